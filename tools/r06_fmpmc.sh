#!/bin/bash
# the FM bank's launch: what do its waves wait for? (config 3's launch beside it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 -L 2>/dev/null | grep -o "SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_LDS[A-Z_]*" | sort -u | tr '\n' ' '; echo
pass() { local tag=$1; shift; local wl=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/fmpmc_$tag -o p -- python3 bench.py $wl --steps 2 --warmup 1 --no-cpu > gpurun_out/fmpmc_$tag.log 2>&1; }
pass fm_a "--workload fm" SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass fm_b "--workload fm" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR
pass c3_b "--no-others --no-dropin --sustain 0" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR
pass c3_a "--no-others --no-dropin --sustain 0" SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
python - <<'PY'
import csv, glob, collections
for d in ("fm_a", "fm_b", "c3_a", "c3_b"):
    for f in glob.glob("gpurun_out/fmpmc_%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if ("fast_kernel<8, 2" in k) or ("false, true, false, true>" in k):
                print(d, k[-40:], {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
