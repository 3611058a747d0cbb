#!/bin/bash
# config 4: kernel trace + one PMC pass (instruction counts, wait cycles) of a step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06b}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c4 -o ${TAG}_c4 -- python3 bench.py --workload config4 --steps 3 --no-cpu > gpurun_out/prof_${TAG}_c4.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_inst -o i -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_wait -o w -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06b"
for f in glob.glob("gpurun_out/prof_%s_c4/**/*kernel_stats.csv" % tag, recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 14: print(",".join(row[:5]))
for d in ("inst", "wait"):
    for f in glob.glob("gpurun_out/pmc_%s_c4_%s/**/*counter_collection.csv" % (tag, d), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        rd = csv.DictReader(open(f))
        for r in rd:
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "fast_kernel" in k: print(k, dict(v))
PY
