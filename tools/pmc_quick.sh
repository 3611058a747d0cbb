#!/bin/bash
# Instruction mix and LDS activity of the headline kernel for one or more library variants (cur = in-tree):
#   tools/pmc_quick.sh <tag> cur r04base ...   -> gpurun_out/pmcq_<tag>_<variant>_{a,b}/
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = cur ]; then export -n SAU_AMD_LIB; unset SAU_AMD_LIB; else export SAU_AMD_LIB="$PWD/saugns_amd/variants/lib_$v.so"; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcq_${TAG}_${v}_a -o a -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin $PMC_ARGS > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmcq_${TAG}_${v}_b -o b -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin $PMC_ARGS > /dev/null 2>&1
  python3 - "$TAG" "$v" <<'PY'
import csv, collections, glob, sys
tag, v = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmcq_{tag}_{v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_kernel" in r["Kernel_Name"] or "chain_kernel" in r["Kernel_Name"] or "mix_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void sauhip::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(v, k, {n: round(sum(x) / len(x)) for n, x in sorted(c.items())}, flush=True)
PY
done
