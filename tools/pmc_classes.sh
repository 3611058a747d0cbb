#!/bin/bash
# Dynamic instruction-class census and stall counters of one workload's kernels (round 5):
#   gpurun --timeout 900 -- 'tools/pmc_classes.sh <tag> [bench.py args, default: config 3]'   -> gpurun_out/pmcc_<tag>_{a,b,c,d}/ + pmcc_<tag>.json
# SQ_INSTS_VALU_* split the vector instructions by class (the classes tools/valu_probe.hip prices); one counter group per run,
# kernel trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
TAG=$1; shift
ARGS=${@:---steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmcc_${TAG}_$d -o $d -- python3 bench.py $ARGS > gpurun_out/pmcc_${TAG}_$d.log 2>&1; }
run a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
run b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run d SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(f"gpurun_out/pmcc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sauhip" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void sauhip::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]), workgroup=int(r["Workgroup_Size"]), grid=int(r["Grid_Size"]))
out = {k: dict(meta[k], launches=max(len(x) for x in c.values()), **{n: sum(x) / len(x) for n, x in sorted(c.items())}) for k, c in agg.items()}
json.dump(out, open(f"gpurun_out/pmcc_{tag}.json", "w"), indent=1)
for k, v in out.items():
    if v.get("SQ_INSTS_VALU", 0) > 1e6:
        t = v["SQ_INSTS_VALU"]
        print(k, {n: round(x / t, 4) for n, x in v.items() if n.startswith("SQ_INSTS")}, flush=True)
PY
