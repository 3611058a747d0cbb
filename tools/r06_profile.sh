#!/bin/bash
# Round 6's final measurements: tools/profile_round.sh (bench lines, kernel traces, PMC passes of every workload) + config 4's
# instruction cache and wait counters + the drop-in settings
#   gpurun --timeout 3000 -- 'tools/r06_profile.sh r06f'
TAG=${1:-r06f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
tail -12 gpurun_out/profile_round_$TAG.log
rocprofv3 -L 2>/dev/null | grep -i "icache\|SQC_INST" | head -40 > gpurun_out/counters_icache_$TAG.txt
timeout 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_icache -o c -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_${TAG}_c4_icache.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_wait -o w -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_${TAG}_c4_wait.log 2>&1
timeout 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_icache -o c -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > gpurun_out/pmc_${TAG}_icache.log 2>&1
python tests/tools/gpu_dropin_settings.py > gpurun_out/dropin_settings_$TAG.txt 2>&1; head -4 gpurun_out/dropin_settings_$TAG.txt
python - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("c4_icache", "c4_wait", "icache"):
    for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, d), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "fast_kernel" in k or "duo" in k:
                print(d, k, {c: sum(x) / len(x) for c, x in v.items()})
PY
