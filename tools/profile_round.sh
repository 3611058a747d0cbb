#!/bin/bash
# One round's measurements on the GPU box, into gpurun_out/ (then: python tools/collect_profile.py <tag> <name>):
#   gpurun --timeout 1500 -- 'tools/profile_round.sh r01h'
# bench line, rocprofv3 kernel trace + stats of the same command, PMC passes (one counter group per
# run, kernel trace only: MI355X_MICROARCH.md, HBM / rocprofv3 section), and a kernel trace of
# BASELINE config 5 (feedback voices: render_kernel).
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_inst -o i -- python3 bench.py --steps 4 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c5 -o ${TAG}_c5 -- python3 tools/gpu_sweep.py c5 > gpurun_out/prof_${TAG}_c5.log 2>&1
tail -1 gpurun_out/bench_$TAG.json | cut -c1-160
tail -2 gpurun_out/prof_${TAG}_c5.log
