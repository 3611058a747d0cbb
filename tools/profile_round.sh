#!/bin/bash
# One round's measurements on the GPU box, into gpurun_out/ (then: python tools/collect_profile.py <tag> <name>):
#   gpurun --timeout 1800 -- 'tools/profile_round.sh r02f'
# Bench lines of the three workloads, rocprofv3 kernel trace + stats of the same commands, and PMC passes
# of the headline command (one counter group per run, kernel trace only: MI355X_MICROARCH.md, HBM /
# rocprofv3 section).
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python bench.py --no-others > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python bench.py --workload config5 > gpurun_out/bench_${TAG}_c5.json 2> gpurun_out/bench_${TAG}_c5.err
python bench.py --workload config4 > gpurun_out/bench_${TAG}_c4.json 2> gpurun_out/bench_${TAG}_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu --no-others > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c5 -o ${TAG}_c5 -- python3 bench.py --workload config5 --steps 3 --no-cpu > gpurun_out/prof_${TAG}_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c4 -o ${TAG}_c4 -- python3 bench.py --workload config4 --steps 3 --no-cpu > gpurun_out/prof_${TAG}_c4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_write -o w -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq -o s -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_inst -o i -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_grbm -o g -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
# the vector instructions by class (round 5: priced with tools/valu_probe.hip's costs by tools/collect_profile.py -> roofline.valu)
CLS_A="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
CLS_B="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
rocprofv3 --pmc $CLS_A --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_cls_a -o a -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
rocprofv3 --pmc $CLS_B --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_cls_b -o b -- python3 bench.py --steps 4 --warmup 1 --sustain 0 --no-cpu --no-others --no-dropin > /dev/null 2>&1
# config 5: HBM bytes and instruction mix of the chain kernel and the passes around it
# (one counter per pass: FETCH_SIZE and WRITE_SIZE together never finished on this pool -- r02a lost 25 minutes to it)
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c5_fetch -o m -- python3 bench.py --workload config5 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c5_write -o m -- python3 bench.py --workload config5 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c5_inst -o i -- python3 bench.py --workload config5 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
# config 4 likewise (HBM bytes per step of the two launches over the voice lists and the few-row mixer)
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_fetch -o m -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_write -o m -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_inst -o i -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
# the two banks on the driver's line since round 4 (carrier-FM bank, config 2): HBM bytes of a step and the instruction mix
for W in fm config2; do D=$W; [ $W = fm ] && D=fmbank   # (pmc_<tag>_fm_inst is the sweep's, below)
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${D}_fetch -o m -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${D}_write -o m -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${D}_inst -o i -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc $CLS_A --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${D}_cls_a -o a -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc $CLS_B --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_${D}_cls_b -o b -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
done
# config 4 likewise: launch cycles and the class census of its two time-parallel launches
timeout 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_grbm -o g -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc $CLS_A --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_cls_a -o a -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
timeout 180 rocprofv3 --pmc $CLS_B --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c4_cls_b -o b -- python3 bench.py --workload config4 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
# running-sum workloads (carrier FM bank, carrier glide, FM + ratio PM stack): sweep lines, kernel trace, instruction mix
TLEVEL=0 timeout 180 python3 tests/tools/gpu_sweep.py c3f fmstack mixed > gpurun_out/sweep_${TAG}_fm.txt 2>&1
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_fm -o ${TAG}_fm -- python3 tests/tools/gpu_sweep.py c3f fmstack > gpurun_out/prof_${TAG}_fm.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fm_inst -o i -- python3 tests/tools/gpu_sweep.py c3f > /dev/null 2>&1
python bench.py > gpurun_out/bench_${TAG}_full.json 2> gpurun_out/bench_${TAG}_full.err   # the driver's command: all three workloads in one line
for f in "" _c5 _c4; do tail -1 gpurun_out/bench_$TAG$f.json | cut -c1-200; done
cat gpurun_out/sweep_${TAG}_fm.txt
