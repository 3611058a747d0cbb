#!/bin/bash
# Round 6's device-vs-compiled-reference sweeps on the final state (as rounds 3-5 did), several processes side by side
# (the reference renders on the host cores; each process has its own generators on the one GPU).
#   gpurun --timeout 2400 -- 'tools/r06_sweeps.sh <tag> <programs per job>'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; N=${2:-3000}; O=${3:-0}
run() { name=$1; shift; SWEEP_OUT=gpurun_out/r06_gpu_vs_ref_${TAG}_$name.json python tests/tools/gpu_vs_ref_sweep.py "$@" > gpurun_out/r06_sweep_${TAG}_$name.log 2>&1 & }
run plain_a $((3000000 + O)) $N
run plain_b $((3100000 + O)) $N
run extreme_a $((3200000 + O)) $N extreme
run extreme_b $((3300000 + O)) $N extreme
run tall $((3400000 + O)) $((N / 4)) tall
run dropin $((3500000 + O)) $N dropin
run varied_a $((3600000 + O)) $N varied
run dropin_b $((3550000 + O)) $N dropin
run dropin_c $((3570000 + O)) $N dropin
run varied_b $((3700000 + O)) $N varied
python tests/tools/gpu_vs_ref_batches.py $((3800000 + O)) $((N / 12)) > gpurun_out/r06_sweep_${TAG}_batches.log 2>&1 &
python tests/tools/gpu_vs_ref_chain_banks.py $((3900 + O / 1000)) 30 > gpurun_out/r06_sweep_${TAG}_chain_banks.log 2>&1 &
wait
cp gpurun_out/gpu_vs_ref_batches.json gpurun_out/r06_gpu_vs_ref_${TAG}_batches.json 2>/dev/null
for f in gpurun_out/r06_sweep_${TAG}_*.log; do echo "== $f"; tail -2 $f | cut -c1-300; done
