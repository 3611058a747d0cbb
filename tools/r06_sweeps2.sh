#!/bin/bash
# the sweeps of tools/r06_sweeps.sh plus the random closed-form banks (the launch that mixes; config 3's two-launch build) and more chain banks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; N=${2:-2500}; O=${3:-500000}
tools/r06_sweeps.sh $TAG $N $O
python tests/tools/gpu_vs_ref_inmix_banks.py $((O + 77)) 300 > gpurun_out/r06_sweep_${TAG}_inmix_banks.log 2>&1
tail -1 gpurun_out/r06_sweep_${TAG}_inmix_banks.log > gpurun_out/r06_gpu_vs_ref_${TAG}_inmix_banks.json
tail -3 gpurun_out/r06_sweep_${TAG}_inmix_banks.log | cut -c1-300
python tests/tools/gpu_vs_ref_chain_banks.py $((4900 + O / 1000)) 60 > gpurun_out/r06_sweep_${TAG}_chain_banks2.log 2>&1
tail -2 gpurun_out/r06_sweep_${TAG}_chain_banks2.log | cut -c1-300
