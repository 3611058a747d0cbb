#!/bin/bash
# rint64's short form and franssgauss32's f32 scaling: config 4, R feedback, the KAT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_units.py -x -q -m gpu -k "rint64 or feedback or r_osc or ras or look_back or noise" 2>&1 | tail -3
for rep in 1 2 3; do
  python bench.py --no-cpu --workload config4 --steps 10 --warmup 2 2>>gpurun_out/r06q_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
done
python tests/tools/gpu_r_feedback_timing.py 2>&1 | tail -5 | cut -c1-100
python tests/tools/gpu_r_feedback_kinds.py 1024 1,7,13,5,2 2>&1 | tail -5 | cut -c1-170
