#!/bin/bash
# config 4 with the steps pipelined two deep (bench.py), against the driver's default full line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  python bench.py --no-cpu --workload config4 --steps 5 --warmup 1 2>>gpurun_out/r06p_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4', d['value'], d['ms_per_step'], d['warmup'], d['roofline'].get('kernel_ms_per_step'))"
done
python bench.py --no-cpu --workload config4 --steps 20 --warmup 2 2>>gpurun_out/r06p_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4 20 steps', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_units.py -x -q -m gpu -k "bench or config4" 2>&1 | tail -2
