#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python tools/gpu_expiry_grid.py 8192 2048 512 1 2>&1 | tail -40
for rep in 1 2; do
  python bench.py --no-cpu --workload config2 --steps 40 --warmup 3 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
done
python bench.py --no-cpu --no-others --steps 50 --warmup 3 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config3', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
