#!/bin/bash
# chain kernels' workgroups on CUs of their own (SAU_AMD_CHAIN_ALONE): R feedback banks, config 5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for a in 1 0; do
    echo "== CHAIN_ALONE=$a"
    SAU_AMD_TUNE=1 SAU_AMD_CHAIN_ALONE=$a python tests/tools/gpu_r_feedback_timing.py 2>&1 | tail -5 | cut -c1-90
    SAU_AMD_TUNE=1 SAU_AMD_CHAIN_ALONE=$a python bench.py --no-cpu --workload config5 --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config5', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
  done
done
