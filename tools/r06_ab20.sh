#!/bin/bash
# rchain_kernel's feeders: increments two batches ahead (rate feeder), counter + split + amount + drain (amount feeder)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests -x -q -m gpu -k "feedback or chain or r_osc or ras" 2>&1 | tail -3
timeout 300 python tests/tools/gpu_r_feedback_kinds.py 1024 0,3,2,4,1,7,5 2>&1 | tail -7 | cut -c1-175
timeout 300 python tests/tools/gpu_r_feedback_timing.py 2>&1 | tail -5 | cut -c1-100
