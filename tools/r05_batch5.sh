#!/bin/bash
cd "$GRAFT_REPO_ROOT"
AB_ARGS="--steps 60" tools/ab_run.sh cur zmin
python tests/tools/gpu_r_feedback_timing.py 2>&1 | tail -5
python -m pytest tests -m gpu -q 2>&1 | tail -6
