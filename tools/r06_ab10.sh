#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "feedback or chain or config5 or r_oscillator or random" > gpurun_out/r06k_tests.txt 2>&1; tail -3 gpurun_out/r06k_tests.txt
echo "--- R feedback, kinds on waves of their own"; python tests/tools/gpu_r_feedback_timing.py 2>&1 | cut -c1-120
python tests/tools/gpu_vs_ref_chain_banks.py 4100 40 2>&1 | tail -2
