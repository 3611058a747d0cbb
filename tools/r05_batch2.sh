#!/bin/bash
cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["config"].get("serial"), j["roofline"].get("all_kernels_ms_per_step"))'
for i in 1 2 3; do python bench.py --workload config5 --steps 8 --no-cpu 2>&1 | tail -1 | python -c "$J" c5; done
python tests/tools/gpu_r_feedback_timing.py 2>&1 | tail -6
python -m pytest tests/test_gpu_units.py -m gpu -x -q -k "r_oscillator or ordered_one_behind or feedback or chain" 2>&1 | tail -5
