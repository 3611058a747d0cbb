#!/bin/bash
cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["config"].get("ordered_two_generators"), j["roofline"].get("all_kernels_ms_per_step"))'
for i in 1 2 3; do python bench.py --workload config5 --steps 8 --no-cpu 2>&1 | tail -1 | python -c "$J" c5; SAU_AMD_TUNE=1 SAU_AMD_NO_SHORT_LAST_CHUNK=1 python bench.py --workload config5 --steps 8 --no-cpu --c5-serial 2>&1 | tail -1 | python -c "$J" c5_long_last; done
AB_ARGS="--steps 60" tools/ab_run.sh cur zmin r04end
python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -m gpu -x -q -k "feedback or chain or config5" 2>&1 | tail -3
