#!/bin/bash
cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f" % j["ms_per_step"], "kernel %.3f" % j["roofline"]["kernel_ms_per_step"])'
for r in 6 5 4 2 6; do SAU_AMD_TUNE=1 SAU_AMD_FAST_ROWS=$r python bench.py --workload fm --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "$J" fm_rows$r; done
for r in 5 4 2; do SAU_AMD_TUNE=1 SAU_AMD_FAST_ROWS=$r python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "$J" c4_rows$r; done
python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "$J" c4_default
