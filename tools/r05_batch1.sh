#!/bin/bash
cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["config"].get("serial"), j["roofline"].get("all_kernels_ms_per_step"))'
for i in 1 2; do python bench.py --workload config5 --steps 8 --no-cpu 2>&1 | tail -1 | python -c "$J" c5; done
for t in 16 8 16 8; do SAU_AMD_TUNE=1 SAU_AMD_NO_CHAIN=1 SAU_AMD_MULTI_TEAMS=$t python bench.py --workload config5 --steps 2 --warmup 1 --no-cpu --c5-serial 2>&1 | tail -1 | python -c "$J" blockloop_teams$t; done
tools/pmc_classes.sh r05fm --workload fm --steps 2 --warmup 1 --no-cpu 2>&1 | tail -4
tools/pmc_classes.sh r05c4 --workload config4 --steps 1 --warmup 0 --no-cpu 2>&1 | tail -6
