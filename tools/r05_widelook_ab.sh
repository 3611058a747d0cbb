#!/bin/bash
# round 5: the look-back build with wide table blocks (where the lean buffer numbering leaves LDS for them) against the narrow one
cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f ms/step" % j["ms_per_step"], "kernel %.3f" % j["roofline"].get("kernel_ms_per_step", 0), j["config"].get("first_step_verified") is not None)'
for rep in 1 2 3; do
SAU_AMD_TUNE=1 SAU_AMD_NO_WIDE_LOOK=1 python bench.py --workload fm --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" fm_narrow
python bench.py --workload fm --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" fm_wide
done
