#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_inmix.py -x -q 2>&1 | tail -5
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f ms/step" % j["ms_per_step"], "launch %.3f" % j["roofline"].get("avg_launch_ms", j["roofline"].get("kernel_ms_per_step", 0)), "sha", j.get("first_step_sha_ok"))'
A="--steps 40 --warmup 3 --no-cpu --no-others --no-dropin --sustain 0"
for rep in 1 2 3; do
SAU_AMD_TUNE=1 SAU_AMD_NO_XCD_QUEUES=1 timeout 300 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" one_counter
timeout 300 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" default_inmix
SAU_AMD_TUNE=1 SAU_AMD_NO_INMIX=1 timeout 300 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" queues_mixer_alone
done
for w in fm config2; do
SAU_AMD_TUNE=1 SAU_AMD_NO_XCD_QUEUES=1 timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" ${w}_one_counter
timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" ${w}_xcd_queues
done
