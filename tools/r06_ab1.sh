#!/bin/bash
# round 6, first measurement: GPU tests, then same-box A/Bs of the look-back placement and the task length
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_tests.txt 2>&1; tail -3 gpurun_out/r06a_tests.txt
one() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>gpurun_out/r06a_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('first_step_sha_ok', d.get('config', {}).get('verified')))")
  echo "$label: $r"
}
for rep in 1 2; do
  one "config4 spread" -- --workload config4 --steps 5 --warmup 1
  one "config4 round-5 placement" SAU_AMD_LOOK_NO_SPREAD=1 -- --workload config4 --steps 5 --warmup 1
done
for rep in 1 2; do
  one "fm" -- --workload fm
  one "config2" -- --workload config2
  one "config2 groups 6" SAU_AMD_DYN_GROUPS=6 -- --workload config2
  one "config3" -- --no-others --no-dropin --sustain 0
  one "config3 groups 6" SAU_AMD_DYN_GROUPS=6 -- --no-others --no-dropin --sustain 0
  one "config3 groups 8" SAU_AMD_DYN_GROUPS=8 -- --no-others --no-dropin --sustain 0
done
one "config5" -- --workload config5
