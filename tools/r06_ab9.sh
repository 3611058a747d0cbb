#!/bin/bash
# round 6: mixer in 64-frame blocks behind a launch that mixes; R feedback rows numbered kind by kind; drop-in first run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/r06j_tests.txt 2>&1; tail -3 gpurun_out/r06j_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06j_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('kernel_ms_per_step'), d.get('dropin_frames_per_s'), d.get('dropin_first_call_ms'))")
  echo "$label: $r"; }
for rep in 1 2 3; do
  one "config3 mix64" -- --no-others --no-dropin --sustain 0
  one "config3 mix256" SAU_AMD_NO_MIX64=1 -- --no-others --no-dropin --sustain 0
  one "config2 mix64" -- --workload config2
  one "config2 mix256" SAU_AMD_NO_MIX64=1 -- --workload config2
done
one "config3 + dropin" -- --no-others --sustain 0
echo "--- R feedback, rows kind by kind"; python tests/tools/gpu_r_feedback_timing.py 2>&1 | cut -c1-200
echo "--- R feedback, rows in voice order"; SAU_AMD_TUNE=1 SAU_AMD_NO_CHAIN_SORT=1 python tests/tools/gpu_r_feedback_timing.py 2>&1 | cut -c1-200
