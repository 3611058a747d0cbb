#!/bin/bash
# round 6: short last chunks of the in-launch mixing, the drop-in generator's run lengths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06f_tests.txt 2>&1; tail -3 gpurun_out/r06f_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06f_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('other_kernels_ms_per_step'), d.get('dropin_frames_per_s'))")
  echo "$label: $r"; }
for rep in 1 2 3; do
  one "config3 taper" -- --no-others --no-dropin --sustain 0
  one "config3 no taper" SAU_AMD_NO_INMIX_TAPER=1 -- --no-others --no-dropin --sustain 0
  one "config2 taper" -- --workload config2
  one "config2 no taper" SAU_AMD_NO_INMIX_TAPER=1 -- --workload config2
done
one "config3 + dropin" -- --no-others --sustain 0
python tests/tools/gpu_dropin_settings.py 2>&1 | head -3
python tests/tools/gpu_dropin_phases.py 2>&1 | tail -3
