#!/bin/bash
# round 6: the look-back launch mixes few-voice streams itself (tailmix) -- GPU tests, then config 4 with and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06e_tests.txt 2>&1; tail -3 gpurun_out/r06e_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06e_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_launch_ms'), d['roofline'].get('other_kernels_ms_per_step'), d['roofline'].get('kernel_ms_per_step'))")
  echo "$label: $r"; }
for rep in 1 2 3; do
  one "config4 tailmix" -- --workload config4 --steps 5 --warmup 1
  one "config4 mix_few alone" SAU_AMD_NO_TAILMIX=1 -- --workload config4 --steps 5 --warmup 1
done
one fm -- --workload fm
one config3 -- --no-others --no-dropin --sustain 0
python tests/tools/debug_tailmix.py 2>&1 | tail -12
