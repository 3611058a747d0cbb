#!/bin/bash
# round 6: bench lines of the four workloads on the in-tree build + where the drop-in generator's time goes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06d_tests.txt 2>&1; tail -3 gpurun_out/r06d_tests.txt
one() { local label=$1; shift
  r=$(python bench.py --no-cpu "$@" 2>>gpurun_out/r06d_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_launch_ms'), d.get('dropin_frames_per_s'))")
  echo "$label: $r"; }
for rep in 1 2; do
  one config3 --no-others --sustain 0
  one config4 --workload config4 --steps 5 --warmup 1
  one fm --workload fm
  one config2 --workload config2
done
SAU_AMD_DEBUG_CREATE=1 python tests/tools/gpu_dropin_phases.py 2>&1 | tail -30
