#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06l_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('kernel_ms_per_step'))")
  echo "$label: $r"; }
for rep in 1 2; do
  for pr in 1 2 3 4; do one "config4 duo prio-setting $pr" SAU_AMD_DUO_PRIO=$pr -- --workload config4 --steps 5 --warmup 1; done
done
