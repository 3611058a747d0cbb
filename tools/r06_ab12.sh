#!/bin/bash
# config 2: the closed-form launch's tasks from the queues (and the launch mixing) at smaller tasks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06m_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('kernel_ms_per_step'))")
  echo "$label: $r"; }
for rep in 1 2; do
  one "config2 default" -- --workload config2 --steps 40 --warmup 3
  for fl in 32 24 16 12 8 4; do
    one "config2 floor $fl mt 1 no inmix" SAU_AMD_DYN_FLOOR=$fl SAU_AMD_DYN_MIN_TASKS=1 SAU_AMD_NO_INMIX=1 -- --workload config2 --steps 40 --warmup 3
  done
  one "config2 floor 12 mt 2 no inmix" SAU_AMD_DYN_FLOOR=12 SAU_AMD_DYN_MIN_TASKS=2 SAU_AMD_NO_INMIX=1 -- --workload config2 --steps 40 --warmup 3
  one "config2 floor 12 mt 1 no inmix no xcd queues" SAU_AMD_DYN_FLOOR=12 SAU_AMD_DYN_MIN_TASKS=1 SAU_AMD_NO_INMIX=1 SAU_AMD_NO_XCD_QUEUES=1 -- --workload config2 --steps 40 --warmup 3
done
