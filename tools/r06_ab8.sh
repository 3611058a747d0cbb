#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q -k "config4" > gpurun_out/r06i_tests.txt 2>&1; tail -2 gpurun_out/r06i_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06i_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('kernel_ms_per_step'))")
  echo "$label: $r"; }
for rep in 1 2; do
  one "config4 two launches" SAU_AMD_NO_DUO=1 -- --workload config4 --steps 5 --warmup 1
  for lw in 5 6 7 8 9; do one "config4 duo lw $lw" SAU_AMD_DUO_LW=$lw -- --workload config4 --steps 5 --warmup 1; done
done
