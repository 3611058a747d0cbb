#!/bin/bash
# the next step's record loaded a step early (FK_PREFETCH=1; round 3 measured it 6 % slower on the forms that spilled scalars)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() { local label=$1 lib=$2; shift 2
  r=$(env ${lib:+SAU_AMD_LIB=$lib} python bench.py --no-cpu "$@" 2>>gpurun_out/r06s_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))")
  echo "$label: $r"; }
V=$GRAFT_REPO_ROOT/saugns_amd/variants
for rep in 1 2; do
  one "fm base" "" --workload fm --steps 30 --warmup 3
  one "fm prefetch" $V/lib_pf.so --workload fm --steps 30 --warmup 3
  one "c4 base" "" --workload config4 --steps 10 --warmup 2
  one "c4 prefetch" $V/lib_pf.so --workload config4 --steps 10 --warmup 2
  one "c3 base" "" --no-others --no-dropin --sustain 0 --steps 100 --warmup 5
  one "c3 prefetch" $V/lib_pf.so --no-others --no-dropin --sustain 0 --steps 100 --warmup 5
done
