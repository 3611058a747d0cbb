#!/bin/bash
# R oscillators in the time-parallel kernels: a row's segment ends kept for the next row while no lane has entered another cycle
# (the working tree's library) against the commit before (saugns_amd/variants/lib_prev.so: tools/ab_variant.sh HEAD prev), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() { local label=$1 lib=$2; shift 2
  r=$(env ${lib:+SAU_AMD_LIB=$lib} python bench.py --no-cpu "$@" 2>>gpurun_out/r06t_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))")
  echo "$label: $r"; }
V=$GRAFT_REPO_ROOT/saugns_amd/variants
for rep in 1 2 3; do
  one "c4 new" "" --workload config4 --steps 10 --warmup 2
  one "c4 prev" $V/lib_prev.so --workload config4 --steps 10 --warmup 2
done
one "fm new" "" --workload fm --steps 30 --warmup 3
one "fm prev" $V/lib_prev.so --workload fm --steps 30 --warmup 3
one "c3 new" "" --no-others --no-dropin --sustain 0 --steps 100 --warmup 5
one "c3 prev" $V/lib_prev.so --no-others --no-dropin --sustain 0 --steps 100 --warmup 5
