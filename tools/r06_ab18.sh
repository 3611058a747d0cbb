#!/bin/bash
# look-back polls: how long an empty poll sleeps (LDS rings: FM bank; words in HBM: config 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() { local label=$1 lib=$2; shift 2
  r=$(env ${lib:+SAU_AMD_LIB=$lib} python bench.py --no-cpu "$@" 2>>gpurun_out/r06r_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))")
  echo "$label: $r"; }
V=$GRAFT_REPO_ROOT/saugns_amd/variants
for rep in 1 2; do
  one "fm base" "" --workload fm --steps 30 --warmup 3
  one "fm lds-sleep 0" $V/lib_l0.so --workload fm --steps 30 --warmup 3
  one "fm lds-sleep 3" $V/lib_l3.so --workload fm --steps 30 --warmup 3
  one "c4 base" "" --workload config4 --steps 10 --warmup 2
  one "c4 hbm-sleep 1" $V/lib_h1.so --workload config4 --steps 10 --warmup 2
  one "c4 hbm-sleep 5" $V/lib_h5.so --workload config4 --steps 10 --warmup 2
done
