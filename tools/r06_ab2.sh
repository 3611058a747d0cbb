#!/bin/bash
# round 6: the two-form group evaluation (FK_SPLIT_MASK): all four builds (in-tree), look-back only (8), none (0) -- same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "config4 or look or running_sum" > gpurun_out/r06c_tests.txt 2>&1; tail -3 gpurun_out/r06c_tests.txt
(python saugns_amd/build.py --variant split8 FK_SPLIT_MASK=8 > gpurun_out/r06c_build8.txt 2>&1) &
(sleep 5; python saugns_amd/build.py --variant split0 FK_SPLIT_MASK=0 > gpurun_out/r06c_build0.txt 2>&1) &
wait
ls -la saugns_amd/variants/
one() { # label, lib, bench args
  local label=$1 lib=$2; shift; shift
  r=$(env SAU_AMD_LIB=$lib python bench.py --no-cpu "$@" 2>>gpurun_out/r06c_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_launch_ms'))")
  echo "$label: $r"
}
V8=$PWD/saugns_amd/variants/lib_split8.so; V0=$PWD/saugns_amd/variants/lib_split0.so
for rep in 1 2 3; do
  for v in cur:"" s8:$V8 s0:$V0; do
    n=${v%%:*}; lib=${v#*:}
    one "config3 $n" "$lib" --no-others --no-dropin --sustain 0
    one "config4 $n" "$lib" --workload config4 --steps 5 --warmup 1
    one "fm $n" "$lib" --workload fm
    one "config2 $n" "$lib" --workload config2
  done
done
