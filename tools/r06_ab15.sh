#!/bin/bash
# config 3's 12-row build as two launches (edges | inner groups in the form without in-segment masks): SAU_AMD_NO_INNER=1 is the one launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_inmix.py -x -q -m gpu 2>&1 | tail -3
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06n_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])")
  echo "$label: $r"; }
for rep in 1 2 3; do
  one "config3 inner" -- --no-others --no-dropin --sustain 0 --steps 60 --warmup 3
  one "config3 one launch" SAU_AMD_NO_INNER=1 -- --no-others --no-dropin --sustain 0 --steps 60 --warmup 3
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3inner2 -o c3 -- python3 bench.py --no-cpu --no-others --no-dropin --sustain 0 --steps 30 --warmup 3 > /dev/null 2>&1
f=$(find gpurun_out/c3inner2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(r['Name'][:75].ljust(75), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(12), r['AverageNs'][:10].rjust(11))
PY
