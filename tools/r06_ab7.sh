#!/bin/bash
# round 6: closed-form and look-back voices in one launch (duo_kernel); analyze_kernel with byte tables; finalize_kernel's bulk walk
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "config4 or look or running_sum or random or corpus" > gpurun_out/r06h_tests.txt 2>&1; tail -3 gpurun_out/r06h_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06h_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('kernel_ms_per_step'), r.get('other_kernels_ms_per_step'), d.get('dropin_frames_per_s'))")
  echo "$label: $r"; }
for rep in 1 2; do
  one "config4 duo" -- --workload config4 --steps 5 --warmup 1
  one "config4 two launches" SAU_AMD_NO_DUO=1 -- --workload config4 --steps 5 --warmup 1
done
one "config3" -- --no-others --sustain 0
one "config2" -- --workload config2
one "fm" -- --workload fm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06h_c4 -o r06h_c4 -- python3 bench.py --workload config4 --steps 3 --no-cpu > gpurun_out/prof_r06h_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06h -o r06h -- python3 bench.py --no-cpu --no-others --no-dropin --sustain 0 --steps 20 > gpurun_out/prof_r06h.log 2>&1
python - <<'PY'
import csv
for f in ('gpurun_out/prof_r06h_c4/r06h_c4_kernel_stats.csv','gpurun_out/prof_r06h/r06h_kernel_stats.csv'):
    print(f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-70s calls %4s avg %10.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
