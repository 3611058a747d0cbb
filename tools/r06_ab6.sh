#!/bin/bash
# round 6: analyze_kernel on LDS copies, finalize_kernel's lattice walk, config 5's chunk length
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06g_tests.txt 2>&1; tail -3 gpurun_out/r06g_tests.txt
one() { local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env SAU_AMD_TUNE=1 "${envs[@]}" python bench.py --no-cpu "$@" 2>>gpurun_out/r06g_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('other_kernels_ms_per_step'), d.get('dropin_frames_per_s'))")
  echo "$label: $r"; }
for rep in 1 2; do
  one "config3" -- --no-others --sustain 0
  one "config3 analyze on HBM" SAU_AMD_ANALYZE_NO_LDS=1 -- --no-others --no-dropin --sustain 0
  one "config4" -- --workload config4 --steps 5 --warmup 1
  one "config4 analyze on HBM" SAU_AMD_ANALYZE_NO_LDS=1 -- --workload config4 --steps 5 --warmup 1
  one "config2" -- --workload config2
  one "fm" -- --workload fm
done
for cf in 16384 8192 4096 32768; do
  one "config5 chunk $cf" SAU_AMD_CHAIN_CHUNK_FRAMES=$cf -- --workload config5
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06g_c4 -o r06g_c4 -- python3 bench.py --workload config4 --steps 3 --no-cpu > gpurun_out/prof_r06g_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06g -o r06g -- python3 bench.py --no-cpu --no-others --no-dropin --sustain 0 --steps 20 > gpurun_out/prof_r06g.log 2>&1
for f in gpurun_out/prof_r06g_c4/*kernel_stats.csv gpurun_out/prof_r06g/*kernel_stats.csv; do echo $f; cut -d, -f1-4 $f | head -12; done
