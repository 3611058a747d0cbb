#!/bin/bash
# the driver's commands on the committed tree: the GPU suite, smoke(), the default bench line (twice)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r06_final_bench_1.json 2> gpurun_out/r06_final_bench_1.err; tail -c 600 gpurun_out/r06_final_bench_1.json
python bench.py > gpurun_out/r06_final_bench_2.json 2> gpurun_out/r06_final_bench_2.err
python - <<'PY'
import json
for i in (1, 2):
    j = json.loads(open(f"gpurun_out/r06_final_bench_{i}.json").read().strip().splitlines()[-1])
    print(i, {k: j.get(k) for k in ("value", "ms_per_step", "valu_frac", "dropin_frames_per_s", "dropin_first_call_ms", "config4_value", "fm_value", "config2_value", "config5_value", "sustained_frames_per_s", "cpu_1core_value", "cpu_all_cores_value")})
    print("  roofline", {k: j["roofline"].get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "avg_launch_ms")}, "cpu", j.get("cpu_baseline", {}).get("value"))
PY
