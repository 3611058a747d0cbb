cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f ms/step" % j["ms_per_step"], "kernel %.3f" % j["roofline"].get("kernel_ms_per_step", 0), j.get("config",{}).get("first_step_verified") is not None or j["config"].get("verified"))'
SAU_AMD_TUNE=1 SAU_AMD_DEBUG=1 python bench.py --workload fm --steps 1 --warmup 0 --no-cpu 2>&1 | grep -E "n_fast" | head -2
SAU_AMD_TUNE=1 SAU_AMD_DEBUG=1 python bench.py --workload config4 --steps 1 --warmup 0 --no-cpu 2>&1 | grep -E "n_fast" | sort | uniq -c | head -4
for rep in 1 2; do
SAU_AMD_TUNE=1 SAU_AMD_NO_LEAN_IDS=1 python bench.py --workload fm --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" fm_full_ids
python bench.py --workload fm --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "$J" fm_lean
SAU_AMD_TUNE=1 SAU_AMD_NO_LEAN_IDS=1 python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "$J" c4_full_ids
python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "$J" c4_lean
done
