cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f ms/step" % j["ms_per_step"], "launch %.3f" % j["roofline"]["avg_launch_ms"])'
A="--steps 60 --warmup 3 --no-cpu --no-others --no-dropin --sustain 0"
for rep in 1 2 3 4; do
for at in 12 13; do
SAU_AMD_TUNE=1 SAU_AMD_INMIX_AT=$at python bench.py $A 2>/dev/null | tail -1 | python -c "$J" inmix_at$at
done
SAU_AMD_TUNE=1 SAU_AMD_NO_INMIX=1 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" queues_alone
done
