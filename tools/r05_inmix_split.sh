cd "$GRAFT_REPO_ROOT"
J='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % j["value"], "%.3f ms/step" % j["ms_per_step"], "launch %.3f" % j["roofline"]["avg_launch_ms"])'
A="--steps 40 --warmup 3 --no-cpu --no-others --no-dropin --sustain 0 --frames 440999"
for rep in 1 2; do
SAU_AMD_TUNE=1 SAU_AMD_NO_INMIX=1 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" queues_mixer_alone
SAU_AMD_TUNE=1 SAU_AMD_INMIX_DRY=1 python bench.py $A 2>/dev/null | tail -1 | python -c "$J" dry
python bench.py $A 2>/dev/null | tail -1 | python -c "$J" inmix
for v in im_samerow im_noload; do
SAU_AMD_LIB=$GRAFT_REPO_ROOT/saugns_amd/variants/lib_$v.so python bench.py $A 2>/dev/null | tail -1 | python -c "$J" $v
done
done
