cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4dbg -o c4 --output-format csv -- python3 bench.py --workload config4 --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_c4dbg.log 2>&1
head -14 $R/gpurun_out/prof_c4dbg/c4_kernel_stats.csv | cut -c1-150
grep -o '"ms_per_step": [0-9.]*\|"segments_per_step": [0-9.]*\|"kernel_ms_per_step": [0-9.]*\|"other_kernels_ms_per_step": {[^}]*}' $R/gpurun_out/prof_c4dbg.log
