#!/bin/bash
# same-box A/B of self-modulation workloads: tools/scratch/ab_c5.sh name1 name2 ("cur" = in-tree build)
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = cur ]; then lib=""; else lib="$PWD/saugns_amd/variants/lib_$v.so"; fi
    echo "$v: $(SAU_AMD_LIB=$lib python tools/gpu_sweep.py c5 2>&1 | tail -1)"
    echo "$v: $(SAU_AMD_LIB=$lib python tools/gpu_corpus_paths.py 2>&1 | grep -i "bass-sounds\|pm_feedback_pm\|total" | tr '\n' ';')"
  done
done
