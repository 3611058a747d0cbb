#!/bin/bash
# Alternate bench runs of library variants on one box: tools/ab_run.sh name1 name2 ... ("cur" = in-tree build)
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = cur ]; then lib=""; else lib="$PWD/saugns_amd/variants/lib_$v.so"; fi
    r=$(SAU_AMD_LIB=$lib python bench.py --no-cpu --no-others --no-dropin --sustain 0 $AB_ARGS 2>&1 | tail -1 | grep -o "avg_launch_ms[^,]*\|\"ms_per_step\": [0-9.]*" | tr '\n' ' ')
    echo "$v: $r"
  done
done
