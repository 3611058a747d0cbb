#!/bin/bash
# All bench workloads of library variants on one box, alternating: tools/ab_all.sh name1 name2 ... ("cur" = in-tree build)
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = cur ]; then lib=""; else lib="$PWD/saugns_amd/variants/lib_$v.so"; fi
    for w in config3 fm config2 config4 config5; do
      r=$(SAU_AMD_LIB=$lib python bench.py --no-cpu --no-others --no-dropin --sustain 0 --workload $w $AB_ARGS 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1)
      echo "$v $w: $r"
    done
  done
done
