#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for n in 1024 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rfb$n -o rfb -- python3 tests/tools/gpu_r_feedback_one.py $n 2>&1 | grep voices
  f=$(find gpurun_out/rfb$n -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(12), r['AverageNs'][:10].rjust(11))
PY
done
