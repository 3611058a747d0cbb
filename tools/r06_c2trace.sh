#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2trace -o c2 -- python3 bench.py --no-cpu --workload config2 --steps 30 --warmup 3 > /dev/null 2>&1
f=$(find gpurun_out/c2trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:75].ljust(75), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(12), r['AverageNs'][:10].rjust(11))
PY
