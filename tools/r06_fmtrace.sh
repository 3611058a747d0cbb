#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fmtrace -o fm -- python3 bench.py --no-cpu --workload fm --steps 20 --warmup 3 > /dev/null 2>&1
f=$(find gpurun_out/fmtrace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r['Name'][:75].ljust(75), r['Calls'].rjust(5), r['TotalDurationNs'].rjust(12), r['AverageNs'][:10].rjust(11))
PY
python bench.py --no-cpu --workload fm --steps 40 --warmup 3 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fm', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
