#!/bin/bash
# config 5, three steps pipelined with sauAmd_Batch_order_after: who runs when (kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c5trace -o t -- python3 bench.py --workload config5 --steps 3 --warmup 0 --no-cpu > gpurun_out/c5trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c5trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sauhip" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void sauhip::", "").replace("sauhip::", ""), r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
an = [i for i, r in enumerate(rows) if r[2] == "analyze_kernel"]
print("analyze launches at (ms):", [round((rows[i][0] - rows[an[0]][0]) / 1e6, 2) for i in an])
# the last three steps = the pipelined ones: print chain_kernel / mix / analyze / finalize lines with queue ids
lo = an[-3]
t0 = rows[lo][0]
for s, e, n, q in rows[lo:]:
    if n in ("chain_kernel", "analyze_kernel", "finalize_kernel", "mix_kernel", "decode_kernel") or (e - s) > 3e5:
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  q{q}  {n}")
PY
tail -c 400 gpurun_out/c5trace.log
