"""Config 5's pipeline from a rocprofv3 --kernel-trace CSV: the last step's chain_kernel and pass launches in start order,
start and end relative to the step's first kernel (ms): who waits for whom.  python tools/c5_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sauhip" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void sauhip::", "").replace("sauhip::", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "analyze_kernel"]
lo = starts[-1]
t0 = rows[lo][0]
busy = {}
for s, e, n in rows[lo:]:
    print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f}  {(e - s) / 1e6:7.3f} ms  {n}")
    busy[n] = busy.get(n, 0) + (e - s) / 1e6
print({k: round(v, 3) for k, v in busy.items()})
