"""Per-step timeline from a rocprofv3 --kernel-trace CSV: kernel durations, gaps between consecutive
kernels and overlaps, averaged over the steps of the run.  python tools/trace_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sauhip" not in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void sauhip::", "").replace("sauhip::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# steps begin at analyze_kernel
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end, prev_name = None, None
for s, e, n, q in rows:
    dur[n].append((e - s) / 1e3)
    if prev_end is not None:
        gap[prev_name + " -> " + n].append((s - prev_end) / 1e3)
    prev_end, prev_name = max(e, prev_end or 0), n
print("kernel durations (us): mean [n]")
for n, v in dur.items():
    v2 = v[len(v) // 4:]
    print(f"  {n:28s} {sum(v2) / len(v2):10.1f} [{len(v)}]")
print("gaps start(next) - end(prev) (us; negative = overlap):")
for n, v in gap.items():
    v2 = v[len(v) // 4:]
    print(f"  {n:50s} {sum(v2) / len(v2):10.1f} [{len(v)}]")
an = [s for s, e, n, q in rows if n == "analyze_kernel"]
if len(an) > 4:
    d = [(b - a) / 1e3 for a, b in zip(an[len(an) // 4:-1], an[len(an) // 4 + 1:])]
    print(f"analyze -> analyze period: {sum(d) / len(d):.1f} us over {len(d)} steps")
