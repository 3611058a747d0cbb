"""Which corpus programs still spend time in the block loop? (timing per kernel kind)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
G = os.path.join(ROOT, "tests", "golden")
index = json.load(open(os.path.join(G, "index.json")))
sa.set_piluts(np.fromfile(os.path.join(G, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
rows = []
for key in sorted(index["corpus"]):
    prg = sa.Program.from_image(open(os.path.join(G, "programs", key + ".saup"), "rb").read())
    b = sa.Batch([prg], index["corpus_rate"])
    b.set_timing(2)
    b.render(stereo=True, chunk=200000)
    t = b.timing_ex()
    rows.append((t["block_ms"], t["fast_ms"], t["segments"], key))
rows.sort(reverse=True)
tb = sum(r[0] for r in rows); tf = sum(r[1] for r in rows)
print(f"total block {tb:.2f} ms fast {tf:.2f} ms over {len(rows)} programs")
for r in rows[:25]:
    print(f"block {r[0]:8.3f} ms  fast {r[1]:8.3f} ms  segments {r[2]:5d}  {r[3]}")
