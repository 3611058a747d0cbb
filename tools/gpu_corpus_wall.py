"""Wall-clock cost of rendering each corpus script through the drop-in API (create .. destroy,
timing off), against the kernels' own time: what the host side adds per segment."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
G = os.path.join(ROOT, "tests", "golden")
index = json.load(open(os.path.join(G, "index.json")))
sa.set_piluts(np.fromfile(os.path.join(G, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
rate = int(os.environ.get("RATE", "44100"))
rows = []
progs = {k: sa.Program.from_image(open(os.path.join(G, "programs", k + ".saup"), "rb").read())
         for k in sorted(index["corpus"])}
for rep in range(2):
    rows = []
    for key, prg in progs.items():
        t0 = time.perf_counter()
        g = sa.Generator(prg, rate)
        pcm = g.render(stereo=True, chunk=11289)
        g.close()
        wall = time.perf_counter() - t0
        b = sa.Batch([prg], rate); b.set_timing(2); b.render(stereo=True, chunk=176400)
        t = b.timing_ex()
        rows.append((wall * 1e3, t["block_ms"] + t["fast_ms"] + t["mix_ms"] + t["aux_ms"], t["segments"], len(pcm) // 2, key))
rows.sort(reverse=True)
print(f"rate {rate}: wall {sum(r[0] for r in rows):.1f} ms, kernels {sum(r[1] for r in rows):.1f} ms, "
      f"segments {sum(r[2] for r in rows)}, frames {sum(r[3] for r in rows)} over {len(rows)} scripts")
for r in rows[:14]:
    print(f"wall {r[0]:8.2f} ms  kernels {r[1]:8.2f} ms  segments {r[2]:5d}  frames {r[3]:9d}  {r[4]}")
