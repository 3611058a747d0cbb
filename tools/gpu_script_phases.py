"""Where the wall time of a corpus render through the drop-in API goes: create, first call, the other
calls, destroy -- per script and summed (tools/gpu_corpus_wall.py measures the total)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
G = os.path.join(ROOT, "tests", "golden")
index = json.load(open(os.path.join(G, "index.json")))
sa.set_piluts(np.fromfile(os.path.join(G, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
rate = int(os.environ.get("RATE", "44100"))
progs = {k: sa.Program.from_image(open(os.path.join(G, "programs", k + ".saup"), "rb").read())
         for k in sorted(index["corpus"])}
buf = np.zeros(2 * 11289, np.int16)
for rep in range(3):
    rows = []
    for key, prg in progs.items():
        t0 = time.perf_counter()
        g = sa.Generator(prg, rate)
        t1 = time.perf_counter()
        more, n = g.run(buf, 11289, True)
        t2 = time.perf_counter()
        calls = 1
        while more:
            more, n = g.run(buf, 11289, True)
            calls += 1
        t3 = time.perf_counter()
        g.close()
        t4 = time.perf_counter()
        rows.append(((t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, calls, key))
    tot = [sum(r[i] for r in rows) for i in range(5)]
    print(f"pass {rep}: wall {tot[0]:.1f} ms = create {tot[1]:.1f} + first call {tot[2]:.1f} + other calls {tot[3]:.1f} + destroy {tot[4]:.1f}"
          f" ({sum(r[5] for r in rows)} calls)")
rows.sort(reverse=True)
for r in rows[:12] + rows[-4:]:
    print(f"wall {r[0]:7.2f} = create {r[1]:6.2f} + first {r[2]:6.2f} + rest {r[3]:7.2f} + destroy {r[4]:5.2f}  calls {r[5]:4d}  {r[6]}")
