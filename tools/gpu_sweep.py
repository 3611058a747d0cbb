"""Timing experiments: frames/s for voice banks of different shape (GPU)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
def run(name, prg, frames=44100, steps=8):
    b = sa.Batch([prg], 44100)
    for _ in range(2): b.run(frames, fetch=False)
    b.sync(); b.timing(reset=True)
    lvl = int(os.environ.get("TLEVEL", "2")); b.set_timing(lvl)
    t0 = time.perf_counter()
    for _ in range(steps): b.run(frames, fetch=False)
    b.sync(); dt = time.perf_counter() - t0
    t = b.timing_ex()
    n = max(1, t["segments"])
    print(f"{name:28s} wall {dt/steps*1e3:7.3f} ms/step  fast {t['fast_ms']/n:6.3f}  block {t['block_ms']/n:6.3f}  mix {t['mix_ms']/n:6.3f}  aux {t['aux_ms']/n:6.3f} ms -> {frames*steps/dt:10.3e} frames/s")
which = sys.argv[1:] or ["c2", "c3", "c3x4"]
if "c2" in which: run("1024 x 1 op (flat)", vb.config2(n=1024, seconds=30))
if "c2b" in which: run("4096 x 1 op (flat)", vb.config2(n=4096, seconds=30))
if "c3" in which: run("1024 x 4 ops (config 3)", vb.config3(n=1024, seconds=30))
if "c3x4" in which: run("4096 x 4 ops", vb.config3(n=4096, seconds=30))
if "c5" in which: run("4096 x 2 ops selfmod (c5)", vb.config5(n=4096, seconds=30), frames=11025, steps=3)
