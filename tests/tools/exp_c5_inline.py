"""Experiment (GPU): config-5-like banks with and without ramps on the chain's own lines, inline feeding on and off --
which part of chain_kernel's batch period is the feeder's line evaluation.  python tests/tools/exp_c5_inline.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_RAMOD
os.environ["SAU_AMD_TUNE"] = "1"
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
def bank(freq_ramp, pma_ramp, n=4096, seconds=10):
    vs = []
    for i in range(n):
        lfo = vb.Op("sin", freq=float(3 + i % 9), amp=1.0)
        f0 = 80.0 + i * 0.211
        vs.append(vb.Op("sin", freq=vb.Line(f0, goal=160.0 + i * 0.1, shape="exp") if freq_ramp else f0,
                        pm_a=vb.Line(0.3 + (i % 8) * 0.1, goal=0.1, shape="lin") if pma_ramp else 0.3 + (i % 8) * 0.1,
                        amp=vb.Line(1.0, goal=0.2, shape="xpe"), amp2=vb.Line(0.2), time_ms=seconds * 1000, mods={POP_RAMOD: [lfo]}))
    return vb.build_program(vs)
for fr, pr in ((True, True), (False, True), (True, False), (False, False)):
    prg = bank(fr, pr)
    for inline in ("", "1"):
        if inline: os.environ.pop("SAU_AMD_NO_CHAIN_INLINE", None)
        else: os.environ["SAU_AMD_NO_CHAIN_INLINE"] = "1"
        best = None
        for rep in range(3):
            b = sa.Batch([prg], 44100); b.set_timing(2)
            t0 = time.perf_counter(); b.run(441000, fetch=False); b.sync(); dt = time.perf_counter() - t0
            tm = b.timing_ex(); b.close()
            if rep and (best is None or dt < best[0]): best = (dt, tm)
        print(f"freq ramp {fr!s:5} pm_a ramp {pr!s:5} inline {inline or '0'}: {best[0]*1e3:7.2f} ms, chain_kernel {best[1]['block_ms']:7.2f} ms ({best[1]['block_ms']*1e6/441000:6.1f} ns per frame), passes {best[1]['fast_ms']:7.2f}", flush=True)
