"""GPU against the compiled reference with as many voices as a sauProgram can state: 20000 and 65535 (vo_count is a uint16) short voices
of four kinds (closed form, PM, frequency ramp, R oscillator), some starting later -- through the drop-in generator, 11289-frame calls."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
import test_gpu_units as T
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
for n, ms in ((20000, 60), (65535, 30)):
    rng = np.random.default_rng(n)
    voices = []
    for k in range(n):
        kind = k % 4
        if kind == 0: v = vb.Op(WAVES[k % len(WAVES)], freq=50.0 + (k % 977) * 1.7, amp=0.9, time_ms=ms - k % 7)
        elif kind == 1: v = vb.Op("sin", freq=100.0 + k % 500, amp=0.8, time_ms=ms, mods={POP_PMOD: [vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.5)]})
        elif kind == 2: v = vb.Op("sin", freq=vb.Line(100.0 + k % 300, goal=400.0, shape="lin"), amp=0.7, time_ms=ms)
        else: v = vb.Op(op_type=POPT_RASEG, ras=("lin", k % 6, 0), seed=k, freq=80.0 + k % 200, amp=0.6, time_ms=ms)
        voices.append(v)
    T._random_starts(rng, voices[:200])
    t0 = time.time(); prg = vb.build_program(voices); t1 = time.time()
    ref = po.ref_render(prg.ptr, 44100, True, chunk=11289); t2 = time.time()
    g = sa.Generator(prg, 44100); got = g.render(stereo=True, chunk=11289); g.close(); t3 = time.time()
    same = len(got) == len(ref) and bool((got == ref).all())
    print(f"{n} voices: build {t1-t0:.1f} s, reference {t2-t1:.1f} s, GPU {t3-t2:.2f} s, {len(ref)} samples,", "identical" if same else "DIFFERS", flush=True)
