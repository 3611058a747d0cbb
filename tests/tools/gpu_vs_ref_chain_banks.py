"""GPU against the compiled reference on banks of voices built around feedback chains: hundreds of voices of the shapes of
tests/test_gpu_units.py::test_feedback_chains_that_running_sums_depend_on (a chain under running sums as in kaboom1.sau, a chain
FM-ing a carrier, chain -> chain, R-oscillator feedback as carrier / PM source / FM source) with parameters drawn per voice.
    python tests/tools/gpu_vs_ref_chain_banks.py [first_seed [banks]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def voice(rng, ms):
    u = rng.uniform
    shape = lambda: LINES[int(rng.integers(len(LINES)))]
    ln = lambda a, b: vb.Line(float(u(a, b)), goal=float(u(a, b)), shape=shape()) if rng.random() < 0.4 else float(u(a, b))
    fb = lambda f, amp, **kw: vb.Op(WAVES[int(rng.integers(len(WAVES)))], freq=f, pm_a=ln(0.05, 0.9), amp=amp, **kw)
    R = lambda **kw: vb.Op(op_type=POPT_RASEG, ras=(shape(), int(rng.integers(6)), int(rng.integers(32))), seed=int(rng.integers(1 << 32)), **kw)
    k = int(rng.integers(7))
    if k == 0:   # kaboom1: chain -> R frequency -> carrier frequency
        return vb.Op("sin", freq=float(u(-600, 600)), freq2=float(u(100, 900)), amp=ln(0.1, 0.9), time_ms=ms,
                     mods={POP_RFMOD: [R(freq=float(u(0.5, 3)), freq2=float(u(5, 20)), mods={POP_RFMOD: [fb(float(u(0.1, 4)), 1.0, phase=float(u(0, 1)))]})]})
    if k == 1:   # a chain FM-ing a carrier
        return vb.Op("tri", freq=float(u(80, 500)), amp=0.7, time_ms=ms, mods={POP_FMOD: [fb(ln(1, 9), float(u(5, 60)))]})
    if k == 2:   # chain -> chain
        return vb.Op("sin", freq=float(u(80, 400)), pm_a=ln(0.1, 0.6), amp=0.7, time_ms=ms, mods={POP_PMOD: [fb(ln(1, 6), float(u(0.2, 0.9)))]})
    if k == 3:   # an early and an ordinary chain in one voice
        return vb.Op("saw", freq=float(u(60, 300)), amp=0.6, time_ms=ms,
                     mods={POP_FMOD: [fb(float(u(1, 5)), float(u(5, 30)))],
                           POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), pm_a=ln(0.1, 0.7), amp=0.7,
                                            mods={POP_PMOD: [vb.Op("tri", freq=vb.Line(3.0, ratio=True), amp=0.3)]})]})
    if k == 4:   # R feedback carrier
        return R(freq=ln(60, 400), pm_a=ln(0.1, 1.2), amp=0.6, time_ms=ms)
    if k == 5:   # R feedback as PM and FM source
        return vb.Op("sin", freq=float(u(100, 400)), time_ms=ms, pm_a=ln(0.0, 0.5),
                     mods={POP_PMOD: [R(freq=vb.Line(float(u(0.5, 3)), ratio=True), pm_a=ln(0.2, 0.9), amp=0.5)],
                           POP_FMOD: [R(freq=float(u(2, 12)), pm_a=ln(0.2, 0.9), amp=float(u(5, 40)))]})
    return fb(ln(80, 600), 0.7, time_ms=ms)  # a plain feedback voice
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(400000 + seed)
    n = int(rng.choice([40, 200, 600]))
    ms = int(rng.integers(150, 900))
    voices = [voice(rng, ms - int(rng.integers(0, 60))) for _ in range(n)]
    prg = vb.build_program(voices)
    stereo = bool(seed & 1); call = int(rng.integers(500, 40000))
    t0 = time.time(); ref = po.ref_render(prg.ptr, 44100, stereo, chunk=call); t1 = time.time()
    b = sa.Batch([prg], 44100); got = b.render(stereo=stereo, chunk=call)[0]; t2 = time.time()
    same = len(got) == len(ref) and bool((got == ref).all())
    bad += not same
    print(f"seed {seed}: {n} voices, {len(ref)} samples, call {call}: reference {t1-t0:.1f} s, GPU {t2-t1:.2f} s,", "identical" if same else "DIFFERS", flush=True)
print("differing banks:", bad)
sys.exit(1 if bad else 0)
