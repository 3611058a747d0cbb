"""GPU against the compiled reference on (a) ramps of many minutes -- positions beyond 2^24 frames, where f32 no longer holds
every integer -- on every line kind and shape, and (b) programs with hundreds of events a few frames apart.
    python tests/tools/gpu_vs_ref_long.py [minutes]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
import test_gpu_units as T
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
ms = int(minutes * 60000)
bad = 0
def cmp(name, prg, rate, stereo, call):
    global bad
    t0 = time.time(); ref = po.ref_render(prg.ptr, rate, stereo, chunk=call); t1 = time.time()
    g = sa.Generator(prg, rate); got = g.render(stereo=stereo, chunk=call); g.close(); t2 = time.time()
    same = len(got) == len(ref) and bool((got == ref).all())
    bad += not same
    print(f"{name}: {len(ref)} samples, reference {t1-t0:.1f} s, GPU {t2-t1:.2f} s, {'identical' if same else 'DIFFERS at %s' % (np.nonzero(got[:len(ref)] != ref[:len(got)])[0][:3],)}", flush=True)
for k, shape in enumerate(LINES):
    m = vb.Op("sin", freq=vb.Line(2.0, goal=0.25, shape=shape, ratio=True), amp=vb.Line(0.1, goal=2.0, shape=LINES[(k + 5) % len(LINES)]))
    r = vb.Op(op_type=POPT_RASEG, ras=(shape, k % 6, (0, 1, 9, 16)[k % 4]), seed=5 + k, freq=vb.Line(3.0, goal=40.0, shape=shape), amp=0.3)
    v = vb.Op(WAVES[k % len(WAVES)], freq=vb.Line(80.0, goal=1200.0, shape=shape), amp=vb.Line(0.9, goal=0.05, shape=shape), time_ms=ms,
              pan=vb.Line(-1.0, goal=1.0, shape=shape), mods={POP_PMOD: [m], POP_AMOD: [r]})
    cmp(f"{minutes:g} min ramps, shape {shape}", vb.build_program([v]), 44100, bool(k & 1), 11289)
# events a few frames apart
for seed in range(12):
    rng = np.random.default_rng(91000 + seed)
    voices = [T._random_voice(rng) for _ in range(3)]
    for v in voices: v.time_ms = 400
    ups = []
    for _ in range(300):
        ups += [u for u in T._random_updates(rng, voices) if u[0] < 400][:2]
    prg = vb.build_program(voices, updates=ups)
    cmp(f"{len(ups)} events in 400 ms, seed {seed}", prg, (44100, 96000, 8000)[seed % 3], bool(seed & 1), int(rng.integers(5, 3000)))
print("differing:", bad)
sys.exit(1 if bad else 0)
