"""Debug aid: the random operator graphs stretched -- seconds instead of tenths of seconds, up to
twelve voices, frequencies up to 20 kHz and below zero -- GPU vs oracle (and, with REF=1 on a box
that has it, oracle vs the compiled reference):  python tests/tools/debug_random_stress.py <lo> <hi>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as tu
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
po.oracle_use_tables(tabs)
use_ref = os.environ.get("REF") == "1"
if not use_ref:
    import saugns_amd as sa
    sa.set_piluts(tabs)
po.oracle().ora_set_fastmath_forms(2 if use_ref else 1)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
def nodes(op, acc):
    acc.append(op)
    for lst in op.mods.values():
        for m in lst: nodes(m, acc)
    return acc
for seed in range(lo, hi):
    rng = np.random.default_rng(70000 + seed)
    voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 13)))]
    scale = float(rng.choice([3.0, 8.0, 20.0]))
    for carr in voices:
        for op in nodes(carr, []):
            if op.time_ms is not None: op.time_ms = int(op.time_ms * scale)
        if rng.random() < 0.3 and hasattr(carr.freq, "v0"):
            carr.freq.v0 = float(rng.choice([-1.0, 1.0])) * float(rng.uniform(2000, 20000))
    tu._random_starts(rng, voices)
    for carr in voices:
        if getattr(carr, "start_ms", 0): carr.start_ms = int(carr.start_ms * scale)
    ups = tu._random_updates(rng, voices)
    rate = int(rng.choice([44100, 48000, 22050]))
    stereo = bool(seed & 1)
    prg = vb.build_program(voices, updates=ups, ampmult=float(rng.choice([1.0, 0.5, 2.0])),
                           amp_div_voices=bool(rng.random() < 0.7))
    for ck in (4000000, int(rng.integers(3000, 30000))):
        want = po.oracle_render(prg.ptr, rate, stereo, chunk=ck)
        got = po.ref_render(prg.ptr, rate, stereo, chunk=ck) if use_ref else sa.Batch([prg], rate).render(stereo=stereo, chunk=ck)[0]
        if len(got) != len(want) or (got != want).any():
            bad.append((seed, ck)); print("FAIL seed", seed, "chunk", ck, "frames", len(want), flush=True)
print("checked", hi - lo, "seeds;", len(bad), "failures", bad)
