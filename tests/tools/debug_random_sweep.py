"""Debug aid: many seeds of the random operator-graph parity check (prints the failing ones)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as tu
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1 if os.environ.get("SAU_AMD_LOOP_TAILS") == "0" else 2)  # the product default reproduces the loop tails: mode 2
lo, hi = int(sys.argv[1]), int(sys.argv[2])
events = len(sys.argv) > 3 and sys.argv[3] == "events"
rate = int(os.environ.get("RATE", "44100"))
bad = []
for seed in range(lo, hi):
    rng = np.random.default_rng((5000 if events else 1000) + seed)
    voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    if events: tu._random_starts(rng, voices)
    ups = tu._random_updates(rng, voices) if events else ()
    chunk = int(rng.integers(700, 3000))
    stereo = bool(seed & 1)
    prg = vb.build_program(voices, updates=ups)
    for ck in (4000000, chunk):
        # the same call size on both sides: the reference's output is not always independent of it
        want = po.oracle_render(prg.ptr, rate, stereo, chunk=ck)
        got = sa.Batch([prg], rate).render(stereo=stereo, chunk=ck)[0]
        if len(got) != len(want) or (got != want).any():
            bad.append((seed, ck)); print("FAIL seed", seed, "chunk", ck, flush=True)
print("checked", hi - lo, "seeds;", len(bad), "failures", bad)
