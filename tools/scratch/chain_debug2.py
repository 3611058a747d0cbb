import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from oracle import pyoracle as po
from lattice_cases import lattice_case
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
rate = 44100
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(77000 + seed)
    prg = lattice_case(rng)
    call, run = int(rng.integers(300, 3000)), int(rng.integers(1000, 30000))
    stereo = bool(seed & 1)
    want = po.oracle_render(prg.ptr, rate, stereo, chunk=call)
    print("seed", seed, "call", call, "run", run, "frames", len(want), flush=True)
    b = sa.Batch([prg], rate); b.set_call_len(call)
    got = b.render(stereo=stereo, chunk=run)[0]
    print("   ok" if len(got) == len(want) and (got == want).all() else "   DIFF", flush=True)
    b.close()
