"""Debug helper: render corpus programs on the GPU and report where they leave the oracle."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from oracle import pyoracle as po
from conftest import load_program, GOLDEN
tabs = np.fromfile(os.path.join(GOLDEN, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
rate = 12000
for key in sys.argv[1:]:
    prg = load_program(sa, key)
    P = prg.struct
    t = 0; evs = []
    for e in range(P.ev_count):
        t += P.events[e].wait_ms; evs.append(t * rate // 1000)
    want = po.oracle_render(prg.ptr, rate, True)
    for chunk in (11289, 256):
        got = sa.Generator(prg, rate).render(stereo=True, chunk=chunk)
        n = min(len(got), len(want))
        d = np.nonzero(got[:n] != want[:n])[0]
        print(key, "chunk", chunk, "len", len(got)//2, len(want)//2, "ndiff", len(d),
              "first", (d[0]//2 if len(d) else None), "events@", evs[:12])
        if len(d):
            i = d[0]//2
            print("   got ", got[2*i-4:2*i+12].tolist()); print("   want", want[2*i-4:2*i+12].tolist())
