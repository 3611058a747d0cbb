"""List corpus programs whose GPU render differs from the oracle (debug aid)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from oracle import pyoracle as po
from conftest import ORACLE_FORMS  # 2 by default: the reference build's loop tails reproduced (the 11289-frame calls of both sides)
G = os.path.join(ROOT, "tests", "golden")
index = json.load(open(os.path.join(G, "index.json")))
tabs = np.fromfile(os.path.join(G, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
rate = index["corpus_rate"]
bad = []
for key in sorted(index["corpus"]):
    prg = sa.Program.from_image(open(os.path.join(G, "programs", key + ".saup"), "rb").read())
    want = po.oracle_render(prg.ptr, rate, True)
    got = sa.Generator(prg, rate).render(stereo=True)
    if len(got) != len(want):
        bad.append((key, "len", len(got), len(want)))
    else:
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        if d.size and d.max() != 0:
            bad.append((key, int(d.max()), int(np.argmax(d != 0)) // 2, len(want) // 2))
print(len(bad), "bad of", len(index["corpus"]))
for b in bad: print(b)
