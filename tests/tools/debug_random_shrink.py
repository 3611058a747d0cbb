"""Debug aid: shrink a failing random graph (remove subtrees / simplify lines while it still fails)."""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as tu
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
seed, vi, ck = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1000 + seed)
voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
root = voices[vi]
def fails(v):
    prg = vb.build_program([copy.deepcopy(v)])
    want = po.oracle_render(prg.ptr, 44100, False)
    got = sa.Batch([prg], 44100).render(stereo=False, chunk=ck)[0]
    return len(got) != len(want) or bool((got != want).any())
assert fails(root)
def nodes(op, acc):
    acc.append(op)
    for lst in op.mods.values():
        for m in lst: nodes(m, acc)
    return acc
changed = True
while changed:
    changed = False
    for op in nodes(root, []):
        for use in list(op.mods):
            for i in range(len(op.mods[use])):
                saved = op.mods[use]
                op.mods[use] = saved[:i] + saved[i + 1:]
                if not op.mods[use]: del op.mods[use]
                if fails(root): changed = True; break
                op.mods[use] = saved
            if changed: break
        if changed: break
    if changed: continue
    for op in nodes(root, []):
        for name in ("amp", "freq", "amp2", "freq2"):
            ln = getattr(op, name)
            if ln is not None and ln.goal is not None:
                g = ln.goal; ln.goal = None
                if fails(root): changed = True; break
                ln.goal = g
        if changed: break
        if op.time_ms is not None and op is not root:
            t = op.time_ms; op.time_ms = None
            if fails(root): changed = True; break
            op.time_ms = t
def show(op, ind=0, use="carrier"):
    def ln(l):
        if l is None: return "-"
        return f"{l.v0!r}" + (f"->{l.goal!r}({l.shape})" if l.goal is not None else "") + ("r" if l.ratio else "")
    print("  " * ind + f"{use}: {op.wave} f={ln(op.freq)} f2={ln(op.freq2)} a={ln(op.amp)} a2={ln(op.amp2)} pma={ln(op.pm_a)} t={op.time_ms} ph={op.phase!r}")
    for u, lst in op.mods.items():
        for m in lst: show(m, ind + 1, str(u))
show(root)
