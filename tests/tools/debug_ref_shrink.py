"""Debug aid (CPU): shrink a random graph on which the restatement and the compiled reference differ
by more than `tol` LSB:  python tests/tools/debug_ref_shrink.py <seed> <voice index> [tol]"""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as tu
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(2)
seed, vi = int(sys.argv[1]), int(sys.argv[2])
tol = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(1000 + seed)
voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
def diff(v):
    prg = vb.build_program([copy.deepcopy(v)])
    a = po.oracle_render(prg.ptr, 44100, False); b = po.ref_render(prg.ptr, 44100, False)
    if len(a) != len(b): return 1 << 20
    return int(np.abs(a.astype(int) - b.astype(int)).max()) if len(a) else 0
if vi < 0:
    for i, v in enumerate(voices): print("voice", i, "diff", diff(v))
    sys.exit(0)
root = voices[vi]
fails = lambda v: diff(v) > tol
assert fails(root), diff(root)
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
        for name in ("amp", "freq", "amp2", "freq2", "pan"):
            ln = getattr(op, name, None)
            if ln is not None and getattr(ln, "goal", None) is not None:
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
        if not hasattr(l, "v0"): return repr(l)
        return f"{l.v0!r}" + (f"->{l.goal!r}({l.shape})" if l.goal is not None else "") + ("r" if l.ratio else "")
    print("  " * ind + f"{use}: type {op.op_type} {op.wave} ras={getattr(op,'ras',None)} noise={getattr(op,'noise',None)} seed={getattr(op,'seed',None)} f={ln(op.freq)} f2={ln(op.freq2)} a={ln(op.amp)} a2={ln(op.amp2)} pma={ln(op.pm_a)} t={op.time_ms} ph={op.phase!r}")
    for u, lst in op.mods.items():
        for m in lst: show(m, ind + 1, str(u))
show(root); print("diff", diff(root))
