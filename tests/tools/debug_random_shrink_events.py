"""Debug aid (GPU): shrink a failing random program with events (drop voices, events, subtrees):
    RATE=96000 python tests/tools/debug_random_shrink_events.py <seed> <chunk>"""
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
seed, ck = int(sys.argv[1]), int(sys.argv[2])
rate = int(os.environ.get("RATE", "44100"))
rng = np.random.default_rng(5000 + seed)
voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
tu._random_starts(rng, voices)
ups = tu._random_updates(rng, voices)
stereo = bool(seed & 1)
def fails(voices, ups):
    try:
        prg = vb.build_program(voices, updates=ups)
    except Exception:
        return False
    want = po.oracle_render(prg.ptr, rate, stereo)
    got = sa.Batch([prg], rate).render(stereo=stereo, chunk=ck)[0]
    return len(got) != len(want) or bool((got != want).any())
assert fails(voices, ups)
def nodes(op, acc):
    acc.append(op)
    for lst in op.mods.values():
        for m in lst: nodes(m, acc)
    return acc
changed = True
while changed:
    changed = False
    for i in range(len(ups)):
        u2 = ups[:i] + ups[i + 1:]
        if fails(voices, u2): ups = u2; changed = True; break
    if changed: continue
    for vi in range(len(voices)):
        if len(voices) == 1: break
        v2 = voices[:vi] + voices[vi + 1:]
        u2 = [(a, (w if w < vi else w - 1), o, x) for (a, w, o, x) in ups if w != vi]
        if fails(v2, u2): voices, ups = v2, u2; changed = True; break
    if changed: continue
    for root in voices:
        for op in nodes(root, []):
            for use in list(op.mods):
                for i in range(len(op.mods[use])):
                    saved = op.mods[use]
                    gone = nodes(saved[i], [])
                    if any(o is u[2] for u in ups for o in gone): continue
                    if any(m is saved[i] for u in ups for lst in u[3].get("mods", {}).values() for m in lst): continue
                    op.mods[use] = saved[:i] + saved[i + 1:]
                    if not op.mods[use]: del op.mods[use]
                    if fails(voices, ups): changed = True; break
                    op.mods[use] = saved
                if changed: break
            if changed: break
        if changed: break
    if changed: continue
    for k, (a, w, o, x) in enumerate(ups):
        for key in list(x):
            x2 = {kk: vv for kk, vv in x.items() if kk != key}
            if not x2: continue
            u2 = ups[:k] + [(a, w, o, x2)] + ups[k + 1:]
            if fails(voices, u2): ups = u2; changed = True; break
        if changed: break
def ln(l):
    if l is None: return "-"
    if not hasattr(l, "v0"): return repr(l)
    return f"{l.v0!r}" + (f"->{l.goal!r}({l.shape})" if l.goal is not None else "") + ("r" if l.ratio else "") + ("" if getattr(l, "state", True) else "[goal only]")
def show(op, ind=0, use="carrier"):
    print("  " * ind + f"{use}: #{getattr(op,'_id','?')} type {op.op_type} {op.wave} ras={getattr(op,'ras',None)} noise={getattr(op,'noise',None)} f={ln(op.freq)} f2={ln(op.freq2)} a={ln(op.amp)} a2={ln(op.amp2)} pma={ln(op.pm_a)} pan={ln(getattr(op,'pan',None))} t={op.time_ms} start={getattr(op,'start_ms',0)} ph={op.phase!r}")
    for u, lst in op.mods.items():
        for m in lst: show(m, ind + 1, str(u))
for v in voices: show(v)
for (a, w, o, x) in ups:
    print("event at", a, "voice", w, "op #", getattr(o, "_id", "?"), {k: (ln(v) if hasattr(v, "v0") else ([getattr(m,'_id','?') for m in sum(v.values(), [])] if k == "mods" else v)) for k, v in x.items()})
print("stereo", stereo, "rate", rate, "chunk", ck)
