"""Debug aid (GPU): shrink a pair of programs of one batch of tests/tools/gpu_vs_ref_batches.py -- program A is rendered
wrong (against the compiled reference) only beside program B -- by removing voices, events, modulator subtrees and ramps
while the difference stays:  python tests/tools/debug_batch_shrink.py <batch seed> <A> <B>"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
bseed, A, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(300000 + bseed)
allv = []
for k in range(12):
    voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    T._random_starts(rng, voices)
    ups = T._random_updates(rng, voices)
    if bseed % 4 == 3:
        T._push_extremes(rng, voices)
    allv.append([voices, list(ups)])
rate = int(rng.choice([44100, 44100, 48000, 96000, 8000]))
stereo = bool(bseed & 1)
call = int(rng.integers(1, 12)) if bseed % 5 == 4 else int(rng.integers(300, 12000))
chunk = call * (1500 if call < 300 else int(rng.integers(1, 6)))
state = {"A": allv[A], "B": allv[B]}

def build(vu):
    voices, ups = copy.deepcopy(vu)
    return vb.build_program(voices, updates=ups)

def differs(st, info=False):
    try:
        pa, pb = build(st["A"]), build(st["B"])
    except Exception as e:  # a reduction that no longer builds
        return False
    ref = po.ref_render(pa.ptr, rate, stereo, chunk=call)
    def gpu(prgs, i):
        b = sa.Batch(prgs, rate); b.set_call_len(call)
        return b.render(stereo=stereo, chunk=chunk)[i]
    alone = gpu([pa], 0)
    if len(alone) < len(ref) or (alone[:len(ref)] != ref).any():
        return False  # (must stay right alone)
    got = gpu([pb, pa], 1)
    n = min(len(got), len(ref))
    d = np.nonzero(got[:n] != ref[:n])[0]
    if info:
        print("differing samples", list(d[:10]), "got", list(got[d[:10]]), "ref", list(ref[d[:10]]), "frames", len(ref) // (2 if stereo else 1))
    return len(d) > 0

def nodes(op, acc):
    acc.append(op)
    for lst in op.mods.values():
        for m in lst: nodes(m, acc)
    return acc

def candidates(st):
    """every one-step reduction of the state, as functions that apply it to a deep copy"""
    out = []
    for key in ("A", "B"):
        voices, ups = st[key]
        for vi in range(len(voices)):
            if len(voices) > 1:
                def rm_voice(s, key=key, vi=vi):
                    v, u = s[key]
                    del v[vi]
                    s[key][1] = [(at, (i if i < vi else i - 1), op, what) for (at, i, op, what) in u if i != vi]
                out.append(("voice %s/%d" % (key, vi), rm_voice))
        for ui in range(len(ups)):
            def rm_up(s, key=key, ui=ui):
                del s[key][1][ui]
            out.append(("update %s/%d" % (key, ui), rm_up))
            for wk in list(ups[ui][3]):
                if len(ups[ui][3]) > 1:
                    def rm_what(s, key=key, ui=ui, wk=wk):
                        del s[key][1][ui][3][wk]
                    out.append(("update %s/%d field %s" % (key, ui, wk), rm_what))
        for vi, v in enumerate(voices):
            ns = nodes(v, [])
            for ni, op in enumerate(ns):
                for use in list(op.mods):
                    for mi in range(len(op.mods[use])):
                        def rm_mod(s, key=key, vi=vi, ni=ni, use=use, mi=mi):
                            o = nodes(s[key][0][vi], [])[ni]
                            gone = nodes(o.mods[use][mi], [])
                            ids = set(id(g) for g in gone)
                            del o.mods[use][mi]
                            if not o.mods[use]: del o.mods[use]
                            s[key][1] = [u for u in s[key][1] if id(u[2]) not in ids]
                            for u in s[key][1]:
                                if "mods" in u[3]:
                                    u[3]["mods"] = {k2: [m for m in l2 if id(m) not in ids] for k2, l2 in u[3]["mods"].items()}
                        out.append(("mod %s/%d node %d use %d #%d" % (key, vi, ni, use, mi), rm_mod))
                for name in ("amp", "freq", "amp2", "freq2", "pan", "pm_a"):
                    ln = getattr(op, name, None)
                    if ln is not None and getattr(ln, "goal", None) is not None:
                        def rm_goal(s, key=key, vi=vi, ni=ni, name=name):
                            getattr(nodes(s[key][0][vi], [])[ni], name).goal = None
                        out.append(("goal %s/%d node %d %s" % (key, vi, ni, name), rm_goal))
    return out

assert differs(state, info=True), "the pair does not differ here"
progress = True
while progress:
    progress = False
    for name, fn in candidates(state):
        trial = copy.deepcopy(state)
        try:
            fn(trial)
        except Exception:
            continue
        if differs(trial):
            state = trial; progress = True
            print("removed", name, flush=True)
            break

def show(op, ind=0, use="carrier"):
    def ln(l):
        if l is None: return "-"
        if not hasattr(l, "v0"): return repr(l)
        return f"{l.v0!r}" + (f"->{l.goal!r}({l.shape},state={l.state})" if l.goal is not None else "") + ("r" if l.ratio else "")
    print("  " * ind + f"{use}: id={id(op) % 100000} type {op.op_type} {op.wave} ras={getattr(op,'ras',None)} nz={getattr(op,'noise',None)} seed={op.seed} f={ln(op.freq)} f2={ln(op.freq2)} a={ln(op.amp)} a2={ln(op.amp2)} pma={ln(op.pm_a)} pan={ln(getattr(op,'pan',None))} t={op.time_ms} ph={op.phase!r} start={getattr(op,'start_ms',None)}")
    for u, lst in op.mods.items():
        for m in lst: show(m, ind + 1, str(u))
print("rate", rate, "stereo", stereo, "call", call, "chunk", chunk)
for key in ("A", "B"):
    print("==== program", key)
    for v in state[key][0]: show(v)
    for (at, vi, op, what) in state[key][1]:
        print("  update at", at, "voice", vi, "op id", id(op) % 100000, {k: (vars(v) if hasattr(v, "__dict__") else v) for k, v in what.items() if k != "mods"}, "mods" if "mods" in what else "")
differs(state, info=True)
import pickle
pickle.dump(state, open(os.path.join(ROOT, "gpurun_out", "r04_case3883_shrunk.pkl"), "wb"))
