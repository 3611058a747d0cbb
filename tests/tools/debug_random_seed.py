"""Debug aid: one seed of test_random_operator_graphs, with the tree printed."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as tu
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
seed = int(sys.argv[1]); chunk_override = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(1000 + seed)
voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
chunk = int(rng.integers(700, 3000))
def show(op, ind=0, use="carrier"):
    def ln(l):
        if l is None: return "-"
        return f"{l.v0:.4g}" + (f"->{l.goal:.4g}({l.shape})" if l.goal is not None else "") + ("r" if l.ratio else "")
    print("  " * ind + f"{use}: {op.wave} f={ln(op.freq)} f2={ln(op.freq2)} a={ln(op.amp)} a2={ln(op.amp2)} pma={ln(op.pm_a)} t={op.time_ms}")
    for u, lst in op.mods.items():
        for m in lst: show(m, ind + 1, str(u))
for v in voices: show(v)
stereo = bool(seed & 1)
prg = vb.build_program(voices)
want = po.oracle_render(prg.ptr, 44100, stereo)
for ck in ([chunk_override] if chunk_override else [4000000, chunk]):
    got = sa.Batch([prg], 44100).render(stereo=stereo, chunk=ck)[0]
    d = np.nonzero(got != want)[0]
    print("chunk", ck, "frames", len(want) // (2 if stereo else 1), "diffs", len(d), "first", d[:4], "max", int(np.abs(got.astype(int) - want.astype(int)).max()) if len(d) else 0)
if len(d):
    print("diff frames:", sorted(set((d // (2 if stereo else 1)).tolist()))[:40])
    # which voice? render each voice alone
    for vi, v in enumerate(voices):
        p1 = vb.build_program([v])
        w1 = po.oracle_render(p1.ptr, 44100, stereo)
        g1 = sa.Batch([p1], 44100).render(stereo=stereo, chunk=ck)[0]
        dd = np.nonzero(g1 != w1)[0]
        print(" voice", vi, "alone: diffs", len(dd), sorted(set((dd // (2 if stereo else 1)).tolist()))[:12])
