import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
po.build(ref=False); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
for n in (1, 2, 8, 64):
    prg = vb.config5(n=n, seconds=30)
    t0 = time.perf_counter(); want = po.oracle_render(prg.ptr, 44100, False); t_cpu = time.perf_counter() - t0
    have_ref = po.have_ref()
    t_ref = None
    if have_ref:
        po.ref()
        t0 = time.perf_counter(); r = po.ref_render(prg.ptr, 44100, False); t_ref = time.perf_counter() - t0
    b = sa.Batch([prg], 44100); b.render(stereo=False, chunk=1323000)  # warm
    b = sa.Batch([prg], 44100)
    t0 = time.perf_counter(); got = b.render(stereo=False, chunk=1323000)[0]; t_gpu = time.perf_counter() - t0
    ok = len(got) == len(want) and (got == want).all()
    print(f"config5 n={n}: {len(want)} frames; oracle {t_cpu*1e3:.1f} ms, compiled reference {t_ref*1e3 if t_ref else float('nan'):.1f} ms, GPU (create excluded, one call) {t_gpu*1e3:.1f} ms; equal {ok}")
