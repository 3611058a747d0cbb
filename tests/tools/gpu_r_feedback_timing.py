"""R-oscillator feedback (rchain_kernel) on the GPU: ns per frame for one voice and for banks, checked against the
oracle (and the compiled reference where it is here).   python tests/tools/gpu_r_feedback_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POPT_RASEG
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
po.build(ref=False); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(2)
SEC = 10
for n in (1, 8, 64, 1024, 4096):
    voices = [vb.Op(op_type=POPT_RASEG, ras=(("lin", "cos", "sqe", "xpe")[k % 4], k % 6, (5 * k) % 32), seed=1234 + 77 * k,
                    freq=vb.Line(90.0 + 11 * (k % 50), goal=300.0 + k % 70, shape="exp") if k % 3 else 140.0 + k % 90,
                    pm_a=vb.Line(0.1 + 0.01 * (k % 40), goal=0.9, shape="lin") if k % 2 else 0.5, amp=0.6, time_ms=SEC * 1000)
              for k in range(n)]
    prg = vb.build_program(voices)
    frames = SEC * 44100
    want = po.oracle_render(prg.ptr, 44100, False, chunk=frames) if n <= 64 else None
    sa.Batch([prg], 44100).render(stereo=False, chunk=frames)  # warm
    b = sa.Batch([prg], 44100)
    b.set_timing(2)
    t0 = time.perf_counter(); got = b.render(stereo=False, chunk=frames)[0]; dt = time.perf_counter() - t0
    tm = b.timing_ex()
    ok = want is None or (len(got) == len(want) and bool((got == want).all()))
    print(f"R feedback, {n} voices x {frames} frames: {dt * 1e3:.1f} ms ({dt / frames * 1e9:.0f} ns per frame); kernels: {tm}; "
          f"{'equal to the oracle' if want is not None and ok else 'DIFFERS' if not ok else 'not compared'}", flush=True)
    assert ok
