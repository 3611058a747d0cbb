"""R-oscillator feedback (rchain_kernel): what does a bank of ONE kind cost, kind by kind? The mixed bank of
gpu_r_feedback_timing.py runs its kinds on waves of their own side by side, so its time is its slowest kind's: this prints the
banks of 1024 like voices for a sample of the kinds of that bank (line shape, function, flags, swept or fixed rate and amount).
    python tests/tools/gpu_r_feedback_kinds.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POPT_RASEG
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
SEC, N = 10, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
KINDS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 14, 17, 19, 22, 23, 29, 31]
frames = SEC * 44100
out = []
for k in KINDS:
    shape, func, flags = ("lin", "cos", "sqe", "xpe")[k % 4], k % 6, (5 * k) % 32
    swept_f, swept_a = bool(k % 3), bool(k % 2)
    voices = [vb.Op(op_type=POPT_RASEG, ras=(shape, func, flags), seed=1234 + 77 * j,
                    freq=vb.Line(90.0 + 11 * (j % 50), goal=300.0 + j % 70, shape="exp") if swept_f else 140.0 + j % 90,
                    pm_a=vb.Line(0.1 + 0.01 * (j % 40), goal=0.9, shape="lin") if swept_a else 0.5, amp=0.6, time_ms=SEC * 1000)
              for j in range(N)]
    prg = vb.build_program(voices)
    sa.Batch([prg], 44100).render(stereo=False, chunk=frames)  # warm
    b = sa.Batch([prg], 44100)
    t0 = time.perf_counter(); b.render(stereo=False, chunk=frames); dt = time.perf_counter() - t0
    b.close()
    rec = dict(voices=N, kind=k, line=shape, func=func, flags=flags, swept_rate=swept_f, swept_amount=swept_a, ns_per_frame=dt / frames * 1e9)
    out.append(rec); print(rec, flush=True)
if len(sys.argv) <= 1:
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_r_feedback_kinds.json"), "w"), indent=1)
