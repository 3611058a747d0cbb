"""One mixed bank of R-feedback voices (the bank of gpu_r_feedback_timing.py), N voices, rendered twice: for a kernel trace.
    rocprofv3 --kernel-trace --stats -d gpurun_out/rfb -- python3 tests/tools/gpu_r_feedback_one.py 1024"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POPT_RASEG
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SEC = 10
voices = [vb.Op(op_type=POPT_RASEG, ras=(("lin", "cos", "sqe", "xpe")[k % 4], k % 6, (5 * k) % 32), seed=1234 + 77 * k,
                freq=vb.Line(90.0 + 11 * (k % 50), goal=300.0 + k % 70, shape="exp") if k % 3 else 140.0 + k % 90,
                pm_a=vb.Line(0.1 + 0.01 * (k % 40), goal=0.9, shape="lin") if k % 2 else 0.5, amp=0.6, time_ms=SEC * 1000)
          for k in range(n)]
prg = vb.build_program(voices)
for rep in range(2):
    b = sa.Batch([prg], 44100)
    t0 = time.perf_counter(); b.render(stereo=False, chunk=SEC * 44100); dt = time.perf_counter() - t0
    b.close()
    print(n, "voices", dt * 1e3, "ms", dt / (SEC * 44100) * 1e9, "ns per frame")
