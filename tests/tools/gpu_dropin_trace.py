"""The drop-in generator's first call on BASELINE config 3, host phases on stderr (SAU_AMD_DEBUG_CREATE): what of the first call is
host work ahead of the device's 2 ms?   SAU_AMD_DEBUG_CREATE=1 python tests/tools/gpu_dropin_trace.py"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
prg = vb.config3(seconds=10)
buf = np.zeros(11289, dtype=np.int16)
for rep in range(3):
    print("== rep", rep, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    g = sa.Generator(prg, 44100)
    t1 = time.perf_counter()
    first = None
    n = 0
    while True:
        ta = time.perf_counter()
        more, got = g.run(buf, 11289, False)
        if first is None:
            first = time.perf_counter() - ta
        n += got
        if not more:
            break
    t2 = time.perf_counter()
    g.close()
    t3 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.3f} first call {1e3*first:.3f} all calls {1e3*(t2-t1):.3f} close {1e3*(t3-t2):.3f} total {1e3*(t3-t0):.3f} ms, {n} frames", file=sys.stderr)
