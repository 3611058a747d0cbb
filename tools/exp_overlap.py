"""Experiment (r03): does the mixer of one batch overlap the time-parallel kernel of another when the
latter leaves register room on the SIMDs?  One batch alone against two batches driven alternately
(each on its own stream), config 3, whole-render steps.  SAU_AMD_LIB selects the library variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
F, K = 441000, int(os.environ.get("K", "12"))
prg = vb.config3(n=1024, seconds=10 * (K + 6))
def go(nb):
    bs = [sa.Batch([prg], 44100) for _ in range(nb)]
    for b in bs:
        for _ in range(2): b.run(F, fetch=False)
    for b in bs: b.sync()
    t0 = time.perf_counter()
    for _ in range(K // nb):
        for b in bs: b.run(F, fetch=False)
    for b in bs: b.sync()
    dt = time.perf_counter() - t0
    n = (K // nb) * nb
    print(f"{os.environ.get('SAU_AMD_LIB', 'default').split('/')[-1]:24s} rows {os.environ.get('SAU_AMD_FAST_ROWS', '8')} batches {nb}: {dt / n * 1e3:7.3f} ms/step -> {F * n / dt:10.4e} frames/s", flush=True)
    for b in bs: b.close()
go(int(os.environ.get("NB", "1")))
