import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["SAU_AMD_DEBUG_CREATE"] = "1"
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank
tabs = np.fromfile("tests/golden/piluts_ref.f32", dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
prg = voicebank.config5(n=4096, seconds=10)
for i in range(4):
    t0 = time.perf_counter(); b = sa.Batch([prg], 44100); t1 = time.perf_counter()
    b.run(441000, stereo=False, fetch=False); t2 = time.perf_counter()
    b.sync(); t3 = time.perf_counter(); b.close(); t4 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} run(enqueue) {1e3*(t2-t1):.3f} sync {1e3*(t3-t2):.3f} close {1e3*(t4-t3):.3f} ms", flush=True)
