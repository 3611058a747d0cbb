import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
prg = vb.config3(n=1024, seconds=30)
g = sa.Generator(prg, 44100)
buf = np.zeros(176400, np.int16)
for i in range(6):
    t0 = time.perf_counter(); g.run(buf, 176400); print("generator direct run 176400: %.3f ms" % (1e3*(time.perf_counter()-t0)))
g.close()
b = sa.Batch([prg], 44100)
for fetch in (False, True, True, False, True):
    b.sync(); t0 = time.perf_counter(); b.run(176400, fetch=fetch); b.sync()
    print("batch run fetch=%s: %.3f ms" % (fetch, 1e3*(time.perf_counter()-t0)))
