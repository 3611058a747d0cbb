import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank
tabs = np.fromfile(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.lib(); sa.set_piluts(tabs)
prg = voicebank.config3()
buf = np.zeros(11289, np.int16)
for rep in range(4):
    t0 = time.perf_counter()
    g = sa.Generator(prg, 44100)
    t1 = time.perf_counter()
    calls = []
    more = True
    while more:
        a = time.perf_counter()
        more, n = g.run(buf, 11289, False)
        calls.append(time.perf_counter() - a)
    t2 = time.perf_counter()
    g.close()
    t3 = time.perf_counter()
    big = sorted(((c, i) for i, c in enumerate(calls)), reverse=True)[:6]
    print("rep %d: create %.3f ms, %d calls %.3f ms, close %.3f ms, total %.3f ms; longest calls (ms, index): %s" % (
        rep, (t1 - t0) * 1e3, len(calls), (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, [(round(c * 1e3, 3), i) for c, i in big]))
