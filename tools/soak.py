"""Soak check: 300 generators created, run to their end and destroyed; device memory must stay flat once the pools are warm."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import saugns_amd as sa
from saugns_amd import voicebank as vb
def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
torch.cuda.init()
base = used()
buf = np.zeros(11289, np.int16)
for rep in range(300):
    prg = vb.config3(n=64 + (rep % 5) * 37, seconds=2)
    g = sa.Generator(prg, 44100)
    while True:
        more, n = g.run(buf, 11289)
        if not more: break
    g.close()
    if rep in (0, 9, 99, 299): print("after", rep + 1, "generators: device memory in use %.0f MiB (start %.0f)" % (used(), base), flush=True)
