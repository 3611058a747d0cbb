"""Soak of the launch that mixes its own rows: N consecutive 441000-frame runs of config 3's bank, every run's PCM
hashed, with the launch mixing and with every frame left to mix_kernel -- the two lists must be equal (a stale read of a voice row
in a tile would show as one differing hash). Also with two generators alternating on the device, so that their launches follow each
other closely.   python tests/tools/gpu_inmix_soak.py [runs]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sa.lib()
sa.set_piluts(np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
os.environ["SAU_AMD_TUNE"] = "1"


def runs(n, two):
    prg = voicebank.config3(seconds=10 * n)
    batches = [sa.Batch([prg], 44100) for _ in range(2 if two else 1)]
    out = []
    for i in range(n):
        for b in batches:
            out.append(hashlib.sha256(np.ascontiguousarray(b.run(441000, stereo=False)[0][0]).tobytes()).hexdigest())
    for b in batches:
        b.close()
    return out


res = {}
for two in (False, True):
    t0 = time.time()
    os.environ.pop("SAU_AMD_NO_INMIX", None)
    a = runs(N if not two else N // 3, two)
    os.environ["SAU_AMD_NO_INMIX"] = "1"
    b = runs(N if not two else N // 3, two)
    res["two generators alternating" if two else "one generator"] = {
        "runs": len(a), "equal": sum(x == y for x, y in zip(a, b)), "distinct": len(set(a)), "seconds": round(time.time() - t0, 1)}
print(json.dumps(res))
sys.exit(0 if all(v["runs"] == v["equal"] for v in res.values()) else 1)
