"""Debug aid: one program of tests/tools/gpu_vs_ref_sweep.py's `extreme` kind by seed -- oracle against the compiled reference on the
CPU, and (with a GPU) the device under several kernel configurations.  python tests/tools/debug_seed_extreme.py <seed> [gpu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SAU_AMD_TUNE"] = "1"
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
seed = int(sys.argv[1]); gpu = len(sys.argv) > 2
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
po.build(ref=False); po.oracle_use_tables(tabs)
rng = np.random.default_rng(20000 + seed)
voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
ups = ()
if seed % 2:
    T._random_starts(rng, voices)
    ups = T._random_updates(rng, voices)
T._push_extremes(rng, voices)
rate = int(rng.choice([1000, 3000, 11025, 44100, 192000, 384000]))
prg = vb.build_program(voices, updates=ups)
call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
stereo = bool(seed & 2)
ref = po.ref_render(prg.ptr, rate, stereo, chunk=call)
po.oracle().ora_set_fastmath_forms(2)
ora = po.oracle_render(prg.ptr, rate, stereo, chunk=call)
d = np.nonzero(ora != ref)[0]
print("rate", rate, "call", call, "stereo", stereo, "samples", len(ref), "oracle vs reference differing:", len(d), d[:8].tolist(), flush=True)
if gpu:
    for env in ({}, {"SAU_AMD_NO_WIDE_TABS": "1"}, {"SAU_AMD_MORE_ROWS": "0"}, {"SAU_AMD_NO_LOOKBACK": "1"}, {"SAU_AMD_NO_SEQ": "1"}, {"SAU_AMD_NO_REPAIR": "1"},
                {"SAU_AMD_FAST_ROWS": "4"}, {"SAU_AMD_NO_DYN": "1"}, {"SAU_AMD_NO_FAST": "1"}):
        for k in ("SAU_AMD_NO_WIDE_TABS", "SAU_AMD_MORE_ROWS", "SAU_AMD_NO_LOOKBACK", "SAU_AMD_NO_SEQ", "SAU_AMD_NO_REPAIR", "SAU_AMD_FAST_ROWS", "SAU_AMD_NO_DYN", "SAU_AMD_NO_FAST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        b = sa.Batch([prg], rate); b.set_timing(2)
        got = b.render(stereo=stereo, chunk=call)[0]
        tm = b.timing_ex(); b.close()
        d = np.nonzero(got != ref)[0]
        print(env, "device vs reference differing:", len(d), d[:6].tolist(), [(int(got[i]), int(ref[i])) for i in d[:3]], "block_ms", round(tm["block_ms"], 3), flush=True)
