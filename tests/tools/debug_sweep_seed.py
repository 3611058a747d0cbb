"""Debug aid: one seed of tests/tools/gpu_vs_ref_sweep.py, rendered every way -- compiled reference, drop-in generator (default
read-ahead, and one run per call), batch API at the same call size -- with the positions of differing samples.
    python tests/tools/debug_sweep_seed.py <seed> [<seed> ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(20000 + seed)
    voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    ups = ()
    if seed % 2:
        T._random_starts(rng, voices)
        ups = T._random_updates(rng, voices)
    prg = vb.build_program(voices, updates=ups)
    call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
    stereo = bool(seed & 2)
    rate = 44100 if seed % 3 else int(rng.choice([8000, 22050, 48000, 96000]))
    ref = po.ref_render(prg.ptr, rate, stereo, chunk=call)
    def cmp(name, got):
        d = np.nonzero(got[:len(ref)] != ref[:len(got)])[0] if len(got) and len(ref) else np.zeros(0, int)
        print(f"seed {seed} call {call} rate {rate} stereo {stereo} {name}: len {len(got)}/{len(ref)} differing {len(d)} first {d[:6].tolist()} last {d[-3:].tolist()}", flush=True)
    g = sa.Generator(prg, rate); cmp("drop-in default", g.render(stereo=stereo, chunk=call)); g.close()
    os.environ["SAU_AMD_READAHEAD"] = "0"
    g = sa.Generator(prg, rate); cmp("drop-in, one run per call", g.render(stereo=stereo, chunk=call)); g.close()
    del os.environ["SAU_AMD_READAHEAD"]
    for ra in ("1000000", str(call * 3)):
        os.environ["SAU_AMD_READAHEAD"] = ra
        g = sa.Generator(prg, rate); cmp(f"drop-in, read-ahead {ra}", g.render(stereo=stereo, chunk=call)); g.close()
        del os.environ["SAU_AMD_READAHEAD"]
    os.environ["SAU_AMD_READAHEAD_RAMP"] = "0"
    g = sa.Generator(prg, rate); cmp("drop-in, no ramp", g.render(stereo=stereo, chunk=call)); g.close()
    del os.environ["SAU_AMD_READAHEAD_RAMP"]
    b = sa.Batch([prg], rate); cmp("batch, runs of the call size", b.render(stereo=stereo, chunk=call)[0]); b.close()
    b = sa.Batch([prg], rate); b.set_call_len(call); cmp("batch, one run of 40 calls", b.render(stereo=stereo, chunk=call * 40)[0]); b.close()
    os.environ["SAU_AMD_TUNE"] = "1"; os.environ["SAU_AMD_NO_FAST"] = "1"
    g = sa.Generator(prg, rate); cmp("drop-in default, block loop only", g.render(stereo=stereo, chunk=call)); g.close()
    del os.environ["SAU_AMD_NO_FAST"]
