"""Random operator graphs in banks of hundreds of voices, with later events, GPU against the oracle:
the random-program suites of tests/test_gpu_units.py have 1-3 voices per program (64 waves per voice,
look-back through HBM words); these banks put the same graphs at 4-13 waves per voice (rings in LDS, voices
across two workgroups) and at one wave per voice.   python tests/tools/gpu_random_soak.py [first_seed [n_seeds]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.build(ref=False); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1 if os.environ.get("SAU_AMD_LOOP_TAILS") == "0" else 2)  # the product default reproduces the loop tails: mode 2
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for seed in range(first, first + count):
    for n in (300, 700, 1024, 2300):
        if n > 1024 and seed % 4: continue
        rng = np.random.default_rng(900000 + 1000 * n + seed)
        voices = [T._random_voice(rng) for _ in range(n)]
        if os.environ.get("SOAK_EXTREME"):  # parameters pushed to extremes (tests/test_gpu_units.py: _push_extremes)
            T._push_extremes(rng, voices)
        T._random_starts(rng, voices)
        ups = T._random_updates(rng, voices[:40])
        prg = vb.build_program(voices, updates=ups)
        stereo = bool(seed & 1)
        chunk = int(rng.integers(2000, 9000)) if seed % 3 else 4000000
        t0 = time.perf_counter()
        want = po.oracle_render(prg.ptr, 44100, stereo, chunk=chunk)
        t1 = time.perf_counter()
        b = sa.Batch([prg], 44100); b.set_timing(2)
        got = b.render(stereo=stereo, chunk=chunk)[0]
        t = b.timing_ex()
        ok = len(got) == len(want) and (got == want).all()
        bad += not ok
        print(f"seed {seed} voices {n} chunk {chunk}: {'ok' if ok else 'DIFFERS'}  oracle {t1-t0:.1f} s, fast {t['fast_ms']:.1f} ms, block loop {t['block_ms']:.1f} ms", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
