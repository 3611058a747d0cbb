"""Experiment (GPU): runs of repeated phases (a carrier of frequency 0 under PM from a sample-and-hold R oscillator whose value
stands for two or three frames at a time) -- every render must equal the oracle's, whether the voices stay on the time-parallel path
or bail to the block loop (block_ms says which).  python tests/tools/exp_hold_runs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SAU_AMD_TUNE"]="1"
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
po.build(ref=False); po.oracle_use_tables(tabs); sa.set_piluts(tabs)
po.oracle().ora_set_fastmath_forms(2)
for n, rf in ((64, 14700.0), (64, 11025.0), (300, 14700.0), (64, 22050.0)):
    vs = []
    for i in range(n):
        r = vb.Op(op_type=POPT_RASEG, ras=("sah", 0, 0), seed=1234 + i, freq=rf, amp=0.3 + 0.001 * i)
        vs.append(vb.Op("sin", freq=0.0, amp=0.5, time_ms=3000, phase=0.1 * (i % 10), mods={POP_PMOD: [r]}))
    prg = vb.build_program(vs)
    want = po.oracle_render(prg.ptr, 44100, False, chunk=50000)
    b = sa.Batch([prg], 44100); b.set_timing(2)
    got = b.render(stereo=False, chunk=50000)[0]
    tm = b.timing_ex(); b.close()
    d = np.nonzero(got != want)[0]
    print(n, rf, "differing", len(d), "of", len(want), "first", d[:5].tolist(), "block_ms", round(tm["block_ms"], 3), flush=True)
