"""Do the hand-made banks of tests/test_gpu_vs_ref.py::test_the_mixer_takes_no_chunk_before_its_rows_are_written show the
defect on a library that has it?  SAU_AMD_LIB=saugns_amd/variants/lib_zmin.so python tests/tools/debug_pin_teeth.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_FMOD, POP_PMOD, POPT_RASEG
from oracle import pyoracle as po
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
nine = vb.Op("sin", freq=vb.Line(200.0, goal=260.0, shape="lin"), time_ms=160, amp=0.5)
cur = nine
for k in range(8):
    m = vb.Op("tri", freq=vb.Line(3.0 + k, goal=5.0 + k, shape="lin"), amp=4.0)
    cur.mods = {POP_FMOD: [m]}
    cur = m
fbv = vb.Op("sin", freq=vb.Line(150.0, goal=300.0, shape="exp"), time_ms=160, pm_a=0.4, amp=0.5)
rcub = vb.Op("sin", freq=220.0, time_ms=160, amp=0.5,
             mods={POP_PMOD: [vb.Op(op_type=POPT_RASEG, ras=("cub", 0, 0), seed=77, freq=30.0, amp=0.6)]})
for name, prg, stereo, call in (("nine sums", vb.build_program([fbv, nine]), False, 6800), ("R cub", vb.build_program([fbv, rcub]), True, 5600)):
    ref = po.ref_render(prg.ptr, 44100, stereo, chunk=call)
    g = sa.Generator(prg, 44100); got = g.render(stereo=stereo, chunk=call); g.close()
    d = np.nonzero(got[:len(ref)] != ref[:len(got)])[0]
    print(name, "len", len(got), len(ref), "differing", len(d), d[:4].tolist())
