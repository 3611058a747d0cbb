"""Experiment (GPU): chain_kernel's waves, where their time goes (a -DCHAIN_PROF build: saugns_amd/variants/lib_chainprof.so).
SAU_AMD_LIB=saugns_amd/variants/lib_chainprof.so python tests/tools/exp_c5_prof.py [freq_ramp pm_a_ramp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SAU_AMD_TUNE"] = "1"
import numpy as np
import saugns_amd as sa
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_RAMOD
fr, pr = (sys.argv[1] == "1", sys.argv[2] == "1") if len(sys.argv) > 2 else (True, True)
vs = []
for i in range(4096):
    lfo = vb.Op("sin", freq=float(3 + i % 9), amp=1.0)
    f0 = 80.0 + i * 0.211
    vs.append(vb.Op("sin", freq=vb.Line(f0, goal=160.0 + i * 0.1, shape="exp") if fr else f0,
                    pm_a=vb.Line(0.3 + (i % 8) * 0.1, goal=0.1, shape="lin") if pr else 0.3 + (i % 8) * 0.1,
                    amp=vb.Line(1.0, goal=0.2, shape="xpe"), amp2=vb.Line(0.2), time_ms=10000, mods={POP_RAMOD: [lfo]}))
prg = vb.build_program(vs)
b = sa.Batch([prg], 44100)
b.run(441000, fetch=False); b.sync(); b.close()
