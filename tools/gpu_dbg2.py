import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
which = sys.argv[1]
if which == "a":
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="lin"))
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_PMOD: [m]})]
elif which == "b":  # modulator amp sweep, modulator used as AM instead of PM
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="lin"))
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_AMOD: [m]})]
elif which == "c":
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="xpe"))
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_PMOD: [m]})]
elif which == "d":
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="xpe"), time_ms=100)
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_PMOD: [m]})]
elif which == "e":
    voices = [vb.Op("sin", freq=146.832, amp=vb.Line(0.5, goal=2.0, shape="xpe"), time_ms=1000)]
elif which == "f":
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="smo"))
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_PMOD: [m]})]
elif which == "g":
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.5, goal=2.0, shape="cos"))
    voices = [vb.Op("sin", freq=146.832, time_ms=20, mods={POP_PMOD: [m]})]
prg = vb.build_program(voices)
want = po.oracle_render(prg.ptr, 44100, False)
got = sa.Generator(prg, 44100).render(chunk=500)
d = np.nonzero(got != want)[0]
print(which, "ndiff", len(d), "first", d[:3], "got", got[:8].tolist(), "want", want[:8].tolist())
