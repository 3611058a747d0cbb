import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_PMOD, POP_RAMOD, POP_APMOD
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
cases = {
 "const freq": [vb.Op("sin", freq=200.0, time_ms=100, pm_a=0.5)],
 "freq ramp": [vb.Op("sin", freq=vb.Line(200.0, goal=400.0, shape="exp"), time_ms=100, pm_a=0.5)],
 "pma ramp": [vb.Op("sin", freq=200.0, time_ms=100, pm_a=vb.Line(0.5, goal=0.1, shape="lin"))],
 "two voices": [vb.Op("sin", freq=200.0, time_ms=100, pm_a=0.5), vb.Op("tri", freq=300.0, time_ms=80, pm_a=0.3)],
 "with pm": [vb.Op("sin", freq=200.0, time_ms=100, pm_a=0.5, mods={POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.3)]})],
 "as modulator": [vb.Op("sin", freq=200.0, time_ms=100, mods={POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.3, pm_a=0.4)]})],
 "apmods": [vb.Op("sin", freq=200.0, time_ms=100, pm_a=0.2, mods={POP_APMOD: [vb.Op("sin", freq=5.0, amp=0.3)]})],
 "config5 x4": None,
}
for name, voices in cases.items():
    prg = vb.config5(n=4, seconds=1) if voices is None else vb.build_program(voices)
    want = po.oracle_render(prg.ptr, 44100, False, chunk=100000)
    got = sa.Batch([prg], 44100).render(stereo=False, chunk=100000)[0]
    d = np.nonzero(got != want)[0] if len(got) == len(want) else [-1]
    print(f"{name:14s} len {len(got)} {len(want)}: {len(d)} differ" + (f", first at {d[0]}: got {got[d[0]:d[0]+5].tolist()} want {want[d[0]:d[0]+5].tolist()}" if len(d) else ""))
