"""Debug aid: render one hand-built voice bank on the GPU and compare with the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_AMOD, POP_RAMOD, POP_FMOD, POP_RFMOD, POP_PMOD
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
def case(name, voices):
    prg = vb.build_program(voices)
    want = po.oracle_render(prg.ptr, 44100, False)
    got = sa.Generator(prg, 44100).render()
    d = np.nonzero(got != want)[0]
    print(name, len(got), len(want), "diffs", len(d), "first", d[:3], got[:6], want[:6])
a1 = vb.Op("sin", freq=7.0, amp=0.3)
case("am", [vb.Op("sin", freq=300.0, amp=0.8, time_ms=120, mods={POP_AMOD: [a1]})])
r1 = vb.Op("tri", freq=3.0, amp=1.0)
case("ram", [vb.Op("sin", freq=300.0, amp=0.8, amp2=0.1, time_ms=120, mods={POP_RAMOD: [r1]})])
case("plain", [vb.Op("sin", freq=300.0, amp=0.8, time_ms=120)])
