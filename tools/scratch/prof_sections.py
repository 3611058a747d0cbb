"""Wave-time per section of the running-sum build (FK_PROF variant; SAU_AMD_LIB=saugns_amd/variants/lib_prof.so)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
from saugns_amd.api import POP_PMOD, POP_FMOD
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
L = sa.lib()
names = ["step load", "common osc", "fvar block", "osc rest", "output", "other steps"]
def run(name, prg, frames=44100, steps=4):
    b = sa.Batch([prg], 44100)
    for _ in range(2): b.run(frames, fetch=False)
    b.sync()
    out = (C.c_uint * 16)()
    L.sauAmd_prof_read(out, 1)
    t0 = time.perf_counter()
    for _ in range(steps): b.run(frames, fetch=False)
    b.sync(); dt = time.perf_counter() - t0
    L.sauAmd_prof_read(out, 1)
    v = np.array(list(out), dtype=np.float64)
    print(f"{name}: {frames*steps/dt:.3e} frames/s")
    for base, label in ((0, "sum passes"), (8, "final pass")):
        tot = v[base:base+6].sum()
        if tot == 0: continue
        print(f"  {label}: " + "  ".join(f"{n} {100*v[base+i]/tot:.1f}%" for i, n in enumerate(names)) + f"  (total {tot*64/1e6:.1f} Mclk)")
voices = []
for i in range(1024):
    m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
    m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
    m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 20.0 + (i % 7) * 5), mods={POP_PMOD: [m2]})
    voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=30000, mods={POP_FMOD: [m1]}))
run("carrier FM", build_program(voices))
voices = []
for i in range(1024):
    m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1))
    voices.append(Op("sin", freq=Line(_num(".4f", 110.0 + i * 0.731), goal=_num(".3f", 220.0 + i * 0.5), shape="exp"),
                     time_ms=30000, mods={POP_PMOD: [m1]}))
run("carrier glide", build_program(voices))
