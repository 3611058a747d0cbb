"""Experiment (r03): what each of rainy_thunder.sau's two voices costs alone, 64 renders per batch, 60 s each
(needs oracle/_ref/libsau_ref.so for the reference's parser)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
po.ref()
rain = "Rlin mg f12.5 p[Rcos mg rpi(10) a10^2 Wsin f20 a10^4] a1/2 t60"
thunder = "Wsin f-50.r+50[Rlin r1.r20[Wsin f1/10]] a0.r1[Wsin f3/20] t60"
for name, script in (("rain", rain), ("thunder", thunder), ("both", rain + "\n" + thunder)):
    ps = [po.ref_build_program(script, predefs={"seed": k}) for k in range(64)]
    prgs = [sa.Program.borrow(p) for p in ps]
    for rep in range(3):
        b = sa.Batch(prgs, 44100); b.set_call_len(11289); b.set_timing(2)
        t0 = time.perf_counter()
        b.run(2646000, fetch=False); b.sync()
        dt = time.perf_counter() - t0
        t = b.timing_ex(); b.close()
    print(f"{name:8s}: {dt*1e3:7.2f} ms per 64 renders  fast {t['fast_ms']:.2f} mix {t['mix_ms']:.2f} aux {t['aux_ms']:.2f} segs {t['segments']}")
