"""Timing experiments: frames/s for voice banks of different shape (GPU)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs)
def run(name, prg, frames=44100, steps=8):
    b = sa.Batch([prg], 44100)
    for _ in range(2): b.run(frames, fetch=False)
    b.sync(); b.timing(reset=True)
    lvl = int(os.environ.get("TLEVEL", "2")); b.set_timing(lvl)
    t0 = time.perf_counter()
    for _ in range(steps): b.run(frames, fetch=False)
    b.sync(); dt = time.perf_counter() - t0
    t = b.timing_ex()
    n = max(1, t["segments"])
    print(f"{name:28s} wall {dt/steps*1e3:7.3f} ms/step  fast {t['fast_ms']/n:6.3f}  block {t['block_ms']/n:6.3f}  mix {t['mix_ms']/n:6.3f}  aux {t['aux_ms']/n:6.3f} ms -> {frames*steps/dt:10.3e} frames/s")
which = sys.argv[1:] or ["c2", "c3", "c3x4"]
if "c2" in which: run("1024 x 1 op (flat)", vb.config2(n=1024, seconds=30))
if "c2b" in which: run("4096 x 1 op (flat)", vb.config2(n=4096, seconds=30))
if "c3" in which: run("1024 x 4 ops (config 3)", vb.config3(n=1024, seconds=30))
if "c3x4" in which: run("4096 x 4 ops", vb.config3(n=4096, seconds=30))
if "c5" in which: run("4096 x 2 ops selfmod (c5)", vb.config5(n=4096, seconds=30), frames=11025, steps=3)
if "c5n" in which:  # config 5 voices in other counts (C5N=<voices>,<voices>...)
    for n in [int(x) for x in os.environ.get("C5N", "3840,1920,960,240,15").split(",")]:
        run(f"{n} x 2 ops selfmod", vb.config5(n=n, seconds=30), frames=11025, steps=3)
if "c4" in which:
    G = os.path.join(ROOT, "tests", "golden", "programs")
    prgs = [sa.Program.from_image(open(os.path.join(G, f"config4_seed{k % 4}.saup"), "rb").read()) for k in range(64)]
    b = sa.Batch(prgs, 44100)
    frames, steps = 44100, 6
    for _ in range(2): b.run(frames, fetch=False)
    b.sync(); b.timing(reset=True); b.set_timing(int(os.environ.get("TLEVEL", "2")))
    t0 = time.perf_counter()
    for _ in range(steps): b.run(frames, fetch=False)
    b.sync(); dt = time.perf_counter() - t0
    t = b.timing_ex(); n = max(1, t["segments"])
    print(f"config 4: 64 renders        wall {dt/steps*1e3:7.3f} ms/step  segs/step {n/steps:.1f} fast {t['fast_ms']/steps:6.3f}  block {t['block_ms']/steps:6.3f}  mix {t['mix_ms']/steps:6.3f}  aux {t['aux_ms']/steps:6.3f} ms/step -> {64*frames*steps/dt:10.3e} frames/s total")
if "c3s" in which:
    from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
    from saugns_amd.api import POP_PMOD
    voices = []
    for i in range(1024):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=30000,
                         amp=Line(1.0, goal=_f32(0.2), shape="lin"), mods={POP_PMOD: [m1]}))
    run("1024 x 4 ops, amp sweep", build_program(voices), frames=44100, steps=4)
if "dropin" in which:
    # the drop-in API exactly as the reference host drives it (saugns.c:589-618): 11289-frame calls,
    # PCM copied to a host buffer every call
    for chunk in (11289, 44100, 176400):
        g = sa.Generator(vb.config3(n=1024, seconds=20), 44100)
        buf = np.zeros(chunk, np.int16)
        for _ in range(3):
            g.run(buf, chunk)
        t0 = time.perf_counter(); n = 0
        while n < 44100 * 12:
            more, got = g.run(buf, chunk); n += got
            if not more: break
        dt = time.perf_counter() - t0
        print(f"drop-in API, config 3, {chunk:6d}-frame calls: {n/dt:10.3e} frames/s ({dt/(n/chunk)*1e3:.3f} ms per call)")
        g.close()
if "file" in which:
    # output stage end to end: config 3, 20 s of audio into a WAV/AU file on tmpfs
    prg = vb.config3(n=1024, seconds=20)
    for fmt, name in ((2, "wav"), (1, "au")):
        path = f"/dev/shm/sau_amd_sweep.{name}"
        sa.render_file(prg, 44100, path, fmt, 1)  # warm
        t0 = time.perf_counter()
        n = sa.render_file(prg, 44100, path, fmt, 1)
        dt = time.perf_counter() - t0
        print(f"render_file {name}: {n} frames in {dt*1e3:.1f} ms -> {n/dt:10.3e} frames/s (create + render + copy + write)")
        os.remove(path)
if "front" in which:
    # what the host pays around the kernels -- generator creation (program
    # conversion, device allocation, state upload), the first call (all t=0 events applied, plans
    # built and uploaded) and a whole 10 s render through the drop-in API, create to destroy
    for name, mk in (("config 3", lambda: vb.config3(n=1024, seconds=10)),
                     ("config 5", lambda: vb.config5(n=4096, seconds=10))):
        for rep in range(2):
            prg = mk()
            buf = np.zeros(176400, np.int16)
            t0 = time.perf_counter()
            g = sa.Generator(prg, 44100)
            t1 = time.perf_counter()
            more, got = g.run(buf, 11289)
            t2 = time.perf_counter(); n = got
            while more:
                more, got = g.run(buf, 176400); n += got
            t3 = time.perf_counter()
            g.close()
            t4 = time.perf_counter()
            print(f"front-end {name} (pass {rep}): create {1e3*(t1-t0):.2f} ms, first call {1e3*(t2-t1):.2f} ms, "
                  f"remaining {n-11289} frames {1e3*(t3-t2):.2f} ms, destroy {1e3*(t4-t3):.2f} ms; "
                  f"whole job {n/(t4-t0):.3e} frames/s")
if "c3x" in which:
    from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
    from saugns_amd.api import POP_PMOD
    voices = []
    for i in range(1024):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4), time_ms=500)
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=30000, mods={POP_PMOD: [m1]}))
    run("1024 x 4 ops, deepest modulator expired", build_program(voices), frames=44100, steps=4)
if "c3f" in which:
    from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
    from saugns_amd.api import POP_PMOD, POP_FMOD
    voices = []
    for i in range(1024):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 20.0 + (i % 7) * 5), mods={POP_PMOD: [m2]})
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=30000, mods={POP_FMOD: [m1]}))
    run("1024 x 4 ops, carrier FM", build_program(voices), frames=44100, steps=4)
    voices = []
    for i in range(1024):
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1))
        voices.append(Op("sin", freq=Line(_num(".4f", 110.0 + i * 0.731), goal=_num(".3f", 220.0 + i * 0.5), shape="exp"),
                         time_ms=30000, mods={POP_PMOD: [m1]}))
    run("1024 x 2 ops, carrier glide", build_program(voices), frames=44100, steps=4)
if "fmstack" in which:
    # FM (vibrato) on the carrier and a 3-deep ratio PM stack under it: 5 operators, long steps
    from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
    from saugns_amd.api import POP_PMOD, POP_FMOD
    voices = []
    for i in range(1024):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("tri", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
        vib = Op("sin", freq=_num(".2f", 4.0 + (i % 9) * 0.5), amp=_num(".1f", 10.0 + i % 30))
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=60000, mods={POP_PMOD: [m1], POP_FMOD: [vib]}))
    run("1024 x 5 ops, FM + ratio PM stack", build_program(voices), frames=441000, steps=2)
if "mixed" in which:
    from saugns_amd.voicebank import Op, Line, build_program, _f32, _num
    from saugns_amd.api import POP_PMOD, POP_FMOD
    voices = []
    for i in range(1024):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
        kind = POP_FMOD if i % 16 == 0 else POP_PMOD   # one voice in sixteen is FM
        if kind == POP_FMOD: m1.amp = Line(30.0)
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=30000, mods={kind: [m1]}))
    run("1024 x 4 ops, 1/16 of the voices FM", build_program(voices), frames=44100, steps=4)
if "f4" in which:
    # what a whole job costs besides rendering. The reference's parser
    # (sau_build_Program through oracle/_ref, where that library is present), this repo's two parser-free
    # builders, generator creation, the first call (every t = 0 event applied), the rest of the render.
    from oracle import pyoracle as po
    scripts = vb.config_scripts()
    for name, mk, voices_of in (("config 3", lambda: vb.config3(n=1024, seconds=10), None),
                                ("config 5", lambda: vb.config5(n=4096, seconds=10), None)):
        key = "config3" if name == "config 3" else "config5"
        parse_ms = None
        if po.have_ref():
            po.ref()
            t0 = time.perf_counter()
            p = po.ref_build_program(scripts[key])
            parse_ms = 1e3 * (time.perf_counter() - t0)
            po.ref_discard_program(p)
        t0 = time.perf_counter(); prg = mk(); py_ms = 1e3 * (time.perf_counter() - t0)
        # the same bank through the C ABI's builder: flatten once (host's own data), then time the call
        import ctypes as C
        if key == "config3":
            from saugns_amd.voicebank import Op, Line, _f32, _num
            from saugns_amd.api import POP_PMOD
            voices = []
            for i in range(1024):
                m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
                m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
                m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
                voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=10000, mods={POP_PMOD: [m1]}))
            arr, n = vb.flatten(voices)
            t0 = time.perf_counter()
            for _ in range(10): q = sa.lib().sauAmd_build_bank(arr, n, 1.0, 1000); sa.lib().sauAmd_free_bank(q)
            c_ms = 1e3 * (time.perf_counter() - t0) / 10
        else:
            c_ms = float("nan")
        for rep in range(2):
            buf = np.zeros(176400, np.int16)
            t0 = time.perf_counter(); g = sa.Generator(prg, 44100); t1 = time.perf_counter()
            more, got = g.run(buf, 11289); t2 = time.perf_counter(); n = got
            while more:
                more, got = g.run(buf, 176400); n += got
            t3 = time.perf_counter(); g.close(); t4 = time.perf_counter()
        print(f"f-4 {name}: reference parser {parse_ms if parse_ms is None else round(parse_ms, 2)} ms | builders: Python {py_ms:.1f} ms, "
              f"sauAmd_build_bank {c_ms:.2f} ms | create {1e3*(t1-t0):.2f} ms, first call {1e3*(t2-t1):.2f} ms, rest of {n} frames "
              f"{1e3*(t3-t2):.2f} ms, destroy {1e3*(t4-t3):.2f} ms -> whole job without parsing {n/(t4-t0):.3e} frames/s"
              + (f", with the reference parser {n/(t4-t0+parse_ms/1e3):.3e}" if parse_ms else ""))
