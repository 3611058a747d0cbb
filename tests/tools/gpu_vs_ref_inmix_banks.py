"""GPU against the compiled reference on closed-form banks large and long enough for the XCD task queues and the launch that mixes
its own rows: 64-400 voices of one nesting depth each bank (so that the launch may mix), PM chains of depth 1-4 with
wave types, ratios, amplitudes and pans drawn per voice, 3-9 s, mono or stereo, one whole-script run or a few uneven ones, launches of
8-48 workgroups. SAU_AMD_INMIX_REPORT is on: the summary says how many banks the launch mixed tiles of.
    python tests/tools/gpu_vs_ref_inmix_banks.py [first_seed [banks]]"""
import json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["SAU_AMD_TUNE"] = "1"
os.environ["SAU_AMD_INMIX_REPORT"] = "1"
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import *
from oracle import pyoracle as po
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def bank(rng, n, seconds, depth):
    voices = []
    for i in range(n):
        op = None
        for d in range(depth - 1):
            op = vb.Op(WAVES[int(rng.integers(len(WAVES)))], freq=vb.Line(float(rng.integers(1, 7)), ratio=True), amp=vb._num(".2f", rng.uniform(0.1, 1.2)),
                       phase=float(rng.uniform(0, 1)), mods={POP_PMOD: [op]} if op else None)
        voices.append(vb.Op(WAVES[int(rng.integers(len(WAVES)))], freq=vb._num(".3f", rng.uniform(40.0, 1500.0)), time_ms=seconds * 1000,
                            amp=vb._num(".2f", rng.uniform(0.2, 1.0)), pan=vb.Line(vb._num(".2f", rng.uniform(0.0, 1.0))),
                            mods={POP_PMOD: [op]} if op else None))
    return vb.build_program(voices)


class Capture:
    """stderr of the library (the report lines) while a bank renders"""
    def __enter__(self):
        sys.stderr.flush()
        self.old = os.dup(2); self.r, self.w = os.pipe(); os.dup2(self.w, 2); return self
    def __exit__(self, *a):
        sys.stderr.flush(); os.dup2(self.old, 2); os.close(self.w); os.close(self.old)
        self.text = b""
        os.set_blocking(self.r, False)
        try:
            while True:
                b = os.read(self.r, 65536)
                if not b: break
                self.text += b
        except BlockingIOError:
            pass
        os.close(self.r)


bad = mixed = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(770000 + seed)
    n, seconds, depth = int(rng.integers(64, 401)), int(rng.integers(3, 10)), int(rng.integers(1, 5))
    stereo = bool(rng.integers(0, 2))
    frames = seconds * 44100
    call = frames if rng.random() < 0.6 else int(rng.integers(frames // 3, frames))
    os.environ["SAU_AMD_FK_GRID"] = str(int(rng.integers(8, 49)))
    prg = bank(rng, n, seconds, depth)
    t0 = time.time(); ref = po.ref_render(prg.ptr, 44100, stereo, chunk=call); t1 = time.time()
    with Capture() as cap:
        b = sa.Batch([prg], 44100); got = b.render(stereo=stereo, chunk=call)[0]; b.close()
    t2 = time.time()
    tiles = sum(int(x) for x in re.findall(rb"tiles (\d+) of", cap.text))
    mixed += tiles > 0
    same = len(got) == len(ref) and bool((got == ref).all())
    bad += not same
    print(f"seed {seed}: {n} voices depth {depth}, {seconds} s {'stereo' if stereo else 'mono'}, call {call}, grid {os.environ['SAU_AMD_FK_GRID']}: tiles mixed inside {tiles}; "
          f"reference {t1-t0:.1f} s, GPU {t2-t1:.2f} s,", "identical" if same else "DIFFERS", flush=True)
print(json.dumps({"banks": count, "identical": count - bad, "banks_the_launch_mixed_tiles_of": mixed, "first_seed": 770000 + first}))
sys.exit(1 if bad else 0)
