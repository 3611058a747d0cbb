"""Mutation fuzz of the host control plane's untrusted-input entry points (run under ASan + UBSan by tests/test_sanitizers.py):
    python tests/tools/fuzz_host_inputs.py [mutations [seed]]
* sauAmd_program_load on mutated program images (bytes flipped / overwritten / truncated / spliced from the corpus images and
  generated banks): it must either reject the image (NULL) or hand back a program -- which is then serialized again and, every
  few hundred images, rendered for a bounded number of frames through the engine over the sequential executor;
* sauAmd_build_bank on mutated operator descriptions (fields overwritten with boundary values): NULL or a program, which is
  then rendered likewise.
Nothing is compared: what is looked for are memory errors and undefined behaviour in program_io.cpp, bank_builder.cpp,
engine.cpp and plan.cpp. Prints one JSON line."""
import ctypes as C
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import saugns_amd as sa
from saugns_amd import api, voicebank as vb

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
L = sa.lib()
api.use_hooks(os.environ.get("SAU_HOOKS_LIB") or os.path.join(ROOT, "tests", "hooks", "libsaugns_amd_hooks.so"))
seqlib = C.CDLL(os.environ.get("SAU_SEQEXEC_LIB") or os.path.join(ROOT, "tests", "seqexec", "libseqexec.so"))
seqlib.seq_backend_create.restype = C.c_void_p
seqlib.seq_backend_create.argtypes = [C.c_uint32]
images = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "programs", "*.saup")))
          if os.path.getsize(f) < 20000][:60]
_bank = vb.config3(n=3, seconds=1)
images.append(sa.Program.borrow(_bank.ptr, owner=_bank).image())
S = {"image_mutations": 0, "images_accepted": 0, "bank_mutations": 0, "banks_accepted": 0, "rendered": 0, "frames": 0, "runs_refused": 0}


def render_some(ptr, owner):
    """a bounded render of an accepted program: 3 calls of 700 frames, mono then stereo"""
    prg = sa.Program.borrow(ptr, owner=owner)
    for stereo in (False, True):
        try:
            b = sa.Batch([prg], 44100, backend=seqlib.seq_backend_create(333))
        except RuntimeError:
            return  # (the engine refused the program: a message, no crash)
        for _ in range(3):
            try:
                pcm, more, lens = b.run(700, stereo)
            except RuntimeError:
                S["runs_refused"] += 1  # (false + a message: an event naming an operator that does not exist, ...)
                break
            S["frames"] += int(lens[0])
            if not more[0]:
                break
        del b
    S["rendered"] += 1


BOUND = np.array([0, 1, 2, 3, 7, 8, 15, 16, 0x7F, 0x80, 0xFF, 0x100, 0x7FFF, 0x8000, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000,
                  0xFFFFFFFF, 0xFFFFFFFE], dtype=np.uint64)
for it in range(N):
    img = bytearray(images[int(rng.integers(len(images)))])
    kind = int(rng.integers(6))
    if kind == 0:    # flip a few bits
        for _ in range(int(rng.integers(1, 5))):
            img[int(rng.integers(len(img)))] ^= 1 << int(rng.integers(8))
    elif kind == 1:  # overwrite an aligned 32-bit word with a boundary value
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(len(img) // 4)) * 4
            img[at:at + 4] = int(BOUND[int(rng.integers(len(BOUND)))] & 0xFFFFFFFF).to_bytes(4, "little")
    elif kind == 2:  # truncate
        img = img[: int(rng.integers(0, len(img)))]
    elif kind == 3:  # random bytes over a span
        at = int(rng.integers(len(img))); n = int(rng.integers(1, 64))
        img[at:at + n] = rng.integers(0, 256, min(n, len(img) - at), dtype=np.uint8).tobytes()
    elif kind == 4:  # splice the tail of another image in
        other = images[int(rng.integers(len(images)))]
        at = int(rng.integers(len(img)))
        img = img[:at] + other[at:]
    else:            # overwrite a 64-bit word (offsets and counts of the image are 64-bit)
        at = int(rng.integers(max(1, len(img) // 8))) * 8
        img[at:at + 8] = int(BOUND[int(rng.integers(len(BOUND)))]).to_bytes(8, "little")
    buf = bytes(img)
    S["image_mutations"] += 1
    p = L.sauAmd_program_load(buf, len(buf))
    if p:
        S["images_accepted"] += 1
        n = L.sauAmd_program_serialize(p, None, 0)
        out = C.create_string_buffer(n)
        assert L.sauAmd_program_serialize(p, out, n) == n
        if S["images_accepted"] % 2 == 0:
            render_some(p, None)
        L.sauAmd_program_free(p)

# operator descriptions for sauAmd_build_bank
base_voices = [vb.Op("sin", freq=200.0 + 10 * k, amp=0.5, time_ms=20 + k,
                     mods={api.POP_PMOD: [vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.7,
                                                mods={api.POP_FMOD: [vb.Op("sin", freq=5.0, amp=3.0)]})],
                           api.POP_AMOD: [vb.Op("sin", freq=3.0, amp=0.2)]}) for k in range(4)]
arr0, n0 = vb.flatten(base_voices)
if True:
    raw0 = bytes(C.string_at(C.addressof(arr0), C.sizeof(arr0)))
    rec = C.sizeof(arr0) // n0
    for it in range(N // 4):
        raw = bytearray(raw0)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(len(raw) // 4)) * 4
            raw[at:at + 4] = int(BOUND[int(rng.integers(len(BOUND)))] & 0xFFFFFFFF).to_bytes(4, "little")
        n_ops = n0 if rng.random() < 0.8 else int(rng.integers(0, n0 + 1))
        cbuf = C.create_string_buffer(bytes(raw), len(raw))
        S["bank_mutations"] += 1
        p = L.sauAmd_build_bank(cbuf, n_ops, C.c_float(float(rng.choice([1.0, 0.0, -1.0, 1e30]))), int(rng.choice([0, 1, 1000, 0xFFFFFFFF])))
        if p:
            S["banks_accepted"] += 1
            if S["banks_accepted"] % 4 == 0:
                render_some(p, None)
            L.sauAmd_free_bank(p)
print(json.dumps(S))
