"""ctypes access to the checkers -- TEST INFRASTRUCTURE ONLY.

Two libraries live under oracle/:

* ``liboracle.so``       this repo's scalar CPU restatement (oracle/sau_oracle.c)
* ``_ref/libsau_ref.so`` the compiled reference (parser + generator), built
  from /root/reference by oracle/Makefile in the build container.  It is the
  only way to turn SAU *script text* into a ``sauProgram`` here; fixtures made
  with it are committed under tests/golden/.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may
import this module; nothing in saugns_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsau_ref.so")


def build(ref=True):
    """Compile liboracle.so and, when the reference sources are present, _ref/."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class _Predef(C.Structure):
    _fields_ = [("key", C.c_char_p), ("len", C.c_uint32), ("val", C.c_double)]


class _ScriptArg(C.Structure):
    # sau/script.h:135-141: str, two 1-bit flags, predef array
    _fields_ = [("str", C.c_char_p), ("bits", C.c_uint8),
                ("predef", C.POINTER(_Predef)), ("predef_count", C.c_size_t)]


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        lib = C.CDLL(ORACLE_SO)
        lib.ora_create.restype = C.c_void_p
        lib.ora_create.argtypes = [C.c_void_p, C.c_uint32]
        lib.ora_run.restype = C.c_bool
        lib.ora_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_bool,
                                C.POINTER(C.c_size_t)]
        lib.ora_destroy.argtypes = [C.c_void_p]
        lib.ora_set_block_len.argtypes = [C.c_void_p, C.c_uint32]
        lib.ora_set_piluts.argtypes = [C.c_void_p]
        lib.ora_get_piluts.restype = C.POINTER(C.c_float)
        lib.ora_set_fastmath_forms.argtypes = [C.c_int]
        lib.ora_herp.restype = C.c_double
        lib.ora_herp.argtypes = [C.c_int, C.c_uint32]
        lib.ora_ramp_fill.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_float,
                                      C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.ora_ramp_map.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_void_p]
        lib.ora_wosc_kat.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t]
        lib.ora_noise_kat.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.ora_rasg_kat.argtypes = [C.c_uint32, C.c_int, C.c_uint, C.c_uint, C.c_uint,
                                     C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ora_franssgauss32.restype = C.c_float
        lib.ora_franssgauss32.argtypes = [C.c_uint32]
        lib.ora_ranfast32.restype = C.c_uint32
        lib.ora_ranfast32.argtypes = [C.c_uint32]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.sau_build_Program.restype = C.c_void_p
        lib.sau_build_Program.argtypes = [C.POINTER(_ScriptArg)]
        lib.sau_discard_Program.argtypes = [C.c_void_p]
        lib.sau_create_Generator.restype = C.c_void_p
        lib.sau_create_Generator.argtypes = [C.c_void_p, C.c_uint32]
        lib.sauGenerator_run.restype = C.c_bool
        lib.sauGenerator_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_bool,
                                         C.POINTER(C.c_size_t)]
        lib.sau_destroy_Generator.argtypes = [C.c_void_p]
        lib.sau_global_init_Wave.argtypes = []
        for name in ("sauLine_fill_funcs", "sauLine_map_funcs"):
            pass
        _ref = lib
    return _ref


def ref_build_program(script, is_path=False, predefs=None, deterministic=True):
    """Parse SAU text (or a file) with the reference parser -> sauProgram* (int)."""
    lib = ref()
    predefs = predefs or {}
    arr = (_Predef * max(1, len(predefs)))()
    keep = []
    for i, (k, v) in enumerate(predefs.items()):
        kb = k.encode()
        keep.append(kb)
        arr[i] = _Predef(kb, len(kb), float(v))
    s = script.encode() if isinstance(script, str) else script
    arg = _ScriptArg(s, (1 if is_path else 0) | (2 if deterministic else 0),
                     arr if predefs else None, len(predefs))
    p = lib.sau_build_Program(C.byref(arg))
    return p


def ref_discard_program(p):
    ref().sau_discard_Program(p)


def ref_piluts():
    """The reference's twelve PILUT arrays as a (12, 2048) float32 array."""
    lib = ref()
    lib.sau_global_init_Wave()
    tabs = (C.POINTER(C.c_float) * 12).in_dll(lib, "sauWave_piluts")
    return np.stack([np.ctypeslib.as_array(tabs[i], shape=(2048,)).copy()
                     for i in range(12)])


def _render(create, run, destroy, prg, srate, stereo, chunk, max_frames, post_create=None):
    g = create(prg, srate)
    if not g:
        raise RuntimeError("generator creation failed")
    if post_create:
        post_create(g)
    ch = 2 if stereo else 1
    buf = np.zeros(chunk * ch, dtype=np.int16)
    n = C.c_size_t()
    out = []
    total = 0
    try:
        while True:
            more = run(g, buf.ctypes.data, chunk, stereo, C.byref(n))
            out.append(buf[: n.value * ch].copy())
            total += n.value
            if not more or (max_frames and total >= max_frames):
                break
    finally:
        destroy(g)
    pcm = np.concatenate(out) if out else np.zeros(0, np.int16)
    if max_frames:
        pcm = pcm[: max_frames * ch]
    return pcm


def ref_render(prg, srate=44100, stereo=False, chunk=11289, max_frames=0):
    lib = ref()
    return _render(lib.sau_create_Generator, lib.sauGenerator_run,
                   lib.sau_destroy_Generator, prg, srate, stereo, chunk, max_frames)


def oracle_render(prg, srate=44100, stereo=False, chunk=11289, max_frames=0, block_len=0):
    lib = oracle()
    post = (lambda g: lib.ora_set_block_len(g, block_len)) if block_len else None
    return _render(lib.ora_create, lib.ora_run, lib.ora_destroy, prg, srate, stereo,
                   chunk, max_frames, post)


def oracle_sndfile_bytes(fmt, channels, srate, pcm):
    """Restatement of the reference's sound file writer (player/sndfile.c): the bytes of a
    raw (0) / AU (1) / WAV (2) file holding interleaved int16 ``pcm``.

    AU: ".snd", header size 28, size field, encoding 3, rate, channels, 4 zero bytes, all
    big-endian (63-72), samples byte-swapped (160-168); on close the size field receives the
    FRAME count (74-80: it stores o->samples, not a byte count) unless that is >= 2^32-1.
    WAV: canonical 44-byte PCM header, little-endian (82-99), RIFF size 36 + bytes and data
    size = channels * frames * 2 as uint32 (101-109)."""
    import struct
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    frames = len(pcm) // channels
    if fmt == 0:
        return pcm.astype("<i2").tobytes()
    if fmt == 1:
        size = frames if frames < 0xFFFFFFFF else 0xFFFFFFFF
        return b".snd" + struct.pack(">IIIIII", 28, size, 3, srate, channels, 0) + pcm.astype(">i2").tobytes()
    nbytes = (channels * frames * 2) & 0xFFFFFFFF
    return (b"RIFF" + struct.pack("<I", (36 + nbytes) & 0xFFFFFFFF) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, channels, srate, channels * srate * 2, channels * 2, 16) +
            b"data" + struct.pack("<I", nbytes) + pcm.astype("<i2").tobytes())


def ref_write_sndfile(path, fmt, channels, srate, pcm, chunk=11289):
    """Write ``pcm`` with the reference's own writer (player/sndfile.c in oracle/_ref)."""
    lib = ref()
    lib.SGS_create_SndFile.restype = C.c_void_p
    lib.SGS_create_SndFile.argtypes = [C.c_char_p, C.c_uint, C.c_uint16, C.c_uint32]
    lib.SGS_SndFile_write.restype = C.c_bool
    lib.SGS_SndFile_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.SGS_close_SndFile.argtypes = [C.c_void_p]
    f = lib.SGS_create_SndFile(os.fsencode(path), fmt, channels, srate)
    if not f:
        raise RuntimeError("SGS_create_SndFile failed")
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    frames = len(pcm) // channels
    for i in range(0, frames, chunk):
        part = pcm[i * channels:(i + chunk) * channels].copy()  # the writer swaps in place
        lib.SGS_SndFile_write(f, part.ctypes.data, len(part) // channels)
    lib.SGS_close_SndFile(f)


def oracle_use_tables(tables):
    """Give the oracle the reference's PILUTs ((12, 2048) float32)."""
    t = np.ascontiguousarray(tables, dtype=np.float32)
    assert t.shape == (12, 2048)
    oracle().ora_set_piluts(t.ctypes.data)
