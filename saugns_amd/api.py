"""ctypes binding of libsaugns_amd.so -- the same calls a C host would make.

``Generator`` mirrors the reference interface for this path
(``sau_create_Generator`` / ``sauGenerator_run`` / ``sau_destroy_Generator``,
sau/generator.h:17-26): same names, argument meaning and error behaviour
(``None``/``RuntimeError`` where the C function returns NULL).  ``Batch`` is
the multi-program extension.  Nothing here computes audio: without the HIP
library and a GPU, creation fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


class SauLine(C.Structure):
    _fields_ = [("v0", C.c_float), ("vt", C.c_float), ("pos", C.c_uint32),
                ("end", C.c_uint32), ("time_ms", C.c_uint32), ("type", C.c_uint8),
                ("flags", C.c_uint8)]


class SauTime(C.Structure):
    _fields_ = [("v_ms", C.c_uint32), ("flags", C.c_uint8)]


class SauRasOpt(C.Structure):
    _fields_ = [("word", C.c_uint32), ("alpha", C.c_uint32)]


class SauMode(C.Union):
    _fields_ = [("main", C.c_uint8), ("ras", SauRasOpt)]


class SauOpData(C.Structure):
    _fields_ = [("id", C.c_uint32), ("params", C.c_uint32), ("time", SauTime),
                ("pan", C.POINTER(SauLine)), ("amp", C.POINTER(SauLine)),
                ("amp2", C.POINTER(SauLine)), ("freq", C.POINTER(SauLine)),
                ("freq2", C.POINTER(SauLine)), ("pm_a", C.POINTER(SauLine)),
                ("phase", C.c_uint32), ("seed", C.c_uint32), ("use_type", C.c_uint8),
                ("type", C.c_uint8), ("mode", SauMode),
                ("camods", C.c_void_p), ("amods", C.c_void_p), ("ramods", C.c_void_p),
                ("fmods", C.c_void_p), ("rfmods", C.c_void_p), ("pmods", C.c_void_p),
                ("apmods", C.c_void_p), ("fpmods", C.c_void_p)]


class SauEvent(C.Structure):
    _fields_ = [("wait_ms", C.c_uint32), ("vo_id", C.c_uint16),
                ("carr_op_id", C.c_uint32), ("op_count", C.c_uint32),
                ("op_data_count", C.c_uint32), ("op_list", C.c_void_p),
                ("op_data", C.POINTER(SauOpData))]


class SauProgram(C.Structure):
    _fields_ = [("events", C.POINTER(SauEvent)), ("ev_count", C.c_size_t),
                ("mode", C.c_uint16), ("vo_count", C.c_uint16), ("op_count", C.c_uint32),
                ("op_nest_depth", C.c_uint8), ("duration_ms", C.c_uint32),
                ("ampmult", C.c_float), ("name", C.c_char_p), ("mp", C.c_void_p),
                ("parse", C.c_void_p)]


assert C.sizeof(SauLine) == 24 and C.sizeof(SauOpData) == 152
assert C.sizeof(SauEvent) == 40 and C.sizeof(SauProgram) == 64

# enum values of include/sau_abi.h
LINES = "cos lin sah exp log xpe lge sqe cub smo ncl nhl uwh".split()
WAVES = "sin tri srs sqr ean cat eto par mto saw hsi spa".split()
NOISES = "wh gw bw tw re vi bv".split()
LP_STATE, LP_STATE_RATIO, LP_GOAL, LP_GOAL_RATIO, LP_TYPE, LP_TIME, LP_TIME_IF_NEW = \
    1, 2, 4, 8, 16, 32, 64
POPT_AMP, POPT_NOISE, POPT_WAVE, POPT_RASEG = 0, 1, 2, 3
POP_CARR, POP_CAMOD, POP_AMOD, POP_RAMOD, POP_FMOD, POP_RFMOD, POP_PMOD, POP_APMOD, \
    POP_FPMOD = range(9)
TIMEP_SET, TIMEP_DEFAULT, TIMEP_IMPLICIT = 1, 2, 4
PMODE_AMP_DIV_VOICES = 1


class LineDesc(C.Structure):
    _fields_ = [("present", C.c_uint8), ("has_goal", C.c_uint8), ("ratio", C.c_uint8), ("shape", C.c_uint8),
                ("v0", C.c_float), ("goal", C.c_float)]


class OpDesc(C.Structure):
    """sauAmdOpDesc (include/saugns_amd.h): one operator of a voice bank for sauAmd_build_bank."""
    _fields_ = [("parent", C.c_uint32), ("use", C.c_uint32), ("type", C.c_uint32), ("mode", C.c_uint32),
                ("time_ms", C.c_uint32), ("start_ms", C.c_uint32), ("phase", C.c_uint32), ("seed", C.c_uint32),
                ("pan", LineDesc), ("amp", LineDesc), ("amp2", LineDesc), ("freq", LineDesc),
                ("freq2", LineDesc), ("pm_a", LineDesc)]


def _declare(L):
    """The C ABI of include/saugns_amd.h on a loaded library."""
    L.sau_create_Generator.restype = C.c_void_p
    L.sau_create_Generator.argtypes = [C.c_void_p, C.c_uint32]
    L.sau_destroy_Generator.argtypes = [C.c_void_p]
    L.sauGenerator_run.restype = C.c_bool
    L.sauGenerator_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_bool,
                                   C.POINTER(C.c_size_t)]
    L.sauAmd_create_Batch.restype = C.c_void_p
    L.sauAmd_create_Batch.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint32]
    if hasattr(L, "sauAmd_create_Batch_on"):  # (SAU_AMD_LIB may name an older build)
        L.sauAmd_create_Batch_on.restype = C.c_void_p
        L.sauAmd_create_Batch_on.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint32]
    L.sauAmd_destroy_Batch.argtypes = [C.c_void_p]
    L.sauAmd_render_file.restype = C.c_bool
    L.sauAmd_render_file.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64)]
    L.sauAmd_Batch_run.restype = C.c_bool
    L.sauAmd_Batch_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_bool,
                                   C.POINTER(C.c_bool), C.POINTER(C.c_size_t)]
    L.sauAmd_Batch_set_call_len.argtypes = [C.c_void_p, C.c_size_t]
    L.sauAmd_Batch_device_pcm.restype = C.c_void_p
    L.sauAmd_Batch_device_pcm.argtypes = [C.c_void_p, C.c_size_t]
    L.sauAmd_Batch_sync.restype = C.c_bool
    L.sauAmd_Batch_sync.argtypes = [C.c_void_p]
    L.sauAmd_Batch_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint64), C.c_int]
    L.sauAmd_Batch_timing_ex.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.sauAmd_Batch_set_timing.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "sauAmd_Batch_order_after"):  # (SAU_AMD_LIB may name an older build: A/B timing against earlier rounds)
        L.sauAmd_Batch_order_after.restype = C.c_bool
        L.sauAmd_Batch_order_after.argtypes = [C.c_void_p, C.c_void_p]
    L.sauAmd_Batch_stream.restype = C.c_void_p
    L.sauAmd_Batch_stream.argtypes = [C.c_void_p]
    L.sauAmd_set_piluts.argtypes = [C.c_void_p]
    L.sauAmd_get_piluts.restype = C.POINTER(C.c_float)
    L.sauAmd_last_error.restype = C.c_char_p
    L.sauAmd_device_count.restype = C.c_int
    L.sauAmd_device_pci_bus_id.restype = C.c_bool
    L.sauAmd_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.sauAmd_program_serialize.restype = C.c_size_t
    L.sauAmd_program_serialize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.sauAmd_program_load.restype = C.c_void_p
    L.sauAmd_program_load.argtypes = [C.c_void_p, C.c_size_t]
    L.sauAmd_program_free.argtypes = [C.c_void_p]
    L.sauAmd_build_bank.restype = C.c_void_p
    L.sauAmd_build_bank.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_uint32]
    L.sauAmd_free_bank.argtypes = [C.c_void_p]
    return L


def lib():
    """Load (building if stale) libsaugns_amd.so and declare its C ABI."""
    global _lib
    if _lib is not None:
        return _lib
    # SAU_AMD_LIB: load another build of the same library (A/B timing of kernel variants)
    path = os.environ.get("SAU_AMD_LIB") or _build.build()
    _lib = _declare(C.CDLL(path))
    if _tables is not None:
        _lib.sauAmd_set_piluts(_tables.ctypes.data)
    return _lib


_hooks = None
_tables = None


def use_hooks(path):
    """tests/ only: load the test-hook library (tests/hooks/libsaugns_amd_hooks.so: the product's object files + the entry
    points that run the host control plane over an injected backend, + the known-answer probes). The product library has
    none of those; objects made with ``backend=...`` live in, and are driven through, the hook library."""
    global _hooks
    if _hooks is None:
        L = _declare(C.CDLL(path))
        L.sauAmd_create_Generator_with_backend.restype = C.c_void_p
        L.sauAmd_create_Generator_with_backend.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.sauAmd_create_Batch_with_backend.restype = C.c_void_p
        L.sauAmd_create_Batch_with_backend.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint32, C.c_void_p]
        L.sauAmd_render_file_with_backend.restype = C.c_bool
        L.sauAmd_render_file_with_backend.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int,
                                                      C.c_void_p, C.POINTER(C.c_uint64)]
        L.sauAmd_Generator_rewinds.restype = C.c_uint
        L.sauAmd_Generator_rewinds.argtypes = [C.c_void_p]
        if _tables is not None:
            L.sauAmd_set_piluts(_tables.ctypes.data)
        _hooks = L
    return _hooks


def hooks():
    if _hooks is None:
        raise RuntimeError("the test-hook library is not loaded (tests/conftest.py: use_hooks)")
    return _hooks


_last_used = None  # the library the most recent Generator / Batch call went to (the hook library for backend=... objects)


def last_error(L=None):
    return (L or _last_used or lib()).sauAmd_last_error().decode()


def _used(L):
    global _last_used
    _last_used = L
    return L


def device_pci_bus_id(device=0):
    """PCI address of HIP device `device` ("0000:c1:00.0"), or None without one."""
    buf = C.create_string_buffer(64)
    return buf.value.decode() if lib().sauAmd_device_pci_bus_id(int(device), buf, 64) else None


def set_piluts(tables):
    global _tables
    t = np.ascontiguousarray(tables, dtype=np.float32).copy()
    assert t.shape == (12, 2048)
    _tables = t
    lib().sauAmd_set_piluts(t.ctypes.data)
    if _hooks is not None:
        _hooks.sauAmd_set_piluts(t.ctypes.data)


SNDFILE_RAW, SNDFILE_AU, SNDFILE_WAV = 0, 1, 2


def render_file(program, srate, path, fmt=SNDFILE_WAV, channels=1, backend=None):
    """sauAmd_render_file: render a whole program into a raw/AU/WAV file -> frames written.
    ``backend`` (tests): a sauengine::Backend* to run the same output stage without a GPU."""
    n = C.c_uint64()
    if backend is None:
        ok = lib().sauAmd_render_file(program.ptr, srate, os.fsencode(path), fmt, channels, C.byref(n))
    else:
        ok = hooks().sauAmd_render_file_with_backend(program.ptr, srate, os.fsencode(path), fmt,
                                                     channels, backend, C.byref(n))
    if not ok:
        raise RuntimeError("sauAmd_render_file failed: " + last_error(None if backend is None else hooks()))
    return n.value


def get_piluts():
    p = lib().sauAmd_get_piluts()
    return np.ctypeslib.as_array(p, shape=(12 * 2048,)).reshape(12, 2048).copy()


class Program:
    """A ``sauProgram`` in host memory: from an image, or borrowed from a parser."""

    def __init__(self, ptr, owner=None, free=None):
        self.ptr = ptr
        self._owner = owner
        self._free = free

    @classmethod
    def from_image(cls, blob):
        buf = bytes(blob)
        p = lib().sauAmd_program_load(buf, len(buf))
        if not p:
            raise ValueError("not a valid SAUPIMG1 program image")
        return cls(p, free=lib().sauAmd_program_free)

    @classmethod
    def borrow(cls, ptr, owner=None):
        return cls(ptr, owner=owner)

    def image(self):
        n = lib().sauAmd_program_serialize(self.ptr, None, 0)
        buf = C.create_string_buffer(n)
        lib().sauAmd_program_serialize(self.ptr, buf, n)
        return buf.raw

    @property
    def struct(self):
        return SauProgram.from_address(self.ptr)

    def __del__(self):
        if getattr(self, "_free", None) and self.ptr:
            self._free(self.ptr)
            self.ptr = None


class Generator:
    """sau_create_Generator / sauGenerator_run / sau_destroy_Generator."""

    def __init__(self, program, srate, backend=None):
        self._prg = program  # borrowed by the C side: keep alive
        self._L = _used(lib() if backend is None else hooks())
        if backend is None:
            self._g = self._L.sau_create_Generator(program.ptr, srate)
        else:  # tests: an injected backend (owned by the generator from here on), in the hook library
            self._g = self._L.sauAmd_create_Generator_with_backend(program.ptr, srate, backend)
        if not self._g:
            raise RuntimeError("sau_create_Generator returned NULL: " + last_error(self._L))

    def run(self, buf, buf_len, stereo=False):
        """-> (more, out_len); buf is an int16 numpy array of buf_len*(1|2)."""
        n = C.c_size_t()
        more = _used(self._L).sauGenerator_run(self._g, buf.ctypes.data, buf_len, stereo, C.byref(n))
        return bool(more), n.value

    def render(self, stereo=False, chunk=11289, max_frames=0):
        ch = 2 if stereo else 1
        buf = np.zeros(chunk * ch, np.int16)
        out, total = [], 0
        while True:
            more, n = self.run(buf, chunk, stereo)
            out.append(buf[: n * ch].copy())
            total += n
            if not more or (max_frames and total >= max_frames):
                break
        pcm = np.concatenate(out) if out else np.zeros(0, np.int16)
        return pcm[: max_frames * ch] if max_frames else pcm

    def rewinds(self):
        """tests: how often a call of another size / channel layout took the read-ahead back (hook library only)"""
        return int(self._L.sauAmd_Generator_rewinds(self._g))

    def close(self):
        if getattr(self, "_g", None):
            self._L.sau_destroy_Generator(self._g)
            self._g = None

    def __del__(self):
        self.close()


class Batch:
    """Many programs rendered in lock step (sauAmd_*Batch*)."""

    def __init__(self, programs, srate, backend=None, device=None):
        """device: the HIP device of this process to render on (sauAmd_create_Batch_on); None: SAU_AMD_DEVICE's, else device 0"""
        self._prgs = list(programs)
        self.n = len(self._prgs)
        arr = (C.c_void_p * self.n)(*[p.ptr for p in self._prgs])
        self._L = _used(lib() if backend is None else hooks())
        if backend is None and device is not None:
            self._b = self._L.sauAmd_create_Batch_on(int(device), arr, self.n, srate)
        elif backend is None:
            self._b = self._L.sauAmd_create_Batch(arr, self.n, srate)
        else:  # tests only: host control plane on an injected executor, in the hook library
            self._b = self._L.sauAmd_create_Batch_with_backend(arr, self.n, srate, backend)
        if not self._b:
            raise RuntimeError("sauAmd_create_Batch returned NULL: " + last_error(self._L))

    def run(self, buf_len, stereo=False, fetch=True):
        """-> (pcm [n, buf_len*ch] or None, more[n], out_len[n])"""
        ch = 2 if stereo else 1
        more = (C.c_bool * self.n)()
        lens = (C.c_size_t * self.n)()
        if fetch:
            pcm = np.zeros((self.n, buf_len * ch), np.int16)
            ptrs = (C.c_void_p * self.n)(*[pcm[i].ctypes.data for i in range(self.n)])
        else:
            pcm, ptrs = None, None
        ok = _used(self._L).sauAmd_Batch_run(self._b, ptrs, buf_len, stereo, more, lens)
        if not ok:
            raise RuntimeError("sauAmd_Batch_run failed: " + last_error(self._L))
        return pcm, [bool(m) for m in more], [int(x) for x in lens]

    def render(self, stereo=False, chunk=11289, max_frames=0):
        """Render every stream to its end -> list of int16 arrays."""
        ch = 2 if stereo else 1
        outs = [[] for _ in range(self.n)]
        alive = [True] * self.n
        total = 0
        while any(alive):
            pcm, more, lens = self.run(chunk, stereo)
            for i in range(self.n):
                if alive[i]:
                    outs[i].append(pcm[i, : lens[i] * ch].copy())
                    alive[i] = more[i]
            total += chunk
            if max_frames and total >= max_frames:
                break
        res = [np.concatenate(o) if o else np.zeros(0, np.int16) for o in outs]
        return [r[: max_frames * ch] for r in res] if max_frames else res

    def set_call_len(self, frames):
        """The sauGenerator_run call size whose block lattice the batch reproduces (0: every run
        is one call)."""
        self._L.sauAmd_Batch_set_call_len(self._b, frames)

    def order_after(self, before):
        """This batch's next run renders on the device only when everything issued for `before` has finished
        (sauAmd_Batch_order_after): scripts one after the other with two generators alive."""
        if not self._L.sauAmd_Batch_order_after(self._b, before._b):
            raise RuntimeError(last_error(self._L))

    def sync(self):
        if not self._L.sauAmd_Batch_sync(self._b):
            raise RuntimeError(last_error(self._L))

    def timing(self, reset=False):
        r, m, n = C.c_double(), C.c_double(), C.c_uint64()
        self._L.sauAmd_Batch_timing(self._b, C.byref(r), C.byref(m), C.byref(n), int(reset))
        return r.value, m.value, n.value

    def set_timing(self, level):
        self._L.sauAmd_Batch_set_timing(self._b, int(level))

    def timing_ex(self, reset=False):
        """-> dict of accumulated kernel times (ms) and the number of segments."""
        out = (C.c_double * 4)()
        n = C.c_uint64()
        self._L.sauAmd_Batch_timing_ex(self._b, out, C.byref(n), int(reset))
        return {"fast_ms": out[0], "block_ms": out[1], "mix_ms": out[2], "aux_ms": out[3],
                "segments": n.value}

    def device_pcm(self, stream):
        return self._L.sauAmd_Batch_device_pcm(self._b, stream)

    def close(self):
        if getattr(self, "_b", None):
            self._L.sauAmd_destroy_Batch(self._b)
            self._b = None

    def __del__(self):
        self.close()
