import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# every plan the engine takes from its cache of graph shapes is compared with a fresh compile (engine.cpp)
os.environ.setdefault("SAU_AMD_PLAN_CHECK", "1")
# tuning and test switches (SAU_AMD_NO_FAST, SAU_AMD_LDS_LIMIT, ... -- everything that is not a product setting) are only
# looked at when SAU_AMD_TUNE is set (engine.h: tune_env); the suite uses many of them to force every kernel build
os.environ.setdefault("SAU_AMD_TUNE", "1")
# The suite runs what ships: the product's default reproduces the compiled reference's loop tails of `cub` (oracle mode 2,
# which depends on the host's call size -- device and oracle are given the same one); `SAU_AMD_LOOP_TAILS=0 pytest ...` runs
# the same suite with the loop bodies' forms everywhere (oracle mode 1), and tests/test_gpu_vs_ref.py and
# tests/test_loop_tails.py keep legs of their own in that setting.
TAILS = os.environ.get("SAU_AMD_LOOP_TAILS", "1") != "0"
ORACLE_FORMS = 2 if TAILS else 1  # ora_set_fastmath_forms(): what the device is compared with


import contextlib


@contextlib.contextmanager
def loop_tails(on):
    """Engines created inside render with the reference build's loop tails of `cub` on or off (SAU_AMD_LOOP_TAILS is read
    when an engine is created)."""
    old = os.environ.get("SAU_AMD_LOOP_TAILS")
    os.environ["SAU_AMD_LOOP_TAILS"] = "1" if on else "0"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("SAU_AMD_LOOP_TAILS", None)
        else:
            os.environ["SAU_AMD_LOOP_TAILS"] = old


@pytest.fixture()
def tails_on():
    with loop_tails(True):
        yield


@pytest.fixture()
def tails_off():
    with loop_tails(False):
        yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests; asking for them explicitly
    (`-m gpu`, as the round-end run does) keeps them, and they then fail loudly without a device --
    the product has no CPU fallback to pass on."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    try:
        import torch
        if torch.cuda.is_available():
            return
    except Exception:
        pass
    skip = pytest.mark.skip(reason="no GPU here (select with -m gpu to insist)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def tables():
    return np.fromfile(os.path.join(GOLDEN, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)


@pytest.fixture(scope="session")
def index():
    return json.load(open(os.path.join(GOLDEN, "index.json")))


@pytest.fixture(scope="session")
def heads():
    return np.load(os.path.join(GOLDEN, "pcm_heads.npz"))


@pytest.fixture(scope="session")
def oracle(tables):
    """The CPU restatement, using the reference's wave tables."""
    from oracle import pyoracle as po
    po.build(ref=False)
    po.oracle_use_tables(tables)
    return po


@pytest.fixture(scope="session")
def sa(tables):
    import saugns_amd
    saugns_amd.lib()
    saugns_amd.set_piluts(tables)
    return saugns_amd


def hooks_path():
    """tests/hooks/libsaugns_amd_hooks.so: the product's own object files + the test entry points (injected backends, the
    read-ahead's rewind count, known-answer probes). SAU_HOOKS_LIB: another build that holds them (tests/test_sanitizers.py)."""
    import subprocess
    path = os.environ.get("SAU_HOOKS_LIB")
    if not path:
        import saugns_amd
        saugns_amd.build()  # (the hook library links saugns_amd/csrc/*.o)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hooks")])
        path = os.path.join(ROOT, "tests", "hooks", "libsaugns_amd_hooks.so")
    return path


@pytest.fixture(scope="session")
def hooks(sa):
    """The hook library, registered with saugns_amd.api (objects made with backend=... live in it)."""
    L = sa.api.use_hooks(hooks_path())
    L.sauAmd_kat_line_host.restype = C.c_int
    L.sauAmd_kat_line_host.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    if hasattr(L, "sauAmd_kat_line_device"):  # (not in the sanitizer build of the host side)
        L.sauAmd_kat_line_device.restype = C.c_int
        L.sauAmd_kat_line_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.sauAmd_kat_div_device.restype = C.c_longlong
        L.sauAmd_kat_div_device.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.sauAmd_kat_rint64_device.restype = C.c_longlong
        L.sauAmd_kat_rint64_device.argtypes = [C.c_int, C.c_void_p]
        L.sauAmd_kat_scan64_device.restype = C.c_int
        L.sauAmd_kat_scan64_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    return L


@pytest.fixture(scope="session")
def seqexec(hooks):
    import subprocess
    # SAU_SEQEXEC_LIB: another build that holds the executor (tests/test_sanitizers.py: the ASan + UBSan library, which is
    # also what SAU_AMD_LIB then points the product's loader at)
    path = os.environ.get("SAU_SEQEXEC_LIB")
    if not path:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "seqexec")])
        path = os.path.join(ROOT, "tests", "seqexec", "libseqexec.so")
    lib = C.CDLL(path)
    lib.seq_backend_create.restype = C.c_void_p
    lib.seq_backend_create.argtypes = [C.c_uint32]
    lib.seq_backend_set_memory.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint]
    lib.seq_backend_segments.argtypes = [C.c_void_p, C.c_void_p]
    return lib


def need_ref(oracle):
    """GPU tests that compare with the compiled reference (oracle/_ref/libsau_ref.so, built here by oracle/Makefile from
    /root/reference and carried to the GPU box with the tree): a missing checker is a FAILURE there, not a skip -- nine of
    the strongest tests would otherwise vanish silently."""
    if not oracle.have_ref():
        pytest.fail("oracle/_ref/libsau_ref.so is not here: build it with `make -C oracle ref` where /root/reference exists; "
                    "it travels to the GPU box with the tree")


@contextlib.contextmanager
def ref_tables(sa, oracle, tables):
    """Device, oracle and compiled reference on the wave tables that very reference library has built: it makes them with
    libm's sin() when it starts (sau/wave.c:105-221) and glibc picks its sin() by CPU -- on the GPU box's host four tables
    differ from the fixture by an ulp here and there. (What a host linked against this backend gets too: the shim adopts the
    host binary's sauWave_piluts, INTEGRATION.md.)"""
    oracle.ref()
    t = oracle.ref_piluts()
    sa.set_piluts(t)
    oracle.oracle_use_tables(t)
    try:
        yield t
    finally:
        sa.set_piluts(tables)
        oracle.oracle_use_tables(tables)


def load_program(sa, key):
    blob = open(os.path.join(GOLDEN, "programs", key + ".saup"), "rb").read()
    return sa.Program.from_image(blob)


def max_diff(a, b):
    assert len(a) == len(b), (len(a), len(b))
    if len(a) == 0:
        return 0
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max())
