"""Build libsaugns_amd.so (host control plane + gfx950 kernels) in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsaugns_amd.so")
SOURCES = ["capi.cpp", "engine.cpp", "plan.cpp", "tables.cpp", "program_io.cpp", "sndout.cpp", "bank_builder.cpp",
           "hip_backend.hip"]
HEADERS = ["engine.h", "hip_backend.h", "capi_internal.h", "sau_dev_math.h", "sau_dev_ops.h",
           "k_common.h", "k_wave_scan.h", "k_block_loop.h", "k_fast_types.h", "k_analyze.h", "k_decode.h", "k_fast_voice.h", "k_fast_group.h",
           "k_chain.h", "k_finish.h",  # parts of hip_backend.hip
           "sau_dev_types.h", "../../include/sau_abi.h", "../../include/saugns_amd.h"]
# -ffp-contract=off: the arithmetic contract forbids FMA contraction
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [__file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Kernel-tuning aid: build saugns_amd/variants/lib_<name>.so with extra -D flags
    (only hip_backend.hip is recompiled; the other objects come from the main build)."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, "hip_backend_%s.o" % name)
    subprocess.check_call([hipcc] + FLAGS + ["-D" + d for d in defines] +
                          ["-c", os.path.join(CSRC, "hip_backend.hip"), "-o", obj])
    objs = [os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o") for s in SOURCES if s != "hip_backend.hip"]
    out = os.path.join(vdir, "lib_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj, "-ldl"])
    return out


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    # several ranks of one node may get here at once: one builds, the others wait and re-check
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(verbose, force=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        # the kernel parts (k_*.h) are included by hip_backend.hip only
        deps = [src] + [h for h in HEADERS if src.endswith(".hip") or not h.startswith("k_")]
        if not force and os.path.exists(obj) and all(
                os.path.getmtime(os.path.join(CSRC, d)) <= os.path.getmtime(obj) for d in deps) and \
                os.path.getmtime(__file__) <= os.path.getmtime(obj):
            continue
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # --variant NAME DEF1 DEF2 ...
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
