#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (tests/test_dist.py): bench.py's N-rank logic -- launcher, sharding, barriers, reductions, the shape of
the result line -- on a box without GPUs: the same script with its batches made over the sequential plan executor
(tests/seqexec) through the test-hook library (tests/hooks). Same arguments as bench.py; `bench.launch_ranks` starts the
ranks as children of THIS script (it re-launches sys.argv[0]). A line made this way says "TEST BACKEND" in `data` and
measures nothing. SAU_SEQEXEC_LIB / SAU_HOOKS_LIB: other builds of the two libraries (a path that does not exist makes the
rank fail, which one test wants)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

_seq = None


def new_batch(sa, prgs):
    global _seq
    if _seq is None:
        sa.api.use_hooks(os.environ.get("SAU_HOOKS_LIB") or os.path.join(ROOT, "tests", "hooks", "libsaugns_amd_hooks.so"))
        _seq = C.CDLL(os.environ.get("SAU_SEQEXEC_LIB") or os.path.join(ROOT, "tests", "seqexec", "libseqexec.so"))
        _seq.seq_backend_create.restype = C.c_void_p
        _seq.seq_backend_create.argtypes = [C.c_uint32]
    return sa.Batch(prgs, 44100, backend=_seq.seq_backend_create(1016))


if __name__ == "__main__":
    bench.HARNESS = {"new_batch": new_batch}
    bench.main()
