"""The host control plane under AddressSanitizer + UndefinedBehaviorSanitizer.

tests/seqexec/Makefile `asan` compiles the seven host sources of libsaugns_amd.so exactly as they ship -- engine.cpp, plan.cpp,
capi.cpp, program_io.cpp, bank_builder.cpp, sndout.cpp, tables.cpp; none of them touches HIP -- together with the sequential
executor into one g++ library with -fsanitize=address,undefined, every finding fatal. A child process with the ASan runtime
preloaded then (i) runs the host tests (tests/test_host.py, tests/test_output.py) on that library and (ii) feeds 10000 mutated
program images to sauAmd_program_load and 2500 mutated operator descriptions to sauAmd_build_bank, rendering what is accepted
(tests/tools/fuzz_host_inputs.py). GPU sanitizers are not available on this pool; the device code is out of this test's reach.
Round 4's first runs found: a shift by an R level beyond 31 from a mutated image (sar32), and image counts of 2^31 operators
that made the engine allocate until std::bad_alloc left through the C ABI (counts are now bounded by what the image holds, and
no C++ exception crosses the ABI)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SEQ = os.path.join(ROOT, "tests", "seqexec")
LIB = os.path.join(SEQ, "libsaugns_host_asan.so")


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def san_env():
    rt = _asan_runtime()
    if not rt:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-s", "-C", SEQ, "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", SAU_AMD_LIB=LIB, SAU_SEQEXEC_LIB=LIB, SAU_HOOKS_LIB=LIB)
    return env


def test_host_tests_under_asan_ubsan(san_env):
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host.py"),
                          os.path.join(ROOT, "tests", "test_output.py"), "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, env=san_env, cwd=ROOT, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "failed" not in out.stdout, tail
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, tail


def test_mutated_inputs_under_asan_ubsan(san_env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_host_inputs.py"), "10000", "4"],
                         capture_output=True, text=True, env=san_env, cwd=ROOT, timeout=900)
    tail = (out.stdout[-500:] + out.stderr[-3000:])
    assert out.returncode == 0, tail
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, tail
    s = json.loads(out.stdout.strip().splitlines()[-1])
    # the fuzz means something only if a fair share of the mutants get past the loaders and into the engine
    assert s["image_mutations"] == 10000 and s["images_accepted"] > 1000 and s["banks_accepted"] > 500 and s["rendered"] > 500, s
