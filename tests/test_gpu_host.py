"""The reference's own command-line host, unchanged, on this library.

oracle/Makefile `hosts` compiles saugns.c and player/sndfile.c from where they lie under
/root/reference and links them twice: `saugns_cpu` with the reference's generator, `saugns_gpu` with
libsaugns_amd.so ahead of the reference archive (generator.o is then never
pulled in). System audio is tests/dropin/null_audiodev.c, a device that plays into a file. The
binaries travel to the GPU box prebuilt (oracle/_ref/ is git-ignored, not gpurun-ignored).
"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
CPU, GPU = os.path.join(REF, "saugns_cpu"), os.path.join(REF, "saugns_gpu")
have_hosts = os.path.exists(CPU) and os.path.exists(GPU)
from conftest import need_ref  # noqa: E402


@pytest.fixture(autouse=True)
def _hosts_are_here(request):
    """The two host binaries travel to the GPU box prebuilt; without them these tests FAIL (a missing checker must not look
    like a pass: VERDICT r04 item 6) -- except the one that needs the reference's script files, which exist in the build
    container only and which says so itself."""
    if not have_hosts and "crash_the_reference_parser" not in request.node.name:
        pytest.fail("oracle/_ref/saugns_cpu / saugns_gpu are not here: `make -C oracle hosts` where /root/reference exists")

# scripts of this repo's own (README.SAU syntax): PM stack, ramps + panning, FM + range AM, feedback with a
# frequency ramp, R and N operators, several timed steps with later events
SCRIPTS = [
    "Wsin",
    "Wsin f220 t0.5 p[Wsin r2 a0.8 p[Wtri r3 a0.3]]",
    "Wsaw f110 t0.4 a0.5[g0 lxpe] c-0.5[g0.5 llin]",
    "Wsin f330 t0.3 f[Wsin f5 a20] a.r0.2[Wsin f7]",
    "Wsin f200[g400 lexp] t0.5 p.a0.5",
    "Rcos f40 t0.5 a0.6\nNwh t0.25 a0.1",
    "Wsin f150 t0.3 a0.4; f300 t0.2; a0.1[g0.5] t0.3 | Wtri f90 t0.4 p[Wsin r3 a0.5]",
]


def run(exe, args, env=None, **kw):
    return subprocess.run([exe] + args, capture_output=True, timeout=300, env=dict(os.environ, **(env or {})), **kw)


def wav_parts(blob):
    assert blob[:4] == b"RIFF" and blob[8:16] == b"WAVEfmt "
    return blob[:44], np.frombuffer(blob[44:], "<i2").astype(np.int32)


def close(a, b, what):
    assert len(a) == len(b) and len(a) > 0, (what, len(a), len(b))
    # (the north star's tolerance is +-1 LSB int16; both hosts make the same calls, so with the reference build's loop
    #  tails reproduced -- the default -- the files are identical)
    assert int(np.abs(a - b).max()) == 0, what


def test_gpu_host_binds_the_generator_here(tmp_path):
    """No GPU needed: the unchanged host's generator calls and the reference parser's sauNoise_names
    resolve to libsaugns_amd.so, the binary holds none of the reference generator's code, and
    without a device it fails the way the reference host reports a failed generator (exit 1)."""
    syms = subprocess.run(["nm", GPU], capture_output=True, text=True).stdout
    assert " U sau_create_Generator" in syms and " U sauGenerator_run" in syms and " U sauNoise_names" in syms
    assert "run_block" not in syms and "sauWOsc" not in syms
    assert "run_block" in subprocess.run(["nm", CPU], capture_output=True, text=True).stdout
    out = run(GPU, ["-m", "-d", "-o", str(tmp_path / "x.wav"), "-e", "Wsin"],
              env={"LD_DEBUG": "bindings", "LD_DEBUG_OUTPUT": str(tmp_path / "ldd")})
    log = "".join(open(tmp_path / f, errors="replace").read() for f in os.listdir(tmp_path) if f.startswith("ldd."))
    bound = [ln.split(" to ")[-1] for ln in log.splitlines() if "`sau_create_Generator'" in ln and " to " in ln]
    assert bound and all("libsaugns_amd.so" in b for b in bound)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode == 1 and b"no CPU fallback" in out.stderr


def test_cpu_host_equals_the_compiled_reference(tmp_path, oracle, sa):
    """The reference host with the reference generator writes what libsau_ref.so renders (pins the
    host build itself: same sources, same flags)."""
    need_ref(oracle)
    for script in SCRIPTS[:4]:
        path = str(tmp_path / "c.wav")
        out = run(CPU, ["-m", "-d", "-r", "44100", "-o", path, "-e", script])
        assert out.returncode == 0, out.stderr
        _, pcm = wav_parts(open(path, "rb").read())
        prg = oracle.ref_build_program(script)
        want = oracle.ref_render(prg, 44100, True, chunk=11289).astype(np.int32)
        assert len(pcm) == len(want) and (pcm == want).all(), script


@pytest.mark.gpu
def test_unchanged_reference_host_on_the_gpu(tmp_path):
    """saugns -o x.wav / -o - (AU on stdout) / --stdout (raw) / --mono, every script through both
    binaries: headers byte-identical, samples within 1 LSB."""
    for k, script in enumerate(SCRIPTS):
        for mono in (False, True):
            flags = ["-m", "-d", "-r", "44100"] + (["--mono"] if mono else [])
            # WAV file
            files = {}
            for name, exe in (("cpu", CPU), ("gpu", GPU)):
                path = str(tmp_path / f"{name}.wav")
                out = run(exe, flags + ["-o", path, "-e", script])
                assert out.returncode == 0, (name, script, out.stderr[-1000:])
                files[name] = open(path, "rb").read()
            hc, pc = wav_parts(files["cpu"])
            hg, pg = wav_parts(files["gpu"])
            assert hc == hg, (script, mono)
            assert struct.unpack_from("<H", hg, 22)[0] == (1 if mono else 2)
            close(pg, pc, (script, mono, "wav"))
            if k % 2:
                continue
            # AU over stdout: big-endian samples, swapped in place by the host (player/sndfile.c:160-168)
            au = {name: run(exe, flags + ["-o", "-", "-e", script]).stdout for name, exe in (("cpu", CPU), ("gpu", GPU))}
            assert au["cpu"][:28] == au["gpu"][:28] and au["gpu"][:4] == b".snd"
            close(np.frombuffer(au["gpu"][28:], ">i2").astype(np.int32),
                  np.frombuffer(au["cpu"][28:], ">i2").astype(np.int32), (script, mono, "au"))
            # raw PCM on stdout
            raw = {name: run(exe, flags + ["--stdout", "-e", script]).stdout for name, exe in (("cpu", CPU), ("gpu", GPU))}
            close(np.frombuffer(raw["gpu"], "<i2").astype(np.int32),
                  np.frombuffer(raw["cpu"], "<i2").astype(np.int32), (script, mono, "raw"))
            assert (np.frombuffer(raw["gpu"], "<i2") == pg).all()


@pytest.mark.gpu
def test_two_generators_at_two_rates(tmp_path):
    """saugns.c:585 (split_gen): audio device on, a file asked for, and the device only supports another
    rate -- the host creates a second generator at the device rate and calls the two alternately, each
    with its own buffer length. File (44.1 kHz) and device stream (48 kHz) both within 1 LSB."""
    for script in (SCRIPTS[1], SCRIPTS[4], SCRIPTS[6]):
        got = {}
        for name, exe in (("cpu", CPU), ("gpu", GPU)):
            wav, dev = str(tmp_path / f"{name}.wav"), str(tmp_path / f"{name}.dev")
            out = run(exe, ["-a", "-d", "-r", "44100", "-o", wav, "-e", script],
                      env={"SAU_NULLDEV_SRATE": "48000", "SAU_NULLDEV_OUT": dev})
            assert out.returncode == 0, (name, out.stderr[-1000:])
            assert b"generating audio twice" in out.stderr
            got[name] = (wav_parts(open(wav, "rb").read()), np.fromfile(dev, "<i2").astype(np.int32))
        (hc, pc), dc = got["cpu"]
        (hg, pg), dg = got["gpu"]
        assert hc == hg
        close(pg, pc, (script, "file"))
        close(dg, dc, (script, "device"))
        assert len(dg) > len(pg)  # 48 kHz against 44.1 kHz


REF_SRC = "/root/reference"
PARSER_CRASHERS = ("devtests/crashes/testbindmultiple.sau", "devtests/crashes/testbindmultiple2.sau",
                   "devtests/crashes/testbindmultiple3.sau", "devtests/warning/label_without_operator.sau")


@pytest.mark.skipif(not have_hosts or not os.path.isdir(os.path.join(REF_SRC, "devtests")),
                    reason="needs the reference hosts and the reference's script files (build container only)")
def test_scripts_that_crash_the_reference_parser_end_both_hosts_alike(tmp_path):
    """Four scripts of the reference's corpus are outside every fixture because its parser crashes on them (devtests/crashes/
    is its collection of such cases; no program ever reaches a generator). The parser stays the reference's (SURVEY section 2,
    out of scope), so the host linked against this library must end exactly as the reference's own does: same signal, no
    output file. No GPU needed -- the crash comes before sau_create_Generator. (The fifth script that used to be skipped,
    alarm-25m, is an ordinary corpus fixture since round 4.)"""
    for rel in PARSER_CRASHERS:
        rc = {}
        for name, exe in (("cpu", CPU), ("gpu", GPU)):
            path = str(tmp_path / f"{name}.wav")
            out = run(exe, ["-m", "-d", "-r", "44100", "-o", path, os.path.join(REF_SRC, rel)])
            rc[name] = out.returncode
            assert not os.path.exists(path) or os.path.getsize(path) <= 44, (rel, name)
        assert rc["cpu"] < 0 and rc["gpu"] == rc["cpu"], (rel, rc)  # (negative: ended by a signal -- SIGSEGV here)
