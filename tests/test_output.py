"""Output stage: raw / AU / WAV files written by
sauAmd_render_file against the reference's player/sndfile.c."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_program

FORMATS = [(0, "raw"), (1, "au"), (2, "wav")]


def test_writer_restatement_vs_reference_files(oracle):
    """oracle_sndfile_bytes == the files the reference's own writer produced (fixture made by
    tests/golden/make_golden.py through oracle/_ref), and == that writer run live when present."""
    kat = np.load(os.path.join(GOLDEN, "sndfile_kat.npz"))
    pcm = kat["pcm"]
    for fmt, name in FORMATS:
        for ch in (1, 2):
            want = kat[f"{name}_{ch}"].tobytes()
            assert oracle.oracle_sndfile_bytes(fmt, ch, 44100, pcm) == want, (name, ch)
    if oracle.have_ref():
        import tempfile
        rng = np.random.default_rng(7)
        pcm = rng.integers(-32768, 32767, 3 * 11289 + 5, dtype=np.int16)  # several writer chunks
        for fmt, name in FORMATS:
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "x")
                oracle.ref_write_sndfile(path, fmt, 1, 48000, pcm)
                assert open(path, "rb").read() == oracle.oracle_sndfile_bytes(fmt, 1, 48000, pcm)


@pytest.mark.parametrize("key", ["config1", "examples__tests__panning"])
def test_render_file_on_cpu_backend(sa, oracle, seqexec, index, tmp_path, key):
    """The output stage over the sequential test backend: headers, size fields, byte order,
    chunking and end of signal -- byte-identical to the restated reference writer fed with
    the oracle's PCM."""
    if key != "config1" and key not in index["corpus"]:
        pytest.skip("fixture program not present")
    oracle.oracle().ora_set_fastmath_forms(1)
    prg = load_program(sa, key)
    rate = 44100 if key == "config1" else index["corpus_rate"]
    for fmt, name in FORMATS:
        for ch in (1, 2):
            path = str(tmp_path / f"{key}_{name}_{ch}")
            n = sa.render_file(prg, rate, path, fmt, ch, backend=seqexec.seq_backend_create(1016))
            pcm = oracle.oracle_render(prg.ptr, rate, ch == 2)
            assert n == len(pcm) // ch
            assert open(path, "rb").read() == oracle.oracle_sndfile_bytes(fmt, ch, rate, pcm), (name, ch)


def test_render_file_bad_arguments(sa, seqexec, tmp_path):
    prg = load_program(sa, "config1")
    with pytest.raises(RuntimeError):
        sa.render_file(prg, 44100, str(tmp_path / "no" / "such" / "dir" / "x.wav"), 2, 1,
                       backend=seqexec.seq_backend_create(256))
    with pytest.raises(RuntimeError):
        sa.render_file(prg, 44100, str(tmp_path / "x.wav"), 2, 3, backend=seqexec.seq_backend_create(256))


@pytest.mark.gpu
def test_render_file_gpu(sa, oracle, index, tmp_path):
    """The product path: device-side byte order, page-locked double buffering, several device
    runs per file (10 s at 44.1 kHz = three 176400-frame runs)."""
    from saugns_amd import voicebank
    if sa.lib().sauAmd_device_count() <= 0:
        pytest.fail("no HIP device: the GPU tests need the real hardware")
    oracle.oracle().ora_set_fastmath_forms(1)
    cases = [(voicebank.config3(n=16, seconds=10), 44100), (load_program(sa, "config1"), 44100)]
    if "examples__tests__panning" in index["corpus"]:
        cases.append((load_program(sa, "examples__tests__panning"), index["corpus_rate"]))
    for prg, rate in cases:
        pcm = {ch: oracle.oracle_render(prg.ptr, rate, ch == 2) for ch in (1, 2)}
        for fmt, name in FORMATS:
            for ch in (1, 2):
                path = str(tmp_path / f"f_{name}_{ch}")
                n = sa.render_file(prg, rate, path, fmt, ch)
                assert n == len(pcm[ch]) // ch
                assert open(path, "rb").read() == oracle.oracle_sndfile_bytes(fmt, ch, rate, pcm[ch]), (name, ch)


@pytest.mark.gpu
def test_render_file_random_programs_gpu(sa, oracle, tmp_path):
    """Randomized programs with events (tests/test_gpu_units.py) through the output stage: the WAV
    and AU files equal the restated writer fed with the oracle's PCM for the stage's own run size."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_units as tu
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(1)
    for seed in range(200, 212):
        rng = np.random.default_rng(5000 + seed)
        voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        tu._random_starts(rng, voices)
        prg = vb.build_program(voices, updates=tu._random_updates(rng, voices))
        ch = 1 + (seed & 1)
        fmt = 1 + (seed >> 1 & 1)
        pcm = oracle.oracle_render(prg.ptr, 48000, ch == 2, chunk=176400)
        path = str(tmp_path / f"r{seed}")
        n = sa.render_file(prg, 48000, path, fmt, ch)
        assert n == len(pcm) // ch
        assert open(path, "rb").read() == oracle.oracle_sndfile_bytes(fmt, ch, 48000, pcm), seed
