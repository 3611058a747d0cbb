"""GPU parity: HIP path (through the C ABI) against the oracle and the reference goldens."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_program, max_diff, ORACLE_FORMS, TAILS

pytestmark = pytest.mark.gpu

# scripts that exercise each part of the path; the full corpus runs in test_gpu_corpus
SMALL = ["devtests__voice-reuse", "examples__dull_seq-fm_pm", "examples__tests__panning",
         "examples__tests__wavetypes", "examples__rainy_thunder"]


def _corpus_keys(index):
    return sorted(index["corpus"].keys())


@pytest.fixture(scope="module")
def gpu(sa):
    if sa.lib().sauAmd_device_count() <= 0:
        pytest.fail("no HIP device: the GPU tests need the real hardware")
    return sa


def test_config1_plumbing(gpu, oracle, index, heads):
    """`-e "Wsin"`: 44100 frames, identical to the reference's own output."""
    from saugns_amd import voicebank
    pcm = gpu.Generator(voicebank.config1(), 44100).render()
    assert len(pcm) == 44100
    assert list(pcm[:8]) == [1001, 1538, 2557, 3565, 4560, 5537, 6492, 7421]
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == index["configs"]["config1"]["sha256"]


@pytest.mark.parametrize("key", SMALL)
def test_scripts_bit_exact_vs_oracle(gpu, oracle, index, key):
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = load_program(gpu, key)
    rate = index["corpus_rate"]
    want = oracle.oracle_render(prg.ptr, rate, True)
    got = gpu.Generator(prg, rate).render(stereo=True)
    assert max_diff(got, want) == 0


def test_gpu_corpus(gpu, oracle, index, heads):
    """Every corpus script: bit-exact vs the oracle and vs the heads of the compiled reference's renders (same call size, 11289)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    rate, head = index["corpus_rate"], index["head_frames"]
    worst_ref = 0
    bad = []
    for key in _corpus_keys(index):
        prg = load_program(gpu, key)
        want = oracle.oracle_render(prg.ptr, rate, True)
        got = gpu.Generator(prg, rate).render(stereo=True)
        if len(got) != len(want) or max_diff(got, want) != 0:
            bad.append(key)
            continue
        assert len(got) // 2 == index["corpus"][key]["frames"]
        g = heads[key]
        d = max_diff(got[: len(g)], g)
        worst_ref = max(worst_ref, d)
    assert not bad, f"GPU != oracle for {bad}"
    # (the north star allows +-1 LSB int16; with the reference build's loop tails reproduced nothing differs at all.
    #  With SAU_AMD_LOOP_TAILS=0 the one `cub` script, bg-drum-01, may differ by an LSB where a tail sample falls)
    assert worst_ref == (0 if TAILS else worst_ref) and worst_ref <= 1


@pytest.mark.parametrize("rate", [8000, 22050, 44100, 48000, 96000])
def test_gpu_corpus_other_sample_rates(gpu, oracle, index, rate):
    """Sample rate enters through the phase coefficients (wosc.h:57, rasg.h:126), every
    ms -> samples conversion with its carry (generator.c:148-160) and the ramp lengths: the
    whole corpus again at other rates, bit-exact vs the oracle (which equals the compiled
    reference bit for bit at these rates too: tests/test_oracle.py)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    bad = []
    for key in _corpus_keys(index):
        prg = load_program(gpu, key)
        stereo = rate in (44100, 96000)
        want = oracle.oracle_render(prg.ptr, rate, stereo)
        got = gpu.Generator(prg, rate).render(stereo=stereo, chunk=50000)
        if len(got) != len(want) or max_diff(got, want) != 0:
            bad.append(key)
    assert not bad, f"GPU != oracle at {rate} Hz for {bad}"


@pytest.mark.parametrize("name,n", [("config2", 11025), ("config3", 11025)])
def test_voicebank_heads(gpu, heads, index, name, n):
    """Configs 2/3 at full voice count, first 0.25 s, against the reference's PCM."""
    from saugns_amd import voicebank
    prg = getattr(voicebank, name)()
    pcm = gpu.Generator(prg, 44100).render(max_frames=n, chunk=n)
    assert hashlib.sha256(pcm[:n].tobytes()).hexdigest() == index["configs"][name]["head_sha256"]


def test_config3_full_checksum(gpu, index):
    """1024 voices x depth-3 PM, full 10 s: SHA-256 of the PCM equals the reference's."""
    from saugns_amd import voicebank
    pcm = gpu.Generator(voicebank.config3(), 44100).render(chunk=44100)
    assert len(pcm) == index["configs"]["config3"]["frames"]
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == index["configs"]["config3"]["sha256"]


def test_config5_head(gpu, heads):
    """4096 voices with feedback FM + range AM + ramps: first 0.25 s, identical to the reference's."""
    from saugns_amd import voicebank
    pcm = gpu.Generator(voicebank.config5(), 44100).render(max_frames=11025, chunk=11025)
    assert max_diff(pcm[:11025], heads["config5"][:11025]) == 0


def test_config4_batch(gpu, heads, index):
    """rainy_thunder x 4 seeds rendered as one batch == the reference, stream by stream."""
    prgs = [load_program(gpu, f"config4_seed{k}") for k in range(4)]
    outs = gpu.Batch(prgs, 44100).render(chunk=44100, max_frames=88200)
    for k, pcm in enumerate(outs):
        assert max_diff(pcm[:88200], heads[f"config4_seed{k}"]) == 0


def test_chunk_size_invariance(gpu, oracle):
    """Caller buffer length must not change the result (SURVEY D-2)."""
    prg = load_program(gpu, "examples__dull_seq-fm_pm")
    a = gpu.Generator(prg, 12000).render(stereo=True, chunk=3072)
    b = gpu.Generator(prg, 12000).render(stereo=True, chunk=997)
    assert max_diff(a, b) == 0


@pytest.mark.parametrize("lds_limit", [None, "98304"])
def test_corpus_single_wave_teams(gpu, lds_limit):
    """The many-voices block-loop geometry (sixteen one-wave teams per workgroup, two frames
    per lane, or one when LDS is short) forced onto every corpus script: bit-exact vs the oracle."""
    import subprocess
    import sys
    env = dict(os.environ, SAU_AMD_MULTI_MIN="1")
    if lds_limit:
        env["SAU_AMD_LDS_LIMIT"] = lds_limit
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_corpus_check.py")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("0 bad of"), out.stdout[-2000:]


def test_config2_full_checksum(gpu, index):
    """256 flat voices, full 10 s: SHA-256 of the PCM equals the reference's."""
    from saugns_amd import voicebank
    pcm = gpu.Generator(voicebank.config2(), 44100).render(chunk=44100)
    assert len(pcm) == index["configs"]["config2"]["frames"]
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == index["configs"]["config2"]["sha256"]


def test_config5_full_checksum_and_call_size_invariance(gpu, index):
    """4096 feedback voices with ramps, full 10 s at full size: one 441000-frame call and
    11289-frame calls give the same PCM, whose SHA-256 is the reference's."""
    from saugns_amd import voicebank
    a = gpu.Generator(voicebank.config5(), 44100).render(chunk=441000)
    b = gpu.Generator(voicebank.config5(), 44100).render(chunk=11289)
    assert len(a) == len(b) == 441000
    sha = hashlib.sha256(a.tobytes()).hexdigest()
    assert sha == hashlib.sha256(b.tobytes()).hexdigest()
    assert sha == index["configs"]["config5"]["sha256"]  # pinned from the compiled reference


def test_config4_full_checksums(gpu, index):
    """rainy_thunder, four seeds, the full 60 s each, rendered as one batch: every stream's
    SHA-256 equals the reference's."""
    prgs = [load_program(gpu, f"config4_seed{k}") for k in range(4)]
    outs = gpu.Batch(prgs, 44100).render(chunk=441000)
    for k, pcm in enumerate(outs):
        ref = index["configs"][f"config4_seed{k}"]
        assert len(pcm) == ref["frames"]
        assert hashlib.sha256(pcm.tobytes()).hexdigest() == ref["sha256"]


@pytest.mark.parametrize("env", [{"SAU_AMD_LDS_LIMIT": "40000"},
                                 {"SAU_AMD_LDS_LIMIT": "30000", "SAU_AMD_NO_FAST": "1"},
                                 {"SAU_AMD_FAST_ROWS": "2", "SAU_AMD_NO_TWO_PASS": "1"}])
def test_corpus_under_tight_lds_and_alternative_builds(gpu, env):
    """Every corpus script with little LDS (wave tables read from HBM, the one-wave one-frame-per-lane
    block-loop geometry for voices with many block buffers) and with the smaller builds of the
    time-parallel kernel: bit-exact vs the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_corpus_check.py")],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("0 bad of"), out.stdout[-2000:]


def test_config4_at_its_stated_size(gpu, index):
    """BASELINE config 4 as one GPU sees it: 64 distinct renders of rainy_thunder.sau (seed = k, here
    the block a rank of an 8-GPU job gets: shard_range(512, 3, 8)) in one batch, the full 60 s each,
    behind the reference host's call size: the SHA-256 of every render equals the compiled
    reference's (tests/golden/config4_seeds.npz)."""
    from saugns_amd.shard import shard_range
    fx = np.load(os.path.join(GOLDEN, "config4_seeds.npz"))
    a, b = shard_range(512, 3, 8)
    assert b - a == 64
    prgs = [gpu.Program.from_image(fx["images"][k].tobytes()) for k in range(a, b)]
    batch = gpu.Batch(prgs, 44100)
    batch.set_call_len(11289)
    outs = batch.render(chunk=441000)
    bad = [a + i for i, pcm in enumerate(outs)
           if len(pcm) != int(fx["frames"][a + i]) or hashlib.sha256(pcm.tobytes()).hexdigest() != str(fx["sha256"][a + i])]
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["config3", "config5"])
def test_banks_built_by_the_c_abi_render_like_the_parsers(sa, oracle, shape):
    """a bank handed over as flat operator descriptions (sauAmd_build_bank, no parser) renders
    on the device exactly as the oracle renders it, and exactly as the bank the Python builder makes -- which is
    pinned on parser-made images -- for the shapes of BASELINE configs 3 and 5 (128 voices, 1 s)."""
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = vb.config3_voices(128, 1) if shape == "config3" else vb.config5_voices(128, 1)
    bank = vb.build_bank_c(voices)
    ref_bank = vb.build_program(voices)
    want = oracle.oracle_render(bank.ptr, 44100, False)
    assert (want == oracle.oracle_render(ref_bank.ptr, 44100, False)).all()
    got = sa.Batch([bank], 44100).render(stereo=False, chunk=44100)[0]
    assert len(got) == len(want) == 44100 and (got == want).all()
    # ... and through the drop-in generator with the reference host's call size
    got2 = sa.Generator(bank, 44100).render(stereo=False, chunk=11289)
    assert len(got2) == len(want) and (got2 == want).all()


@pytest.mark.gpu
def test_generators_in_a_row_give_the_same_pcm(sa, oracle):
    """A process that renders one script after another gets pooled buffers back at once. Found in round 3: the PCM
    block was cleared with a memset on the null stream, which is not ordered with the generator's own (non-blocking)
    stream and may still be at work when the first mixer writes -- zeros from some page on, in the second and later
    generators of a process, about nine renders in ten for the scripts below (many short segments: the first mixer
    comes early). Twelve renders in a row of each, through both APIs, every one identical to the oracle's."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for key, rate, stereo in (("examples__tests__tone_seq-v1", 44100, True), ("examples__tests__tone_seq-v2_label", 48000, False)):
        prg = load_program(sa, key)
        want = oracle.oracle_render(prg.ptr, rate, stereo, chunk=11289)
        for rep in range(12):
            got = sa.Generator(prg, rate).render(stereo=stereo, chunk=11289)
            assert len(got) == len(want) and (got == want).all(), (key, rep, "generator")
            got = sa.Batch([prg], rate).render(stereo=stereo, chunk=11289)[0]
            assert len(got) == len(want) and (got == want).all(), (key, rep, "batch")
