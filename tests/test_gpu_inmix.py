"""The closed-form launch's task queues, one per XCD, and the mixer inside that launch (saugns_amd/csrc/k_fast_types.h, round 5):
a bank of many voices mixed into one stream can have its tiles mixed by the launch that renders the rows, chunk by chunk,
with mix_kernel taking what is left. The sums are the reference's ordered f32 sums either way (generator.c:749-825): every
PCM here is compared with the oracle's or with the reference's SHA-256, and SAU_AMD_INMIX_REPORT tells how much the launch
mixed itself -- a test that passed on the plain mixer alone would say nothing."""
import hashlib
import re

import numpy as np
import pytest

from conftest import ORACLE_FORMS

pytestmark = pytest.mark.gpu

LINE = re.compile(r"inmix: voices (\d+) frames (\d+) chunks (\d+) x (\d+) frames, tiles (\d+) of (\d+), chunks mixed whole (\d+), guard (\d+) (\d+)")


def _reports(capfd):
    return [tuple(int(x) for x in m.groups()) for m in LINE.finditer(capfd.readouterr().err)]


@pytest.fixture()
def report(monkeypatch):
    """The launch reports what it mixed itself."""
    monkeypatch.setenv("SAU_AMD_INMIX_REPORT", "1")


def test_config3_whole_render_in_one_run(sa, index, report, capfd):
    """BASELINE config 3 as bench.py renders it -- all 441000 frames in one engine run: SHA-256 of the PCM equals the
    reference's, and most of it was mixed inside the launch."""
    from saugns_amd import voicebank
    batch = sa.Batch([voicebank.config3()], 44100)
    pcm = batch.run(441000, stereo=False)[0][0]
    batch.close()
    assert hashlib.sha256(np.ascontiguousarray(pcm[:441000]).tobytes()).hexdigest() == index["configs"]["config3"]["sha256"]
    reps = _reports(capfd)
    assert reps, "the launch did not mix at all"
    voices, frames, nch, cf, tiles, of, whole, g0, g1 = reps[0]
    assert voices == 1024 and frames == 441000 and g0 == 0 and g1 == 0
    assert tiles * 2 > of, reps[0]


@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("grid", ["16", "24"])
def test_banks_against_the_oracle(sa, oracle, report, capfd, monkeypatch, stereo, grid):
    """Smaller banks, so that the oracle can render them: 96 voices x 4 s with pans of their own, a launch of few workgroups
    (SAU_AMD_FK_GRID: tasks enough per wave for the counter to deal them out) -- mono and stereo, frames not a multiple of
    anything."""
    from saugns_amd import voicebank as vb
    monkeypatch.setenv("SAU_AMD_FK_GRID", grid)
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = vb.config3_voices(96, 4)
    for i, v in enumerate(voices):
        v.pan = vb.Line(vb._num(".2f", ((i * 37) % 100) / 100.0))
    prg = vb.build_program(voices)
    want = oracle.oracle_render(prg.ptr, 44100, stereo, chunk=176400)
    batch = sa.Batch([prg], 44100)
    got = batch.run(176400, stereo=stereo)[0][0]
    batch.close()
    n = len(want)
    assert (np.asarray(got).reshape(-1)[:n] == want).all()
    reps = _reports(capfd)
    assert reps and reps[0][4] > 0, reps


def test_voices_of_different_depth_leave_it_to_the_mixer(sa, oracle, report, capfd, monkeypatch):
    """Voices whose lead-in differs (nesting depth 1 and 3) cut their row groups at different frames: premix_kernel says so
    and the launch mixes nothing; same PCM."""
    from saugns_amd import voicebank as vb
    monkeypatch.setenv("SAU_AMD_FK_GRID", "16")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = vb.config3_voices(64, 4) + [vb.Op("sin", freq=vb._num(".4f", 55.0 * (1 + i)), time_ms=4000) for i in range(32)]
    prg = vb.build_program(voices)
    want = oracle.oracle_render(prg.ptr, 44100, False, chunk=176400)
    batch = sa.Batch([prg], 44100)
    got = batch.run(176400, stereo=False)[0][0]
    batch.close()
    assert (np.asarray(got)[:len(want)] == want).all()
    reps = _reports(capfd)
    assert reps and reps[0][4] == 0 and reps[0][8] != 0, reps


def test_the_same_pcm_whoever_mixes_and_however_tasks_are_dealt(sa, oracle, capfd, monkeypatch):
    """Tasks from one queue per XCD, chunk-major, with the launch mixing (what ships) == the same with every frame left to
    mix_kernel (SAU_AMD_NO_INMIX) == the one counter in voice order (SAU_AMD_NO_XCD_QUEUES) == the oracle."""
    from saugns_amd import voicebank as vb
    monkeypatch.setenv("SAU_AMD_FK_GRID", "16")
    monkeypatch.setenv("SAU_AMD_INMIX_REPORT", "1")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.config3(96, 4)
    want = oracle.oracle_render(prg.ptr, 44100, False, chunk=176400)

    def render():
        batch = sa.Batch([prg], 44100)
        pcm = np.array(batch.run(176400, stereo=False)[0][0], copy=True)
        batch.close()
        return pcm[:len(want)]

    assert (render() == want).all()
    assert _reports(capfd)
    monkeypatch.setenv("SAU_AMD_NO_INMIX", "1")
    assert (render() == want).all()
    assert not _reports(capfd)
    monkeypatch.setenv("SAU_AMD_NO_XCD_QUEUES", "1")
    assert (render() == want).all()


@pytest.mark.parametrize("stereo", [False, True])
def test_edge_groups_and_inner_groups_in_launches_of_their_own(sa, oracle, report, capfd, monkeypatch, stereo):
    """The 12-row closed-form build renders a bank in two launches since round 6 -- every voice's first and last row group by the plain
    build, the groups between by the build that holds only the form without in-segment masks (k_fast_voice.h: INNER) -- or in one
    (SAU_AMD_NO_INNER): the same PCM, the oracle's, for voices that end at different frames (their last groups differ), in one run and
    in runs that cut the voices' groups elsewhere."""
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = vb.config3_voices(80, 5)
    for i, v in enumerate(voices):
        v.time_ms = 5000 - 37 * (i % 23)  # (different last groups)
        v.pan = vb.Line(vb._num(".2f", ((i * 29) % 100) / 100.0))
    prg = vb.build_program(voices)
    for chunk in (220500, 70001):
        want = oracle.oracle_render(prg.ptr, 44100, stereo, chunk=chunk)
        for no_inner in (False, True):
            if no_inner:
                monkeypatch.setenv("SAU_AMD_NO_INNER", "1")
            else:
                monkeypatch.delenv("SAU_AMD_NO_INNER", raising=False)
            batch = sa.Batch([prg], 44100)
            got = batch.render(stereo=stereo, chunk=chunk)[0]
            batch.close()
            assert len(got) == len(want) and (np.asarray(got) == want).all(), (chunk, no_inner)


def test_step_after_step_of_config3(sa, monkeypatch):
    """Six consecutive 441000-frame runs of a 60 s config-3 bank (the control words and the voice rows are reused from run to
    run): the launch that mixes and the mixer alone give the same PCM, run by run."""
    from saugns_amd import voicebank

    def runs(n):
        batch = sa.Batch([voicebank.config3(seconds=10 * n)], 44100)
        out = [hashlib.sha256(np.ascontiguousarray(batch.run(441000, stereo=False)[0][0]).tobytes()).hexdigest() for _ in range(n)]
        batch.close()
        return out

    a = runs(6)
    monkeypatch.setenv("SAU_AMD_NO_INMIX", "1")
    b = runs(6)
    assert a == b and len(set(a)) == 6


def _bank(vb, rng, n, seconds, depth):
    """n voices of one nesting depth (so that their row groups are cut at the same frames), frequencies, amplitudes and pans drawn"""
    voices = []
    for i in range(n):
        op = None
        for d in range(depth - 1):
            op = vb.Op("sin", freq=vb.Line(float(rng.integers(1, 6)), ratio=True), amp=vb._num(".2f", rng.uniform(0.2, 0.9)),
                       mods={vb.POP_PMOD: [op]} if op else None)
        voices.append(vb.Op("sin", freq=vb._num(".3f", rng.uniform(60.0, 900.0)), time_ms=seconds * 1000,
                            pan=vb.Line(vb._num(".2f", rng.uniform(0.0, 1.0))), mods={vb.POP_PMOD: [op]} if op else None))
    return vb.build_program(voices)


@pytest.mark.parametrize("seed", range(8))
def test_drawn_banks_against_the_oracle(sa, oracle, report, capfd, monkeypatch, seed):
    """Banks drawn at random -- 64 to 200 voices of depth 1 to 4, 3 to 5 s, mono or stereo, a run that ends short of the
    script or past it, launches of 8 to 32 workgroups -- identical to the oracle's PCM."""
    from saugns_amd import voicebank as vb
    rng = np.random.default_rng(9100 + seed)
    n, seconds, depth = int(rng.integers(64, 201)), int(rng.integers(3, 6)), int(rng.integers(1, 5))
    stereo, grid = bool(rng.integers(0, 2)), int(rng.integers(8, 33))
    run = int(seconds * 44100 + rng.integers(-5000, 3000))
    monkeypatch.setenv("SAU_AMD_FK_GRID", str(grid))
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = _bank(vb, rng, n, seconds, depth)
    want = oracle.oracle_render(prg.ptr, 44100, stereo, chunk=run)
    batch = sa.Batch([prg], 44100)
    pcm, more, lens = batch.run(run, stereo=stereo)
    got = [np.array(pcm[0][: lens[0] * (2 if stereo else 1)], copy=True)]
    while more[0]:
        pcm, more, lens = batch.run(run, stereo=stereo)
        got.append(np.array(pcm[0][: lens[0] * (2 if stereo else 1)], copy=True))
    batch.close()
    got = np.concatenate(got)
    assert len(got) == len(want) and (got == want).all(), (n, seconds, depth, stereo, grid, run)
    print("mixed inside the launch:", _reports(capfd))  # (short runs of shallow voices have too few tasks for the queues: pytest -s shows which)


def test_a_line_the_host_took_for_one_value_sends_the_voice_to_the_block_loop(sa, oracle, monkeypatch):
    """Round 5's lean buffer numbering (tests/test_plan_buffers.py): the host leaves out the buffers of frequency lines and range
    ends it knows to be one value. SAU_AMD_LEAN_IDS_LIE makes it forget the sweeps it has seen: decode_kernel then finds a
    line it must keep without a buffer, and the voice is rendered by the block loop -- same PCM as the oracle's, with the
    switch and without."""
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)

    def voice(i):
        m3 = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=vb._f32(0.4))
        m2 = vb.Op("sin", freq=vb.Line(2.0, goal=5.0, ratio=True), amp=vb._f32(0.7), mods={vb.POP_PMOD: [m3]})
        m1 = vb.Op("sin", freq=vb.Line(1.0, ratio=True), amp=20.0 + i, amp2=vb.Line(40.0), mods={vb.POP_PMOD: [m2]})
        return vb.Op("sin", freq=vb._num(".2f", 110.0 + 3.7 * i), freq2=vb.Line(vb._num(".2f", 220.0 + i), goal=330.0), time_ms=1500,
                     mods={vb.POP_FMOD: [m1]})

    prg = vb.build_program([voice(i) for i in range(24)])
    want = oracle.oracle_render(prg.ptr, 44100, False, chunk=11289)
    for lie in (False, True):
        if lie:
            monkeypatch.setenv("SAU_AMD_LEAN_IDS_LIE", "1")
        got = sa.Batch([prg], 44100).render(stereo=False, chunk=11289)[0]
        assert len(got) == len(want) and (got == want).all(), lie


@pytest.mark.parametrize("channels", [1, 2])
def test_files_written_with_the_launch_mixing(sa, oracle, report, capfd, monkeypatch, tmp_path, channels):
    """sauAmd_render_file over a bank the launch mixes itself: the AU file's big-endian PCM (the tiles swap bytes as mix_kernel
    does) and the WAV file's, byte for byte the restated writer's over the oracle's PCM."""
    from saugns_amd import voicebank as vb
    monkeypatch.setenv("SAU_AMD_FK_GRID", "16")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = vb.config3_voices(96, 8)
    for i, v in enumerate(voices):
        v.pan = vb.Line(vb._num(".2f", ((i * 29) % 100) / 100.0))
    prg = vb.build_program(voices)
    pcm = oracle.oracle_render(prg.ptr, 44100, channels == 2)
    for fmt, name in ((sa.api.SNDFILE_AU, "au"), (sa.api.SNDFILE_WAV, "wav")):
        path = str(tmp_path / f"bank.{name}")
        n = sa.render_file(prg, 44100, path, fmt, channels)
        assert n == len(pcm) // channels
        assert open(path, "rb").read() == oracle.oracle_sndfile_bytes(fmt, channels, 44100, pcm), name
    reps = _reports(capfd)
    assert reps and max(r[4] for r in reps) > 0, reps
