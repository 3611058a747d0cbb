"""Closed-form voices and running-sum (look-back) voices of a segment in ONE launch (duo_kernel, saugns_amd/csrc/k_fast_voice.h,
round 6): a workgroup's waves split into the two roles by the lengths of analyze_kernel's lists. Whichever launches render the
rows, the PCM is the reference's (generator.c:505-729, wosc.h:135-169, rasg.h:299-671): every render here is compared with the
oracle's, in the setting that ships, with the two launches apart (SAU_AMD_NO_DUO) and with the split forced to its extremes
(SAU_AMD_DUO_LW) -- and SAU_AMD_DEBUG_DUO says whether the joint launch ran, so that a test passing without it says so."""
import re

import numpy as np
import pytest

from conftest import ORACLE_FORMS
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_AMOD, POP_FMOD, POP_PMOD, POPT_RASEG

pytestmark = pytest.mark.gpu

LINE = re.compile(r"\[sau-amd\] duo (\d):")


class _Duos(list):
    text = ""


def _duos(capfd):
    err = capfd.readouterr().err
    d = _Duos(int(m.group(1)) for m in LINE.finditer(err))
    d.text = "\n".join(l for l in err.splitlines() if "duo" in l)
    return d


def _pm_voice(i, ms):
    m2 = vb.Op("sin", freq=vb.Line(float(2 + i % 3), ratio=True), amp=vb._f32(0.7))
    m1 = vb.Op(("sin", "tri", "sqr")[i % 3], freq=vb.Line(float(1 + i % 5), ratio=True), amp=vb._num(".2f", 0.5 + (i % 7) * 0.1),
               mods={POP_PMOD: [m2]})
    return vb.Op("sin", freq=vb._num(".4f", 110.0 + i * 0.731), time_ms=ms, pan=vb.Line(vb._num(".2f", ((i * 37) % 100) / 100.0)),
                 mods={POP_PMOD: [m1]})


def _fm_voice(i, ms):
    m2 = vb.Op("sin", freq=vb.Line(float(2 + i % 3), ratio=True), amp=vb._f32(0.7))
    m1 = vb.Op("sin", freq=vb.Line(float(1 + i % 5), ratio=True), amp=vb._num(".1f", 20.0 + (i % 7) * 5.0), mods={POP_PMOD: [m2]})
    return vb.Op(("sin", "saw")[i % 2], freq=vb._num(".4f", 140.0 + i * 1.377), time_ms=ms, mods={POP_FMOD: [m1]})


def _r_voice(i, ms):
    """a carrier whose frequency an R oscillator with a modulated rate moves (64-bit cycle counters through the look-back), AM beside"""
    rate = vb.Op("sin", freq=0.7 + 0.1 * (i % 5), amp=3.0)
    r = vb.Op(op_type=POPT_RASEG, ras=(("lin", "cos", "sqe")[i % 3], i % 4, 0), seed=77 + 5 * i, freq=9.0 + i % 6, amp=25.0,
              mods={POP_FMOD: [rate]})
    am = vb.Op("sin", freq=2.0 + i % 3, amp=0.3)
    return vb.Op("sin", freq=200.0 + 3.1 * i, time_ms=ms, mods={POP_FMOD: [r], POP_AMOD: [am]})


def _render(sa, prgs, frames, stereo):
    batch = sa.Batch(prgs, 44100)
    pcm = [np.array(p, copy=True) for p in batch.run(frames, stereo=stereo)[0]]
    batch.close()
    return pcm


@pytest.mark.parametrize("stereo", [False, True])
def test_a_bank_of_both_kinds(sa, oracle, capfd, monkeypatch, stereo):
    """96 PM voices and 48 FM / R voices, interleaved, 3 s in one engine run: the joint launch, the two launches apart and the
    split at 2 and at 14 look-back waves of a workgroup's 16 all give the oracle's PCM."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    monkeypatch.setenv("SAU_AMD_DEBUG_DUO", "1")
    voices = []
    for i in range(48):
        voices += [_pm_voice(2 * i, 3000), _fm_voice(i, 3000) if i % 3 else _r_voice(i, 3000), _pm_voice(2 * i + 1, 3000)]
    prg = vb.build_program(voices)
    want = oracle.oracle_render(prg.ptr, 44100, stereo, chunk=132300)
    got = _render(sa, [prg], 132300, stereo)[0]
    assert (np.asarray(got).reshape(-1)[:len(want)] == want).all()
    assert 1 in _duos(capfd), "the joint launch did not run"
    monkeypatch.setenv("SAU_AMD_NO_DUO", "1")
    got = _render(sa, [prg], 132300, stereo)[0]
    assert (np.asarray(got).reshape(-1)[:len(want)] == want).all()
    assert 1 not in _duos(capfd)
    monkeypatch.delenv("SAU_AMD_NO_DUO")
    for lw in ("2", "14"):
        monkeypatch.setenv("SAU_AMD_DUO_LW", lw)
        got = _render(sa, [prg], 132300, stereo)[0]
        assert (np.asarray(got).reshape(-1)[:len(want)] == want).all(), lw
        assert 1 in _duos(capfd)


def test_a_batch_of_small_scripts(sa, oracle, capfd, monkeypatch):
    """Twelve scripts of three or four voices each, one of them a look-back voice (the shape of BASELINE config 4's renders): few
    look-back voices, each spread over the waves of several workgroups, beside the closed-form voices' tasks; scripts of different
    lengths, so that later segments hold fewer voices of either kind."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    monkeypatch.setenv("SAU_AMD_DEBUG_DUO", "1")
    # (config 4's scripts use five wave tables, which leaves its closed-form voices 8 rows per pass on narrow table blocks; these
    #  use one, and would get 12 rows on wide blocks in a launch of their own)
    monkeypatch.setenv("SAU_AMD_MORE_ROWS", "0")
    monkeypatch.setenv("SAU_AMD_NO_WIDE_TABS", "1")
    prgs = []
    for k in range(12):
        ms = 2000 + 250 * (k % 5)
        # (sine tables only: with four wave tables in LDS the launches take 6 rows per pass, and the joint launch is the 8-row one)
        voices = [_pm_voice(3 * k, ms), _r_voice(k, ms) if k % 2 else _fm_voice(2 * k, ms), _pm_voice(3 * k + 3, ms - 400)]
        if k % 3 == 0:
            voices.append(vb.Op("sin", freq=55.0 + k, time_ms=ms, amp=vb.Line(0.8, goal=0.1, shape="exp")))
        prgs.append(vb.build_program(voices))
    want = [oracle.oracle_render(p.ptr, 44100, False, chunk=66150) for p in prgs]

    def render():
        batch = sa.Batch(prgs, 44100)
        out = batch.render(stereo=False, chunk=66150)
        batch.close()
        return out

    got = render()
    for g, w in zip(got, want):
        assert len(g) == len(w) and (np.asarray(g) == w).all()
    d = _duos(capfd)
    assert 1 in d, "the joint launch did not run: " + d.text
    monkeypatch.setenv("SAU_AMD_NO_DUO", "1")
    got = render()
    for g, w in zip(got, want):
        assert (np.asarray(g) == w).all()


def test_only_one_kind_present(sa, oracle, capfd, monkeypatch):
    """The host knows which voices MAY take the look-back, analyze_kernel which do: a segment whose voices all take the look-back,
    and runs whose later segments hold closed-form voices only (the swept ones have ended) -- the joint launch gives every wave to
    the list that has voices."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    monkeypatch.setenv("SAU_AMD_DEBUG_DUO", "1")
    only_look = [_fm_voice(i, 1500) for i in range(40)]
    prg = vb.build_program(only_look)
    want = oracle.oracle_render(prg.ptr, 44100, False, chunk=66150)
    got = _render(sa, [prg], 66150, False)[0]
    assert (np.asarray(got)[:len(want)] == want).all()
    # voices with a swept frequency (running sums) that end early beside PM voices, in runs short enough that later segments hold
    # closed-form voices only
    mixed = [vb.Op("sin", freq=vb.Line(200.0 + i, goal=300.0 + 2 * i, shape=("lin", "exp")[i % 4 // 2]), time_ms=600) if i % 2 else _pm_voice(i, 1500)
             for i in range(64)]
    prg = vb.build_program(mixed)
    want = oracle.oracle_render(prg.ptr, 44100, False, chunk=22050)
    batch = sa.Batch([prg], 44100)
    got = batch.render(stereo=False, chunk=22050)[0]
    batch.close()
    assert len(got) == len(want) and (np.asarray(got) == want).all()


def test_two_batches_in_flight(sa, oracle, monkeypatch):
    """bench.py's config 4 since round 6: the next batch's generators are created and their runs issued while the device still
    renders this batch's (two batches in flight, each on a stream of its own; their joint launches take turns through the event
    chain of hip_backend.hip: SpreadLaunchOrder). Two batches of six small scripts, both issued before either is waited for, five
    times over; the PCM read from the device afterwards is the oracle's."""
    import ctypes as C
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    monkeypatch.setenv("SAU_AMD_MORE_ROWS", "0")
    monkeypatch.setenv("SAU_AMD_NO_WIDE_TABS", "1")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    sets = []
    for half in range(2):
        prgs = []
        for k in range(6):
            ms = 1500 + 200 * k
            prgs.append(vb.build_program([_pm_voice(3 * k + half, ms), _r_voice(k + 7 * half, ms) if k % 2 else _fm_voice(2 * k + half, ms),
                                          _pm_voice(3 * k + 3 + half, ms)]))
        frames = 44100 * 3
        want = [oracle.oracle_render(p.ptr, 44100, False, chunk=frames) for p in prgs]
        sets.append((prgs, want, frames))
    for rep in range(5):
        live = []
        for prgs, want, frames in sets:  # both issued, neither waited for
            b = sa.Batch(prgs, 44100)
            b.run(frames, stereo=False, fetch=False)
            live.append((b, want, frames))
        for b, want, frames in live:
            b.sync()
            for i, w in enumerate(want):
                got = np.zeros(len(w), np.int16)
                assert hip.hipMemcpy(got.ctypes.data, b.device_pcm(i), got.nbytes, 2) == 0
                assert (got == w).all(), (rep, i)
            b.close()
