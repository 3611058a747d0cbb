"""The compiled reference's loop tails (`cub` lines) -- the product's default, SAU_AMD_LOOP_TAILS=1.

gcc gives the scalar epilogues of two loops of sau/line.c another association than the vector bodies: the last sample
of a sauLine_fill_cub call of odd length and the last len % 4 samples of a sauLine_map_cub call (saugns_amd/csrc/
sau_dev_math.h: sweep_cub_tail, TailCtx). Where those samples lie depends on the reference's blocks -- 1024 frames from
the start of a span, cut where an operator or the voice stops and where the sweep ends -- so on the host's call size.
Here: the host control plane over the sequential executor against the oracle's mode 2 and the compiled reference at
equal call sizes, bit for bit (CPU); tests/test_gpu_vs_ref.py and the gpu-marked tests below do the same on the device."""
import os

import numpy as np
import pytest

from saugns_amd import voicebank as vb
from saugns_amd.api import (POP_PMOD, POP_FMOD, POP_AMOD, POP_RAMOD, POP_RFMOD, POP_CAMOD, POP_APMOD, POPT_RASEG)
import test_gpu_units as tu

RATE = 44100


def cub_programs():
    """`cub` on every kind of line, under time clipping, with events in mid-sweep, and in R segments."""
    C = lambda v0, g, **kw: vb.Line(v0, goal=g, shape="cub", **kw)
    out = []
    out.append(("amp", [vb.Op("sin", freq=330.0, amp=C(0.0, 1.0), time_ms=77)], ()))
    out.append(("freq", [vb.Op("saw", freq=C(100.0, 900.0), time_ms=91)], ()))
    m = vb.Op("sin", freq=C(2.0, 0.5, ratio=True), amp=C(0.2, 3.0), time_ms=41)  # a modulator shorter than its carrier
    out.append(("nested_clipped", [vb.Op("tri", freq=220.0, amp=0.8, time_ms=93, mods={POP_PMOD: [m]})], ()))
    lfo = vb.Op("sin", freq=7.0, amp=1.0)
    out.append(("range_am_fm", [vb.Op("sin", freq=C(200.0, 400.0), freq2=C(300.0, 100.0), amp=C(1.0, 0.1), amp2=C(0.1, 0.9),
                                      time_ms=120, mods={POP_RAMOD: [lfo], POP_RFMOD: [vb.Op("sin", freq=3.0, amp=1.0)]})], ()))
    out.append(("pan", [vb.Op("sin", freq=500.0, amp=0.7, time_ms=60, pan=C(-1.0, 1.0)),
                        vb.Op("sqr", freq=120.0, amp=0.5, time_ms=85, pan=C(0.5, -0.5),
                              mods={POP_CAMOD: [vb.Op("sin", freq=9.0, amp=0.3)]})], ()))
    out.append(("selfmod", [vb.Op("sin", freq=180.0, pm_a=C(0.1, 0.9), amp=C(0.9, 0.2), time_ms=70)], ()))
    for func in (0, 1, 4):
        for flags in (0, 1, 9, 16):
            out.append((f"Rcub_f{func}_o{flags}", [vb.Op(freq=150.0 + 13 * flags, amp=0.8, time_ms=66 + func, op_type=POPT_RASEG,
                                                         seed=77 + flags, ras=("cub", func, flags)),
                                                   vb.Op("sin", freq=300.0, time_ms=58, mods={POP_PMOD: [
                                                       vb.Op(freq=vb.Line(0.5, ratio=True), amp=2.0, time_ms=33, op_type=POPT_RASEG,
                                                             seed=5, ras=("cub", func, flags))]})], ()))
    # events in mid-sweep: a new goal samples the running `cub` sweep (a fill of length 1: the scalar tail)
    carr = vb.Op("sin", freq=C(300.0, 600.0), amp=C(0.2, 1.0), time_ms=150)
    ups = [(40, 0, carr, {"amp": vb.Line(0.0, goal=0.5, shape="cub", state=False)}),
           (75, 0, carr, {"freq": vb.Line(0.0, goal=200.0, shape="cub", state=False)}),
           (110, 0, carr, {"amp": vb.Line(0.0, goal=0.9, shape="lin", state=False)})]
    out.append(("regoal", [carr], ups))
    return out


def _cases():
    for name, voices, ups in cub_programs():
        yield name, vb.build_program(voices, updates=ups), True
    for seed in (5, 9, 15, 22):  # random graphs known to hold a `cub` line behind heavy modulation
        rng = np.random.default_rng(1000 + seed)
        voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        yield f"graph{seed}", vb.build_program(voices), bool(seed & 1)
    for seed in (1, 23, 26, 28, 35, 39):
        rng = np.random.default_rng(5000 + seed)
        voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        tu._random_starts(rng, voices)
        yield f"events{seed}", vb.build_program(voices, updates=tu._random_updates(rng, voices)), bool(seed & 1)


CALLS = (11289, 1746, 1023, 4000000)


def test_sequential_executor_reproduces_the_loop_tails(sa, oracle, seqexec, tails_on):
    oracle.oracle().ora_set_fastmath_forms(2)
    try:
        for name, prg, stereo in _cases():
            for chunk in CALLS:
                want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
                if oracle.have_ref():
                    assert (want == oracle.ref_render(prg.ptr, RATE, stereo, chunk=chunk)).all(), (name, chunk)
                for blk in (1016, 333):
                    got = sa.Batch([prg], RATE, backend=seqexec.seq_backend_create(blk)).render(stereo=stereo, chunk=chunk)[0]
                    assert len(got) == len(want) and (got == want).all(), (name, chunk, blk)
                # the drop-in generator: engine runs that cover several host calls of this size
                if chunk < 100000:
                    g = sa.Generator(prg, RATE, backend=seqexec.seq_backend_create(777))
                    got = g.render(stereo=stereo, chunk=chunk)
                    assert len(got) == len(want) and (got == want).all(), (name, chunk, "drop-in")
    finally:
        oracle.oracle().ora_set_fastmath_forms(1)


def test_loop_tails_matter_and_can_be_switched_off(sa, oracle, seqexec, tails_off):
    """The two forms really differ on these programs (else the test above proves nothing), and with the switch off the
    executor gives the loop bodies' forms (mode 1), whatever the call size."""
    n_diff = 0
    for name, prg, stereo in _cases():
        oracle.oracle().ora_set_fastmath_forms(2)
        m2 = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=1746)
        oracle.oracle().ora_set_fastmath_forms(1)
        m1 = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=1746)
        n_diff += bool((m1 != m2).any())
        got = sa.Batch([prg], RATE, backend=seqexec.seq_backend_create(1016)).render(stereo=stereo, chunk=1746)[0]
        assert (got == m1).all(), name  # (SAU_AMD_LOOP_TAILS=0: the tails_off fixture)
    assert n_diff >= 3, n_diff  # (an ulp in a ramp reaches the int16 output only behind enough modulation)


@pytest.mark.gpu
def test_device_reproduces_the_loop_tails(sa, oracle, tails_on):
    oracle.oracle().ora_set_fastmath_forms(2)
    try:
        for name, prg, stereo in _cases():
            for chunk in CALLS:
                want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
                got = sa.Batch([prg], RATE).render(stereo=stereo, chunk=chunk)[0]
                assert len(got) == len(want) and (got == want).all(), (name, chunk)
                if chunk < 100000:
                    got = sa.Generator(prg, RATE).render(stereo=stereo, chunk=chunk)
                    assert len(got) == len(want) and (got == want).all(), (name, chunk, "drop-in")
        # programs side by side (other programs' events cut this program's segments, not its spans)
        prgs = [p for _, p, st in _cases() if not st][:10]
        b = sa.Batch(prgs, RATE)
        b.set_call_len(1746)
        outs = b.render(stereo=False, chunk=1746 * 3)
        for p, got in zip(prgs, outs):
            want = oracle.oracle_render(p.ptr, RATE, False, chunk=1746)
            assert len(got) == len(want) and (got == want).all()
    finally:
        oracle.oracle().ora_set_fastmath_forms(1)


def _rcub_bank(n=16):
    """R oscillators with `cub` segments that the time-parallel path takes (steady, nothing modulated): carriers, one with
    a shorter R-cub modulator nested in it (the operator's stop cuts the reference's block: TailCtx.rem)."""
    voices = [vb.Op(op_type=POPT_RASEG, ras=("cub", k % 6, (0, 1, 9, 16)[k % 4]), seed=11 + k, freq=90.0 + 7 * k, amp=0.9,
                    time_ms=400) for k in range(n)]
    voices.append(vb.Op("sin", freq=250.0, time_ms=300, mods={POP_PMOD: [
        vb.Op(op_type=POPT_RASEG, ras=("cub", 1, 9), seed=3, freq=vb.Line(1.5, ratio=True), amp=0.7, time_ms=170)]}))
    # ... and with running-sum phases (the look-back build with the tail code, fast_kernel<4, 2, true>)
    voices.append(vb.Op(op_type=POPT_RASEG, ras=("cub", 0, 1), seed=21, freq=vb.Line(120.0, goal=420.0, shape="lin"), amp=0.8, time_ms=350))
    voices.append(vb.Op(op_type=POPT_RASEG, ras=("cub", 3, 0), seed=22, freq=200.0, amp=0.8, time_ms=350,
                        mods={POP_FMOD: [vb.Op("sin", freq=5.0, amp=40.0)]}))
    return voices


def test_r_segment_map_tails_matter_at_tiny_call_sizes(oracle):
    """With host calls of a few frames almost every sample is among the last len % 4 of a sauLine_map_cub call: there the
    two forms differ in the int16 output (at ordinary call sizes they practically never do -- which is how the
    time-parallel build could lack the map's tails for most of round 3 without any test noticing)."""
    n = 0
    for v in _rcub_bank():
        prg = vb.build_program([v])
        oracle.oracle().ora_set_fastmath_forms(2)
        m2 = oracle.oracle_render(prg.ptr, RATE, False, chunk=3)
        oracle.oracle().ora_set_fastmath_forms(1)
        m1 = oracle.oracle_render(prg.ptr, RATE, False, chunk=3)
        n += int((m1 != m2).sum())
    assert n >= 20, n


@pytest.mark.gpu
def test_device_reproduces_the_map_tails_at_tiny_call_sizes(sa, oracle, tails_on):
    """R oscillators with `cub` segments on the time-parallel path (fast_kernel<4, 0, true>: FT_CUBTAIL) with host calls of
    3 and 7 frames, engine runs of thousands of such calls: bit-exact vs the oracle's mode 2 (= the compiled reference),
    voice by voice and as one bank; and off the block loop."""
    oracle.oracle().ora_set_fastmath_forms(2)
    try:
        bank = _rcub_bank()
        for call in (3, 7):
            for vs in [[v] for v in bank] + [bank]:
                prg = vb.build_program(vs)
                want = oracle.oracle_render(prg.ptr, RATE, False, chunk=call)
                b = sa.Batch([prg], RATE)
                b.set_call_len(call)
                b.set_timing(2)
                got = b.render(stereo=False, chunk=call * 3000)[0]
                assert len(got) == len(want) and (got == want).all(), (call, len(vs), int((got != want).sum()))
            assert b.timing_ex()["block_ms"] < 1.0  # (the bank: every voice but the one whose modulator stops stays closed-form)
    finally:
        oracle.oracle().ora_set_fastmath_forms(1)
