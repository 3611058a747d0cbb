"""GPU unit parity: one feature of the operator graph at a time, bit-exact vs the oracle."""
import os

import numpy as np
import pytest

from conftest import max_diff, ORACLE_FORMS
from saugns_amd import voicebank as vb
from saugns_amd.api import (LINES, POP_AMOD, POP_APMOD, POP_CAMOD, POP_FMOD, POP_FPMOD, POP_PMOD,
                            POP_RAMOD, POP_RFMOD, POPT_NOISE, POPT_RASEG, POPT_WAVE, WAVES)

pytestmark = pytest.mark.gpu
RATE = 44100


def check(sa, oracle, voices, frames=None, stereo=False, chunk=4000):
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.build_program(voices)
    # (both sides make calls of `chunk` frames: the reference build's loop tails -- oracle mode 2, the suite's default --
    #  fall where its blocks end, which depends on the host's call size)
    want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
    got = sa.Generator(prg, RATE).render(stereo=stereo, chunk=chunk)
    assert len(got) == len(want)
    d = np.nonzero(got != want)[0]
    assert len(d) == 0, (f"{len(d)} samples differ, first at {d[0]}: got "
                         f"{got[d[0]:d[0]+6].tolist()} want {want[d[0]:d[0]+6].tolist()}")


@pytest.mark.parametrize("shape", LINES)
def test_carrier_amp_sweep(sa, oracle, shape):
    check(sa, oracle, [vb.Op("sin", freq=440.0, amp=vb.Line(0.0, goal=2.0, shape=shape), time_ms=100)])


@pytest.mark.parametrize("shape", LINES)
def test_carrier_freq_sweep(sa, oracle, shape):
    check(sa, oracle, [vb.Op("sin", freq=vb.Line(220.0, goal=1760.0, shape=shape), time_ms=100)])


@pytest.mark.parametrize("wave", WAVES)
def test_wave_types(sa, oracle, wave):
    check(sa, oracle, [vb.Op(wave, freq=333.3, time_ms=60, phase=0.123)])


def test_pm_mod_amp_sweep(sa, oracle):
    m = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=vb.Line(0.0, goal=2.0, shape="xpe"))
    check(sa, oracle, [vb.Op("sin", freq=146.832, time_ms=100, mods={POP_PMOD: [m]})])


def test_two_pmods(sa, oracle):
    m1 = vb.Op("sin", freq=vb.Line(3 / 7, ratio=True), amp=0.8)
    m2 = vb.Op("tri", freq=vb.Line(1.5, ratio=True), amp=0.5)
    check(sa, oracle, [vb.Op("sin", freq=146.832, time_ms=100, mods={POP_PMOD: [m1, m2]})])


def test_expiring_modulator(sa, oracle):
    m1 = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0, time_ms=30)
    m2 = vb.Op("sin", freq=1000 / 3, amp=vb.Line(1.0, goal=0.0, shape="xpe"))
    check(sa, oracle, [vb.Op("sin", freq=146.832, time_ms=100, mods={POP_PMOD: [m1, m2]})])


def test_fm_and_range_fm(sa, oracle):
    f1 = vb.Op("sin", freq=5.0, amp=30.0)
    r1 = vb.Op("sin", freq=vb.Line(0.5, ratio=True), amp=1.0)
    check(sa, oracle, [vb.Op("saw", freq=200.0, freq2=400.0, time_ms=120,
                             mods={POP_FMOD: [f1], POP_RFMOD: [r1]})])


def test_am_and_range_am(sa, oracle):
    a1 = vb.Op("sin", freq=7.0, amp=0.3)
    r1 = vb.Op("tri", freq=3.0, amp=1.0)
    check(sa, oracle, [vb.Op("sin", freq=300.0, amp=0.8, amp2=0.1, time_ms=120,
                             mods={POP_AMOD: [a1], POP_RAMOD: [r1]})])


def test_freq_scaled_pm(sa, oracle):
    p = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=2.0)
    check(sa, oracle, [vb.Op("sin", freq=444.0, time_ms=80, mods={POP_FPMOD: [p]})])
    q = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0)
    p2 = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=2.0)
    check(sa, oracle, [vb.Op("sin", freq=444.0, time_ms=80, mods={POP_FPMOD: [p2], POP_PMOD: [q]})])


def test_self_modulation(sa, oracle):
    check(sa, oracle, [vb.Op("sin", freq=220.0, pm_a=0.7, time_ms=100)])
    check(sa, oracle, [vb.Op("saw", freq=220.0, pm_a=vb.Line(0.1, goal=2.0, shape="lin"), time_ms=100)])
    ap = vb.Op("sin", freq=2.0, amp=0.5)
    check(sa, oracle, [vb.Op("sin", freq=220.0, pm_a=0.3, time_ms=100, mods={POP_APMOD: [ap]})])


def test_zero_frequency_fill_forward(sa, oracle):
    """dphase == 0 runs: the differentiator holds its previous output (wosc.h:251-252)."""
    check(sa, oracle, [vb.Op("sin", freq=vb.Line(0.0), amp=1.0, time_ms=50, phase=0.3)])
    check(sa, oracle, [vb.Op("sin", freq=vb.Line(300.0, goal=-300.0, shape="lin"), time_ms=200)])


def test_many_voices_stereo(sa, oracle):
    voices = [vb.Op(WAVES[i % 12], freq=100.0 + 37.7 * i, time_ms=40 + 3 * i, phase=i * 0.1)
              for i in range(40)]
    check(sa, oracle, voices, stereo=True)


def test_line_arithmetic_device_vs_host(sa, hooks):
    """sau_dev_math.h compiled for gfx950 == compiled for the host, bit for bit (the probes live in the hook library)."""
    L = hooks
    rng = np.random.default_rng(7)
    n = 1016
    mul = rng.uniform(50, 5000, n).astype(np.float32)
    for shape in range(13):
        for (v0, vt, pos, end, flags) in [(0.0, 2.0, 0, 44100, 0x35), (1.0, 0.0, 100, 4410, 0x35),
                                          (0.5, 3.0, 0, 300, 0x3d), (220.0, 880.0, 7, 2000, 0x37),
                                          (3.0, 0.0, 0, 0, 0x33), (1.0, 0.0, 0, 0, 0x31)]:
            for m in (None, mul):
                st = np.zeros(6, np.uint32)
                st[:2] = np.array([v0, vt], np.float32).view(np.uint32)
                st[2:] = [pos, end, shape, flags]
                sd, sh = st.copy(), st.copy()
                od, oh = np.zeros(n, np.float32), np.zeros(n, np.float32)
                assert L.sauAmd_kat_line_device(sd.ctypes.data, n, m.ctypes.data if m is not None else None, od.ctypes.data)
                L.sauAmd_kat_line_host(sh.ctypes.data, n, m.ctypes.data if m is not None else None, oh.ctypes.data)
                bad = np.nonzero(od.view(np.uint32) != oh.view(np.uint32))[0]
                assert len(bad) == 0, (shape, v0, vt, pos, end, hex(flags), m is not None, bad[:4], od[bad[:4]], oh[bad[:4]])
                assert (sd == sh).all(), (shape, sd, sh)


def check_runs(sa, oracle, voices, chunk, stereo=False):
    """As check(), but through the batch API with one engine run per `chunk` frames, so that
    later runs start from the state earlier ones left (expired operators, finished ramps)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.build_program(voices)
    want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)  # (a run of the batch API stands for one host call)
    got = sa.Batch([prg], RATE).render(stereo=stereo, chunk=chunk)[0]
    assert len(got) == len(want)
    d = np.nonzero(got != want)[0]
    assert len(d) == 0, f"{len(d)} samples differ, first at {d[0]}"


@pytest.mark.parametrize("chunk", [1500, 4410])
def test_operators_out_of_time_across_runs(sa, oracle, chunk):
    """A modulator that has run out of time yields silence and stands still (generator.c:686-700),
    as first or later member of its list, with a subtree of its own, in PM and AM lists."""
    first = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0, time_ms=30)
    later = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.6)
    check_runs(sa, oracle, [vb.Op("sin", freq=146.832, time_ms=200, mods={POP_PMOD: [first, later]})], chunk)
    first = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0)
    later = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.6, time_ms=45)
    check_runs(sa, oracle, [vb.Op("sin", freq=146.832, time_ms=200, mods={POP_PMOD: [first, later]})], chunk)
    inner = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.5, time_ms=150)
    outer = vb.Op("sin", freq=vb.Line(1.5, ratio=True), amp=0.9, time_ms=50, mods={POP_PMOD: [inner]})
    check_runs(sa, oracle, [vb.Op("sin", freq=220.0, time_ms=200, mods={POP_PMOD: [outer]})], chunk)
    am = vb.Op("sin", freq=7.0, amp=0.4, time_ms=60)
    check_runs(sa, oracle, [vb.Op("sin", freq=330.0, time_ms=200, mods={POP_AMOD: [am]}),
                            vb.Op("sin", freq=110.0, time_ms=120, amp=vb.Line(1.0, goal=0.1, shape="lin"))],
               chunk, stereo=True)


def test_differentiator_division_exhaustive(sa, hooks):
    """diff_scale / (float)dphase (wosc.h:253) is computed as v_rcp_f32 plus one residual
    correction; that is correctly rounded for these operands only, so it is checked against
    IEEE division for every f32 divisor of magnitude 1..2^31 (both signs) and the diff_scale
    of every wave. The probe is shown to discriminate: the uncorrected product fails."""
    import ctypes as C
    fn = hooks.sauAmd_kat_div_device
    for wave in range(12):
        first = C.c_uint32()
        assert fn(wave, 0, C.byref(first)) == 0, (wave, hex(first.value))
        assert fn(wave, 1, C.byref(first)) > 0


def test_a_batch_on_a_named_device(sa, oracle):
    """sauAmd_create_Batch_on (include/saugns_amd.h, round 6): a batch on the HIP device the caller names -- the same PCM as the
    default batch's and the oracle's on device 0, two of them rendering side by side; a device the process does not have, and a
    negative one, are refused with a message and NULL, as a constructor of the reference's is (generator.c:191-231)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.build_program(vb.config3_voices(8, 1) + [vb.Op("sin", freq=220.0, pm_a=0.6, amp=0.4, time_ms=1000)])
    want = oracle.oracle_render(prg.ptr, RATE, False, chunk=44100)
    a = sa.Batch([prg], RATE, device=0)
    b = sa.Batch([prg], RATE, device=0)
    got_a = a.render(stereo=False, chunk=44100)[0]
    got_b = b.render(stereo=False, chunk=44100)[0]
    a.close(); b.close()
    assert len(got_a) == len(want) and (np.asarray(got_a) == want).all() and (np.asarray(got_b) == want).all()
    n = sa.lib().sauAmd_device_count()
    for bad in (n, -1):
        with pytest.raises(Exception):
            sa.Batch([prg], RATE, device=bad)
        assert sa.last_error()


@pytest.mark.parametrize("scattered", [0, 1])
def test_rint64_and_the_gauss_noise_conversions_on_every_bit_pattern(hooks, scattered):
    """rint64() (llrintf kept as int64, sau/math.h:63-64: phase increments of R oscillators and swept frequencies) has a short form
    for waves whose values are all below 2^31 (round 6) beside the compiler's eleven-instruction conversion, and franssgauss32()
    (noise.h:90-98) scales its integers in f32 where the reference multiplies in double and rounds: both against independent
    forms on the device, for all 2^32 bit patterns, with a wave's lanes alike in magnitude and with them scattered."""
    import ctypes as C
    fb = C.c_uint32(0)
    bad = hooks.sauAmd_kat_rint64_device(scattered, C.byref(fb))
    assert bad == 0, (bad, hex(fb.value))


def test_wave_scan_of_64_bit_values_counts_its_carries(hooks):
    """k_wave_scan.h: the inclusive 64-bit scan (two 32-bit DPP scans + the wraps of the low words counted from one compare) and the
    64-bit sum, against numpy's wrapping cumsum -- positive and negative increments (high words 0 and ~0: rasg.h:154-155 with a
    negative rate), low words that wrap at every lane or never, random 64-bit values."""
    rng = np.random.default_rng(606)
    waves = []
    waves.append(rng.integers(0, 1 << 32, 64, dtype=np.uint64))                       # small positive: hi = 0, wraps now and then
    waves.append((-rng.integers(1, 1 << 32, 64).astype(np.int64)).astype(np.uint64))  # small negative: hi = ~0
    waves.append(np.full(64, 0xffffffff, dtype=np.uint64))                            # wraps at every lane but the first
    waves.append(np.full(64, 0x80000000, dtype=np.uint64))                            # wraps at every second lane
    waves.append(np.zeros(64, dtype=np.uint64))
    waves.append(np.full(64, 0xffffffffffffffff, dtype=np.uint64))
    w = np.zeros(64, dtype=np.uint64); w[17] = 0xffffffff; w[18] = 1; waves.append(w)  # one wrap, exactly to zero
    for _ in range(57):
        waves.append(rng.integers(0, 1 << 63, 64, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 64, dtype=np.uint64))
    a = np.ascontiguousarray(np.stack(waves))
    scan = np.zeros_like(a)
    tot = np.zeros(len(waves), dtype=np.uint64)
    assert hooks.sauAmd_kat_scan64_device(a.ctypes.data, scan.ctypes.data, tot.ctypes.data, len(waves))
    with np.errstate(over="ignore"):
        want = np.cumsum(a, axis=1, dtype=np.uint64)
    assert (scan == want).all(), np.argwhere(scan != want)[:4]
    assert (tot == want[:, 63]).all()


@pytest.mark.parametrize("chunk", [1500, 20000])
def test_running_sum_phases_across_runs(sa, oracle, chunk):
    """Ramped and modulated frequencies (phase = running sum of per-frame increments,
    wosc.h:135-169) through the time-parallel kernel: two-pass voices (FM by constant-frequency
    modulators, glide with ratio children, range-FM) and in-order voices (nested FM), with the
    state handed from one engine run to the next."""
    fm = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=40.0,
               mods={POP_PMOD: [vb.Op("tri", freq=vb.Line(3.0, ratio=True), amp=0.5)]})
    check_runs(sa, oracle, [vb.Op("sin", freq=220.0, time_ms=300, mods={POP_FMOD: [fm]})], chunk)
    child = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.7)
    check_runs(sa, oracle, [vb.Op("sin", freq=vb.Line(110.0, goal=440.0, shape="exp"), time_ms=300,
                                  mods={POP_PMOD: [child]})], chunk)
    r1 = vb.Op("sin", freq=vb.Line(0.5, ratio=True), amp=1.0)
    f1 = vb.Op("sin", freq=5.0, amp=30.0)
    check_runs(sa, oracle, [vb.Op("saw", freq=200.0, freq2=400.0, time_ms=300,
                                  mods={POP_RFMOD: [r1], POP_FMOD: [f1]})], chunk)
    inner = vb.Op("sin", freq=7.0, amp=15.0)
    outer = vb.Op("sin", freq=55.0, amp=60.0, mods={POP_FMOD: [inner]})
    check_runs(sa, oracle, [vb.Op("sin", freq=330.0, time_ms=300, mods={POP_FMOD: [outer]}),
                            vb.Op("sin", freq=vb.Line(300.0, goal=100.0, shape="lin"), time_ms=200,
                                  mods={POP_FPMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=1.5)]})],
               chunk, stereo=True)


def _random_voice(rng, depth=0):
    """A random operator tree: every modulator list kind, ramps on any line, ratio and absolute
    frequencies, operators shorter than their carrier, now and then self-modulation."""
    lists = [POP_PMOD, POP_FMOD, POP_AMOD, POP_RAMOD, POP_RFMOD, POP_FPMOD, POP_APMOD]
    shape = lambda: LINES[int(rng.integers(len(LINES)))]
    if depth == 0:
        f = float(rng.uniform(60, 900))
        freq = vb.Line(f, goal=float(rng.uniform(60, 900)), shape=shape()) if rng.random() < 0.3 else f
    else:
        r = float(rng.choice([0.25, 0.5, 1.0, 1.5, 2.0, 3.0]))
        if rng.random() < 0.6:
            freq = vb.Line(r, goal=r * float(rng.uniform(0.5, 2)), shape=shape(), ratio=True) \
                if rng.random() < 0.2 else vb.Line(r, ratio=True)
        else:
            freq = float(rng.uniform(0.5, 400))
    a0 = float(rng.uniform(0.1, 1.0)) * (40.0 if rng.random() < 0.15 else 1.0)
    amp = vb.Line(a0, goal=float(rng.uniform(0, 1)), shape=shape()) if rng.random() < 0.3 else a0
    mods = {}
    if depth < 3:
        for use in lists:
            if rng.random() < (0.35 if depth == 0 else 0.2):
                mods[use] = [_random_voice(rng, depth + 1) for _ in range(int(rng.integers(1, 3)))]
    if depth == 0 and rng.random() < 0.25:  # pan modulators (generator.c:749-788)
        mods[POP_CAMOD] = [_random_voice(rng, depth + 1) for _ in range(int(rng.integers(1, 3)))]
    kw = {}
    if POP_RAMOD in mods:
        kw["amp2"] = float(rng.uniform(0, 1))
    if POP_RFMOD in mods:
        kw["freq2"] = float(rng.uniform(60, 900)) if depth == 0 else vb.Line(float(rng.uniform(0.5, 3)), ratio=True)
    if rng.random() < 0.08 or POP_APMOD in mods:  # feedback, constant or ramped amount
        p0 = float(rng.uniform(0.1, 0.9))
        kw["pm_a"] = vb.Line(p0, goal=float(rng.uniform(0, 1.5)), shape=shape()) if rng.random() < 0.3 else p0
    time_ms = int(rng.integers(40, 160)) if depth == 0 else (int(rng.integers(10, 120)) if rng.random() < 0.2 else None)
    if depth == 0 and rng.random() < 0.35:  # carrier pan, held or ramped
        p0 = float(rng.uniform(-1, 1))
        kw["pan"] = vb.Line(p0, goal=float(rng.uniform(-1, 1)), shape=shape()) if rng.random() < 0.6 else p0
    kind = rng.random()
    if kind < 0.15:  # R oscillator: random line shape, function and function flags
        kw.update(op_type=POPT_RASEG, seed=int(rng.integers(1 << 32)),
                  ras=(LINES[int(rng.integers(len(LINES)))], int(rng.integers(6)), int(rng.integers(32))))
    elif kind < 0.22 and depth > 0 and not mods:  # noise source
        kw.pop("freq2", None)
        return vb.Op(amp=amp, time_ms=time_ms, op_type=POPT_NOISE, seed=int(rng.integers(1 << 32)),
                     noise=int(rng.integers(7)), **{k: v for k, v in kw.items() if k == "amp2"})
    return vb.Op(WAVES[int(rng.integers(len(WAVES)))], freq=freq, amp=amp, time_ms=time_ms,
                 phase=float(rng.uniform(0, 1)), mods=mods, **kw)


def _push_extremes(rng, voices):
    """Frequencies of zero, a few microhertz, beyond Nyquist and negative; amplitudes of thousands and negative ones;
    modulation amounts of hundreds of cycles; operators of 0, 1 and 2 ms."""
    def walk(op, top):
        for name in ("freq", "freq2", "amp", "amp2", "pm_a"):
            ln = getattr(op, name, None)
            if ln is None or rng.random() > 0.3:
                continue
            if name in ("freq", "freq2") and not ln.ratio:
                pick = [0.0, 1e-6, -1e-6, 3e4, -3e4, 1e6, 22050.0, 0.5]
            elif name in ("freq", "freq2"):
                pick = [0.0, 1e-4, 1e3, -2.0, 7.25]
            elif name == "pm_a":
                pick = [0.0, 1e-7, 40.0, 2.5, -1.0]
            else:
                pick = [0.0, -1.0, 1e4, 1e-30, 300.0, -77.0]
            ln.v0 = float(rng.choice(pick))
            if ln.goal is not None and rng.random() < 0.5:
                ln.goal = float(rng.choice(pick))
        if not top and rng.random() < 0.2:
            op.time_ms = int(rng.choice([0, 1, 2, 5]))
        for lst in op.mods.values():
            for m in lst:
                walk(m, False)
    for v in voices:
        walk(v, True)


def _tall_tree(rng):
    """A chain of 2..256 operators, each level's list kind, operator type and parameters drawn at random; now and then a
    second member in a list (a leaf)."""
    depth = int(rng.choice([2, 10, 40, 100, 121, 130, 200, 255, 256]))
    op = None
    for d in range(depth):
        top = d == depth - 1
        node = _random_voice(rng, depth=3 if not top else 0)  # (depth 3: no random children of its own)
        node.mods = {}
        if top:
            node.time_ms = int(rng.integers(20, 60))
        elif node.time_ms is not None and rng.random() < 0.9:
            node.time_ms = None
        if op is not None:
            use = int(rng.choice([POP_PMOD, POP_FMOD, POP_AMOD, POP_RAMOD, POP_RFMOD, POP_FPMOD] + ([POP_APMOD] if node.op_type != POPT_NOISE else [])))
            if node.op_type == POPT_NOISE and use in (POP_PMOD, POP_FMOD, POP_RFMOD, POP_FPMOD):
                use = POP_AMOD
            members = [op] + ([_random_voice(rng, depth=3)] if rng.random() < 0.05 else [])  # (a leaf: the path stays within 256)
            node.mods = {use: members}
            if use == POP_RAMOD and node.amp2 is None: node.amp2 = vb.Line(float(rng.uniform(0, 1)))
            if use == POP_RFMOD and node.freq2 is None and node.freq is not None:
                node.freq2 = vb.Line(float(rng.uniform(0.5, 3)), ratio=True) if not top else vb.Line(float(rng.uniform(60, 900)))
            if use == POP_APMOD and node.pm_a is None: node.pm_a = vb.Line(float(rng.uniform(0.1, 0.9)))
        op = node
    return op


def _random_starts(rng, voices):
    """Some voices begin later than the first (a script's timing separators)."""
    for carr in voices[1:]:
        if rng.random() < 0.5:
            carr.start_ms = int(rng.integers(1, 90))


def _random_updates(rng, voices):
    """Later events on random operators of random voices: new values and/or new ramps for
    amplitude, frequency or pan, and new durations for carriers (as compound steps give)."""
    def nodes(op, acc):
        acc.append(op)
        for lst in op.mods.values():
            for m in lst:
                nodes(m, acc)
        return acc
    shape = lambda: LINES[int(rng.integers(len(LINES)))]
    ups = []
    for vi, carr in enumerate(voices):
        t0 = getattr(carr, "start_ms", 0) or 0
        t_end = t0 + carr.time_ms
        for _ in range(int(rng.integers(0, 4))):
            at = int(rng.integers(t0 + 5, max(t0 + 6, t_end - 5)))
            ops = nodes(carr, [])
            op = ops[int(rng.integers(len(ops)))]
            what = {}
            kind = rng.random()
            has_freq = op.op_type != POPT_NOISE and op.freq is not None
            if kind < 0.45 or not has_freq:
                g = float(rng.uniform(0, 1))
                what["amp"] = vb.Line(float(rng.uniform(0.1, 1)), goal=g if rng.random() < 0.7 else None,
                                      shape=shape(), state=bool(rng.random() < 0.5) or rng.random() < 0.3)
                if what["amp"].goal is None:
                    what["amp"].state = True
            elif kind < 0.85:
                ratio = op.freq.ratio
                v = float(rng.choice([0.5, 1.0, 2.0, 3.0])) if ratio else float(rng.uniform(50, 800))
                g = v * float(rng.uniform(0.5, 2.0))
                what["freq"] = vb.Line(v, goal=g if rng.random() < 0.6 else None, shape=shape(), ratio=ratio,
                                       state=bool(rng.random() < 0.6))
                if what["freq"].goal is None:
                    what["freq"].state = True
                elif not what["freq"].state and op is not carr and rng.random() < 0.25:
                    # a goal of the other kind than the state (absolute <-> ratio of the parent): the
                    # ramp starts from the state rescaled by the parent's frequency (sau/line.c:358-370)
                    what["freq"].ratio = not ratio
                    what["freq"].goal = float(rng.uniform(50, 800)) if ratio else float(rng.choice([0.5, 1.0, 2.0, 3.0]))
            elif op is carr:
                what["pan"] = vb.Line(float(rng.uniform(-1, 1)), goal=float(rng.uniform(-1, 1)) if rng.random() < 0.6 else None,
                                      shape=shape())
            else:
                what["amp"] = vb.Line(float(rng.uniform(0.1, 1)))
            if op is carr and rng.random() < 0.5:
                what["time_ms"] = int(rng.integers(10, 120))
                t_end = at + what["time_ms"]
            # what else an event may carry (generator.c:283-343): a new wave / noise / R option
            # word, a phase or seed reset, a self-modulation amount, replaced modulator lists
            if rng.random() < 0.25:
                if op.op_type == POPT_NOISE:
                    what["noise"] = int(rng.integers(7))
                elif op.op_type == POPT_RASEG:
                    what["ras"] = (LINES[int(rng.integers(len(LINES)))], int(rng.integers(6)), int(rng.integers(32)))
                else:
                    what["wave"] = WAVES[int(rng.integers(len(WAVES)))]
            if rng.random() < 0.2 and op.op_type != POPT_NOISE:
                what["phase"] = float(rng.uniform(0, 1))
            if rng.random() < 0.15 and op.op_type != POPT_WAVE:
                what["seed"] = int(rng.integers(1 << 32))
            if rng.random() < 0.12 and op.op_type != POPT_NOISE and op.pm_a is not None:
                what["pm_a"] = vb.Line(float(rng.uniform(0, 1)), goal=float(rng.uniform(0, 1)) if rng.random() < 0.5 else None,
                                      shape=shape())
            if rng.random() < 0.2 and op.mods:
                use = sorted(op.mods)[int(rng.integers(len(op.mods)))]
                if use != POP_CAMOD or op is carr:
                    keep_n = int(rng.integers(0, len(op.mods[use]) + 1))
                    what["mods"] = {use: list(op.mods[use][:keep_n])}
            ups.append((at, vi, op, what))
    return ups


def extreme_program(seed):
    """Program `seed` of tests/tools/gpu_vs_ref_sweep.py's `extreme` mode: a random graph with parameters pushed to extremes.
    -> (program, rate, call size)"""
    rng = np.random.default_rng(20000 + seed)
    voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    ups = ()
    if seed % 2:
        _random_starts(rng, voices)
        ups = _random_updates(rng, voices)
    _push_extremes(rng, voices)
    rate = int(rng.choice([1000, 3000, 11025, 44100, 192000, 384000]))
    prg = vb.build_program(voices, updates=ups)
    call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
    return prg, rate, call


# seeds of that sweep that differed from the compiled reference before round 3's fixes: frequency-scaled PM beyond 2^63 phase
# units (the host's float -> int64 conversion gives 0x8000...0 there, the device's saturates), feedback that runs to infinity
# (a NaN in the mix leaves the reference build's clamp as -1)
EXTREME_SEEDS = (211, 278, 344, 449, 612, 820, 871, 1168, 1274, 1377, 1417, 1482, 1498, 1808, 2127, 2274, 2802, 2865, 10687)


@pytest.mark.parametrize("seed", EXTREME_SEEDS)
def test_extreme_parameters(sa, oracle, seed):
    """Frequencies of zero, microhertz, megahertz and negative, amplitudes of tens of thousands, modulation amounts of hundreds
    of cycles, operators of 0-2 ms, sample rates of 1 kHz to 384 kHz: the programs that exposed where the device's arithmetic
    left the host's at the edges (conversions out of range, NaN in the mix). Bit-exact vs the oracle (pinned against the
    compiled reference on the same programs, tests/test_oracle.py), batch API and drop-in generator."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg, rate, call = extreme_program(seed)
    for stereo in (False, True):
        want = oracle.oracle_render(prg.ptr, rate, stereo, chunk=call)
        b = sa.Batch([prg], rate)
        if call < 300:
            b.set_call_len(call)
        got = b.render(stereo=stereo, chunk=call * 1500 if call < 300 else call)[0]
        assert len(got) == len(want) and (got == want).all(), (seed, stereo, int((got != want).sum()))
    g = sa.Generator(prg, rate)
    got = g.render(stereo=False, chunk=call)
    g.close()
    want = oracle.oracle_render(prg.ptr, rate, False, chunk=call)
    assert len(got) == len(want) and (got == want).all(), (seed, "drop-in")


@pytest.mark.parametrize("seed", range(48))
def test_random_graphs_with_later_events(sa, oracle, seed):
    """Random graphs whose operators receive later events (new values, new ramps, new durations):
    the state every kernel leaves behind must be what the next segment's kernels expect."""
    rng = np.random.default_rng(5000 + seed)
    voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    _random_starts(rng, voices)
    ups = _random_updates(rng, voices)
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.build_program(voices, updates=ups)
    stereo = bool(seed & 1)
    for chunk in (4000000, int(rng.integers(700, 3000))):
        # the same call size on both sides: the reference's output is not always independent of
        # it (found at 96 kHz: a duration change, a value set and a goal-only ramp on one line)
        want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
        got = sa.Batch([prg], RATE).render(stereo=stereo, chunk=chunk)[0]
        assert len(got) == len(want)
        d = np.nonzero(got != want)[0]
        assert len(d) == 0, f"chunk {chunk}: {len(d)} samples differ, first at {d[0]}"


@pytest.mark.parametrize("seed", range(64))
def test_random_operator_graphs(sa, oracle, seed):
    """Random graphs through whichever kernels they land in (closed form, running sums in one,
    two or three passes or in order, block loop), in one engine run and cut into several."""
    rng = np.random.default_rng(1000 + seed)
    voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    check(sa, oracle, voices, stereo=bool(seed & 1))
    check_runs(sa, oracle, voices, chunk=int(rng.integers(700, 3000)), stereo=bool(seed & 1))


@pytest.mark.gpu
def test_deep_and_wide_graphs(sa, oracle):
    """Nesting at the limit (64 levels of PM) and 300 modulators in one list, on the device."""
    from saugns_amd.voicebank import Op, Line
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    op = None
    for d in range(64):
        top = d == 63
        op = Op("sin", freq=200.0 if top else Line(2.0, ratio=True), amp=0.5, time_ms=300 if top else None,
                mods={POP_PMOD: [op]} if op else {})
    mods = [Op("sin", freq=Line(float(1 + i % 7), ratio=True), amp=0.1) for i in range(300)]
    wide = Op("sin", freq=200.0, amp=0.5, time_ms=300, mods={POP_PMOD: mods})
    for voices in ([op], [wide]):
        prg = vb.build_program(voices)
        want = oracle.oracle_render(prg.ptr, RATE, False)
        for env in ({}, {"SAU_AMD_NO_FAST": "1"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                got = sa.Batch([prg], RATE).render(stereo=False, chunk=5000)[0]
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            assert len(got) == len(want) and (got == want).all(), env


@pytest.mark.gpu
def test_two_generators_alternately_on_device(sa, oracle):
    """Two generators of one program at two rates, called in turn from one thread (the host's
    `split_gen` arrangement, saugns.c:585), each rendering ahead on its own stream and pooled
    buffers: neither disturbs the other."""
    from conftest import load_program
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = load_program(sa, "examples__dull_seq-fm_pm")
    rates = (44100, 48000)
    want = [oracle.oracle_render(prg.ptr, r, True) for r in rates]
    gens = [sa.Generator(prg, r) for r in rates]
    outs = [[], []]
    bufs = [np.zeros(2 * 11289, np.int16), np.zeros(2 * 12288, np.int16)]
    alive = [True, True]
    while any(alive):
        for i, g in enumerate(gens):
            if alive[i]:
                more, n = g.run(bufs[i], len(bufs[i]) // 2, True)
                outs[i].append(bufs[i][: 2 * n].copy())
                alive[i] = more
    for g in gens:
        g.close()
    for i in range(2):
        got = np.concatenate(outs[i])
        assert len(got) == len(want[i]) and (got == want[i]).all()


def _through_zero_bank(n, seconds):
    """Slow carriers under slow, deep PM: the phase step passes through zero twice per modulator
    cycle and changes by only a few units per sample there, so exactly repeated phases
    (wosc.h:251-252: the output holds) happen every few seconds per voice."""
    from saugns_amd.voicebank import Op, Line
    voices = []
    for i in range(n):
        m = Op("sin", freq=Line(0.5, ratio=True), amp=vb._f32(0.7 + 0.3 * ((i * 7) % 16) / 16.0))
        voices.append(Op("sin", freq=vb._num(".4f", 0.8 + i * 0.0131), amp=1.0, time_ms=seconds * 1000,
                         mods={POP_PMOD: [m]}))
    return vb.build_program(voices)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", ["8", "6", "5", "4"])
def test_repeated_phases_at_row_starts_stay_on_the_fast_path(sa, oracle, rows, monkeypatch):
    """Hundreds of exactly repeated phases, some of them on the first lane an operator is defined
    in: bit-exact, and made good by repair_kernel (the row group once more, shifted) -- no voice
    is handed to the block loop (which would take milliseconds per voice here). At every number of
    rows per pass that has a build of its own."""
    monkeypatch.setenv("SAU_AMD_FAST_ROWS", rows)
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = _through_zero_bank(256, 20)
    want = oracle.oracle_render(prg.ptr, RATE, False)
    b = sa.Batch([prg], RATE)
    b.set_timing(2)
    got = b.render(stereo=False, chunk=441000)[0]
    t = b.timing_ex()
    assert len(got) == len(want) and (got == want).all()
    assert t["block_ms"] < 1.0, t
    if rows != "8":
        return
    # the case is real: without the repair pass some voices do go to the block loop (still bit-exact)
    os.environ["SAU_AMD_NO_REPAIR"] = "1"
    try:
        b = sa.Batch([prg], RATE)
        b.set_timing(2)
        got = b.render(stereo=False, chunk=441000)[0]
        t = b.timing_ex()
    finally:
        del os.environ["SAU_AMD_NO_REPAIR"]
    assert (got == want).all()
    assert t["block_ms"] > 1.0, t


@pytest.mark.parametrize("wave", WAVES)
def test_wave_types_modulated_and_fed_back(sa, oracle, wave):
    """SURVEY 8c fixture list: each wave type x {plain, PM, self-modulation} (plain: test_wave_types)."""
    m = vb.Op(wave, freq=vb.Line(2.0, ratio=True), amp=0.6)
    check(sa, oracle, [vb.Op(wave, freq=277.2, time_ms=60, phase=0.2, mods={POP_PMOD: [m]})])
    check(sa, oracle, [vb.Op(wave, freq=277.2, time_ms=60, pm_a=0.4)])
    inner = vb.Op(wave, freq=vb.Line(3.0, ratio=True), amp=0.5, pm_a=0.3)
    check(sa, oracle, [vb.Op(wave, freq=138.6, time_ms=60, mods={POP_PMOD: [inner]})])


def test_noise_types(sa, oracle):
    """The seven noise generators (noise.h:41-185), as carriers and as phase modulators."""
    from saugns_amd.api import POPT_NOISE
    for nz in range(7):
        for seed in (0, 1, 0x9e3779b9):
            check(sa, oracle, [vb.Op(amp=0.7, time_ms=40, op_type=POPT_NOISE, noise=nz, seed=seed)], chunk=700)
        n = vb.Op(amp=0.2, op_type=POPT_NOISE, noise=nz, seed=7)
        check(sa, oracle, [vb.Op("sin", freq=300.0, time_ms=40, mods={POP_PMOD: [n]})])


def test_red_noise_stays_on_the_time_parallel_path(sa, oracle):
    """Red noise (noise.h:136-147) is a wrapping running sum of a counter hash: prefixes by look-back, like a
    running-sum phase -- as carrier, as PM source of an FM'd carrier (two sums in one voice), with an amplitude
    ramp, over several engine runs (the sum carried from one to the next), 1 to 64 waves per voice; and none of
    it in the block loop."""
    from saugns_amd.api import POPT_NOISE
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for n_voices, ms in ((1, 700), (3, 300), (40, 120), (300, 60)):
        voices = []
        for k in range(n_voices):
            red = vb.Op(amp=0.3 + 0.01 * (k % 5), op_type=POPT_NOISE, noise=4, seed=1000 + k)
            kind = k % 3
            if kind == 0:
                voices.append(vb.Op(amp=vb.Line(0.8, goal=0.1, shape="lin"), time_ms=ms, op_type=POPT_NOISE, noise=4, seed=k))
            elif kind == 1:
                voices.append(vb.Op("sin", freq=200.0 + k, time_ms=ms, mods={POP_PMOD: [red]}))
            else:
                voices.append(vb.Op("tri", freq=vb.Line(150.0 + k, goal=300.0, shape="exp"), time_ms=ms,
                                    mods={POP_PMOD: [red], POP_FMOD: [vb.Op("sin", freq=5.0, amp=10.0)]}))
        prg = vb.build_program(voices)
        want = oracle.oracle_render(prg.ptr, RATE, False)
        for chunk in (4000000, 1733):
            b = sa.Batch([prg], RATE)
            b.set_timing(2)
            got = b.render(stereo=False, chunk=chunk)[0]
            assert len(got) == len(want) and (got == want).all(), (n_voices, chunk)
            if chunk > len(want):
                assert b.timing_ex()["block_ms"] < 0.5, (n_voices, b.timing_ex())  # (a token launch only)


@pytest.mark.parametrize("line", LINES)
def test_r_oscillator_options(sa, oracle, line):
    """R oscillator: every line shape x segment function x a spread of function flags
    (rasg.h:299-743), plain, as a PM source with PM of its own, and self-modulated
    (rasg.h:242-294). tests/test_oracle.py pins the same grid against the compiled reference."""
    from saugns_amd.api import POPT_RASEG
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for func in range(6):
        voices = []
        for k, flags in enumerate((0, 1, 2, 4, 8, 16, 9, 25, 31)):
            voices.append(vb.Op(freq=180.0 + 31.0 * k, amp=0.8, time_ms=25 + k, op_type=POPT_RASEG,
                                seed=12345 + k, ras=(line, func, flags)))
        check(sa, oracle, voices, stereo=True, chunk=900)
        m = vb.Op("sin", freq=vb.Line(0.5, ratio=True), amp=0.4)
        r = vb.Op(freq=vb.Line(2.0, ratio=True), amp=0.6, op_type=POPT_RASEG, seed=99, ras=(line, func, 9),
                  mods={POP_PMOD: [m]})
        check(sa, oracle, [vb.Op("sin", freq=220.0, time_ms=40, mods={POP_PMOD: [r]})])
        fed_back = [vb.Op(freq=150.0 + 17.0 * k, amp=0.7, time_ms=40, op_type=POPT_RASEG, seed=5 + flags,
                          ras=(line, func, flags), pm_a=(0.5, 1.5)[k & 1])
                    for k, flags in enumerate((0, 1, 2, 4, 8, 16, 9, 25, 31))]
        check(sa, oracle, fed_back, chunk=1100)


def amp_operator_cases():
    """The A (amplitude) operator, generator.c:505-520 run_block_amp / README.SAU:61-62: a source
    of 1.0 scaled by its amplitude parameter -- every ramp shape, amplitude and range-amplitude
    modulators of its own, a duration shorter than its parent's -- in every role: carrier (a DC
    level), amplitude / range-amplitude / frequency / range-frequency / phase / frequency-scaled
    phase / self-modulation-amount / pan modulator, alone and layered beside other modulators.
    -> list of (name, voices, later events)."""
    from saugns_amd.api import POPT_AMP
    A = lambda amp, **kw: vb.Op(amp=amp, op_type=POPT_AMP, **kw)
    cases = []
    for shape in LINES:
        cases.append((f"carrier ramp {shape}", [A(vb.Line(0.1, goal=0.9, shape=shape), time_ms=30)], ()))
        cases.append((f"amod ramp {shape}", [vb.Op("sin", freq=300.0, amp=0.2, time_ms=30,
                                                  mods={POP_AMOD: [A(vb.Line(0.0, goal=0.7, shape=shape))]})], ()))
    lfo = lambda f=6.0, a=0.5: vb.Op("sin", freq=f, amp=a)
    cases += [
        ("carrier held", [A(0.6, time_ms=25), A(-0.3, time_ms=40, pan=0.5)], ()),
        ("carrier with amods and ramods", [A(0.4, amp2=0.9, time_ms=60, mods={POP_AMOD: [lfo(9.0, 0.2)], POP_RAMOD: [lfo(4.0, 1.0)]})], ()),
        ("ramod source", [vb.Op("tri", freq=220.0, amp=0.2, amp2=0.9, time_ms=50,
                                mods={POP_RAMOD: [A(vb.Line(0.0, goal=1.0, shape="cos"))]})], ()),
        ("ramod source layered", [vb.Op("tri", freq=220.0, amp=0.2, amp2=0.9, time_ms=50,
                                        mods={POP_RAMOD: [lfo(5.0, 1.0), A(0.5)]})], ()),
        ("fmod source", [vb.Op("sin", freq=200.0, time_ms=50, mods={POP_FMOD: [A(vb.Line(0.0, goal=150.0, shape="lin"))]})], ()),
        ("rfmod source", [vb.Op("sin", freq=200.0, freq2=400.0, time_ms=50,
                                mods={POP_RFMOD: [A(vb.Line(0.0, goal=1.0, shape="sqe"))]})], ()),
        ("pmod source", [vb.Op("sin", freq=200.0, time_ms=50, mods={POP_PMOD: [A(vb.Line(0.0, goal=2.0, shape="xpe")), lfo(3.0, 0.5)]})], ()),
        ("fpmod source", [vb.Op("sin", freq=200.0, time_ms=50, mods={POP_FPMOD: [A(vb.Line(0.0, goal=30.0, shape="lin"))]})], ()),
        ("apmod source", [vb.Op("sin", freq=200.0, time_ms=50, pm_a=0.3, mods={POP_APMOD: [A(vb.Line(0.0, goal=0.6, shape="lin"))]})], ()),
        ("camod source", [vb.Op("sin", freq=200.0, time_ms=50, pan=-0.4, mods={POP_CAMOD: [A(vb.Line(0.0, goal=0.8, shape="lin"))]})], ()),
        ("shorter than its parent", [vb.Op("sin", freq=250.0, amp=0.1, time_ms=60,
                                           mods={POP_AMOD: [A(0.5, time_ms=20), lfo(7.0, 0.2)], POP_PMOD: [A(0.25, time_ms=35)]})], ()),
        ("nested", [vb.Op("sin", freq=250.0, time_ms=60, mods={POP_PMOD: [
            vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.1, mods={POP_AMOD: [A(0.7, mods={POP_AMOD: [lfo(11.0, 0.3)]})]})]})], ()),
        ("in an R oscillator", [vb.Op(freq=180.0, amp=0.3, time_ms=50, op_type=POPT_RASEG, seed=77, ras=("cos", 1, 0),
                                      mods={POP_AMOD: [A(vb.Line(0.0, goal=0.5, shape="lin"))], POP_FMOD: [A(40.0)]})], ()),
    ]
    # later events on an A operator: new value, new ramp, new duration
    a1 = A(0.5)
    c1 = vb.Op("sin", freq=300.0, amp=0.2, time_ms=80, mods={POP_AMOD: [a1]})
    cases.append(("events on a modulator", [c1], [(20, 0, a1, {"amp": vb.Line(0.1)}),
                                                  (35, 0, a1, {"amp": vb.Line(0.0, goal=0.9, shape="cos", state=False)}),
                                                  (60, 0, a1, {"amp": vb.Line(0.3, goal=0.0, shape="lin"), "time_ms": 10})]))
    c2 = A(0.3, time_ms=40)
    cases.append(("events on a carrier", [c2], [(15, 0, c2, {"amp": vb.Line(0.0, goal=0.8, shape="lin", state=False)}),
                                                 (30, 0, c2, {"amp": vb.Line(0.5), "time_ms": 30})]))
    return cases


def test_amp_operator(sa, oracle):
    """A operator in every role (amp_operator_cases), bit-exact vs the oracle; tests/test_oracle.py
    pins the oracle on the same cases against the compiled reference."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for name, voices, ups in amp_operator_cases():
        prg = vb.build_program(voices, updates=ups)
        for stereo, chunk in ((True, 4000000), (False, 777)):
            want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
            for path in ("drop-in", "batch"):
                got = (sa.Generator(prg, RATE).render(stereo=stereo, chunk=chunk) if path == "drop-in"
                       else sa.Batch([prg], RATE).render(stereo=stereo, chunk=chunk)[0])
                assert len(got) == len(want) and (got == want).all(), (name, stereo, chunk, path)
        assert np.abs(want.astype(np.int32)).max() > 0, name


def test_pan_modulators(sa, oracle):
    """Pan modulator lists (generator.c:749-788): lasting as long as the carrier, shorter (the voice
    goes on with the pan line alone), longer, nested, beside a pan ramp; stereo and mono."""
    for mod_ms in (None, 30, 70, 200):
        m = vb.Op("spa", freq=vb.Line(1.0, ratio=True), amp=0.48, time_ms=mod_ms)
        inner = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=0.5, time_ms=50)
        m2 = vb.Op("sin", freq=7.0, amp=0.3, mods={POP_PMOD: [inner]})
        voices = [vb.Op("tri", freq=144.3, amp=0.7, time_ms=136, pan=0.25, mods={POP_CAMOD: [m]}),
                  vb.Op("sin", freq=300.0, amp=0.5, time_ms=100, pan=vb.Line(-0.5, goal=0.5, shape="cos"),
                        mods={POP_CAMOD: [m2, vb.Op("saw", freq=2.0, amp=0.2, time_ms=mod_ms)]})]
        for stereo in (True, False):
            check(sa, oracle, voices, stereo=stereo, chunk=4000000)
            check(sa, oracle, voices, stereo=stereo, chunk=777)


@pytest.mark.parametrize("stereo", [False, True])
def test_batch_of_different_programs(sa, oracle, stereo):
    """sauAmd_create_Batch over 24 random programs with their own event timelines, lengths and
    voice counts: every stream equals its own single render (streams share launches, nothing else)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prgs, wants = [], []
    for seed in range(24):
        rng = np.random.default_rng(9000 + seed)
        voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        _random_starts(rng, voices)
        prg = vb.build_program(voices, updates=_random_updates(rng, voices))
        prgs.append(prg)
    for chunk in (4000000, 1777):
        outs = sa.Batch(prgs, RATE).render(stereo=stereo, chunk=chunk)
        for k, (got, prg) in enumerate(zip(outs, prgs)):
            want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)  # (the same call size: loop tails)
            assert len(got) >= len(want) and (got[:len(want)] == want).all() and not got[len(want):].any(), (k, chunk)


def test_ratio_chains_below_modulated_frequencies(sa, oracle):
    """Ratio-frequency modulators nested deeper than the modulators that write into an ancestor's
    frequency block (a classic FM patch: vibrato on the carrier, a ratio PM stack under it): the
    rows get extra lead-in lanes instead of the voice going to the block loop. With frequency-scaled
    PM (one lane more), range-FM, and an FM'd modulator inside the stack."""
    def stack(fpm=False):
        m3 = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=0.4)
        m2 = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.7, mods={(POP_FPMOD if fpm else POP_PMOD): [m3]})
        return vb.Op("sin", freq=vb.Line(1.0, ratio=True), amp=0.9, mods={POP_PMOD: [m2]})
    vib = lambda f, a: vb.Op("sin", freq=f, amp=a)
    voices = [
        vb.Op("sin", freq=220.0, time_ms=400, mods={POP_PMOD: [stack()], POP_FMOD: [vib(5.0, 30.0)]}),
        vb.Op("sin", freq=330.0, time_ms=350, mods={POP_PMOD: [stack(fpm=True)], POP_FMOD: [vib(7.0, 12.0)]}),
        vb.Op("saw", freq=110.0, freq2=220.0, time_ms=300,
              mods={POP_PMOD: [stack()], POP_RFMOD: [vb.Op("sin", freq=vb.Line(0.25, ratio=True), amp=1.0)]}),
    ]
    inner_fm = vb.Op("sin", freq=vb.Line(0.5, ratio=True), amp=0.8,
                     mods={POP_FMOD: [vib(3.0, 20.0)], POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.5,
                           mods={POP_PMOD: [vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=0.3)]})]})
    voices.append(vb.Op("sin", freq=180.0, time_ms=300, mods={POP_PMOD: [inner_fm], POP_FMOD: [vib(4.0, 25.0)]}))
    for chunk in (4000000, 5000, 997):
        check(sa, oracle, voices, stereo=True, chunk=chunk)
    b = sa.Batch([vb.build_program(voices)], RATE)
    b.set_timing(2)
    b.render(stereo=False, chunk=4000000)
    assert b.timing_ex()["block_ms"] < 1.0  # none of them needed the block loop


@pytest.mark.parametrize("env", [{"SAU_AMD_FAST_ROWS": "2"}, {"SAU_AMD_FAST_ROWS": "6"}, {"SAU_AMD_FAST_ROWS": "5"}, {"SAU_AMD_NO_TWO_PASS": "1"}, {"SAU_AMD_NO_SEQ": "1"},
                                 {"SAU_AMD_LDS_LIMIT": "65536"}, {"SAU_AMD_MULTI_MIN": "1"},
                                 {"SAU_AMD_NO_LOOKBACK": "1"}, {"SAU_AMD_NO_LOOKBACK": "1", "SAU_AMD_NO_INC_ROWS": "1"},
                                 {"SAU_AMD_LOOK_MIN_VOICES": "1"}, {"SAU_AMD_LOOK_MIN_VOICES": "1", "SAU_AMD_LOOK_ROWS": "4"},
                                 {"SAU_AMD_LOOK_NO_LDS": "1"}, {"SAU_AMD_NO_DYN": "1"}, {"SAU_AMD_DYN_GROUPS": "1"},
                                 {"SAU_AMD_NO_LEAN": "1"},
                                 # round 4's forms of the closed-form build: narrow tables, 8 and 10 rows per pass
                                 {"SAU_AMD_NO_WIDE_TABS": "1"}, {"SAU_AMD_MORE_ROWS": "0"}, {"SAU_AMD_MORE_ROWS": "10"}])
def test_random_graphs_in_other_kernel_configurations(sa, oracle, env):
    """The random programs with events through the other builds and modes of the kernels: two rows per
    pass, running sums by one wave in order, no running sums in the time-parallel path at all, a
    tight LDS budget, single-wave teams in the block loop, running sums in several passes instead of
    one pass with look-back (with and without the saved increments), and -- few voices as these programs
    have -- one pass with look-back where the default keeps several passes (8 and 4 rows per pass)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for seed in range(100, 124):
            rng = np.random.default_rng(5000 + seed)
            voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
            _random_starts(rng, voices)
            prg = vb.build_program(voices, updates=_random_updates(rng, voices))
            stereo = bool(seed & 1)
            chunk = int(rng.integers(700, 3000))
            want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=chunk)
            got = sa.Batch([prg], RATE).render(stereo=stereo, chunk=chunk)[0]
            assert len(got) == len(want) and (got == want).all(), (seed, env)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("depth", ["2", "1"])
def test_dropin_generator_on_random_programs(sa, oracle, depth, monkeypatch):
    """sau_create_Generator / sauGenerator_run with the host's 11289-frame calls (and odd sizes) on
    randomized programs with events; long ones, so that the read-ahead hands out several device runs
    from its three buffers (two runs in flight, or one: SAU_AMD_READAHEAD_DEPTH). The device renders runs of 176400
    frames ahead, each a whole number of the host's calls; the oracle makes the host's calls."""
    monkeypatch.setenv("SAU_AMD_READAHEAD_DEPTH", depth)
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for seed in range(300, 308):
        rng = np.random.default_rng(5000 + seed)
        voices = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        for carr in voices:  # 4 .. 16 s: more than one 176400-frame run
            carr.time_ms = int(carr.time_ms * 100)
        _random_starts(rng, voices)
        prg = vb.build_program(voices, updates=_random_updates(rng, voices))
        stereo = bool(seed & 1)
        for call in (11289, 4099):
            # (the oracle makes the host's calls; the device renders runs of many of them ahead)
            want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=call)
            got = sa.Generator(prg, RATE).render(stereo=stereo, chunk=call)
            assert len(got) == len(want) and (got == want).all(), (seed, call)


def test_deep_nesting(sa, oracle):
    """Nesting as deep as a sauProgram can state (op_nest_depth is a uint8: 256 operators on a path,
    sau/program.h:259). Straight PM chains of 33, 65 and 120 levels (8-bit buffer ids; one wave, one frame per
    lane, about 240 block buffers in LDS), of 130 (a wide plan -- step pairs with 16-bit ids -- still in LDS) and
    of 255 and 256 levels (512 buffers in HBM: render_kernel<1, 1, 1, true>); chains through the other
    modulator lists at 256 levels (range modulators hold three buffers per level); a wide voice beside ordinary
    voices in one segment, several calls; bit-exact vs the oracle."""
    def chain(depth, use=POP_PMOD, ms=40):
        op = None
        for d in range(depth):
            top = d == depth - 1
            if use in (POP_FMOD, POP_RFMOD):
                f = 200.0 if top else 30.0 + 11.0 * (d % 7)
            else:
                f = 200.0 if top else vb.Line(1.0 + (d % 4) * 0.5, ratio=True)
            op = vb.Op("sin", freq=f, amp=20.0 if use in (POP_FMOD, POP_RFMOD) else 0.5,
                       time_ms=ms if top else None, mods={use: [op]} if op else {})
        return op
    for depth in (33, 65, 120, 130, 255, 256):
        check(sa, oracle, [chain(depth)], chunk=1500)
    for use in (POP_FMOD, POP_RFMOD, POP_AMOD, POP_RAMOD, POP_FPMOD):
        check(sa, oracle, [chain(256, use)], chunk=1700)
    check(sa, oracle, [chain(150), vb.Op("saw", freq=110.0, amp=0.3, time_ms=80), chain(256, POP_AMOD, ms=55),
                       vb.Op("sin", freq=330.0, time_ms=70, mods={POP_PMOD: [vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.4)]})],
          stereo=True, chunk=997)


@pytest.mark.parametrize("seed", [266, 270, 273, 274] + list(range(600, 620)))
def test_tall_random_trees(sa, oracle, seed):
    """Operator trees of 2..256 levels with the list kind, operator type and parameters drawn per level (program `seed` of
    tests/tools/gpu_vs_ref_sweep.py's `tall` mode): wide plans with range modulators, R and N operators, feedback -- the first
    four are the ones that found render_kernel<1, 1, 1, true> reading its block buffers in HBM through LDS-typed pointers in
    the W feedback loop (a memory fault at worst). Bit-exact vs the oracle."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    rng = np.random.default_rng(20000 + seed)
    _ = [_random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    voices = [_tall_tree(rng) for _ in range(int(rng.integers(1, 3)))]
    ups = ()
    if seed % 2:
        _random_starts(rng, voices)
        ups = _random_updates(rng, voices)
    prg = vb.build_program(voices, updates=ups)
    call = int(rng.integers(300, 12000))
    stereo = bool(seed & 2)
    want = oracle.oracle_render(prg.ptr, RATE, stereo, chunk=call)
    got = sa.Batch([prg], RATE).render(stereo=stereo, chunk=call)[0]
    assert len(got) == len(want) and (got == want).all(), (seed, int((got != want).sum()))


def test_voices_with_thousands_of_operators(sa, oracle):
    """What bounds a voice's tree is memory, not LDS: when block buffers, operator records (256 B each) and steps of a voice do
    not fit a workgroup's LDS they all go to the workgroup's area in HBM (render_kernel<1, 1, 1, true>). A carrier with 40 PM
    modulators of 40 AM modulators each (1641 operators), one with 700 modulators in one list and feedback, and both beside
    ordinary voices; bit-exact vs the oracle."""
    big = vb.Op("sin", freq=200.0, amp=0.5, time_ms=30, mods={POP_PMOD: [
        vb.Op(("sin", "tri", "saw")[i % 3], freq=vb.Line(1.0 + (i % 5) * 0.5, ratio=True), amp=0.02 + 0.001 * i, mods={POP_AMOD: [
            vb.Op("sin", freq=3.0 + j + i, amp=0.01 + 0.002 * j) for j in range(40)]}) for i in range(40)]})
    flat = vb.Op("tri", freq=150.0, amp=0.6, pm_a=0.3, time_ms=25, mods={POP_PMOD: [
        vb.Op("sin", freq=vb.Line(float(1 + i % 7), ratio=True), amp=0.004, phase=(i % 10) / 10) for i in range(700)]})
    small = vb.Op("saw", freq=110.0, amp=0.3, time_ms=40)
    check(sa, oracle, [big], chunk=700)
    check(sa, oracle, [flat], chunk=900)
    check(sa, oracle, [small, big, vb.Op("sin", freq=330.0, time_ms=35), flat], stereo=True, chunk=555)


@pytest.mark.parametrize("chunks", ["1", "2", "16"])
def test_feedback_chains_at_other_pipeline_depths(sa, oracle, chunks, monkeypatch):
    """chain_kernel beside the time-parallel passes: segments with feedback voices cut into
    1, 2 or 16 chunks instead of the default (by length), chains fed from their own lines by the feeder waves (the default
    since round 4) and, in one leg, through rows written by a chain-input pass instead -- bit-exact vs the oracle, frequency /
    amount / amplitude ramps, constant-frequency chains and a feedback modulator included."""
    monkeypatch.setenv("SAU_AMD_CHAIN_CHUNKS", chunks)
    if chunks == "2":
        monkeypatch.setenv("SAU_AMD_NO_CHAIN_INLINE", "1")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = vb.config5(n=96, seconds=2)
    want = oracle.oracle_render(prg.ptr, RATE, False)
    got = sa.Batch([prg], RATE).render(stereo=False, chunk=88200)[0]
    assert len(got) == len(want) and (got == want).all()
    inner = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.4, pm_a=vb.Line(0.2, goal=0.6, shape="cos"))
    voices = [vb.Op("sin", freq=vb.Line(150.0 + 7 * k, goal=300.0, shape="exp"), time_ms=900 + 10 * k, pm_a=0.3 + 0.05 * k,
                    mods={POP_PMOD: [inner]} if k == 3 else {}) for k in range(6)]
    check(sa, oracle, voices, chunk=50000)
    # chains of one frequency (no accumulator staged between chunks: the closed form), with amount ramps and without
    flat = [vb.Op(("sin", "tri", "saw")[k % 3], freq=97.0 + 31 * k, time_ms=1500 + 40 * k,
                  pm_a=(vb.Line(0.1 + 0.1 * k, goal=0.9, shape=("lin", "xpe", "cos")[k % 3]) if k & 1 else 0.4 + 0.1 * k)) for k in range(7)]
    check(sa, oracle, flat, chunk=200000)


@pytest.mark.parametrize("early", ["on", "off"])
def test_feedback_chains_that_running_sums_depend_on(sa, oracle, early, monkeypatch):
    """A feedback chain whose inputs are its own lines runs before every pass when something the passes compute
    depends on it (FastInfo.early): the shape of examples/sounds/kaboom1.sau -- a self-modulating W
    oscillator range-modulating the frequency of an R oscillator that range-modulates the carrier's frequency --,
    a chain FM-ing a W carrier, a chain whose phase is modulated by such a chain (the outer one runs in chunks as
    usual and reads the inner one's row), frequency and amount ramps on the early chain, an early and an ordinary
    chain in one voice. Bit-exact vs the oracle, also with SAU_AMD_NO_EARLY_CHAINS=1 (such voices then take the block loop:
    kaboom1 71 against 32 ms, tools/gpu_script_phases.py)."""
    if early == "off":
        monkeypatch.setenv("SAU_AMD_NO_EARLY_CHAINS", "1")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    def fb(f=0.2, a=0.5, amp=1.0, **kw):
        return vb.Op("sin", freq=f, pm_a=a, amp=amp, **kw)
    kaboom = vb.Op("sin", freq=-500.0, freq2=500.0, amp=vb.Line(0.0, goal=1.0, shape="exp"), time_ms=1500,
                   mods={POP_RFMOD: [vb.Op(op_type=POPT_RASEG, ras=("smo", 0, 0), seed=7, freq=1.0, freq2=10.0,
                                           mods={POP_RFMOD: [fb(phase=0.25)]})]})
    fm = vb.Op("tri", freq=220.0, time_ms=900, mods={POP_FMOD: [fb(3.0, vb.Line(0.2, goal=0.7, shape="lin"), amp=40.0)]})
    nested = vb.Op("sin", freq=150.0, pm_a=0.4, time_ms=800,
                   mods={POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, goal=5.0, shape="cos"), pm_a=0.6, amp=0.8)]})
    both = vb.Op("saw", freq=110.0, time_ms=700,
                 mods={POP_FMOD: [fb(2.5, 0.3, amp=25.0)],
                       POP_PMOD: [vb.Op("sin", freq=vb.Line(2.0, ratio=True), pm_a=0.5, amp=0.7,
                                        mods={POP_PMOD: [vb.Op("tri", freq=vb.Line(3.0, ratio=True), amp=0.3)]})]})
    # R-oscillator feedback (rasg.h:242-294), fed from its own lines: rchain_kernel, always early -- as a carrier with
    # frequency and amount ramps, as a PM source, as the modulator of a running sum, beside a W chain
    rfb = vb.Op(op_type=POPT_RASEG, ras=("cos", 1, 9), seed=11, freq=vb.Line(90.0, goal=240.0, shape="exp"),
                pm_a=vb.Line(0.3, goal=1.2, shape="lin"), amp=0.6, time_ms=600)
    rpm = vb.Op("sin", freq=200.0, time_ms=500, pm_a=0.3,
                mods={POP_PMOD: [vb.Op(op_type=POPT_RASEG, ras=("lin", 3, 2), seed=12, freq=vb.Line(1.5, ratio=True), pm_a=0.8, amp=0.5)],
                      POP_FMOD: [vb.Op(op_type=POPT_RASEG, ras=("smo", 0, 16), seed=13, freq=7.0, pm_a=0.5, amp=30.0)]})
    voices = [kaboom, fm, nested, both, rfb, rpm]
    for vs in ([kaboom], [fm], [nested], [both], [rfb], [rpm], voices):
        prg = vb.build_program(vs)
        want = oracle.oracle_render(prg.ptr, RATE, True)
        for chunk in (1000000, 30000):
            got = sa.Batch([prg], RATE).render(stereo=True, chunk=chunk)[0]
            d = np.nonzero(got != want)[0]
            assert len(got) == len(want) and len(d) == 0, (len(vs), chunk, len(d), d[:4])


def test_r_feedback_bank_of_many_kinds(sa, oracle):
    """R oscillators with self-modulation (rasg.h:242-294: rchain_kernel, lanes = voices) in a bank whose neighbours differ in line
    shape, function and flags -- 24 kinds over 200 voices, interleaved, a few W feedback voices among them: the host numbers the
    chains' rows kind by kind (engine.cpp, round 6) so that a wave's 64 chains agree and take the scalarised copy of the loop;
    the PCM is the oracle's whatever the numbering."""
    from saugns_amd.api import POPT_RASEG
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    voices = []
    for k in range(200):
        if k % 25 == 7:
            voices.append(vb.Op("sin", freq=110.0 + k, pm_a=0.4 + 0.002 * k, amp=0.5, time_ms=400))
        else:
            voices.append(vb.Op(op_type=POPT_RASEG, ras=(("lin", "cos", "sqe", "xpe")[k % 4], (k // 4) % 6, (0, 5, 17, 31)[(k // 24) % 4]),
                                seed=900 + 13 * k, freq=vb.Line(90.0 + 7 * (k % 40), goal=240.0 + k % 60, shape="exp") if k % 3 else 150.0 + k % 80,
                                pm_a=vb.Line(0.1 + 0.01 * (k % 30), goal=0.8, shape="lin") if k % 2 else 0.45, amp=0.5, time_ms=400))
    check(sa, oracle, voices, chunk=1000000)
    check(sa, oracle, voices, chunk=7000, stereo=True)


@pytest.mark.parametrize("lookback", ["on", "off"])
def test_running_sums_by_look_back(sa, oracle, lookback, monkeypatch):
    """Running-sum phases in one pass (every wave publishes its row group's sum and looks back
    for its prefix) against the several-pass form of the same voices: FM nested deeper than the sum
    passes reach, nine running-sum oscillators in one voice (one more than the look-back arrays hold:
    that voice goes to one wave in order), R oscillators with swept and modulated frequencies (64-bit
    counters), a single long voice whose waves sit in different workgroups, and many short voices."""
    from saugns_amd.api import POPT_RASEG
    if lookback == "off":
        monkeypatch.setenv("SAU_AMD_NO_LOOKBACK", "1")
    else:
        monkeypatch.setenv("SAU_AMD_LOOK_MIN_VOICES", "1")  # (by default segments with few voices keep the several-pass form)
    def nest(depth, f0=3.0):
        m = vb.Op("sin", freq=f0, amp=9.0)
        for d in range(depth - 1):
            m = vb.Op(("tri", "sin", "saw")[d % 3], freq=f0 * (d + 2) * 1.7, amp=14.0 + 5 * d, mods={POP_FMOD: [m]})
        return m
    check(sa, oracle, [vb.Op("sin", freq=330.0, time_ms=700, mods={POP_FMOD: [nest(5)]})], chunk=1000000)
    wide = lambda n: [vb.Op("sin", freq=vb.Line(2.0 + k, goal=5.0 + 2 * k, shape=("lin", "exp", "cos")[k % 3]), amp=6.0 + k)
                      for k in range(n)]
    check(sa, oracle, [vb.Op("sin", freq=vb.Line(200.0, goal=260.0, shape="lin"), time_ms=500, mods={POP_FMOD: wide(7)}),
                       vb.Op("sin", freq=vb.Line(300.0, goal=150.0, shape="exp"), time_ms=400, mods={POP_FMOD: wide(8)})],
          stereo=True, chunk=30000)
    r_voices = [vb.Op(freq=vb.Line(120.0 + 40 * k, goal=480.0 - 30 * k, shape="exp"), amp=0.7, time_ms=350 + 20 * k,
                      op_type=POPT_RASEG, seed=77 + k, ras=(("lin", "cos", "sah")[k % 3], k % 6, (0, 9, 25, 31)[k % 4]),
                      mods={POP_FMOD: [vb.Op("sin", freq=6.0, amp=25.0)]} if k & 1 else {}) for k in range(6)]
    check(sa, oracle, r_voices, chunk=100000)
    r_mod = vb.Op(freq=vb.Line(3.0, goal=11.0, shape="lin"), amp=30.0, op_type=POPT_RASEG, seed=5, ras=("cos", 1, 9))
    check(sa, oracle, [vb.Op("sin", freq=440.0, time_ms=600, mods={POP_FMOD: [r_mod]})], chunk=1000000)
    long_one = vb.Op("sin", freq=vb.Line(55.0, goal=1760.0, shape="exp"), time_ms=6000,
                     mods={POP_FMOD: [vb.Op("sin", freq=4.0, amp=8.0, mods={POP_FMOD: [vb.Op("sin", freq=0.7, amp=2.0)]})]})
    check(sa, oracle, [long_one], chunk=1000000)
    many = [vb.Op("sin", freq=vb.Line(100.0 + 3 * k, goal=200.0 + k, shape="lin"), time_ms=40 + k % 17,
                  mods={POP_FMOD: [vb.Op("sin", freq=5.0 + k % 5, amp=10.0)]}) for k in range(300)]
    check(sa, oracle, many, chunk=1000000)
    # 300 voices long enough for 13 waves each: most of them sit across two workgroups (words in HBM, not rings in LDS)
    across = [vb.Op(("sin", "tri")[k & 1], freq=vb.Line(90.0 + 2 * k, goal=400.0 - k, shape=("exp", "lin")[k % 2]), time_ms=190 + k % 23,
                    mods={POP_FMOD: [vb.Op("sin", freq=4.0 + k % 7, amp=12.0)]} if k % 3 else {}) for k in range(300)]
    check(sa, oracle, across, chunk=1000000)
    # banks too big for more than one wave per voice (2500 voices: a wave each, sums carried in LDS; 4200: waves
    # take several voices in turn)
    for n in (2500, 4200):
        bank = [vb.Op("sin", freq=vb.Line(80.0 + 0.1 * k, goal=160.0 + 0.05 * k, shape="lin"), time_ms=30 + k % 11,
                      mods={POP_FMOD: [vb.Op("sin", freq=6.0 + k % 3, amp=9.0)]} if k % 2 else {}) for k in range(n)]
        check(sa, oracle, bank, chunk=1000000)
    b = sa.Batch([vb.build_program([vb.Op("sin", freq=330.0, time_ms=700, mods={POP_FMOD: [nest(5)]}), long_one])], RATE)
    b.set_timing(2)
    b.render(stereo=False, chunk=4000000)
    assert b.timing_ex()["block_ms"] < 1.0  # the time-parallel kernel took them, not the block loop


@pytest.mark.timeout(120)
def test_look_back_wait_is_bounded(sa, oracle, monkeypatch):
    """a wave that waits for another workgroup's sums (look-back words in HBM) must not spin forever when that
    workgroup never comes -- a second process on the device, a CU-masked one. SAU_AMD_LOOK_WITHHOLD makes group 1 of every
    spread voice keep its words to itself: the wave behind it gives up after LOOK_SPIN_MAX empty polls (k_common.h),
    publishes so that nobody waits for *it*, and the voice's segment is redone by the block loop -- slow, exact, never hung."""
    monkeypatch.setenv("SAU_AMD_LOOK_MIN_VOICES", "1")
    monkeypatch.setenv("SAU_AMD_LOOK_NO_LDS", "1")  # (every voice through the words in HBM)
    monkeypatch.setenv("SAU_AMD_LOOK_WITHHOLD", "1")
    long_one = vb.Op("sin", freq=vb.Line(55.0, goal=1760.0, shape="exp"), time_ms=3000,
                     mods={POP_FMOD: [vb.Op("sin", freq=4.0, amp=8.0, mods={POP_FMOD: [vb.Op("sin", freq=0.7, amp=2.0)]})]})
    from saugns_amd.api import POPT_RASEG
    r_one = vb.Op(freq=vb.Line(120.0, goal=480.0, shape="exp"), amp=0.7, time_ms=2500, op_type=POPT_RASEG, seed=77, ras=("cos", 1, 9))
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for voices in ([long_one], [r_one], [long_one, r_one]):
        prg = vb.build_program(voices)
        want = oracle.oracle_render(prg.ptr, RATE, False, chunk=1000000)
        b = sa.Batch([prg], RATE)
        b.set_timing(2)
        got = b.render(stereo=False, chunk=1000000)[0]
        assert len(got) == len(want) and (got == want).all()
        assert b.timing_ex()["block_ms"] > 1.0  # (the block loop did redo them: the hook really withheld the words)


def test_batches_ordered_one_behind_the_other(sa, oracle):
    """sauAmd_Batch_order_after: scripts one after the other with two generators alive -- the next script is created and
    its run issued while the device still renders the current one, its rendering kernels ordered behind the current
    script's (bench.py's config-5 steps). Feedback voices (chunks, chains on their second stream, the mixer following the
    chunks), FM voices (look-back across workgroups) and plain ones; every render equals the oracle's, whatever order the
    host then collects them in."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    fm = [vb.Op("sin", freq=vb.Line(120.0 + 9 * k, goal=260.0, shape="exp"), time_ms=1200,
                mods={POP_FMOD: [vb.Op("sin", freq=4.0 + k % 3, amp=12.0)]}) for k in range(24)]
    prgs = [vb.config5(n=200, seconds=2), vb.build_program(fm), vb.config3(n=64, seconds=2), vb.config5(n=70, seconds=3)]
    want = [oracle.oracle_render(p.ptr, RATE, False, chunk=150000) for p in prgs]
    order = [0, 1, 2, 3, 0, 3, 1]
    cur = sa.Batch([prgs[order[0]]], RATE)
    pend = [(cur, order[0], [])]
    outs = []
    for nxt_i in order[1:] + [None]:
        b, i, parts = pend[-1]
        nb = None
        if nxt_i is not None:
            nb = sa.Batch([prgs[nxt_i]], RATE)
            nb.order_after(b)
        # the current script to its end, fetched (each run's PCM block holds one run), the next one issued behind it
        while True:
            pcm, more, lens = b.run(150000, stereo=False)
            parts.append(pcm[0, :lens[0]].copy())
            if not more[0]:
                break
        outs.append((i, np.concatenate(parts)))
        b.close()
        if nb is not None:
            pend.append((nb, nxt_i, []))
    for i, got in outs:
        assert len(got) == len(want[i]) and (got == want[i]).all(), i
    # both in flight: the first issued and not fetched, the second ordered behind it and fetched, then the first's next run
    a, b = sa.Batch([prgs[0]], RATE), sa.Batch([prgs[3]], RATE)
    a.run(50000, stereo=False, fetch=False)
    b.order_after(a)
    pcm, more, lens = b.run(132300, stereo=False)
    assert lens[0] == 132300 and (pcm[0] == want[3][:132300]).all()
    pcm, more, lens = a.run(38200, stereo=False)
    assert lens[0] == 38200 and (pcm[0] == want[0][50000:88200]).all()
    a.close(); b.close()


@pytest.mark.timeout(180)
def test_two_generators_at_once_with_running_sums(sa, oracle):
    """Two host threads, a generator each, rendering banks of FM voices at the same time: their single-pass
    launches have voices spread over several workgroups (up to 64 waves per voice), which wait for each other
    across workgroups -- the process lets such launches take turns on a device (SpreadLaunchOrder), everything
    else of the two generators overlaps. Both renders equal the oracle's."""
    import threading
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    def bank(seed):
        return [vb.Op("sin", freq=vb.Line(100.0 + 7 * k + seed, goal=300.0 + 3 * k, shape="exp"), time_ms=1500 + 10 * k,
                      mods={POP_FMOD: [vb.Op("sin", freq=3.0 + k % 5, amp=15.0 + seed,
                                             mods={POP_FMOD: [vb.Op("tri", freq=0.5 + 0.1 * k, amp=1.5)]})]})
                for k in range(40)]
    prgs = [vb.build_program(bank(s)) for s in (0, 11)]
    want = [oracle.oracle_render(p.ptr, RATE, False) for p in prgs]
    got = [None, None]
    def work(i):
        out = []
        for rep in range(3):
            out.append(sa.Batch([prgs[i]], RATE).render(stereo=False, chunk=20000 + 7000 * i)[0])
        got[i] = out
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(2):
        for pcm in got[i]:
            assert len(pcm) == len(want[i]) and (pcm == want[i]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["1", "real"])
def test_feedback_voices_render_when_the_chain_rows_cannot_be_allocated(tmp_path, how):
    """the chains' rows are the one large allocation that depends on how the engine cut the segment. When the device
    cannot give them (SAU_AMD_CHAIN_ROWS_FAIL makes an engine's first such allocation fail) the segment's feedback voices take
    the block loop, the process's budget is halved and later segments are cut shorter -- the render stays exact. In a
    process of its own: the lowered budget is process-wide. "real" (ADVICE r04): the allocation is asked of hipMalloc and refused
    by it (64 TiB) -- the failed call's error stays the thread's last one on this runtime, and the launches behind it check
    hipGetLastError(): pool_alloc consumes it (and gives the pool's idle blocks back before it gives up)."""
    import subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
os.environ["SAU_AMD_TUNE"] = "1"; os.environ["SAU_AMD_CHAIN_ROWS_FAIL"] = {how!r}
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
tabs = np.fromfile(os.path.join({ROOT!r}, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
po.build(ref=False); po.oracle_use_tables(tabs); sa.set_piluts(tabs)
po.oracle().ora_set_fastmath_forms(2)
prg = vb.config5(n=70, seconds=7)
want = po.oracle_render(prg.ptr, 44100, False, chunk=150000)
for rep in range(2):  # the first engine meets the failure, the second the lowered budget
    b = sa.Batch([prg], 44100)
    got = b.render(stereo=False, chunk=150000)[0]
    tm = b.timing_ex(); b.close()
    assert len(got) == len(want) and (got == want).all(), (rep, int((got != want).sum()))
print("ROWS FALLBACK OK")
"""
    f = tmp_path / "rows.py"
    f.write_text(script)
    out = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ROWS FALLBACK OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
