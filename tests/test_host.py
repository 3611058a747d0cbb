"""Host logic without a GPU: C ABI surface, program images, control plane + plan
compiler (through the sequential test executor), and the loud failure without HIP."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_program, max_diff, ORACLE_FORMS


def test_library_exports_every_declared_symbol(sa):
    hdr = open(os.path.join(ROOT, "include", "saugns_amd.h")).read()
    names = re.findall(r"SAU_AMD_API[^;(]*?\b(sau\w+)\s*[\[(]", hdr)
    assert {"sau_create_Generator", "sau_destroy_Generator", "sauGenerator_run",
            "sauNoise_names"} <= set(names)
    L = sa.lib()
    for n in names:
        assert hasattr(L, n) or C.c_void_p.in_dll(L, n) is not None, n
    noise = (C.c_char_p * 8).in_dll(L, "sauNoise_names")
    assert [x.decode() for x in noise[:7]] == ["wh", "gw", "bw", "tw", "re", "vi", "bv"]
    assert noise[7] is None


def test_no_gpu_means_loud_failure(sa):
    """Without a HIP device the product refuses to run: no CPU fallback."""
    if sa.lib().sauAmd_device_count() > 0:
        pytest.skip("a GPU is present")
    from saugns_amd import voicebank
    with pytest.raises(RuntimeError, match="NULL"):
        sa.Generator(voicebank.config1(), 44100)
    assert "no HIP device" in sa.last_error()


def test_program_image_round_trip(sa):
    for key in ("examples__rainy_thunder", "devtests__voice-reuse", "examples__misc3-2pm_R"):
        blob = open(os.path.join(GOLDEN, "programs", key + ".saup"), "rb").read()
        prg = sa.Program.from_image(blob)
        assert prg.image() == blob
    with pytest.raises(ValueError):
        sa.Program.from_image(b"not an image at all.............................")


def test_abi_struct_sizes():
    from saugns_amd import api
    assert C.sizeof(api.SauLine) == 24 and C.sizeof(api.SauOpData) == 152
    assert C.sizeof(api.SauEvent) == 40 and C.sizeof(api.SauProgram) == 64
    assert api.SauOpData.pan.offset == 16 and api.SauOpData.camods.offset == 88
    assert api.SauOpData.mode.offset == 76 and api.SauProgram.ampmult.offset == 32


HOST_KEYS = ["devtests__voice-reuse", "devtests__pm-addremaddrem", "examples__dull_seq-fm_pm",
             "examples__rainy_thunder", "examples__tests__panning", "examples__tests__wavetypes",
             "examples__sounds__unnamed2", "examples__tests__scales", "examples__misc3-2pm_R",
             "examples__tests__sin_ramp_f-exp_log", "examples__random-blip_thump",
             "examples__tests__line_noisy", "devtests__compnest", "examples__sounds__stereo_static"]


@pytest.mark.parametrize("key", HOST_KEYS)
def test_control_plane_and_plans_vs_oracle(sa, oracle, seqexec, key):
    """Engine (events, voices, end detection) + plan compiler + shared arithmetic,
    executed sequentially, against the oracle: bit-exact, any block length."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = load_program(sa, key)
    want = oracle.oracle_render(prg.ptr, 12000, True)
    for block, chunk in ((1016, 11289), (64, 1000)):
        got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(block)).render(
            stereo=True, chunk=chunk)[0]
        assert max_diff(got, want) == 0, (key, block)


def test_run_return_values_match_reference_semantics(sa, oracle, seqexec):
    """(more, out_len) per call, like sauGenerator_run (generator.c:953-972)."""
    prg = load_program(sa, "devtests__voice-reuse")
    lib = oracle.oracle()
    g = lib.ora_create(prg.ptr, 12000)
    b = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(1016))
    buf = np.zeros(3001, np.int16)
    n = C.c_size_t()
    for _ in range(100):
        more_o = lib.ora_run(g, buf.ctypes.data, 3001, False, C.byref(n))
        pcm, more, lens = b.run(3001)
        assert (bool(more_o), n.value) == (more[0], lens[0])
        assert (pcm[0] == buf).all()
        if not more_o:
            break
    else:
        pytest.fail("never ended")
    lib.ora_destroy(g)


@pytest.mark.parametrize("readahead", ["", "0", "3000", "20000"])
def test_dropin_generator_call_sizes(sa, oracle, seqexec, readahead, monkeypatch):
    """sauGenerator_run hands the PCM of larger engine runs out piecewise from two buffers:
    (more, out_len) and the samples of every call equal the reference generator's
    (generator.c:905-973), whatever the call size and the size of a run."""
    if readahead:
        monkeypatch.setenv("SAU_AMD_READAHEAD", readahead)
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    lib = oracle.oracle()
    for key, stereo, call in (("devtests__voice-reuse", False, 3001), ("examples__tests__scales", True, 11289),
                              ("examples__dull_seq-fm_pm", False, 25000), ("devtests__voice-reuse", True, 1)):
        prg = load_program(sa, key)
        ch = 2 if stereo else 1
        o = lib.ora_create(prg.ptr, 12000)
        g = sa.Generator(prg, 12000, backend=seqexec.seq_backend_create(1016))
        want = np.zeros(call * ch, np.int16)
        got = np.full(call * ch, 77, np.int16)
        n = C.c_size_t()
        for i in range(200000 if call > 1 else 5000):
            more_o = bool(lib.ora_run(o, want.ctypes.data, call, stereo, C.byref(n)))
            more, out_len = g.run(got, call, stereo)
            assert (more, out_len) == (more_o, n.value), (key, i)
            assert (got == want).all(), (key, i)
            if not more_o:
                break
        else:
            assert call == 1
        # past the end: silence, zero frames, false -- as often as asked
        more, out_len = g.run(got, call, stereo)
        if call > 1:
            assert (more, out_len) == (False, 0) and not got.any()
        g.close()
        lib.ora_destroy(o)


def test_batch_streams_are_independent(sa, oracle, seqexec):
    """Programs with different event timelines in one batch == each alone."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    keys = ["devtests__voice-reuse", "examples__tests__scales", "examples__dull_seq-fm_pm"]
    prgs = [load_program(sa, k) for k in keys]
    outs = sa.Batch(prgs, 12000, backend=seqexec.seq_backend_create(1016)).render(stereo=False, chunk=5000)
    for k, prg, got in zip(keys, prgs, outs):
        want = oracle.oracle_render(prg.ptr, 12000, False)
        assert max_diff(got, want) == 0, k


def test_voicebank_builder_equals_parser_images(sa):
    """The parser-free builder lays out config 1 exactly like the reference parser."""
    from saugns_amd import voicebank
    want = sa.Program.from_image(open(os.path.join(GOLDEN, "programs", "config1.saup"), "rb").read())
    got = voicebank.config1()
    a, b = want.struct, got.struct
    assert (a.ev_count, a.vo_count, a.op_count, a.mode, a.duration_ms) == \
           (b.ev_count, b.vo_count, b.op_count, b.mode, b.duration_ms)
    oa, ob = a.events[0].op_data[0], b.events[0].op_data[0]
    for f in ("id", "params", "phase", "seed", "use_type", "type"):
        assert getattr(oa, f) == getattr(ob, f), f
    for ln in ("pan", "amp", "freq"):
        la, lb = getattr(oa, ln).contents, getattr(ob, ln).contents
        assert (la.v0, la.vt, la.time_ms, la.type, la.flags) == (lb.v0, lb.vt, lb.time_ms, lb.type, lb.flags)


def test_block_buffer_numbering(sa, seqexec):
    """Host side of the time-parallel path: block buffers renumbered by liveness when a plan is
    compiled (fast_slot_compact). A depth-3 PM chain needs one buffer where the block loop's plan
    names five; frequency blocks are only counted for voices that may have running-sum phases."""
    import ctypes as C
    from saugns_amd import voicebank as vb
    from saugns_amd.api import POP_FMOD, POP_PMOD

    def counts(voices):
        prg = vb.build_program(voices)
        b = sa.Batch([prg], 44100, backend=seqexec.seq_backend_create(256))
        b.run(512, stereo=False, fetch=False)
        out = (C.c_uint32 * 5)()
        seqexec.seq_backend_last_counts(out)
        return dict(zip(("n_main", "n_fast", "n_fast_full", "may_scan", "serial"), list(out)))

    m3 = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=0.4)
    m2 = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.7, mods={POP_PMOD: [m3]})
    m1 = vb.Op("sin", freq=vb.Line(1.0, ratio=True), amp=0.5, mods={POP_PMOD: [m2]})
    c = counts([vb.Op("sin", freq=110.0, time_ms=100, mods={POP_PMOD: [m1]})])
    assert c["n_main"] == 5 and c["n_fast"] == 1 and not c["may_scan"] and not c["serial"]
    fm = vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=30.0)
    c = counts([vb.Op("sin", freq=220.0, time_ms=100, mods={POP_FMOD: [fm]})])
    assert c["may_scan"] and c["n_fast_full"] >= 1 and c["n_fast_full"] >= c["n_fast"]
    c = counts([vb.Op("sin", freq=220.0, pm_a=0.5, time_ms=100)])
    assert c["serial"]


def _pm_chain(depth):
    from saugns_amd.voicebank import Op, Line
    from saugns_amd.api import POP_PMOD
    op = None
    for d in range(depth):
        top = d == depth - 1
        op = Op("sin", freq=200.0 if top else Line(2.0, ratio=True), amp=0.5, time_ms=50 if top else None,
                mods={POP_PMOD: [op]} if op else {})
    return op


def _chain(depth, use, ratio=True, wave="sin"):
    """A straight chain of `depth` operators, each the only member of its parent's list `use`."""
    from saugns_amd.voicebank import Op, Line
    op = None
    for d in range(depth):
        top = d == depth - 1
        kw = {}
        if top:
            kw = dict(freq=200.0, time_ms=50)
        else:
            kw = dict(freq=Line(1.0 + (d % 5) * 0.5, ratio=True) if ratio else 30.0 + 11.0 * (d % 7))
        op = Op(wave, amp=0.5 if use != "f" else 20.0, mods={use: [op]} if op else {}, **kw)
    return op


def test_nesting_as_deep_as_the_reference(sa, oracle, seqexec):
    """sauProgram.op_nest_depth is a uint8 (sau/program.h:259; generator.c:133 gives every level its 7 buffers):
    the reference takes 256 operators on a path. A straight PM chain renders bit-exact at 63, 65, 120 (plans
    with 8-bit buffer ids) and at 130, 255, 256 levels (wide plans: step pairs with 16-bit ids);
    chains through the other modulator lists -- FM, range FM (three buffers per level), AM, range AM,
    frequency-scaled PM -- at 200 and 256 levels; a path of 257 operators is not a sauProgram: refused with a
    message and silence (the reference host then simply ends: generator.c:905 cannot fail). Any number of
    modulators in one list is fine (they share a buffer)."""
    from saugns_amd import voicebank
    from saugns_amd.voicebank import Op, Line
    from saugns_amd.api import POP_PMOD, POP_FMOD, POP_RFMOD, POP_AMOD, POP_RAMOD, POP_FPMOD
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for depth in (63, 65, 120, 130, 255, 256):
        prg = voicebank.build_program([_pm_chain(depth)])
        got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(256)).render()[0]
        assert max_diff(got, oracle.oracle_render(prg.ptr, 12000, False)) == 0, depth
    for use in (POP_FMOD, POP_RFMOD, POP_AMOD, POP_RAMOD, POP_FPMOD):
        for depth in (200, 256):
            prg = voicebank.build_program([_chain(depth, use, ratio=use not in (POP_FMOD, POP_RFMOD))])
            got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(333)).render()[0]
            want = oracle.oracle_render(prg.ptr, 12000, False)
            assert len(got) == len(want) and max_diff(got, want) == 0, (use, depth)
    # a wide plan beside ordinary voices, events on both, several calls
    voices = [_pm_chain(150), Op("saw", freq=110.0, amp=0.3, time_ms=80), _chain(140, POP_AMOD)]
    prg = voicebank.build_program(voices)
    want = oracle.oracle_render(prg.ptr, 12000, True)
    for chunk in (0, 257):
        got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(200)).render(stereo=True, **({"chunk": chunk} if chunk else {}))[0]
        assert len(got) == len(want) and max_diff(got, want) == 0
    prg = voicebank.build_program([_pm_chain(257)])
    with pytest.raises(RuntimeError, match="nesting deeper than 256 levels"):
        sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(256)).render()
    g = sa.Generator(prg, 12000, backend=seqexec.seq_backend_create(256))
    buf = np.full(1000, 5, np.int16)
    assert g.run(buf, 1000) == (False, 0) and not buf.any()
    assert "nesting deeper than 256 levels" in sa.last_error()
    g.close()
    mods = [Op("sin", freq=Line(float(1 + i % 7), ratio=True), amp=0.1) for i in range(300)]
    prg = voicebank.build_program([Op("sin", freq=200.0, amp=0.5, time_ms=50, mods={POP_PMOD: mods})])
    got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(256)).render()[0]
    assert max_diff(got, oracle.oracle_render(prg.ptr, 12000, False)) == 0


def test_two_generators_alternately(sa, oracle, seqexec):
    """saugns.c:585 creates a second generator at the device rate (`split_gen`) and calls both in
    turn from one thread: the two must not share mutable state (threading row)."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prg = load_program(sa, "examples__dull_seq-fm_pm")
    rates = (12000, 16000)
    want = [oracle.oracle_render(prg.ptr, r, True) for r in rates]
    gens = [sa.Generator(prg, r, backend=seqexec.seq_backend_create(1016)) for r in rates]
    outs = [[], []]
    bufs = [np.zeros(2 * 3000, np.int16), np.zeros(2 * 4000, np.int16)]
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
        assert max_diff(np.concatenate(outs[i]), want[i]) == 0


def test_pan_modulator_shorter_than_its_carrier(sa, oracle, seqexec):
    """generator.c:762-771: pan modulators run for the carrier's length and add nothing once their
    own time is over; the voice goes on with the pan line alone (the steps after a pan modulator
    get the carrier's length back)."""
    from saugns_amd import voicebank
    from saugns_amd.voicebank import Op, Line
    from saugns_amd.api import POP_CAMOD
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for mod_ms in (None, 30, 70, 200):
        m = Op("spa", freq=Line(1.0, ratio=True), amp=0.48, time_ms=mod_ms)
        c = Op("tri", freq=144.3, amp=0.7, time_ms=136, pan=0.25, mods={POP_CAMOD: [m]})
        prg = voicebank.build_program([c])
        want = oracle.oracle_render(prg.ptr, 12000, True)
        for block, chunk in ((1016, 11289), (64, 500)):
            got = sa.Batch([prg], 12000, backend=seqexec.seq_backend_create(block)).render(stereo=True, chunk=chunk)[0]
            assert max_diff(got, want) == 0, (mod_ms, block)


def test_program_images_of_random_programs(sa, oracle):
    """sauAmd_program_serialize / sauAmd_program_load over programs with every operator type,
    modulator list and event kind: the loaded image renders the same samples (oracle) and
    serializes to the same bytes again."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_units as tu
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for seed in range(40):
        rng = np.random.default_rng(5000 + seed)
        voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        tu._random_starts(rng, voices)
        prg = vb.build_program(voices, updates=tu._random_updates(rng, voices))
        blob = prg.image()
        back = sa.Program.from_image(blob)
        assert back.image() == blob, seed
        a = oracle.oracle_render(prg.ptr, 22050, True)
        b = oracle.oracle_render(back.ptr, 22050, True)
        assert len(a) == len(b) and max_diff(a, b) == 0, seed


@pytest.mark.parametrize("rate", [44100, 96000])
def test_line_positions_follow_the_reference_block_lattice(sa, oracle, seqexec, rate):
    """Same program, call sizes 1746 / 11289 / whole: the oracle (which renders in the reference's
    own blocks) and the engine agree sample for sample at each -- whatever blocks the backend
    itself renders in, alone or beside another program whose events cut the segments elsewhere,
    and through the read-ahead of the drop-in generator. (That the renders differ between call
    sizes is the reference's behaviour; asserted so that the programs keep exercising it.)"""
    from saugns_amd import voicebank as vb
    from lattice_cases import expiry_value_goal_program
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    seen_dependence = False
    for seed in range(6):
        prg = expiry_value_goal_program(seed)
        renders = {}
        for chunk in (1746, 11289, 400000):
            want = oracle.oracle_render(prg.ptr, rate, True, chunk=chunk)
            renders[chunk] = want
            if oracle.have_ref():
                oracle.oracle().ora_set_fastmath_forms(2)
                ref = oracle.ref_render(prg.ptr, rate, True, chunk=chunk)
                assert max_diff(oracle.oracle_render(prg.ptr, rate, True, chunk=chunk), ref) == 0, (seed, chunk)
                oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
            for block in ((333, 1016, 50000) if seed < 2 else (1016,)):
                got = sa.Batch([prg], rate, backend=seqexec.seq_backend_create(block)).render(stereo=True, chunk=chunk)[0]
                assert max_diff(got, want) == 0, (seed, chunk, block)
        if any((renders[1746] != renders[c]).any() for c in (11289, 400000)):
            seen_dependence = True
        # beside a program with many events of its own (segments end where they fall)
        busy_voices = [vb.Op("sin", freq=150.0, time_ms=3500)]
        busy = vb.build_program(busy_voices, updates=[(k * 53 + 7, 0, busy_voices[0], {"amp": vb.Line(0.5 + 0.01 * (k % 7))})
                                                       for k in range(60)])
        got = sa.Batch([prg, busy], rate, backend=seqexec.seq_backend_create(1016)).render(stereo=True, chunk=11289)[0]
        assert max_diff(got, renders[11289]) == 0, (seed, "batch")
        # drop-in generator: engine runs of 4 host calls
        os.environ["SAU_AMD_READAHEAD"] = str(4 * 1746 + 100)
        try:
            got = sa.Generator(prg, rate, backend=seqexec.seq_backend_create(1016)).render(stereo=True, chunk=1746)
        finally:
            del os.environ["SAU_AMD_READAHEAD"]
        assert max_diff(got, renders[1746]) == 0, (seed, "read-ahead")
    assert seen_dependence, "none of the programs depends on the call size: the test has lost its subject"


@pytest.mark.parametrize("rate", [8000, 44100, 96000])
def test_random_lattice_programs(sa, oracle, seqexec, rate):
    """Random operator trees with (value-only, then goal-only) events on lines whose time has run out
    (tests/lattice_cases.py): engine runs of any length over host calls of another, backend blocks
    of a third -- sample for sample the oracle at that call size; the oracle itself equals the
    compiled reference there."""
    from lattice_cases import lattice_case
    dep = 0
    for seed in range(40):
        rng = np.random.default_rng(77000 + seed)
        prg = lattice_case(rng)
        call, run = int(rng.integers(300, 3000)), int(rng.integers(1000, 30000))
        block = int(rng.choice([64, 1016, 4000]))
        stereo = bool(seed & 1)
        oracle.oracle().ora_set_fastmath_forms(0)
        a = oracle.oracle_render(prg.ptr, rate, stereo, chunk=call)
        b = oracle.oracle_render(prg.ptr, rate, stereo, chunk=4000000)
        dep += len(a) != len(b) or bool((a != b).any())
        if oracle.have_ref() and seed < 12:
            oracle.oracle().ora_set_fastmath_forms(2)
            assert max_diff(oracle.oracle_render(prg.ptr, rate, stereo, chunk=call),
                            oracle.ref_render(prg.ptr, rate, stereo, chunk=call)) == 0, seed
        oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
        want = oracle.oracle_render(prg.ptr, rate, stereo, chunk=call)
        bt = sa.Batch([prg], rate, backend=seqexec.seq_backend_create(block))
        bt.set_call_len(call)
        got = bt.render(stereo=stereo, chunk=run)[0]
        assert max_diff(got, want) == 0, (seed, call, run, block)
    assert dep >= 8, "too few of the programs depend on the call size"


def test_malformed_program_images_are_rejected(sa):
    """sauAmd_program_load on damaged images: NULL (ValueError here), never a wild access."""
    import struct
    blob = bytearray(open(os.path.join(GOLDEN, "programs", "config1.saup"), "rb").read())
    good = sa.Program.from_image(bytes(blob))
    assert good.struct.ev_count >= 1
    total = struct.unpack_from("<Q", blob, 8)[0]
    assert total == len(blob)

    def rejected(b):
        with pytest.raises(ValueError):
            sa.Program.from_image(bytes(b))

    for bad_total in (0, 8, 16, 79, total + 8, 2 ** 63):  # size field vs header and buffer
        b = bytearray(blob); struct.pack_into("<Q", b, 8, bad_total); rejected(b)
    ev_off = struct.unpack_from("<Q", blob, 16)[0]       # sauProgram.events (offset form)
    b = bytearray(blob); struct.pack_into("<Q", b, 16, 0); rejected(b)             # events counted, none there
    b = bytearray(blob); struct.pack_into("<Q", b, 16, ev_off + 4); rejected(b)    # misaligned
    b = bytearray(blob); struct.pack_into("<Q", b, 16, 8); rejected(b)             # inside the header
    b = bytearray(blob); struct.pack_into("<Q", b, 24, 2 ** 61); rejected(b)       # ev_count * 40 wraps
    b = bytearray(blob); struct.pack_into("<Q", b, 24, total); rejected(b)         # more events than bytes
    # first event: op_data_count / op_data offset (sauProgramEvent: wait u32, vo u16, carr u32, op_count u32,
    # op_data_count u32, op_list ptr, op_data ptr)
    b = bytearray(blob); struct.pack_into("<I", b, ev_off + 16, 0x7fffffff); rejected(b)
    b = bytearray(blob); struct.pack_into("<Q", b, ev_off + 32, 0); rejected(b)
    b = bytearray(blob); struct.pack_into("<Q", b, ev_off + 32, total - 8); rejected(b)
    rejected(blob[: total // 2])
    rejected(b"SAUPIMG1" + bytes(200))
    # every single-byte corruption of the pointer-bearing part either loads or is rejected cleanly
    rng = np.random.default_rng(5)
    for _ in range(400):
        b = bytearray(blob)
        b[int(rng.integers(8, len(b)))] = int(rng.integers(256))
        try:
            sa.Program.from_image(bytes(b))
        except ValueError:
            pass


def test_amp_operator_through_the_plan(sa, oracle, seqexec):
    """A operator cases (tests/test_gpu_units.py::amp_operator_cases) through the engine and the
    plan format, executed sequentially: bit-exact vs the oracle."""
    import test_gpu_units as tu
    from saugns_amd import voicebank as vb
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for name, voices, ups in tu.amp_operator_cases():
        prg = vb.build_program(voices, updates=ups)
        want = oracle.oracle_render(prg.ptr, 44100, True, chunk=777)
        for block in (64, 1016):
            got = sa.Batch([prg], 44100, backend=seqexec.seq_backend_create(block)).render(stereo=True, chunk=777)[0]
            assert max_diff(got, want) == 0, (name, block)


def test_bank_builder_in_the_c_abi(sa, oracle):
    """sauAmd_build_bank (parser-free voice banks) lays programs out as the Python builder
    does (which tests/test_host.py::test_voicebank_builder_equals_parser_images pins on parser-made
    images): the oracle renders both to the same PCM, for the BASELINE banks and random trees; bad
    descriptions are refused."""
    import test_gpu_units as tu
    from saugns_amd import voicebank as vb
    from saugns_amd.api import OpDesc
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)

    def voices_of(kind):
        if kind == "config3":
            return [vb.Op("sin", freq=110.0 + i * 0.731, time_ms=300, mods={6: [
                vb.Op("sin", freq=vb.Line(float(1 + i % 5), ratio=True), amp=0.5, mods={6: [
                    vb.Op("sin", freq=vb.Line(2.0, ratio=True), amp=0.7)]})]}) for i in range(24)]
        rng = np.random.default_rng(kind)
        vs = [tu._random_voice(rng) for _ in range(3)]
        for k, v in enumerate(vs):
            v.start_ms = 10 * k
        return vs

    for kind in ["config3"] + list(range(12)):
        a = vb.build_program(voices_of(kind))
        b = vb.build_bank_c(voices_of(kind))
        assert (a.struct.ev_count, a.struct.vo_count, a.struct.op_count, a.struct.op_nest_depth, a.struct.duration_ms) == \
               (b.struct.ev_count, b.struct.vo_count, b.struct.op_count, b.struct.op_nest_depth, b.struct.duration_ms), kind
        pa = oracle.oracle_render(a.ptr, 44100, True)
        pb = oracle.oracle_render(b.ptr, 44100, True)
        assert len(pa) == len(pb) and (pa == pb).all(), kind
    bad = (OpDesc * 2)()
    bad[0].use, bad[1].use, bad[1].parent = 0, 6, 7  # parent out of range
    assert not sa.lib().sauAmd_build_bank(bad, 2, 1.0, 1000)
    bad[1].parent = 1  # its own parent
    assert not sa.lib().sauAmd_build_bank(bad, 2, 1.0, 1000)
    bad[0].use = 6; bad[0].parent = 1; bad[1].parent = 0  # no carrier
    assert not sa.lib().sauAmd_build_bank(bad, 2, 1.0, 1000)

    def one(**kw):  # a well-formed one-operator bank with one field spoiled
        d = (OpDesc * 1)()
        d[0].use, d[0].type, d[0].mode, d[0].time_ms = 0, 2, 0, 100
        d[0].freq.present, d[0].freq.v0, d[0].amp.present, d[0].amp.v0 = 1, 440.0, 1, 1.0
        for k, v in kw.items():
            obj, name = d[0], k
            if "__" in k:
                obj, name = getattr(d[0], k.split("__")[0]), k.split("__")[1]
            setattr(obj, name, v)
        p = sa.lib().sauAmd_build_bank(d, 1, 1.0, 1000)
        if p:
            sa.lib().sauAmd_free_bank(p)
        return bool(p)

    assert one()
    assert not one(type=4)                      # no such operator type
    assert not one(mode=12)                     # no such wave
    assert not one(type=1, mode=7)              # no such noise
    assert not one(type=3, mode=13)             # R: no such line
    assert not one(type=3, mode=6 << 16)        # R: no such function
    assert not one(freq__shape=13)              # no such ramp shape
    assert not one(time_ms=0)                   # a carrier without a length
    assert not one(start_ms=0xFFFFFFF0, time_ms=100)  # duration beyond 32 bits


def _calls_pattern(rng, n):
    """(frames, stereo) per call: runs of equal calls broken by changes of size, of layout, or of both"""
    calls, size, stereo = [], int(rng.integers(1, 4000)), bool(rng.integers(2))
    while len(calls) < n:
        for _ in range(int(rng.integers(1, 9)) if size >= 40 else int(rng.integers(1, 60))):
            calls.append((size, stereo))
        what = int(rng.integers(3))
        if what != 1:
            size = int([rng.integers(1, 40), rng.integers(40, 1200), rng.integers(40, 1200), rng.integers(1200, 4000),
                        rng.integers(4000, 30000)][int(rng.integers(5))])
        if what != 0:
            stereo = not stereo
    return calls[:n]


@pytest.mark.parametrize("readahead", ["", "6000", "60000"])
def test_dropin_generator_follows_changes_of_call_size_and_layout(sa, oracle, seqexec, readahead, monkeypatch):
    """sauGenerator_run takes buf_len and stereo per call (sau/generator.c:905-913): a host may change either from one call to
    the next, and what it gets depends on both -- the block lattice restarts at every call (854-878), mono is (L + R) / 2 before
    rounding. The drop-in generator renders ahead for calls like the last one; a call of another kind takes the engine back to
    the start of the run being handed out (Engine::snapshot / restore), renders what was consumed again and goes on in the new
    size / layout: every call's (more, out_len) and samples equal the reference generator's, with the read-ahead on (VERDICT r04
    item 2; until round 5 a layout change failed and a size change kept the old lattice for up to two runs). The programs
    are ones whose sound depends on the lattice (tests/lattice_cases.py)."""
    from lattice_cases import expiry_value_goal_program, lattice_case
    if readahead:
        monkeypatch.setenv("SAU_AMD_READAHEAD", readahead)
    lib = oracle.oracle()
    lib.ora_set_fastmath_forms(ORACLE_FORMS)
    rewinds = 0
    for seed in range(6):
        rng = np.random.default_rng(4100 + seed)
        prg = expiry_value_goal_program(seed) if seed < 3 else lattice_case(rng)
        calls = _calls_pattern(rng, 4000)
        o = lib.ora_create(prg.ptr, 44100)
        g = sa.Generator(prg, 44100, backend=seqexec.seq_backend_create(1016))
        n = C.c_size_t()
        for i, (size, stereo) in enumerate(calls):
            ch = 2 if stereo else 1
            want = np.zeros(size * ch, np.int16)
            got = np.full(size * ch, 77, np.int16)
            more_o = bool(lib.ora_run(o, want.ctypes.data, size, stereo, C.byref(n)))
            more, out_len = g.run(got, size, stereo)
            assert (more, out_len) == (more_o, n.value), (seed, i, size, stereo)
            d = np.nonzero(got != want)[0]
            assert len(d) == 0, (seed, i, size, stereo, len(d), int(d[0]), got[d[:4]].tolist(), want[d[:4]].tolist())
            if not more_o:
                break
        else:
            raise AssertionError("the script did not end within the pattern")
        rewinds += g.rewinds()
        g.close()
        lib.ora_destroy(o)
    assert rewinds > 30, rewinds  # (the read-ahead was on and was taken back, not bypassed)


def test_every_backend_has_its_own_budget_for_the_chains_rows(sa, oracle, seqexec):
    """VERDICT r05 item 8 / ADVICE r04: the budget of the feedback chains' rows -- which decides where the engine cuts segments with
    feedback voices -- belongs to the backend (its device's free memory, its own failed allocations), not to the process: two
    sequential executors stand in for two devices, one roomy, one nearly full and with two failed allocations behind it. The same
    4 feedback voices are cut into segments of different lengths, each by its own budget, and a third engine on the roomy
    "device" afterwards is not affected by the other's failures. Same PCM every time."""
    import ctypes as C
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    from saugns_amd import voicebank as vb
    prg = vb.config5(n=4, seconds=8)  # 4 chains x 8 s at 48 kHz: 384000 frames
    want = oracle.oracle_render(prg.ptr, 48000, False)

    def run(free_bytes, failures):
        be = seqexec.seq_backend_create(1024)
        seqexec.seq_backend_set_memory(be, free_bytes, failures)
        b = sa.Batch([prg], 48000, backend=be)
        got = b.render(stereo=False, chunk=len(want))[0]
        seg = (C.c_uint32 * 2)()
        seqexec.seq_backend_segments(be, seg)
        b.close()
        assert max_diff(got, want) == 0
        return seg[0], seg[1]

    roomy = run(64 << 30, 0)          # an eighth of 64 GiB: the whole render is one segment (4 chains x 8 B x 384000 frames)
    tight = run(256 << 20, 2)         # 256 MiB free, two failures: (32 MiB >> 2) / 32 B = 262144 frames per segment
    assert roomy[1] >= len(want) and roomy[0] <= 2, roomy
    assert tight[1] == 262144 and tight[0] >= 2, tight
    assert run(64 << 30, 0) == roomy  # the tight device's failures are its own


_EXPIRY_CHILD = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_PMOD
import hashlib
import conftest
sa.lib(); sa.set_piluts(np.fromfile(os.path.join(sys.argv[1], "tests", "golden", "piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
sa.api.use_hooks(conftest.hooks_path())
lib = C.CDLL(sys.argv[2])
lib.seq_backend_create.restype = C.c_void_p; lib.seq_backend_create.argtypes = [C.c_uint32]
lib.seq_backend_segments.argtypes = [C.c_void_p, C.c_void_p]
short = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0, time_ms=700)    # runs out at frame 30870
shorter = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.6, time_ms=250)  # ... and at 11025
prg = vb.build_program([vb.Op("sin", freq=146.832, time_ms=2000, mods={POP_PMOD: [short]}),
                        vb.Op("sin", freq=220.0, time_ms=2000, mods={POP_PMOD: [shorter]})])
be = lib.seq_backend_create(1024)
b = sa.Batch([prg], 44100, backend=be)
pcm = b.render(stereo=False, chunk=88200)[0]
seg = (C.c_uint32 * 2)(); lib.seq_backend_segments(be, seg); b.close()
print("RESULT " + json.dumps({"segments": seg[0], "longest": seg[1], "sha": hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest()}))
"""


def test_a_long_segment_ends_soon_after_an_operator_runs_out_of_time(seqexec, oracle):
    """generator.c:686-700: an operator out of time stands still and yields silence while its voice plays on. The time-parallel
    path renders a voice up to the first such frame and the block loop takes it from there to the segment's end, so the engine cuts
    a long segment on a grid of 8192 frames after that frame (engine.cpp); SAU_AMD_EXPIRY_GRID=1 cuts at the frame itself. Two
    modulators running out at frames 11025 and 30870 of an 88200-frame run: 16384 | 32768 | the rest on the grid, 11025 | 30870 | the
    rest without it; the same PCM, which is the oracle's."""
    import json
    import subprocess
    import sys
    import hashlib
    from saugns_amd import voicebank as vb
    from saugns_amd.api import POP_PMOD
    path = os.environ.get("SAU_SEQEXEC_LIB") or os.path.join(ROOT, "tests", "seqexec", "libseqexec.so")

    def run(grid):
        env = dict(os.environ, SAU_AMD_TUNE="1")
        env.pop("SAU_AMD_EXPIRY_GRID", None)
        if grid:
            env["SAU_AMD_EXPIRY_GRID"] = grid
        out = subprocess.run([sys.executable, "-c", _EXPIRY_CHILD, ROOT, path], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, out.stderr[-2000:]
        return json.loads(line[0][7:])

    on_grid, exact = run(None), run("1")
    assert on_grid["segments"] == 3 and on_grid["longest"] == 88200 - 32768, on_grid
    assert exact["segments"] == 3 and exact["longest"] == 88200 - 30870, exact
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    short = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=1.0, time_ms=700)
    shorter = vb.Op("tri", freq=vb.Line(2.0, ratio=True), amp=0.6, time_ms=250)
    prg = vb.build_program([vb.Op("sin", freq=146.832, time_ms=2000, mods={POP_PMOD: [short]}),
                            vb.Op("sin", freq=220.0, time_ms=2000, mods={POP_PMOD: [shorter]})])
    want = hashlib.sha256(np.ascontiguousarray(oracle.oracle_render(prg.ptr, 44100, False, chunk=88200)).tobytes()).hexdigest()
    assert on_grid["sha"] == want and exact["sha"] == want


def test_product_library_exports_only_the_abi():
    """test probes and hooks are not exports of libsaugns_amd.so (they live in tests/hooks). What it
    exports is include/saugns_amd.h + the reference's four generator symbols (+ the kernels' host stubs, which hipcc emits
    with default visibility)."""
    import re
    import subprocess
    import saugns_amd
    out = subprocess.run(["nm", "-D", "--defined-only", saugns_amd.build()], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    bad = [x for x in names if re.search(r"kat_|with_backend|_rewinds|seq_backend", x)]
    assert not bad, bad
    hdr = open(os.path.join(ROOT, "include", "saugns_amd.h")).read()
    declared = set(re.findall(r"\b(sauAmd_\w+|sau_create_Generator|sau_destroy_Generator|sauGenerator_run|sauNoise_names)\s*[\(\[]", hdr))
    exported = {x for x in names if x.startswith(("sauAmd_", "sau_", "sauGenerator_", "sauNoise_"))}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
