"""GPU parity where the render depends on the caller's buffer size: the reference's block-lattice
line positions (sau/line.c:385-398, 430-445, 305-309; tests/lattice_cases.py)."""
import numpy as np
import pytest

from conftest import max_diff, ORACLE_FORMS
from lattice_cases import expiry_value_goal_program, lattice_case

pytestmark = pytest.mark.gpu


def _same(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    d = np.nonzero(got != want)[0]
    assert len(d) == 0, f"{what}: {len(d)} samples differ, first at {d[0]}: got {got[d[0]:d[0]+4].tolist()} want {want[d[0]:d[0]+4].tolist()}"


@pytest.mark.parametrize("rate", [44100, 96000])
@pytest.mark.parametrize("path", ["", "block-loop"])
def test_expiry_value_goal_sequences(sa, oracle, rate, path, monkeypatch):
    """Time expiry -> value-only event -> goal-only event (the ramp's length is what the block
    lattice made of the position): the drop-in generator behind 1746- and 11289-frame host calls
    (read-ahead runs of many calls each) and a single whole-script call, the batch API with one run
    per call -- each equal to the oracle at that call size, which equals the compiled reference
    there (tests/test_host.py)."""
    if path:
        monkeypatch.setenv("SAU_AMD_NO_FAST", "1")
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    differs = 0
    for seed in range(6):
        prg = expiry_value_goal_program(seed)
        renders = {}
        for call in (1746, 11289, 400000):
            want = oracle.oracle_render(prg.ptr, rate, True, chunk=call)
            renders[call] = want
            g = sa.Generator(prg, rate)
            _same(g.render(stereo=True, chunk=call), want, (seed, call, "drop-in"))
            g.close()
            if seed < 3:
                _same(sa.Batch([prg], rate).render(stereo=True, chunk=call)[0], want, (seed, call, "batch"))
        differs += bool((renders[1746] != renders[11289]).any() or (renders[1746] != renders[400000]).any())
    assert differs, "none of the programs depends on the call size"


@pytest.mark.parametrize("rate", [44100, 96000])
def test_random_lattice_programs(sa, oracle, rate):
    """Random trees with such event sequences on any line of any operator: engine runs of one
    length over host calls of another (sauAmd_Batch_set_call_len), and the drop-in generator."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    for seed in range(40):
        rng = np.random.default_rng(77000 + seed)
        prg = lattice_case(rng)
        call, run = int(rng.integers(300, 3000)), int(rng.integers(1000, 30000))
        stereo = bool(seed & 1)
        want = oracle.oracle_render(prg.ptr, rate, stereo, chunk=call)
        b = sa.Batch([prg], rate)
        b.set_call_len(call)
        _same(b.render(stereo=stereo, chunk=run)[0], want, (seed, call, run))
        b.close()
        if seed % 4 == 0:
            g = sa.Generator(prg, rate)
            _same(g.render(stereo=stereo, chunk=call), want, (seed, call, "drop-in"))
            g.close()


def test_lattice_programs_in_one_batch(sa, oracle):
    """Twelve such programs side by side: every program's events cut the others' segments."""
    oracle.oracle().ora_set_fastmath_forms(ORACLE_FORMS)
    prgs = [lattice_case(np.random.default_rng(77100 + k)) for k in range(12)]
    outs = sa.Batch(prgs, 44100).render(stereo=False, chunk=2500)
    for k, (prg, got) in enumerate(zip(prgs, outs)):
        _same(got, oracle.oracle_render(prg.ptr, 44100, False, chunk=2500), k)


@pytest.fixture()
def on_ref_tables(sa, oracle, tables):
    """Where the compiled reference is here (the GPU box): device and oracle on the wave tables that library built (conftest.py:
    ref_tables) for the test's duration -> True; else the fixture tables -> False."""
    from conftest import ref_tables
    if not oracle.have_ref():
        yield False
        return
    with ref_tables(sa, oracle, tables):
        yield True


def _render_calls(create, run, destroy, prg, rate, calls):
    """A host that changes its calls from one to the next: (frames, stereo) in turn, the last one until the script ends.
    -> the calls' PCM, one after the other."""
    import ctypes as C
    g = create(prg, rate)
    assert g
    n, out, k = C.c_size_t(), [], 0
    while True:
        size, stereo = calls[min(k, len(calls) - 1)]
        ch = 2 if stereo else 1
        k += 1
        buf = np.zeros(size * ch, np.int16)
        more = run(g, buf.ctypes.data, size, stereo, C.byref(n))
        out.append(buf[: n.value * ch].copy())
        if not more:
            break
    destroy(g)
    return np.concatenate(out)


def _render_pattern(create, run, destroy, prg, rate, stereo, sizes):
    return _render_calls(create, run, destroy, prg, rate, [(s, stereo) for s in sizes])


PATTERNS = [[1746, 300, 11289, 5, 1024, 4000], [11289, 11289, 11289, 700, 700, 20000], [1, 2, 3, 50000],
            [11289] * 9 + [1746] * 30 + [11289]]
# ... and ones that change the channel layout as well (True: stereo)
MIXED = [[(11289, False)] * 3 + [(11289, True)] * 4 + [(11289, False)],
         [(1746, True), (1746, True), (300, False), (300, False), (300, True), (11289, False), (5000, True)],
         [(4000, False)] * 20 + [(4000, True)] * 20 + [(1000, False)] * 7 + [(25000, True)]]


def test_a_host_that_changes_its_call_size(sa, oracle, on_ref_tables, monkeypatch):
    """The reference's output can depend on the size of the host's calls (its blocks restart at every call), and
    sauGenerator_run takes the size per call (sau/generator.c:905-913). The drop-in generator follows any pattern of sizes
    exactly -- with one engine run per call (SAU_AMD_READAHEAD=0) and, since round 5, in its default setting too: a call of
    another size takes the read-ahead back to the start of the run being handed out and goes on from what the host has
    consumed in the new lattice (capi.cpp: generator_rewind). Programs whose sound depends on the lattice
    (tests/lattice_cases.py), against the oracle -- which renders in the reference's own blocks -- and, where the compiled
    reference is here (it is on the GPU box), against that."""
    import saugns_amd.api as api
    lib, ora = api.lib(), oracle.oracle()
    ora.ora_set_fastmath_forms(ORACLE_FORMS)
    ref = oracle.ref() if on_ref_tables else None
    dep = [expiry_value_goal_program(s) for s in range(4)] + [lattice_case(np.random.default_rng(77300 + k)) for k in range(4)]
    differs = 0
    for stereo in (False, True):
        for k, prg in enumerate(dep):
            base = None
            for sizes in PATTERNS:
                want = _render_pattern(ora.ora_create, ora.ora_run, ora.ora_destroy, prg.ptr, 44100, stereo, sizes)
                if ref is not None and ORACLE_FORMS == 2:
                    _same(_render_pattern(ref.sau_create_Generator, ref.sauGenerator_run, ref.sau_destroy_Generator,
                                          prg.ptr, 44100, stereo, sizes), want, (k, sizes, "oracle vs compiled reference"))
                monkeypatch.setenv("SAU_AMD_READAHEAD", "0")
                got = _render_pattern(lib.sau_create_Generator, lib.sauGenerator_run, lib.sau_destroy_Generator,
                                      prg.ptr, 44100, stereo, sizes)
                _same(got, want, (k, sizes, "one run per call"))
                monkeypatch.delenv("SAU_AMD_READAHEAD")
                got = _render_pattern(lib.sau_create_Generator, lib.sauGenerator_run, lib.sau_destroy_Generator,
                                      prg.ptr, 44100, stereo, sizes)
                _same(got, want, (k, sizes, "read-ahead (default)"))
                differs += base is not None and (len(base) != len(want) or bool((base != want).any()))
                base = want
    assert differs > 8, differs  # (the patterns really render differently: the test means something)
    # a program without anything the lattice moves: exact under every pattern as well
    from saugns_amd import voicebank
    prg = voicebank.config3(n=8, seconds=1)
    for sizes in PATTERNS:
        want = _render_pattern(ora.ora_create, ora.ora_run, ora.ora_destroy, prg.ptr, 44100, False, sizes)
        got = _render_pattern(lib.sau_create_Generator, lib.sauGenerator_run, lib.sau_destroy_Generator,
                              prg.ptr, 44100, False, sizes)
        _same(got, want, (sizes, "read-ahead"))


def test_a_host_that_changes_its_channel_layout(sa, oracle, on_ref_tables):
    """`stereo` is an argument of every sauGenerator_run call as well (sau/generator.c:905): mono <-> stereo flips in
    mid-stream, with and without a change of size, with frames buffered for the other layout (until round 5: a failure,
    silence + false, which a host reads as the end of the script). Against the compiled reference itself."""
    import saugns_amd.api as api
    from conftest import need_ref
    need_ref(oracle)
    assert on_ref_tables
    lib, ref, ora = api.lib(), oracle.ref(), oracle.oracle()
    ora.ora_set_fastmath_forms(ORACLE_FORMS)
    from saugns_amd import voicebank
    prgs = [expiry_value_goal_program(s) for s in (1, 3)] + [lattice_case(np.random.default_rng(77400 + k)) for k in range(3)] + \
           [voicebank.config3(n=6, seconds=2), voicebank.config5(n=5, seconds=2)]
    for k, prg in enumerate(prgs):
        for calls in MIXED:
            want = _render_calls(ref.sau_create_Generator, ref.sauGenerator_run, ref.sau_destroy_Generator, prg.ptr, 44100, calls)
            if ORACLE_FORMS == 2:
                _same(_render_calls(ora.ora_create, ora.ora_run, ora.ora_destroy, prg.ptr, 44100, calls), want, (k, "oracle"))
                got = _render_calls(lib.sau_create_Generator, lib.sauGenerator_run, lib.sau_destroy_Generator, prg.ptr, 44100, calls)
                _same(got, want, (k, calls[:8], "device vs compiled reference"))
