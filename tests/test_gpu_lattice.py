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
