"""Model check of the look-back protocol of saugns_amd/csrc/k_common.h (lookback32 with the ring in LDS): the
waves of one voice as state machines, every read and write of a ring word its own atomic step, interleaved by
an adversarial random scheduler (including waves that stall for long stretches). Checked: every group gets the
exact prefix of the totals before it, nobody needs a ring entry that has been overwritten (which would show
as a wave that never finishes), and the ring of 4 x waves entries is needed -- with 2 x waves the same
schedules do go wrong. CPU only."""
import random

import pytest

AGG, PREFIX = 1, 2


def run(n_waves, n_groups, ring, seed, stall=0.0, max_steps=2_000_000):
    rng = random.Random(seed)
    totals = [rng.randrange(1 << 32) for _ in range(n_groups)]
    slots = [(0, 0, 0)] * ring  # (tag, status, value), one atomic word each
    result = [None] * n_groups

    def wave(w):
        """generator: one yield per atomic access"""
        cg = w
        while cg < n_groups:
            tot = totals[cg]
            if cg == 0:
                slots[0] = (1, PREFIX, tot); yield
                result[0] = 0
            else:
                slots[cg % ring] = (cg + 1, AGG, tot); yield
                excl, p = 0, cg - 1
                while True:
                    window = []
                    for lane in range(64):  # lane l looks at group p - l; each read is its own step
                        idx = p - lane
                        if idx < 0:
                            window.append((PREFIX, 0))
                        elif cg - idx > ring:
                            window.append((0, 0))  # further back than the ring: reads as empty
                        else:
                            tag, st, val = slots[idx % ring]; yield
                            window.append((st, val) if tag == idx + 1 else (0, 0))
                    first_pref = next((i for i, e in enumerate(window) if e[0] == PREFIX), 64)
                    first_none = next((i for i, e in enumerate(window) if e[0] == 0), 64)
                    upto = first_pref + 1 if first_pref < first_none else first_none
                    excl = (excl + sum(e[1] for e in window[:upto])) & 0xFFFFFFFF
                    if first_pref < first_none:
                        break
                    p -= upto
                    yield
                slots[cg % ring] = (cg + 1, PREFIX, (excl + tot) & 0xFFFFFFFF); yield
                result[cg] = excl
            for _ in range(rng.randrange(0, 6)):  # the rest of the row group's work
                yield
            cg += n_waves

    live = {w: wave(w) for w in range(n_waves)}
    stalled = {}
    steps = 0
    while live:
        steps += 1
        if steps > max_steps:
            return None  # somebody waits for an entry that will never come back
        w = rng.choice(list(live))
        if stalled.get(w, 0) > 0:
            stalled[w] -= 1
            if len(live) > 1 and any(stalled.get(x, 0) == 0 for x in live if x != w):
                continue
        elif rng.random() < stall:
            stalled[w] = rng.randrange(50, 3000)
            continue
        try:
            next(live[w])
        except StopIteration:
            del live[w]
    want, acc = [], 0
    for t in totals:
        want.append(acc)
        acc = (acc + t) & 0xFFFFFFFF
    return result == want


@pytest.mark.parametrize("n_waves", [2, 3, 4, 7, 9, 16])
def test_ring_of_four_times_the_waves_serves_every_schedule(n_waves):
    for seed in range(3):
        for stall in (0.0, 0.002, 0.02):
            assert run(n_waves, 30 * n_waves + seed, 4 * n_waves, 1000 * n_waves + seed, stall) is True


def test_a_ring_of_twice_the_waves_is_too_small():
    """the bound is not slack: with fewer entries some schedule overwrites what a slow wave still has to read"""
    outcomes = [run(n, 60 * n, 2 * n, 77 + s, 0.02, max_steps=400_000) for n in (2, 3, 4) for s in range(12)]
    assert any(o is not True for o in outcomes)
