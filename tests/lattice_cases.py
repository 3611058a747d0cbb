"""Programs whose sound depends on the reference's block-lattice line positions (test data).

sau/line.c:385-398 (advance_len), 430-445 (sauLine_run) and 456-473 (sauLine_skip) move the position
of a line without a sweep once per <= 1024-frame block, restarting it at 0 for the whole block when
its time runs out or its goal is reached; sauLine_copy (305-309) reads it when a later event sets only
a goal with inherited time. Blocks restart at every sauGenerator_run call and at every event
(generator.c:854-878, 917-946), so such renders depend on the caller's buffer size.
"""
import numpy as np

import test_gpu_units as tu
from saugns_amd import voicebank as vb
from saugns_amd.api import POP_AMOD, POP_PMOD, POPT_NOISE


def lattice_case(rng):
    voices = [tu._random_voice(rng) for _ in range(int(rng.integers(1, 3)))]
    tu._random_starts(rng, voices)
    def nodes(op, acc):
        acc.append(op)
        for lst in op.mods.values():
            for m in lst: nodes(m, acc)
        return acc
    ups = []
    mod_ms = int(rng.integers(20, 120))
    for vi, carr in enumerate(voices):
        t0 = getattr(carr, "start_ms", 0) or 0
        first = carr.time_ms  # 40..160: lines given at the start run out then
        carr_lines = ["amp"] + (["freq"] if carr.freq is not None else []) + ["pan"]
        # keep the carrier going: new durations through events on one line, so the others' times run out
        t = t0 + int(first * 0.6)
        keep = carr_lines[int(rng.integers(len(carr_lines)))]
        total_end = t0 + int(rng.integers(500, 900))
        ups.append((t, vi, carr, {keep: vb.Line(float(rng.uniform(0.2, 1.0)) * (300.0 if keep == "freq" else 1.0)), "time_ms": total_end - t}))
        ops = nodes(carr, [])
        for _ in range(int(rng.integers(2, 6))):
            op = ops[int(rng.integers(len(ops)))]
            names = ["amp"]
            if op.op_type != POPT_NOISE and op.freq is not None: names.append("freq")
            if op is carr: names.append("pan")
            if op.amp2 is not None: names.append("amp2")
            if op.op_type != POPT_NOISE and op.freq2 is not None: names.append("freq2")
            if op.op_type != POPT_NOISE and op.pm_a is not None: names.append("pm_a")
            name = names[int(rng.integers(len(names)))]
            ratio = name in ("freq", "freq2") and getattr(op, name).ratio
            scale = 1.0 if (ratio or name not in ("freq", "freq2")) else 300.0
            lo = t0 + (first if op is carr else mod_ms) + 5
            ta = int(rng.integers(lo, max(lo + 1, total_end - 200)))
            ups.append((ta, vi, op, {name: vb.Line(float(rng.uniform(0.2, 1.5)) * scale, ratio=ratio)}))
            tb = ta + int(rng.integers(5, 150))
            ups.append((tb, vi, op, {name: vb.Line(0.0, goal=float(rng.uniform(0.1, 1.5)) * scale, ratio=ratio,
                                                     shape=["lin", "cos", "xpe", "uwh"][int(rng.integers(4))], state=False)}))
    return vb.build_program(voices, updates=ups, default_mod_ms=mod_ms)


def expiry_value_goal_program(seed):
    """Event sequences whose sound depends on the reference's block-lattice line positions
    (sau/line.c:385-398, 430-445, 305-309): a line's time runs out, its position keeps cycling
    block by block, a later event sets only a value (new time, old position), a still later one only
    a goal with inherited time (`end -= pos`): the ramp's length is what the position was."""
    rng = np.random.default_rng(seed)
    voices, updates = [], []
    for v in range(3):
        m = vb.Op("sin", freq=vb.Line(float(rng.integers(1, 4)), ratio=True), amp=float(rng.uniform(0.5, 2.0)),
                  time_ms=None if v != 1 else 2600)
        a = vb.Op("tri", freq=float(rng.uniform(2, 9)), amp=vb.Line(0.3, goal=1.0, shape="lin"), time_ms=int(rng.integers(150, 700)))
        c = vb.Op("sin", freq=float(rng.uniform(100, 400)), amp=vb.Line(0.8, goal=0.4, shape="cos") if v == 2 else 0.9,
                  time_ms=3000, mods={POP_PMOD: [m], POP_AMOD: [a]}, pan=0.3 * v - 0.3)
        c.start_ms = int(rng.integers(0, 300)) if v else 0
        voices.append(c)
        t = c.start_ms + 1000 + int(rng.integers(20, 400))  # the default 1000 ms line times have run out
        for op, name in ((m, "amp"), (c, "amp"), (c, "freq"), (c, "pan")):
            t1 = t + int(rng.integers(0, 300))
            updates.append((t1, v, op, {name: vb.Line(float(rng.uniform(0.2, 1.0)) * (200.0 if name == "freq" else 1.0))}))
            t2 = t1 + int(rng.integers(30, 500))
            shape = ["lin", "cos", "xpe", "sqe"][int(rng.integers(0, 4))]
            updates.append((t2, v, op, {name: vb.Line(0.0, goal=float(rng.uniform(0.1, 1.5)) * (300.0 if name == "freq" else 1.0),
                                                        shape=shape, state=False)}))
            if rng.integers(0, 2):  # and once more, after that ramp's inherited time is over
                t3 = t2 + 1000 + int(rng.integers(50, 300))
                updates.append((t3, v, op, {name: vb.Line(0.0, goal=float(rng.uniform(0.1, 1.0)) * (250.0 if name == "freq" else 1.0),
                                                            shape="lin", state=False)}))
    return vb.build_program(voices, updates=updates)
