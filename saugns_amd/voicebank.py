"""Build ``sauProgram`` structures directly (no parser) for synthetic voice banks.

The structures are laid out exactly as the reference parser lays them out for
the equivalent script (one event per voice, operator data in post-order,
carrier last; line/time flag conventions of sau/parser.c) -- tests check the
builder against parser-made images of BASELINE configs 2, 3 and 5
(SURVEY.md section 8d gives the script text of each config).
"""
import ctypes as C
import math

import numpy as np

from . import api
from .api import (LINES, LP_GOAL, LP_STATE, LP_STATE_RATIO, LP_TIME, LP_TIME_IF_NEW, LP_TYPE,
                  POP_AMOD, POP_APMOD, POP_CAMOD, POP_CARR, POP_FMOD, POP_FPMOD, POP_PMOD,
                  POP_RAMOD, POP_RFMOD, POPT_WAVE, SauEvent, SauLine, SauOpData, SauProgram,
                  TIMEP_DEFAULT, TIMEP_IMPLICIT, TIMEP_SET, WAVES)

_LIST_FIELDS = {POP_CAMOD: "camods", POP_AMOD: "amods", POP_RAMOD: "ramods",
                POP_FMOD: "fmods", POP_RFMOD: "rfmods", POP_PMOD: "pmods",
                POP_APMOD: "apmods", POP_FPMOD: "fpmods"}


def cyclepos(x):
    """sau_cyclepos_dtoui32 (sau/math.h:70-72): cycle fraction -> u32 phase."""
    v = math.remainder(x, 1.0) * 2.0 ** 32
    r = np.rint(v)  # round-half-even like lrint in the default mode
    return int(r) & 0xFFFFFFFF


class Line:
    def __init__(self, v0, goal=None, shape="lin", ratio=False, state=True):
        self.v0, self.goal, self.shape, self.ratio, self.state = v0, goal, shape, ratio, state


class Op:
    """One operator of a voice tree (only what configs 1-5 style banks need)."""

    def __init__(self, wave="sin", freq=None, amp=1.0, time_ms=None, phase=0.0,
                 amp2=None, freq2=None, pm_a=None, mods=None, op_type=POPT_WAVE,
                 ras=None, noise=0, seed=0, pan=None):
        """W operator by default. op_type=POPT_RASEG: ``ras`` = (line shape name, function id,
        function flags) as sauRasOpt (sau/program.h:126-163); op_type=POPT_NOISE: ``noise`` = id."""
        self.ras, self.noise, self.seed = ras, noise, seed
        self.pan = pan if isinstance(pan, Line) or pan is None else Line(pan)  # carriers only
        self.wave = wave
        self.freq = freq if isinstance(freq, Line) or freq is None else Line(freq)
        self.amp = amp if isinstance(amp, Line) else Line(amp)
        self.amp2 = amp2 if isinstance(amp2, Line) or amp2 is None else Line(amp2)
        self.freq2 = freq2 if isinstance(freq2, Line) or freq2 is None else Line(freq2)
        self.pm_a = pm_a if isinstance(pm_a, Line) or pm_a is None else Line(pm_a)
        self.time_ms = time_ms
        self.phase = phase
        self.mods = mods or {}  # use type -> [Op]
        self.op_type = op_type


class BuiltProgram(api.Program):
    """A Program whose memory is owned by Python objects kept in ``_keep``."""

    def __init__(self, prg, keep):
        super().__init__(C.addressof(prg))
        self._keep = keep
        self._prg = prg


def _mk_line(keep, ln, time_ms, is_r_par=False):
    if ln is None:
        return None
    s = SauLine()
    s.v0 = ln.v0
    s.vt = ln.goal if ln.goal is not None else 0.0
    s.pos = 0
    s.end = 0
    s.time_ms = time_ms
    s.type = LINES.index(ln.shape)
    fl = LP_TIME | LP_TIME_IF_NEW
    if ln.state:
        fl |= LP_STATE
    if not is_r_par:
        fl |= LP_TYPE
    if ln.ratio:
        fl |= LP_STATE_RATIO
    if ln.goal is not None:
        fl |= LP_GOAL
    s.flags = fl
    keep.append(s)
    return C.pointer(s)


def build_program(voices, ampmult=1.0, default_mod_ms=1000, updates=(), amp_div_voices=True):
    """voices: list of carrier Ops (each with .time_ms) -> BuiltProgram.

    updates: later events in the shape the parser gives compound steps (``Wsin f200 t1; f300[g400]``):
    tuples (at_ms, voice index, Op of that voice's tree, {"freq"|"amp"|"amp2"|"freq2"|"pan"|"pm_a": Line,
    "time_ms": int, "wave": name (W) | "noise": index (N) | "ras": (line, func, flags) (R),
    "phase": cycles, "seed": int, "mods": {list use: [Ops already in the tree, possibly none]}});
    lines with ``state=False`` only set a goal. What an event may carry: generator.c:283-343."""
    keep = []
    updates = sorted(updates, key=lambda u: u[0])
    events = (SauEvent * (len(voices) + len(updates)))()
    next_id = [0]
    depth_max = [0]
    dur = 0

    def assign_ids(op):
        op._id = next_id[0]
        next_id[0] += 1
        for use in sorted(op.mods):
            for m in op.mods[use]:
                assign_ids(m)

    def emit(op, use, depth, out):
        depth_max[0] = max(depth_max[0], depth)
        for u in sorted(op.mods):
            for m in op.mods[u]:
                emit(m, u, depth + 1, out)
        od = SauOpData()
        od.id = op._id
        od.params = 0xF
        carrier = use == POP_CARR
        t_ms = op.time_ms if op.time_ms is not None else default_mod_ms
        od.time.v_ms = t_ms
        od.time.flags = TIMEP_SET if op.time_ms is not None else \
            (TIMEP_SET | TIMEP_DEFAULT | TIMEP_IMPLICIT)
        p = _mk_line(keep, (op.pan or Line(0.0)) if carrier else None, t_ms)
        if p: od.pan = p
        for name, ln, rp in (("amp", op.amp, False), ("amp2", op.amp2, True),
                             ("freq", op.freq, False), ("freq2", op.freq2, True),
                             ("pm_a", op.pm_a, False)):
            p = _mk_line(keep, ln, t_ms, rp)
            if p: setattr(od, name, p)
        od.phase = cyclepos(op.phase)
        od.seed = op.seed
        od.use_type = use
        od.type = op.op_type
        od.mode.ras.word = 0
        od.mode.ras.alpha = 0
        if op.op_type == api.POPT_RASEG:
            line, func, flags = op.ras
            # line | flags << 8 | func << 18 | level << 24, with the "set" bits of what is given
            od.mode.ras.word = LINES.index(line) | ((flags & 0x3f) | (1 << 6) | (1 << 7)) << 8 | (func & 0x3f) << 18
        elif op.op_type == api.POPT_NOISE:
            od.mode.main = op.noise
        else:
            od.mode.main = WAVES.index(op.wave)
        for u, lst in op.mods.items():
            arr = (C.c_uint32 * (1 + len(lst)))(len(lst), *[m._id for m in lst])
            keep.append(arr)
            setattr(od, _LIST_FIELDS[u], C.addressof(arr))
        out.append(od)

    for v, carr in enumerate(voices):
        assign_ids(carr)
    # events in time order: a voice may start later (`carrier.start_ms`, as a script's `/t` gives);
    # at equal times voices come first, then updates, each in the order given
    timeline = [(getattr(carr, "start_ms", 0) or 0, 0, v) for v, carr in enumerate(voices)] + \
               [(u[0], 1, k) for k, u in enumerate(updates)]
    timeline.sort()
    slot_of_voice = {v: i for i, (_, kind, v) in enumerate(timeline) if kind == 0}
    slot_of_update = {k: i for i, (_, kind, k) in enumerate(timeline) if kind == 1}
    time_of_slot = [t for t, _, _ in timeline]
    for v, carr in enumerate(voices):
        ods = []
        emit(carr, POP_CARR, 0, ods)
        arr = (SauOpData * len(ods))(*ods)
        keep.append(arr)
        i = slot_of_voice[v]
        ev = events[i]
        ev.wait_ms = time_of_slot[i] - (time_of_slot[i - 1] if i else 0)
        ev.vo_id = v
        ev.carr_op_id = carr._id
        ev.op_count = 0
        ev.op_data_count = len(ods)
        ev.op_list = None
        ev.op_data = arr
        dur = max(dur, time_of_slot[i] + carr.time_ms)
    use_of = {}

    def note_uses(op, use):
        use_of[id(op)] = use
        for u, lst in op.mods.items():
            for m in lst:
                note_uses(m, u)

    for carr in voices:
        note_uses(carr, POP_CARR)
    for k, (at_ms, vi, op, what) in enumerate(updates):
        od = SauOpData()
        od.id = op._id
        od.params = 0
        t_ms = what.get("time_ms")
        if t_ms is not None:
            od.params |= 1  # SAU_POPP_TIME
            od.time.v_ms = t_ms
            od.time.flags = TIMEP_SET
        line_ms = t_ms if t_ms is not None else (op.time_ms if op.time_ms is not None else default_mod_ms)
        for name, rp in (("amp", False), ("amp2", True), ("freq", False), ("freq2", True), ("pan", False),
                         ("pm_a", False)):
            if name in what:
                p = _mk_line(keep, what[name], line_ms, rp)
                setattr(od, name, p)
        od.mode.ras.word = 0
        od.mode.ras.alpha = 0
        if "wave" in what:
            od.params |= 2  # SAU_POPP_MODE
            od.mode.main = WAVES.index(what["wave"])
        if "noise" in what:
            od.params |= 2
            od.mode.main = what["noise"]
        if "ras" in what:
            od.params |= 2
            line, func, flags = what["ras"]
            od.mode.ras.word = LINES.index(line) | ((flags & 0x3f) | (1 << 6) | (1 << 7)) << 8 | (func & 0x3f) << 18
        if "phase" in what:
            od.params |= 4  # SAU_POPP_PHASE
            od.phase = cyclepos(what["phase"])
        if "seed" in what:
            od.params |= 8  # SAU_POPP_SEED
            od.seed = what["seed"]
        for u, lst in what.get("mods", {}).items():
            arr = (C.c_uint32 * (1 + len(lst)))(len(lst), *[m._id for m in lst])
            keep.append(arr)
            setattr(od, _LIST_FIELDS[u], C.addressof(arr))
        od.use_type = use_of[id(op)]
        od.type = op.op_type
        arr = (SauOpData * 1)(od)
        keep.append(arr)
        i = slot_of_update[k]
        ev = events[i]
        ev.wait_ms = time_of_slot[i] - (time_of_slot[i - 1] if i else 0)
        ev.vo_id = vi
        ev.carr_op_id = voices[vi]._id
        ev.op_count = 0
        ev.op_data_count = 1
        ev.op_list = None
        ev.op_data = arr
        if t_ms is not None and op is voices[vi]:
            dur = max(dur, at_ms + t_ms)
    prg = SauProgram()
    prg.events = events
    prg.ev_count = len(voices) + len(updates)
    prg.mode = api.PMODE_AMP_DIV_VOICES if amp_div_voices else 0
    prg.vo_count = len(voices)
    prg.op_count = next_id[0]
    prg.op_nest_depth = depth_max[0]
    prg.duration_ms = dur
    prg.ampmult = ampmult
    prg.name = b"voicebank"
    keep.append(events)
    return BuiltProgram(prg, keep)


def _f32(x):
    return float(np.float32(x))


def _num(fmt, x):
    """The value a script would carry: formatted, parsed back, stored as f32."""
    return _f32(float(format(x, fmt)))


# ---- BASELINE configs (script text in SURVEY.md section 8d) ---------------------

def config1():
    """`Wsin`: 1 voice, 440 Hz default, 1 s."""
    return build_program([Op("sin", freq=440.0, amp=1.0, time_ms=1000)])


def config2(n=256, seconds=10):
    """n x `Wsin f.. p.. t10`: flat batch, no modulation."""
    voices = []
    for i in range(n):
        f = _num(".4f", 55.0 * (1 + i % 64) + (i // 64) * 0.37)
        ph = float(format((i * 0.6180339887) % 1, ".6f"))
        voices.append(Op("sin", freq=f, time_ms=seconds * 1000, phase=ph))
    return build_program(voices)


def config3(n=1024, seconds=10, first=0):
    """n voices, each carrier + 3-deep PM chain (the headline workload); `first` shifts the
    voice indices (voices first .. first+n-1 of a larger bank)."""
    return build_program(config3_voices(n, seconds, first))


def config3_voices(n=1024, seconds=10, first=0):
    voices = []
    for i in range(first, first + n):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7),
                mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True),
                amp=_num(".2f", 0.5 + (i % 7) * 0.1), mods={POP_PMOD: [m2]})
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=seconds * 1000,
                         mods={POP_PMOD: [m1]}))
    return voices


def config3_fm(n=1024, seconds=10):
    """Config 3's voices with the carrier's modulator list an FM list (`f[...]` for `p[...]`, deviations of 20-50 Hz): the
    carrier's phase is a running sum of per-frame increments (wosc.h:135-169), its modulator keeps the depth-2 PM chain --
    the metric's "depth-3 FM" read literally. Script text: config_scripts()["fm"]."""
    voices = []
    for i in range(n):
        m3 = Op("sin", freq=Line(float(3 + i % 4), ratio=True), amp=_f32(0.4))
        m2 = Op("sin", freq=Line(float(2 + i % 3), ratio=True), amp=_f32(0.7), mods={POP_PMOD: [m3]})
        m1 = Op("sin", freq=Line(float(1 + i % 5), ratio=True), amp=_num(".1f", 20.0 + (i % 7) * 5.0), mods={POP_PMOD: [m2]})
        voices.append(Op("sin", freq=_num(".4f", 110.0 + i * 0.731), time_ms=seconds * 1000, mods={POP_FMOD: [m1]}))
    return build_program(voices)


def config5(n=4096, seconds=10):
    """n voices: self-feedback FM carrier with ramps + range-AM modulator."""
    return build_program(config5_voices(n, seconds))


def config5_voices(n=4096, seconds=10):
    voices = []
    for i in range(n):
        lfo = Op("sin", freq=float(3 + i % 9), amp=1.0)
        carr = Op("sin",
                  freq=Line(_num(".4f", 80.0 + i * 0.211), goal=_num(".3f", 160.0 + i * 0.1),
                            shape="exp"),
                  pm_a=Line(_num(".2f", 0.3 + (i % 8) * 0.1), goal=_f32(0.1), shape="lin"),
                  amp=Line(1.0, goal=_f32(0.2), shape="xpe"),
                  amp2=Line(_f32(0.2)),
                  time_ms=seconds * 1000, mods={POP_RAMOD: [lfo]})
        voices.append(carr)
    return voices


def config_scripts():
    """Script text of configs 2, 3, 5 and the carrier-FM bank (for the reference parser; fixtures only)."""
    c2 = "\n".join(f"Wsin f{55.0*(1+i%64)+(i//64)*0.37:.4f} p{(i*0.6180339887)%1:.6f} t10"
                   for i in range(256))
    c3 = "\n".join(f"Wsin f{110.0+i*0.731:.4f} t10 p[Wsin r{1+i%5} a{0.5+(i%7)*0.1:.2f} "
                   f"p[Wsin r{2+i%3} a0.7 p[Wsin r{3+i%4} a0.4]]]" for i in range(1024))
    c5 = "\n".join(f"Wsin f{80.0+i*0.211:.4f}[g{160.0+i*0.1:.3f} lexp] "
                   f"p.a{0.3+(i%8)*0.1:.2f}[g0.1 llin] a1[g0.2 lxpe].r0.2[Wsin f{3+i%9}] t10"
                   for i in range(4096))
    fm = "\n".join(f"Wsin f{110.0+i*0.731:.4f} t10 f[Wsin r{1+i%5} a{20.0+(i%7)*5.0:.1f} "
                   f"p[Wsin r{2+i%3} a0.7 p[Wsin r{3+i%4} a0.4]]]" for i in range(1024))
    return {"config2": c2, "config3": c3, "config5": c5, "fm": fm}


# ---- the same banks through the C ABI's builder (sauAmd_build_bank) --------------------------

def _line_desc(ln):
    d = api.LineDesc()
    if ln is not None:
        d.present, d.has_goal, d.ratio = 1, int(ln.goal is not None), int(bool(ln.ratio))
        d.shape, d.v0, d.goal = LINES.index(ln.shape), ln.v0, (ln.goal if ln.goal is not None else 0.0)
    return d


def flatten(voices):
    """Op trees -> (ctypes array of sauAmdOpDesc, count) for sauAmd_build_bank."""
    flat = []

    def visit(op, use, parent):
        me = len(flat)
        flat.append((op, use, parent))
        for u in sorted(op.mods):
            for m in op.mods[u]:
                visit(m, u, me)

    for carr in voices:
        visit(carr, POP_CARR, 0)
    arr = (api.OpDesc * len(flat))()
    for i, (op, use, parent) in enumerate(flat):
        d = arr[i]
        d.parent, d.use, d.type = parent, use, op.op_type
        if op.op_type == api.POPT_RASEG:
            line, func, flags = op.ras
            d.mode = LINES.index(line) | (flags & 0x3f) << 8 | (func & 0x3f) << 16
        elif op.op_type == api.POPT_NOISE:
            d.mode = op.noise
        else:
            d.mode = WAVES.index(op.wave)
        d.time_ms = op.time_ms or 0
        d.start_ms = getattr(op, "start_ms", 0) or 0
        d.phase, d.seed = cyclepos(op.phase), op.seed
        d.pan, d.amp, d.amp2 = _line_desc(op.pan), _line_desc(op.amp), _line_desc(op.amp2)
        d.freq, d.freq2, d.pm_a = _line_desc(op.freq), _line_desc(op.freq2), _line_desc(op.pm_a)
    return arr, len(flat)


def build_bank_c(voices, ampmult=1.0, default_mod_ms=1000):
    """build_program(voices) through sauAmd_build_bank (no later events)."""
    arr, n = flatten(voices)
    p = api.lib().sauAmd_build_bank(arr, n, ampmult, default_mod_ms)
    if not p:
        raise ValueError("sauAmd_build_bank rejected the description")
    return api.Program(p, free=api.lib().sauAmd_free_bank)
