"""Block buffers of the time-parallel path's running-sum numbering (saugns_amd/csrc/sau_dev_types.h: fast_slot_compact, the
lean form -- round 5): rows per pass are what LDS holds of them, so the count is performance; that the device copes when
the host's count is too small is tests/test_gpu_inmix.py's neighbour below (GPU). Host logic only: plans are compiled by the
engine over the sequential test executor and dumped with SAU_AMD_PLAN_DUMP."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import os, sys
sys.path.insert(0, {root!r} + "/tests"); sys.path.insert(0, {root!r})
import ctypes as C
import numpy as np
import saugns_amd as sa
import conftest
from saugns_amd import voicebank as vb
sa.lib(); sa.set_piluts(np.fromfile({root!r} + "/tests/golden/piluts_ref.f32", dtype="<f4").reshape(12, 2048))
sa.api.use_hooks(conftest.hooks_path())
import subprocess
subprocess.check_call(["make", "-s", "-C", {root!r} + "/tests/seqexec"])
seq = C.CDLL({root!r} + "/tests/seqexec/libseqexec.so")
seq.seq_backend_create.restype = C.c_void_p; seq.seq_backend_create.argtypes = [C.c_uint32]
{make}
b = sa.Batch([prg], 44100, backend=seq.seq_backend_create(44100))
b.run(512, stereo=False)
b.close()
'''

FM_VOICE = """
def fm(freq2):
    m3 = vb.Op("sin", freq=vb.Line(3.0, ratio=True), amp=vb._f32(0.4))
    m2 = vb.Op("sin", freq=freq2, amp=vb._f32(0.7), mods={vb.POP_PMOD: [m3]})
    m1 = vb.Op("sin", freq=vb.Line(1.0, ratio=True), amp=20.0, mods={vb.POP_PMOD: [m2]})
    return vb.Op("sin", freq=110.0, time_ms=1000, mods={vb.POP_FMOD: [m1]})
"""


def _dump(make, env=None):
    e = dict(os.environ, SAU_AMD_TUNE="1", SAU_AMD_PLAN_DUMP="1")
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", SNIPPET.format(root=ROOT, make=make)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _dump.freq_line_ids = [int(x) for x in re.findall(r"kind 1 flags \S+ op \d+ which 3 .*full ids: out (-?\d+)", r.stderr)]
    return [(int(a), int(b)) for a, b in re.findall(r"plan: voice \d+ steps \d+ n_fast (\d+) n_fast_full (\d+)", r.stderr)]


def test_a_carrier_fm_voice_needs_two_buffers():
    """carrier <- FM <- PM <- PM, every modulator frequency a fixed ratio: the carrier's frequency block and one for the chain
    (four before round 5: one per frequency line)"""
    make = FM_VOICE + "prg = vb.build_program([fm(vb.Line(2.0, ratio=True))])"
    assert _dump(make)[-1][1] == 2
    assert _dump(make, {"SAU_AMD_NO_LEAN_IDS": "1"})[-1][1] == 4


def test_a_modulator_whose_frequency_sweeps_keeps_its_buffer():
    """the frequency lines that have a step of their own, in plan order: the carrier's (added into: a buffer), m1's, m2's (the
    innermost modulator's is evaluated inside its oscillator step). Fixed ratios get no buffer; m2's gets one once an event has
    given it a sweep"""
    _dump(FM_VOICE + "prg = vb.build_program([fm(vb.Line(2.0, ratio=True))])")
    assert [i >= 0 for i in _dump.freq_line_ids] == [True, False, False], _dump.freq_line_ids
    _dump(FM_VOICE + "prg = vb.build_program([fm(vb.Line(2.0, goal=5.0, ratio=True))])")
    assert [i >= 0 for i in _dump.freq_line_ids] == [True, False, True], _dump.freq_line_ids


def test_config4s_second_voice_needs_three_buffers():
    """rainy_thunder's range-FM carrier with a range-modulated R rate and range AM: the carrier's frequency block, the R
    oscillator's rate block and one modulator output -- the range ends are one value each and folded into their blends
    (five before round 5)"""
    make = 'prg = conftest.load_program(sa, "config4_seed0")'
    got = _dump(make)
    assert got[-1] == (3, 3), got
    assert _dump(make, {"SAU_AMD_NO_LEAN_IDS": "1"})[-1][1] == 5
