"""Debug aid (GPU): programs of one batch of tests/tools/gpu_vs_ref_batches.py rendered in subsets and under the tuning switches,
each compared with the compiled reference:  python tests/tools/debug_batch_pair.py <batch seed> <program> [<other program>]"""
import json, os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(2)
bseed = int(sys.argv[1]); pk = int(sys.argv[2]); other = int(sys.argv[3]) if len(sys.argv) > 3 else None
rng = np.random.default_rng(300000 + bseed)
prgs = []
for k in range(12):
    voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    T._random_starts(rng, voices)
    ups = T._random_updates(rng, voices)
    if bseed % 4 == 3:
        T._push_extremes(rng, voices)
    prgs.append(vb.build_program(voices, updates=ups))
rate = int(rng.choice([44100, 44100, 48000, 96000, 8000]))
stereo = bool(bseed & 1)
call = int(rng.integers(1, 12)) if bseed % 5 == 4 else int(rng.integers(300, 12000))
chunk = call * (1500 if call < 300 else int(rng.integers(1, 6)))
print("rate", rate, "stereo", stereo, "call", call, "chunk", chunk, flush=True)
ref = po.ref_render(prgs[pk].ptr, rate, stereo, chunk=call)
ora = po.oracle_render(prgs[pk].ptr, rate, stereo, chunk=call)
print("oracle == reference:", len(ora) == len(ref) and bool((ora == ref).all()), "frames", len(ref) // (2 if stereo else 1), flush=True)

def run(idx, env=None, chunk_=None, tag=""):
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k); os.environ[k] = v
    try:
        b = sa.Batch([prgs[i] for i in idx], rate)
        b.set_call_len(call)
        outs = b.render(stereo=stereo, chunk=chunk_ or chunk)
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    got = outs[idx.index(pk)]
    n = min(len(got), len(ref))
    d = np.nonzero(got[:n] != ref[:n])[0]
    ok = len(got) >= len(ref) and len(d) == 0
    desc = "identical" if ok else "DIFFERS at %s got %s ref %s" % (list(d[:8]), list(got[d[:8]]), list(ref[d[:8]]))
    print("%-60s %s" % ("programs %s %s %s" % (idx, env or "", tag), desc), flush=True)
    return ok

run([pk])
run(list(range(12)))
if other is None:
    for o in range(12):
        if o != pk: run(sorted([pk, o]))
else:
    pair = sorted([pk, other])
    run(pair)
    run([other, pk] if pk < other else [pk, other], tag="(other order)")
    for c in (call, 2 * call, 3 * call, 4 * call, 5 * call, 20 * call):
        run(pair, chunk_=c, tag="chunk %d" % c)
    for e in ("SAU_AMD_NO_SEQ", "SAU_AMD_NO_FAST", "SAU_AMD_NO_CHAIN", "SAU_AMD_NO_EARLY_CHAINS", "SAU_AMD_NO_LOOKBACK", "SAU_AMD_NO_TWO_PASS",
              "SAU_AMD_NO_INC_ROWS", "SAU_AMD_NO_LEAN", "SAU_AMD_NO_DYN", "SAU_AMD_NO_REPAIR", "SAU_AMD_LOOK_NO_LDS", "SAU_AMD_NO_PLAN_CACHE",
              "SAU_AMD_NO_CHAIN_INLINE"):
        run(pair, {e: "1"})
    for r in ("8", "6", "5", "4", "2"):
        run(pair, {"SAU_AMD_FAST_ROWS": r})
        run([pk], {"SAU_AMD_FAST_ROWS": r})
    for l in ("65536", "98304", "131072"):
        run(pair, {"SAU_AMD_LDS_LIMIT": l})
        run([pk], {"SAU_AMD_LDS_LIMIT": l})
    run([pk], {"SAU_AMD_NO_LOOKBACK": "1"})
    run([pk], {"SAU_AMD_NO_TWO_PASS": "1"})
    run([pk], {"SAU_AMD_CHAIN_CHUNKS": "1"})
    run(pair, {"SAU_AMD_CHAIN_CHUNKS": "1"})
    run(pair, {"SAU_AMD_CHAIN_CHUNK_FRAMES": "4096"})
    print("---- debug dump of the pair", flush=True)
    sys.stderr.flush()
    run(pair, {"SAU_AMD_DEBUG": "1"})
