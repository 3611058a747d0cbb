"""GPU against the compiled reference, many programs per batch (sauAmd_create_Batch): twelve random programs with events side by
side in one engine -- every program's events cut the others' segments, never their spans of the reference's block lattice --
each compared with the reference's render of that program alone at the same call size.
    python tests/tools/gpu_vs_ref_batches.py [first_seed [batches]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T
os.environ["SAU_AMD_LOOP_TAILS"] = "1"
po.ref(); tabs = po.ref_piluts(); sa.set_piluts(tabs)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S = {"batches": 0, "programs": 0, "identical": 0, "samples": 0, "differing": []}
t0 = time.time()
for bseed in range(first, first + count):
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
    b = sa.Batch(prgs, rate)
    b.set_call_len(call)
    outs = b.render(stereo=stereo, chunk=call * (1500 if call < 300 else int(rng.integers(1, 6))))
    S["batches"] += 1
    for k, (p, got) in enumerate(zip(prgs, outs)):
        ref = po.ref_render(p.ptr, rate, stereo, chunk=call)
        # (a batch renders until its longest program ends: shorter programs are followed by silence)
        same = len(got) >= len(ref) and bool((got[:len(ref)] == ref).all()) and not got[len(ref):].any()
        S["programs"] += 1; S["identical"] += same; S["samples"] += len(ref)
        if not same:
            S["differing"].append({"batch": bseed, "program": k, "rate": rate, "call": call, "stereo": stereo})
            print("DIFFERS", S["differing"][-1], flush=True)
S["seconds"] = round(time.time() - t0, 1)
json.dump(S, open(os.path.join(ROOT, "gpurun_out", "gpu_vs_ref_batches.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in S.items() if k != "differing"}))
sys.exit(0 if S["identical"] == S["programs"] else 1)
