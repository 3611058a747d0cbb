"""GPU against the compiled reference (oracle/_ref/libsau_ref.so) on random programs beyond the suite's 112:
    python tests/tools/gpu_vs_ref_sweep.py [first_seed [count [dropin]]]
Random operator graphs (every modulator list, ramps, R / N / A operators, feedback) with later events and random
start times, one to three voices, random call sizes, mono and stereo, 44.1 kHz and (every third) 8 / 22.05 / 48 / 96 kHz; the product's default (the reference build's
loop tails reproduced). Every render must equal the reference's bit for bit; the summary goes to
gpurun_out/gpu_vs_ref_sweep.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from saugns_amd import voicebank as vb
from oracle import pyoracle as po
import test_gpu_units as T

os.environ["SAU_AMD_LOOP_TAILS"] = "1"
if not po.have_ref():
    sys.exit("oracle/_ref/libsau_ref.so is not here (built from /root/reference by oracle/Makefile)")
po.ref()
tabs = po.ref_piluts()  # the tables this very reference library built (glibc picks its sin() by CPU)
sa.set_piluts(tabs)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tall = len(sys.argv) > 3 and sys.argv[3] == "tall"  # operator trees up to 256 levels deep, the list kind drawn per level (wide plans)
extreme = len(sys.argv) > 3 and sys.argv[3] == "extreme"  # the same graphs with parameters pushed to extremes
corpus = len(sys.argv) > 3 and sys.argv[3] == "corpus"  # the reference's 95 scripts (program images), `count` passes over them
varied = len(sys.argv) > 3 and sys.argv[3] == "varied"  # round 5: the host changes the size and the channel layout of its calls as it goes (the drop-in generator's read-ahead is taken back and re-issued: capi.cpp generator_rewind)
dropin = corpus or varied or len(sys.argv) > 3 and sys.argv[3] == "dropin"  # through sau_create_Generator / sauGenerator_run (read-ahead runs) instead of the batch API
S = {"programs": 0, "identical": 0, "samples": 0, "samples_differing": 0, "max_abs_diff": 0, "differing": [],
     "first_seed": first, "loop_tails": True, "api": "drop-in generator, call sizes and channel layouts changing in mid-stream" if varied else "drop-in generator" if dropin else "batch", "programs_are": "the 95 corpus scripts" if corpus else "random graphs, extreme parameters" if extreme else "trees up to 256 levels deep" if tall else "random graphs"}
t0 = time.time()
_push = T._push_extremes
_tall = T._tall_tree
def cases():
    if corpus:
        index = json.load(open(os.path.join(ROOT, "tests", "golden", "index.json")))
        keys = sorted(index["corpus"])
        progs = {k: sa.Program.from_image(open(os.path.join(ROOT, "tests", "golden", "programs", k + ".saup"), "rb").read()) for k in keys}
        for p_ in range(first, first + count):
            rng = np.random.default_rng(70000 + p_)
            for k in keys:
                yield (p_, k), progs[k], bool(rng.integers(2)), int(rng.integers(300, 20000)), int(rng.choice([22050, 44100, 44100, 48000, 96000]))
        return
    for seed in range(first, first + count):
        rng = np.random.default_rng(20000 + seed)
        voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
        if tall:
            voices = [_tall(rng) for _ in range(int(rng.integers(1, 3)))]
        ups = ()
        if seed % 2:
            T._random_starts(rng, voices)
            ups = T._random_updates(rng, voices)
        rate_x = None
        if extreme:
            _push(rng, voices)
            if os.environ.get("SWEEP_PAN"):  # carriers' pan positions beyond [-1, 1] too (a generator of its own: the programs stay what they were otherwise)
                r2 = np.random.default_rng(777000 + seed)
                for v in voices:
                    if r2.random() < 0.5:
                        v.pan = vb.Line(float(r2.choice([-3.0, 0.0, 1.0, 7.5, 1e4, -1e-3])), goal=float(r2.choice([-2.0, 0.5, 30.0])) if r2.random() < 0.5 else None, shape="lin")
            rate_x = int(rng.choice([1000, 3000, 11025, 44100, 192000, 384000]))
        if os.environ.get("SWEEP_AMP"):  # the program's amplitude multiplier and its division by the voice count (sauProgram.ampmult, mode)
            r3 = np.random.default_rng(778000 + seed)
            prg = vb.build_program(voices, updates=ups, ampmult=float(r3.choice([1.0, 0.25, 3.0, 1e-3, 50.0])), amp_div_voices=bool(r3.integers(2)))
        else:
            prg = vb.build_program(voices, updates=ups)
        # every fifth program with a host call of a few frames: where the reference's blocks end -- and with them the loop
        # tails of `cub` -- then falls on almost every sample (round 3: the R-segment map's tails were missing from the
        # time-parallel build and showed at such call sizes only)
        call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
        yield seed, prg, bool(seed & 2), call, (rate_x or (44100 if seed % 3 else int(rng.choice([8000, 22050, 48000, 96000]))))
def render_calls(create, run, destroy, prg, rate, calls):
    import ctypes as C
    g = create(prg, rate)
    n, out, k = C.c_size_t(), [], 0
    while True:
        size, st = calls[min(k, len(calls) - 1)]
        k += 1
        buf = np.zeros(size * (2 if st else 1), np.int16)
        more = run(g, buf.ctypes.data, size, st, C.byref(n))
        out.append(buf[: n.value * (2 if st else 1)].copy())
        if not more:
            break
    destroy(g)
    return np.concatenate(out)


for seed, prg, stereo, chunk, rate in cases():
    if varied:
        r4 = np.random.default_rng(555000 + (seed if not isinstance(seed, tuple) else seed[0]))
        calls, size, st = [], chunk, stereo
        for _ in range(int(r4.integers(3, 9))):
            calls += [(size, st)] * int(r4.integers(1, 7))
            what = int(r4.integers(3))
            if what != 1:
                size = int([r4.integers(1, 30), r4.integers(30, 3000), r4.integers(3000, 40000)][int(r4.integers(3))])
            if what != 0:
                st = not st
        L, Rl = sa.lib(), po.ref()
        ref = render_calls(Rl.sau_create_Generator, Rl.sauGenerator_run, Rl.sau_destroy_Generator, prg.ptr, rate, calls)
        gpu = render_calls(L.sau_create_Generator, L.sauGenerator_run, L.sau_destroy_Generator, prg.ptr, rate, calls)
    else:
        ref = po.ref_render(prg.ptr, rate, stereo, chunk=chunk)
    if varied:
        pass
    elif dropin:
        g = sa.Generator(prg, rate)
        gpu = g.render(stereo=stereo, chunk=chunk)
        g.close()
    else:
        b = sa.Batch([prg], rate)
        if chunk < 300:  # engine runs of many such calls
            b.set_call_len(chunk)
            gpu = b.render(stereo=stereo, chunk=chunk * 1500)[0]
        else:
            gpu = b.render(stereo=stereo, chunk=chunk)[0]
    same = len(gpu) == len(ref) and bool((gpu == ref).all())
    S["programs"] += 1; S["identical"] += same; S["samples"] += len(ref)
    if not same:
        n = min(len(gpu), len(ref))
        d = np.abs(gpu[:n].astype(np.int32) - ref[:n].astype(np.int32))
        S["samples_differing"] += int((d > 0).sum()) + abs(len(gpu) - len(ref))
        S["max_abs_diff"] = max(S["max_abs_diff"], int(d.max()) if n else 0)
        S["differing"].append({"seed": list(seed) if isinstance(seed, tuple) else seed, "rate": rate, "call_size": chunk, "stereo": stereo, "lengths": [len(gpu), len(ref)]})
        print("seed", seed, "DIFFERS", S["differing"][-1], flush=True)
S["seconds"] = round(time.time() - t0, 1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(S, open(os.path.join(ROOT, "gpurun_out", "gpu_vs_ref_sweep%s.json" % ("_corpus" if corpus else "_tall" if tall else "_extreme" if extreme else "_varied" if varied else "_dropin" if dropin else "")) if not os.environ.get("SWEEP_OUT") else os.environ["SWEEP_OUT"], "w"), indent=1)
print(json.dumps({k: v for k, v in S.items() if k != "differing"}))
sys.exit(0 if S["identical"] == S["programs"] else 1)
