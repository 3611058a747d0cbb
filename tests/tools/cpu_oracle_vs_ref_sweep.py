"""CPU: the restatement (oracle/sau_oracle.c, mode 2) against the compiled reference (oracle/_ref/libsau_ref.so) on the random
programs of the GPU sweeps -- the hop the device's parity rests on (GPU == oracle in the suite, GPU == reference in the sweeps).
    python tests/tools/cpu_oracle_vs_ref_sweep.py [first_seed [count [plain|extreme|batch [workers]]]]
`batch` draws the programs of tests/tools/gpu_vs_ref_batches.py (twelve per seed, every fourth seed extreme, every fifth
with calls of 1-11 frames); `extreme` also moves carriers' pan positions beyond [-1, 1]. Summary on stdout as JSON."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def programs(seed, mode):
    from saugns_amd import voicebank as vb
    import test_gpu_units as T
    if mode == "batch":
        rng = np.random.default_rng(300000 + seed)
        prgs = []
        for k in range(12):
            voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
            T._random_starts(rng, voices)
            ups = T._random_updates(rng, voices)
            if seed % 4 == 3:
                T._push_extremes(rng, voices)
            prgs.append(vb.build_program(voices, updates=ups))
        rate = int(rng.choice([44100, 44100, 48000, 96000, 8000]))
        stereo = bool(seed & 1)
        call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
        return [((seed, k), p, stereo, call, rate) for k, p in enumerate(prgs)]
    rng = np.random.default_rng(20000 + seed)
    voices = [T._random_voice(rng) for _ in range(int(rng.integers(1, 4)))]
    ups = ()
    if seed % 2:
        T._random_starts(rng, voices)
        ups = T._random_updates(rng, voices)
    rate_x = None
    if mode == "extreme":
        T._push_extremes(rng, voices)
        r2 = np.random.default_rng(777000 + seed)
        for v in voices:
            if r2.random() < 0.5:
                v.pan = vb.Line(float(r2.choice([-3.0, 0.0, 1.0, 7.5, 1e4, -1e-3])),
                                goal=float(r2.choice([-2.0, 0.5, 30.0])) if r2.random() < 0.5 else None, shape="lin")
        rate_x = int(rng.choice([1000, 3000, 11025, 44100, 192000, 384000]))
    prg = vb.build_program(voices, updates=ups)
    call = int(rng.integers(1, 12)) if seed % 5 == 4 else int(rng.integers(300, 12000))
    return [(seed, prg, bool(seed & 2), call, (rate_x or (44100 if seed % 3 else int(rng.choice([8000, 22050, 48000, 96000])))))]


def work(args):
    first, count, mode = args
    from oracle import pyoracle as po
    po.ref(); tabs = po.ref_piluts(); po.oracle_use_tables(tabs)
    po.oracle().ora_set_fastmath_forms(2)
    out = {"programs": 0, "identical": 0, "samples": 0, "differing": []}
    for seed in range(first, first + count):
        for key, prg, stereo, call, rate in programs(seed, mode):
            ref = po.ref_render(prg.ptr, rate, stereo, chunk=call)
            ora = po.oracle_render(prg.ptr, rate, stereo, chunk=call)
            same = len(ref) == len(ora) and bool((ref == ora).all())
            out["programs"] += 1; out["identical"] += same; out["samples"] += len(ref)
            if not same:
                n = min(len(ref), len(ora))
                d = np.nonzero(ref[:n] != ora[:n])[0]
                out["differing"].append({"seed": key, "rate": rate, "call": call, "stereo": stereo, "samples": int(len(d)),
                                         "first": int(d[0]) if len(d) else -1, "lens": [len(ref), len(ora)]})
    return out


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, (os.cpu_count() or 2) - 1)
    import multiprocessing as mp
    step = max(1, min(25, count // workers or 1))
    jobs = [(s, min(step, first + count - s), mode) for s in range(first, first + count, step)]
    t0 = time.time()
    S = {"mode": mode, "first_seed": first, "seeds": count, "programs": 0, "identical": 0, "samples": 0, "differing": []}
    with mp.get_context("spawn").Pool(workers) as pool:
        for r in pool.imap_unordered(work, jobs):
            for k in ("programs", "identical", "samples"):
                S[k] += r[k]
            S["differing"] += r["differing"]
            for d in r["differing"]:
                print("DIFFERS", d, flush=True)
    S["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(S))
    sys.exit(0 if S["identical"] == S["programs"] else 1)
