"""Operators that run out of time inside a segment (generator.c:686-700): the segment is cut on a grid of frames after the
earliest such frame and the voice spends the frames in between in the block loop. What does the grid cost, what would cutting
at the frame itself cost? One process per setting (SAU_AMD_EXPIRY_GRID), every corpus script through a Batch in runs of 176400
frames: kernel time by kind, segments, wall time of the whole render (timing off)."""
import sys, os, json, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def child(grid):
    import numpy as np
    import saugns_amd as sa
    index = json.load(open(os.path.join(G, "index.json")))
    sa.set_piluts(np.fromfile(os.path.join(G, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048))
    rate = index["corpus_rate"]
    out = {}
    progs = {k: sa.Program.from_image(open(os.path.join(G, "programs", k + ".saup"), "rb").read()) for k in sorted(index["corpus"])}
    for rep in range(2):  # (the second pass is the one that counts: pools warm)
        for key, prg in progs.items():
            b = sa.Batch([prg], rate)
            t0 = time.perf_counter()
            pcm = b.render(stereo=True, chunk=176400)
            wall = time.perf_counter() - t0
            b.close()
            b = sa.Batch([prg], rate); b.set_timing(2); b.render(stereo=True, chunk=176400)
            t = b.timing_ex(); b.close()
            import hashlib
            out[key] = dict(wall_ms=wall * 1e3, block_ms=t["block_ms"], fast_ms=t["fast_ms"], mix_ms=t["mix_ms"], aux_ms=t["aux_ms"],
                            segments=t["segments"], sha=hashlib.sha256(np.ascontiguousarray(pcm[0]).tobytes()).hexdigest()[:16])
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    res = {}
    for grid in sys.argv[1:] or ["8192", "2048", "512", "1"]:
        env = dict(os.environ, SAU_AMD_TUNE="1", SAU_AMD_EXPIRY_GRID=grid)
        p = subprocess.run([sys.executable, __file__, "child", grid], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(grid, "failed", p.stderr[-2000:]); continue
        res[grid] = json.loads(line[0][7:])
    base = res.get("8192")
    summary = {}
    for grid, r in res.items():
        summary[grid] = dict(wall_ms=sum(x["wall_ms"] for x in r.values()), block_ms=sum(x["block_ms"] for x in r.values()),
                             fast_ms=sum(x["fast_ms"] for x in r.values()), kernels_ms=sum(x["block_ms"] + x["fast_ms"] + x["mix_ms"] + x["aux_ms"] for x in r.values()),
                             segments=sum(x["segments"] for x in r.values()), scripts_with_block_loop=sum(1 for x in r.values() if x["block_ms"] > 0.02),
                             same_pcm_as_8192=all(base and base[k]["sha"] == x["sha"] for k, x in r.items()))
        print(grid, summary[grid])
    if base:
        worst = sorted(base, key=lambda k: -base[k]["block_ms"])[:12]
        for k in worst:
            print(k, {g: (round(res[g][k]["wall_ms"], 2), round(res[g][k]["block_ms"], 2), res[g][k]["segments"]) for g in res})
    json.dump({"what": __doc__, "summary": summary, "per_script": res}, open(os.path.join(ROOT, "gpurun_out", "r06_expiry_grid.json"), "w"), indent=1)
