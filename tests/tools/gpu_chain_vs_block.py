"""A few feedback voices (BASELINE config 5's) through the chain kernels and through the block loop (SAU_AMD_NO_CHAIN): ns per
frame of each, and two corpus scripts whose single voice modulates its own phase by a modulated amount (block loop today).
    python tests/tools/gpu_chain_vs_block.py"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import saugns_amd as sa
    from saugns_amd import voicebank as vb
    tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
    sa.set_piluts(tabs)
    G = os.path.join(ROOT, "tests", "golden")
    out = {}
    for n in (1, 8, 64):
        prg = vb.config5(n=n, seconds=10)
        sa.Batch([prg], 44100).render(stereo=False, chunk=441000)
        b = sa.Batch([prg], 44100); b.set_timing(2)
        t0 = time.perf_counter(); b.render(stereo=False, chunk=441000); dt = time.perf_counter() - t0
        out[f"config5 x{n}"] = dict(ns_per_frame=dt / 441000 * 1e9, **{k: round(v, 2) for k, v in b.timing_ex().items()})
    index = json.load(open(os.path.join(G, "index.json")))
    for key in ("examples__sounds__bass-sounds", "examples__sounds__pm_feedback_pm"):
        prg = sa.Program.from_image(open(os.path.join(G, "programs", key + ".saup"), "rb").read())
        sa.Batch([prg], index["corpus_rate"]).render(stereo=True, chunk=200000)
        b = sa.Batch([prg], index["corpus_rate"]); b.set_timing(2)
        t0 = time.perf_counter(); pcm = b.render(stereo=True, chunk=200000)[0]; dt = time.perf_counter() - t0
        out[key] = dict(ns_per_frame=dt / (len(pcm) // 2) * 1e9, frames=len(pcm) // 2, **{k: round(v, 2) for k, v in b.timing_ex().items()})
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(); sys.exit(0)
    for label, env in (("chains", {}), ("block loop", {"SAU_AMD_TUNE": "1", "SAU_AMD_NO_CHAIN": "1"})):
        p = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print("==", label)
        if not line:
            print(p.stderr[-1500:]); continue
        for k, v in json.loads(line[0][7:]).items():
            print("  ", k, v)
