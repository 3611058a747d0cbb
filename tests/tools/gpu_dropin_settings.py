"""The drop-in generator's wall time on BASELINE config 3 (create -> 11289-frame calls -> destroy) under its read-ahead settings:
    python tests/tools/gpu_dropin_settings.py            (runs itself once per setting: the settings are read at creation)"""
import os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import numpy as np
    import saugns_amd as sa
    from saugns_amd import voicebank
    tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
    sa.lib(); sa.set_piluts(tabs)
    prg = voicebank.config3()
    buf = np.zeros(11289, np.int16)
    best = []
    for rep in range(8):
        t0 = time.perf_counter()
        g = sa.Generator(prg, 44100)
        t1 = time.perf_counter()
        more, first = True, None
        while more:
            more, n = g.run(buf, 11289, False)
            if first is None:
                first = time.perf_counter()
        t2 = time.perf_counter()
        g.close()
        t3 = time.perf_counter()
        if rep >= 2:
            best.append(((t3 - t0) * 1e3, (t1 - t0) * 1e3, (first - t1) * 1e3, (t2 - first) * 1e3, (t3 - t2) * 1e3))
    b = min(best)
    print("total %.3f ms (create %.3f, first call %.3f, other calls %.3f, destroy %.3f) -> %.3g frames/s" % (b + (441000 / b[0] * 1e3,)))
else:
    for env in ({}, {"SAU_AMD_READAHEAD": "338670"}, {"SAU_AMD_READAHEAD": "451560"}, {"SAU_AMD_READAHEAD_GROW": "3"}, {"SAU_AMD_READAHEAD_GROW": "4"},
                {"SAU_AMD_READAHEAD_GROW": "6"}, {"SAU_AMD_TUNE": "1", "SAU_AMD_NO_SNAPSHOT": "1"}, {"SAU_AMD_READAHEAD_DEPTH": "1"}, {"SAU_AMD_READAHEAD_RAMP": "0"},
                {"SAU_AMD_READAHEAD_RAMP": "0", "SAU_AMD_READAHEAD": "451560"}):
        r = subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        print(env, r.stdout.strip() or r.stderr[-300:])
