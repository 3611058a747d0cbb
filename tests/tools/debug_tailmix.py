"""tailmix on / off on a small config-4 batch: where do the PCMs differ?"""
import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1:
    import saugns_amd as sa
    tabs = np.fromfile(os.path.join(ROOT, "tests/golden/piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
    sa.lib(); sa.set_piluts(tabs)
    fx = np.load(os.path.join(ROOT, "tests/golden/config4_seeds.npz"))
    prgs = [sa.Program.from_image(fx["images"][k].tobytes()) for k in range(8)]
    b = sa.Batch(prgs, 44100)
    pcm = b.run(88200, stereo=False)[0]
    np.save(sys.argv[1], np.asarray(pcm))
else:
    for name, env in (("on", {"SAU_AMD_TUNE": "1", "SAU_AMD_TAILMIX": "1"}), ("off", {})):
        subprocess.check_call([sys.executable, __file__, "/tmp/tm_%s.npy" % name], env=dict(os.environ, **env))
    a, b = np.load("/tmp/tm_on.npy"), np.load("/tmp/tm_off.npy")
    print("shapes", a.shape, b.shape)
    for s in range(a.shape[0]):
        d = np.nonzero(a[s] != b[s])[0]
        print("stream", s, "differing frames", len(d), "first", d[:12], "last", d[-4:] if len(d) else [])
        if len(d):
            i = d[0]
            print("   on ", a[s][i:i + 8], "\n   off", b[s][i:i + 8])
