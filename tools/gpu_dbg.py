import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import saugns_amd as sa
from oracle import pyoracle as po
from conftest import load_program, GOLDEN
tabs = np.fromfile(os.path.join(GOLDEN, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
sa.set_piluts(tabs); po.oracle_use_tables(tabs); po.oracle().ora_set_fastmath_forms(1)
key, nframes = sys.argv[1], int(sys.argv[2])
prg = load_program(sa, key)
want = po.oracle_render(prg.ptr, 12000, True)
got = sa.Generator(prg, 12000).render(stereo=True, chunk=nframes, max_frames=nframes)
print("got ", got[:24].tolist()); print("want", want[:24].tolist())
