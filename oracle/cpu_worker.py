"""oracle/cpu_worker.py -- TEST INFRASTRUCTURE ONLY: one CPU process of bench.py's multi-core
`cpu_baseline` leg. Renders voices [first, first+count) of BASELINE config 3 for `frames` frames
with the compiled reference (oracle/_ref) when present, else the restatement, and prints the
seconds the render took (start-up excluded).

    python oracle/cpu_worker.py <first> <count> <frames> <go-file>
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    first, count, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    go = sys.argv[4]
    import numpy as np
    from oracle import pyoracle as po
    from saugns_amd import voicebank
    prg = voicebank.config3(n=count, seconds=frames // 44100 + 2, first=first)
    if po.have_ref():
        po.ref()
        render = lambda: po.ref_render(prg.ptr, 44100, False, max_frames=frames, chunk=11289)
    else:
        tabs = np.fromfile(os.path.join(ROOT, "tests", "golden", "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
        po.oracle_use_tables(tabs)
        po.oracle().ora_set_fastmath_forms(1)
        render = lambda: po.oracle_render(prg.ptr, 44100, False, max_frames=frames, chunk=11289)
    print("ready", flush=True)
    while not os.path.exists(go):  # all workers start together
        time.sleep(0.005)
    t0 = time.perf_counter()
    render()
    print(time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
