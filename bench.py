#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the generator hot path over one batch: every rank
renders `--frames` frames (default 441000 = the whole 10 s of the script at 44.1 kHz)
of BASELINE config 3 (1024 voices, each carrier + 3-deep PM chain; 4096 operators) with
program state, block buffers and PCM resident in HBM.  Ranks hold independent
voice banks (the path has no exchange step: SURVEY.md 8e), so scaling is weak
and `value` is the sum over ranks of mixed mono output frames per second.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(tabs, voices):
    """The reference's own generator (oracle/_ref, built from its sources by oracle/Makefile)
    when that library is present, else this repo's CPU restatement; bounded sample."""
    from oracle import pyoracle as po
    from saugns_amd import voicebank
    prg = voicebank.config3(n=voices, seconds=30)
    if po.have_ref():
        kind = "reference"
        po.ref()

        def render(frames):
            return po.ref_render(prg.ptr, 44100, False, max_frames=frames, chunk=11289)
    else:
        kind = "port"
        po.oracle_use_tables(tabs)
        po.oracle().ora_set_fastmath_forms(1)

        def render(frames):
            return po.oracle_render(prg.ptr, 44100, False, max_frames=frames, chunk=11289)
    # calibrate, then run ~15 s of CPU work
    frames = 11025
    t0 = time.perf_counter()
    render(frames)
    dt = time.perf_counter() - t0
    frames = int(min(44100 * 20, max(frames, frames * 15.0 / max(dt, 1e-3))))
    t0 = time.perf_counter()
    render(frames)
    dt = time.perf_counter() - t0
    multi = cpu_baseline_all_cores(voices, frames_1core=frames / dt)
    return {"value": frames / dt, "unit": "mixed mono int16 frames/s", "cores": 1,
            "kind": kind, "all_cores": multi,
            "sample": f"config 3 ({voices} voices x depth-3 PM), first {frames} frames "
                      f"({frames * voices * 4:.3g} operator-samples), {dt:.1f} s on 1 host thread"
                      + (" (sauGenerator_run of the compiled reference, -O3 -ffast-math as its Makefile)"
                         if kind == "reference" else " (oracle/sau_oracle.c)"),
            "operator_samples_per_s": frames * voices * 4 / dt}


def cpu_baseline_all_cores(voices, frames_1core):
    """the same bank with its voices partitioned over all host cores, one process
    each (the reference is single-threaded; its voices only meet in the final per-frame sum)."""
    import subprocess
    import tempfile
    n = max(1, min(os.cpu_count() or 1, voices))
    per = [voices // n + (1 if i < voices % n else 0) for i in range(n)]
    # about 8 s per worker at the single-core rate measured above
    frames = int(min(44100 * 60, max(2205, 8.0 * frames_1core * voices / max(per))))
    go = os.path.join(tempfile.mkdtemp(prefix="sau_bench_"), "go")
    procs, first = [], 0
    for c in per:
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"),
                                       str(first), str(c), str(frames), go],
                                      stdout=subprocess.PIPE, text=True))
        first += c
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("worker failed to start")
        open(go, "w").close()
        secs = [float(p.stdout.readline()) for p in procs]
    except (RuntimeError, ValueError) as e:
        for p in procs:
            p.kill()
        return {"error": str(e)}
    finally:
        for p in procs:
            p.wait()
        try:
            os.remove(go)
            os.rmdir(os.path.dirname(go))
        except OSError:
            pass
    return {"value": frames / max(secs), "unit": "mixed mono int16 frames/s", "cores": n,
            "sample": f"same bank, voices partitioned over {n} processes ({min(per)}-{max(per)} voices each), "
                      f"first {frames} frames, slowest process {max(secs):.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=441000)
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU; SAU_BENCH_BACKEND=gloo lets the N>1 logic be exercised on a box with
    # fewer GPUs than ranks (ranks then share devices; rendezvous and reductions on the CPU)
    backend = os.environ.get("SAU_BENCH_BACKEND", "nccl")
    dev = local_rank % max(1, torch.cuda.device_count())
    os.environ.setdefault("SAU_AMD_DEVICE", str(dev))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    import saugns_amd as sa
    from saugns_amd import voicebank
    tabs = np.fromfile(os.path.join(ROOT, "tests", "golden", "piluts_ref.f32"),
                       dtype="<f4").reshape(12, 2048)
    sa.set_piluts(tabs)

    seconds = max(1, (args.frames * (args.steps + args.warmup + 1)) // 44100 + 2)
    prg = voicebank.config3(n=args.voices, seconds=seconds)
    batch = sa.Batch([prg], 44100)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        batch.sync()

    for _ in range(args.warmup):
        batch.run(args.frames, stereo=False, fetch=False)
    barrier()
    batch.timing_ex(reset=True)
    batch.set_timing(1)  # HIP events around the dominant kernel only, on its own stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(args.frames, stereo=False, fetch=False)
    barrier()
    dt = time.perf_counter() - t0
    tm = batch.timing_ex()
    if world > 1:
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_total = args.frames * args.steps * world
    value = frames_total / dt
    # the only exchange of the whole job is this after-the-fact reduction of
    # {frames rendered, PCM checksum} for the scaling report (outside the timed region)
    pcm = batch.run(args.frames, stereo=False)[0]
    tally = [args.frames * args.steps, int(np.asarray(pcm, dtype=np.int64).sum() & 0x7FFFFFFF)]
    mine = list(tally)
    if world > 1:
        t = torch.tensor(tally, device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tally = [int(x) for x in t.tolist()]
    if tally[0] != frames_total or tally[1] != mine[1] * world:
        raise SystemExit(f"rank {rank}: ranks disagree on the rendered PCM ({tally} vs {mine} x {world})")
    if rank == 0:
        n_ops = args.voices * 4
        # 8 B per operator-sample (one f32 write + one f32 read of every
        # operator's block output) + 2 B per output frame, per launch of the kernel
        alg_bytes = (n_ops * 8 + 2) * args.frames
        launch_s = (tm["fast_ms"] / 1e3) / max(1, tm["segments"])
        achieved = alg_bytes / launch_s / 1e9 if launch_s > 0 else 0.0
        # HBM bytes per launch of the same kernel on the same workload, from the committed
        # PMC passes (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); counters
        # cannot be collected from inside this process, so null when the workload differs
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_m_pmc_summary.json")))
            wl = pmc["workload"]
            if wl["voices"] == args.voices and wl["frames_per_step"] == args.frames:
                k = [v for n, v in pmc["kernels"].items() if n.startswith("sauhip::fast_kernel<")]
                traffic = k[0]["hbm_bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError, IndexError):
            pass
        out = {
            "metric": "mono samples/sec/GPU @ N voices (depth-3 FM)",
            "value": value, "unit": "mixed mono int16 frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f64 table interpolation, u32 phase)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 3: {args.voices} voices x (carrier + 3-deep "
                                   f"PM chain) = {n_ops} operators, 44.1 kHz mono, "
                                   f"{args.frames} frames per step, per GPU",
                       "voices": args.voices, "operators": n_ops,
                       "frames_per_step": args.frames,
                       "frames_all_ranks": tally[0], "pcm_checksum_all_ranks": tally[1],
                       "voice_samples_per_s": value * args.voices,
                       "operator_samples_per_s": value * n_ops},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "fast_kernel<8, false>", "avg_launch_ms": launch_s * 1e3,
                         "launches": tm["segments"],
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(tabs, args.voices)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
