#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload config3|config4|config5]

`--gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment) starts the N ranks
itself: N fresh child processes of this script, one per GPU, before this process has touched the GPU;
rank 0's JSON line is relayed. Under `python -m torch.distributed.run --nproc-per-node N` the ranks
are the launcher's. Every rank checks that the world it finds is the `--gpus` it was given.

The default (config 3) line also carries `other_workloads`: short runs of config 5 and config 4 in
the same process, outside config 3's timed region, each with its own value / ms_per_step / roofline /
cpu_baseline and the same SHA-256 checks (`--no-others` leaves them out, e.g. under rocprofv3).

One step = one pass of the generator hot path over one batch, program state, block buffers and PCM
resident in HBM:

* config3 (default, the configuration the metric is quoted on): every rank renders `--frames`
  frames (default 441000 = the script's whole 10 s at 44.1 kHz) of BASELINE config 3 -- 1024
  voices, each a carrier with a 3-deep PM chain, 4096 operators. Ranks hold independent voice
  banks (the path has no exchange step): weak scaling, `value` = mixed mono output
  frames per second summed over ranks. The first step's PCM is checked against the reference's
  SHA-256 (tests/golden/index.json) before anything is timed.
* config4 (the north star's multi-GPU case): examples/rainy_thunder.sau with seed = k, k = 0..511
  (tests/golden/config4_seeds.npz), 64 renders per GPU: rank r renders seeds
  shard_range(64 * N, r, N) as one batch; a step = those 64 scripts from generator creation to
  their last frame (60 s each). No data-path collective; the ranks all-reduce {frames, checksum}
  afterwards, and every render's SHA-256 is compared with the reference's.
* config5: 4096 voices with self-feedback FM + range AM + ramps (the feedback-recurrence stress);
  a step = the script's whole 10 s from generator creation on; the PCM's SHA-256 is checked.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def kernel_source_hash():
    """Identity of the kernels a PMC summary under profiles/ was collected on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "saugns_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(("k_", "sau_dev_")) and f.endswith(".h"):  # device code: the kernels' parts and the arithmetic they share
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_traffic(workload, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    command (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; counters cannot be read
    from inside the process). Only a summary collected on these very kernel sources counts."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)):
        if not f.endswith("_pmc_summary.json"):
            continue
        try:
            pmc = json.load(open(os.path.join(pdir, f)))
            if pmc.get("kernel_source_sha") != kernel_source_hash():
                continue
            if pmc.get("workload", {}).get("name") != workload:
                continue
            # (the dominant launch of the prefix: config 3's step has two of fast_kernel<12, ...> since round 6, edge groups and the rest)
            k = sorted((v for n, v in pmc["kernels"].items() if kernel_prefix in n),
                       key=lambda v: -v.get("hbm_bytes_per_launch_corrected", 0.0))
            # every kernel of a step is launched once per step (one segment per step): their sum is the step's HBM bytes
            step_bytes = sum(v.get("hbm_bytes_per_launch_corrected", 0.0) for v in pmc["kernels"].values())
            best = (k[0]["hbm_bytes_per_launch_corrected"], "profiles/" + f, k[0].get("valu"), step_bytes)
        except (OSError, KeyError, ValueError, IndexError):
            pass
    return best if best else (None, None, None, None)


def profile_step_traffic(workload):
    """HBM bytes per step of one of the other workloads (every kernel of the step, FETCH_SIZE doubled + WRITE_SIZE),
    from the committed PMC passes of that command -- only from a summary collected on these very kernel sources."""
    pdir = os.path.join(ROOT, "profiles")
    best = (None, None)
    for f in sorted(os.listdir(pdir)):
        if not f.endswith("_pmc_summary.json"):
            continue
        try:
            pmc = json.load(open(os.path.join(pdir, f)))
            if pmc.get("kernel_source_sha") == kernel_source_hash() and pmc.get("workload", {}).get("name") == workload \
                    and pmc.get("hbm_bytes_per_step_corrected"):
                best = (pmc["hbm_bytes_per_step_corrected"], "profiles/" + f)
        except (OSError, KeyError, ValueError, AttributeError):
            pass
    return best


def profile_step_valu(workload, kernel_prefix):
    """The VALU side (tools/collect_profile.py: valu_side) of one kernel of another workload's step, from the hash-matched
    PMC summary of that command, or None."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for f in sorted(os.listdir(pdir)):
        if not f.endswith("_pmc_summary.json"):
            continue
        try:
            pmc = json.load(open(os.path.join(pdir, f)))
            if pmc.get("kernel_source_sha") == kernel_source_hash() and pmc.get("workload", {}).get("name") == workload:
                for n, v in pmc["kernels"].items():
                    # (the step's dominant launch: token launches of other builds match the prefix too)
                    if kernel_prefix in n and v.get("valu") and (best is None or best["file"] != f or
                                                                  v["valu"].get("weighted_cycles_per_launch", 0) > best.get("weighted_cycles_per_launch", 0)):
                        best = dict(v["valu"], kernel=n, source="profiles/" + f, file=f)
        except (OSError, KeyError, ValueError, AttributeError):
            pass
    return best


def sha_checked(rec):
    """Did this run compare a step's PCM with the compiled reference's SHA-256 (and pass: a mismatch aborts the run)?
    False when the run had no such comparison (a reduced workload): never a claim that did not run."""
    cfg = rec.get("config", {})
    v = cfg.get("first_step_verified") or cfg.get("verified")
    if isinstance(v, dict):
        return bool(v.get("sha256"))
    return isinstance(v, str) and v.startswith("SHA-256 of every one")


def profile_chain_issue():
    """config 5's chain_kernel from the hash-matched PMC summary: vector instructions per chain-wave cycle -- how busy the one
    wave per 64 chains keeps its SIMD (the recurrence is a dependent chain: most cycles issue nothing)."""
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir), reverse=True):
        if not f.endswith("_c5_pmc_summary.json"):
            continue
        try:
            pmc = json.load(open(os.path.join(pdir, f)))
            if pmc.get("kernel_source_sha") != kernel_source_hash():
                continue
            k = [v for n, v in pmc["kernels"].items() if n.endswith("::chain_kernel") or "::chain_kernel" in n][0]
            # SQ_WAVE_CYCLES counts quad-cycles per resident wave (3 waves per workgroup: the chain wave and its two feeders)
            return {"valu_insts_per_launch": k["SQ_INSTS_VALU"], "wave_quad_cycles_per_launch": k["SQ_WAVE_CYCLES"],
                    "valu_insts_per_wave_cycle": k["SQ_INSTS_VALU"] / (4.0 * k["SQ_WAVE_CYCLES"]),
                    "is": "vector instructions issued per cycle of a resident wave of chain_kernel (chain wave + 2 feeder waves per 64 chains); "
                          "a wave that issued back to back would show about 0.25", "source": "profiles/" + f}
        except (OSError, KeyError, ValueError, IndexError):
            pass
    return None


def hbm_convention(achieved, alg, key="bytes_per_step"):
    """SURVEY 8d's byte model as a nested record: a convention (operator blocks priced as HBM traffic), not a roof the
    time-parallel kernels touch -- they keep those blocks in LDS."""
    return {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, key: alg,
            "is": "SURVEY 8d's 8 B per operator-sample + 2 B per frame over the kernel time: a convention, not this kernel's traffic"}


def cpu_reference(make_prg, what, voices, ops_per_voice, tabs, all_cores=False, budget_s=12.0):
    """The reference's own generator (oracle/_ref, built from its sources by oracle/Makefile) when
    that library is present, else this repo's CPU restatement; a bounded sample of the workload."""
    from oracle import pyoracle as po
    prg = make_prg()
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
    frames = 2205  # calibrate (the first call also pays for one-time set-up), then run ~budget_s of CPU work
    render(frames)
    frames = 8820
    t0 = time.perf_counter()
    render(frames)
    dt = time.perf_counter() - t0
    frames = int(min(44100 * 30, max(frames, frames * budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    render(frames)
    dt = time.perf_counter() - t0
    out = {"value": frames / dt, "unit": "mixed mono int16 frames/s", "cores": 1, "kind": kind,
           "sample": f"{what}, first {frames} frames ({frames * voices * ops_per_voice:.3g} operator-samples), "
                     f"{dt:.1f} s on 1 host thread"
                     + (" (sauGenerator_run of the compiled reference, -O3 -ffast-math as its Makefile)"
                        if kind == "reference" else " (oracle/sau_oracle.c)"),
           "operator_samples_per_s": frames * voices * ops_per_voice / dt}
    if all_cores:
        out["all_cores"] = cpu_baseline_all_cores(voices, frames_1core=frames / dt)
    return out


def cpu_baseline_all_cores(voices, frames_1core):
    """the config-3 bank with its voices partitioned over all host cores, one process
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


def sha256(a):
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# tests/tools/bench_seqexec.py (tests/test_dist.py) imports this module, sets HARNESS to {"new_batch": fn} and calls main():
# the N-rank logic of this script -- launcher, sharding, barriers, reductions -- then runs on a box without GPUs over the
# tests' sequential plan executor. Lines produced that way say so in `data` and measure nothing. Nothing in this file
# loads a test library or reads a test switch.
HARNESS = None


def test_backend():
    return HARNESS is not None


def new_batch(sa, prgs):
    if HARNESS is not None:
        return HARNESS["new_batch"](sa, prgs)
    return sa.Batch(prgs, 44100)


class Ranks:
    """One process per GPU; torch.distributed (backend "nccl" = RCCL) only for the barrier, the
    max-over-ranks clock and the after-the-fact report."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # SAU_BENCH_BACKEND=gloo lets the N>1 logic be exercised on a box with fewer GPUs than
        # ranks (ranks then share devices; rendezvous and reductions on the CPU)
        self.backend = os.environ.get("SAU_BENCH_BACKEND", "nccl")
        self.cuda = not test_backend()
        dev = local_rank % max(1, torch.cuda.device_count())
        os.environ.setdefault("SAU_AMD_DEVICE", str(dev))
        if self.cuda:
            torch.cuda.set_device(dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(self.backend)
        self.tdev = "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self, batches=()):
        if self.world > 1:
            self.dist.barrier()
        if self.cuda:
            self.torch.cuda.synchronize()
        for b in batches:
            b.sync()

    def max(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], device=self.tdev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, ints):
        if self.world == 1:
            return [int(x) for x in ints]
        t = self.torch.tensor([int(x) for x in ints], device=self.tdev, dtype=self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]

    def gather(self, obj):
        """every rank's `obj`, in rank order, on every rank"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def identity(self, sa, ms_per_step):
        """Who rendered what: rank, host, the HIP device it used and that device's PCI address, its own ms per step (the
        line's value uses the slowest rank's clock). With RCCL as the backend two ranks on one board are an error: the
        scaling figure would be that of a shared GPU."""
        import socket
        dev = int(os.environ.get("SAU_AMD_DEVICE", "0"))
        bus = None if test_backend() else sa.api.device_pci_bus_id(dev)
        mine = {"rank": self.rank, "host": socket.gethostname(), "device": dev, "pci_bus_id": bus,
                "ms_per_step": ms_per_step}
        ranks = self.gather(mine)
        if self.backend == "nccl" and self.world > 1:
            seen = {}
            for r in ranks:
                key = (r["host"], r["pci_bus_id"])
                if r["pci_bus_id"] is not None and key in seen:  # (an address the runtime would not give proves nothing either way)
                    raise SystemExit(f"rank {self.rank}: ranks {seen.get(key)} and {r['rank']} report the same GPU {key}: "
                                     "one process per GPU is the contract")
                seen[key] = r["rank"]
        return ranks

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def dropin_rate(sa, args):
    """Config 3's 10 s through sau_create_Generator / sauGenerator_run as the reference host drives them (saugns.c:589-618:
    11289-frame calls into a host buffer), creation to the last frame, best of twelve -- host copies and PCIe included. The loop
    is the C host's: the three entry points called through pre-bound ctypes handles (a numpy `.ctypes.data` and a wrapper
    object per call cost 0.3 ms of the 3 ms in rounds 3-5's figure, which a C host does not pay)."""
    import ctypes as C
    import numpy as np
    from saugns_amd import voicebank
    prg = voicebank.config3(n=args.voices, seconds=10)
    buf = np.zeros(11289, np.int16)
    L = sa.lib()
    create, run, destroy = L.sau_create_Generator, L.sauGenerator_run, L.sau_destroy_Generator
    bufp, got = C.c_void_p(buf.ctypes.data), C.c_size_t()
    gotp = C.byref(got)
    best = None
    for _ in range(12):
        t0 = time.perf_counter()
        g = create(prg.ptr, 44100)
        if not g:
            raise SystemExit("sau_create_Generator returned NULL: " + sa.last_error())
        t1 = time.perf_counter()
        more = run(g, bufp, 11289, False, gotp)
        n = got.value
        t2 = time.perf_counter()
        while more:
            more = run(g, bufp, 11289, False, gotp)
            n += got.value
        t3 = time.perf_counter()
        destroy(g)
        t4 = time.perf_counter()
        if best is None or t4 - t0 < best[0]:
            best = (t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3)
    return {"value": n / best[0], "unit": "mixed mono int16 frames/s", "frames": n, "seconds": best[0],
            "phases_ms": {"create": best[1] * 1e3, "first_call": best[2] * 1e3, "other_calls": best[3] * 1e3, "destroy": best[4] * 1e3,
                          "is": "sau_create_Generator; the first sauGenerator_run (the events at t = 0, the plans, the first engine run "
                                "and the wait for its PCM); the other calls (copies out of the read-ahead buffers, the waits for the "
                                "later runs); sau_destroy_Generator"},
            "what": "sau_create_Generator -> sauGenerator_run in 11289-frame calls into host memory -> sau_destroy_Generator, "
                    "the script's whole 10 s, best of 12"}


def run_bank(args, R, sa, tabs, name, steps=20, warmup=2):
    """A voice bank rendered like config 3 -- one step = the script's whole 10 s (441000 frames), state and PCM resident in
    HBM, the first step's SHA-256 checked against the compiled reference's -- for the two workloads the r03 verdict asked onto
    the driver's line: BASELINE config 2 (256 flat sines) and the carrier-FM bank (config 3 with the carrier's modulator
    list an FM list: running-sum phases, look-back build)."""
    import numpy as np
    from saugns_amd import voicebank
    index = json.load(open(os.path.join(GOLDEN, "index.json")))
    frames = 441000
    spec = {"config2": dict(make=lambda s: voicebank.config2(n=256, seconds=s), voices=256, ops=1,
                            what="BASELINE config 2: 256 independent Wsin voices, no modulation",
                            metric="mono samples/sec/GPU @ 256 flat voices", kernel="fast_kernel<8, 0>"),
            "fm": dict(make=lambda s: voicebank.config3_fm(n=1024, seconds=s), voices=1024, ops=4,
                       what="carrier-FM bank: 1024 voices x (carrier taking FM from a modulator with a 2-deep PM chain) = 4096 "
                            "operators (config 3 with f[...] for p[...]; the carrier's phase is a running sum)",
                       metric="mono samples/sec/GPU @ 1024 voices (carrier FM + depth-2 PM under it)",
                       kernel="fast_kernel<T, 2> (running sums in one pass, decoupled look-back)")}[name]
    prg = spec["make"](10 * (steps + warmup + 3) + 2)
    batch = new_batch(sa, [prg])
    pcm = batch.run(frames, stereo=False)[0]
    got, want = sha256(pcm[0]), index["configs"][name]["sha256"]
    if got != want:
        raise SystemExit(f"rank {R.rank}: {name}: first step's PCM {got[:16]} is not the reference's {want[:16]}")
    verified = {"sha256": got, "equals": f"tests/golden/index.json configs.{name}.sha256 (compiled reference)"}
    for _ in range(warmup):
        batch.run(frames, stereo=False, fetch=False)
    R.barrier([batch])
    batch.timing_ex(reset=True)
    batch.set_timing(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(frames, stereo=False, fetch=False)
    R.barrier([batch])
    dt = R.max(time.perf_counter() - t0)
    tm = batch.timing_ex()
    batch.close()
    if R.rank != 0:
        return None
    n_ops = spec["voices"] * spec["ops"]
    alg = (n_ops * 8 + 2) * frames
    launch_s = tm["fast_ms"] / 1e3 / steps  # (every time-parallel launch of a step)
    achieved = alg / launch_s / 1e9 if launch_s > 0 else 0.0
    out = {"metric": spec["metric"], "value": frames * steps * R.world / dt, "unit": "mixed mono int16 frames/s",
           "n_gpus": R.world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f64 table interpolation, u32 phase)", "data": "synthetic",
           "config": {"workload": spec["what"] + f", 44.1 kHz mono, {frames} frames per step, per GPU", "voices": spec["voices"],
                      "operators": n_ops, "frames_per_step": frames, "first_step_verified": verified,
                      "operator_samples_per_s": frames * steps * R.world / dt * n_ops},
           "roofline": {"traffic": None, "kernel": spec["kernel"], "kernel_ms_per_step": launch_s * 1e3,
                        "segments_per_step": tm["segments"] / steps, "algorithmic": hbm_convention(achieved, alg)}}
    tr, src = profile_step_traffic(name)
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, src
    out["hbm_real_frac"] = tr / (dt / steps) / 8e12 if tr else None
    if name == "config2":
        # one operator per voice: the voice row IS the operator's block, so here the byte model is the real traffic and HBM
        # is the roof that counts -- the step's measured bytes over the step's time
        out["roofline"].update({"bound": "hbm", "achieved": (tr / (dt / steps) / 1e9) if tr else achieved, "peak": 8000.0, "unit": "GB/s",
                                "frac": out["hbm_real_frac"] if tr else achieved / 8000.0,
                                "is": "measured HBM bytes of a step over the step time" if tr else "algorithmic bytes over the kernel time"})
    else:
        v = profile_step_valu(name, "fast_kernel<")
        out["roofline"].update(bound="valu", achieved=v["weighted_cycles_per_launch"] if v else None,
                               peak=v["simd_cycles_per_launch"] if v else None, unit="SIMD-cycles per launch",
                               frac=v["frac"] if v else None, valu=v)
        out["valu_frac"] = v["frac"] if v else None
    out["first_step_sha_ok"] = bool(verified and verified.get("sha256"))
    if not args.no_cpu and R.world == 1:
        out["cpu_baseline"] = cpu_reference(lambda: spec["make"](30), spec["what"].split(":")[0], spec["voices"], spec["ops"],
                                            tabs, budget_s=5.0)
    return out


def run_config3(args, R, sa, tabs):
    import numpy as np
    from saugns_amd import voicebank
    index = json.load(open(os.path.join(GOLDEN, "index.json")))
    sustain = 0 if test_backend() else max(0, args.sustain)
    seconds = max(1, (args.frames * (args.steps + args.warmup + sustain + 2)) // 44100 + 2)
    prg = voicebank.config3(n=args.voices, seconds=seconds)
    batch = new_batch(sa, [prg])
    # what is about to be timed is the reference's render: the first step (the script's first 10 s)
    verified = None
    pcm = batch.run(args.frames, stereo=False)[0]
    if args.voices == 1024 and args.frames == 441000:
        got, want = sha256(pcm[0]), index["configs"]["config3"]["sha256"]
        if got != want:
            raise SystemExit(f"rank {R.rank}: first step's PCM {got[:16]} is not the reference's {want[:16]}")
        verified = {"sha256": got, "equals": "tests/golden/index.json configs.config3.sha256 (compiled reference)"}
    for _ in range(args.warmup):
        batch.run(args.frames, stereo=False, fetch=False)
    R.barrier([batch])
    batch.timing_ex(reset=True)
    batch.set_timing(1)  # HIP events around the dominant kernel only, on its own stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(args.frames, stereo=False, fetch=False)
    R.barrier([batch])
    dt_mine = time.perf_counter() - t0
    dt = R.max(dt_mine)
    tm = batch.timing_ex()
    ranks = R.identity(sa, dt_mine / args.steps * 1e3)
    frames_total = args.frames * args.steps * R.world
    # The same step a thousand times over, a few seconds of nothing but these kernels: what the timed K steps give
    # when the device stays busy (clocks, thermals), and long enough for a sampled GPU-utilisation reading to see it.
    # Reported beside `value`, never as it.
    sustained = None
    if sustain:
        batch.set_timing(0)
        R.barrier([batch])
        t1 = time.perf_counter()
        for _ in range(sustain):
            batch.run(args.frames, stereo=False, fetch=False)
        R.barrier([batch])
        ds = R.max(time.perf_counter() - t1)
        sustained = {"steps": sustain, "seconds": ds, "ms_per_step": ds / sustain * 1e3,
                     "value": args.frames * sustain * R.world / ds}
    # the only exchange of the whole job is this after-the-fact reduction of
    # {frames rendered, PCM checksum} for the scaling report (outside the timed region)
    pcm = batch.run(args.frames, stereo=False)[0]
    mine = [args.frames * args.steps, int(np.asarray(pcm, dtype=np.int64).sum() & 0x7FFFFFFF)]
    tally = R.sum(mine)
    if tally[0] != frames_total or tally[1] != mine[1] * R.world:
        raise SystemExit(f"rank {R.rank}: ranks disagree on the rendered PCM ({tally} vs {mine} x {R.world})")
    dropin = dropin_rate(sa, args) if R.world == 1 and not test_backend() and not args.no_dropin else None
    if R.rank != 0:
        return None
    n_ops = args.voices * 4
    # 8 B per operator-sample (one f32 write + one f32 read of every operator's
    # block output) + 2 B per output frame, per launch of the kernel
    alg_bytes = (n_ops * 8 + 2) * args.frames
    launch_s = (tm["fast_ms"] / 1e3) / max(1, tm["segments"])
    achieved = alg_bytes / launch_s / 1e9 if launch_s > 0 else 0.0
    traffic, source, valu, step_bytes = profile_traffic("config3", "fast_kernel<") if verified else (None, None, None, None)
    # What binds this kernel (DESIGN.md 4.1 *Roofline*): vector-instruction issue. `valu` comes from the
    # hash-matched PMC summary (tools/collect_profile.py): SQ_INSTS_VALU_* split the launch's vector instructions by class, each
    # class is priced with the issue cost tools/valu_probe.hip measured on this part (profiles/r05_valu_costs.json; four waves
    # per SIMD, as this kernel runs), and the sum is set against 1024 SIMDs x the launch's cycles. SURVEY 8d's byte model is
    # kept beside it as `algorithmic` -- a convention (operator blocks priced as HBM traffic), not a bound this kernel meets.
    valu_frac = valu.get("frac") if valu else None
    hbm_real_frac = (step_bytes / (dt / args.steps) / 8e12) if step_bytes else None
    out = {
        "metric": "mono samples/sec/GPU @ N voices (depth-3 FM)",
        "value": frames_total / dt, "unit": "mixed mono int16 frames/s",
        "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f64 table interpolation, u32 phase)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config 3: {args.voices} voices x (carrier + 3-deep "
                               f"PM chain) = {n_ops} operators, 44.1 kHz mono, "
                               f"{args.frames} frames per step, per GPU",
                   "voices": args.voices, "operators": n_ops, "frames_per_step": args.frames,
                   "frames_all_ranks": tally[0], "pcm_checksum_all_ranks": tally[1],
                   "first_step_verified": verified, "sustained": sustained,
                   # the PCIe-inclusive rate through the drop-in API (never `value`): sauGenerator_run with the reference
                   # host's 11289-frame calls, PCM copied into the caller's buffer every call
                   "dropin_value": dropin,
                   "voice_samples_per_s": frames_total / dt * args.voices,
                   "operator_samples_per_s": frames_total / dt * n_ops},
        "roofline": {"bound": "valu",
                     "achieved": valu.get("weighted_cycles_per_launch") if valu else None,
                     "peak": valu.get("simd_cycles_per_launch") if valu else None,
                     "unit": "SIMD-cycles per launch (vector instructions by class x measured issue cost, against 1024 SIMDs x launch cycles)",
                     "frac": valu_frac, "traffic": traffic, "traffic_source": source,
                     "valu": valu,
                     "kernel": "fast_kernel<12, 0, false, true, false, true> (closed-form build, 12 rows per pass, wide table blocks in LDS: the row groups "
                               "away from the segment's ends, in the form without in-segment masks; tasks from one queue per XCD, and the launch mixes "
                               "about three quarters of the output frames itself -- DESIGN.md 4.1; avg_launch_ms also holds the 25 us launch of the "
                               "plain build that renders every voice's first and last group ahead of it)",
                     "avg_launch_ms": launch_s * 1e3, "launches": tm["segments"],
                     # SURVEY 8d's convention: every operator's block output priced as one f32 write + one f32 read in HBM. This
                     # kernel keeps those blocks in LDS -- HBM carries the voice rows only (`traffic`, about an eighth) -- so this
                     # "fraction" is against a roof the kernel does not touch and may exceed 1; kept because 8d defines it
                     "algorithmic": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                                     "bytes_per_launch": alg_bytes,
                                     "is": "SURVEY 8d's 8 B per operator-sample + 2 B per frame over the launch time: a convention, not this kernel's traffic"},
                     "hbm_real": {"bytes_per_step": step_bytes, "frac": hbm_real_frac,
                                  "is": "HBM bytes of every kernel of one step (PMC: FETCH_SIZE x 2 + WRITE_SIZE) over the step time, against 8 TB/s"}},
        # the same facts as top-level scalars (the driver's record keeps no nested objects: VERDICT r04 item 6)
        "first_step_sha_ok": bool(verified), "valu_frac": valu_frac, "hbm_real_frac": hbm_real_frac,
        "algorithmic_hbm_frac": achieved / 8000.0,
        "dropin_frames_per_s": dropin["value"] if dropin else None,
        "dropin_create_ms": dropin["phases_ms"]["create"] if dropin else None,
        "dropin_first_call_ms": dropin["phases_ms"]["first_call"] if dropin else None,
        "dropin_other_calls_ms": dropin["phases_ms"]["other_calls"] if dropin else None,
        "dropin_destroy_ms": dropin["phases_ms"]["destroy"] if dropin else None,
        "sustained_frames_per_s": sustained["value"] if sustained else None,
        "ranks": ranks,
    }
    if not args.no_cpu and R.world == 1:
        out["cpu_baseline"] = cpu_reference(lambda: voicebank.config3(n=args.voices, seconds=30),
                                            f"config 3 ({args.voices} voices x depth-3 PM)", args.voices, 4,
                                            tabs, all_cores=True)
        # the same as top-level scalars (the driver's record keeps no nested objects; VERDICT r05 weak 7: the all-cores figure
        # and its core count were lost with the nested one)
        cb = out["cpu_baseline"]
        out["cpu_1core_value"] = cb["value"]
        ac = cb.get("all_cores") or {}
        out["cpu_all_cores_value"] = ac.get("value")
        out["cpu_cores"] = ac.get("cores")
        out["cpu_kind"] = cb["kind"]
    return out


def cpu_reference_config4(fx, tabs, budget_s=10.0):
    """Config 4 on one host core: whole 60 s renders of seeds 0, 1, 2, ... through the compiled reference (or the
    oracle port), as many as fit the time budget (at least 2, at most 64)."""
    import numpy as np
    from oracle import pyoracle as po
    import saugns_amd as sa
    frames_each = int(fx["frames"][0])
    if po.have_ref():
        kind = "reference"
        po.ref()

        def render(prg):
            return po.ref_render(prg.ptr, 44100, False, chunk=11289)
    else:
        kind = "port"
        po.oracle_use_tables(tabs)
        po.oracle().ora_set_fastmath_forms(1)

        def render(prg):
            return po.oracle_render(prg.ptr, 44100, False, chunk=11289)
    render(sa.Program.from_image(fx["images"][0].tobytes()))  # one-time set-up
    n, frames, ok = 0, 0, True
    t0 = time.perf_counter()
    while n < 64 and (n < 2 or time.perf_counter() - t0 < budget_s):
        pcm = render(sa.Program.from_image(fx["images"][n].tobytes()))
        frames += len(pcm)
        ok = ok and (kind != "reference" or sha256(np.asarray(pcm)) == str(fx["sha256"][n]))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "mixed mono int16 frames/s summed over renders", "cores": 1, "kind": kind,
            "sample": f"config 4, seeds 0..{n - 1} of rainy_thunder.sau, whole {frames_each}-frame renders one after "
                      f"another, {dt:.1f} s on 1 host thread (hash of each hashed-render equals the fixture's: {ok})",
            "operator_samples_per_s": frames * 7 / dt}


def run_config4(args, R, sa, tabs, steps=None, warmup=None):
    import numpy as np
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from saugns_amd.shard import shard_range
    fx = np.load(os.path.join(GOLDEN, "config4_seeds.npz"))
    per_gpu = args.renders
    total = min(512, per_gpu * R.world)
    a, b = shard_range(total, R.rank, R.world)
    prgs = [sa.Program.from_image(fx["images"][k].tobytes()) for k in range(a, b)]
    frames_each = int(fx["frames"][0])
    run_len = args.c4_run if args.c4_run else frames_each  # one engine run per render unless told otherwise
    if args.c4_frames:  # tests: the head of every render only (then compared with the fixtures' PCM heads)
        frames_each = run_len = min(frames_each, args.c4_frames)

    def step(fetch, timing=None, keep=False):
        batch = new_batch(sa, prgs)
        batch.set_call_len(11289)  # the reference host's call size (saugns.c:471,526)
        if timing is not None:
            batch.set_timing(2)
        outs = [[] for _ in prgs]
        alive, n = True, 0
        while alive:
            pcm, more, lens = batch.run(run_len, stereo=False, fetch=fetch)
            if fetch:
                for i in range(len(prgs)):
                    outs[i].append(pcm[i, :lens[i]].copy())
            n += sum(lens)
            alive = any(more) and not args.c4_frames
        batch.sync()
        if timing is not None:
            t = batch.timing_ex()
            for k in timing:
                timing[k] += t[k]
        if keep:
            return n, outs, batch
        batch.close()
        return n, outs

    # The timed steps are pipelined two deep, as a host that renders batch after batch would drive them: step n + 1's generators
    # are created and its engine runs issued (host work: 64 programs' conversion, their t = 0 events, plans, uploads -- 0.2-0.3 ms)
    # while the device renders step n; a step's batch is synchronised with and destroyed once the next one is under way. Until
    # round 6 every step ended with a synchronisation and the device idled through the next one's set-up. Both sides of the timed
    # region are synchronised as ever. Two batches live at a time: the warm-up runs two steps at least, so that the pools hold
    # both sets of buffers before the clock starts.
    def issue(timing=None):
        batch = new_batch(sa, prgs)
        batch.set_call_len(11289)
        if timing is not None:
            batch.set_timing(2)
        alive, n = True, 0
        while alive:
            _pcm, more, lens = batch.run(run_len, stereo=False, fetch=False)
            n += sum(lens)
            alive = any(more) and not args.c4_frames
        return batch, n

    def finish(batch, timing=None):
        batch.sync()
        if timing is not None:
            t = batch.timing_ex()
            for k in timing:
                timing[k] += t[k]
        batch.close()

    def pipelined(count, timing=None):
        frames, pending = 0, None
        for _ in range(count):
            batch, n = issue(timing)
            frames += n
            if pending is not None:
                finish(pending, timing)
            pending = batch
        if pending is not None:
            finish(pending, timing)
        return frames

    warmup_done = max(warmup, 2) if not test_backend() else warmup
    pipelined(warmup_done)
    R.barrier()
    t0 = time.perf_counter()
    frames_mine = pipelined(steps)
    R.barrier()
    dt = R.max(time.perf_counter() - t0)
    # the kernels' own times (events around every launch) from `steps_timed` steps of their own, one at a time: with two batches in
    # flight a launch's events also span its wait for the other batch's workgroups
    tm = {"fast_ms": 0.0, "block_ms": 0.0, "mix_ms": 0.0, "aux_ms": 0.0, "segments": 0}
    steps_timed = min(steps, 2)
    for _ in range(steps_timed):
        pipelined(1, tm)
    # after the timed region: every render against the reference's SHA-256, then the ranks' report
    n, outs = step(True)
    if args.c4_frames:
        heads = np.load(os.path.join(GOLDEN, "pcm_heads.npz"))
        bad = [a + i for i, o in enumerate(outs) if f"config4_seed{a + i}" in heads and
               np.abs(np.concatenate(o).astype(int) - heads[f"config4_seed{a + i}"][:frames_each].astype(int)).max() > 1]
    else:
        bad = [a + i for i, o in enumerate(outs) if sha256(np.concatenate(o)) != str(fx["sha256"][a + i])]
    if bad:
        raise SystemExit(f"rank {R.rank}: renders {bad[:8]} differ from the reference's SHA-256")
    checksum = sum(int(np.concatenate(o).astype(np.int64).sum()) for o in outs) & 0x7FFFFFFFFFFF
    tally = R.sum([frames_mine, n, checksum, len(prgs)])
    if tally[0] != steps * frames_each * total or tally[3] != total:
        raise SystemExit(f"rank {R.rank}: frame count {tally} does not add up to {total} renders")
    gathered = pcm_gather(args, R, sa, step, fx, total, frames_each) if args.gather_pcm else None
    if R.rank != 0:
        return None
    # 7 operators per render (2 voices): 8 B per operator-sample + 2 B per output frame
    alg = (7 * 8 + 2) * frames_each * len(prgs)
    kern_s = tm["fast_ms"] / 1e3 / max(1, steps_timed)
    achieved = alg / kern_s / 1e9 if kern_s > 0 else 0.0
    full4 = not args.c4_frames and args.renders == 64 and not args.c4_run
    traffic4, source4 = profile_step_traffic("config4") if full4 else (None, None)
    # the step's dominant launch: since round 6 duo_kernel (closed-form and look-back voices in one launch); else the look-back build
    v4 = (profile_step_valu("config4", "duo_kernel") or profile_step_valu("config4", ", 2, false, false")) if full4 else None
    out = {
        "metric": "mono samples/sec, examples/rainy_thunder.sau x 512 renders sharded over GPUs",
        "value": tally[0] / dt, "unit": "mixed mono int16 frames/s summed over renders",
        "n_gpus": R.world, "steps": steps, "warmup": warmup_done,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (u64 cycle counters, u32 phase)", "data": "synthetic",
        "config": {"workload": f"BASELINE config 4: rainy_thunder.sau with seed=k, {len(prgs)} renders per GPU "
                               f"(seeds shard_range({total}, rank, {R.world})), 60 s at 44.1 kHz mono each; one step = "
                               f"the rank's renders from generator creation to the last frame; steps pipelined two deep (the next step's "
                               f"generators are set up while the device renders this one's)",
                   "renders_all_ranks": tally[3], "frames_all_ranks": tally[0],
                   "pcm_checksum_all_ranks": tally[2],
                   "verified": (f"SHA-256 of every one of the {total} renders equals the compiled reference's "
                                "(tests/golden/config4_seeds.npz)") if not args.c4_frames else
                               f"first {frames_each} frames of seeds 0..3 within 1 LSB of tests/golden/pcm_heads.npz"},
        "roofline": {"bound": "valu", "achieved": (v4 or {}).get("weighted_cycles_per_launch"), "peak": (v4 or {}).get("simd_cycles_per_launch"),
                     "unit": "SIMD-cycles per launch of the dominant kernel (" + ((v4 or {}).get("kernel") or "duo_kernel") + ")",
                     "frac": (v4 or {}).get("frac"), "valu": v4, "algorithmic": hbm_convention(achieved, alg),
                     "traffic": traffic4, "traffic_source": source4,
                     "traffic_is": "HBM bytes of every kernel of one step (64 renders)",
                     "kernel": "duo_kernel: the closed-form voices (as fast_kernel<8, 0>) and the look-back voices (as fast_kernel<8, 2>) of "
                               "analyze_kernel's two lists in one launch, a workgroup's waves split between them by the lists' lengths",
                     "kernel_ms_per_step": kern_s * 1e3, "other_kernels_ms_per_step":
                     {k: tm[k] / max(1, steps_timed) for k in ("block_ms", "mix_ms", "aux_ms")},
                     "segments_per_step": tm["segments"] / max(1, steps_timed),
                     "kernel_times_from": f"{steps_timed} steps of their own after the timed region, one batch at a time",
                     "note": "64 renders per GPU are 128 voices; one engine run (one segment) per render since r03"},
        "valu_frac": (v4 or {}).get("frac"), "hbm_real_frac": (traffic4 / (dt / steps) / 8e12) if traffic4 else None,
    }
    if gathered:
        out["config"]["pcm_gather"] = gathered
    if not args.no_cpu and R.world == 1:
        out["cpu_baseline"] = cpu_reference_config4(fx, tabs)
    return out


def pcm_gather(args, R, sa, step, fx, total, frames_each):
    """--gather-pcm (the optional exchange): after everything else, render the rank's share once more, leave
    the PCM in HBM and send it straight to rank 0 (RCCL send/recv inside one group; with gloo, tests, from host memory).
    Rank 0 then holds all `total` renders and checks every one against the reference's SHA-256 -- or, with --c4-frames,
    its own share against what it rendered. Never part of a timed step."""
    import numpy as np
    torch = R.torch
    from saugns_amd import shard
    if R.cuda and R.backend == "nccl":
        if args.c4_run and args.c4_run < frames_each:
            # (the batch's PCM block in HBM holds the last engine run only: ADVICE r04)
            raise SystemExit("--gather-pcm takes the PCM where the batch left it in HBM: one engine run per render (no --c4-run below the render's length)")
        n, _, batch = step(False, keep=True)
        local = torch.stack([shard.device_pcm_tensor(batch, i, frames_each) for i in range(n // frames_each)])
    else:
        batch = None
        n, outs = step(True)
        local = torch.from_numpy(np.stack([np.concatenate(o)[:frames_each] for o in outs]))
    # what every rank holds, render by render, told to rank 0 on the side (objects, not the data path): the gathered block
    # must hold exactly these, in rank order -- seeds shard_range(total, r, world) of rank r at rows [a_r, b_r)
    sums = [int(x.to(torch.int64).sum().item()) for x in local]
    all_sums = [x for part in R.gather(sums) for x in part]
    R.barrier()
    t0 = time.perf_counter()
    everything = shard.gather_renders_to_root(local, 0)
    R.barrier()
    dt = time.perf_counter() - t0
    if batch is not None:
        batch.close()
    if R.rank != 0:
        return None
    host = everything.cpu().numpy()
    if len(host) != total:
        raise SystemExit(f"rank 0: gathered {len(host)} renders, expected {total}")
    got_sums = [int(host[k].astype(np.int64).sum()) for k in range(total)]
    if got_sums != all_sums:
        bad = [k for k in range(total) if got_sums[k] != all_sums[k]]
        raise SystemExit(f"rank 0: gathered renders {bad[:8]} are not what their ranks rendered (rank order / content)")
    if not args.c4_frames:
        bad = [k for k in range(total) if sha256(host[k]) != str(fx["sha256"][k])]
        if bad:
            raise SystemExit(f"rank 0: gathered renders {bad[:8]} differ from the reference's SHA-256")
    nbytes = int(host.nbytes)
    return {"renders": int(len(host)), "bytes": nbytes, "seconds": dt, "GB_per_s": nbytes / dt / 1e9, "rank_order_checked": True,
            "distinct_renders": len(set(got_sums)),
            "backend": R.backend, "what": "torch.distributed.gather of int16 PCM to rank 0 (one direct send per rank), after "
            "the timed region; every gathered render's SHA-256 checked on rank 0" + (" (heads only: not hashed)" if args.c4_frames else "")}


def run_config5(args, R, sa, tabs, steps=None, warmup=None):
    import numpy as np
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from saugns_amd import voicebank
    index = json.load(open(os.path.join(GOLDEN, "index.json")))
    prg = voicebank.config5(n=args.voices5, seconds=10)
    frames = 441000

    def step(fetch, timing=None):
        batch = new_batch(sa, [prg])
        if timing is not None:
            batch.set_timing(2)
        pcm = batch.run(frames, stereo=False, fetch=fetch)[0]
        batch.sync()
        if timing is not None:
            t = batch.timing_ex()
            for k in timing:
                timing[k] += t[k]
        batch.close()
        return pcm

    pcm = step(True)
    verified = None
    if args.voices5 == 4096:
        got, want = sha256(pcm[0]), index["configs"]["config5"]["sha256"]
        if got != want:
            raise SystemExit(f"rank {R.rank}: config 5 PCM {got[:16]} is not the reference's {want[:16]}")
        verified = {"sha256": got, "equals": "tests/golden/index.json configs.config5.sha256 (compiled reference)"}
    for _ in range(warmup):
        step(False)
    R.barrier()
    tm = {"fast_ms": 0.0, "block_ms": 0.0, "mix_ms": 0.0, "aux_ms": 0.0, "segments": 0}
    # One step = the script's whole 10 s from generator creation to its last frame (creation 0.05 ms, the first run's host
    # work -- t = 0 events, operator records, plans -- 0.6 ms before the first kernel, 43 ms of feedback chains with the passes
    # and the mixer beside them, the last chunk's tail).
    def make():
        b = new_batch(sa, [prg])
        b.set_timing(2)
        return b

    def timed(pipelined):
        tm = {"fast_ms": 0.0, "block_ms": 0.0, "mix_ms": 0.0, "aux_ms": 0.0, "segments": 0}
        R.barrier()
        t0 = time.perf_counter()
        cur = make()
        cur.run(frames, stereo=False, fetch=False)  # (returns when everything is enqueued)
        last_pcm = None
        for i in range(steps):
            nxt = None
            if pipelined and i + 1 < steps:
                nxt = make()
                nxt.order_after(cur)  # its kernels start when this script's have finished; its host work starts now
                # (the last script's PCM is fetched -- 882 KB -- and checked below: what ran two deep is the reference's render too)
                r = nxt.run(frames, stereo=False, fetch=i + 2 == steps)
                last_pcm = r[0] if i + 2 == steps else last_pcm
            cur.sync()
            t = cur.timing_ex()
            for k in tm:
                tm[k] += t[k]
            cur.close()
            if not pipelined and i + 1 < steps:
                nxt = make()
                nxt.run(frames, stereo=False, fetch=False)
            cur = nxt
        R.barrier()
        dtx = R.max(time.perf_counter() - t0)
        if last_pcm is not None and verified and sha256(last_pcm[0]) != verified["sha256"]:
            raise SystemExit(f"rank {R.rank}: config 5: the last pipelined step's PCM is not the reference's")
        return dtx, tm

    # `value`: the K scripts strictly one after the other (each created, issued, drained). Beside it, for the record, the
    # same K with script k + 1 created and issued while script k renders, ordered behind it (sauAmd_Batch_order_after):
    # it hides the host work of a script's start (0.6 ms of device idle time per step) and measured no faster -- the
    # recurrence itself runs 2 % slower on a device that never idles (profiles/r05_headline_ab.json)
    dt_ordered, _ = timed(True) if args.c5_ordered else (None, None)
    dt, tm = timed(False)
    mine = [frames * steps, int(np.asarray(pcm, dtype=np.int64).sum() & 0x7FFFFFFF)]
    tally = R.sum(mine)
    if tally[0] != frames * steps * R.world or tally[1] != mine[1] * R.world:
        raise SystemExit(f"rank {R.rank}: ranks disagree ({tally} vs {mine} x {R.world})")
    if R.rank != 0:
        return None
    n_ops = args.voices5 * 2
    alg = (n_ops * 8 + 2) * frames
    dom = max(("block_ms", "fast_ms"), key=lambda k: tm[k])
    kern_s = tm[dom] / 1e3 / steps
    achieved = alg / kern_s / 1e9 if kern_s > 0 else 0.0
    traffic5, source5 = profile_step_traffic("config5") if args.voices5 == 4096 else (None, None)
    out = {
        "metric": "mono samples/sec/GPU @ N voices (self-feedback FM + AM + ramps)",
        "value": tally[0] / dt, "unit": "mixed mono int16 frames/s",
        "n_gpus": R.world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (f64 table interpolation, u32 phase)", "data": "synthetic",
        "config": {"workload": f"BASELINE config 5: {args.voices5} voices x (self-modulated carrier with frequency, "
                               f"amplitude and feedback ramps + range-AM modulator) = {n_ops} operators, 44.1 kHz mono; "
                               f"one step = the script's whole 10 s ({frames} frames) from generator creation on",
                   "voices": args.voices5, "operators": n_ops, "frames_per_step": frames,
                   "frames_all_ranks": tally[0], "pcm_checksum_all_ranks": tally[1], "verified": verified,
                   "steps_are": "K scripts strictly one after the other: each created, issued and drained before the next is created",
                   "ordered_two_generators": ({"value": tally[0] / dt_ordered, "ms_per_step": dt_ordered / steps * 1e3,
                                               "is": "script k + 1 created and issued while script k renders, its kernels ordered behind "
                                                     "script k's (sauAmd_Batch_order_after)"} if dt_ordered else None),
                   "operator_samples_per_s": tally[0] / dt * n_ops},
        # what binds this workload is the latency of one dependent chain per voice (wosc.h:273-310): nanoseconds per sample step
        # of chain_kernel against the bare recurrence on this part (tools/chain_probe2.hip: 96.9 ns with the wide table entries)
        "roofline": {"bound": "latency", "achieved": kern_s * 1e9 / frames, "peak": 96.9, "unit": "ns per sample step of the feedback recurrence (lower is better)",
                     "frac": (96.9 / (kern_s * 1e9 / frames)) if kern_s > 0 else None,
                     #  `peak` is this kernel's OWN bare loop -- the same 36 dependent instructions without their
                     # surroundings --, so `frac` says how little the kernel adds to its recurrence, not how far the machine is:
                     "peak_is": "own bare loop (tools/chain_probe2.hip): the recurrence's dependent instructions alone on this part",
                     # ... which is mostly idle: one wave per 64 chains runs the recurrence
                     "waves_resident": (args.voices5 + 63) // 64, "simds": 1024,
                     "simds_occupied_frac": ((args.voices5 + 63) // 64) / 1024.0,
                     "chain_wave_issue": profile_chain_issue(),
                     "algorithmic": hbm_convention(achieved, alg), "traffic": traffic5, "traffic_source": source5,
                     "traffic_is": "HBM bytes of every kernel of one step",
                     "kernel": "feedback recurrence + block loop (" + dom + ")",
                     "kernel_ms_per_step": kern_s * 1e3,
                     "all_kernels_ms_per_step": {k: tm[k] / steps for k in ("fast_ms", "block_ms", "mix_ms", "aux_ms")},
                     "segments_per_step": tm["segments"] / steps},
        "hbm_real_frac": (traffic5 / (dt / steps) / 8e12) if traffic5 else None,
    }
    if not args.no_cpu and R.world == 1:
        out["cpu_baseline"] = cpu_reference(lambda: voicebank.config5(n=args.voices5, seconds=10),
                                            f"config 5 ({args.voices5} feedback voices)", args.voices5, 2, tabs)
    return out


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this script, one per
    GPU (LOCAL_RANK selects it), and relay rank 0's JSON line. This process never initialises the GPU (nothing
    below imports torch or the library), so no process that has is ever replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):  # a rank that dies takes the others with it (they would wait at a barrier)
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()  # exactly the children started above
                    rcs[r] = p.wait()
        time.sleep(0.05)
    reader.join(timeout=10)
    line = got[0] if got else ""
    sys.stdout.write(line)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed (rank, exit code): {bad}")
    if not line.strip():
        raise SystemExit("bench.py: rank 0 printed no result line")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["config3", "config4", "config5", "config2", "fm"], default="config3")
    ap.add_argument("--frames", type=int, default=441000, help="config3: frames per step")
    ap.add_argument("--voices", type=int, default=1024, help="config3: voices")
    ap.add_argument("--renders", type=int, default=64, help="config4: renders per GPU")
    ap.add_argument("--voices5", type=int, default=4096, help="config5: voices")
    ap.add_argument("--c4-run", type=int, default=0, help="config4: frames per engine run (default: the whole 60 s)")
    ap.add_argument("--c4-frames", type=int, default=0, help="config4 (tests): render only the first frames of each script")
    ap.add_argument("--sustain", type=int, default=1000, help="config3: steps of the sustained run after the timed region "
                    "(reported under `sustained`, never `value`; 0: none)")
    ap.add_argument("--c5-serial", action="store_true", help="(accepted for old command lines: the default)")
    ap.add_argument("--c5-ordered", action="store_true", help="config5: also measure the steps with script k + 1 issued while script k renders, ordered behind it")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gather-pcm", action="store_true",
                    help="--workload config4: afterwards send every rank's finished PCM to rank 0 (SURVEY 8e, optional) and check it there")
    ap.add_argument("--no-dropin", action="store_true", help="config3: skip the drop-in API's PCIe-inclusive rate")
    ap.add_argument("--no-others", action="store_true", help="config3: no short config 5 / config 4 runs after it")
    ap.add_argument("--force-others", action="store_true", help="tests: the other workloads also beside a reduced config 3")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)  # before anything touches the GPU
    defaults = {"config3": (500, 20), "config4": (5, 1), "config5": (5, 1), "config2": (20, 2), "fm": (20, 2)}[args.workload]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]

    import numpy as np
    R = Ranks()
    if R.world != args.gpus:
        raise SystemExit(f"rank {R.rank}: --gpus {args.gpus} but the launcher's world has {R.world} ranks")
    import saugns_amd as sa
    tabs = np.fromfile(os.path.join(GOLDEN, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
    sa.set_piluts(tabs)
    if args.workload in ("config2", "fm"):
        out = run_bank(args, R, sa, tabs, args.workload, steps=args.steps, warmup=args.warmup)
    else:
        out = {"config3": run_config3, "config4": run_config4, "config5": run_config5}[args.workload](args, R, sa, tabs)
    others = {}
    if args.workload == "config3" and not args.no_others and ((args.voices == 1024 and args.frames == 441000) or args.force_others):
        # the other two BASELINE workloads, every rank alike (their barriers are collective), short runs
        for name, fn in (("config5", run_config5), ("config4", run_config4)):
            if name == "config5" and test_backend():
                continue  # (4096 feedback voices on the CPU plan executor of the rank-logic tests: minutes)
            o = fn(args, R, sa, tabs, steps=8 if name == "config4" else 3, warmup=1)  # (config 4: 3.5 ms steps, two deep)
            if o is not None:
                o["roofline"]["kernel_source_sha"] = kernel_source_hash()
                others[name] = o
        if not test_backend():  # (the CPU plan executor of the rank-logic tests would take minutes over these)
            for name in ("fm", "config2"):
                o = run_bank(args, R, sa, tabs, name, steps=40, warmup=5)  # (0.3-3 ms steps: ten of them were a sample of 3 ms)
                if o is not None:
                    o["roofline"]["kernel_source_sha"] = kernel_source_hash()
                    others[name] = o
    if out is not None:
        out["roofline"]["kernel_source_sha"] = kernel_source_hash()
        if test_backend():
            out["data"] = "TEST BACKEND (CPU plan executor of tests/seqexec): rank logic only, not a measurement"
        if others:
            out["other_workloads"] = others
            # ... and their headline figures as top-level scalars (the driver's record keeps no nested objects: VERDICT r04 item 6)
            for name, key in (("config5", "config5"), ("config4", "config4"), ("fm", "fm"), ("config2", "config2")):
                if name in others:
                    out[key + "_value"] = others[name]["value"]
                    out[key + "_ms_per_step"] = others[name]["ms_per_step"]
                    # from the run's own record: a SHA-256 comparison with the compiled reference's that ran and held
                    # -- config 5 at another voice count and config 4 cut short (--c4-frames) have none, and say so
                    out[key + "_sha_ok"] = sha_checked(others[name])
            if R.world > 1 and "config4" in others:
                # the north star's multi-GPU case at the top of an N > 1 line: 64 renders of rainy_thunder.sau per GPU,
                # seeds shard_range(64 N, rank, N), no data-path collective (the full record stays under other_workloads)
                c4 = others["config4"]
                out["config4_sharded"] = {k: c4[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "scaling")}
                out["config4_sharded"].update(renders_all_ranks=c4["config"]["renders_all_ranks"],
                                              frames_all_ranks=c4["config"]["frames_all_ranks"],
                                              verified=c4["config"]["verified"])
        print(json.dumps(out))
    R.close()


if __name__ == "__main__":
    main()
