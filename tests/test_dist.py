"""Multi-GPU path on CPU: sharding of independent renders over ranks (gloo, world size 2).

BASELINE config 4 shards whole renders; there is no data-path collective
 -- only the metadata all-reduce of {frames, checksum} that the
scaling report uses."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def test_shard_ranges_cover_everything():
    from saugns_amd.shard import shard_range
    for total in (1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                a, b = shard_range(total, r, world)
                got += list(range(a, b))
            assert got == list(range(total))


def test_two_rank_gloo_batch(tmp_path, seqexec):  # (the fixture rebuilds libseqexec.so when stale)
    script = textwrap.dedent("""
        import os, sys, ctypes as C
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, torch, torch.distributed as dist
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        import saugns_amd as sa
        from saugns_amd.shard import shard_range, reduce_report
        from conftest import load_program, GOLDEN
        tabs = np.fromfile(os.path.join(GOLDEN, "piluts_ref.f32"), dtype="<f4").reshape(12, 2048)
        sa.set_piluts(tabs)
        sa.api.use_hooks(os.path.join(%r, "tests", "hooks", "libsaugns_amd_hooks.so"))
        seq = C.CDLL(os.path.join(%r, "tests", "seqexec", "libseqexec.so"))
        seq.seq_backend_create.restype = C.c_void_p; seq.seq_backend_create.argtypes = [C.c_uint32]
        # as bench.py --workload config4 shards them: seeds of tests/golden/config4_seeds.npz (all 512
        # of BASELINE config 4), here two renders per rank
        fx = np.load(os.path.join(GOLDEN, "config4_seeds.npz"))
        keys = ["config4_seed%%d" %% k for k in range(4)]
        a, b = shard_range(2 * world, rank, world)
        assert (a, b) == (2 * rank, 2 * rank + 2)
        prgs = [sa.Program.from_image(fx["images"][k].tobytes()) for k in range(a, b)]
        batch = sa.Batch(prgs, 44100, backend=seq.seq_backend_create(1016))
        batch.set_call_len(11289)
        outs = batch.render(chunk=11025, max_frames=11025)
        heads = np.load(os.path.join(GOLDEN, "pcm_heads.npz"))
        for k, pcm in zip(keys[a:b], outs):
            assert int(np.abs(pcm.astype(int) - heads[k][:11025].astype(int)).max()) <= 1, k
        frames, checksum = reduce_report(sum(len(o) for o in outs), sum(int(o.astype(np.int64).sum()) for o in outs))
        if rank == 0:
            want = sum(int(heads[k][:11025].astype(np.int64).sum()) for k in keys)
            assert frames == 4 * 11025, frames
            assert abs(checksum - want) <= 4 * 11025, (checksum, want)
            print("OK", frames)
        dist.destroy_process_group()
    """ % (ROOT, ROOT, ROOT, ROOT))
    f = tmp_path / "rank.py"
    f.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          "29533", str(f)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OK 44100" in out.stdout


def _bench(argv, seqexec, launcher=False, env_extra=None, timeout=600):
    """bench.py's N-rank logic on a box without GPUs: the host control plane over the sequential executor, gloo."""
    env = dict(os.environ, SAU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", "29541"]
    # (tests/tools/bench_seqexec.py = bench.py with its batches made over the sequential executor)
    return subprocess.run(cmd + [os.path.join(ROOT, "tests", "tools", "bench_seqexec.py")] + argv, capture_output=True, text=True,
                          env=env, timeout=timeout)


def _line(out):
    import json
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_config4(seqexec):
    """`python bench.py --gpus 2 --workload config4` -- the driver's command shape, no launcher around it -- runs two
    ranks that render shard_range(2 * 2, r, 2) of BASELINE config 4's seeds and reduce {frames, checksum}."""
    j = _line(_bench(["--gpus", "2", "--workload", "config4", "--renders", "2", "--c4-frames", "11025",
                      "--steps", "1", "--warmup", "0", "--no-cpu"], seqexec))
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["config"]["renders_all_ranks"] == 4 and j["config"]["frames_all_ranks"] == 4 * 11025
    assert "TEST BACKEND" in j["data"]  # a line made this way never passes for a measurement


def test_bench_gpus_2_starts_two_ranks_config3(seqexec):
    j = _line(_bench(["--gpus", "2", "--voices", "8", "--frames", "4410", "--steps", "2", "--warmup", "1",
                      "--no-cpu"], seqexec))
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["frames_all_ranks"] == 2 * 2 * 4410
    one = _line(_bench(["--gpus", "1", "--voices", "8", "--frames", "4410", "--steps", "2", "--warmup", "1",
                        "--no-cpu"], seqexec))
    assert one["n_gpus"] == 1 and one["config"]["frames_all_ranks"] == 2 * 4410
    # the same bank on every rank: the all-rank checksum is the single rank's, twice
    assert j["config"]["pcm_checksum_all_ranks"] == 2 * one["config"]["pcm_checksum_all_ranks"]
    # who rendered: one record per rank (host, device, PCI address -- none on the CPU test backend --, its own clock)
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and len(one["ranks"]) == 1
    assert all(r["host"] and r["ms_per_step"] > 0 and r["pci_bus_id"] is None for r in j["ranks"])
    assert max(r["ms_per_step"] for r in j["ranks"]) <= j["ms_per_step"] * 1.001  # (the line's clock is the slowest rank's)


def test_bench_gpus_2_gathers_finished_pcm_to_rank_0(seqexec):
    """SURVEY.md 8e's optional exchange, `--gather-pcm`: after the timed region every rank sends its finished renders
    straight to rank 0 (torch.distributed.gather; gloo from host memory here, RCCL from HBM on GPUs), which then holds
    all of them in rank order."""
    j = _line(_bench(["--gpus", "2", "--workload", "config4", "--renders", "2", "--c4-frames", "11025", "--steps", "1",
                      "--warmup", "0", "--no-cpu", "--gather-pcm"], seqexec))
    g = j["config"]["pcm_gather"]
    assert g["renders"] == 4 and g["bytes"] == 4 * 11025 * 2 and g["backend"] == "gloo" and g["seconds"] > 0


def test_bench_gpus_8_config4_at_the_north_stars_split(seqexec):
    """The north star's multi-GPU case with its real geometry, as far as a box without GPUs can run it:
    eight ranks (gloo), rank r renders seeds shard_range(512, r, 8) = [64 r, 64 r + 64) of BASELINE config 4 -- the heads of the
    renders, over the sequential executor --, the ranks tally {frames, renders} and, with --gather-pcm, send their finished PCM
    to rank 0, which must end up holding all 512 renders in rank order (checked against what every rank says it rendered).
    The day an 8-GPU node exists the only new thing is RCCL itself."""
    j = _line(_bench(["--gpus", "8", "--workload", "config4", "--renders", "64", "--c4-frames", "2205", "--steps", "1",
                      "--warmup", "0", "--no-cpu", "--gather-pcm"], seqexec, timeout=1500))
    assert j["n_gpus"] == 8 and j["scaling"] == "weak"
    assert j["config"]["renders_all_ranks"] == 512 and j["config"]["frames_all_ranks"] == 512 * 2205
    assert "shard_range(512, rank, 8)" in j["config"]["workload"]
    g = j["config"]["pcm_gather"]
    assert g["renders"] == 512 and g["bytes"] == 512 * 2205 * 2 and g["backend"] == "gloo" and g["rank_order_checked"]
    assert g["distinct_renders"] > 500  # (seed = k: every render is its own)


def test_gather_renders_to_root_keeps_rank_order(tmp_path):
    """shard.gather_renders_to_root with two gloo ranks: rank 0 gets [rank 0's renders, rank 1's], rank 1 nothing."""
    script = f"""
import sys
sys.path.insert(0, {ROOT!r})
import torch, torch.distributed as dist
from saugns_amd.shard import gather_renders_to_root
dist.init_process_group("gloo")
r = dist.get_rank()
local = (torch.arange(6, dtype=torch.int16).reshape(2, 3) + 100 * r)
got = gather_renders_to_root(local, 0)
if r == 0:
    assert got.tolist() == [[0, 1, 2], [3, 4, 5], [100, 101, 102], [103, 104, 105]], got
    print("GATHER OK")
else:
    assert got is None
dist.destroy_process_group()
"""
    f = tmp_path / "g.py"
    f.write_text(script)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(f)], capture_output=True, text=True,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert out.returncode == 0 and "GATHER OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_gpus_2_default_workload_carries_config4_sharded(seqexec):
    """With N > 1 the default command's line has the north star's multi-GPU case at its top: config 4, 64 renders per GPU
    sharded by seed (here 2 per rank, heads only), next to config 3's value."""
    j = _line(_bench(["--gpus", "2", "--voices", "8", "--frames", "4410", "--steps", "1", "--warmup", "0", "--no-cpu",
                      "--force-others", "--renders", "2", "--c4-frames", "11025"], seqexec))
    c4 = j["config4_sharded"]
    assert c4["n_gpus"] == 2 and c4["renders_all_ranks"] == 4 and c4["frames_all_ranks"] == 8 * 4 * 11025  # (eight timed steps of the side workload)
    assert c4["scaling"] == "weak" and c4["value"] > 0
    assert j["other_workloads"]["config4"]["config"]["renders_all_ranks"] == 4
    one = _line(_bench(["--gpus", "1", "--voices", "8", "--frames", "4410", "--steps", "1", "--warmup", "0", "--no-cpu",
                        "--force-others", "--renders", "2", "--c4-frames", "11025"], seqexec))
    assert "config4_sharded" not in one and one["other_workloads"]["config4"]["config"]["renders_all_ranks"] == 2


def test_bench_under_a_launcher_and_world_mismatch(seqexec):
    """Under torch.distributed.run the ranks are the launcher's; --gpus must agree with the world it made."""
    j = _line(_bench(["--gpus", "2", "--voices", "8", "--frames", "4410", "--steps", "1", "--warmup", "0",
                      "--no-cpu"], seqexec, launcher=True))
    assert j["n_gpus"] == 2
    bad = _bench(["--gpus", "4", "--voices", "8", "--frames", "4410", "--steps", "1", "--warmup", "0", "--no-cpu"],
                 seqexec, launcher=True)
    assert bad.returncode != 0 and "--gpus 4" in bad.stdout + bad.stderr


def test_bench_launcher_reports_a_failed_rank(seqexec):
    """A rank that cannot run (here: a test backend that does not exist) makes `bench.py --gpus 2` exit non-zero."""
    out = _bench(["--gpus", "2", "--voices", "8", "--frames", "4410", "--steps", "1", "--warmup", "0", "--no-cpu"],
                 seqexec, env_extra={"SAU_SEQEXEC_LIB": "/nonexistent/libseqexec.so"}, timeout=300)
    assert out.returncode != 0


import pytest


@pytest.mark.gpu
def test_bench_gpus_2_on_one_gpu_box_with_cpu_rendezvous():
    """The same command on a GPU box: `bench.py --gpus 2` starts two ranks that share the one device (rendezvous and
    reductions through gloo), each rendering its shard of BASELINE config 4 at full length on the HIP backend, every
    render's SHA-256 checked against the compiled reference's; then config 3, where both ranks render the same bank."""
    env = dict(os.environ, SAU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)

    def run(argv):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                             env=env, timeout=900)
        return _line(out)

    j = run(["--gpus", "2", "--workload", "config4", "--renders", "4", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert j["n_gpus"] == 2 and j["config"]["renders_all_ranks"] == 8
    assert j["config"]["frames_all_ranks"] == 8 * 2646000 and "TEST BACKEND" not in j["data"]
    assert "SHA-256 of every one of the 8 renders" in j["config"]["verified"]
    # the optional exchange, both ways: two ranks through gloo from host memory; one rank with the PCM taken where the
    # batch left it in HBM (the tensor RCCL would send) -- every gathered render's SHA-256 checked on rank 0
    g = run(["--gpus", "2", "--workload", "config4", "--renders", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--gather-pcm"])
    assert g["config"]["pcm_gather"]["renders"] == 4 and g["config"]["pcm_gather"]["bytes"] == 4 * 2646000 * 2
    env.pop("SAU_BENCH_BACKEND")
    g = run(["--gpus", "1", "--workload", "config4", "--renders", "3", "--steps", "1", "--warmup", "0", "--no-cpu", "--gather-pcm"])
    assert g["config"]["pcm_gather"]["renders"] == 3 and g["config"]["pcm_gather"]["backend"] == "nccl"
    env["SAU_BENCH_BACKEND"] = "gloo"
    k = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-others"])
    assert k["n_gpus"] == 2 and k["config"]["frames_all_ranks"] == 2 * 3 * 441000
    assert k["config"]["first_step_verified"]["sha256"].startswith("3211740aca595248")
