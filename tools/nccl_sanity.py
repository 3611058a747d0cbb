"""One-rank RCCL sanity check of the collectives bench.py uses at N>1 (barrier, MAX and SUM
all-reduce on a device tensor): python tools/nccl_sanity.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([7, 9], device="cuda", dtype=torch.int64); dist.all_reduce(u, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
print("rccl ok", t.item(), u.tolist())
dist.destroy_process_group()
