"""One-rank RCCL sanity check of the collectives bench.py issues at N > 1 (int64 SUM, float64 MAX, MIN, barrier), on the
GPU box: python scripts/experiments/rccl_sanity.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
c = torch.tensor([1, 2, 3 << 40], dtype=torch.int64, device=dev)
t = torch.tensor([0.25], dtype=torch.float64, device=dev)
n = torch.tensor([7], dtype=torch.int64, device=dev)
dist.all_reduce(c, op=dist.ReduceOp.SUM)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(n, op=dist.ReduceOp.MIN)
dist.barrier()
torch.cuda.synchronize()
assert c.tolist() == [1, 2, 3 << 40] and t.item() == 0.25 and n.item() == 7
print("rccl ok", c.tolist(), t.item(), n.item())
dist.destroy_process_group()
