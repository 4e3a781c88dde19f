#!/usr/bin/env python3
"""One-rank RCCL exercise of the collectives bench.py / sharding.py use at N > 1 (gather, reduce, all_reduce, barrier):
checks on a one-GPU box that the nccl backend initialises and runs them; not a measurement."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
blk = torch.arange(40 * 1000, dtype=torch.uint8, device=dev)
out = [torch.empty_like(blk)]
w = dist.gather(blk, gather_list=out, dst=0, async_op=True)
w.wait()
assert torch.equal(out[0], blk)
c = torch.ones(10000, dtype=torch.int32, device=dev)
dist.reduce(c, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.5 and int(c.sum().item()) == 10000
dist.destroy_process_group()
print("rccl single-rank collectives ok")
