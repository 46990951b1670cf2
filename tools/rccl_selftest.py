"""Single-rank RCCL self-test (the GPU boxes available to the build have one GPU): initialises the nccl backend
exactly as bench.py / train_bench.py do for N > 1 and runs the collectives they use."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1 << 20, device="cuda")
w = dist.all_reduce(t, async_op=True)
w.wait()
e = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(e, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(t.sum()), float(e))
dist.destroy_process_group()
