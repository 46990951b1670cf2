"""The bench step (cfg3 forward + Chamfer) under one value of a library debug knob, for profiling:
    rocprofv3 --kernel-trace --stats ... -- python tools/step_knob_run.py --knob mpsr_debug_set_wino3_form --value 0 [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from monopsr_amd import _lib  # noqa: E402
from monopsr_amd.core import device_net as dn  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--knob", default="")
ap.add_argument("--value", type=int, default=-1)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--rccl", action="store_true", help="create a one-rank RCCL communicator first (what an N > 1 rank has alive)")
args = ap.parse_args()
device = torch.device("cuda", 0)
if args.rccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    t = torch.ones(4, device=device)
    dist.all_reduce(t)  # (the communicator is created by its first collective)
    torch.cuda.synchronize()
if args.knob:
    getattr(_lib.lib(), args.knob)(args.value)
net = dn.DeviceNet(W.synthetic_weights(seed=0), device=device)
inp, _ = bench.make_inputs(args.batch, 1024, 0, device)
step = bench.Step(net, inp, 1024)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
print("%s(%d): %.3f ms per step" % (args.knob, args.value, (time.perf_counter() - t0) / args.steps * 1e3))
