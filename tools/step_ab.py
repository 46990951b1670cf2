"""A/B of the bench step under library debug knobs, interleaved in one process (boxes and clocks drift between runs):
    python tools/step_ab.py --knob mpsr_debug_set_conv_pointwise --values 0,-1 [--rounds 5] [--steps 10]
Prints ms per step (median over rounds) for every value of the knob."""
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
ap.add_argument("--knob", default="mpsr_debug_set_conv_pointwise")
ap.add_argument("--values", default="0,-1")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
device = torch.device("cuda", 0)
net = dn.DeviceNet(W.synthetic_weights(seed=0), device=device)
inp, _ = bench.make_inputs(args.batch, 1024, 0, device)
step = bench.Step(net, inp, 1024)
knob = getattr(_lib.lib(), args.knob)
values = [int(v) for v in args.values.split(",")]
for _ in range(3):
    step()
samples = {v: [] for v in values}
for _ in range(args.rounds):
    for v in values:
        knob(v)
        net.fcache = {}  # (transformed filters kept across calls belong to one setting of the knobs)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        samples[v].append((time.perf_counter() - t0) / args.steps * 1e3)
for v in values:
    s = sorted(samples[v])
    print("%s(%d): %.3f ms per step (median of %d; min %.3f)" % (args.knob, v, s[len(s) // 2], len(s), s[0]))
