"""nn_distance forward timing (interleaved rounds) and bit-exactness against the C oracle on a slice.
    [MPSR_LIB_PATH=...] python tools/nn_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd  # noqa: E402
from oracle import ops as orc  # noqa: E402

for b, n, m in ((256, 1024, 1024), (32, 2304, 2304), (2048, 1024, 1024)):
    x1, x2 = torch.randn((b, n, 3), device="cuda"), torch.randn((b, m, 3), device="cuda")
    r = nnd.nn_distance(x1, x2)
    ref = orc.nn_distance(x1[:2].cpu().numpy(), x2[:2].cpu().numpy())
    ok = all((a[:2].cpu().numpy() == b_).all() for a, b_ in zip(r, ref))
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            nnd.nn_distance(x1, x2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = sorted(ts)[3]
    print("%d x %d x %d: %.1f us  %.2f T pairs/s  bit-exact %s" % (b, n, m, t * 1e3, 2.0 * b * n * m / (t * 1e-3) / 1e12, ok))
