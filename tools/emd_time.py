"""Interleaved timing rounds of the EMD entry points (256 x 2048^2 = BASELINE cfg5's per-GPU share, and the model's
own 32 x 2304^2), with the exhausted-receiver skipping of the passes on (default) and off (mpsr_debug_set_emd_skip(0):
every pair evaluated, the round-3 kernels).  Same bits either way (tests/test_ops_gpu.py)."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from monopsr_amd import _lib  # noqa: E402
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


lib = _lib.lib()
for b, n in ((256, 2048), (32, 2304)):
    y1 = torch.rand((b, n, 3), device="cuda") * 2 - 1
    y2 = torch.rand((b, n, 3), device="cuda") * 2 - 1
    for rnd in range(1):
        for skip in (0, 1, 2):
            lib.mpsr_debug_set_emd_skip(skip)
            print("%d x %d^2 skip=%d (round %d): approx_match %.2f ms  fused loss %.2f  cost only %.2f" % (
                b, n, skip, rnd, timeit(lambda: am.approx_match(y1, y2), 5), timeit(lambda: am.emd_loss_fwd_bwd(y1, y2), 5),
                timeit(lambda: am.emd_loss_fwd_bwd(y1, y2, want_grads=False), 5)))
lib.mpsr_debug_set_emd_skip(2)

# r06: level culling of the fused loss (mpsr_debug_set_emd_cull: 0 plain evaluation in the caller's point order, 1 the
# default -- Morton-sorted clouds, far chunks of the four steepest levels skipped --, 2 sorted but nothing skipped), on
# bench.py's uniform clouds and on surface-like clouds (points on the faces of a car-sized box + 1 cm noise)
def surface(b, n, gen):
    p = (torch.rand((b, n, 3), device="cuda", generator=gen) * 2 - 1) * torch.tensor([2.0, 0.8, 0.9], device="cuda")
    ax = torch.randint(0, 3, (b, n), device="cuda", generator=gen)
    side = torch.randint(0, 2, (b, n), device="cuda", generator=gen).float() * 2 - 1
    for d, half in enumerate((2.0, 0.8, 0.9)):
        p[..., d] = torch.where(ax == d, side * half, p[..., d])
    return p + torch.randn(p.shape, device="cuda", generator=gen) * 0.01


gen = torch.Generator(device="cuda").manual_seed(6)
for name, b, n in (("uniform", 256, 2048), ("surface", 256, 2048), ("uniform", 32, 2304), ("surface", 32, 2304)):
    if name == "uniform":
        y1 = torch.rand((b, n, 3), device="cuda", generator=gen) * 2 - 1
        y2 = torch.rand((b, n, 3), device="cuda", generator=gen) * 2 - 1
    else:
        y1, y2 = surface(b, n, gen), surface(b, n, gen)
    for rnd in range(2):
        for cull in (0, 1, 2):
            lib.mpsr_debug_set_emd_cull(cull)
            print("%s %d x %d^2 cull=%d (round %d): fused loss %.3f ms  cost only %.3f" % (
                name, b, n, cull, rnd, timeit(lambda: am.emd_loss_fwd_bwd(y1, y2), 5),
                timeit(lambda: am.emd_loss_fwd_bwd(y1, y2, want_grads=False), 5)))
lib.mpsr_debug_set_emd_cull(0)  # (the default)
