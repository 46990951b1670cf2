"""Per-workgroup timeline of one conv_igemm_kernel launch (library built with -DMPSR_TRACE, see tools/README.md):
which CU every workgroup ran on, when it started, how long its prologue / K loop / epilogue took.

    MPSR_LIB_PATH=abl/trace.so python tools/conv_trace.py [--shape 12,12,1024,256,1,1] [--batch 256] [--tile 3]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="12,12,1024,256,1,1")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tile", type=int, default=3)
    ap.add_argument("--depth", type=int, default=-1)
    args = ap.parse_args()
    H, W, C, N, k, dil = [int(v) for v in args.shape.split(",")]
    B = args.batch
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_tile(args.tile)
    lib.mpsr_debug_set_conv_sched(0, 0)
    lib.mpsr_debug_set_conv_depth(args.depth)
    dev = torch.device("cuda")
    x = torch.randn((B, H, W, C), device=dev)
    w = torch.randn((N, k * k * C), device=dev) * 0.05
    y = torch.empty((B, H, W, N), device=dev)
    nrec = 1 << 16
    ws = torch.zeros((nrec * 16,), dtype=torch.float32, device=dev)

    def run():
        _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, C, w.data_ptr(), None, None, y.data_ptr(), N, k, k,
                                            dil, 1, 1, ws.data_ptr(), ws.numel(), _lib.stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ws.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    rec = ws.view(torch.int64).cpu().numpy().reshape(-1, 8)
    rec = rec[rec[:, 3] != 0]
    t0, t1, t2, t3, hw, xcc, r0, r1 = [rec[:, i] for i in range(8)]
    cu = ((xcc & 15) << 8) | ((hw >> 8) & 0xff)  # xcc, se/sh/cu bits
    clk = (t3 - t0) / np.maximum((r1 - r0) / 100e6, 1e-9) / 1e9
    print("launch %.1f us, %d workgroups traced, %d distinct CUs; shader clock while resident %.2f GHz (median)"
          % (e0.elapsed_time(e1) * 1e3, len(rec), len(set(cu.tolist())), float(np.median(clk))))
    base = r0.min()
    start_us = (r0 - base) / 100.0
    end_us = (r1 - base) / 100.0
    print("start times (us): min %.1f p50 %.1f p90 %.1f max %.1f; end: p50 %.1f max %.1f"
          % (start_us.min(), np.median(start_us), np.percentile(start_us, 90), start_us.max(), np.median(end_us),
             end_us.max()))
    late = start_us > 5.0
    print("workgroups starting late (>5 us): %d on %d CUs; per-CU count of late workgroups: %s"
          % (late.sum(), len(set(cu[late].tolist())), np.bincount(np.unique(cu[late], return_counts=True)[1]).tolist()
             if late.any() else []))
    print("per-CU total workgroups histogram (count -> CUs): %s"
          % np.bincount(np.unique(cu, return_counts=True)[1]).tolist())
    for name, a, b in (("prologue", t0, t1), ("k loop", t1, t2), ("epilogue", t2, t3), ("total", t0, t3)):
        d = (b - a).astype(np.float64)
        print("%-9s cycles: early wgs p50 %8.0f  late wgs p50 %8.0f" % (name, np.median(d[~late]),
                                                                       np.median(d[late]) if late.any() else 0))
    # busy CUs over time
    edges = np.arange(0, end_us.max() + 10, 10.0)
    active = [(int(((start_us <= t) & (end_us > t)).sum()), len(set(cu[(start_us <= t) & (end_us > t)].tolist())))
              for t in edges]
    print("t(us): resident workgroups / CUs with work: " + "  ".join("%d:%d/%d" % (t, a, c)
                                                                     for t, (a, c) in zip(edges, active)))


if __name__ == "__main__":
    main()
