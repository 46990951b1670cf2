"""Timing of the point-cloud ops on the GPU (HIP events) with their algorithmic-bytes / pair-rate figures, next to
the CPU oracle on a bounded sample.

    python tools/ops_bench.py [--b 256] [--chamfer-n 1024] [--emd-n 2048] [--emd-b 256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am  # noqa: E402
from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd  # noqa: E402
from oracle import ops as orc  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=256)
    ap.add_argument("--chamfer-n", type=int, default=1024)
    ap.add_argument("--emd-b", type=int, default=256)
    ap.add_argument("--emd-n", type=int, default=2048)
    ap.add_argument("--cpu-clouds", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda")
    out = {}

    b, n = args.b, args.chamfer_n
    x1, x2 = torch.randn((b, n, 3), device=dev), torch.randn((b, n, 3), device=dev)
    with torch.no_grad():
        d1, i1, d2, i2 = nnd.nn_distance(x1, x2)
        t_f = timeit(lambda: nnd.nn_distance(x1, x2), 20)
        ones = torch.ones_like(d1)
        t_b = timeit(lambda: nnd.nn_distance_grad(x1, x2, ones, i1, ones, i2), 20)
    pairs = 2.0 * b * n * n
    out["chamfer"] = {"shape": [b, n, n], "fwd_us": round(t_f * 1e6, 1), "bwd_us": round(t_b * 1e6, 1),
                      "fwd_alg_GBps": round(b * 2 * n * 20 / t_f / 1e9, 1),
                      "bwd_alg_GBps": round(b * 2 * n * 32 / t_b / 1e9, 1),
                      "fwd_Gpairs_per_s": round(pairs / t_f / 1e9, 1),
                      "fwd_valu_lane_ops_frac_of_78.6T": round(pairs * 9 / t_f / 78.6e12, 3)}
    c = args.cpu_clouds
    a1, a2 = x1[:c].cpu().numpy(), x2[:c].cpu().numpy()
    t0 = time.perf_counter()
    r = orc.nn_distance(a1, a2)
    t1 = time.perf_counter()
    orc.nn_distance_grad(a1, a2, np.ones_like(r[0]), r[1], np.ones_like(r[2]), r[3])
    t2 = time.perf_counter()
    out["chamfer"]["cpu_oracle_1thread"] = {"clouds": c, "fwd_clouds_per_s": round(c / (t1 - t0), 1),
                                            "bwd_clouds_per_s": round(c / (t2 - t1), 1)}
    out["chamfer"]["gpu_fwd_clouds_per_s"] = round(b / t_f, 1)

    b, n = args.emd_b, args.emd_n
    y1 = torch.rand((b, n, 3), device=dev) * 2 - 1
    y2 = torch.rand((b, n, 3), device=dev) * 2 - 1
    with torch.no_grad():
        match = am.approx_match(y1, y2)
        t_m = timeit(lambda: am.approx_match(y1, y2), 3)
        t_c = timeit(lambda: am.match_cost(y1, y2, match), 5)
        t_g = timeit(lambda: am.match_cost_grad(y1, y2, match), 5)
    mbytes = 4.0 * b * n * n
    out["emd"] = {"shape": [b, n, n], "approx_match_ms": round(t_m * 1e3, 2), "match_cost_ms": round(t_c * 1e3, 2),
                  "match_cost_grad_ms": round(t_g * 1e3, 2),
                  "approx_match_alg_GBps": round(mbytes / t_m / 1e9, 1),
                  "approx_match_Gexp_per_s": round(40.0 * b * n * n / t_m / 1e9, 1),
                  "match_cost_GBps": round(mbytes / t_c / 1e9, 1),
                  "match_cost_grad_GBps": round(2 * mbytes / t_g / 1e9, 1),
                  "clouds_per_s(match+cost+grad)": round(b / (t_m + t_c + t_g), 1)}
    with torch.no_grad():
        t_f = timeit(lambda: am.emd_loss_fwd_bwd(y1, y2), 3)
        t_fc = timeit(lambda: am.emd_loss_fwd_bwd(y1, y2, want_grads=False), 3)
        t_cp = timeit(lambda: am.approx_match(y1, y2, temp_floats=b * 2 * n * 2), 2)
        hb = max(1, b // 16)
        t_h = timeit(lambda: am.emd_loss_fwd_bwd(y1[:hb], y2[:hb], semantics="host"), 1) * (b / hb)
    # pair evaluations of the fused loss: 21 passes x 1 exponential + 2 loss passes x 3 exponentials
    out["emd"].update({"fused_loss_ms": round(t_f * 1e3, 2), "fused_cost_only_ms": round(t_fc * 1e3, 2),
                       "approx_match_compact_scratch_ms": round(t_cp * 1e3, 2),
                       "fused_loss_host_semantics_ms(extrapolated from %d clouds)" % hb: round(t_h * 1e3, 1),
                       "fused_loss_clouds_per_s": round(b / t_f, 1),
                       "fused_loss_Gexp_per_s": round((21.0 + 6.0) * b * n * n / t_f / 1e9, 1)})
    c = 2
    a1, a2 = y1[:c].cpu().numpy(), y2[:c].cpu().numpy()
    t0 = time.perf_counter()
    mt = orc.approx_match(a1, a2, "cpu")
    t1 = time.perf_counter()
    out["emd"]["cpu_oracle_1thread_clouds_per_s(approx_match, cpu semantics)"] = round(c / (t1 - t0), 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
