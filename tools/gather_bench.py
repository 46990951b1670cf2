"""The decoder's two upsampled convolutions (conv2_1: 12x12x512 -> 24x24x256; conv3_1: 24x24x256 -> 48x48x128, B = 256;
reference net_builder.py:72-77, :81-85) through mpsr_conv3x3_upsampled_f32, per kernel: the tap GEMMs (pw_conv_kernel)
and upconv_gather_kernel, durations from torch.profiler (roctracer records), whole call from HIP events.
Knock-out builds (tools/build_variant.sh <name> upconv.hip -DUPC_SKIP_SUM / -DUPC_SKIP_LOAD / -DUPC_SKIP_STORE, then
MPSR_LIB_PATH=abl/<name>.so) say which of the gather's three streams (LDS reads + arithmetic, z loads, result stores)
its time follows.  usage: python tools/gather_bench.py [--batch 256] [--reps 10]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from monopsr_amd.core import device_net as dn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--band", type=int, default=0, help="mpsr_debug_set_upconv_band: output rows per band (0 = the rule)")
    args = ap.parse_args()
    from monopsr_amd import _lib
    _lib.lib().mpsr_debug_set_upconv_band(args.band)
    dev = torch.device("cuda", 0)
    B = args.batch
    out = {"lib": os.environ.get("MPSR_LIB_PATH", "default"), "batch": B, "band": args.band}
    for name, (h, C, N) in (("conv2_1", (12, 512, 256)), ("conv3_1", (24, 256, 128))):
        g = torch.Generator(device=dev).manual_seed(h)
        x = torch.randn((B, h, h, C), device=dev, generator=g).clamp_(min=0)
        w = torch.randn((N, 9 * C), device=dev, generator=g) * 0.02
        bias = torch.randn((N,), device=dev, generator=g)
        f = lambda: dn.conv3x3_upsampled(x, (2 * h, 2 * h), w, bias, relu=True)
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        rec = {"call_us": round(e0.elapsed_time(e1) * 1e3 / args.reps, 1)}
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(args.reps):
                    f()
                torch.cuda.synchronize()
            for ev in prof.key_averages():
                us = getattr(ev, "self_device_time_total", None) or getattr(ev, "device_time_total", None) or 0
                short = next((k for k in ("upconv_gather_kernel", "pw_conv_kernel", "upconv_weights_kernel")
                              if k in ev.key), None)
                if us and short:
                    rec[short] = [round(us / args.reps, 1), ev.count // args.reps]
        except Exception as e:  # noqa: BLE001
            rec["profiler_error"] = repr(e)[:200]
        z_mb = B * h * h * 9 * N * 4 / 1e6
        y_mb = B * 4 * h * h * N * 4 / 1e6
        rec["gather_bytes_MB"] = round(z_mb + y_mb, 1)
        out[name] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
