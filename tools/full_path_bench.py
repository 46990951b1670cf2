"""Per-image timing of the FULL instance path: KITTI-size image + proposal boxes in ->
preprocess, proposal crops, crop trunk + full-image trunk, feature crop + pool, squash/decoder/xyz, heads out.

    python tools/full_path_bench.py [--boxes 32] [--images 8]

`--images n` runs n images back to back per timed pass (the reference processes one image of 32 boxes per step,
configs/monopsr_model_000.yaml:14-17).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd.core import config_utils  # noqa: E402
from monopsr_amd.core import device_net as dn  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402
from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=32)
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--graph", action="store_true",
                    help="also time one image's launches captured into a HIP graph (torch.cuda.CUDAGraph) and replayed")
    ap.add_argument("--knob", action="append", default=[],
                    help="library debug knob for this run, e.g. --knob mpsr_debug_set_conv_sched=1 (repeatable)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    from monopsr_amd import _lib
    _lib.set_conv_math(args.math)
    for kv in args.knob:
        name, val = kv.split("=")
        getattr(_lib.lib(), name)(*[int(v) for v in val.split(",")])
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = dn.DeviceNet(weights, device=dev, full_trunk=True)
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, "test")
    rng = np.random.default_rng(0)
    H, Wd, B = 375, 1242, args.boxes
    samples = []
    for _ in range(args.images):
        h, w = rng.uniform(20, 200, B), rng.uniform(20, 200, B)
        y1, x1 = rng.uniform(0, H - 1 - h), rng.uniform(0, Wd - 1 - w)
        boxes = np.stack([y1, x1, y1 + h, x1 + w], 1).astype(np.float32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        samples.append(dict(
            rgb_image=t(rng.integers(0, 256, (H, Wd, 3)).astype(np.float32)), boxes_2d=t(boxes),
            boxes_2d_norm=t(boxes / np.array([H, Wd, H, Wd], np.float32)),
            cam_p=t(np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791],
                              [0, 0, 1, 0.002745884]], np.float32)),
            est_view_angs=t(rng.uniform(-0.6, 0.6, B).astype(np.float32)),
            class_indices=torch.ones((B, 1), dtype=torch.int32, device=dev),
            mean_lwh=t(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
            prop_cen_z_offset=torch.full((B,), 2.178, device=dev)))

    def run():
        for s in samples:
            model.build(dict(s))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (args.reps * args.images)
    batched = {}
    for n in (1, 2, 4, 8, 16):
        if n > args.images:
            break
        model.build_batch([dict(s) for s in samples[:n]])
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            model.build_batch([dict(s) for s in samples[:n]])
        e1.record()
        torch.cuda.synchronize()
        batched[str(n)] = round(e0.elapsed_time(e1) / (args.reps * n), 3)
    gflop = 2 * 167.1 + B * 12.393  # full-image trunk + per-crop path
    out = {"workload": "full path: 375x1242 image + %d boxes" % B, "knobs": args.knob, "ms_per_image": round(ms, 3),
           "images_per_s": round(1e3 / ms, 1), "crops_per_s": round(B * 1e3 / ms, 1),
           "algorithmic_GFLOP_per_image": round(gflop, 1), "TFLOP_per_s": round(gflop / ms, 1),
           "build_batch_ms_per_image_by_images_per_call": batched}
    if args.graph:
        # the ~250 short launches of one image as ONE graph launch: inputs stay in place (static buffers), the outputs
        # of the captured pass are rewritten by every replay
        try:
            static = dict(samples[0])
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    model.build(dict(static))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                graph_out = model.build(dict(static))
            torch.cuda.synchronize()
            ref = model.build(dict(static))
            g.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(graph_out[0]["centroids"], ref[0]["centroids"]))
            e0.record()
            n = args.reps * args.images
            for _ in range(n):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            gms = e0.elapsed_time(e1) / n
            out["hip_graph"] = {"ms_per_image": round(gms, 3), "images_per_s": round(1e3 / gms, 1),
                                "same_result_as_eager": same}
        except Exception as e:  # noqa: BLE001
            out["hip_graph"] = {"error": repr(e)[:300]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
