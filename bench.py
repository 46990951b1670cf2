#!/usr/bin/env python3
"""bench.py -- instance-crops/s through MonoPSR's per-instance hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic instances per GPU (BASELINE.json config 3):
    256 synthetic 48x48 RGB crops  -> ResNet-101 trunk to block3 (output stride 4)
    + synthetic (256,12,12,1024) full-image feature crop -> 1x1 squash, map decoder, xyz-map head (N x 3 cloud)
    -> centroid / lwh / alpha heads
    -> Chamfer (nn_distance) forward + backward between the first 1024 predicted points and a 1024-point GT cloud.
Everything runs through the C ABI of libmonopsr_hip.so on the current HIP stream; inputs and weights are resident
in HBM before the timed region.  Multi-GPU: instances are sharded, one process per GPU, no data-path collective
(weak scaling; `--allreduce-grads` adds an RCCL all-reduce of a gradient-sized buffer per step).

`--gpus N` with N > 1: under `python -m torch.distributed.run --nproc-per-node N` (WORLD_SIZE set) every process is
one rank; started plainly (`python bench.py --gpus N`) this process only starts N rank processes of itself -- before
torch is imported, so nothing here has touched a GPU -- waits for them and exits with their status.

Prints ONE JSON line on rank 0 (contract in the task description); `roofline`, `cpu_baseline`, `nn_distance` and
`emd` are extra objects.
"""
import argparse
import json
import os
import sys
import time


def _launch_ranks(argv):
    """`bench.py --gpus N` outside a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment), wait, exit non-zero if any rank failed.  Runs before anything
    GPU-related is imported; never re-executes a process that has initialised the GPU.  Returns only when this
    process is itself a rank (N == 1, or a launcher already set WORLD_SIZE)."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args(argv)[0].gpus
    if n < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != n:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; launch with "
                             "--nproc-per-node %d" % (n, os.environ["WORLD_SIZE"], n))
        return
    if n == 1:
        return
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait())
    finally:
        for p in procs:
            if p.poll() is None:  # a rank died: do not leave its peers waiting in a collective
                p.terminate()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or len(codes) != n:
        raise SystemExit("bench.py: rank(s) failed (rank, exit code): %s" % bad)
    raise SystemExit(0)


if __name__ == "__main__":
    _launch_ranks(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CROP = 2 * 6196658176  # trunk + squash + decoder + xyz + heads, MACs x 2
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense (v_mfma_f32_32x32x16_bf16)
P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
               [0.0, 0.0, 1.0, 0.002745884]], np.float32)  # a KITTI P2 (values of training/calib/000000.txt)


def make_inputs(B, npts, rank, device):
    """Seeded synthetic inputs of SURVEY.md 8(d) cfg 2/3 (seeds offset by rank)."""
    def rng(seed):
        return np.random.default_rng(seed + 1000 * rank)
    crops = (rng(1).standard_normal((B, 48, 48, 3), dtype=np.float32) * 50)
    full_feat = np.maximum(rng(2).standard_normal((B, 12, 12, 1024), dtype=np.float32), 0)
    r = rng(3)
    h, w = r.uniform(20, 200, B), r.uniform(20, 200, B)
    y1, x1 = r.uniform(0, 375 - h), r.uniform(0, 1242 - w)
    boxes = np.stack([y1, x1, y1 + h, x1 + w], 1).astype(np.float32)
    view = np.arctan2((boxes[:, 1] + boxes[:, 3]) / 2 - P2[0, 2], P2[0, 0]).astype(np.float32)
    gt = rng(5).standard_normal((B, npts, 3), dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return dict(crops=t(crops), full_feat=t(full_feat), boxes=t(boxes), cam_p=t(P2), view=t(view),
                cls=torch.ones((B, 1), dtype=torch.int32, device=device),
                mean_lwh=t(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                z_off=torch.full((B,), 2.17799973487854, dtype=torch.float32, device=device), gt=t(gt)), \
        dict(crops=crops, full_feat=full_feat, boxes=boxes, view=view, gt=gt)


class MultiStreamStep:
    """The same pass with the batch split into `n` contiguous instance shards, each on its own HIP stream with its
    own scratch: the short, partially filled tail of one shard's kernels overlaps the other shards' kernels
    (instances are independent, so this is the single-GPU form of the multi-GPU sharding)."""

    def __init__(self, net, inp, npts, n):
        self.main = torch.cuda.current_stream()
        self.streams = [torch.cuda.Stream() for _ in range(n)]
        B = inp["crops"].shape[0]
        per_inst = {k for k, v in inp.items() if v.dim() > 0 and v.shape[0] == B and k != "cam_p"}
        self.steps = []
        for i in range(n):
            lo, hi = i * B // n, (i + 1) * B // n
            shard = {k: (v[lo:hi].contiguous() if k in per_inst else v) for k, v in inp.items()}
            self.steps.append(Step(net.clone_with_own_scratch(), shard, npts))

    def __call__(self):
        ev = torch.cuda.Event()
        ev.record(self.main)
        outs = []
        for s, step in zip(self.streams, self.steps):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                outs.append(step())
        for s in self.streams:
            self.main.wait_stream(s)
        return outs


class Step:
    """One hot-path pass; all launches go to torch's current stream through the C ABI."""

    def __init__(self, net, inp, npts):
        from monopsr_amd.tf_ops.nn_distance import tf_nndistance
        self.net, self.inp, self.npts, self.nnd = net, inp, npts, tf_nndistance
        B = inp["crops"].shape[0]
        self.ones = torch.ones((B, npts), dtype=torch.float32, device=inp["crops"].device)

    def forward_net(self):
        i = self.inp
        crop_feat = self.net.trunk(i["crops"])
        fb, _, xyz = self.net.squash_decoder(crop_feat, i["full_feat"], (48, 48), want_feat_map=False)
        out = self.net.heads_fwd(fb, i["boxes"], i["cam_p"], i["view"], i["cls"], i["mean_lwh"], i["z_off"])
        return xyz, out

    def __call__(self):
        xyz, out = self.forward_net()
        B = xyz.shape[0]
        pred = xyz.reshape(B, -1, 3)[:, :self.npts].contiguous()
        with torch.no_grad():
            d1, i1, d2, i2 = self.nnd.nn_distance(pred, self.inp["gt"])
            g1, g2 = self.nnd.nn_distance_grad(pred, self.inp["gt"], self.ones, i1, self.ones, i2)
        return out["centroids"], d1, d2, g1, g2


def conv_replay(net, B):
    """Every conv/FC launch of one step (same shapes, same order) with nothing in between, so that HIP events
    around it time the dominant kernel alone.  Returns (callable, launches per call)."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    dev = net.device
    jobs = []

    def add(part, idx, M_hw, alg_cin=None):
        r = part.records[idx]
        Bq, H, W = M_hw
        x = torch.empty((Bq * H * W * r["cin"],), dtype=torch.float32, device=dev).normal_()
        y = torch.empty((Bq * H * W * r["cout"],), dtype=torch.float32, device=dev)
        jobs.append((x, y, part.blob, r, Bq, H, W, alg_cin or r["cin"]))  # alg_cin: K without zero padding

    tr = net.crop_trunk
    add(tr, 0, (B * 576, 1, 1), 147)
    for k in range(1, tr.n):
        add(tr, k, (B, 12, 12))
    dec = net.decoder
    for k, hw in enumerate([(12, 12), (12, 12), (24, 24), (24, 24), (48, 48), (48, 48), (48, 48)]):
        add(dec, k, (B, hw[0], hw[1]))
    hd = net.heads
    for k in range(1, hd.n):  # img_fc (split-K + reduce kernel) is left out: it is not a pure conv launch
        add(hd, k, (B, 1, 1), {1: 1043, 4: 1060}.get(k))

    # the network entry points hand every layer the scheduling scratch and leave the schedule to the library
    # (split_k = 0); the replay does the same, so it launches the kernels a step launches
    nws = max(lib.mpsr_conv2d_scratch_floats(Bq, H, W, r["cout"]) for _, _, _, r, Bq, H, W, _ in jobs)
    ws = torch.empty((nws,), dtype=torch.float32, device=dev)

    def run():
        s = _lib.stream()
        for x, y, blob, r, Bq, H, W, _ in jobs:
            bias = blob.data_ptr() + 4 * r["b_off"] if r["b_off"] >= 0 else None
            _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), Bq, H, W, r["cin"], blob.data_ptr() + 4 * r["w_off"],
                                                bias, None, y.data_ptr(), r["cout"], r["kh"], r["kw"], r["dilation"],
                                                r["relu"], 0, ws.data_ptr(), nws, s))
    flops = sum(2.0 * Bq * H * W * cin * r["kh"] * r["kw"] * r["cout"] for _, _, _, r, Bq, H, W, cin in jobs)
    # multiply-add FLOPs the library's kernels really issue for these launches: the Winograd kernel (decoder 3x3
    # layers) 16/36 of the direct count, the atrous layers only their in-image taps
    import ctypes
    executed, kinds = 0.0, {0: 0, 1: 0, 2: 0}
    for _, _, _, r, Bq, H, W, _ in jobs:
        kind, ex = ctypes.c_int(0), ctypes.c_double(0.0)
        _lib.check(lib.mpsr_conv2d_plan(Bq, H, W, r["cin"], r["cout"], r["kh"], r["kw"], r["dilation"],
                                        ctypes.byref(kind), ctypes.byref(ex)))
        executed += ex.value
        kinds[kind.value] += 1
    # algorithmic HBM bytes of a launch: its input, weights and output once each
    alg_bytes = sum(4.0 * (Bq * H * W * (r["cin"] + r["cout"]) + r["cout"] * r["kh"] * r["kw"] * r["cin"])
                    for _, _, _, r, Bq, H, W, _ in jobs)
    return run, len(jobs), flops, alg_bytes, executed, kinds


def mfma_box_peak(device):
    """fp32 MFMA rate this board sustains with nothing but v_mfma_f32_32x32x2_f32 (4 waves per SIMD): MI355X boards
    differ in sustained clock, so the roofline fraction is quoted against the 157.3 TFLOP/s datasheet peak AND next to
    this figure measured in the same process."""
    from monopsr_amd import _lib
    import ctypes
    lib = _lib.lib()
    fn = lib.mpsr_debug_mfma_peak
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    out = torch.zeros(4, device=device)
    waves, iters = 4, 2000
    _lib.check(fn(out.data_ptr(), cus, waves, 4, 100, _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        _lib.check(fn(out.data_ptr(), cus, waves, 4, iters, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    return cus * 4 * waves * iters * 16 * 4096.0 * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def emd_object(device, b=256, n=2048):
    """BASELINE config 5's per-GPU share (256 clouds x 2048^2 points) through the EMD loss: the fused path
    (mpsr_emd_loss: 21 passes + 2 loss passes, no match tensor) and the materialising ops next to it."""
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    g = torch.Generator(device=device).manual_seed(6)
    x1 = torch.rand((b, n, 3), device=device, generator=g) * 2 - 1
    x2 = torch.rand((b, n, 3), device=device, generator=g) * 2 - 1

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
    t_f = timeit(lambda: am.emd_loss_fwd_bwd(x1, x2), 5)
    match = am.approx_match(x1, x2)
    t_m = timeit(lambda: am.approx_match(x1, x2), 3)
    t_c = timeit(lambda: am.match_cost(x1, x2, match), 3)
    t_g = timeit(lambda: am.match_cost_grad(x1, x2, match), 3)
    pairs = float(b) * n * n
    exps = (21.0 + 6.0) * pairs  # one exponential per pair and pass, three per pair in each of the two loss passes
    return {"shape": [b, n, n], "workload": "BASELINE cfg5 per-GPU share: approx_match + match_cost + gradients",
            "fused_loss_ms": round(t_f * 1e3, 3), "fused_clouds_per_s": round(b / t_f, 1),
            "fused_exp_per_s": float("%.4g" % (exps / t_f)), "exp_peak_per_s": 9.83e12,
            "fused_exp_frac_of_quarter_rate_peak": round(exps / t_f / 9.83e12, 3),
            "materialising_ops_ms": {"approx_match": round(t_m * 1e3, 3), "match_cost": round(t_c * 1e3, 3),
                                     "match_cost_grad": round(t_g * 1e3, 3)},
            "match_bytes": int(4 * pairs), "approx_match_write_GBps": round(4 * pairs / t_m / 1e9, 1),
            "match_cost_read_GBps": round(4 * pairs / t_c / 1e9, 1),
            "match_cost_grad_read_GBps": round(8 * pairs / t_g / 1e9, 1), "hbm_peak_GBps": 8000,
            "bound": "v_exp_f32 issue (passes), HBM (materialised match)"}


def cpu_baseline(weights, host, sample, npts):
    """The CPU restatement (oracle/net.py on torch CPU, all host cores) + the C Chamfer oracle (1 thread) on the
    first `sample` instances of the same workload.  Reported baseline only."""
    from oracle import net as onet
    from oracle import ops as orc
    if sample <= 0:  # size the sample for roughly 15 s of CPU work from a warm 8-instance probe of the trunk
        sc = "FirstStageFeatureExtractor_crop/resnet_v1_101"
        with torch.no_grad():
            onet.resnet101_block3(torch.from_numpy(host["crops"][:2]), weights, sc)  # warm-up (thread pool, oneDNN)
            t0 = time.perf_counter()
            onet.resnet101_block3(torch.from_numpy(host["crops"][:8]), weights, sc)
        per = (time.perf_counter() - t0) / 8 * 1.6  # trunk is ~64 % of the per-crop work
        sample = int(max(8, min(host["crops"].shape[0], 15.0 / per)))
    s = slice(0, sample)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = onet.instance_path(host["crops"][s], host["full_feat"][s], host["boxes"][s], P2, host["view"][s],
                                 np.ones((sample, 1), np.int32), np.tile(np.array([[3.88, 1.63, 1.53]], np.float32),
                                                                         (sample, 1)),
                                 np.full((sample,), 2.17799973487854, np.float32), weights)
    pred = ref["inst_xyz_map_local"].reshape(sample, -1, 3)[:, :npts].contiguous().numpy()
    d1, i1, d2, i2 = orc.nn_distance(pred, host["gt"][s])
    orc.nn_distance_grad(pred, host["gt"][s], np.ones_like(d1), i1, np.ones_like(d2), i2)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "first %d instances of the same workload: oracle/net.py (PyTorch-CPU fp32 restatement of the "
                      "TF1 graph, %d threads) + oracle C Chamfer fwd/bwd (1 thread), %.1f s" %
                      (sample, torch.get_num_threads(), dt)}, ref


def training_step_object(device, batch, inp, steps=3, warmup=2):
    """ms per training step of `batch` instances on this GPU (crop trunk + decoder + heads trainable, 72.8 M
    parameters in one flat buffer)."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=0), device=device, decoder_bn="batch")
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0)
    sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                  cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                  prop_cen_z_offset=inp["z_off"])
    sample.update(trainer.synthetic_ground_truth(sample, seed=7))
    losses = [float(tr.step(sample)) for _ in range(warmup)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timed = [tr.step(sample) for _ in range(steps)]  # device scalars: no host sync inside the timed region
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses += [float(v) for v in timed]
    return {"ms_per_step": round(dt * 1e3, 2), "crops_per_s": round(batch / dt, 1), "steps": steps,
            "params": int(net.params.numel()), "grad_bytes": int(net.grads.numel() * 4),
            "what": "fwd + configured losses (incl. global-map projection) + bwd + clip + Adam + EMA, fp32, "
                    "decoder BatchNorm on batch statistics; single rank (no all-reduce)",
            "loss_per_step": [round(v, 1) for v in losses],
            "note": "random-initialised weights and synthetic targets: the first Adam steps are a transient (the same "
                    "run reaches 40 % of the initial loss after 10 steps, tools/train_bench.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="instances per GPU")
    ap.add_argument("--points", type=int, default=1024, help="points per cloud for Chamfer")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="instances timed on the host CPU (0 = skip, -1 = size for ~15 s of CPU work)")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="also all-reduce a 100,204,832-float buffer per step (size of the model's gradient)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-emd", action="store_true", help="skip the extra EMD (BASELINE config 5) object")
    ap.add_argument("--no-train-step", action="store_true", help="skip the extra training_step object")
    ap.add_argument("--streams", type=int, default=1,
                    help="split each GPU's batch into this many instance shards on separate HIP streams")
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16x3"],
                    help="contraction arithmetic of the timed run: fp32 (default, the headline) or the opt-in "
                         "split-bfloat16 mode (include/monopsr_hip.h MPSR_MATH_BF16X3)")
    ap.add_argument("--no-allreduce-probe", action="store_true",
                    help="N > 1: skip the extra region that repeats the step with the gradient-sized all-reduce")
    ap.add_argument("--no-fast-mode", action="store_true",
                    help="skip the extra bf16x3_mode measurement appended to a default fp32 run")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MPSR_BENCH_RENDEZVOUS_ONLY"):
        # test hook (tests/test_bench_launch.py): the launch + rendezvous + max-over-ranks plumbing of an N-rank run
        # on CPU (gloo), without the hot path
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(seen, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "max_rank_plus_1": float(seen.item())}),
                  flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # MPSR_BENCH_SHARE_GPU: test hook for 1-GPU boxes -- every rank computes on cuda:0 and the ranks meet over gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing, the N-rank code path does
    share_gpu = bool(os.environ.get("MPSR_BENCH_SHARE_GPU"))
    backend = "gloo" if share_gpu else "nccl"
    if share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local_rank,
                                                                                torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("MPSR_BENCH_FORCE_DIST"):  # the env knob exercises the N > 1 code on one GPU
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = world
    red_dev = device if backend == "nccl" else torch.device("cpu")  # where the timing reductions live

    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    _lib.set_conv_math(args.math)
    weights = W.synthetic_weights(seed=0)
    net = dn.DeviceNet(weights, device=device)
    inp, host = make_inputs(args.batch, args.points, rank, device)
    step = Step(net, inp, args.points) if args.streams <= 1 else MultiStreamStep(net, inp, args.points, args.streams)
    grad_buf = torch.zeros((100204832,), dtype=torch.float32, device=device) if args.allreduce_grads else None

    def one_step():
        out = step()
        if grad_buf is not None and dist is not None:
            dist.all_reduce(grad_buf)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = {
        "metric": "instance-crops/sec (fwd+Chamfer)",
        "value": round(args.batch * n_gpus * args.steps / elapsed, 2),
        "unit": "crops/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.math == "fp32" else "bf16x3 (split-bfloat16 products, f32 accumulate, f32 tensors)",
        "data": "synthetic",
        "config": {"workload": "BASELINE cfg3: batch=%d/GPU synthetic 48x48 crops -> ResNet-101 trunk(block3, os4) + "
                               "squash/map-decoder/xyz + heads fwd, + %d-pt nn_distance Chamfer fwd/bwd" %
                               (args.batch, args.points),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * n_gpus, "points": args.points,
                   "streams_per_gpu": args.streams,
                   "sharding": "instances/%d, no data-path collective" % n_gpus +
                               (" + all-reduce(401 MB synthetic grad buffer)" if args.allreduce_grads else "")},
    }

    if dist is not None and backend == "nccl" and not args.allreduce_grads and not args.no_allreduce_probe:
        # BASELINE config 4 adds "data-parallel RCCL all-reduce" to the sharded step.  The metric path itself has no
        # exchange (value above); this extra region repeats the step with an all-reduce of a buffer the size of the
        # model's fp32 gradient (100,204,832 floats, synthetic contents -- the forward path produces no parameter
        # gradient; the real one is exercised by tools/train_bench.py), launched asynchronously so RCCL overlaps the
        # step's kernels as a trainer's bucketed reduce overlaps backward.
        try:
            gbuf = torch.zeros((100204832,), dtype=torch.float32, device=device)

            def ar_step():
                work = dist.all_reduce(gbuf, async_op=True)
                out = step()
                work.wait()
                return out
            for _ in range(2):
                ar_step()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k2 = max(3, min(10, args.steps))
            for _ in range(k2):
                ar_step()
            torch.cuda.synchronize()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
            result["with_grad_allreduce"] = {
                "value": round(args.batch * n_gpus * k2 / el2, 2), "unit": "crops/s",
                "ms_per_step": round(1e3 * el2 / k2, 3), "allreduce_bytes": int(gbuf.numel() * 4),
                "note": "same step + async RCCL all-reduce(sum) of a gradient-sized synthetic fp32 buffer per step"}
            del gbuf
        except Exception as e:
            result["with_grad_allreduce"] = {"error": repr(e)}

    if rank == 0 and not args.no_roofline:
        # dominant kernel alone: every conv/FC launch of one step, timed with events on the launch stream
        run, launches, flops, alg_bytes, executed, kinds = conv_replay(net, args.batch)
        run()
        torch.cuda.synchronize()
        reps = max(3, min(10, args.steps))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        avg_s = e0.elapsed_time(e1) * 1e-3 / (reps * launches)
        achieved = flops / launches / avg_s / 1e12
        # HBM traffic and MFMA-busy cycles cannot be counted from inside the process: they come from the newest
        # committed rocprofv3 --pmc passes over this same command (tools/collect_profiles.sh -> profiles/
        # rNN_pmc_traffic.json), quoted with the round and commit they were collected at
        traffic = busy = src = None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(path) as f:
                    pm = json.load(f)
                traffic = round(pm["hbm_bytes_per_launch"])
                busy = pm.get("mfma_busy")
                src = {"file": os.path.relpath(path, ROOT), "round": pm.get("round"), "commit": pm.get("commit"),
                       "collected": pm.get("collected")}
                break
            except Exception:
                pass
        peak = PEAK_F32_MFMA_TFLOPS if args.math == "fp32" else PEAK_BF16_MFMA_TFLOPS
        kname = "conv_igemm_kernel + wino_conv_kernel (fp32 MFMA 32x32x2: implicit GEMM; Winograd F(2x2,3x3) for the " \
                "decoder's 3x3 layers)" if args.math == "fp32" else \
            "conv_igemm_kernel (3x bf16 MFMA 32x32x16 per fp32 product; achieved counts each product once)"
        result["roofline"] = {"bound": "mfma", "kernel": kname,
                              "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                              "frac": round(achieved / peak, 4), "traffic": traffic if args.math == "fp32" else None,
                              "launches_per_step": launches, "avg_launch_us": round(avg_s * 1e6, 2),
                              "flops_per_launch": round(flops / launches),
                              "algorithmic_bytes": round(alg_bytes / launches)}
        if args.math == "fp32":
            # `achieved` counts the ALGORITHMIC multiply-adds of the direct convolution (the contract's definition);
            # the decoder's 3x3 layers run as Winograd F(2x2,3x3) (16 products where the direct form has 36) and the
            # atrous layers skip their out-of-image taps, so the matrix pipes issue fewer: `executed` is what they do
            result["roofline"]["executed"] = round(executed / launches / avg_s / 1e12, 2)
            result["roofline"]["executed_frac"] = round(executed / launches / avg_s / 1e12 / peak, 4)
            result["roofline"]["executed_flops_per_launch"] = round(executed / launches)
            result["roofline"]["launch_kinds"] = {"implicit_gemm": kinds[0], "winograd_f2x2_3x3": kinds[1],
                                                  "direct_narrow": kinds[2]}
            result["roofline"]["note"] = ("achieved / frac count the ALGORITHMIC direct-convolution FLOPs (SURVEY 8(d)), "
                                          "so they can exceed the peak: the 4 Winograd launches issue 16/36 of theirs and "
                                          "the atrous layers skip out-of-image taps; executed / executed_frac and "
                                          "mfma_busy (PMC) are the utilisation of the matrix pipes")
            if traffic:
                result["roofline"]["traffic_over_algorithmic"] = round(traffic / (alg_bytes / launches), 3)
                result["roofline"]["mfma_busy"] = busy
                result["roofline"]["traffic_source"] = src
            try:
                box = mfma_box_peak(device)
                result["roofline"]["peak_measured_on_this_board"] = round(box, 1)
                result["roofline"]["frac_of_measured"] = round(achieved / box, 4)
            except Exception as e:
                result["roofline"]["peak_measured_on_this_board"] = repr(e)
    if rank == 0 and not args.no_roofline:
        # the Chamfer op alone, as the north star asks ("achieved HBM GB/s on nn_distance"): algorithmic bytes =
        # b*(n+m)*20 forward (12 read + 8 written per point), b*(n+m)*32 backward; the kernel is VALU-bound, so the
        # pair-evaluation rate is reported next to it
        try:
            from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd
            b, n = args.batch, args.points
            c1 = torch.randn((b, n, 3), device=device)
            c2 = torch.randn((b, n, 3), device=device)
            ones = torch.ones((b, n), device=device)
            d1, i1, d2, i2 = nnd.nn_distance(c1, c2)
            nnd.nn_distance_grad(c1, c2, ones, i1, ones, i2)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            reps = 20
            ev[0].record()
            for _ in range(reps):
                nnd.nn_distance(c1, c2)
            ev[1].record()
            for _ in range(reps):
                nnd.nn_distance_grad(c1, c2, ones, i1, ones, i2)
            ev[2].record()
            torch.cuda.synchronize()
            tf_, tb_ = ev[0].elapsed_time(ev[1]) * 1e-3 / reps, ev[1].elapsed_time(ev[2]) * 1e-3 / reps
            result["nn_distance"] = {
                "shape": [b, n, n], "fwd_us": round(tf_ * 1e6, 1), "bwd_us": round(tb_ * 1e6, 1),
                "fwd_alg_GBps": round(b * 2 * n * 20 / tf_ / 1e9, 1), "bwd_alg_GBps": round(b * 2 * n * 32 / tb_ / 1e9, 1),
                "hbm_peak_GBps": 8000, "fwd_pair_evals_per_s": float("%.4g" % (2.0 * b * n * n / tf_)),
                "bound": "valu (fwd), hbm/lds (bwd)"}
        except Exception as e:
            result["nn_distance"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline and not args.no_emd:
        try:
            result["emd"] = emd_object(device)
        except Exception as e:
            result["emd"] = {"error": repr(e)}
    if rank == 0 and n_gpus == 1 and args.math == "fp32" and not args.no_fast_mode:
        # the same step in the opt-in bf16x3 contraction mode, with the drift of its outputs against the fp32 run
        # on the same inputs (NOT the headline: `value` above is fp32)
        try:
            ref = None
            if isinstance(step, Step):
                xyz32, out32 = step.forward_net()
                ref = (xyz32.clone(), out32["centroids"].clone())
            _lib.set_conv_math("bf16x3")
            for _ in range(2):
                one_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = max(3, min(10, args.steps))
            for _ in range(k):
                one_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            fast = {"value": round(args.batch * k / dt, 2), "unit": "crops/s", "ms_per_step": round(1e3 * dt / k, 3),
                    "arithmetic": "hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 tensors"}
            if ref is not None:
                xyz, out = step.forward_net()
                rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
                fast["max_rel_drift_vs_fp32"] = {"inst_xyz_map_local": float("%.3g" % rel(xyz, ref[0])),
                                                 "centroids": float("%.3g" % rel(out["centroids"], ref[1]))}
            result["bf16x3_mode"] = fast
        except Exception as e:  # the side measurement must never cost the headline line
            result["bf16x3_mode"] = {"error": repr(e)}
        finally:
            _lib.set_conv_math("fp32")
    if rank == 0 and n_gpus == 1 and args.math == "fp32" and not args.no_train_step and not args.no_roofline:
        # one data-parallel TRAINING step of the same 256 instances (forward, the reference's configured
        # loss set, backward, per-variable clip, Adam + moving average; map-decoder BatchNorm on batch statistics),
        # timed like tools/train_bench.py.  A side measurement: never part of `value`.
        try:
            result["training_step"] = training_step_object(device, args.batch, inp)
        except Exception as e:
            result["training_step"] = {"error": repr(e)}
    if rank == 0 and args.cpu_sample != 0 and n_gpus == 1:
        result["cpu_baseline"], _ = cpu_baseline(weights, host, min(args.cpu_sample, args.batch), args.points)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
