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
    import signal
    procs = []
    got_signal = []

    def on_signal(signum, frame):  # a SIGTERM / SIGINT to the launcher goes to the ranks: none is left behind
        got_signal.append(signum)
    # (installed BEFORE the first rank exists: a signal that arrives while the ranks are being started must not take the
    # launcher down by the default action and orphan them)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
    # cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument` (the task environment exports
    # it already; a value the caller set is kept)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))

    def stop_all(sig=signal.SIGTERM, grace=10.0):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass
        t_end = time.monotonic() + grace
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()  # the exact children started above, never a pattern
                p.wait()

    # poll ALL ranks: the first one that exits non-zero (or a signal to this process) ends the others, which would
    # otherwise sit in the rendezvous or in an RCCL collective until its timeout
    codes = [None] * n
    bad = []
    while any(c is None for c in codes) and not bad and not got_signal:
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0):
                    bad.append((r, codes[r]))
        if any(c is None for c in codes) and not bad:
            time.sleep(0.05)
    if bad or got_signal:
        stop_all()
        if got_signal:
            raise SystemExit("bench.py: launcher received signal %d; ranks stopped" % got_signal[0])
        raise SystemExit("bench.py: rank(s) failed (rank, exit code): %s; the other ranks were stopped" % bad)
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

    overlap_heads = False  # DeviceNet.forward_instances: True = the FC heads on a second stream next to the map decoder

    def forward_net(self):
        i = self.inp
        return self.net.forward_instances(i["crops"], i["full_feat"], i["boxes"], i["cam_p"], i["view"], i["cls"],
                                          i["mean_lwh"], i["z_off"], overlap_heads=self.overlap_heads)

    def __call__(self):
        xyz, out = self.forward_net()
        B = xyz.shape[0]
        pred = xyz.reshape(B, -1, 3)[:, :self.npts].contiguous()
        with torch.no_grad():
            d1, i1, d2, i2 = self.nnd.nn_distance(pred, self.inp["gt"])
            g1, g2 = self.nnd.nn_distance_grad(pred, self.inp["gt"], self.ones, i1, self.ones, i2)
        return out["centroids"], d1, d2, g1, g2


def conv_replay(net, B):
    """Every matrix-pipe launch of one step -- same kernels, same variants, same order -- with nothing in between, so
    that HIP events around it time the dominant kernel family alone: the trunk's 94 layers (conv3 / squash launches WITH
    their residual operand), the decoder's layers as mpsr_squash_decoder_plan says the step runs them (conv2_1 / conv3_1
    as their tap GEMMs, 1152 columns per launch: the gather kernels that follow are vector-ALU work and are not
    replayed), the xyz head, and the seven head FCs incl. img_fc (stream-K).  Returns (callable, launches, algorithmic
    FLOPs, algorithmic bytes, executed FLOPs, launches by kind)."""
    import ctypes
    from monopsr_amd import _lib
    from monopsr_amd.core import weights as W
    lib = _lib.lib()
    dev = net.device
    jobs = []

    def plan(Bq, H, Wd, cin, cout, kh, kw, dil):
        kind, ex = ctypes.c_int(0), ctypes.c_double(0.0)
        _lib.check(lib.mpsr_conv2d_plan(Bq, H, Wd, cin, cout, kh, kw, dil, ctypes.byref(kind), ctypes.byref(ex)))
        return kind.value, ex.value

    # Operands with the RESIDENCY of the step's own (r06; review weak 9: the replay ran ~5 % slower than the same launches
    # in situ): every launch reads what the launch in front of it just wrote -- the trunk's real data flow, bottleneck by
    # bottleneck, with the unit's input (or its projection shortcut) as conv3's residual -- out of a small rotating pool
    # of buffers per size, like the network entry points' ping-pong scratch; a fresh relu(N(0,1)) tensor stands in only
    # where the producing kernel is not a matrix launch (the pooled root output is a view of the root's own result, the
    # full-image feature crop, the two upsampling gathers' results, the heads' concat rows).
    pool = {}

    def out_buffer(n):
        ring = pool.setdefault(n, [[], 0])
        if len(ring[0]) < 4:
            ring[0].append(torch.empty((n,), dtype=torch.float32, device=dev))
            return ring[0][-1]
        ring[1] = (ring[1] + 1) % 4
        return ring[0][ring[1]]

    def stand_in(n, raw=False):
        x = torch.empty((n,), dtype=torch.float32, device=dev).normal_()
        return x if raw else x.clamp_(min=0)

    def add(part, idx, M_hw, alg_cin=None, x=None, res=None, override=None):
        """-> the launch's output tensor (the last part's for a tap GEMM)."""
        r = part.records[idx]
        Bq, H, Wd = M_hw
        n_in = Bq * (override[0] * override[1] if override is not None else H * Wd) * r["cin"]
        if x is None:
            x = stand_in(n_in)
        assert x.numel() >= n_in, (idx, x.numel(), n_in)
        bias = part.blob.data_ptr() + 4 * r["b_off"] if r["b_off"] >= 0 else None
        alg = 2.0 * Bq * H * Wd * (alg_cin or r["cin"]) * r["kh"] * r["kw"] * r["cout"]
        alg_b = 4.0 * (Bq * H * Wd * (r["cin"] + r["cout"] * (2 if res is not None else 1)) +
                       r["cout"] * r["kh"] * r["kw"] * r["cin"])
        if override is not None:  # the layer runs as `parts` 1x1 GEMMs of the source map with 1152 outputs each
            sh, sw, parts = override
            wgt = torch.randn((parts * 1152 * r["cin"],), dtype=torch.float32, device=dev) * 0.02
            assert x.numel() >= Bq * sh * sw * r["cin"]
            for pi in range(parts):
                y = torch.empty((Bq * sh * sw * 1152,), dtype=torch.float32, device=dev)
                jobs.append(dict(x=x, y=y, w=wgt.data_ptr() + 4 * pi * 1152 * r["cin"], bias=None, res=None, keep=wgt,
                                 shape=(Bq, sh, sw, r["cin"], 1152, 1, 1, 1, 0), alg=alg / parts, alg_b=alg_b / parts,
                                 kind=7, ex=2.0 * Bq * sh * sw * r["cin"] * 1152))
            return y
        y = out_buffer(Bq * H * Wd * r["cout"])
        assert y.data_ptr() != x.data_ptr() and (res is None or res.data_ptr() != y.data_ptr())
        kind, ex = plan(Bq, H, Wd, r["cin"], r["cout"], r["kh"], r["kw"], r["dilation"])
        jobs.append(dict(x=x, y=y, w=part.blob.data_ptr() + 4 * r["w_off"], bias=bias, res=res, keep=None,
                         shape=(Bq, H, Wd, r["cin"], r["cout"], r["kh"], r["kw"], r["dilation"], r["relu"]),
                         alg=alg, alg_b=alg_b, kind=kind, ex=ex))
        return y

    tr = net.crop_trunk
    roles = [sp["role"] for sp in W.scaled_trunk_specs(W.CROP_SCOPE, net.width_div)]
    y0 = add(tr, 0, (B * 576, 1, 1), 147, x=stand_in(B * 576 * tr.records[0]["cin"], raw=True))
    cur, sc, a = y0, None, None  # (the 3x3 / 2 max-pool keeps a quarter of the root's pixels: a view of its result)
    for k in range(1, tr.n):
        if roles[k] == "shortcut":
            sc = add(tr, k, (B, 12, 12), x=cur)
        elif roles[k] == "conv1":
            a = add(tr, k, (B, 12, 12), x=cur)
        elif roles[k] == "conv2":
            a = add(tr, k, (B, 12, 12), x=a)
        else:  # conv3: + the unit's input (or its projection), ReLU
            cur = add(tr, k, (B, 12, 12), x=a, res=sc if sc is not None else cur)
            sc = None
    dec = net.decoder
    kinds7, ex7 = (ctypes.c_int * 7)(), (ctypes.c_double * 7)()
    _lib.check(lib.mpsr_squash_decoder_plan(B, 12, 12, 48, 48, dec.layers, dec.n, kinds7, ex7))
    prev = None
    for k, hw in enumerate([(12, 12), (12, 12), (24, 24), (24, 24), (48, 48), (48, 48), (48, 48)]):
        if kinds7[k] == 7:  # tap GEMM on the source map (half the size), one launch per 128 output channels
            add(dec, k, (B, hw[0], hw[1]), x=prev, override=(hw[0] // 2, hw[1] // 2, dec.records[k]["cout"] // 128))
            prev = None  # (its gather is vector work and is not replayed: the next layer reads a stand-in)
        else:
            # squash = two GEMMs over the K halves: crop features, then the full-image feature crop + the first as residual
            # (k = 2 / 4 outside the tap-GEMM form read a bilinear upsampling -- a vector kernel's result: stand-in)
            xin = cur if k == 0 else None if k in (1, 2, 4) else prev
            prev = add(dec, k, (B, hw[0], hw[1]), x=xin, res=prev if k == 1 else None)
            jobs[-1]["kind"], jobs[-1]["ex"] = kinds7[k], ex7[k]
    hd = net.heads
    prev = None
    for k in range(hd.n):
        chained = k in (2, 3, 5, 6)  # layers whose input IS the previous launch's output (the others read concat rows)
        prev = add(hd, k, (B, 1, 1), {1: 1043, 4: 1060}.get(k), x=prev if chained else None)

    # the network entry points hand every layer the scheduling scratch and leave the schedule to the library
    # (split_k = 0); the replay does the same, so it launches the kernels a step launches
    nws = max(lib.mpsr_conv2d_scratch_floats(j["shape"][0], j["shape"][1], j["shape"][2], j["shape"][4]) for j in jobs)
    ws = torch.empty((nws,), dtype=torch.float32, device=dev)

    def run():
        s = _lib.stream()
        for j in jobs:
            Bq, H, Wd, cin, cout, kh, kw, dil, relu = j["shape"]
            _lib.check(lib.mpsr_conv2d_nhwc_f32(j["x"].data_ptr(), Bq, H, Wd, cin, j["w"], j["bias"],
                                                j["res"].data_ptr() if j["res"] is not None else None, j["y"].data_ptr(),
                                                cout, kh, kw, dil, relu, 0, ws.data_ptr(), nws, s))
    flops = sum(j["alg"] for j in jobs)
    executed = sum(j["ex"] for j in jobs)
    kinds = {}
    for j in jobs:
        kinds[j["kind"]] = kinds.get(j["kind"], 0) + 1
    alg_bytes = sum(j["alg_b"] for j in jobs)
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


def roofline_object(net, args, device, ms_per_step):
    """The dominant kernel family alone: every conv/FC launch of one step, timed with events on the launch stream.
    `achieved` / `frac` are what the matrix pipes EXECUTE (mpsr_conv2d_plan: the Winograd launches issue 16/36 --
    36/144 for F(4x4,3x3), 16/81 for block3's atrous sub-grids as zero-padded tiles, 25/81 for F(3x3,3x3) with halos -- of
    their direct-convolution products and
    the other atrous layers skip out-of-image taps), so
    0 < frac <= 1 and it is comparable with the PMC's MFMA-busy fraction; the ALGORITHMIC direct-convolution rate of
    SURVEY 8(d) (12.393 GFLOP per crop) is next to it as algorithmic_achieved / algorithmic_frac and may exceed 1."""
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
    pass_s = e0.elapsed_time(e1) * 1e-3 / reps
    avg_s = pass_s / launches
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
    fp32 = args.math == "fp32"
    peak = PEAK_F32_MFMA_TFLOPS if fp32 else PEAK_BF16_MFMA_TFLOPS
    kname = "fp32 MFMA 32x32x2 convolution / FC kernels: pw_conv_kernel (wide 1x1 layers and the decoder's tap GEMMs), " \
            "wino3z_conv_kernel (block3's atrous 3x3 layers: every 3x3 pixel sub-grid as ONE zero-padded tile in sixteen products, one wave owning all 16 positions of its tile block), wino3h_conv_kernel (F(3x3,3x3) on tiles with halos: the 3x3 layers of blocks 1-2), wino4_conv_kernel (F(4x4,3x3), decoder conv2_2 / " \
            "conv3_2), conv_igemm_kernel / conv_sk_kernel (blocks 1-2, root, img_fc), fc_rows_kernel, " \
            "conv3x3_narrow_mfma_kernel" if fp32 else \
        "conv_igemm_kernel (3x bf16 MFMA 32x32x16 per fp32 product; achieved counts each product once)"
    alg = flops / launches / avg_s / 1e12
    exe = executed / launches / avg_s / 1e12 if fp32 else alg
    out = {"bound": "mfma", "kernel": kname, "achieved": round(exe, 2), "peak": peak, "unit": "TFLOP/s",
           "frac": round(exe / peak, 4), "traffic": traffic if fp32 else None,
           "launches_per_step": launches, "avg_launch_us": round(avg_s * 1e6, 2),
           "flops_per_launch": round((executed if fp32 else flops) / launches),
           "algorithmic_flops_per_launch": round(flops / launches),
           "algorithmic_achieved": round(alg, 2), "algorithmic_frac": round(alg / peak, 4),
           "algorithmic_bytes": round(alg_bytes / launches),
           "kernel_ms_per_step": round(pass_s * 1e3, 3),
           "kernel_ms_per_step_le_ms_per_step": bool(pass_s * 1e3 <= ms_per_step),
           "replay": "the step's matrix-pipe launches back to back -- the variants the step launches: residual operands "
                     "on conv3 / squash, the decoder's tap GEMMs, img_fc on stream-K -- in the step's own data flow: every "
                     "launch reads what the launch in front of it wrote (bottleneck by bottleneck, rotating buffers like "
                     "the entry points' ping-pong scratch), relu(N(0,1)) stand-ins only where the producer is a vector "
                     "kernel; not replayed: the vector-ALU kernels between them (im2col, pools, the two upsampling "
                     "gathers, head glue, Chamfer) -- `in_situ` has both families measured inside the timed loop"}
    if fp32:
        out["launch_kinds"] = {"implicit_gemm": kinds.get(0, 0), "winograd_f2x2_3x3": kinds.get(1, 0),
                               "direct_narrow": kinds.get(2, 0), "winograd_f4x4_3x3": kinds.get(3, 0),
                               "winograd_f3x3_3x3_subgrid_tiles": kinds.get(4, 0),
                               "pointwise_persistent": kinds.get(5, 0), "fc_few_rows": kinds.get(6, 0),
                               "upsampled_conv_tap_gemm": kinds.get(7, 0)}
        out["note"] = ("achieved / frac = multiply-adds the matrix pipes execute per launch / launch time (/ peak); "
                       "algorithmic_* count the direct-convolution FLOPs of SURVEY 8(d) for the same launches; "
                       "mfma_busy is the PMC's SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE from the quoted collection -- the latter "
                       "advances at a fixed ~2.45 GHz, so it is a fraction of TIME at the nominal clock like frac; the step "
                       "is power-limited (~1.3 kW package power: the shader clock sits near 2.0-2.1 GHz under the dense "
                       "kernels), which both figures contain")
        if traffic:
            out["traffic_over_algorithmic"] = round(traffic / (alg_bytes / launches), 3)
            out["mfma_busy"] = busy
            out["traffic_source"] = src
        try:
            box = mfma_box_peak(device)
            out["peak_measured_on_this_board"] = round(box, 1)
            out["frac_of_measured"] = round(exe / box, 4)
        except Exception as e:
            out["peak_measured_on_this_board"] = repr(e)
    return out


MATRIX_KERNELS = ("pw_conv_kernel", "wino3z_conv_kernel", "wino3h_conv_kernel", "wino3w_conv_kernel", "wino3_conv_kernel",
                  "wino3p_conv_kernel", "wino4_conv_kernel", "wino4s_conv_kernel", "wino_conv_kernel", "wino_conv8_kernel",
                  "conv_igemm_kernel", "conv_sk_kernel", "fc_rows_kernel", "conv3x3_narrow_mfma_kernel")


def in_situ_object(one_step, executed_flops_per_step, ms_per_step, steps=4, expect_matrix_launches=None):
    """Per-kernel durations of the TIMED step itself (not of a replay): `steps` more steps of the same loop under
    torch.profiler (roctracer activity records of every kernel this process launches, the library's included), summed
    by kernel family.  matrix_ms_per_step + vector_ms_per_step <= ms_per_step must hold: durations exclude the gaps
    between launches.  This is what answers "how busy are the matrix pipes IN the step"; the replay above (HIP events,
    as the measurement contract asks) runs the same launches on other buffers and is a few percent slower."""
    from torch.profiler import ProfilerActivity, profile
    one_step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        one_step()  # (the tracer's own start-up falls on this one)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        torch.cuda.synchronize()
        traced_ms = 1e3 * (time.perf_counter() - t0) / steps
    steps += 1  # every launch recorded, the untimed first pass included
    fam = {}
    for ev in prof.key_averages():
        us = None
        for attr in ("self_device_time_total", "device_time_total", "self_cuda_time_total", "cuda_time_total"):
            v = getattr(ev, attr, None)
            if v:
                us = float(v)
                break
        if not us:
            continue
        name = ev.key
        short = next((k for k in MATRIX_KERNELS if k in name), None)
        if short is not None and "<" in name:  # keep the template arguments that tell the variants apart
            short += name[name.index("<"):name.index(">") + 1] if ">" in name else ""
        if short is None:  # "void (anonymous namespace)::name<...>(args)" -> name
            base = name.replace("(anonymous namespace)::", "").replace("void ", "")
            base = base.split("(")[0].split("<")[0].split("::")[-1].strip() or name[:60]
        key = ("matrix", short) if short is not None else ("vector", base[:60])
        d = fam.setdefault(key, [0.0, 0])
        d[0] += us
        d[1] += int(ev.count)
    if not fam:
        raise RuntimeError("the profiler returned no device activity")
    mat = sum(v[0] for k, v in fam.items() if k[0] == "matrix") / steps / 1e3
    vec = sum(v[0] for k, v in fam.items() if k[0] == "vector") / steps / 1e3
    nmat = sum(v[1] for k, v in fam.items() if k[0] == "matrix") / steps
    # (under another tracer -- rocprofv3 around this process -- the records are incomplete: say so instead of dividing)
    if (expect_matrix_launches is not None and abs(nmat - expect_matrix_launches) > 0.5) or \
            not (0.2 * ms_per_step < mat + vec < 2.0 * ms_per_step):
        raise RuntimeError("incomplete kernel records (%.1f matrix launches per step, %.3f ms of records for a %.3f ms "
                           "step): another tracer owns the device activity" % (nmat, mat + vec, ms_per_step))
    top = sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]
    return {"how": "torch.profiler (roctracer kernel records) over %d further steps of the timed loop.  A record spans "
                   "dispatch to completion and consecutive records overlap by the next launch's ramp-up, so their sum "
                   "exceeds even the traced steps' own wall clock by a few percent (records_over_traced_wall): the "
                   "absolute times bound the kernels from above (frac_in_situ from below), the SHARES are what the "
                   "records measure; *_scaled_to_untraced applies the matrix share to the untraced ms_per_step (no "
                   "gaps assumed: an upper bound on the fraction).  The event-timed replay (roofline.frac) sits between "
                   "the two: its launches run back to back without the vector kernels' pauses, i.e. at the lower clock "
                   "of an uninterrupted power-limited stream, which is why kernel_ms_per_step + vector_ms_per_step can "
                   "exceed ms_per_step by 2-3 %%" % steps,
            "matrix_ms_per_step": round(mat, 3), "vector_ms_per_step": round(vec, 3),
            "sum_ms_per_step": round(mat + vec, 3), "traced_ms_per_step": round(traced_ms, 3),
            "ms_per_step": ms_per_step, "records_over_traced_wall": round((mat + vec) / traced_ms, 4),
            "matrix_share_of_kernel_time": round(mat / (mat + vec), 4),
            "matrix_ms_per_step_scaled_to_untraced": round(ms_per_step * mat / (mat + vec), 3),
            "matrix_launches_per_step": round(sum(v[1] for k, v in fam.items() if k[0] == "matrix") / steps, 1),
            "vector_launches_per_step": round(sum(v[1] for k, v in fam.items() if k[0] == "vector") / steps, 1),
            "frac_in_situ": round(executed_flops_per_step / (mat * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "frac_in_situ_scaled_to_untraced": round(executed_flops_per_step / (ms_per_step * mat / (mat + vec) * 1e-3)
                                                     / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "kernels_ms_per_step": {("%s:%s" % k): [round(v[0] / steps / 1e3, 3), round(v[1] / steps, 1)]
                                    for k, v in top}}


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
    culled = None
    try:  # the opt-in level-culling form of the fused loss (include/monopsr_hip.h, mpsr_emd_loss_temp_floats), same clouds
        from monopsr_amd import _lib
        _lib.lib().mpsr_debug_set_emd_cull(1)
        culled = timeit(lambda: am.emd_loss_fwd_bwd(x1, x2), 5)
    except Exception:
        pass
    finally:
        _lib.lib().mpsr_debug_set_emd_cull(0)
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
            "level_culling_opt_in": None if culled is None else {
                "fused_loss_ms": round(culled * 1e3, 3),
                "note": "Morton-sorted clouds, far 32-point chunks of the four steepest levels skipped (exactly zero "
                        "terms); off by default: the sorted summation order moves isolated gradient elements by up to "
                        "~1e-3 of the largest (tests/test_ops_gpu.py::test_emd_level_culling_skips_only_exact_zeros)"},
            "materialising_ops_ms": {"approx_match": round(t_m * 1e3, 3), "match_cost": round(t_c * 1e3, 3),
                                     "match_cost_grad": round(t_g * 1e3, 3)},
            "match_bytes": int(4 * pairs), "approx_match_write_GBps": round(4 * pairs / t_m / 1e9, 1),
            "match_cost_read_GBps": round(4 * pairs / t_c / 1e9, 1),
            "match_cost_grad_read_GBps": round(8 * pairs / t_g / 1e9, 1), "hbm_peak_GBps": 8000,
            "bound": "v_exp_f32 issue (passes), HBM (materialised match)"}


def ops_at_reference_shape_object(device, b=32, n=2304):
    """The two point-cloud ops at the shape the REFERENCE evaluates them on: (32, 2304, 3) clouds -- 32 boxes per image
    (configs/monopsr_model_000.yaml:14-17), every pixel of the 48 x 48 xyz map a point
    (/root/reference/src/monopsr/core/models/monopsr/monopsr_model.py:1127-1164: masked maps reshaped to (B, 2304, 3), then
    approx_match + match_cost and nn_distance).  Parity at this shape: tests/test_ops_gpu.py (2304^2 against the oracle)."""
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd
    g = torch.Generator(device=device).manual_seed(16)
    x1 = torch.rand((b, n, 3), device=device, generator=g) * 2 - 1
    x2 = torch.rand((b, n, 3), device=device, generator=g) * 2 - 1
    ones = torch.ones((b, n), device=device)

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
    d1, i1, d2, i2 = nnd.nn_distance(x1, x2)
    t_nf = timeit(lambda: nnd.nn_distance(x1, x2), 50)
    t_nb = timeit(lambda: nnd.nn_distance_grad(x1, x2, ones, i1, ones, i2), 50)
    match = am.approx_match(x1, x2)
    t_m = timeit(lambda: am.approx_match(x1, x2), 10)
    t_c = timeit(lambda: am.match_cost(x1, x2, match), 10)
    t_g = timeit(lambda: am.match_cost_grad(x1, x2, match), 10)
    t_f = timeit(lambda: am.emd_loss_fwd_bwd(x1, x2), 10)
    pairs = float(b) * n * n
    return {"shape": [b, n, n],
            "workload": "the reference's evaluation shape: 32 boxes x 2304-point clouds (every pixel of the 48x48 map)",
            "nn_distance": {"fwd_us": round(t_nf * 1e6, 1), "bwd_us": round(t_nb * 1e6, 1),
                            "fwd_alg_GBps": round(b * 2 * n * 20 / t_nf / 1e9, 1),
                            "bwd_alg_GBps": round(b * 2 * n * 32 / t_nb / 1e9, 1),
                            "fwd_pair_evals_per_s": float("%.4g" % (2.0 * pairs / t_nf)),
                            "clouds_per_s_fwd_bwd": round(b / (t_nf + t_nb), 1)},
            "emd": {"approx_match_ms": round(t_m * 1e3, 3), "match_cost_ms": round(t_c * 1e3, 3),
                    "match_cost_grad_ms": round(t_g * 1e3, 3),
                    "approx_match_plus_match_cost_clouds_per_s": round(b / (t_m + t_c), 1),
                    "fused_loss_fwd_bwd_ms": round(t_f * 1e3, 3), "fused_clouds_per_s": round(b / t_f, 1),
                    "match_bytes": int(4 * pairs), "approx_match_write_GBps": round(4 * pairs / t_m / 1e9, 1)},
            "hbm_peak_GBps": 8000,
            "note": "32 clouds are 32 workgroup columns of the EMD passes and 4.5 workgroups per cloud of the Chamfer "
                    "search: the chip is partly filled at this shape (the cfg3 / cfg5 objects above are the filled ones)"}


def cfg5_step_object(step, inp, device, steps):
    """BASELINE config 5 as ONE timed step on this GPU: the forward pass of the batch's crops (trunk, squash / decoder /
    xyz, heads) + the approximate-EMD loss with its gradients (mpsr_emd_loss: approx_match + match_cost + match_cost_grad
    fused, reference losses_custom.py:135-165) between the first 2048 points of every predicted N x 3 cloud and a
    2048-point U(-1,1) ground-truth cloud.  A side measurement: `value` is config 3."""
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    B = inp["crops"].shape[0]
    g = torch.Generator(device=device).manual_seed(7)
    gt = torch.rand((B, 2048, 3), device=device, generator=g) * 2 - 1

    def one():
        xyz, out = step.forward_net()
        pred = xyz.reshape(B, -1, 3)[:, :2048].contiguous()
        cost, g1, g2 = am.emd_loss_fwd_bwd(pred, gt)
        return out["centroids"], cost, g1
    for _ in range(2):
        one()
    torch.cuda.synchronize()
    k = max(3, min(10, steps))
    t0 = time.perf_counter()
    for _ in range(k):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    return {"workload": "BASELINE cfg5 per-GPU share: batch=%d crops fwd + 2048-pt approxmatch EMD loss fwd/bwd, one step"
                        % B, "ms_per_step": round(dt * 1e3, 3), "crops_per_s": round(B / dt, 1), "steps": k,
            "points": 2048, "emd": "mpsr_emd_loss (fused: no match tensor), device semantics"}


def full_image_path_object(device, boxes=32, images=8, reps=3):
    """SURVEY 8(f) row 1 / the reference's own step shape (configs/monopsr_model_000.yaml:14-17): ONE 375 x 1242 image +
    32 proposal boxes in -> preprocess, proposal crops, crop trunk AND full-image trunk (on a second stream), feature
    crop + pool, squash / decoder / xyz, heads out (MonoPSRModel.build); and the crop-only step at the same batch of 32."""
    from monopsr_amd.core import config_utils
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = dn.DeviceNet(weights, device=device, full_trunk=True)
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, "test")
    rng = np.random.default_rng(0)
    H, Wd, B = 375, 1242, boxes
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    samples = []
    for _ in range(images):
        h, w = rng.uniform(20, 200, B), rng.uniform(20, 200, B)
        y1, x1 = rng.uniform(0, H - 1 - h), rng.uniform(0, Wd - 1 - w)
        bx = np.stack([y1, x1, y1 + h, x1 + w], 1).astype(np.float32)
        samples.append(dict(rgb_image=t(rng.integers(0, 256, (H, Wd, 3)).astype(np.float32)), boxes_2d=t(bx),
                            boxes_2d_norm=t(bx / np.array([H, Wd, H, Wd], np.float32)), cam_p=t(P2),
                            est_view_angs=t(rng.uniform(-0.6, 0.6, B).astype(np.float32)),
                            class_indices=torch.ones((B, 1), dtype=torch.int32, device=device),
                            mean_lwh=t(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                            prop_cen_z_offset=torch.full((B,), 2.178, device=device)))

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * n)
    ms = 1e3 * timed(lambda: [model.build(dict(sm)) for sm in samples], images)
    # the same images in ONE pass (MonoPSRModel.build_batch: one full-image trunk call with batch N, box_ind-routed crops,
    # one crop-trunk / decoder / heads call over all N x 32 boxes)
    per_call = {"1": {"ms_per_image": round(ms, 3), "images_per_s": round(1e3 / ms, 1)}}
    for n in (2, 4, 8):
        if n > images:
            break
        try:
            msn = 1e3 * timed(lambda: model.build_batch([dict(sm) for sm in samples[:n]]), n)
            per_call[str(n)] = {"ms_per_image": round(msn, 3), "images_per_s": round(1e3 / msn, 1)}
        except Exception as e:  # noqa: BLE001
            per_call[str(n)] = {"error": repr(e)[:200]}
    inp32, _ = make_inputs(B, 1024, 0, device)
    st = Step(net, inp32, 1024)
    ms32 = 1e3 * timed(lambda: [st() for _ in range(images)], images)
    gflop = 2 * 167.1 + B * 12.393  # full-image trunk + per-crop path, algorithmic
    return {"workload": "375x1242 image + %d boxes -> both trunks, feature crop, squash / decoder / xyz, heads" % B,
            "ms_per_image": round(ms, 3), "images_per_s": round(1e3 / ms, 1), "crops_per_s": round(B * 1e3 / ms, 1),
            "algorithmic_GFLOP_per_image": round(gflop, 1), "algorithmic_TFLOP_per_s": round(gflop / ms, 1),
            "images_per_call": per_call,
            "crop_only_step_batch_%d" % B: {"ms_per_step": round(ms32, 3), "crops_per_s": round(B * 1e3 / ms32, 1),
                                             "what": "the cfg3 step (synthetic full-image feature crop + Chamfer) at the "
                                                     "reference's 32 boxes per image"}}


def two_batches_object(net, inp, args, step):
    """Throughput with two batches in flight on two streams of the one GPU (tools/pipeline_steps.py as a bench object)."""
    steps = [step, Step(net.clone_with_own_scratch(), inp, args.points)]
    from monopsr_amd.core import device_net as dn
    streams = [torch.cuda.Stream()]
    streams.append(dn.concurrent_stream(streams[0].device, streams[0]))  # (one that does not share its hardware queue)
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
    k = max(4, min(20, args.steps)) // 2 * 2
    for i in range(2):  # warm-up: the second net's scratch and filter cache
        with torch.cuda.stream(streams[i]):
            steps[i]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        with torch.cuda.stream(streams[i % 2]):
            steps[i % 2]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in streams:
        main.wait_stream(s)
    return {"value": round(args.batch * k / dt, 2), "unit": "crops/s", "ms_per_batch": round(1e3 * dt / k, 3),
            "batches": k, "what": "consecutive batches alternate between two HIP streams (own scratch each); the headline "
                                  "`value` is the one-stream rate"}


def winograd_off_object(one_step, args, device, policy="off"):
    """The step under mpsr_set_winograd_policy(MPSR_WINOGRAD_OFF / _ACCURATE) + the element-wise error those policies
    are for."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    out = {"policy": "MPSR_WINOGRAD_OFF: no F(3x3,3x3) / F(4x4,3x3) kernel (border-class implicit GEMM and the exact tap "
                     "GEMMs instead)" if policy == "off" else
                     "MPSR_WINOGRAD_ACCURATE: only the transform-domain forms that keep an element-wise 1e-3 on heavy-tailed "
                     "maps -- sixteen-product tiles (block3), F(2x2,3x3) (decoder conv2_2 / conv3_2); no F(4x4,3x3), no "
                     "F(3x3,3x3) halo tiles (blocks 1-2 on the direct kernels); the exact tap GEMMs as always"}
    _lib.set_winograd_policy(policy)
    for _ in range(2):
        one_step()
    torch.cuda.synchronize()
    k = max(3, min(10, args.steps))
    t0 = time.perf_counter()
    for _ in range(k):
        one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update({"value": round(args.batch * k / dt, 2), "unit": "crops/s", "ms_per_step": round(1e3 * dt / k, 3)})
    _lib.set_winograd_policy("auto")
    # element-wise error on a heavy-tailed map, both policies, one layer each of the two Winograd kernels' shapes
    rng = np.random.default_rng(11)

    def hostile(shape):
        x = np.abs(rng.standard_normal(shape)).astype(np.float32)
        x *= rng.random(shape) >= 0.9
        x *= np.where(rng.random(shape) < 0.01, 1e3, 1.0).astype(np.float32)
        return x.astype(np.float32)
    errs = {}
    for name, (B, H, C, N, dil) in (("decoder conv3_2 (48x48, 128 -> 128)", (32, 48, 128, 128, 1)),
                                    ("block3 conv2 (12x12, 256 -> 256, dilation 4)", (64, 12, 256, 256, 4))):
        x = hostile((B, H, H, C))
        w = rng.standard_normal((N, 3, 3, C)) * np.sqrt(2.0 / (9 * C)) * \
            np.exp(rng.uniform(np.log(0.1), np.log(10.0), N))[:, None, None, None]
        w = w.astype(np.float32)
        bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
        ref = torch.relu(torch.nn.functional.conv2d(
            torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(0, 3, 1, 2),
            torch.from_numpy(bias).double(), padding=dil, dilation=dil)).permute(0, 2, 3, 1).numpy()
        scale = np.abs(ref).max()
        big = np.abs(ref) > 1e-3 * scale
        xd, wd, bd = torch.from_numpy(x).to(device), torch.from_numpy(w.reshape(N, -1)).to(device), \
            torch.from_numpy(bias).to(device)
        row = {}
        for pol in ("auto", policy):
            got = dn.conv2d(xd, wd, bd, None, 3, 3, dil, True, split_k=0, winograd_policy=pol).cpu().double().numpy()
            row[pol] = {"tensor_scale": float("%.3g" % (np.abs(got - ref).max() / scale)),
                        "element_wise": float("%.3g" % (np.abs(got - ref)[big] / np.abs(ref)[big]).max())}
        errs[name] = row
    out["error_vs_float64_on_heavy_tailed_map"] = dict(
        errs, what="max|err| / max|ref| and the largest relative error of an element with |ref| > 1e-3 max|ref|; map: "
                   "relu(N(0,1)), 90 % zeros, 1 % of the entries x1000; filters: He-scaled x per-channel gains over two "
                   "decades.  The default policy misses an ELEMENT-WISE 1e-3 on such maps by up to ~4x (its tensor-scale "
                   "error is ~1e-5); this mode keeps it")
    return out


def _timed_threads(fn, parts, workers):
    """Run fn(part) for every part on `workers` host threads (the C oracle is called through ctypes, which releases
    the GIL: the threads are as independent as the processes SURVEY 8(d) describes); -> wall seconds."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    if workers <= 1:
        for p in parts:
            fn(p)
    else:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(fn, parts))
    return time.perf_counter() - t0


def cpu_baseline(weights, host, sample, npts, budget_s=18.0):
    """The CPU side of SURVEY 8(d), all `kind: port` (the oracle restatements; /root/reference does not exist on the
    GPU box and TensorFlow 1.8 runs nowhere): `conv` = oracle/net.py on torch CPU with all host threads; `chamfer_1core`
    / `chamfer_nproc` = the C restatement of the reference's single-threaded CPU kernel on one core and as one
    independent worker per core; `emd_cfg5_subset` = the C restatement of approxmatch_cpu + matchcost + gradient on a
    few 2048-point clouds.  The top-level value blends conv (all threads) + Chamfer (1 thread) over the same instances
    as before.  Bounded: about `budget_s` seconds in all.  Reported baseline only."""
    from oracle import net as onet
    from oracle import ops as orc
    ncores = os.cpu_count() or 1
    sc = "FirstStageFeatureExtractor_crop/resnet_v1_101"
    B = host["crops"].shape[0]
    if sample <= 0:  # size the conv sample for roughly half the budget from a warm probe of the trunk
        with torch.no_grad():
            onet.resnet101_block3(torch.from_numpy(host["crops"][:2]), weights, sc)  # warm-up (thread pool, oneDNN)
            t0 = time.perf_counter()
            onet.resnet101_block3(torch.from_numpy(host["crops"][:8]), weights, sc)
        per = (time.perf_counter() - t0) / 8 * 1.6  # trunk is ~64 % of the per-crop work
        sample = int(max(8, min(B, 0.5 * budget_s / per)))
    s = slice(0, sample)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = onet.instance_path(host["crops"][s], host["full_feat"][s], host["boxes"][s], P2, host["view"][s],
                                 np.ones((sample, 1), np.int32), np.tile(np.array([[3.88, 1.63, 1.53]], np.float32),
                                                                         (sample, 1)),
                                 np.full((sample,), 2.17799973487854, np.float32), weights)
    t_conv = time.perf_counter() - t0
    pred = ref["inst_xyz_map_local"].reshape(sample, -1, 3)[:, :npts].contiguous().numpy()
    t0 = time.perf_counter()
    d1, i1, d2, i2 = orc.nn_distance(pred, host["gt"][s])
    orc.nn_distance_grad(pred, host["gt"][s], np.ones_like(d1), i1, np.ones_like(d2), i2)
    t_ch = time.perf_counter() - t0
    dt = t_conv + t_ch
    out = {"value": round(sample / dt, 3), "unit": "crops/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "first %d instances of the same workload: oracle/net.py (PyTorch-CPU fp32 restatement of the "
                     "TF1 graph, %d threads) + oracle C Chamfer fwd/bwd (1 thread), %.1f s" %
                     (sample, torch.get_num_threads(), dt),
           "host_cores": ncores,
           "conv": {"value": round(sample / t_conv, 3), "unit": "crops/s", "cores": torch.get_num_threads(),
                    "kind": "port", "sample": "%d instances, trunk + squash + decoder + heads, %.1f s"
                                              % (sample, t_conv)}}

    # Chamfer fwd + bwd of the whole batch: the reference's CPU kernel is single-threaded by construction
    def chamfer(sl):
        a, b = host["gt"][sl], host["gt2"][sl]
        e1, j1, e2, j2 = orc.nn_distance(a, b)
        orc.nn_distance_grad(a, b, np.ones_like(e1), j1, np.ones_like(e2), j2)
    host = dict(host, gt2=np.random.default_rng(4).standard_normal(host["gt"].shape, dtype=np.float32))
    t1 = _timed_threads(chamfer, [slice(0, B)], 1)
    out["chamfer_1core"] = {"value": round(B / t1, 2), "unit": "clouds/s", "cores": 1, "kind": "port",
                            "sample": "%d cloud pairs of %d points, fwd + bwd, %.2f s" % (B, npts, t1)}
    w = min(ncores, B)
    parts = [slice(i * B // w, (i + 1) * B // w) for i in range(w)]
    # (one pass of the batch over all cores is a 0.05 s sample -- +-15 % between runs; every worker repeats its share
    # until the sample is about a second)
    reps = max(1, min(64, int(1.0 / max(t1 / w, 1e-3)) + 1))

    def chamfer_reps(sl):
        for _ in range(reps):
            chamfer(sl)
    tn = _timed_threads(chamfer_reps, parts, w)
    out["chamfer_nproc"] = {"value": round(B * reps / tn, 2), "unit": "clouds/s", "cores": w, "kind": "port",
                            "sample": "the same %d pairs %d times over, as %d independent single-threaded workers, %.2f s"
                                      % (B, reps, w, tn)}
    # EMD at BASELINE cfg5's cloud size (2048 points, U(-1,1)): CPU semantics (11 levels, double state)
    n5 = 2048
    r6, r7 = np.random.default_rng(6), np.random.default_rng(7)
    k1 = 2
    k = max(k1, min(ncores, 64))
    c1 = (r6.random((k, n5, 3), dtype=np.float32) * 2 - 1)
    c2 = (r7.random((k, n5, 3), dtype=np.float32) * 2 - 1)

    def emd(sl):
        mt = orc.approx_match(c1[sl], c2[sl], semantics="cpu")
        orc.match_cost(c1[sl], c2[sl], mt, semantics="cpu")
        orc.match_cost_grad(c1[sl], c2[sl], mt, semantics="cpu")
    te1 = _timed_threads(emd, [slice(0, k1)], 1)
    ten = _timed_threads(emd, [slice(i, i + 1) for i in range(k)], k)
    out["emd_cfg5_subset"] = {
        "value": round(k / ten, 3), "unit": "clouds/s", "cores": k, "kind": "port",
        "one_core_clouds_per_s": round(k1 / te1, 3),
        "sample": "%d-cloud subset of cfg5's 256 x 2048^2 per-GPU share (approx_match + match_cost + grad, the "
                  "reference CPU kernel's semantics), one cloud per worker on %d workers: %.2f s; 1 core on %d clouds: "
                  "%.2f s; the full 256-cloud share would take %.0f s at the all-core rate" %
                  (k, k, ten, k1, te1, 256 * ten / k)}
    return out, ref


def training_step_object(device, batch, inp, steps=3, warmup=2, dist=None, red_dev=None, make_trainer=None,
                         bucket_mib=64, full_trunk=False, allreduce="rccl", decoder_bn="batch", one_rank_rccl=False):
    """ms per training step of `batch` instances per rank.  full_trunk=False: crop trunk + decoder + heads trainable
    (72.8 M parameters, 291 MB gradient; the full-image branch enters as a synthetic feature crop).  full_trunk=True: BOTH
    ResNet-101 trunks trainable from one raw 375x1242 image per rank + `batch` boxes -- the reference's whole trainable
    set (net_builder.py:44-52, core/trainer.py:71-81), the 100 M-parameter / 401 MB gradient SURVEY 8(d) cfg4 names; the
    default of an N > 1 run.  With a process group (N > 1) every rank trains its own shard and the flat gradient goes
    through core/trainer.ReverseBucketReducer (64 MiB buckets launched from the end of the buffer as backward reports
    layers ready, averaged, THEN per-variable clip as the reference does: core/trainer.py:76-81); the step is timed
    with and without the reduce (MAX over ranks), the difference being the exposed, non-overlapped part.
    one_rank_rccl (N = 1): a ONE-rank "nccl" process group is created for the duration of this object and the step runs
    the same RCCL calls an N > 1 step issues (force_collectives; every collective an identity) -- the library, the
    sizes and the stream ordering are the real ones, the wire is not."""
    world = dist.get_world_size() if dist is not None else 1
    own_pg = None
    if one_rank_rccl and dist is None:
        try:
            import socket
            import torch.distributed as own_pg
            if own_pg.is_initialized():
                raise RuntimeError("a process group exists already")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            own_pg.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            one_rank = {"backend": own_pg.get_backend(), "rccl_version": ".".join(map(str, torch.cuda.nccl.version()))}
        except Exception as e:  # the measurement still runs, without collectives
            own_pg, one_rank = None, {"error": repr(e)[:300]}
    try:
        return _training_step_object(device, batch, inp, steps, warmup, dist, red_dev, make_trainer, bucket_mib,
                                     full_trunk, allreduce, decoder_bn, world,
                                     one_rank if one_rank_rccl and dist is None else None, own_pg)
    finally:
        if own_pg is not None:
            try:
                own_pg.destroy_process_group()
            except Exception:
                pass


def _training_step_object(device, batch, inp, steps, warmup, dist, red_dev, make_trainer, bucket_mib, full_trunk,
                          allreduce, decoder_bn, world, one_rank, own_pg):
    if make_trainer is None:
        from monopsr_amd.core import config_utils, train_net, trainer
        from monopsr_amd.core import weights as W
        cfg = config_utils.default_config()
        if full_trunk:  # both trunks trainable: the 100.3 M-parameter / 401 MB gradient of BASELINE cfg4
            net = train_net.TrainNet(W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE)), device=device,
                                     decoder_bn=decoder_bn, full_trunk=True)
        else:
            net = train_net.TrainNet(W.synthetic_weights(seed=0), device=device, decoder_bn=decoder_bn)
        tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0,
                                     bucket_bytes=bucket_mib << 20, allreduce=allreduce,
                                     force_collectives=own_pg is not None)
        sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                      cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"],
                      mean_lwh=inp["mean_lwh"], prop_cen_z_offset=inp["z_off"])
        if full_trunk:  # the full-image branch needs the image itself and the boxes normalised to it
            g = torch.Generator(device=device).manual_seed(8)
            sample["rgb_image"] = torch.randint(0, 256, (375, 1242, 3), device=device, generator=g).float()
            sample["boxes_2d_norm"] = inp["boxes"] / torch.tensor([375.0, 1242.0, 375.0, 1242.0], device=device)
            del sample["full_img_feature_crop"], sample["rgb_image_crops"]  # (as tools/train_bench.py --full-image)
        sample.update(trainer.synthetic_ground_truth(sample, seed=7))
    else:
        tr, sample = make_trainer()
        net = tr.net
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)

    def timed(k):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        vals = [tr.step(sample) for _ in range(k)]  # device scalars: no host sync inside the timed region
        sync()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / k, [float(v) for v in vals]

    out = {"params": int(net.params.numel()), "grad_bytes": int(net.grads.numel() * 4), "steps": steps,
           "ranks": world,
           "trainable": ("both ResNet-101 trunks + decoder + heads = the reference's whole trainable set (SURVEY 8(d) "
                         "cfg4: 100,204,832 parameters / 400,819,328 gradient bytes in TF's parameterisation; here "
                         "BatchNorm is folded into one bias per channel and every tensor padded to 256 bytes); input: "
                         "one synthetic 375x1242 image + %d boxes per rank" % batch) if full_trunk else
                        "crop trunk + decoder + heads (the full-image branch enters as a synthetic feature crop)",
           "what": "fwd + configured losses (incl. global-map projection) + bwd + clip + Adam + EMA, fp32; decoder "
                   "BN: %s" % ("batch statistics pooled over all ranks (fp64 sums all-reduced per layer and direction: the "
                               "whole step's batch, as the reference's single-process step)" if decoder_bn == "batch_global"
                               else "per-rank batch statistics (no cross-rank statistics exchange; --decoder-bn "
                                    "batch_global pools them)"),
           "streams": "weight gradients on a second HIP stream of the same GPU (leaves of the backward pass), joined "
                      "at the end of backward and before every gradient bucket leaves"}
    if getattr(tr, "hardware_queues", None) is not None:
        out["hardware_queues"] = dict(tr.hardware_queues, what="device_net.blocked_by_collectives: does a waiting collective "
                                      "hold up launches on that stream (= it shares the communicator stream's hardware "
                                      "queue); the weight-gradient stream is the first candidate that is not held up")
    losses = [float(tr.step(sample)) for _ in range(warmup)]
    if world > 1:
        tr.reducer.enabled = False
        dt_local, _ = timed(steps)
        tr.reducer.enabled = True
        timed(1)  # first reduced step outside the timed region (communicator warm-up)
        dt, vals = timed(steps)
        losses += vals
        out.update({"ms_per_step": round(dt * 1e3, 2), "crops_per_s": round(batch * world / dt, 1),
                    "ms_per_step_without_allreduce": round(dt_local * 1e3, 2),
                    "exposed_allreduce_ms": round((dt - dt_local) * 1e3, 2),
                    "allreduce": "ReverseBucketReducer (%s): %d buckets of <= %d MiB, async, launched as backward reports "
                                 "layers ready; average, then per-variable clip_by_norm, then Adam" %
                                 ("one all_reduce per bucket" if tr.reducer.mode == "rccl" else
                                  "reduce_scatter_tensor + all_gather_into_tensor per bucket",
                                  len(tr.reducer.buckets), tr.reducer.bucket_bytes >> 20),
                    "allreduce_mode": tr.reducer.mode,
                    "nccl_env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS",
                                                                "RCCL_MSCCL_ENABLE") if os.environ.get(k) is not None},
                    "gradients": "real (this step's backward), %d floats" % net.grads.numel()})
        # the same buffer all-reduced alone: what the wire costs when nothing overlaps it
        try:
            dist.all_reduce(net.grads)
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                dist.all_reduce(net.grads)
            sync()
            ta = (time.perf_counter() - t0) / 3
            t = torch.tensor([ta], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ta = float(t.item())
            nb = net.grads.numel() * 4
            out["allreduce_alone_ms"] = round(ta * 1e3, 3)
            out["allreduce_alone_busbw_GBps"] = round(nb / ta * 2 * (world - 1) / world / 1e9, 1)
        except Exception as e:
            out["allreduce_alone_ms"] = repr(e)
    else:
        dt, vals = timed(steps)
        losses += vals
        out.update({"ms_per_step": round(dt * 1e3, 2), "crops_per_s": round(batch / dt, 1)})
        if one_rank is not None:
            if own_pg is not None:  # the step above ran the N > 1 step's RCCL calls on a one-rank communicator
                issued = list(getattr(tr.reducer, "last_issued", []))
                one_rank.update({"collectives_per_step": {k: issued.count(k) for k in sorted(set(issued))},
                                 "mode": tr.reducer.mode, "buckets": len(tr.reducer.buckets),
                                 "bucket_mib": tr.reducer.bucket_bytes >> 20})
                tr.reducer.enabled = False
                dt0, _ = timed(max(2, steps // 3))
                tr.reducer.enabled = True
                one_rank["ms_per_step_without_collectives"] = round(dt0 * 1e3, 2)
                one_rank["note"] = ("world_size 1: every collective is an identity -- what is exercised is RCCL itself, "
                                    "the bucket sizes, async issue from backward and the average -> clip -> Adam order; "
                                    "it says nothing about xGMI")
            out["rccl_one_rank"] = one_rank
    out["loss_per_step"] = [round(v, 1) for v in losses]
    out["note"] = ("random-initialised weights and synthetic targets: the first Adam steps are a transient (every "
                   "parameter moves by the learning rate whatever its gradient); the list is the loss of every step "
                   "run, warm-up included")
    return out


def rank_proof(dist, rank, world, backend, red_dev):
    """Evidence that N ranks on N distinct devices took part: an all-reduce of ones, every rank's device, and the
    collective library's version."""
    ones = torch.ones((1,), dtype=torch.float32, device=red_dev)
    dist.all_reduce(ones)
    cuda = torch.cuda.is_available()
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(),
          "device": torch.cuda.current_device() if cuda else None}
    if cuda:
        pr = torch.cuda.get_device_properties(me["device"])
        me["name"] = pr.name
        for k in ("uuid", "pci_bus_id", "pci_device_id"):
            if hasattr(pr, k):
                me[k] = str(getattr(pr, k))
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    ids = {(r.get("uuid") or r.get("pci_bus_id"), r["device"]) for r in ranks}
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            ver = repr(e)
    return {"allreduce_of_ones": float(ones.item()), "world_size": world,
            "backend": "nccl (RCCL)" if backend == "nccl" else backend, "rccl_version": ver,
            "distinct_devices": len(ids), "ranks": ranks}


def gather_rank_objects(dist, world, fn):
    """Every rank runs fn() (errors become {"error": ...} so that no rank skips the collective) -> list by rank."""
    try:
        mine = fn()
    except Exception as e:
        mine = {"error": repr(e)}
    objs = [None] * world
    dist.all_gather_object(objs, mine)
    return objs


_OUT = None  # the stream the ONE JSON line goes to (main(): the process's original stdout)


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too: RCCL prints a five-line version block
    ("RCCL version : ...", HIP / ROCm version, hostname, library path) to the C stdout when a communicator is created
    -- buffered, so it lands BEHIND the JSON line at exit (seen in r06 on the first run that created a communicator
    in this process).  So the process keeps its original stdout for the line alone and points file descriptor 1 at
    stderr for everybody else (C libraries, child processes, stray prints)."""
    global _OUT
    if _OUT is None:
        sys.stdout.flush()
        _OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _OUT


def emit_line(line):
    out = _OUT if _OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


class Emitter:
    """Rank 0 prints exactly ONE JSON line.  Once the headline exists a watchdog is armed: if the side measurements
    that follow (extra objects, collectives with the other ranks) have not finished by the deadline, the watchdog
    prints the line as it stands and ends the process -- a hang in an extra never costs the headline."""

    def __init__(self, rank, deadline_s):
        import threading
        self.rank, self.lock, self.done, self.result = rank, threading.RLock(), False, None  # (re-entrant: the signal handler runs on the main thread, possibly inside emit())
        self.timer = threading.Timer(deadline_s, self._expired)
        self.timer.daemon = True
        self.deadline_s = deadline_s

    def arm(self, result):
        self.result = result
        self.timer.start()
        if self.rank == 0:
            # the launcher stops every rank with SIGTERM when another rank dies (GPU fault, OOM kill) during the extras:
            # the measured headline must still come out
            import signal

            def on_term(signum, frame):
                sys.stderr.write("bench.py: rank 0 received signal %d after the headline was measured; emitting it "
                                 "without the unfinished extras\n" % signum)
                self._flush(reason="terminated_by_signal_%d" % signum)
                # non-zero: the run WAS terminated (another rank died); the line carries extras_incomplete and the
                # measured headline, the status tells the launcher's caller that the extras are missing for a reason
                os._exit(128 + signum)
            try:
                signal.signal(signal.SIGTERM, on_term)
            except ValueError:  # not the main thread
                pass

    def _flush(self, reason):
        with self.lock:
            if self.done or self.result is None:
                return
            keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config")
            try:
                line = json.dumps(dict(self.result, extras_incomplete=reason))
            except (RuntimeError, TypeError, ValueError):
                line = json.dumps(dict({k: self.result[k] for k in keys if k in self.result}, extras_incomplete=reason))
            emit_line(line)
            self.done = True

    def _expired(self):
        sys.stderr.write("bench.py: rank %d: the extra measurements did not finish within %.0f s; %s\n" %
                         (self.rank, self.deadline_s, "printing the headline without them" if self.rank == 0 else "exiting"))
        with self.lock:
            if not self.done and self.rank == 0 and self.result is not None:
                line = None
                for _ in range(20):
                    try:
                        line = json.dumps(dict(self.result, extras_timed_out_after_s=self.deadline_s))
                        break
                    except (RuntimeError, TypeError, ValueError):  # the main thread was adding a key / a half-built object
                        time.sleep(0.01)
                if line is None:  # fall back to the contract's own keys, which were complete before the watchdog was armed
                    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                            "scaling", "vs_baseline", "dtype", "data", "config")
                    line = json.dumps(dict({k: self.result[k] for k in keys if k in self.result},
                                           extras_timed_out_after_s=self.deadline_s))
                emit_line(line)
                self.done = True
        os._exit(0)  # every rank: the launcher must not turn a printed headline into a failed run

    def emit(self, result):
        with self.lock:
            if not self.done and self.rank == 0:
                emit_line(json.dumps(result))
            self.done = True
        self.timer.cancel()


def rendezvous_only(rank, world):
    """Test hook (tests/test_bench_launch.py): the whole N-rank plumbing of this file on CPU over gloo -- launch,
    rendezvous, max-over-ranks timing, rank proof, per-rank gathers, the reducer-driven training-step object and the
    all-rank EMD object -- with stand-in workloads where the hot path would run (it has no CPU form)."""
    import torch.distributed as dist
    from monopsr_amd.core.trainer import ReverseBucketReducer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    em = Emitter(rank, float(os.environ.get("MPSR_BENCH_TEST_DEADLINE", "300")))
    seen = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(seen, op=dist.ReduceOp.MAX)
    result = {"rendezvous_only": True, "n_gpus": world, "max_rank_plus_1": float(seen.item())}
    em.arm(result)
    if os.environ.get("MPSR_BENCH_TEST_HANG_EXTRAS"):  # test hook: an extra that never returns
        time.sleep(3600)
    result["rank_proof"] = rank_proof(dist, rank, world, "gloo", cpu)
    result["per_rank_ms"] = [o["ms"] for o in gather_rank_objects(dist, world, lambda: {"ms": 1.0 + rank})]

    class FakeNet:
        params = torch.zeros(4096)
        grads = torch.zeros(4096)

    class FakeTrainer:  # four "layers" whose backward fills its slice of the flat gradient, last layer first
        def __init__(self):
            self.net = FakeNet()
            self.spans = [(0, 1024), (1024, 2048), (2048, 3072), (3072, 4096)]
            self.reducer = ReverseBucketReducer(self.net.grads, self.spans, bucket_bytes=4096)

        def step(self, sample):
            for li in (3, 2, 1, 0):
                lo, hi = self.spans[li]
                self.net.grads[lo:hi] = float(rank + 1)
                self.reducer.layer_ready(li)
            self.reducer.finish(average=True)
            return self.net.grads.mean()
    result["training_step"] = training_step_object(cpu, 4, None, steps=2, warmup=1, dist=dist, red_dev=cpu,
                                                   make_trainer=lambda: (FakeTrainer(), None))
    result["emd"] = emd_all_ranks(dist, world, lambda: {"fused_loss_ms": 1.0, "fused_clouds_per_s": 10.0 * (rank + 1)})
    em.emit(result)
    dist.barrier()
    dist.destroy_process_group()


def emd_all_ranks(dist, world, fn):
    """BASELINE cfg5 across ranks: every rank runs its 256-cloud share concurrently (no exchange: clouds are
    independent); rank 0's object + the sum over ranks."""
    dist.barrier()
    objs = gather_rank_objects(dist, world, fn)
    out = dict(objs[0])
    ok = [o for o in objs if "fused_clouds_per_s" in o]
    out["all_ranks"] = {"ranks_ok": len(ok), "fused_clouds_per_s_sum": round(sum(o["fused_clouds_per_s"] for o in ok), 1),
                        "per_rank_fused_loss_ms": [o.get("fused_loss_ms") for o in objs],
                        "errors": [o["error"] for o in objs if "error" in o],
                        "note": "each rank timed its own share while the others ran theirs; no collective on the "
                                "data path"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="instances per GPU")
    ap.add_argument("--points", type=int, default=1024, help="points per cloud for Chamfer")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="instances timed on the host CPU (0 = skip, -1 = size for ~9 s of conv work)")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="also all-reduce a 100,204,832-float buffer per step (size of the model's gradient)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-emd", action="store_true", help="skip the extra EMD (BASELINE config 5) object")
    ap.add_argument("--no-train-step", action="store_true", help="skip the extra training_step object")
    ap.add_argument("--train-batch", type=int, default=0, help="instances per rank of the training_step object "
                                                               "(0 = --batch)")
    ap.add_argument("--streams", type=int, default=1,
                    help="split each GPU's batch into this many instance shards on separate HIP streams")
    ap.add_argument("--pws-per-cu", type=int, default=0,
                    help="tuning knob with --streams: persistent pointwise workgroups per CU of every launch (default 2)")
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16x3"],
                    help="contraction arithmetic of the timed run: fp32 (default, the headline) or the opt-in "
                         "split-bfloat16 mode (include/monopsr_hip.h MPSR_MATH_BF16X3)")
    ap.add_argument("--no-allreduce-probe", action="store_true",
                    help="N > 1: skip the extra region that repeats the step with the gradient-sized all-reduce")
    ap.add_argument("--no-fast-mode", action="store_true",
                    help="skip the extra bf16x3_mode measurement appended to a default fp32 run")
    ap.add_argument("--bucket-mib", type=int, default=64, choices=[16, 32, 64, 128],
                    help="N > 1: bucket size of the training step's gradient all-reduce (ReverseBucketReducer)")
    ap.add_argument("--train-full-trunk", action="store_true",
                    help="N = 1: make `training_step` itself the both-trunk step (it is the default for N > 1, and the "
                         "N = 1 line carries it as `training_step_full` anyway)")
    ap.add_argument("--train-crop-trunk-only", action="store_true",
                    help="N > 1: training_step on the crop-trunk net (72.8 M parameters, 291 MB gradient) instead of the "
                         "both-trunk net (the reference's whole trainable set, 401 MB: SURVEY 8(d) cfg4)")
    ap.add_argument("--no-train-step-full", action="store_true", help="N = 1: skip the training_step_full object")
    ap.add_argument("--no-full-image", action="store_true", help="skip the extra full_image_path object")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "direct"],
                    help="N > 1: the training step's gradient exchange per bucket -- rccl: one all_reduce (the library "
                         "picks ring / tree / ...), direct: reduce_scatter + all_gather of 1 / N shards (every peer's "
                         "own xGMI link once per phase, SURVEY 5)")
    ap.add_argument("--decoder-bn", default="batch", choices=["batch", "batch_global"],
                    help="training_step: map-decoder BatchNorm statistics per rank, or pooled over all ranks (the "
                         "reference's whole-step batch, net_builder.py:78-79,86-87)")
    ap.add_argument("--extras-deadline", type=float, default=900.0,
                    help="seconds after the headline is measured before the watchdog prints it without the "
                         "unfinished extra objects")
    args = ap.parse_args()

    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MPSR_BENCH_TEST_FAIL_RANK") == str(rank):  # test hooks of tests/test_bench_launch.py
        raise SystemExit(7)
    if os.environ.get("MPSR_BENCH_TEST_HANG"):
        time.sleep(600)
    if os.environ.get("MPSR_BENCH_RENDEZVOUS_ONLY"):
        return rendezvous_only(rank, world)
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
    multi = dist is not None and world > 1
    emitter = Emitter(rank, args.extras_deadline)

    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    _lib.set_conv_math(args.math)
    weights = W.synthetic_weights(seed=0)
    net = dn.DeviceNet(weights, device=device)
    inp, host = make_inputs(args.batch, args.points, rank, device)
    if args.pws_per_cu > 0:
        _lib.lib().mpsr_debug_set_pointwise_per_cu(args.pws_per_cu)
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
    t_own = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
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
    emitter.arm(result)  # from here on a hang in a side measurement costs the extras, not the line

    if multi:
        # ---- N > 1 only: who took part, and how evenly
        try:
            result["rank_proof"] = rank_proof(dist, rank, world, backend, red_dev)
        except Exception as e:
            result["rank_proof"] = {"error": repr(e)}
        try:
            ms = gather_rank_objects(dist, world, lambda: {"ms": round(1e3 * t_own / args.steps, 3)})
            result["per_rank_ms_per_step"] = [o.get("ms") for o in ms]
            result["ms_per_step_max_over_ranks"] = result["ms_per_step"]
        except Exception as e:
            result["per_rank_ms_per_step"] = {"error": repr(e)}

    if multi and backend == "nccl" and not args.allreduce_grads and not args.no_allreduce_probe:
        # BASELINE config 4, forward form: the sharded metric step with an asynchronous all-reduce of a buffer the
        # size of the FULL model's fp32 gradient (both trunks: 100,204,832 floats, synthetic contents -- the forward
        # path produces no parameter gradient) overlapping it.  The REAL gradient exchange is the training_step
        # object below.
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
                "note": "same forward step + async RCCL all-reduce(sum) of a full-model-gradient-sized SYNTHETIC fp32 "
                        "buffer per step; the real-gradient exchange is in training_step"}
            del gbuf
        except Exception as e:
            result["with_grad_allreduce"] = {"error": repr(e)}

    if rank == 0 and not args.no_roofline:
        try:
            result["roofline"] = roofline_object(net, args, device, result["ms_per_step"])
        except Exception as e:
            result["roofline"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline and args.math == "fp32" and isinstance(result.get("roofline"), dict) \
            and "flops_per_launch" in result["roofline"]:
        try:
            rf = result["roofline"]
            rf["in_situ"] = in_situ_object(one_step, rf["flops_per_launch"] * rf["launches_per_step"],
                                           result["ms_per_step"], expect_matrix_launches=rf["launches_per_step"])
        except Exception as e:
            result["roofline"]["in_situ"] = {"error": repr(e)[:300]}
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
            reps = 50
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
    if not args.no_roofline and not args.no_emd:
        if multi:  # cfg5 across ranks: every rank its own share, concurrently
            try:
                result["emd"] = emd_all_ranks(dist, world, lambda: emd_object(device))
            except Exception as e:
                result["emd"] = {"error": repr(e)}
        elif rank == 0:
            try:
                result["emd"] = emd_object(device)
            except Exception as e:
                result["emd"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline and not args.no_emd:
        try:
            result["ops_at_reference_shape"] = ops_at_reference_shape_object(device)
        except Exception as e:
            result["ops_at_reference_shape"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline and not args.no_emd and isinstance(step, Step):
        try:
            result["cfg5_step"] = cfg5_step_object(step, inp, device, args.steps)
        except Exception as e:
            result["cfg5_step"] = {"error": repr(e)}
    if rank == 0 and n_gpus == 1 and not args.no_roofline and not args.no_full_image and args.math == "fp32":
        try:
            result["full_image_path"] = full_image_path_object(device)
        except Exception as e:
            result["full_image_path"] = {"error": repr(e)}
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
    if rank == 0 and n_gpus == 1 and args.math == "fp32" and not args.no_fast_mode:
        # The element-wise-safe configuration (MPSR_WINOGRAD_OFF: direct / implicit-GEMM kernels instead of the F(3x3,3x3)
        # and F(4x4,3x3) ones; the tap GEMMs of the upsampled layers are exact and stay) with its throughput, next to what
        # it buys: the element-wise error of the decoder's conv3_2-shaped layer on a heavy-tailed map (90 % zeros, 1 % of
        # the entries x1000, trained-like filter scales: tests/test_hostile_inputs_gpu.py) under both policies, against a
        # float64 torch convolution on the host.  NOT the headline (`value` above is the default policy).
        try:
            result["winograd_off_mode"] = winograd_off_object(one_step, args, device)
        except Exception as e:
            result["winograd_off_mode"] = {"error": repr(e)}
        finally:
            _lib.set_winograd_policy("auto")
        try:  # r06: the policy between the two -- element-wise-safe transform-domain forms only
            result["winograd_accurate_mode"] = winograd_off_object(one_step, args, device, policy="accurate")
        except Exception as e:
            result["winograd_accurate_mode"] = {"error": repr(e)}
        finally:
            _lib.set_winograd_policy("auto")
    if rank == 0 and n_gpus == 1 and args.math == "fp32" and not args.no_fast_mode and isinstance(step, Step):
        # A serving configuration, NOT the headline: consecutive batches (independent by construction) issued alternately on
        # two HIP streams with their own scratch, so that one batch's launch edges (a 180 us launch pays ~20-30 us of
        # ramp-up and drain) overlap the other's matrix work.  `value` above keeps one stream: a step there is
        # one batch start to end.
        try:
            result["two_batches_in_flight"] = two_batches_object(net, inp, args, step)
        except Exception as e:
            result["two_batches_in_flight"] = {"error": repr(e)}
    if args.math == "fp32" and not args.no_train_step and not args.no_roofline:
        # SURVEY 8(f3) / BASELINE cfg4: one data-parallel TRAINING step of the same instances on EVERY rank (forward,
        # the reference's configured loss set, backward, all-reduce of the real gradient, per-variable clip, Adam +
        # moving average; map-decoder BatchNorm on per-rank batch statistics).  A side measurement: never part of
        # `value`.  All ranks enter (it holds collectives); failures are symmetric (same code, same sizes).
        if multi or rank == 0:
            try:
                tb = args.train_batch or args.batch
                tinp = inp if tb == args.batch else make_inputs(tb, args.points, rank, device)[0]
                # N = 1: 2 + 18 steps, so that the line shows the loss past the first Adam steps' transient
                # N > 1: the both-trunk net (cfg4's 401 MB gradient) unless --train-crop-trunk-only
                full = (not args.train_crop_trunk_only) if multi else args.train_full_trunk
                ts = training_step_object(device, tb, tinp, steps=6 if multi else 18, dist=dist if multi else None,
                                          red_dev=red_dev, bucket_mib=args.bucket_mib, full_trunk=full,
                                          allreduce=args.allreduce, decoder_bn=args.decoder_bn)
                result["training_step"] = ts
            except Exception as e:
                result["training_step"] = {"error": repr(e)}
        if not multi and rank == 0 and not args.no_train_step_full and not args.train_full_trunk:
            # BASELINE cfg4's per-GPU share as specified: the both-trunk, 100 M-parameter step (401 MB gradient), on a
            # one-rank RCCL communicator so that the N > 1 step's collective calls run for real (identities)
            try:
                tb = args.train_batch or args.batch
                tinp = inp if tb == args.batch else make_inputs(tb, args.points, rank, device)[0]
                result["training_step_full"] = training_step_object(
                    device, tb, tinp, steps=6, dist=None, red_dev=red_dev, bucket_mib=args.bucket_mib, full_trunk=True,
                    allreduce=args.allreduce, decoder_bn=args.decoder_bn, one_rank_rccl=True)
            except Exception as e:
                result["training_step_full"] = {"error": repr(e)}
    if rank == 0 and args.cpu_sample != 0 and n_gpus == 1:
        try:
            result["cpu_baseline"], _ = cpu_baseline(weights, host, min(args.cpu_sample, args.batch), args.points)
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}

    emitter.emit(result)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
