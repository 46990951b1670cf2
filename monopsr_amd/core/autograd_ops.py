"""Differentiable wrappers (torch.autograd.Function) over the HIP forward / backward kernels.

Weights are NOT autograd leaves: a layer's weight and bias live in a flat parameter buffer and their gradients are
accumulated by the wgrad / bias-grad kernels straight into the matching slice of a flat gradient buffer (the buffer
the data-parallel all-reduce and the Adam kernel run over).  Autograd only routes activation gradients.
The caller zeroes the flat gradient buffer once per step.
"""
import torch
import torch.nn.functional as F

from monopsr_amd import _lib
from monopsr_amd.core import device_net as dn


# split_k argument of every forward / data-gradient convolution: 0 = the library schedules the launch (Winograd for
# the decoder's dense 3x3 layers and their data gradients, stream-K for the long-K FC layers): 88 -> 81 ms per step
_WGRAD_SCRATCH = {}
dn._SCRATCH_CACHES.append(_WGRAD_SCRATCH)
_SCHED = 0


class LayerRef:
    """One conv / FC layer: views into the flat parameter and gradient buffers (BatchNorm already folded)."""

    def __init__(self, w, b, dw, db, cin, cout, kh, kw, dilation, relu):
        self.w, self.b, self.dw, self.db = w, b, dw, db
        self.cin, self.cout, self.kh, self.kw, self.dilation, self.relu = cin, cout, kh, kw, dilation, relu
        # batch_norm: None, or a BatchNormState -- the convolution then runs without bias / activation and `b` / `db`
        # are the BatchNorm beta and its gradient (still this layer's second variable for clipping and all-reduce)
        self.batch_norm = None


class BatchNormState:
    """Moving statistics of one training-mode BatchNorm (slim.batch_norm defaults: decay 0.999, no scale).
    `sync_group`: None = statistics over this process's instances; a torch.distributed group (or True for the default
    group) = over the instances of ALL its ranks -- the reference normalises over the whole step's batch
    (net_builder.py:78-79,86-87), so a sharded batch must pool its sums to give the same result."""

    def __init__(self, moving_mean, moving_variance, eps=1e-3, decay=0.999, sync_group=None):
        self.moving_mean, self.moving_variance, self.eps, self.decay = moving_mean, moving_variance, eps, decay
        self.sync_group = sync_group

    def _group(self):
        """The process group to pool sums over, or None when there is nothing to pool."""
        import torch.distributed as dist
        if self.sync_group is None or not (dist.is_available() and dist.is_initialized()):
            return None
        g = None if self.sync_group is True else self.sync_group
        return (g, dist) if dist.get_world_size(g) > 1 else None

    def all_reduce_sums(self, t):
        """Sum a small fp64 tensor over the group in place (device tensor; through the host where the backend has no
        device collectives: gloo in the CPU / shared-GPU tests)."""
        gd = self._group()
        if gd is None:
            return t
        g, dist = gd
        if dist.get_backend(g) == "gloo" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=g)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=g)
        return t


def _relu_grad(dy, y):
    dx = torch.empty_like(dy)
    _lib.check(_lib.lib().mpsr_relu_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dx), dy.numel(), _lib.stream()))
    return dx


class Conv2dFn(torch.autograd.Function):
    """y = act(conv(x, L.w) + L.b + residual); backward deposits dW, db into L.dw, L.db."""

    @staticmethod
    def forward(ctx, x, residual, layer, token):
        # `token` is a dummy scalar that requires grad: it puts the layer into the autograd graph even when neither
        # x nor residual needs a gradient (first layer), so that backward still deposits dW / db
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        bn = layer.batch_norm is not None
        y = dn.conv2d(x, layer.w, None if bn else layer.b, res, layer.kh, layer.kw, layer.dilation,
                      layer.relu and not bn, split_k=_SCHED)
        ctx.layer = layer
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if (layer.relu and not bn) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = ctx.layer
        x, y = ctx.saved_tensors
        g = _masked_grad(L, dy, y)
        _deposit_weight_grad(L, x, g)
        dx = _data_grad(L, g, x.shape[3]) if ctx.needs_input_grad[0] else None
        return dx, (g if ctx.has_res else None), None, None


def _masked_grad(L, dy, y):
    """dy through the layer's own ReLU (one streaming pass; the bias gradient rides in the weight-gradient kernel)."""
    g = dy.contiguous()
    if L.relu and L.batch_norm is None:
        gm = torch.empty_like(g)
        _lib.check(_lib.lib().mpsr_act_bias_grad(_lib.ptr(g), _lib.ptr(y), _lib.ptr(gm), None, g.numel() // L.cout,
                                                 L.cout, _lib.stream()))
        g = gm
    return g


def _pad4(L, g):
    """The kernels read rows of N floats with 16-byte loads: a narrow output (the 3-channel xyz head) is padded to 4."""
    pad = (-L.cout) % 4
    return (F.pad(g, (0, pad)) if pad else g), pad


# Weight gradients are LEAVES of the backward pass: nothing later in the pass reads dW, while the data gradients form the
# chain every other launch waits for.  r06: they are launched on a second HIP stream, so that the ramp-up and the drain of
# each launch -- ~30 us of a 1x1 weight gradient's 180 are fixed cost, not loop time -- overlap with the
# other stream's work instead of idling the chip.  The side stream waits for the event that marks x and g ready; x and g
# are pinned to it (record_stream); the main stream joins it at the end of the backward pass (an autograd engine
# callback queued by the first deposit of the pass) and before the data-parallel reducer issues a bucket.
# A/B switch (tools/train_bench.py --wgrad-main-stream).
WGRAD_SIDE_STREAM = True
WGRAD_SIDE_STREAMS = 1  # how many side streams the launches rotate over
WGRAD_STREAM_PRIORITY = 0  # A/B: -1 = a high-priority side stream
_WGRAD_STREAMS = {}   # device index -> [torch.cuda.Stream, ...]
_WGRAD_NEXT = {}      # device index -> launches so far (the rotation)
_JOIN_QUEUED = set()  # device indices whose end-of-backward join is queued in the running pass
_STREAM_OBJECTS = {}  # (device index, raw stream handle) -> torch.cuda.Stream (a handle names one stream for the process's life)
_PASS_STREAMS = {}    # device index -> {stream id: stream} the running pass's deposits were launched FROM (a TrainNet with
#                       both trunks runs the full-image trunk, forward and therefore backward, on its own stream)


def _wgrad_stream(device, main=None):
    pool = _WGRAD_STREAMS.setdefault(device.index, [])
    k = _WGRAD_NEXT.get(device.index, 0)
    _WGRAD_NEXT[device.index] = k + 1
    k %= max(1, WGRAD_SIDE_STREAMS)
    while len(pool) <= k:
        # a stream that shares its hardware queue with `main` would serialise with it: picked by measurement, once
        pool.append(dn.concurrent_stream(device, main, priority=WGRAD_STREAM_PRIORITY))
    return pool[k]


def join_wgrad_stream(device=None):
    """Make the current stream wait for every weight gradient launched on the side stream so far (no-op when the side
    stream was never used).  Called by the backward pass's own end-of-pass callback, by the reducer before it issues a
    bucket, and by anyone who reads the flat gradient buffer from inside a backward pass."""
    idx = None if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    _JOIN_QUEUED.discard(idx)
    cur = torch.cuda.current_stream(idx)
    for s in list(_PASS_STREAMS.pop(idx, {}).values()) + list(_WGRAD_STREAMS.get(idx, ())):
        if s.cuda_stream != cur.cuda_stream:
            cur.wait_stream(s)


def settle_wgrad_stream(device, group=None, candidates=4):
    """For a rank whose gradient exchange runs on RCCL: make sure the weight-gradient stream does not share a hardware queue
    with the communicator's stream (device_net.blocked_by_collectives) -- a waiting or running collective would hold up
    every weight gradient behind it.  Probes the caller's current stream once and exactly `candidates` streams (one tiny
    all_reduce each: the SAME number on every rank, whatever they find), keeps the first candidate that overlaps the
    caller's stream and is not held up.  -> a report for the bench line / logs."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", idx)
    main = torch.cuda.current_stream(device)
    helper = dn.concurrent_stream(device, main)
    report = {"caller_stream_held_up_by_collectives": bool(dn.blocked_by_collectives(main, group, helper))}
    pool = _WGRAD_STREAMS.setdefault(idx, [])
    cands = list(pool[:1])
    while len(cands) < candidates:
        cands.append(dn.concurrent_stream(device, main, priority=WGRAD_STREAM_PRIORITY))
    held = [bool(dn.blocked_by_collectives(c, group)) for c in cands]
    pick = held.index(False) if False in held else 0
    report["wgrad_stream_candidates_held_up"] = held
    report["wgrad_stream_picked"] = pick
    if pool:
        pool[0] = cands[pick]
    else:
        pool.append(cands[pick])
    return report


def wgrad_stream_if_any(device):
    """The side stream the weight gradients of `device` run on, or None (switched off / never used / several of them)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    pool = _WGRAD_STREAMS.get(idx, ())
    return pool[0] if (WGRAD_SIDE_STREAM and len(pool) == 1) else None


def order_behind_pass_streams(stream, device):
    """`stream` waits for every stream the running pass has deposited gradients from so far."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    for s in _PASS_STREAMS.get(idx, {}).values():
        if s.cuda_stream != stream.cuda_stream:
            stream.wait_stream(s)


def _deposit_weight_grad(L, x, g):
    """dW (and db) of one layer from its input x and the gradient g at its pre-activation output, accumulated into the
    layer's slices of the flat gradient buffer; then the layer's `on_grad_ready` hook (the data-parallel reducer)."""
    if x.is_cuda:
        dev = x.device
        # (host cost matters here -- 109 deposits per step, and a step of 8-32 instances is bound by the host: the stream
        # OBJECT of a raw handle is looked up, not rebuilt (torch.cuda.current_stream() takes ~8 us), and the side stream
        # is made current with two set_stream calls instead of a context manager)
        raw = _lib.stream(dev.index)
        main = _STREAM_OBJECTS.get((dev.index, raw))
        if main is None:
            main = _STREAM_OBJECTS[(dev.index, raw)] = torch.cuda.current_stream(dev)
        _PASS_STREAMS.setdefault(dev.index, {})[raw] = main
        in_pass = True
        if dev.index not in _JOIN_QUEUED:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda: join_wgrad_stream(dev))
                _JOIN_QUEUED.add(dev.index)
            except RuntimeError:  # not inside a backward pass (a backward formula called by hand): join right away
                in_pass = False
    if WGRAD_SIDE_STREAM and x.is_cuda:
        side = _wgrad_stream(dev, main)
        side.wait_stream(main)  # x, g (and the step's zeroed gradient buffer) are ready on the main stream here
        torch.cuda.set_stream(side)
        try:
            _launch_weight_grad(L, x, g)
        finally:
            torch.cuda.set_stream(main)
        x.record_stream(side)
        g.record_stream(side)
        if not in_pass:
            join_wgrad_stream(dev)
    else:
        _launch_weight_grad(L, x, g)
    ready = getattr(L, "on_grad_ready", None)
    if ready is not None:
        ready()  # e.g. the data-parallel reducer: this layer's slice of the flat gradient is complete (or on its way on
        #          the side stream: the reducer joins it before a bucket leaves)


def _launch_weight_grad(L, x, g):
    lib = _lib.lib()
    s = _lib.stream()
    B, H, W, C = x.shape
    N = L.cout
    bn = L.batch_norm is not None
    g4, pad = _pad4(L, g)
    N4 = N + pad
    db4 = None if bn else L.db  # with BatchNorm, db is the beta gradient and BatchNormReluFn deposits it
    if pad:
        dw4 = torch.zeros((N4, L.w.shape[1]), dtype=torch.float32, device=x.device)
        if db4 is not None:
            db4 = torch.zeros((N4,), dtype=torch.float32, device=x.device)
    else:
        dw4 = L.dw
    # scratch for the layers that take the Winograd-domain weight gradient (the decoder's dense 3x3 layers), kept per
    # (device, stream) like the forward scheduler's: launches on one stream are ordered
    nws = lib.mpsr_conv2d_wgrad_scratch_floats(B, H, W, C, N4, L.kh, L.kw, L.dilation)
    ws = dn.stream_scratch(_WGRAD_SCRATCH, x.device, nws) if nws else None
    _lib.check(lib.mpsr_conv2d_wgrad_ws_f32(_lib.ptr(x), _lib.ptr(g4), B, H, W, C, N4, L.kh, L.kw, L.dilation,
                                            _lib.ptr(dw4), _lib.ptr(db4), _lib.ptr(ws),
                                            ws.numel() if ws is not None else 0, s))
    if pad and db4 is not None:
        L.db.add_(db4[:N])
    if pad:
        L.dw.add_(dw4[:N])


class DgradBank:
    """The data-gradient layouts (mpsr_conv2d_dgrad_pack) of ALL layers in one flat buffer, refreshed by ONE launch
    per step (mpsr_conv2d_dgrad_pack_batch) instead of one launch per layer inside backward (105 launches of ~5 us
    and as many gaps per step).  The weights only change in the optimizer step: the trainer calls refresh() before
    backward and invalidate() after it; while `fresh`, the _data_grad* helpers read a layer's `wd` view instead of
    packing.  Anything else that runs backward (tests, a caller's own loop) never sees a stale layout: without
    refresh() the per-layer path runs as before."""

    def __init__(self, layers, device, skip=()):
        """skip: indices of layers whose data gradient is never computed (a trunk's root convolution reads the image,
        which needs no gradient): they get no slice -- should one ever be asked for, _dgrad_filter packs it per call."""
        lib = _lib.lib()
        skip = set(skip)
        banked = [(i, L) for i, L in enumerate(layers) if i not in skip]
        sizes = []
        for _, L in banked:
            n4 = L.cout + (-L.cout) % 4
            sizes.append(L.cin * L.kh * L.kw * n4)
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 63) // 64 * 64)  # 256-byte aligned slices (16-byte loads in the kernels)
        self.flat = torch.empty((offs[-1],), dtype=torch.float32, device=device)
        self.skipped = sorted(skip)
        jobs = (_lib.PackJob * len(banked))()
        for j, (_, L) in enumerate(banked):
            n4 = L.cout + (-L.cout) % 4
            L.wd = self.flat[offs[j]:offs[j] + sizes[j]].view(L.cin, L.kh * L.kw * n4)
            L.dgrad_bank = self
            jobs[j].w, jobs[j].wd = L.w.data_ptr(), L.wd.data_ptr()
            jobs[j].N, jobs[j].Nd, jobs[j].T, jobs[j].C = L.cout, n4, L.kh * L.kw, L.cin
        nbytes = lib.mpsr_dgrad_pack_table_bytes(jobs, len(banked))
        if not nbytes:
            raise _lib.InvalidArgumentError("DgradBank: bad layer table")
        import ctypes
        host = torch.zeros(((nbytes + 7) // 8,), dtype=torch.int64)
        n_chunks = ctypes.c_longlong(0)
        _lib.check(lib.mpsr_dgrad_pack_table_build(jobs, len(banked), host.data_ptr(), ctypes.byref(n_chunks)))
        self.table = host.to(device)
        self.n_chunks = int(n_chunks.value)
        self.fresh = False

    def refresh(self):
        _lib.check(_lib.lib().mpsr_conv2d_dgrad_pack_batch(_lib.ptr(self.table), self.n_chunks, _lib.stream()))
        self.fresh = True

    def invalidate(self):
        self.fresh = False


def _dgrad_filter(L, C, N4, pad, device):
    """wd (C, taps x N4) of layer L: its slice of a fresh DgradBank, else packed here."""
    bank = getattr(L, "dgrad_bank", None)
    if bank is not None and bank.fresh and C == L.cin:
        return L.wd
    w4 = F.pad(L.w, (0, 0, 0, pad)) if pad else L.w
    wd = torch.empty((C, L.kh * L.kw * N4), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().mpsr_conv2d_dgrad_pack(_lib.ptr(w4.contiguous()), N4, L.kh, L.kw, C, _lib.ptr(wd),
                                                 _lib.stream()))
    return wd


def _data_grad(L, g, C, residual=None):
    """dX of one layer = the forward convolution of g with the taps flipped and the channel roles swapped
    (mpsr_conv2d_dgrad_pack); `residual` (same shape as dX) is added in that convolution's epilogue -- the gradient of
    another branch that read the same tensor, for free instead of by an elementwise launch."""
    g4, pad = _pad4(L, g)
    N4 = L.cout + pad
    wd = _dgrad_filter(L, C, N4, pad, g.device)
    B = g4.shape[0]
    if (L.kh == 1 and L.kw == 1 and g4.shape[1] == 1 and g4.shape[2] == 1 and residual is None and C >= 4096
            and B % 4 == 0 and B <= 1024):
        # a few-row FC layer with a WIDE input (img_fc: 256 x 1024 -> 256 x 18432): as written the GEMM has 256 rows
        # and runs on the few-row kernel at 54 TF/s (178 us); transposed -- rows = the 18432 input features, wd IS
        # that operand (C, N4), g the (B, N4) "filter" -- it is a plain 1x1 convolution with tiles for every CU
        # (100 us) and one small transpose of the result
        dxt = dn.conv2d(wd.view(1, C, 1, N4), g4.reshape(B, N4), None, None, 1, 1, 1, False, split_k=_SCHED)
        return dxt.view(C, B).t().contiguous().view(B, 1, 1, C)
    return dn.conv2d(g4, wd, None, residual, L.kh, L.kw, L.dilation, False, split_k=_SCHED)


class UnitLink:
    """Handshake between two consecutive bottleneck units of a chain (the tensor between them has NO other consumer).
    The later unit's backward runs first; when it can hand its input gradient over already multiplied by the ReLU mask
    of that tensor (masked store path of the pointwise kernel) it says so here and the earlier unit skips its own
    elementwise ReLU-gradient pass over the 1024-channel gradient (23 x ~90 us per step in block3)."""

    def __init__(self):
        self.premasked = False
        self.bits = None  # the ReLU bit mask of the tensor between the units, written by the earlier unit's conv3 launch


_MASK_SCRATCH = {}
dn._SCRATCH_CACHES.append(_MASK_SCRATCH)
# A/B switch (tools/train_bench.py --unfused-relu-grads): False = every ReLU gradient inside a unit is an elementwise pass
FUSED_RELU_GRADS = True


def _conv1x1_with_mask(x, L, residual):
    """y = L(x) + residual through ReLU AND the bit mask of y from the same launch (mpsr_conv1x1_relu_bitmask_f32), or
    None when that launch does not take the layer."""
    lib = _lib.lib()
    B, H, W, C = x.shape
    M, N = B * H * W, L.cout
    if L.kh != 1 or L.kw != 1 or not L.relu or not lib.mpsr_conv1x1_masked_applies(M, C, N):
        return None
    y = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device)
    bits = torch.empty((lib.mpsr_relu_bitmask_words(M, N),), dtype=torch.int32, device=x.device)
    _lib.check(lib.mpsr_conv1x1_relu_bitmask_f32(_lib.ptr(x), M, C, _lib.ptr(L.w), _lib.ptr(L.b), _lib.ptr(residual), 1,
                                                 _lib.ptr(y), _lib.ptr(bits), N, _lib.stream()))
    return y, bits


def _data_grad_masked(L, g, x, residual, bits=None):
    """dX of the 1x1 layer L times the ReLU mask of x (= L's post-ReLU input), or None when the masked pointwise launch
    does not take the shape: mpsr_relu_bitmask(x) (unless the producer of x left its `bits`) -> mpsr_conv1x1_masked_f32."""
    lib = _lib.lib()
    C, N = x.shape[3], L.cout
    M = x.numel() // C
    if L.kh != 1 or L.kw != 1 or N % 4 or not lib.mpsr_conv1x1_masked_applies(M, N, C):
        return None
    s = _lib.stream()
    if bits is None:
        bits = dn.stream_scratch(_MASK_SCRATCH, x.device, lib.mpsr_relu_bitmask_words(M, C))
        _lib.check(lib.mpsr_relu_bitmask(_lib.ptr(x), M, C, _lib.ptr(bits), s))
    wd = _dgrad_filter(L, C, N, 0, g.device)
    dx = torch.empty_like(x)
    _lib.check(lib.mpsr_conv1x1_masked_f32(_lib.ptr(g), M, N, _lib.ptr(wd), None, _lib.ptr(residual), _lib.ptr(bits),
                                           _lib.ptr(dx), C, s))
    return dx


def _data_grad_relu_masked(L, g, y_in):
    """dX of layer L times the ReLU mask of its (post-ReLU) input y_in in ONE library call
    (mpsr_conv2d_relu_masked_f32: the F(3x3,3x3) kernel applies the mask in its epilogue)."""
    lib = _lib.lib()
    g4, pad = _pad4(L, g)
    N4 = L.cout + pad
    C = y_in.shape[3]
    s = _lib.stream()
    wd = _dgrad_filter(L, C, N4, pad, g.device)
    B, H, W, _ = g4.shape
    ws = dn.stream_scratch(dn._SCHED_SCRATCH, g.device, lib.mpsr_conv2d_scratch_floats(B, H, W, C))
    dx = torch.empty_like(y_in)
    _lib.check(lib.mpsr_conv2d_relu_masked_f32(_lib.ptr(g4.contiguous()), B, H, W, N4, _lib.ptr(wd), _lib.ptr(y_in),
                                               _lib.ptr(dx), C, L.kh, L.kw, L.dilation, _lib.ptr(ws), ws.numel(), s))
    return dx


class BottleneckFn(torch.autograd.Function):
    """One ResNet-v1 bottleneck unit (object_detection/nets/resnet_v1.py:79-139) as ONE autograd node:
    y = relu(conv3(conv2(conv1(x))) + shortcut(x)), shortcut = x or a 1x1 projection.  Layer by layer, autograd has to
    SUM the two gradients that reach x (through conv1 and through the shortcut) with an elementwise launch per unit
    (30 per trunk and step, 1.9 ms); here the shortcut branch's gradient is the `residual` operand of conv1's
    data-gradient convolution and is added in its epilogue.  Same kernels otherwise, gradients deposited in the same
    order (conv3, conv2, conv1, projection).  `in_link` / `out_link` (UnitLink or None) chain consecutive units: with an
    in_link the unit may return its input gradient already masked by (x > 0) and flags that on the link; with an
    out_link it takes `dy` as already masked when the next unit flagged it."""

    @staticmethod
    def forward(ctx, x, c1, c2, c3, sc, token, in_link=None, out_link=None):
        x = x.contiguous()
        res = dn.conv2d(x, sc.w, sc.b, None, sc.kh, sc.kw, sc.dilation, sc.relu, split_k=_SCHED) if sc is not None else x
        t1 = dn.conv2d(x, c1.w, c1.b, None, c1.kh, c1.kw, c1.dilation, c1.relu, split_k=_SCHED)
        t2 = dn.conv2d(t1, c2.w, c2.b, None, c2.kh, c2.kw, c2.dilation, c2.relu, split_k=_SCHED)
        # (a linked unit's conv3 launch also leaves the bit mask of y for the next unit's backward)
        ym = _conv1x1_with_mask(t2, c3, res.contiguous()) if out_link is not None else None
        if ym is not None:
            y, out_link.bits = ym
        else:
            y = dn.conv2d(t2, c3.w, c3.b, res, c3.kh, c3.kw, c3.dilation, c3.relu, split_k=_SCHED)
        ctx.layers = (c1, c2, c3, sc)
        ctx.links = (in_link, out_link)
        ctx.save_for_backward(x, t1, t2, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, c2, c3, sc = ctx.layers
        in_link, out_link = ctx.links
        x, t1, t2, y = ctx.saved_tensors
        if out_link is not None and out_link.premasked:
            out_link.premasked = False
            g3 = dy.contiguous()  # the next unit's data gradient left through the mask of y already
        else:
            g3 = _masked_grad(c3, dy, y)  # also the gradient that enters the shortcut branch
        _deposit_weight_grad(c3, t2, g3)
        # conv3's data gradient through conv2's ReLU mask, conv2's through conv1's: in the launches' store paths
        g2 = _data_grad_masked(c3, g3, t2, None) if (c2.relu and FUSED_RELU_GRADS) else None
        if g2 is None:
            g2 = _masked_grad(c2, _data_grad(c3, g3, t2.shape[3]), t2)
        _deposit_weight_grad(c2, t1, g2)
        if c1.relu and FUSED_RELU_GRADS:
            g1 = _data_grad_relu_masked(c2, g2, t1)
        else:
            g1 = _masked_grad(c1, _data_grad(c2, g2, t1.shape[3]), t1)
        _deposit_weight_grad(c1, x, g1)
        dx = None
        residual = g3
        if sc is not None:
            gs = _masked_grad(sc, g3, None)  # (a projection shortcut has no activation)
            _deposit_weight_grad(sc, x, gs)
            if ctx.needs_input_grad[0]:
                residual = _data_grad(sc, gs, x.shape[3])
        if ctx.needs_input_grad[0]:
            if in_link is not None:
                dx = _data_grad_masked(c1, g1, x, residual, in_link.bits)
                in_link.premasked = dx is not None
                in_link.bits = None
            if dx is None:
                dx = _data_grad(c1, g1, x.shape[3], residual=residual)
        return dx, None, None, None, None, None, None, None


_UPCONV_SCRATCH = {}
dn._SCRATCH_CACHES.append(_UPCONV_SCRATCH)


class UpsampledConv2dFn(torch.autograd.Function):
    """tf.image.resize_bilinear(x, size, align_corners) -> 3x3 SAME conv with layer L as ONE operator that never forms
    the upsampled map (csrc/upconv.hip; the map decoder's conv2_1 / conv3_1, net_builder.py:72-85).  Forward: tap GEMM on
    the source map + gather.  Backward: the transposed gather of dy, then a 1x1 weight gradient and a 1x1 data-gradient
    GEMM, all at the SOURCE resolution -- in place of the resize gradient, the 3x3 data gradient and the (Winograd-domain)
    weight gradient on the 4x larger map.  With a BatchNormState on the layer the convolution runs bare (BatchNormReluFn
    follows), otherwise bias + ReLU are part of the operator."""

    @staticmethod
    def forward(ctx, x, layer, size, align_corners, token):
        x = x.contiguous()
        bn = layer.batch_norm is not None
        y = dn.conv3x3_upsampled(x, size, layer.w, None if bn else layer.b, layer.relu and not bn, align_corners)
        ctx.layer, ctx.cfg = layer, (tuple(size), bool(align_corners))
        ctx.save_for_backward(x, y if (layer.relu and not bn) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = ctx.layer
        x, y = ctx.saved_tensors
        (oh, ow), ac = ctx.cfg
        g = _masked_grad(L, dy, y)
        lib = _lib.lib()
        B, h, w, C = x.shape
        s = _lib.stream()
        if L.batch_norm is None:  # (with BatchNorm, db is the beta gradient and BatchNormReluFn deposits it)
            _lib.check(lib.mpsr_bias_grad(_lib.ptr(g), g.numel() // L.cout, L.cout, _lib.ptr(L.db), s))
        nws = lib.mpsr_conv3x3_upsampled_bwd_scratch_floats(B, h, w, C, L.cout)
        ws = dn.stream_scratch(_UPCONV_SCRATCH, x.device, nws)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(lib.mpsr_conv3x3_upsampled_bwd_f32(_lib.ptr(x), _lib.ptr(g), B, h, w, C, oh, ow, int(ac), _lib.ptr(L.w),
                                                      L.cout, _lib.ptr(L.dw), _lib.ptr(dx), _lib.ptr(ws), ws.numel(), s))
        ready = getattr(L, "on_grad_ready", None)
        if ready is not None:
            ready()
        return dx, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, padding):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.cfg = (k, s, padding)
        return dn.max_pool(x, k, s, padding)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        k, s, padding = ctx.cfg
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().mpsr_max_pool_grad(_lib.ptr(x), _lib.ptr(dy.contiguous()), B, H, W, C, k, s,
                                                 int(padding == "SAME"), _lib.ptr(dx), _lib.stream()))
        return dx, None, None, None


class ResizeBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners):
        ctx.in_shape = tuple(x.shape)
        ctx.cfg = (tuple(size), align_corners)
        return dn.resize_bilinear(x, tuple(size), align_corners)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.in_shape
        (oh, ow), ac = ctx.cfg
        if (H, W) == (oh, ow):
            return dy, None, None
        dx = torch.empty(ctx.in_shape, dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().mpsr_resize_bilinear_grad(_lib.ptr(dy.contiguous()), B, H, W, C, oh, ow, int(ac),
                                                        _lib.ptr(dx), _lib.stream()))
        return dx, None, None


BN_MASK_FROM_Z = True  # A/B switch (tools/train_bench.py --bn-mask-from-y): BatchNorm's backward rebuilds the ReLU mask from z


class BatchNormReluFn(torch.autograd.Function):
    """y = relu((z - mean) / sqrt(var + eps) + beta) with BATCH statistics over (N, H, W) (slim.batch_norm,
    is_training=True, scale=False, net_builder.py:76-87); updates the layer's moving statistics like the fused
    TensorFlow kernel (biased variance to normalise, unbiased into the moving average); deposits d(beta) into L.db."""

    @staticmethod
    def forward(ctx, z, layer, token):
        st = layer.batch_norm
        z = z.contiguous()
        C = z.shape[-1]
        M = z.numel() // C
        lib = _lib.lib()
        sums = torch.empty((2, C), dtype=torch.float64, device=z.device)
        _lib.check(lib.mpsr_batch_norm_stats(_lib.ptr(z), M, C, sums[0].data_ptr(), sums[1].data_ptr(), _lib.stream()))
        Mtot = M
        if st._group() is None:
            # mean / variance, the moving statistics and the two float vectors of the apply pass in ONE launch
            # (r06: the torch expressions below were ~16 launches of 3-5 us per layer and step)
            stats = torch.empty((2, C), dtype=torch.float32, device=z.device)
            mean32, inv = stats[0], stats[1]
            _lib.check(lib.mpsr_batch_norm_finalize(sums[0].data_ptr(), sums[1].data_ptr(), _lib.ptr(z), M, C,
                                                    float(st.eps), float(st.decay), _lib.ptr(st.moving_mean),
                                                    _lib.ptr(st.moving_variance), mean32.data_ptr(), inv.data_ptr(),
                                                    _lib.stream()))
        else:
            # the kernel's sums are of (z - z[0]) per rank: un-shift them to plain first / second moments in fp64, pool
            # [sum z | sum z^2 | count] over the ranks, then mean and (biased) variance of the WHOLE step's batch
            sh = z.reshape(M, C)[0].double()
            pooled = torch.cat([sums[0] + M * sh, sums[1] + 2.0 * sh * sums[0] + M * sh * sh,
                                torch.full((1,), float(M), dtype=torch.float64, device=z.device)])
            st.all_reduce_sums(pooled)
            Mtot = pooled[2 * C]
            mean = pooled[:C] / Mtot
            var = torch.clamp(pooled[C:2 * C] / Mtot - mean * mean, min=0.0)
            with torch.no_grad():
                st.moving_mean.mul_(st.decay).add_(mean.float(), alpha=1.0 - st.decay)
                unbias = Mtot / torch.clamp(Mtot - 1, min=1.0)
                st.moving_variance.mul_(st.decay).add_((var * unbias).float(), alpha=1.0 - st.decay)
            mean32 = mean.float().contiguous()
            inv = torch.rsqrt(var + st.eps).float().contiguous()
        y = torch.empty_like(z)
        _lib.check(lib.mpsr_batch_norm_apply(_lib.ptr(z), M, C, _lib.ptr(mean32), _lib.ptr(inv), _lib.ptr(layer.b),
                                             int(layer.relu), _lib.ptr(y), _lib.stream()))
        ctx.layer = layer
        ctx.count = Mtot  # rows the statistics were taken over (all ranks' when pooled)
        ctx.save_for_backward(z, y if layer.relu else None, mean32, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = ctx.layer
        z, y, mean32, inv = ctx.saved_tensors
        dy = dy.contiguous()
        C = z.shape[-1]
        M = z.numel() // C
        lib = _lib.lib()
        sums = torch.empty((2, C), dtype=torch.float64, device=z.device)
        # with ReLU: the mask y > 0 is rebuilt from z inside both passes (the forward's own fused multiply-add, same
        # bits) instead of streaming y through them -- 2 of the 7 tensor-sized streams of a layer's backward less
        from_z = L.relu and BN_MASK_FROM_Z
        if from_z:
            _lib.check(lib.mpsr_batch_norm_grad_sums_z(_lib.ptr(dy), _lib.ptr(z), M, C, _lib.ptr(mean32), _lib.ptr(inv),
                                                       _lib.ptr(L.b), sums[0].data_ptr(), sums[1].data_ptr(),
                                                       _lib.stream()))
        else:
            _lib.check(lib.mpsr_batch_norm_grad_sums(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C, _lib.ptr(mean32),
                                                     _lib.ptr(inv), sums[0].data_ptr(), sums[1].data_ptr(),
                                                     _lib.stream()))
        # d(beta) is this rank's share (the gradient all-reduce pools it like every other parameter's); the two means of
        # the normalisation's backward are over the rows the statistics came from
        if L.batch_norm._group() is None:
            means = torch.empty((2, C), dtype=torch.float32, device=z.device)
            _lib.check(lib.mpsr_batch_norm_grad_finalize(sums[0].data_ptr(), sums[1].data_ptr(), float(ctx.count), C,
                                                         _lib.ptr(L.db), means[0].data_ptr(), means[1].data_ptr(),
                                                         _lib.stream()))
        else:
            L.db.add_(sums[0].float())
            L.batch_norm.all_reduce_sums(sums)
            means = (sums / ctx.count).float().contiguous()
        dz = torch.empty_like(z)
        if from_z:
            _lib.check(lib.mpsr_batch_norm_grad_z(_lib.ptr(dy), _lib.ptr(z), M, C, _lib.ptr(mean32), _lib.ptr(inv),
                                                  _lib.ptr(L.b), means[0].data_ptr(), means[1].data_ptr(), _lib.ptr(dz),
                                                  _lib.stream()))
        else:
            _lib.check(lib.mpsr_batch_norm_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C, _lib.ptr(mean32),
                                                _lib.ptr(inv), means[0].data_ptr(), means[1].data_ptr(), _lib.ptr(dz),
                                                _lib.stream()))
        return dz, None, None


class CropAndResizeFn(torch.autograd.Function):
    """tf.image.crop_and_resize, differentiable w.r.t. the image (boxes are inputs of the graph)."""

    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop_size, extrapolation_value):
        image = image.contiguous()
        boxes = boxes.to(torch.float32).contiguous()
        box_ind = box_ind.contiguous().int() if box_ind is not None else None
        ctx.save_for_backward(boxes, box_ind)
        ctx.cfg = (tuple(image.shape), tuple(crop_size))
        return dn.crop_and_resize(image, boxes, box_ind, tuple(crop_size), extrapolation_value)

    @staticmethod
    def backward(ctx, dy):
        boxes, box_ind = ctx.saved_tensors
        (nimg, H, W, C), (ch, cw) = ctx.cfg
        dimg = torch.empty((nimg, H, W, C), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().mpsr_crop_and_resize_grad(_lib.ptr(dy.contiguous()), nimg, H, W, C, _lib.ptr(boxes),
                                                        _lib.ptr(box_ind), boxes.shape[0], ch, cw, _lib.ptr(dimg),
                                                        _lib.stream()))
        return dimg, None, None, None, None


_TOKEN = {}


def _token(device):
    if device not in _TOKEN:
        _TOKEN[device] = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
    return _TOKEN[device]


def conv2d(x, layer, residual=None):
    """One layer: convolution (+ bias + residual + ReLU), or convolution -> training-mode BatchNorm -> ReLU when the
    layer carries a BatchNormState."""
    y = Conv2dFn.apply(x, residual, layer, _token(x.device))
    if layer.batch_norm is not None:
        y = BatchNormReluFn.apply(y, layer, _token(x.device))
    return y


def upsampled_conv_applies(x_shape, layer, size, align_corners=True):
    """True when resize -> 3x3 conv with `layer` can run (forward AND backward) as the single tap-GEMM operator."""
    B, h, w, C = x_shape
    return layer.kh == 3 and layer.kw == 3 and layer.dilation == 1 and _lib.get_conv_math() == "fp32" and \
        _lib.lib().mpsr_conv3x3_upsampled_applies(B, h, w, C, size[0], size[1], layer.cout, int(align_corners)) == 2


def upsampled_conv2d(x, layer, size, align_corners=True):
    """resize_bilinear(x, size) -> conv(layer) (-> training-mode BatchNorm -> ReLU when the layer carries a
    BatchNormState) without the upsampled map; see UpsampledConv2dFn."""
    y = UpsampledConv2dFn.apply(x, layer, tuple(size), align_corners, _token(x.device))
    if layer.batch_norm is not None:
        y = BatchNormReluFn.apply(y, layer, _token(x.device))
    return y


def bottleneck(x, c1, c2, c3, shortcut=None, in_link=None, out_link=None):
    """One bottleneck unit of frozen-BatchNorm (folded) layers as a single autograd node (BottleneckFn).  in_link: x is
    the output of the previous unit of a chain (created with the same UnitLink as its out_link) and feeds NOTHING else."""
    for L in (c1, c2, c3, shortcut):
        if L is not None and L.batch_norm is not None:
            raise _lib.InvalidArgumentError("bottleneck(): the trunk's layers are trained in BatchNorm-folded form")
    return BottleneckFn.apply(x, c1, c2, c3, shortcut, _token(x.device), in_link, out_link)


def max_pool(x, k, s, padding):
    return MaxPoolFn.apply(x, k, s, padding)


def resize_bilinear(x, size, align_corners):
    return ResizeBilinearFn.apply(x, size, align_corners)


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value=0.0):
    return CropAndResizeFn.apply(image, boxes, box_ind, crop_size, extrapolation_value)
