"""Device-resident packed weights and scratch for the network entry points of libmonopsr_hip.so."""
import ctypes

import numpy as np
import torch

from monopsr_amd import _lib
from monopsr_amd.core import weights as W


def _layer_array(records):
    arr = (_lib.Layer * len(records))()
    for i, r in enumerate(records):
        arr[i] = _lib.Layer(r["cin"], r["cout"], r["kh"], r["kw"], r["dilation"], r["relu"], r["w_off"], r["b_off"])
    return arr


class PackedPart:
    def __init__(self, blob, records, device):
        self.blob = torch.from_numpy(blob).to(device)
        self.records = records
        self.layers = _layer_array(records)
        self.n = len(records)


class FilterCache:
    """Transformed filters of one weight blob kept across calls (mpsr_net_opts.filter_cache + filter_cache_tags): filled
    by the first native call, only read afterwards."""

    def __init__(self, part, device):
        n = _lib.lib().mpsr_filter_cache_floats(part.layers, part.n)
        self.buf = torch.empty((int(n),), dtype=torch.float32, device=device) if n else None
        # per-layer notes of what each slice holds: the library keeps them, so the cache survives a change of the
        # arithmetic mode / Winograd policy / batch size (another kernel, another form of the filters) by itself
        self.tags = (ctypes.c_int32 * part.n)()
        self.filled = False

    def opts(self, key=None, event=None, math=None, winograd_policy=None):
        """Options of one native call.  The caller reports the call's outcome with `mark_filled()` once it has returned
        MPSR_OK: a failed first call must not leave a cache that later calls trust."""
        o = _lib.NetOpts()
        if self.buf is not None:
            o.filter_cache, o.filter_cache_floats = self.buf.data_ptr(), self.buf.numel()
            o.filter_cache_valid = 1 if self.filled else 0
            o.filter_cache_tags = self.tags
            self._tags_before = bytes(self.tags)  # (mark_filled: did this call re-write a slice?)
        o.ready_event = event
        o.math = _lib.CALL_MATH[math]
        o.winograd_policy = _lib.CALL_WINOGRAD[winograd_policy]
        return o

    def mark_filled(self):
        """The native call that was handed this cache returned MPSR_OK: its launches (which fill, or RE-fill, slices) are
        queued on the calling stream.  Calls on OTHER streams must not read the slices before those launches have run:
        the LAST stream that wrote the cache is remembered with an event and later callers on another stream wait for
        it.  A call re-writes slices whenever the kernel choice of a layer changes (arithmetic mode, Winograd policy, a
        batch size on the other side of a kernel's threshold): the library records that in `tags`, so a call after
        which the tags differ from the snapshot `opts()` took was a writer.  One DeviceNet is driven by
        one host thread at a time (`tags` is plain host memory the library updates without a lock;
        clone_with_own_scratch() gives every thread / stream its own cache)."""
        if self.buf is None:
            return
        wrote = not self.filled or bytes(self.tags) != getattr(self, "_tags_before", None)
        self.filled = True
        if wrote:
            self.fill_stream = torch.cuda.current_stream(self.buf.device)
            self.fill_event = torch.cuda.Event()
            self.fill_event.record(self.fill_stream)

    def before_call(self):
        """Orders a call on another stream than the last one that wrote the cache behind that write."""
        if self.filled and getattr(self, "fill_event", None) is not None:
            cur = torch.cuda.current_stream(self.buf.device)
            if cur != self.fill_stream:
                cur.wait_event(self.fill_event)


class Workspace:
    """Grow-only scratch buffer on one device (allocation stays outside the C ABI)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty((int(nbytes),), dtype=torch.uint8, device=self.device)
        return self.buf


class DeviceNet:
    """Packs a TF-named weight dict (monopsr_amd.core.weights) and runs trunk / decoder / heads through the C ABI."""

    def __init__(self, weights, device="cuda", width_div=1, full_trunk=False):
        self.device = torch.device(device)
        self.width_div = width_div
        self.crop_trunk = PackedPart(*W.pack_trunk(weights, W.CROP_SCOPE, width_div), self.device)
        self.full_trunk = (PackedPart(*W.pack_trunk(weights, W.FULL_SCOPE, width_div), self.device)
                           if full_trunk else None)
        self.decoder = PackedPart(*W.pack_decoder(weights, width_div), self.device)
        feat_elems = weights["output/proposal_fc/proposal_fc/img_fc/weights"].shape[0]
        self.heads = PackedPart(*W.pack_heads(weights, feat_elems), self.device)
        self.feat_elems = feat_elems
        self.ws_trunk = Workspace(self.device)
        self.ws_trunk_full = Workspace(self.device)  # own scratch: the two trunks may run on two streams at once
        self.side_stream = None                      # created on first use (full-image branch, net_builder)
        self.ws_dec = Workspace(self.device)
        self.ws_heads = Workspace(self.device)
        self._weights = weights
        self._fc_cache = {}
        self.fcache = {}          # FilterCache per weight blob, created on first use
        self.heads_stream = None  # second HIP stream of forward_instances (created on first use)
        # arithmetic mode / Winograd policy of THIS net's calls (mpsr_net_opts, ABI 5): None = the process-wide default,
        # "fp32" / "bf16x3", "auto" / "off".  Two nets with different settings can run concurrently on two threads.
        self.math = None
        self.winograd_policy = None

    def _filter_cache(self, which, part):
        """Weights of a DeviceNet never change after packing: the Winograd-transformed filters are computed once."""
        d = self.__dict__.setdefault("fcache", {})
        if d.get(which) is None:
            d[which] = FilterCache(part, self.device)
        return d[which]

    def clone_with_own_scratch(self):
        """Same device weights, separate workspaces: one clone per HIP stream when instance shards run
        concurrently on several streams."""
        import copy
        other = copy.copy(self)
        other.ws_trunk, other.ws_dec, other.ws_heads = Workspace(self.device), Workspace(self.device), \
            Workspace(self.device)
        other.ws_trunk_full, other.side_stream = Workspace(self.device), None
        other.fcache = {}  # (a shared cache would be filled by two streams at once)
        other.heads_stream = None
        return other

    # ------------------------------------------------------------------ single FC layers (output builder)
    def fully_connected(self, x, name, relu):
        """slim.fully_connected `name` on x (B, fin) through the HIP GEMM; the input is zero-padded to a multiple
        of 32 channels (what the fused heads entry point pads its concat rows to: the same few-row FC kernel then
        serves both, csrc/pointwise.hip)."""
        if name not in self._fc_cache:
            w = self._weights[name + "/weights"]  # (in, out)
            kpad = (w.shape[0] + 31) // 32 * 32
            w_ok = np.zeros((w.shape[1], kpad), np.float32)
            w_ok[:, :w.shape[0]] = w.T
            self._fc_cache[name] = (torch.from_numpy(w_ok).to(self.device),
                                    torch.from_numpy(self._weights[name + "/biases"].astype(np.float32)).to(self.device),
                                    w.shape[0], kpad)
        w_ok, bias, fin, kpad = self._fc_cache[name]
        B = x.shape[0]
        if x.shape[1] != fin:
            raise _lib.InvalidArgumentError("%s expects %d input features, got %d" % (name, fin, x.shape[1]))
        if kpad != fin:
            xp = torch.zeros((B, kpad), dtype=torch.float32, device=self.device)
            xp[:, :fin] = x
            x = xp
        split = 8 if kpad >= 8192 else 1
        return conv2d(x.reshape(B, 1, 1, kpad), w_ok, bias, None, 1, 1, 1, relu, split_k=split).reshape(B, -1)

    # ------------------------------------------------------------------ trunk
    MAX_CHUNK = 512  # instances per native call: keeps every activation of one call below the 4 GiB that the
    #                  kernels' 32-bit buffer offsets address (largest: B x 48 x 48 x 256 fp32 = 2.4 MB per instance)

    def trunk(self, img, which="crop"):
        """img (B,H,W,3) -> block3 (B,H/4,W/4,C)."""
        part = self.crop_trunk if which == "crop" else self.full_trunk
        if part is None:
            raise _lib.MpsrError("DeviceNet was built without the full-image trunk")
        img = img.contiguous()
        B, H, Wd, _ = img.shape
        if B > self.MAX_CHUNK:
            return torch.cat([self.trunk(img[i:i + self.MAX_CHUNK], which) for i in range(0, B, self.MAX_CHUNK)], 0)
        oh, ow = (H + 6 - 7) // 2 + 1, (Wd + 6 - 7) // 2 + 1
        ph, pw = (oh + 1) // 2, (ow + 1) // 2
        cout = part.records[-1]["cout"]
        out = torch.empty((B, ph, pw, cout), dtype=torch.float32, device=self.device)
        lib = _lib.lib()
        nbytes = lib.mpsr_trunk_workspace_bytes(B, H, Wd)
        ws = (self.ws_trunk if which == "crop" else self.ws_trunk_full).get(nbytes)
        fc = self._filter_cache(which, part)
        fc.before_call()
        opts = fc.opts((B, H, Wd), math=getattr(self, 'math', None), winograd_policy=getattr(self, 'winograd_policy', None))
        _lib.check(lib.mpsr_trunk_fwd_ex(_lib.ptr(img), B, H, Wd, _lib.ptr(part.blob), part.layers, part.n,
                                         _lib.ptr(out), _lib.ptr(ws), ws.numel(), ctypes.byref(opts), _lib.stream()))
        fc.mark_filled()
        return out

    # ------------------------------------------------------------------ squash + decoder (+ xyz head)
    def squash_decoder(self, crop_feat, full_feat, map_size=(48, 48), want_feat_map=True, want_xyz=True,
                       box3d_event=None):
        """-> (features_for_box_3d, features_for_map or None, inst_xyz_map_local).  box3d_event: a recorded
        torch.cuda.Event that the native call re-records on the current stream as soon as features_for_box_3d is
        complete (single-chunk batches only) -- another stream may then run the heads next to the map decoder."""
        crop_feat, full_feat = crop_feat.contiguous(), full_feat.contiguous()
        B, fh, fw, _ = crop_feat.shape
        if B > self.MAX_CHUNK:
            parts = [self.squash_decoder(crop_feat[i:i + self.MAX_CHUNK], full_feat[i:i + self.MAX_CHUNK], map_size,
                                         want_feat_map, want_xyz) for i in range(0, B, self.MAX_CHUNK)]
            if box3d_event is not None:
                box3d_event.record()
            return tuple(torch.cat([p[k] for p in parts], 0) if parts[0][k] is not None else None for k in range(3))
        mh, mw = map_size
        recs = self.decoder.records
        csq, c3 = recs[1]["cout"], recs[5]["cout"]
        feat_box = torch.empty((B, fh // 2, fw // 2, csq), dtype=torch.float32, device=self.device)
        feat_map = torch.empty((B, mh, mw, c3), dtype=torch.float32, device=self.device) if want_feat_map else None
        xyz = torch.empty((B, mh, mw, recs[6]["cout"]), dtype=torch.float32, device=self.device)
        lib = _lib.lib()
        nbytes = lib.mpsr_decoder_workspace_bytes(B, fh, fw, mh, mw)
        ws = self.ws_dec.get(nbytes)
        fc = self._filter_cache("dec", self.decoder)
        fc.before_call()
        opts = fc.opts((B, fh, fw, mh, mw, want_feat_map), box3d_event.cuda_event if box3d_event is not None else None,
                       math=getattr(self, 'math', None), winograd_policy=getattr(self, 'winograd_policy', None))
        _lib.check(lib.mpsr_squash_decoder_fwd_ex(_lib.ptr(crop_feat), _lib.ptr(full_feat), B, fh, fw, mh, mw,
                                                  _lib.ptr(self.decoder.blob), self.decoder.layers, self.decoder.n,
                                                  _lib.ptr(feat_box), _lib.ptr(feat_map), _lib.ptr(xyz), _lib.ptr(ws),
                                                  ws.numel(), ctypes.byref(opts), _lib.stream()))
        fc.mark_filled()
        return feat_box, feat_map, xyz

    # ------------------------------------------------------------------ the whole instance path
    def forward_instances(self, crops, full_feat, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset,
                          map_size=(48, 48), overlap_heads=False, **head_kw):
        """crops (B,48,48,3) + full-image feature crop (B,12,12,1024) + the box scalars -> (inst_xyz_map_local,
        head outputs): trunk, squash + map decoder + xyz head, FC heads (net_builder.py:30-96 +
        monopsr_output_builder.py:95-661).  The heads only need features_for_box_3d, which exists after the squash
        layer's max-pool: with `overlap_heads` they run on a second HIP stream NEXT TO the map decoder (their ~10 small
        launches fill the decoder kernels' partially occupied rounds) and the current stream waits for them at the
        end; results are bit-identical to the one-stream order."""
        crop_feat = self.trunk(crops)
        if not overlap_heads:
            fb, _, xyz = self.squash_decoder(crop_feat, full_feat, map_size, want_feat_map=False)
            return xyz, self.heads_fwd(fb, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, **head_kw)
        main = torch.cuda.current_stream(self.device)
        if self.heads_stream is None:
            self.heads_stream = concurrent_stream(self.device)
            self._box3d_event = torch.cuda.Event()
        ev = self._box3d_event
        ev.record(main)  # (creates the native event on first use; the native call records it again, later)
        fb, _, xyz = self.squash_decoder(crop_feat, full_feat, map_size, want_feat_map=False, box3d_event=ev)
        side = self.heads_stream
        side.wait_event(ev)
        with torch.cuda.stream(side):
            out = self.heads_fwd(fb, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, **head_kw)
        fb.record_stream(side)
        main.wait_stream(side)
        for t in out.values():
            t.record_stream(main)
        return xyz, out

    # ------------------------------------------------------------------ N images x their boxes in one pass
    def forward_images(self, images, boxes_2d_norm, box_ind, boxes_2d, cam_p, view_angs, class_idx, mean_lwh,
                       cen_z_offset, img_roi_size=(48, 48), map_roi_size=(48, 48), resized_full_img_shape=(160, 608),
                       **head_kw):
        """The reference's step (one preprocessed image + its boxes: monopsr_model.py:222-237, net_builder.py:44-60,
        configs/monopsr_model_000.yaml:14-17) for N images at once: images (N,H,W,3) preprocessed, boxes of all images
        concatenated -- boxes_2d_norm (B,4) normalised by each box's OWN original image size, box_ind (B) int32 = image
        of the box, boxes_2d (B,4) pixels, cam_p (N,3,4), the per-box scalars (B).  One proposal crop_and_resize routed
        by box_ind, ONE full-image trunk call with batch N (block3's 1x1 layers then have N x 6080 rows: at N = 1 they
        fill half the chip), one crop-trunk call with batch B, one feature crop_and_resize routed by box_ind, one
        squash / decoder / heads call.  -> (inst_xyz_map_local (B,48,48,3), head outputs); every box's outputs equal
        the single-image call's (each operator is per box / per image)."""
        if self.full_trunk is None:
            raise _lib.MpsrError("DeviceNet was built without the full-image trunk")
        box_ind = box_ind.to(dtype=torch.int32, device=self.device).contiguous()
        half = (map_roi_size[0] // 2, map_roi_size[1] // 2)
        main = torch.cuda.current_stream(self.device)
        if self.side_stream is None:
            self.side_stream = concurrent_stream(self.device)
        side = self.side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):  # the two trunks are independent (net_builder.py:44-52): second stream
            full_img = resize_bilinear(images, tuple(resized_full_img_shape), align_corners=True)
            full_feat_map = self.trunk(full_img, "full")
            large = crop_and_resize(full_feat_map, boxes_2d_norm, box_ind, half)
            full_feat = max_pool(large, 2, 2, "VALID")
        crops = crop_and_resize(images, boxes_2d_norm, box_ind, tuple(img_roi_size))
        crop_feat = self.trunk(crops, "crop")
        main.wait_stream(side)
        full_feat.record_stream(main)
        fb, _, xyz = self.squash_decoder(crop_feat, full_feat, tuple(map_roi_size), want_feat_map=False)
        out = self.heads_fwd(fb, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, cam_index=box_ind,
                             **head_kw)
        return xyz, out

    # ------------------------------------------------------------------ heads
    def heads_fwd(self, feat_box3d, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset,
                  image_shape=(320, 1216), max_depth=45.0, num_classes=1, num_alpha_bins=12,
                  cen_y_norm=1.666754, cen_y_class_offset=0.0648, cam_index=None):
        """cam_p (3,4) / (12,): one projection matrix for all boxes (the reference's step: one image); or (n_cams,3,4)
        with cam_index (B) int32: the boxes of several images in one call (mpsr_heads_fwd_cams)."""
        B = feat_box3d.shape[0]
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        feat = feat_box3d.contiguous().reshape(B, -1)
        boxes_2d = boxes_2d.to(**f32).contiguous()
        cam_p = cam_p.to(**f32).contiguous().reshape(-1, 12)
        n_cams = cam_p.shape[0]
        if cam_index is None:
            if n_cams != 1:
                raise _lib.InvalidArgumentError("heads_fwd: %d projection matrices need a cam_index" % n_cams)
        else:
            cam_index = cam_index.to(dtype=torch.int32, device=dev).contiguous().reshape(B)
        view = view_angs.to(**f32).contiguous().reshape(B)
        cls = class_idx.to(dtype=torch.int32, device=dev).contiguous().reshape(B)
        mean_lwh = mean_lwh.to(**f32).contiguous()
        z_off = cen_z_offset.to(**f32).contiguous().reshape(B)
        nb = num_alpha_bins
        out = {
            "lwh": torch.empty((B, 3), **f32), "lwh_offs": torch.empty((B, 3), **f32),
            "alpha_bins": torch.empty((B, nb), **f32), "alpha_regs": torch.empty((B, nb), **f32),
            "prop_cen_z": torch.empty((B, 1), **f32), "cen_y": torch.empty((B, 1), **f32),
            "cen_y_offs": torch.empty((B, 1), **f32), "cen_z": torch.empty((B, 1), **f32),
            "cen_z_offs": torch.empty((B, 1), **f32), "cen_x": torch.empty((B, 1), **f32),
            "centroids": torch.empty((B, 3), **f32),
        }
        consts = _lib.HeadConsts(float(image_shape[0]), float(image_shape[1]), float(max_depth), float(cen_y_norm),
                                 float(cen_y_class_offset), int(num_classes), int(nb))
        outs = _lib.HeadOutputs(*[_lib.ptr(out[k]) for k, _ in _lib.HeadOutputs._fields_])
        lib = _lib.lib()
        nbytes = lib.mpsr_heads_workspace_bytes(B, feat.shape[1])
        ws = self.ws_heads.get(nbytes)
        _lib.check(lib.mpsr_heads_fwd_cams(_lib.ptr(feat), B, feat.shape[1], _lib.ptr(boxes_2d), _lib.ptr(cam_p), n_cams,
                                           _lib.ptr(cam_index), _lib.ptr(view), _lib.ptr(cls), _lib.ptr(mean_lwh),
                                           _lib.ptr(z_off), ctypes.byref(consts), _lib.ptr(self.heads.blob),
                                           self.heads.layers, self.heads.n, ctypes.byref(outs), _lib.ptr(ws), ws.numel(),
                                           _lib.stream()))
        out["view_ang"] = view.reshape(B, 1)
        return out


# ---------------------------------------------------------------------- single-layer helpers (tests, tools)

_SCHED_SCRATCH = {}
_SCRATCH_CACHES = []  # every per-(device, stream) scratch cache of the package (pinned_stream_scratch walks them)


def pinned_stream_scratch(device, stream):
    """The scratch tensors the caches hold for (device, stream), for a caller that must keep them alive: a captured HIP
    graph has their addresses baked into its launches -- holding the tensors keeps the memory from being handed to
    anyone else when a cache later evicts the entry (more than `keep` streams)."""
    key = (device.index, stream.cuda_stream)
    return [c[key] for c in [_SCHED_SCRATCH] + _SCRATCH_CACHES if key in c]


def stream_scratch(cache, device, floats, keep=8):
    """A float32 scratch of at least `floats` elements for the current stream of `device`, kept in `cache` per (device,
    stream): launches on one stream are ordered, so they can share it.  At most `keep` streams are remembered (the
    oldest entry goes first: its memory returns to the caching allocator once the launches using it have run), so
    short-lived streams do not pin 100 MB each for the life of the process."""
    key = (device.index, _lib.stream(device.index))
    ws = cache.get(key)
    if ws is None or ws.numel() < floats:
        cache.pop(key, None)
        while len(cache) >= keep:
            old = cache.pop(next(iter(cache)))
            old.record_stream(torch.cuda.current_stream(device))
        ws = cache[key] = torch.empty((floats,), dtype=torch.float32, device=device)
    return ws


def runs_concurrently(main, cand, cycles=400000):
    """Do launches on `cand` overlap launches on `main`?  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
    4 by default; a new stream takes the least-used one) and a hardware queue runs its packets one after the other: two
    streams that share a queue do not overlap at all.  Which queue a stream lands on depends on what else the process has
    created -- with an RCCL communicator alive the weight-gradient stream of a training step landed on the main stream's
    queue and the step went from 45.0 to 48.9 ms, slower than with one stream (r06).  Measured, not guessed: a spin kernel
    on each stream behind a common event takes one spin when they overlap and two when they are serialised."""
    dev = cand.device
    def one(both):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        e0.record(main)
        cand.wait_event(e0)
        with torch.cuda.stream(main):
            torch.cuda._sleep(cycles)
            e1.record(main)
        if both:
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
                e2.record(cand)
        torch.cuda.synchronize(dev)
        return max(e0.elapsed_time(e1), e0.elapsed_time(e2) if both else 0.0)
    one(True)  # (first use of a stream: queue creation, not timing)
    alone = min(one(False), one(False))
    together = min(one(True), one(True))
    return together < 1.5 * alone


def concurrent_stream(device, main=None, priority=0, tries=12):
    """A new stream of `device` that really overlaps `main` (default: the current stream): candidates are created until one
    passes runs_concurrently (torch hands out pool streams round-robin, so they land on different hardware queues); the
    first one if none does (a single hardware queue).  Costs a few milliseconds, once per side stream."""
    device = torch.device(device)
    if main is None:
        main = torch.cuda.current_stream(device)
    first = None
    for _ in range(tries):
        s = torch.cuda.Stream(device=device, priority=priority)
        if first is None:
            first = s
        if s.cuda_stream != main.cuda_stream and runs_concurrently(main, s):
            return s
    return first


def blocked_by_collectives(stream, group=None, helper=None, long_cycles=6000000, short_cycles=200000):
    """Does a collective that is waiting for its inputs hold up launches on `stream`?  It does when the communicator's own
    stream shares `stream`'s hardware queue: the queue is in order, so the collective's "wait for the producer's event"
    packet sits in front of whatever `stream` enqueues next -- and on a real node the collective itself (milliseconds)
    would.  Measured on any world size, ONE tiny all_reduce per call (every rank must make the same calls): a helper stream
    spins ~3 ms and issues the all_reduce behind the spin, so the communicator's stream waits ~3 ms; a short spin enqueued
    on `stream` meanwhile finishes in its own time, or after those 3 ms.  -> True when it was held up."""
    import torch.distributed as dist
    dev = stream.device
    if helper is None:
        helper = concurrent_stream(dev, stream)
    t = torch.ones(4, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(stream)
    helper.wait_event(e0)
    with torch.cuda.stream(helper):
        torch.cuda._sleep(long_cycles)
        work = dist.all_reduce(t, group=group, async_op=True)
    with torch.cuda.stream(stream):
        torch.cuda._sleep(short_cycles)
        e1.record(stream)
    work.wait()
    torch.cuda.synchronize(dev)
    took = e0.elapsed_time(e1)
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(helper):
        e2.record(helper)
        torch.cuda._sleep(long_cycles)
        e3.record(helper)
    torch.cuda.synchronize(dev)
    return took > 0.5 * e2.elapsed_time(e3)


_CONSTANTS = {}


def device_constant(values, device, dtype=torch.float32):
    """A small constant tensor on `device`, uploaded ONCE per (values, dtype, device): a host -> device copy inside a
    step is a synchronising call and cannot be captured into a HIP graph."""
    arr = np.asarray(values)
    key = (arr.shape, arr.astype(np.float64).tobytes(), str(dtype), str(device))
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = torch.as_tensor(arr, dtype=dtype, device=device)
    return t


def conv2d(x, w_ok, bias=None, residual=None, kh=1, kw=1, dilation=1, relu=False, split_k=1, math=None,
           winograd_policy=None):
    """mpsr_conv2d_nhwc_f32 on torch tensors: x (B,H,W,C), w_ok (N, kh*kw*C).  math ("fp32" / "bf16x3") and
    winograd_policy ("auto" / "off"): options of THIS call (mpsr_conv2d_nhwc_f32_ex); None = the process-wide defaults."""
    x, w_ok = x.contiguous(), w_ok.contiguous()
    B, H, Wd, C = x.shape
    N = w_ok.shape[0]
    y = torch.empty((B, H, Wd, N), dtype=torch.float32, device=x.device)
    # split_k == 0: the library schedules the launch itself (stream-K / Winograd) inside a scratch that is kept per
    # (device, stream) -- launches on one stream are ordered, so they can share it, and a training step issues
    # hundreds of these calls from Python
    if split_k == 0:
        nws = _lib.lib().mpsr_conv2d_scratch_floats(B, H, Wd, N)
        ws = stream_scratch(_SCHED_SCRATCH, x.device, nws)
    else:
        nws = split_k * B * H * Wd * N if split_k > 1 else 0
        ws = torch.empty((nws,), dtype=torch.float32, device=x.device) if nws else None
    if math is None and winograd_policy is None:
        _lib.check(_lib.lib().mpsr_conv2d_nhwc_f32(
            _lib.ptr(x), B, H, Wd, C, _lib.ptr(w_ok), _lib.ptr(bias.contiguous()) if bias is not None else None,
            _lib.ptr(residual.contiguous()) if residual is not None else None, _lib.ptr(y), N, kh, kw, dilation,
            int(relu), split_k, _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream()))
        return y
    opts = _lib.ConvOpts(_lib.CALL_MATH[math], _lib.CALL_WINOGRAD[winograd_policy])
    _lib.check(_lib.lib().mpsr_conv2d_nhwc_f32_ex(
        _lib.ptr(x), B, H, Wd, C, _lib.ptr(w_ok), _lib.ptr(bias.contiguous()) if bias is not None else None,
        _lib.ptr(residual.contiguous()) if residual is not None else None, _lib.ptr(y), N, kh, kw, dilation,
        int(relu), split_k, _lib.ptr(ws), ws.numel() if ws is not None else 0, ctypes.byref(opts), _lib.stream()))
    return y


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value=0.0):
    image, boxes = image.contiguous(), boxes.to(torch.float32).contiguous()
    nimg, H, Wd, C = image.shape
    nb = boxes.shape[0]
    ind = box_ind.to(torch.int32).contiguous() if box_ind is not None else None
    out = torch.empty((nb, crop_size[0], crop_size[1], C), dtype=torch.float32, device=image.device)
    _lib.check(_lib.lib().mpsr_crop_and_resize(_lib.ptr(image), nimg, H, Wd, C, _lib.ptr(boxes), _lib.ptr(ind), nb,
                                               crop_size[0], crop_size[1], float(extrapolation_value), _lib.ptr(out),
                                               _lib.stream()))
    return out


def resize_bilinear(x, size, align_corners=False):
    x = x.contiguous()
    B, H, Wd, C = x.shape
    if (H, Wd) == tuple(size):
        return x
    out = torch.empty((B, size[0], size[1], C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mpsr_resize_bilinear(_lib.ptr(x), B, H, Wd, C, size[0], size[1], int(align_corners),
                                               _lib.ptr(out), _lib.stream()))
    return out


def conv3x3_upsampled(x, size, w_ok, bias=None, relu=False, align_corners=True):
    """tf.image.resize_bilinear(x, size, align_corners) -> 3x3 SAME conv (weights (N, 9 C) as for conv2d), without
    forming the upsampled map (csrc/upconv.hip; net_builder.py:72-77).  x (B,h,w,C) -> (B,size[0],size[1],N)."""
    x = x.contiguous()
    B, h, w, C = x.shape
    N = w_ok.shape[0]
    lib = _lib.lib()
    nws = lib.mpsr_conv3x3_upsampled_scratch_floats(B, h, w, C, N)
    ws = torch.empty((nws,), dtype=torch.float32, device=x.device)
    out = torch.empty((B, size[0], size[1], N), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpsr_conv3x3_upsampled_f32(_lib.ptr(x), B, h, w, C, size[0], size[1], int(align_corners),
                                              _lib.ptr(w_ok.contiguous()), _lib.ptr(bias), int(relu), _lib.ptr(out), N,
                                              _lib.ptr(ws), ws.numel(), _lib.stream()))
    return out


def max_pool(x, k, s, padding="VALID"):
    x = x.contiguous()
    B, H, Wd, C = x.shape
    if padding == "SAME":
        oh, ow = -(-H // s), -(-Wd // s)
    else:
        oh, ow = (H - k) // s + 1, (Wd - k) // s + 1
    out = torch.empty((B, oh, ow, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mpsr_max_pool(_lib.ptr(x), B, H, Wd, C, k, s, int(padding == "SAME"), _lib.ptr(out),
                                        _lib.stream()))
    return out
