"""Debug aid (tests and tools only; nothing in the product path calls it): make every read of uninitialised device
memory deterministic.

`poison_uninitialised()` replaces torch.empty / torch.empty_like / Tensor.new_empty for the duration of a `with` block by
versions that fill floating-point results with NaN (integer results are left alone: a poisoned index would fault the
GPU instead of failing a comparison), and `poison_stream_scratch()` does the same to the per-(device, stream) scratch
buffers this package keeps across calls (the split-K slabs, Winograd-domain partial sums, transformed filters behind
them ...).  A kernel that reads an element nobody wrote -- a K-split "no zero-fill" path whose slices do not cover the
output, an accumulate-into-scratch launch without its memset -- then returns NaN every time instead of whatever the
allocator handed out, in eager steps as well as in replays of a captured graph (the fills are launches like any other
and are captured with the step)."""
import contextlib

import torch


def _nan_fill(t):
    if torch.is_tensor(t) and t.is_floating_point() and t.numel() and t.device.type == "cuda":
        t.fill_(float("nan"))
    return t


@contextlib.contextmanager
def poison_uninitialised():
    real_empty, real_empty_like, real_new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def empty(*a, **k):
        return _nan_fill(real_empty(*a, **k))

    def empty_like(*a, **k):
        return _nan_fill(real_empty_like(*a, **k))

    def new_empty(self, *a, **k):
        return _nan_fill(real_new_empty(self, *a, **k))

    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty
    try:
        yield
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = real_empty, real_empty_like, real_new_empty


def poison_stream_scratch():
    """NaN into every scratch buffer the package's per-stream caches hold right now (call between steps)."""
    from monopsr_amd.core import device_net as dn
    n = 0
    for cache in [dn._SCHED_SCRATCH] + list(dn._SCRATCH_CACHES):
        for t in cache.values():
            _nan_fill(t)
            n += 1
    return n


def poison_workspaces(net):
    """All-ones bytes (a NaN in every float) into the grow-only scratch of a DeviceNet (trunk / decoder / heads
    workspaces: activations, split-K slabs, partial sums).  The filter caches are NOT touched: they hold results."""
    n = 0
    for name in ("ws_trunk", "ws_trunk_full", "ws_dec", "ws_heads"):
        ws = getattr(net, name, None)
        if ws is not None and ws.buf is not None:
            ws.buf.fill_(255)
            n += 1
    return n
