"""Instance-level data parallelism for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests).

The reference has no distributed code at all.  Every instance (crop, cloud pair) is independent
through trunk, decoder, heads, Chamfer and EMD -- BatchNorm runs on moving statistics -- so the batch is split into
contiguous shards, weights are replicated, and the forward / metric path needs NO collective; results are gathered
once at the end.  The only real exchange step of a training iteration is the gradient all-reduce, provided here as
a bucketed, asynchronous all-reduce over a flat fp32 buffer (100,204,832 parameters = 401 MB for the full model).
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU), so buckets are kept large (64 MiB default): few,
big messages keep every link busy and let RCCL pick its direct all-to-all style algorithms.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world_size):
    """Contiguous split of n instances; the first n % world_size ranks take one extra."""
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sample(sample, rank, world_size, per_instance_keys=None):
    """Slice the per-instance entries of a MonoPSRModel.build sample dict (first dim = instance); shared entries
    (camera matrix, full image) are passed through."""
    n = sample['boxes_2d'].shape[0]
    lo, hi = shard_range(n, rank, world_size)
    shared = {'cam_p', 'rgb_image'}
    out = {}
    for k, v in sample.items():
        per_inst = (k in per_instance_keys) if per_instance_keys is not None else (
            k not in shared and hasattr(v, 'shape') and len(v.shape) > 0 and v.shape[0] == n)
        out[k] = v[lo:hi] if per_inst else v
    return out


def gather_instances(t, total, group=None, force_collective=False):
    """All ranks receive the concatenation over ranks (in rank order) of a per-instance tensor whose shards follow
    shard_range(total, ...).  Uneven shards are padded to the largest one for the collective.  force_collective: issue
    the all_gather on a one-rank group too (an identity: the real RCCL call on a single-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return t
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    big = max(sizes)
    pad = torch.zeros((big,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def reduce_metric_sums(values, group=None, force_collective=False):
    """Sum scalar / small metric tensors over ranks (e.g. per-rank Chamfer sums and valid-pixel counts)."""
    t = torch.stack([v.reshape(()).to(torch.float64) for v in values])
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        dist.all_reduce(t, group=group)
    return t


class BucketedAllReduce:
    """Sum a flat fp32 gradient buffer over ranks in large buckets, asynchronously.

    start() launches one all-reduce per bucket in order (the heads' 42 M parameters come first in backward order,
    so the first buckets can go out while the trunk is still running); finish() waits for all of them and
    optionally averages.  On the GPU box the collectives run on RCCL's own stream and overlap with compute."""

    def __init__(self, flat, bucket_bytes=64 << 20, group=None):
        assert flat.dim() == 1
        self.flat = flat
        self.group = group
        n = max(1, bucket_bytes // flat.element_size())
        self.buckets = [(i, min(i + n, flat.numel())) for i in range(0, flat.numel(), n)]
        self._pending = []

    def start(self, upto=None):
        """Launch buckets not yet launched whose end is <= `upto` elements (all when None)."""
        done = len(self._pending)
        for lo, hi in self.buckets[done:]:
            if upto is not None and hi > upto:
                break
            self._pending.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))

    def finish(self, average=False):
        self.start()
        for w in self._pending:
            w.wait()
        self._pending = []
        if average:
            self.flat.div_(dist.get_world_size(self.group))
        return self.flat
