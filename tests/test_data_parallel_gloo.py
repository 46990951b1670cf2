"""world_size-2 gloo test of the instance-sharded data-parallel helpers (runs on CPU, 127.0.0.1 rendezvous)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monopsr_amd.core import data_parallel as dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _per_instance_work(sample):
    """Stand-in for the per-instance path on CPU: anything that treats instances independently."""
    b = sample["boxes_2d"]
    cen = torch.stack([b[:, 0] + b[:, 2], b[:, 1] * 2 + sample["est_view_angs"], b[:, 3] - sample["cam_p"][0, 0]], 1)
    cloud = sample["clouds"] * 2 + cen[:, None, :]
    return cen, cloud


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)  # same seed on every rank: the global batch
        sample = {"boxes_2d": torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)),
                  "est_view_angs": torch.from_numpy(rng.standard_normal(n).astype(np.float32)),
                  "clouds": torch.from_numpy(rng.standard_normal((n, 16, 3)).astype(np.float32)),
                  "cam_p": torch.eye(3, 4) * 700}
        ref_cen, ref_cloud = _per_instance_work(sample)
        mine = dp.shard_sample(sample, rank, world)
        lo, hi = dp.shard_range(n, rank, world)
        assert mine["boxes_2d"].shape[0] == hi - lo and mine["cam_p"].shape == (3, 4)
        cen, cloud = _per_instance_work(mine)
        all_cen = dp.gather_instances(cen, n)
        all_cloud = dp.gather_instances(cloud, n)
        ok = torch.equal(all_cen, ref_cen) and torch.equal(all_cloud, ref_cloud)
        # metric sums
        s = dp.reduce_metric_sums([cloud.double().sum(), torch.tensor(float(hi - lo))])
        want = float(ref_cloud.double().sum())
        ok = ok and abs(float(s[0]) - want) <= 1e-9 * abs(want) and int(s[1]) == n
        # bucketed gradient all-reduce: 3 buckets, launched in two stages
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red = dp.BucketedAllReduce(flat, bucket_bytes=400 * 4)
        assert len(red.buckets) == 3
        red.start(upto=400)
        out = red.finish(average=False)
        ok = ok and torch.equal(out, torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert all(res.values()), res


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 32, 2048):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_ranks_even_split():
    _run(2, 32)


def test_two_ranks_ragged_split():
    _run(2, 7)


def test_eight_ranks_even_and_ragged_split():
    _run(8, 2048)  # BASELINE cfg4 / cfg5: 256 instances per rank
    _run(8, 13)    # fewer than two instances on some ranks
