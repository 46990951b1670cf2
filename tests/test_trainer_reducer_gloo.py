"""ReverseBucketReducer (core/trainer.py): bucket launch order and all-reduce result, world_size 2 over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, mode="rccl"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monopsr_amd.core.trainer import ReverseBucketReducer
        flat = torch.zeros(1000)
        spans = [(0, 300), (300, 450), (450, 700), (700, 1000)]  # four "layers"
        red = ReverseBucketReducer(flat, spans, bucket_bytes=250 * 4, mode=mode)
        assert len(red.buckets) == 4
        order = []
        orig = red._launch
        red._launch = lambda bi: (order.append(bi), orig(bi))[1]
        # backward visits layers last -> first; each writes its slice then reports
        for li in (3, 2, 1, 0):
            lo, hi = spans[li]
            flat[lo:hi] = float(rank + 1) * (li + 1)
            red.layer_ready(li)
        red.finish(average=True)
        want = torch.zeros(1000)
        for li, (lo, hi) in enumerate(spans):
            want[lo:hi] = (li + 1) * sum(range(1, world + 1)) / world
        # bucket 3 = [750,1000) needs only layer 3; bucket 2 = [500,750) layers 2+3; buckets 1 and 0 both wait for
        # layer 0 (span [0,300)) and go out together when it reports
        ok = torch.allclose(flat, want) and order[:2] == [3, 2] and sorted(order[2:]) == [0, 1]
        # second step reuses the reducer
        flat.fill_(float(rank))
        for li in (3, 2, 1, 0):
            red.layer_ready(li)
        red.finish(average=False)
        ok = ok and torch.allclose(flat, torch.full((1000,), float(sum(range(world)))))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_reverse_bucket_reducer_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert all(res.values()), res


def test_reverse_bucket_reducer_direct_mode_two_and_three_ranks():
    """mode="direct": reduce_scatter_tensor + all_gather_into_tensor per bucket (SURVEY 5's mesh-aware exchange) gives the
    all-reduce's result with the same launch order; at three ranks the 250-element buckets do not divide into shards
    and fall back to all_reduce bucket by bucket -- same result."""
    ctx = mp.get_context("spawn")
    for world in (2, 3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, "direct")) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        res = dict(q.get(timeout=5) for _ in range(world))
        assert all(res.values()), (world, res)


def test_reverse_bucket_reducer_eight_ranks_both_modes():
    """The node the scaling run uses has eight GPUs: the same exchange at world_size 8 ("direct": 250-element buckets do
    not divide by eight and fall back to all_reduce, as a last odd bucket of the real 64 MiB ones may)."""
    ctx = mp.get_context("spawn")
    for mode in ("rccl", "direct"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 8, port, q, mode)) for r in range(8)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        res = dict(q.get(timeout=5) for _ in range(8))
        assert all(res.values()), (mode, res)


def test_reducer_rejects_an_unknown_mode():
    import pytest
    from monopsr_amd.core.trainer import ReverseBucketReducer
    with pytest.raises(ValueError):
        ReverseBucketReducer(torch.zeros(8), [(0, 8)], mode="ring")


def _bias_worker(rank, world, port, q):
    """Flat layout [w0 | b0 | w1 | b1] with a bucket boundary exactly between w0 and b0 (the bias slice then
    sits in a bucket whose members must still include layer 0)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monopsr_amd.core.trainer import ReverseBucketReducer
        flat = torch.zeros(400)
        spans = [[(0, 100), (100, 110)], [(110, 390), (390, 400)]]  # per layer: weight span, bias span
        red = ReverseBucketReducer(flat, spans, bucket_bytes=100 * 4)
        assert red.buckets == [(0, 100), (100, 200), (200, 300), (300, 400)]
        assert red.members[1] == {0, 1}, red.members  # bucket [100,200) holds b0: it waits for layer 0 too
        order = []
        orig = red._launch
        red._launch = lambda bi: (order.append(bi), orig(bi))[1]
        # layer 1's backward completes first
        flat[110:400] = float(rank + 1)
        red.layer_ready(1)
        launched_early = list(order)
        # only now does layer 0 write its bias and weights
        flat[0:110] = 10.0 * (rank + 1)
        red.layer_ready(0)
        red.finish(average=False)
        tot = float(sum(range(1, world + 1)))
        want = torch.cat([torch.full((110,), 10 * tot), torch.full((290,), tot)])
        q.put((rank, bool(torch.equal(flat, want) and 1 not in launched_early and 0 not in launched_early
                          and sorted(launched_early) == [2, 3])))
    finally:
        dist.destroy_process_group()


def test_bucket_boundary_on_a_bias_slice():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bias_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert all(res.values()), res


def test_reducer_single_process_is_a_noop():
    from monopsr_amd.core.trainer import ReverseBucketReducer
    flat = torch.arange(10, dtype=torch.float32)
    red = ReverseBucketReducer(flat, [(0, 5), (5, 10)], bucket_bytes=16)
    red.layer_ready(1)
    red.layer_ready(0)
    red.finish()
    assert torch.equal(flat, torch.arange(10, dtype=torch.float32))


def _one_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from monopsr_amd.core import data_parallel
        from monopsr_amd.core.trainer import ReverseBucketReducer
        spans = [(0, 300), (300, 450), (450, 700), (700, 1000)]
        res = {}
        for mode in ("rccl", "direct"):
            flat = torch.arange(1000, dtype=torch.float32)
            idle = ReverseBucketReducer(flat, spans, bucket_bytes=250 * 4, mode=mode)
            forced = ReverseBucketReducer(flat, spans, bucket_bytes=250 * 4, mode=mode, force_active=True)
            for red in (idle, forced):
                for li in (3, 2, 1, 0):
                    red.layer_ready(li)
                red.finish(average=True)
            res[mode] = (idle._active(), forced._active(), list(idle.last_issued), sorted(set(forced.last_issued)),
                         len(forced.last_issued), bool(torch.equal(flat, torch.arange(1000, dtype=torch.float32))))
        x = torch.arange(12.0).reshape(4, 3)
        res["gather"] = bool(torch.equal(data_parallel.gather_instances(x, 4, force_collective=True), x))
        res["sums"] = data_parallel.reduce_metric_sums([torch.tensor(2.0), torch.tensor(3.0)],
                                                       force_collective=True).tolist()
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_forced_collectives_on_a_one_rank_group_are_identities():
    """force_active (the switch tests/test_rccl_one_rank_gpu.py uses to run the real RCCL calls on a single-GPU box): on a
    one-rank group an unforced reducer issues nothing, a forced one issues every bucket's collectives and leaves the
    buffer as it was; same for the metric path's helpers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    res = q.get(timeout=5)
    assert res["rccl"] == (False, True, [], ["all_reduce"], 4, True), res["rccl"]
    assert res["direct"] == (False, True, [], ["all_gather_into_tensor", "reduce_scatter_tensor"], 8, True), res["direct"]
    assert res["gather"] and res["sums"] == [2.0, 3.0]
