"""The METRIC path of BASELINE cfg2-cfg5 across ranks, with values (instances are independent, the batch is
split in contiguous shards, weights replicated, no data-path collective; the r05 review lists the multi-rank configurations
as untested beyond their flow): two ranks on cuda:0 (gloo: RCCL refuses two ranks per device) each run the instance path
-- both trunks' features in, decoder, heads, centroids -- plus the Chamfer search and the EMD loss on THEIR 8 of 16
instances.  Gathered back (data_parallel.gather_instances) the per-instance outputs equal the unsharded run's; the
rank-summed metrics (data_parallel.reduce_metric_sums) equal the unsharded sums.  Tolerances, not bits: a launch of 8
instances may tile and split its reductions differently from one of 16."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, NPTS, DIV = 16, 256, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    rng = np.random.default_rng(61)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    return dict(crops=dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                full_feat=dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // DIV)), 0).astype(np.float32)),
                boxes_2d=dev(boxes),
                cam_p=dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                view=dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                cls=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                mean_lwh=dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                z_off=torch.full((B,), 2.178, device="cuda"),
                gt=dev(rng.standard_normal((B, NPTS, 3)).astype(np.float32)))


def _path(net, i):
    """-> per-instance centroids (b, 3), per-instance Chamfer sums (b,), per-instance EMD costs (b,)."""
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance
    with torch.no_grad():
        xyz, out = net.forward_instances(i["crops"], i["full_feat"], i["boxes_2d"], i["cam_p"], i["view"], i["cls"],
                                         i["mean_lwh"], i["z_off"])
        pred = xyz.reshape(xyz.shape[0], -1, 3)[:, :NPTS].contiguous()
        d1, _, d2, _ = tf_nndistance.nn_distance(pred, i["gt"])
        emd = tf_approxmatch.emd_cost(pred, i["gt"])
    return out["centroids"].reshape(xyz.shape[0], -1), d1.sum(1) + d2.sum(1), emd.reshape(-1)


def _net():
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    return dn.DeviceNet(W.synthetic_weights(seed=62, width_div=DIV), device=torch.device("cuda", 0), width_div=DIV)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monopsr_amd.core import data_parallel as dp
        torch.cuda.set_device(0)
        inp = _inputs()
        mine = dp.shard_sample(inp, rank, world, per_instance_keys=[k for k in inp if k != "cam_p"])
        assert mine["crops"].shape[0] == B // world and mine["cam_p"].shape == (3, 4)
        cen, cham, emd = _path(_net(), mine)
        # (gloo gathers host tensors; on RCCL the same calls take the device tensors)
        all_cen = dp.gather_instances(cen.cpu(), B)
        all_cham = dp.gather_instances(cham.cpu(), B)
        sums = dp.reduce_metric_sums([cham.sum().cpu(), emd.sum().cpu(), torch.tensor(float(cen.shape[0]))])
        q.put((rank, all_cen.numpy(), all_cham.numpy(), dict(zip(("chamfer", "emd", "instances"), sums.tolist()))))
    finally:
        dist.destroy_process_group()


def test_two_rank_metric_path_equals_the_unsharded_run():
    cen, cham, emd = _path(_net(), _inputs())
    cen, cham, emd = cen.cpu().numpy(), cham.cpu().numpy(), emd.cpu().numpy()
    assert np.isfinite(cen).all() and np.isfinite(cham).all() and np.isfinite(emd).all() and emd.min() > 0
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=600)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        all_cen, all_cham, sums = res[r]
        assert all_cen.shape == cen.shape and all_cham.shape == cham.shape
        assert np.abs(all_cen - cen).max() <= 1e-4 * max(1.0, np.abs(cen).max()), r   # every instance, in batch order
        assert np.abs(all_cham - cham).max() <= 1e-4 * np.abs(cham).max(), r
        assert sums["instances"] == B
        assert abs(sums["chamfer"] - float(cham.sum())) <= 1e-4 * float(cham.sum()), (r, sums)
        assert abs(sums["emd"] - float(emd.sum())) <= 2e-3 * float(emd.sum()), (r, sums)   # (the EMD's own 1e-3 parity bound)
    assert res[0][2] == res[1][2]  # both ranks hold the same reduced numbers
