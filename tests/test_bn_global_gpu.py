"""Cross-rank statistics of the map decoder's training-mode BatchNorm (TrainNet(decoder_bn='batch_global')).

The reference normalises the decoder's activations over ALL instances of a step (one process, 32 boxes:
builders/net_builder.py:78-79,86-87 with is_training=True).  A data-parallel step shards the instances, so per-rank
statistics give different activations and gradients than the reference's on the same inputs; 'batch_global' pools the
fp64 sums of mpsr_batch_norm_stats / mpsr_batch_norm_grad_sums over the ranks before normalising.

Test (one GPU): two ranks over gloo, both computing on cuda:0 (RCCL refuses two ranks on one device; the collective is
2 C + 1 fp64 numbers per layer), 2 x 16 instances sharded against 32 instances in one process: decoder outputs, the
input gradient, every parameter gradient (summed over the ranks) and the moving statistics must agree to 1e-5.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, DIV = 32, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    rng = np.random.default_rng(5)
    crop_feat = np.maximum(rng.standard_normal((B, 12, 12, 1024 // DIV)), 0).astype(np.float32)
    # (instances of very different scale: per-shard statistics would differ visibly from the pooled ones)
    crop_feat *= np.linspace(0.3, 3.0, B, dtype=np.float32)[:, None, None, None]
    full_feat = np.maximum(rng.standard_normal((B, 12, 12, 1024 // DIV)), 0).astype(np.float32)
    probe = rng.standard_normal((B, 48, 48, 128 // DIV)).astype(np.float32)  # the loss is <features_for_map, probe> / B
    return crop_feat, full_feat, probe


def _run(net, crop_feat, full_feat, probe, scale):
    net.zero_grad()
    x = crop_feat.clone().requires_grad_(True)
    fb, fm, xyz = net.squash_decoder(x, full_feat)
    loss = (fm * probe).sum() * scale + (xyz * xyz).sum() * scale
    loss.backward()
    return fm.detach(), xyz.detach(), x.grad.detach()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monopsr_amd.core import train_net
        from monopsr_amd.core import weights as W
        torch.cuda.set_device(0)
        net = train_net.TrainNet(W.synthetic_weights(seed=121, width_div=DIV), width_div=DIV, decoder_bn='batch_global')
        crop_feat, full_feat, probe = [torch.from_numpy(a).cuda() for a in _inputs()]
        n = B // world
        sl = slice(rank * n, (rank + 1) * n)
        fm, xyz, gx = _run(net, crop_feat[sl], full_feat[sl], probe[sl], 1.0 / B)
        grads = net.grads.cpu()
        dist.all_reduce(grads)  # what the trainer's reducer does (before its division by the world size)
        bn = [L.batch_norm for L in net.layers if L.batch_norm is not None]
        q.put((rank, fm.cpu().numpy(), xyz.cpu().numpy(), gx.cpu().numpy(), grads.numpy(),
               [b.moving_mean.cpu().numpy() for b in bn], [b.moving_variance.cpu().numpy() for b in bn]))
    finally:
        dist.destroy_process_group()


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("world", [2])
def test_sharded_batch_with_pooled_statistics_equals_the_whole_batch(world):
    from monopsr_amd.core import train_net
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=121, width_div=DIV)
    crop_feat, full_feat, probe = [torch.from_numpy(a).cuda() for a in _inputs()]
    whole = train_net.TrainNet(weights, width_div=DIV, decoder_bn='batch')
    fm, xyz, gx = _run(whole, crop_feat, full_feat, probe, 1.0 / B)
    want_grads = whole.grads.cpu().numpy()
    bn = [L.batch_norm for L in whole.layers if L.batch_norm is not None]
    assert len(bn) == 4

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
    n = B // world
    for r in range(world):
        sfm, sxyz, sgx, sgrads, smm, smv = res[r]
        sl = slice(r * n, (r + 1) * n)
        assert _rel(sfm, fm[sl].cpu().numpy()) < 1e-5, "features_for_map of rank %d" % r
        assert _rel(sxyz, xyz[sl].cpu().numpy()) < 1e-5
        assert _rel(sgx, gx[sl].cpu().numpy()) < 1e-5, "input gradient of rank %d" % r
        assert _rel(sgrads, want_grads) < 1e-5, "parameter gradients (summed over ranks)"
        for k in range(4):
            assert _rel(smm[k], bn[k].moving_mean.cpu().numpy()) < 1e-5
            assert _rel(smv[k], bn[k].moving_variance.cpu().numpy()) < 1e-5

    # and per-rank statistics really are something else on these inputs (the test has teeth)
    shard = train_net.TrainNet(weights, width_div=DIV, decoder_bn='batch')
    fm0, _, _ = _run(shard, crop_feat[:n], full_feat[:n], probe[:n], 1.0 / B)
    assert _rel(fm0.cpu().numpy(), fm[:n].cpu().numpy()) > 1e-3
