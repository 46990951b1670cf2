"""Chamfer nearest-neighbour op, mirroring the reference wrapper tf_ops/nn_distance/tf_nndistance.py:15-40.

Same name, argument meaning, outputs and gradient wiring; tensors are torch CUDA tensors and the compute is
libmonopsr_hip.so (mpsr_nn_distance_fwd / mpsr_nn_distance_bwd).
"""
import torch

from monopsr_amd import _lib


def _check_clouds(op, xyz1, xyz2):
    # the reference's OP_REQUIRES checks (tf_nndistance.cpp:51-58)
    if xyz1.dim() != 3:
        raise _lib.InvalidArgumentError("%s requires xyz1 be of shape (batch,#points,3)" % op)
    if xyz1.shape[2] != 3:
        raise _lib.InvalidArgumentError("%s only accepts 3d point set xyz1" % op)
    if xyz2.dim() != 3:
        raise _lib.InvalidArgumentError("%s requires xyz2 be of shape (batch,#points,3)" % op)
    if xyz2.shape[2] != 3:
        raise _lib.InvalidArgumentError("%s only accepts 3d point set xyz2" % op)
    if xyz2.shape[0] != xyz1.shape[0]:
        raise _lib.InvalidArgumentError("%s expects xyz1 and xyz2 have same batch size" % op)
    if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
        raise _lib.InvalidArgumentError("%s expects float32 clouds" % op)


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """Op NnDistanceGrad (tf_nndistance.cpp:10-18): -> grad_xyz1 (b,n,3), grad_xyz2 (b,m,3)."""
    _check_clouds("NnDistanceGrad", xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    for name, t, shape in (("grad_dist1", grad_dist1, (b, n)), ("idx1", idx1, (b, n)),
                           ("grad_dist2", grad_dist2, (b, m)), ("idx2", idx2, (b, m))):
        if tuple(t.shape) != shape:
            raise _lib.InvalidArgumentError("NnDistanceGrad requires %s be of shape(batch,#points)" % name)
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    grad_dist1 = grad_dist1.contiguous().float()
    grad_dist2 = grad_dist2.contiguous().float()
    idx1, idx2 = idx1.contiguous().int(), idx2.contiguous().int()
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _lib.check(_lib.lib().mpsr_nn_distance_bwd(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(grad_dist1),
                                               _lib.ptr(idx1), _lib.ptr(grad_dist2), _lib.ptr(idx2),
                                               _lib.ptr(g1), _lib.ptr(g2), _lib.stream()))
    return g1, g2


class _NnDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _check_clouds("NnDistance", xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
        idx1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
        dist2 = torch.empty((b, m), dtype=torch.float32, device=xyz1.device)
        idx2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
        _lib.check(_lib.lib().mpsr_nn_distance_fwd(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1),
                                                   _lib.ptr(idx1), _lib.ptr(dist2), _lib.ptr(idx2), _lib.stream()))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, grad_dist1, grad_idx1, grad_dist2, grad_idx2):
        # the reference's _nn_distance_grad (tf_nndistance.py:34-40)
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        if grad_dist1 is None:
            grad_dist1 = torch.zeros(idx1.shape, dtype=torch.float32, device=xyz1.device)
        if grad_dist2 is None:
            grad_dist2 = torch.zeros(idx2.shape, dtype=torch.float32, device=xyz1.device)
        return nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2)


def nn_distance(xyz1, xyz2):
    """Computes the distance of nearest neighbors for a pair of point clouds.

    input: xyz1: (batch_size,#points_1,3)  the first point cloud
    input: xyz2: (batch_size,#points_2,3)  the second point cloud
    output: dist1: (batch_size,#point_1)   squared distance from first to second
    output: idx1:  (batch_size,#point_1)   nearest neighbor from first to second (int32)
    output: dist2: (batch_size,#point_2)   squared distance from second to first
    output: idx2:  (batch_size,#point_2)   nearest neighbor from second to first (int32)
    """
    return _NnDistance.apply(xyz1, xyz2)
