"""Approximate Earth Mover's Distance ops, mirroring the reference wrapper
tf_ops/approxmatch/tf_approxmatch.py:15-71 (approx_match has no gradient; match_cost differentiates w.r.t. both
clouds with match held constant).  Compute: libmonopsr_hip.so (mpsr_approx_match / mpsr_match_cost /
mpsr_match_cost_grad), GPU-kernel semantics of the reference (10 levels, fp32, match laid out (b, m, n)).
"""
import torch

from monopsr_amd import _lib


def _check(op, xyz1, xyz2, match=None):
    # tf_approxmatch.cpp:152-165, 183-191
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise _lib.InvalidArgumentError("%s expects (batch_size,num_points,3) xyz1 shape" % op)
    if xyz2.dim() != 3 or xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise _lib.InvalidArgumentError("%s expects (batch_size,num_points,3) xyz2 shape, and batch_size must "
                                        "match" % op)
    if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
        raise _lib.InvalidArgumentError("%s expects float32 clouds" % op)
    if match is not None:
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        if match.dim() != 3 or tuple(match.shape) != (b, m, n):
            raise _lib.InvalidArgumentError("%s expects (batch_size,#query,#dataset) match shape" % op)


def approx_match(xyz1, xyz2):
    """
    input:
        xyz1 : batch_size * #dataset_points * 3
        xyz2 : batch_size * #query_points * 3
    returns:
        match : batch_size * #query_points * #dataset_points
    """
    _check("ApproxMatch", xyz1, xyz2)
    with torch.no_grad():  # ops.NoGradient('ApproxMatch'), tf_approxmatch.py:26
        xyz1, xyz2 = xyz1.detach().contiguous(), xyz2.detach().contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = torch.empty((b, m, n), dtype=torch.float32, device=xyz1.device)
        temp = torch.empty((_lib.lib().mpsr_approx_match_temp_floats(b, n, m),), dtype=torch.float32,
                           device=xyz1.device)
        _lib.check(_lib.lib().mpsr_approx_match(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                                _lib.ptr(temp), _lib.stream()))
    return match


def match_cost_grad(xyz1, xyz2, match):
    """Op MatchCostGrad: -> grad1 (b,n,3), grad2 (b,m,3)."""
    _check("MatchCostGrad", xyz1, xyz2, match)
    xyz1, xyz2, match = xyz1.contiguous(), xyz2.contiguous(), match.contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _lib.check(_lib.lib().mpsr_match_cost_grad(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                               _lib.ptr(g1), _lib.ptr(g2), _lib.stream()))
    return g1, g2


class _MatchCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, match):
        _check("MatchCost", xyz1, xyz2, match)
        xyz1, xyz2, match = xyz1.contiguous(), xyz2.contiguous(), match.contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
        _lib.check(_lib.lib().mpsr_match_cost(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                              _lib.ptr(cost), _lib.stream()))
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        # _match_cost_grad (tf_approxmatch.py:52-71): scale by grad_cost[:, None, None]; no gradient for match
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = match_cost_grad(xyz1, xyz2, match)
        scale = grad_cost.reshape(-1, 1, 1)
        return g1 * scale, g2 * scale, None


def match_cost(xyz1, xyz2, match):
    """
    input:
        xyz1 : batch_size * #dataset_points * 3
        xyz2 : batch_size * #query_points * 3
        match : batch_size * #query_points * #dataset_points
    returns:
        cost : batch_size
    """
    return _MatchCost.apply(xyz1, xyz2, match)
