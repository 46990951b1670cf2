"""Approximate Earth Mover's Distance ops, mirroring the reference wrapper
tf_ops/approxmatch/tf_approxmatch.py:15-71 (approx_match has no gradient; match_cost differentiates w.r.t. both
clouds with match held constant).  Compute: libmonopsr_hip.so (mpsr_approx_match / mpsr_match_cost /
mpsr_match_cost_grad / mpsr_emd_loss).

semantics="device" (default): the reference's GPU kernel (10 levels, fp32, match laid out (b, m, n)).
semantics="host": the reference's CPU kernel (tf_approxmatch.cpp:23-140: 11 levels, double state, match laid out
(b, n, m)) -- what a TF1-CPU run of the reference computes (BASELINE config 1).
"""
import torch

from monopsr_amd import _lib

SEMANTICS = {"device": 0, "host": 1}


def _sem(semantics):
    if semantics not in SEMANTICS:
        raise _lib.InvalidArgumentError("unknown EMD semantics %r (choose from %s)" % (semantics, sorted(SEMANTICS)))
    return SEMANTICS[semantics]


def _temp(b, n, m, sem, device, loss=False):
    # float64 storage: 8-byte aligned for the host semantics' double state
    # (loss: the fused loss' scratch -- with the sorted clouds behind the state it culls the steep levels' far pairs)
    lib = _lib.lib()
    nfl = lib.mpsr_emd_loss_temp_floats(b, n, m, sem) if loss else lib.mpsr_emd_temp_floats(b, n, m, sem)
    return torch.empty(((nfl + 1) // 2,), dtype=torch.float64, device=device), nfl


def _check(op, xyz1, xyz2, match=None, host=False):
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
        if match.dim() != 3 or tuple(match.shape) != ((b, n, m) if host else (b, m, n)):
            raise _lib.InvalidArgumentError("%s expects (batch_size,#query,#dataset) match shape" % op)


def approx_match(xyz1, xyz2, semantics="device", temp_floats=None):
    """
    input:
        xyz1 : batch_size * #dataset_points * 3
        xyz2 : batch_size * #query_points * 3
    returns:
        match : batch_size * #query_points * #dataset_points   (semantics="host": batch * #dataset * #query)
    temp_floats: size of the scratch handed to the library (default: what the fast path asks for; b*(n+m)*2, the
    reference op shell's allocation, selects the compact path).
    """
    _check("ApproxMatch", xyz1, xyz2)
    sem = _sem(semantics)
    with torch.no_grad():  # ops.NoGradient('ApproxMatch'), tf_approxmatch.py:26
        xyz1, xyz2 = xyz1.detach().contiguous(), xyz2.detach().contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = torch.empty((b, n, m) if sem else (b, m, n), dtype=torch.float32, device=xyz1.device)
        temp, nfl = _temp(b, n, m, sem, xyz1.device)
        if temp_floats is not None:
            nfl = int(temp_floats)
            temp = torch.empty(((nfl + 1) // 2 + 1,), dtype=torch.float64, device=xyz1.device)
        _lib.check(_lib.lib().mpsr_approx_match_ex(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                                   _lib.ptr(temp), nfl, sem, _lib.stream()))
    return match


def match_cost_grad(xyz1, xyz2, match, semantics="device"):
    """Op MatchCostGrad: -> grad1 (b,n,3), grad2 (b,m,3)."""
    host = bool(_sem(semantics))
    _check("MatchCostGrad", xyz1, xyz2, match, host)
    xyz1, xyz2, match = xyz1.contiguous(), xyz2.contiguous(), match.contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    if host:  # a (b,n,m) match is the (b,m',n') layout of the swapped clouds
        _lib.check(_lib.lib().mpsr_match_cost_grad(b, m, n, _lib.ptr(xyz2), _lib.ptr(xyz1), _lib.ptr(match),
                                                   _lib.ptr(g2), _lib.ptr(g1), _lib.stream()))
    else:
        _lib.check(_lib.lib().mpsr_match_cost_grad(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                                   _lib.ptr(g1), _lib.ptr(g2), _lib.stream()))
    return g1, g2


class _MatchCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, match, semantics):
        host = bool(_sem(semantics))
        _check("MatchCost", xyz1, xyz2, match, host)
        xyz1, xyz2, match = xyz1.contiguous(), xyz2.contiguous(), match.contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
        if host:
            _lib.check(_lib.lib().mpsr_match_cost(b, m, n, _lib.ptr(xyz2), _lib.ptr(xyz1), _lib.ptr(match),
                                                  _lib.ptr(cost), _lib.stream()))
        else:
            _lib.check(_lib.lib().mpsr_match_cost(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(match),
                                                  _lib.ptr(cost), _lib.stream()))
        ctx.save_for_backward(xyz1, xyz2, match)
        ctx.semantics = semantics
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        # _match_cost_grad (tf_approxmatch.py:52-71): scale by grad_cost[:, None, None]; no gradient for match
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = match_cost_grad(xyz1, xyz2, match, ctx.semantics)
        scale = grad_cost.reshape(-1, 1, 1)
        return g1 * scale, g2 * scale, None, None


def match_cost(xyz1, xyz2, match, semantics="device"):
    """
    input:
        xyz1 : batch_size * #dataset_points * 3
        xyz2 : batch_size * #query_points * 3
        match : batch_size * #query_points * #dataset_points   (semantics="host": batch * #dataset * #query)
    returns:
        cost : batch_size
    """
    return _MatchCost.apply(xyz1, xyz2, match, semantics)


def emd_loss_fwd_bwd(xyz1, xyz2, semantics="device", want_grads=True):
    """cost (b), d cost / d xyz1 (b,n,3), d cost / d xyz2 (b,m,3) of match_cost(xyz1, xyz2, approx_match(xyz1, xyz2))
    with match held constant -- in one native call that never materialises the (b,m,n) match tensor.
    want_grads=False: cost only (gradients returned as None)."""
    _check("EmdLoss", xyz1, xyz2)
    sem = _sem(semantics)
    with torch.no_grad():
        xyz1, xyz2 = xyz1.detach().contiguous(), xyz2.detach().contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = torch.empty((b,), dtype=torch.float32, device=xyz1.device)
        g1, g2 = (torch.empty_like(xyz1), torch.empty_like(xyz2)) if want_grads else (None, None)
        temp, nfl = _temp(b, n, m, sem, xyz1.device, loss=True)
        _lib.check(_lib.lib().mpsr_emd_loss(b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(cost), _lib.ptr(g1),
                                            _lib.ptr(g2), _lib.ptr(temp), nfl, sem, _lib.stream()))
    return cost, g1, g2


class _EmdCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, semantics):
        cost, g1, g2 = emd_loss_fwd_bwd(xyz1, xyz2, semantics)
        ctx.save_for_backward(g1, g2)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        g1, g2 = ctx.saved_tensors
        scale = grad_cost.reshape(-1, 1, 1)
        return g1 * scale, g2 * scale, None


def emd_cost(xyz1, xyz2, semantics="device"):
    """Differentiable fused form of match_cost(xyz1, xyz2, approx_match(xyz1, xyz2)): same value and gradients
    (match held constant, tf_approxmatch.py:26,52-71), no match tensor in memory."""
    return _EmdCost.apply(xyz1, xyz2, semantics)
