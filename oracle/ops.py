"""ctypes front-end of the CPU oracle for the point-cloud ops (oracle_ops.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under monopsr_amd/ may import this module.

numpy in, numpy out; shapes and names follow the reference's op wrappers
(/root/reference/src/tf_ops/nn_distance/tf_nndistance.py:15-25,
 /root/reference/src/tf_ops/approxmatch/tf_approxmatch.py:15-43).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ops.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile liboracle_ops.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_ops.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return a.ctypes.data_as(_F)


def _i(a):
    return a.ctypes.data_as(_I)


def _clouds(xyz1, xyz2):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    if xyz1.ndim != 3 or xyz1.shape[2] != 3 or xyz2.ndim != 3 or xyz2.shape[2] != 3:
        raise ValueError("clouds must be (batch, #points, 3)")
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("clouds must have the same batch size")
    return xyz1, xyz2


def nn_distance(xyz1, xyz2):
    """-> dist1 (b,n) f32, idx1 (b,n) i32, dist2 (b,m) f32, idx2 (b,m) i32."""
    xyz1, xyz2 = _clouds(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32)
    i2 = np.empty((b, m), np.int32)
    lib().orc_nn_distance(b, n, m, _f(xyz1), _f(xyz2), _f(d1), _i(i1), _f(d2), _i(i2))
    return d1, i1, d2, i2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """-> grad_xyz1 (b,n,3), grad_xyz2 (b,m,3)."""
    xyz1, xyz2 = _clouds(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gd1 = np.ascontiguousarray(grad_dist1, np.float32).reshape(b, n)
    gd2 = np.ascontiguousarray(grad_dist2, np.float32).reshape(b, m)
    i1 = np.ascontiguousarray(idx1, np.int32).reshape(b, n)
    i2 = np.ascontiguousarray(idx2, np.int32).reshape(b, m)
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().orc_nn_distance_grad(b, n, m, _f(xyz1), _f(xyz2), _f(gd1), _i(i1), _f(gd2), _i(i2), _f(g1), _f(g2))
    return g1, g2


def approx_match(xyz1, xyz2, semantics="gpu"):
    """semantics='gpu' -> match (b,m,n), 10 levels, fp32 (tf_approxmatch_g.cu); 'cpu' -> match stored (b,n,m),
    11 levels, double state (tf_approxmatch.cpp)."""
    xyz1, xyz2 = _clouds(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if semantics == "gpu":
        match = np.empty((b, m, n), np.float32)
        lib().orc_approxmatch_gpusem(b, n, m, _f(xyz1), _f(xyz2), _f(match))
    elif semantics == "cpu":
        match = np.empty((b, n, m), np.float32)
        lib().orc_approxmatch_cpu(b, n, m, _f(xyz1), _f(xyz2), _f(match))
    else:
        raise ValueError(semantics)
    return match


def match_cost(xyz1, xyz2, match, semantics="gpu"):
    xyz1, xyz2 = _clouds(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.ascontiguousarray(match, np.float32)
    assert match.size == b * n * m
    cost = np.empty((b,), np.float32)
    fn = lib().orc_matchcost_gpusem if semantics == "gpu" else lib().orc_matchcost_cpu
    fn(b, n, m, _f(xyz1), _f(xyz2), _f(match), _f(cost))
    return cost


def match_cost_grad(xyz1, xyz2, match, semantics="gpu"):
    xyz1, xyz2 = _clouds(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.ascontiguousarray(match, np.float32)
    assert match.size == b * n * m
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    fn = lib().orc_matchcostgrad_gpusem if semantics == "gpu" else lib().orc_matchcostgrad_cpu
    fn(b, n, m, _f(xyz1), _f(xyz2), _f(match), _f(g1), _f(g2))
    return g1, g2
