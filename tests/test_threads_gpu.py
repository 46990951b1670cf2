"""Concurrent callers: four host threads, each with its own HIP stream and scratch, drive the Chamfer ops, a
convolution with the library-scheduled (split_k = 0) path and the EMD loss at the same time; every thread must get
what a single-threaded run gives (include/monopsr_hip.h, "Threads")."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_four_threads_four_streams():
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd
    rng = np.random.default_rng(0)
    jobs = []
    for t in range(4):
        x1 = torch.from_numpy(rng.uniform(-1, 1, (6, 700 + 13 * t, 3)).astype(np.float32)).cuda()
        x2 = torch.from_numpy(rng.uniform(-1, 1, (6, 500 + 7 * t, 3)).astype(np.float32)).cuda()
        a = torch.from_numpy(rng.standard_normal((4, 12, 12, 64 + 32 * t)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((96, 9 * (64 + 32 * t))) / 30).astype(np.float32)).cuda()
        jobs.append((x1, x2, a, w))

    def work(job):
        x1, x2, a, w = job
        d1, i1, d2, i2 = nnd.nn_distance(x1, x2)
        g1, g2 = nnd.nn_distance_grad(x1, x2, torch.ones_like(d1), i1, torch.ones_like(d2), i2)
        y = dn.conv2d(a, w, None, None, 3, 3, 2, True, split_k=0)
        cost, e1, e2 = am.emd_loss_fwd_bwd(x1, x2)
        return [d1, i1, d2, i2, g1, g2, y, cost, e1, e2]
    want = [work(j) for j in jobs]
    torch.cuda.synchronize()
    got, errors = [None] * 4, []

    def runner(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(5):
                    got[k] = work(jobs[k])
            s.synchronize()
        except Exception as e:  # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=runner, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        for i, (a, b) in enumerate(zip(got[k], want[k])):
            if i in (4, 5, 7):  # LDS / global fp32 atomics: summation order varies run to run
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
            else:
                assert torch.equal(a, b), (k, i)


def test_two_threads_two_policies_do_not_see_each_other():
    """Options per CALL, not per process (ABI 5: mpsr_conv_opts / mpsr_net_opts.math, .winograd_policy; the reference's
    launchers are stateless, tf_nndistance.cpp:168): two host threads on two streams run the same layers concurrently,
    one with the Winograd policy "auto", the other "off" (and one instance path each with fp32 / bf16x3 arithmetic) --
    every call returns the bits of its own single-threaded run, whatever the other thread is doing, and the
    process-wide defaults are untouched."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(3)
    x3 = torch.from_numpy(rng.standard_normal((96, 12, 12, 128)).astype(np.float32)).cuda()   # F(3x3,3x3) when "auto"
    w3 = torch.from_numpy((rng.standard_normal((128, 9 * 128)) / 34).astype(np.float32)).cuda()
    x4 = torch.from_numpy(rng.standard_normal((32, 48, 48, 64)).astype(np.float32)).cuda()    # F(4x4,3x3) when "auto"
    w4 = torch.from_numpy((rng.standard_normal((64, 9 * 64)) / 24).astype(np.float32)).cuda()
    weights = W.synthetic_weights(seed=9, width_div=4)
    nets = {m: dn.DeviceNet(weights, width_div=4) for m in ("fp32", "bf16x3")}
    for m, net in nets.items():
        net.math = m
    crops = torch.from_numpy((rng.standard_normal((6, 48, 48, 3)) * 50).astype(np.float32)).cuda()

    def work(policy, math):
        a = dn.conv2d(x3, w3, None, None, 3, 3, 4, True, split_k=0, winograd_policy=policy)
        b = dn.conv2d(x4, w4, None, None, 3, 3, 1, True, split_k=0, winograd_policy=policy)
        c = nets[math].trunk(crops)
        return [a, b, c]
    cfgs = [("auto", "fp32"), ("off", "bf16x3")]
    want = [work(*c) for c in cfgs]
    torch.cuda.synchronize()
    # the two settings really are different evaluations
    assert not torch.equal(want[0][0], want[1][0]) and not torch.equal(want[0][1], want[1][1])
    assert not torch.equal(want[0][2], want[1][2])
    got, errors = [None, None], []
    barrier = threading.Barrier(2)

    def runner(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                barrier.wait()
                for _ in range(20):
                    got[k] = work(*cfgs[k])
            s.synchronize()
        except Exception as e:  # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=runner, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        for i, (a, b) in enumerate(zip(got[k], want[k])):
            assert torch.equal(a, b), (cfgs[k], i)
    assert _lib.get_conv_math() == "fp32" and _lib.lib().mpsr_get_winograd_policy() == 0
    # an unknown option value is refused, not silently inherited
    opts = _lib.ConvOpts(7, 0)
    import ctypes
    y = torch.empty((32, 48, 48, 64), device="cuda")
    rc = _lib.lib().mpsr_conv2d_nhwc_f32_ex(x4.data_ptr(), 32, 48, 48, 64, w4.data_ptr(), None, None, y.data_ptr(), 64, 3,
                                            3, 1, 0, 1, None, 0, ctypes.byref(opts), _lib.stream())
    assert rc == 1 and b"opts.math" in _lib.lib().mpsr_last_error()
