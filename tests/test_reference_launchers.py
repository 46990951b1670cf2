"""The reference-side binding of INTEGRATION.md section 1 is real code: examples/reference_op_launchers.cpp defines
the five launcher functions the reference's op shells declare (tf_nndistance.cpp:168,208; tf_approxmatch.cpp:141-143)
on top of include/monopsr_hip.h.  CPU: it compiles with a plain C++11 compiler against the header and links against
libmonopsr_hip.so.  GPU: called with device buffers, it gives what the library's own wrappers give."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "reference_op_launchers.cpp")
# Itanium-mangled names of the reference's declarations
SYMS = {
    "NmDistanceKernelLauncher": "_Z24NmDistanceKernelLauncheriiPKfiS0_PfPiS1_S2_",
    "NmDistanceGradKernelLauncher": "_Z28NmDistanceGradKernelLauncheriiPKfiS0_S0_PKiS0_S2_PfS3_",
    "approxmatchLauncher": "_Z19approxmatchLauncheriiiPKfS0_PfS1_",
    "matchcostLauncher": "_Z17matchcostLauncheriiiPKfS0_S0_Pf",
    "matchcostgradLauncher": "_Z21matchcostgradLauncheriiiPKfS0_S0_PfS1_",
}


def _build(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "monopsr_amd", "csrc")], stdout=subprocess.DEVNULL)
    obj = str(tmp_path / "launchers.o")
    so = str(tmp_path / "libmonopsr_tf_launchers.so")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fPIC", "-I", os.path.join(ROOT, "include"), "-c",
                           SRC, "-o", obj])
    libdir = os.path.join(ROOT, "monopsr_amd")
    subprocess.check_call(["g++", "-shared", "-o", so, obj, "-L" + libdir, "-lmonopsr_hip", "-Wl,-rpath," + libdir,
                           "-Wl,--no-undefined"])
    return so


def test_launchers_compile_and_link_against_the_header(tmp_path):
    so = _build(tmp_path)
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    for name, mangled in SYMS.items():
        assert " T " + mangled in out, (name, out)
    # the mangled names really are the reference's signatures
    dem = subprocess.check_output(["c++filt"] + list(SYMS.values())).decode().splitlines()
    assert dem[0] == "NmDistanceKernelLauncher(int, int, float const*, int, float const*, float*, int*, float*, int*)"
    assert dem[2] == "approxmatchLauncher(int, int, int, float const*, float const*, float*, float*)"


@pytest.mark.gpu
def test_launchers_on_device_buffers(tmp_path):
    import torch
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd
    lib = ctypes.CDLL(_build(tmp_path))
    P, I = ctypes.c_void_p, ctypes.c_int
    rng = np.random.default_rng(0)
    b, n, m = 3, 200, 150
    x1 = torch.from_numpy(rng.uniform(-1, 1, (b, n, 3)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.uniform(-1, 1, (b, m, 3)).astype(np.float32)).cuda()
    d1, d2 = torch.empty((b, n), device="cuda"), torch.empty((b, m), device="cuda")
    i1 = torch.empty((b, n), dtype=torch.int32, device="cuda")
    i2 = torch.empty((b, m), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    f = getattr(lib, SYMS["NmDistanceKernelLauncher"])
    f.argtypes, f.restype = [I, I, P, I, P, P, P, P, P], None
    f(b, n, x1.data_ptr(), m, x2.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr())
    torch.cuda.synchronize()
    r = nnd.nn_distance(x1, x2)
    assert torch.equal(d1, r[0]) and torch.equal(i1, r[1]) and torch.equal(d2, r[2]) and torch.equal(i2, r[3])
    g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
    ones1, ones2 = torch.ones_like(d1), torch.ones_like(d2)
    f = getattr(lib, SYMS["NmDistanceGradKernelLauncher"])
    f.argtypes, f.restype = [I, I, P, I, P, P, P, P, P, P, P], None
    f(b, n, x1.data_ptr(), m, x2.data_ptr(), ones1.data_ptr(), i1.data_ptr(), ones2.data_ptr(), i2.data_ptr(),
      g1.data_ptr(), g2.data_ptr())
    torch.cuda.synchronize()
    h1, h2 = nnd.nn_distance_grad(x1, x2, ones1, i1, ones2, i2)
    torch.testing.assert_close(g1, h1, rtol=0, atol=1e-5 * float(h1.abs().max()))
    torch.testing.assert_close(g2, h2, rtol=0, atol=1e-5 * float(h2.abs().max()))
    # EMD with the reference shell's scratch: TensorShape{b, (n+m)*2}
    match = torch.empty((b, m, n), device="cuda")
    temp = torch.empty((b, (n + m) * 2), device="cuda")
    f = getattr(lib, SYMS["approxmatchLauncher"])
    f.argtypes, f.restype = [I, I, I, P, P, P, P], None
    f(b, n, m, x1.data_ptr(), x2.data_ptr(), match.data_ptr(), temp.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(match, am.approx_match(x1, x2))
    cost = torch.empty((b,), device="cuda")
    f = getattr(lib, SYMS["matchcostLauncher"])
    f.argtypes, f.restype = [I, I, I, P, P, P, P], None
    f(b, n, m, x1.data_ptr(), x2.data_ptr(), match.data_ptr(), cost.data_ptr())
    f = getattr(lib, SYMS["matchcostgradLauncher"])
    f.argtypes, f.restype = [I, I, I, P, P, P, P, P], None
    f(b, n, m, x1.data_ptr(), x2.data_ptr(), match.data_ptr(), g1.data_ptr(), g2.data_ptr())
    torch.cuda.synchronize()
    torch.testing.assert_close(cost, am.match_cost(x1, x2, match), rtol=1e-5, atol=0)
    e1, e2 = am.match_cost_grad(x1, x2, match)
    assert torch.equal(g1, e1) and torch.equal(g2, e2)
