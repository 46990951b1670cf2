import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from monopsr_amd import _lib
from monopsr_amd.core import device_net as dn
lib = _lib.lib()
rng = np.random.default_rng(0)
for (B, H, W, C, N, res, relu) in [(4,12,12,256,1024,True,True),(3,12,12,128,512,True,True),(5,7,9,128,160,False,False),(16,12,12,256,384,True,True),(1,3,5,128,32,True,False),(40,12,12,256,1024,True,True),(2,12,12,512,256,False,True)]:
    x = torch.from_numpy(rng.standard_normal((B,H,W,C)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((N,C))/np.sqrt(C)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
    r = torch.from_numpy(rng.standard_normal((B,H,W,N)).astype(np.float32)).cuda() if res else None
    ref = x.double().reshape(-1,C) @ w.double().t() + b.double()
    if res: ref = ref + r.double().reshape(-1,N)
    if relu: ref = ref.clamp_(min=0)
    lib.mpsr_debug_set_conv_pointwise(2)
    got = dn.conv2d(x, w, b, r, 1, 1, 1, relu, split_k=0)
    got2 = dn.conv2d(x, w, b, r, 1, 1, 1, relu, split_k=0)
    lib.mpsr_debug_set_conv_pointwise(-1)
    err = (got.double().reshape(-1,N) - ref).abs().max().item()
    print((B,H,W,C,N,res,relu), "err %.3g" % err, "det", torch.equal(got, got2))
