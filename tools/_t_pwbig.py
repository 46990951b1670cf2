import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from monopsr_amd import _lib
from monopsr_amd.core import device_net as dn
lib = _lib.lib()
torch.manual_seed(0)
for (B, H, W, C, N, res, relu) in [(256,12,12,256,1024,True,True),(256,12,12,128,512,True,True),(128,12,12,256,1024,True,False),(100,12,12,256,512,True,False)]:
    x = torch.randn((B,H,W,C), device='cuda')
    w = torch.randn((N,C), device='cuda')/np.sqrt(C)
    b = torch.randn((N,), device='cuda')
    r = torch.randn((B,H,W,N), device='cuda') if res else None
    lib.mpsr_debug_set_conv_pointwise(0)
    ref = dn.conv2d(x, w, b, r, 1, 1, 1, relu, split_k=0)
    lib.mpsr_debug_set_conv_pointwise(1)
    got = dn.conv2d(x, w, b, r, 1, 1, 1, relu, split_k=0)
    lib.mpsr_debug_set_conv_pointwise(-1)
    d = (got-ref).abs().reshape(B*H*W, N)
    bad = (d.max(dim=1).values > 1e-3).nonzero().flatten()
    print((B,H,W,C,N,res,relu), "max err %.3g" % d.max().item(), "bad rows", bad.numel(), bad[:8].tolist(), sorted(set(((bad // 96) // 8).tolist())) if bad.numel() else "", "cols", (d.max(dim=0).values > 1e-3).nonzero().flatten()[:6].tolist(), int((d.max(dim=0).values > 1e-3).sum()))
