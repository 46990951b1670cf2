import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from monopsr_amd import _lib
lib = _lib.lib()
def run(B, H, W, C, N, wino, x, dy):
    dw = torch.zeros((N, 9 * C), device='cuda'); db = torch.zeros((N,), device='cuda')
    nws = lib.mpsr_conv2d_wgrad_scratch_floats(B, H, W, C, N, 3, 3, 1)
    ws = torch.empty((max(nws, 1),), device='cuda')
    lib.mpsr_debug_set_wgrad_winograd(wino)
    def call():
        dw.zero_(); db.zero_()
        _lib.check(lib.mpsr_conv2d_wgrad_ws_f32(x.data_ptr(), dy.data_ptr(), B, H, W, C, N, 3, 3, 1, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, _lib.stream()))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): call()
    e1.record(); torch.cuda.synchronize()
    lib.mpsr_debug_set_wgrad_winograd(1)
    return dw.clone(), db.clone(), e0.elapsed_time(e1) / 3 * 1e3, nws
for (B, H, W, C, N) in [(256, 24, 24, 512, 256), (256, 24, 24, 256, 256), (256, 48, 48, 256, 128), (256, 48, 48, 128, 128)]:
    torch.manual_seed(1)
    x = torch.randn((B, H, W, C), device='cuda').clamp_(min=0); dy = torch.randn((B, H, W, N), device='cuda')
    dw0, db0, us0, _ = run(B, H, W, C, N, 0, x, dy)
    dw1, db1, us1, nws = run(B, H, W, C, N, 1, x, dy)
    sc = dw0.abs().max().item()
    print((B, H, W, C, N), "direct %.0f us  winograd %.0f us" % (us0, us1), "max diff / scale %.2e" % ((dw1 - dw0).abs().max().item() / sc))
