import sys, os, torch
sys.path.insert(0, os.getcwd())
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for b, n in ((256, 2048), (32, 2304)):
    y1 = torch.rand((b, n, 3), device="cuda") * 2 - 1
    y2 = torch.rand((b, n, 3), device="cuda") * 2 - 1
    print(b, n, "approx_match %.2f ms  fused %.2f  cost-only %.2f" % (timeit(lambda: am.approx_match(y1, y2), 5), timeit(lambda: am.emd_loss_fwd_bwd(y1, y2), 5), timeit(lambda: am.emd_loss_fwd_bwd(y1, y2, want_grads=False), 5)))
