"""Weight gradient of the full-image trunk's 1x1 layers (ONE 40x152 feature map: 6080 pixel rows -- a short reduction) for
different slicing targets.  python tools/wgrad_small_m.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

SHAPES = [("b3 conv1 1024->256", 1024, 256), ("b3 conv3 256->1024", 256, 1024), ("b2 conv3 128->512", 128, 512)]


def main():
    lib = _lib.lib()
    B, H, W = 1, 40, 152
    for target in (0, 768, 384, 256, 192, 128, 96):  # 0: the shipped rule (at least 12 steps per slice), then fixed targets
        lib.mpsr_debug_set_wgrad_min_steps(12 if target == 0 else 0)
        lib.mpsr_debug_set_wgrad_target(target or 768)
        row = []
        for name, C, N in SHAPES:
            x = torch.randn((B, H, W, C), device="cuda")
            dy = torch.randn((B, H, W, N), device="cuda")
            dw = torch.zeros((N, C), device="cuda")
            db = torch.zeros((N,), device="cuda")
            s = _lib.stream()

            def run():
                _lib.check(lib.mpsr_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), B, H, W, C, N, 1, 1, 1, _lib.ptr(dw),
                                                     _lib.ptr(db), s))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            row.append("%s %.1f us" % (name, e0.elapsed_time(e1) * 50))
        print("target %4d: %s" % (target, "   ".join(row)))
    lib.mpsr_debug_set_wgrad_target(768)
    lib.mpsr_debug_set_wgrad_min_steps(12)


if __name__ == "__main__":
    main()
