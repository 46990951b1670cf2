"""One fused EMD loss (256 x 2048^2) under rocprofv3 --kernel-trace: prints every launch's duration in order.
    rocprofv3 --kernel-trace --output-format csv -d out -o emd -- python3 tools/emd_pass_times.py [skip]
    python tools/emd_pass_times.py --report out/emd_kernel_trace.csv"""
import csv
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if "emd_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-23:]  # the last call
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")
        print("%-40s %8.1f us" % (name[:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    sys.exit(0)
import torch  # noqa: E402
sys.path.insert(0, os.getcwd())
from monopsr_amd import _lib  # noqa: E402
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am  # noqa: E402
_lib.lib().mpsr_debug_set_emd_skip(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator(device="cuda").manual_seed(6)
y1 = torch.rand((256, 2048, 3), device="cuda", generator=g) * 2 - 1
y2 = torch.rand((256, 2048, 3), device="cuda", generator=g) * 2 - 1
for _ in range(3):
    am.emd_loss_fwd_bwd(y1, y2)
torch.cuda.synchronize()
