"""Per-kernel durations from a rocprofv3 --kernel-trace CSV: median us of every kernel whose name contains one of the
given substrings, by launch position within a step if --per-step N launches repeat.
    python tools/kernel_times.py <kernel_trace.csv> upconv_gather pw_conv wino4"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
by = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    for k in keys:
        if k in n:
            short = n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            by[(short, r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v = sorted(v)
    print("%-60s grid %-8s x %-5s n=%4d  median %8.1f us  min %8.1f" % (k[0][:60], k[1], k[2], len(v), v[len(v) // 2], v[0]))
