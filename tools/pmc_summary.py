"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: launches, average duration, per-launch counter
averages.  usage: python tools/pmc_summary.py gpurun_out/pmc_r1/*/pmc_counter_collection.csv"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def main(paths):
    for path in paths:
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))  # kernel -> counter -> [n, sum, dur]
        with open(path) as f:
            for row in csv.DictReader(f):
                a = agg[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        print("# " + path)
        counters = sorted({c for k in agg.values() for c in k})
        print("| kernel | launches | avg_us | " + " | ".join(counters) + " |")
        print("|---|---|---|" + "---|" * len(counters))
        for k, cs in sorted(agg.items(), key=lambda kv: -max(v[2] for v in kv[1].values())):
            n = max(v[0] for v in cs.values())
            dur = max(v[2] for v in cs.values()) / n
            print("| %s | %d | %.1f | %s |" % (k, n, dur, " | ".join("%.4g" % (cs[c][1] / max(cs[c][0], 1)) if c in cs
                                                                   else "" for c in counters)))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
