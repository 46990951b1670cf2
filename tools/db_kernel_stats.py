"""Per-kernel totals from a rocprofv3 results database (ROCm 7.2 writes <name>_results.db): calls, total ms, average us.
    python tools/db_kernel_stats.py out_results.db [--top 25] [--csv]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--csv", action="store_true")
args = ap.parse_args()
cur = sqlite3.connect(args.db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
total = sum(r[2] for r in rows)
if args.csv:
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage"')
    for n, c, t, a, p in rows:
        print('"%s",%d,%d,%.1f,%.4f' % (n.replace('"', "'"), c, round(t * 1e3), a * 1e3, p))
else:
    print("total kernel time %.3f ms" % (total / 1e3))
    for n, c, t, a, p in rows[:args.top]:
        print("%6d x %9.1f us = %9.3f ms  %5.2f %%  %s" % (c, a, t / 1e3, p, n[:110]))
