"""rocprofv3 --pmc counter_collection CSVs of bench.py -> one JSON: per conv kernel instantiation the launches,
average duration, HBM bytes per launch and MFMA-busy fraction; the same over all conv launches (what bench.py quotes).

    python tools/pmc_reduce.py <round tag> <commit> <pass1.csv> <pass2.csv> ...

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are KiB per dispatch, and on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM).
MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); clock = GUI_ACTIVE / 8 / duration.
"""
import csv
import json
import os
import re
import sys
import time
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main(tag, commit, paths):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))  # kernel -> counter -> [n, sum, dur_us]
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                a = agg[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    kernels = {}
    tot = defaultdict(float)
    for k, cs in agg.items():
        if os.environ.get("PMC_EXTRA_PREFIXES") and k.startswith(tuple(os.environ["PMC_EXTRA_PREFIXES"].split(","))):
            pass  # (r06: the training step's own kernels -- weight gradients, BatchNorm, gathers -- for r06_train_pmc.json)
        elif not (k.startswith("conv_") or k.startswith("wino_conv") or k.startswith("wino4_conv") or k.startswith("wino4s_conv") or k.startswith("wino3_conv") or k.startswith("wino3w_conv") or k.startswith("wino3z_conv") or k.startswith("wino3h_conv") or k.startswith("pw_conv") or k.startswith("fc_rows") or k.startswith("conv3x3_narrow_mfma")):  # the matrix-pipe kernels of a step (bench.py's conv_replay launches the same set)
            continue
        g = lambda c: (cs[c][1] / cs[c][0]) if c in cs and cs[c][0] else None
        n = max(v[0] for v in cs.values())
        passes = max(1, len([c for c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "TCC_HIT_sum") if c in cs]))
        dur = sum(v[2] for c, v in cs.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "TCC_HIT_sum")) / \
            max(1, sum(v[0] for c, v in cs.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "TCC_HIT_sum")))
        rec = {"launches_per_pass": n, "avg_us": round(dur, 1)}
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            rec["fetch_KiB"] = round(g("FETCH_SIZE"), 1)
            rec["write_KiB"] = round(g("WRITE_SIZE"), 1)
            rec["hbm_bytes_per_launch"] = round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
            tot["hbm"] += rec["hbm_bytes_per_launch"] * n
            tot["n_hbm"] += n
        if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            rec["mfma_busy"] = round((g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024) / (g("GRBM_GUI_ACTIVE") / 8), 4)
            rec["clock_GHz"] = round(g("GRBM_GUI_ACTIVE") / 8 / (cs["GRBM_GUI_ACTIVE"][2] / cs["GRBM_GUI_ACTIVE"][0]) / 1e3, 3)
            tot["busy_cycles"] += g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 * n
            tot["active_cycles"] += g("GRBM_GUI_ACTIVE") / 8 * n
            tot["active_us"] += cs["GRBM_GUI_ACTIVE"][2] / cs["GRBM_GUI_ACTIVE"][0] * n
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            rec["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        kernels[k] = rec
    out = {"round": tag, "commit": commit, "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
           "command": os.environ.get("PMC_COMMAND", "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 2 "
                                                     "--warmup 1 --cpu-sample 0 --no-roofline --no-fast-mode (one counter "
                                                     "set per pass)"),
           "formula": "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction); mfma_busy = "
                      "(SQ_VALU_MFMA_BUSY_CYCLES/1024)/(GRBM_GUI_ACTIVE/8)",
           "kernels": kernels}
    if tot["n_hbm"]:
        out["hbm_bytes_per_launch"] = round(tot["hbm"] / tot["n_hbm"])
    if tot["active_cycles"]:
        out["mfma_busy"] = round(tot["busy_cycles"] / tot["active_cycles"], 4)
        # NOTE on units: GRBM_GUI_ACTIVE advances at a fixed ~2.45 GHz on these boards whatever the shader clock does
        # (every long kernel of every collection gives GUI_ACTIVE / 8 / duration = 2.41-2.46 GHz), while
        # SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles: mfma_busy is therefore a fraction of TIME at the nominal clock
        # -- it already contains the clock the power limit takes away -- and compares directly with bench.py's
        # roofline.frac.  The fraction of a wave's own CYCLES spent issuing MFMAs comes from the kernels' cycle stamps
        # (tools/wino3w_trace.py: 0.86 for the wave-owned F(3x3,3x3) kernel at 0.73 here).
        out["gui_active_GHz"] = round(tot["active_cycles"] / tot["active_us"] / 1e3, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
