#!/bin/bash
# FETCH_SIZE per byte read, for 16-byte coalesced loads and for the F(4x4) kernel's 4-byte patch requests.
# usage (GPU box): tools/fetch_calibration.sh [outdir]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/fetch_cal}; mkdir -p $OUT
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p -o cal -- python3 tools/fetch_calibration.py > $OUT/run.log 2>&1 || echo "calibration pass failed"
python3 - "$OUT" <<'PY'
import csv, sys, glob
out = sys.argv[1]
rows = {}
for path in glob.glob(out + "/p/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "fetch_calibration" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            mode = "16B coalesced" if "<0>" in r["Kernel_Name"] else "4B patch pattern"
            rows.setdefault(mode, []).append(float(r["Counter_Value"]))
for mode, v in sorted(rows.items()):
    kib = sum(v) / len(v)
    print("%-18s FETCH_SIZE %.0f KiB for 1048576 KiB read -> counter reports %.3f of the bytes (multiply by %.2f)"
          % (mode, kib, kib / 1048576.0, 1048576.0 / kib))
PY
