#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs of profiles/run_pmc.sh to one JSON: per-launch averages for the
march kernel, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md (HBM section)."""
import csv
import glob
import json
import sys
from collections import defaultdict

out_dir = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_march"
sums, disp = defaultdict(float), defaultdict(set)
for f in glob.glob(out_dir + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        c = row["Counter_Name"]
        sums[c] += float(row["Counter_Value"])
        disp[c].add(row["Dispatch_Id"])
res = {"kernel": kern, "per_launch": {c: sums[c] / max(1, len(disp[c])) for c in sums},
       "launches": {c: len(disp[c]) for c in sums}}
pl = res["per_launch"]
if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports half of a wide streaming read
    res["hbm_bytes_per_launch"] = (2.0 * pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0
    res["hbm_bytes_per_launch_uncorrected"] = (pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0
if "SQ_THREAD_CYCLES_VALU" in pl and "SQ_ACTIVE_INST_VALU" in pl and pl["SQ_ACTIVE_INST_VALU"]:
    res["valu_lane_utilisation"] = pl["SQ_THREAD_CYCLES_VALU"] / (64.0 * pl["SQ_ACTIVE_INST_VALU"])
if "SQ_ACTIVE_INST_VALU" in pl and "SQ_BUSY_CYCLES" in pl and pl["SQ_BUSY_CYCLES"]:
    res["valu_busy_fraction_raw"] = pl["SQ_ACTIVE_INST_VALU"] / pl["SQ_BUSY_CYCLES"]
print(json.dumps(res, indent=1))
