#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs of profiles/run_pmc_march.sh to one JSON: per-launch averages for
the march kernel with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md (HBM section), the
per-event instruction counts bench.py prices its roofline with, and the identity of the sources the
profiled library was built from."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_march"
cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sums, disp = defaultdict(float), defaultdict(set)
for f in glob.glob(out_dir + "/*/*/*counter_collection.csv"):
    if "/stats/" in f:
        continue
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if kern + "(" not in name and kern + "<" not in name:   # k_march / k_march<K>, not k_march_finish / k_march_cull
            continue
        c = row["Counter_Name"]
        sums[c] += float(row["Counter_Value"])
        disp[c].add(row["Dispatch_Id"])
# the pre-pass of the path cull (k_cull_level: several launches per frame, one per level) beside the march
pre, pre_disp = defaultdict(float), defaultdict(set)
for f in glob.glob(out_dir + "/*/*/*counter_collection.csv"):
    if "/stats/" in f:
        continue
    for row in csv.DictReader(open(f)):
        if "k_cull_level" in row["Kernel_Name"]:
            pre[row["Counter_Name"]] += float(row["Counter_Value"])
            pre_disp[row["Counter_Name"]].add(row["Dispatch_Id"])
res = {"kernel": kern, "config": cfg,
       "per_launch": {c: sums[c] / max(1, len(disp[c])) for c in sums},
       "launches": {c: len(disp[c]) for c in sums}}
if pre:
    res["cull_prepass"] = {"kernel": "k_cull_level", "launches": {c: len(pre_disp[c]) for c in pre},
                           "sum_over_launches": dict(pre)}
# ... and the audit of what the table drops (k_cull_audit: one launch per table; round 6)
aud, aud_disp = defaultdict(float), defaultdict(set)
for f in glob.glob(out_dir + "/*/*/*counter_collection.csv"):
    if "/stats/" in f:
        continue
    for row in csv.DictReader(open(f)):
        if "k_cull_audit" in row["Kernel_Name"]:
            aud[row["Counter_Name"]] += float(row["Counter_Value"])
            aud_disp[row["Counter_Name"]].add(row["Dispatch_Id"])
if aud:
    res["cull_audit"] = {"kernel": "k_cull_audit", "launches": {c: len(aud_disp[c]) for c in aud}, "sum_over_launches": dict(aud)}
h = hashlib.sha256()
for f in ("lens-flare_amd/csrc/lf_march.hip", "lens-flare_amd/csrc/lf_cull.hip", "lens-flare_amd/csrc/lf_march_common.h",
          "lens-flare_amd/csrc/lf_march_events.h", "lens-flare_amd/csrc/lf_internal.h"):
    h.update(open(os.path.join(root, f), "rb").read())
mk = open(os.path.join(root, "lens-flare_amd", "Makefile")).read()
h.update(mk[mk.index("FLAGS  :="):mk.index("SRCS   :=")].encode())   # the compile flags, as bench.py hashes them
res["source_sha"] = h.hexdigest()[:16]
pl = res["per_launch"]
if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports half of a wide streaming read
    res["hbm_bytes_per_launch"] = (2.0 * pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0
    res["hbm_bytes_per_launch_uncorrected"] = (pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0
# the bench line printed by the SQ pass: executed events of that very run
try:
    line = [l for l in open(os.path.join(out_dir, "sq.json")) if l.startswith("{")][-1]
    b = json.loads(line)
    ev = b["config"]["events_executed_per_frame"]
    res["executed_events_per_launch"] = ev
    res["ms_per_launch_under_pmc"] = b["roofline"]["avg_launch_ms"]
    if "SQ_INSTS_VALU" in pl:
        res["valu_wave_instr_per_launch"] = pl["SQ_INSTS_VALU"]
        res["valu_wave_instr_per_executed_event"] = pl["SQ_INSTS_VALU"] / ev
        res["valu_lane_slots_per_event"] = pl["SQ_INSTS_VALU"] * 64.0 / ev
    if "SQ_INSTS_SALU" in pl:
        sc = pl["SQ_INSTS_SALU"] + pl.get("SQ_INSTS_BRANCH", 0.0) + pl.get("SQ_INSTS_SMEM", 0.0)
        res["scalar_instr_per_launch"] = sc
        res["scalar_instr_per_executed_event"] = sc / ev
    if "GRBM_GUI_ACTIVE" in pl:
        res["clock_ghz"] = pl["GRBM_GUI_ACTIVE"] / 8.0 / (b["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
except Exception as e:  # noqa: BLE001
    res["note"] = f"no bench line next to the counters: {e}"
try:
    line = [l for l in open(os.path.join(out_dir, "stats.json")) if l.startswith("{")][-1]
    b = json.loads(line)
    res["unprofiled_style_run"] = {"ms_per_step": b["ms_per_step"], "march_avg_launch_ms": b["roofline"]["avg_launch_ms"],
                                   "value": b["value"]}
except Exception:
    pass
print(json.dumps(res, indent=1))
