#!/bin/bash
# Counters of the kernels of the reference-faithful frame (k_flare_layer, k_ghost_raster, k_tonemap) on
# profiles/flare_frame_timing.py: rocprofv3 kernel stats + PMC in SEPARATE passes.
#   bash profiles/run_pmc_flare.sh <tag>            (GPU box, repo root; nothing is built here)
set -e
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_flare_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 profiles/flare_frame_timing.py > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 profiles/flare_frame_timing.py > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 profiles/flare_frame_timing.py > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 profiles/flare_frame_timing.py > $OUT/write.json 2> $OUT/write.err
python3 - $OUT <<'P'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for kern in ("k_flare_layer", "k_ghost_raster", "k_tonemap"):
    # the script renders 1080p (counter, mt19937) then 4K (counter, mt19937), 14 frames each: split by grid size
    per = defaultdict(lambda: {"sums": defaultdict(float), "n": defaultdict(int), "t": []})
    for f in glob.glob(out + "/*/*/*counter_collection.csv"):
        if "/stats/" in f:
            continue
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                g = r.get("Grid_Size", "?")
                per[g]["sums"][r["Counter_Name"]] += float(r["Counter_Value"]); per[g]["n"][r["Counter_Name"]] += 1
    for f in glob.glob(out + "/stats/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
                per[g]["t"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    res[kern] = {}
    for g, v in per.items():
        pl = {c: v["sums"][c] / max(1, v["n"][c]) for c in v["sums"]}
        e = {"launches_timed": len(v["t"]), "ms_per_launch": (sum(v["t"]) / len(v["t"])) if v["t"] else None, "per_launch": pl}
        if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
            e["hbm_bytes_per_launch"] = (2.0 * pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0   # KiB; gfx950 FETCH_SIZE x2
        res[kern]["grid_" + str(g)] = e
print(json.dumps(res, indent=1))
P
