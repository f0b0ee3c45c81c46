#!/bin/bash
# Profile of the scene-radiance kernel k_scene_term on the 18.5 k-primitive timing
# scene of profiles/scene_term_timing.py: rocprofv3 kernel stats + PMC counters in SEPARATE passes.
#   bash profiles/run_pmc_scene.sh <tag>            (GPU box, repo root; nothing is built here)
set -e
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_scene_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 profiles/scene_term_timing.py > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 profiles/scene_term_timing.py > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 profiles/scene_term_timing.py > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 profiles/scene_term_timing.py > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cache -- python3 profiles/scene_term_timing.py > $OUT/cache.json 2> $OUT/cache.err || true
python3 - $OUT <<'P'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for kern in ("k_scene_term<false>", "k_scene_term<true>"):
    sums, disp, t = defaultdict(float), defaultdict(set), []
    for f in glob.glob(out + "/*/*/*counter_collection.csv"):
        if "/stats/" in f:
            continue
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                sums[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for f in glob.glob(out + "/stats/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                t.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    pl = {c: sums[c] / max(1, len(disp[c])) for c in sums}
    e = {"launches_timed": len(t), "ms_per_launch": (sum(t) / len(t)) if t else None, "per_launch": pl}
    if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
        e["hbm_bytes_per_launch"] = (2.0 * pl["FETCH_SIZE"] + pl["WRITE_SIZE"]) * 1024.0   # KiB; gfx950 FETCH_SIZE x2
    res[kern] = e
try:
    res["timing_script"] = json.loads(open(out + "/stats.json").read())
except Exception as ex:
    res["timing_script_error"] = str(ex)
print(json.dumps(res, indent=1))
P
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
