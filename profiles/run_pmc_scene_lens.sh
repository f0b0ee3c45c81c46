#!/bin/bash
# Counters of the lens-imaged scene term on a C4-shaped frame (bench.py --config c4_1gpu: 4K, 256 samples per pixel
# offered, pyramid.dae; c4_maxplanck_1gpu: 50 801 triangles): round 5's k_scene_lens (scene rays compacted, the
# default), round 4's k_scene_term<SOFT, true> (LF_SCENE_COMPACT=0: one traversal per lane's own sample) and the same
# frame with the reference's pinhole (LF_BENCH_PINHOLE_SCENE=1): rocprofv3 kernel trace + SQ counters in their own
# pass.   bash profiles/run_pmc_scene_lens.sh <tag> [config]      (GPU box, repo root)
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=${1:-r05}
CFG=${2:-c4_1gpu}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_scene_lens_$TAG
rm -rf $OUT; mkdir -p $OUT
for mode in lens lens_per_lane pinhole; do
  if [ $mode = pinhole ]; then export LF_BENCH_PINHOLE_SCENE=1; else unset LF_BENCH_PINHOLE_SCENE; fi
  if [ $mode = lens_per_lane ]; then export LF_SCENE_COMPACT=0; else unset LF_SCENE_COMPACT; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq_$mode -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu > $OUT/sq_$mode.json 2> $OUT/sq_$mode.err
done
unset LF_BENCH_PINHOLE_SCENE LF_SCENE_COMPACT
python3 - $OUT $CFG <<'Q'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = {"frame": "bench.py --config " + sys.argv[2] + " (3840x2160, 256 samples per pixel offered)"}
for mode in ("lens", "lens_per_lane", "pinhole"):
    sums, disp, t = defaultdict(float), defaultdict(set), []
    for f in glob.glob(out + f"/sq_{mode}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_scene_" in r["Kernel_Name"] and "trace_ray" not in r["Kernel_Name"]:
                sums[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
                name = "k_scene_" + r["Kernel_Name"].split("k_scene_")[1].split("(")[0]
    for f in glob.glob(out + f"/sq_{mode}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_scene_" in r["Kernel_Name"] and "trace_ray" not in r["Kernel_Name"]:
                t.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    e = {"kernel": name if sums else None, "ms_per_launch_under_pmc": sum(t) / len(t) if t else None,
         "per_launch": {c: sums[c] / max(1, len(disp[c])) for c in sums}}
    try:
        b = json.loads([l for l in open(out + f"/sq_{mode}.json") if l.startswith("{")][-1])
        e["scene_term"] = b["scene_term"]
    except Exception as ex:  # noqa: BLE001
        e["note"] = str(ex)
    res[mode] = e
print(json.dumps(res, indent=1))
Q
