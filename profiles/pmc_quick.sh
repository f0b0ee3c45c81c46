#!/bin/bash
# One rocprofv3 --pmc pass (8 SQ counters) + kernel time for variants of k_march (experiments):
#   bash profiles/pmc_quick.sh <tag> "<name>=<env assignments>" ...
# nothing is built here: the libraries must exist before the profiler starts
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  OUT=gpurun_out/pq_${TAG}_${name}
  rm -rf $OUT; mkdir -p $OUT
  for kv in $envs; do export "$kv"; done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench.json 2> $OUT/err.txt
  for kv in $envs; do unset "${kv%%=*}"; done
  python3 - "$name" $OUT <<'P'
import csv, glob, sys, json
from collections import defaultdict
s = defaultdict(float); t = 0.0
for f in glob.glob(sys.argv[2] + "/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            s[r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(sys.argv[2] + "/sq/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            t += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
ms = None
try:
    b = json.loads([l for l in open(sys.argv[2] + "/bench.json") if l.startswith("{")][-1])
    ms = b["roofline"]["avg_launch_ms"]
except Exception:
    pass
g = lambda k: s.get(k, 0.0)
print(f"{sys.argv[1]:14s} ms {ms if ms is None else round(ms,1)} trace_ms {t:.1f} VALU {g('SQ_INSTS_VALU'):.4g} SALU {g('SQ_INSTS_SALU'):.4g} BR {g('SQ_INSTS_BRANCH'):.4g} SMEM {g('SQ_INSTS_SMEM'):.4g} "
      f"wave_cyc {g('SQ_WAVE_CYCLES'):.4g} wait_inst {g('SQ_WAIT_INST_ANY'):.4g} wait_any {g('SQ_WAIT_ANY'):.4g} act_valu {g('SQ_ACTIVE_INST_VALU'):.4g}")
P
done
