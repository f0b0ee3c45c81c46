#!/bin/bash
# Extra SQ counters (issue breakdown) for the march kernel.  bash profiles/run_pmc2.sh <tag>
set -e
# the library must exist BEFORE the profiler starts: nothing may build (exec hipcc) under rocprofv3
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing: run __graft_entry__.build() first" >&2; exit 1; }
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc2_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.json 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.json 2> $OUT/b.err
python3 - "$OUT" <<'P'
import csv, glob, sys, json
from collections import defaultdict
s=defaultdict(float)
for f in glob.glob(sys.argv[1]+"/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march(" in r["Kernel_Name"] or "k_marchE" in r["Kernel_Name"] or ("k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]):
            s[r["Counter_Name"]]+=float(r["Counter_Value"])
print(json.dumps(s, indent=1))
P
