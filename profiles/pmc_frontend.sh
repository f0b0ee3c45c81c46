#!/bin/bash
# Front-end counters of k_march (instruction cache, instruction fetch, scalar data cache): two
# rocprofv3 --pmc passes over one bench frame.  Usage (GPU box, repo root): bash profiles/pmc_frontend.sh <tag>
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=${1:-fe}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fe_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu ${FE_ARGS}"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.json 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQC_TC_INST_REQ SQC_TC_STALL --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.json 2> $OUT/b.err
python3 - $OUT <<'P'
import csv, glob, sys, json
from collections import defaultdict
s = defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            s[r["Counter_Name"]] += float(r["Counter_Value"])
print(json.dumps(dict(s), indent=1))
P
