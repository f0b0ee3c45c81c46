#!/bin/bash
# HBM-side traffic of k_march for variants (experiments): FETCH_SIZE and WRITE_SIZE in their own passes
#   bash profiles/pmc_hbm.sh <tag> "<name>=<env assignments>" ...
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  name=${v%%=*}; envs=${v#*=}
  OUT=gpurun_out/ph_${TAG}_${name}
  rm -rf $OUT; mkdir -p $OUT
  for kv in $envs; do export "$kv"; done
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/$c.json 2> $OUT/$c.err
  done
  for kv in $envs; do unset "${kv%%=*}"; done
  python3 - "$name" $OUT <<'P'
import csv, glob, sys
from collections import defaultdict
s, n = defaultdict(float), defaultdict(int)
for f in glob.glob(sys.argv[2] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            s[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
f, w = s["FETCH_SIZE"] / max(1, n["FETCH_SIZE"]), s["WRITE_SIZE"] / max(1, n["WRITE_SIZE"])
print(f"{sys.argv[1]:12s} FETCH_SIZE {f:.4g} KiB  WRITE_SIZE {w:.4g} KiB  -> HBM bytes/launch (2 x fetch + write) {(2 * f + w) * 1024 / 1e6:.1f} MB")
P
done
