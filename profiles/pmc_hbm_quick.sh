#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_march in their own passes (MI355X_MICROARCH.md: separate --pmc passes; gfx950
# FETCH_SIZE x2) for the current library or LF_LIB:  bash profiles/pmc_hbm_quick.sh <tag> [env assignments]
[ -f lens-flare_amd/liblensflare_hip.so ] || { echo "liblensflare_hip.so missing" >&2; exit 1; }
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/hbm_$TAG; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/$c.json 2> $OUT/$c.err
done
python3 - $OUT <<'Q'
import csv, glob, sys
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, n = 0.0, set()
    for f in glob.glob(sys.argv[1] + f"/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_march" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print(c, "KiB per launch", v / max(1, len(n)))
Q
