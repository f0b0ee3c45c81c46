#!/usr/bin/env python3
"""Install a summary produced by profiles/run_pmc_march.sh as the profile bench.py prices its roofline
with:  python3 profiles/install_pmc.py gpurun_out/pmc_<tag>/summary.json [config] [round, default r03]"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
dst = os.path.join(root, "profiles", f"{rnd}_march_pmc.json")
s = json.load(open(sys.argv[1]))
cfg = sys.argv[2] if len(sys.argv) > 2 else s.get("config", "c3")
allp = json.load(open(dst)) if os.path.exists(dst) else {}
allp[cfg] = s
allp["_how"] = ("bash profiles/run_pmc_march.sh <tag> <config> on the GPU box: rocprofv3 --kernel-trace --pmc in separate "
                "passes (SQ / FETCH_SIZE / WRITE_SIZE / misc) over `bench.py --steps 1 --warmup 0 --no-cpu`, reduced by "
                "profiles/summarize_pmc.py (FETCH_SIZE x2 gfx950 correction); source_sha = sha256 of lf_march.hip + lf_march_events.h + "
                "lf_internal.h + the Makefile's compile flags at profile time")
json.dump(allp, open(dst, "w"), indent=1)
print("installed", cfg, "source_sha", s.get("source_sha"))
