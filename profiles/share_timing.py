#!/usr/bin/env python3
"""How long does ONE rank's share of the bench frame take on one GPU?  (tile rows t % N == r of the 1080p / 256 spp
frame, N = 1, 2, 4, 8; the slowest rank r counts.)  N x share / whole frame is the compute-side strong-scaling efficiency
before the exchanges.  Round 5: a share = this rank's slab of the cull pre-pass (lf_set_cull_share: blocks b % N == r)
+ the culled march of its tile rows under the COMPLETE table; `replicated` = what the frame would cost if every rank
built the whole table (the pre-pass does not shrink with N).
Usage (GPU box, repo root): python3 profiles/share_timing.py > gpurun_out/r05_share_timing.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
from goldenlib import load_texels  # noqa: E402

pkg = g.load_package()
W, H, spp = 1920, 1080, 256
SUN = ([0.08, 0.05, -1.0], [1.0, 0.9, 0.5], 0.05)
mask = load_texels("pentbig500_14.png")
lens = pkg.load_lens_file("dgauss11.lens")


def ctx():
    lf = pkg.LensFlare(0)
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(*SUN)
    lf.set_ghost_pairs(None, True)
    lf.timing_enable(True)
    return lf


def timed(lf, what, fn, reps=3):
    fn()
    lf.synchronize()
    lf.timing_reset()
    for _ in range(reps):
        fn()
    lf.synchronize()
    n, ms = lf.timing_get(what)
    return ms / max(1, n)


out = {"frame": f"{W}x{H}, {spp} spp, primary + 45 pairs x 3 wavelengths, sun {SUN[0]}", "ranks": {}}
# the march of a share under the complete table: mode 1 keeps the table between launches, so only the march is timed
full = ctx()
full.set_march_culling(1)
whole_prepass = None
for n in (1, 2, 4, 8):
    march, prepass = [], []
    for r in range(n):
        full.set_row_interleave(r, n)
        march.append(timed(full, "march", lambda: full.trace_ghosts(spp, 1)))
        if n == 1:
            one = ctx()
            one.set_march_culling(2)
            prepass.append(timed(one, "cull_prepass", lambda: one.trace_ghosts(spp, 1)))
            one.close()
        else:
            sl = ctx()
            sl.set_cull_share(r, n)
            prepass.append(timed(sl, "cull_prepass", lambda: sl.cull_prepare(spp)))
            sl.close()
    if n == 1:
        whole_prepass = prepass[0]
    out["ranks"][n] = {"march_ms_per_rank": march, "prepass_slab_ms_per_rank": prepass,
                       "slowest_share_ms": max(m + p for m, p in zip(march, prepass)),
                       "slowest_share_ms_if_the_table_were_replicated": max(march) + whole_prepass}
base = out["ranks"][1]["slowest_share_ms"]
out["efficiency_shared"] = {n: base / (n * out["ranks"][n]["slowest_share_ms"]) for n in out["ranks"]}
out["efficiency_replicated"] = {n: base / (n * out["ranks"][n]["slowest_share_ms_if_the_table_were_replicated"]) for n in out["ranks"]}
out["note"] = ("compute side only, one GPU playing each rank in turn: the two all-gathers per frame (table slabs: 16.7 MB "
               "in total; finished tile rows: 49.8 MB, overlapped with the next frame) and the flare layer (0.13 ms) are not in it")
full.close()
print(json.dumps(out, indent=1))
