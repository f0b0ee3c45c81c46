#!/usr/bin/env python3
"""How long does ONE rank's share of the bench frame take on one GPU?  N = 1, 2, 4, 8; the slowest rank counts;
N x share / whole frame is the compute-side strong-scaling efficiency before the exchange of the finished frame.
Round 6, two deals of the 1080p / 256 spp frame beside each other:
  blocks  (lf_set_block_deal, what bench.py runs): the 64 x 64-pixel blocks b % N == r -- the rank's pre-pass builds, and its
          audit checks, only the rows of its own blocks, its march reads only those: one launch of lf_trace_ghosts (mode 2)
          holds the whole share, NOTHING is exchanged but the finished blocks;
  rows    (lf_set_row_interleave, rounds 1-5): tile rows t % N == r; the rank builds its slab of the table
          (lf_set_cull_share), one all-gather completes it, every rank audits the whole table, marches its rows.
Usage (GPU box, repo root): python3 profiles/share_timing.py > gpurun_out/r06_share_timing.json"""
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


def timed(lf, fn, reps=3):
    """-> ms per call of every timed kernel class"""
    fn()
    lf.synchronize()
    lf.timing_reset()
    for _ in range(reps):
        fn()
    lf.synchronize()
    return {k: lf.timing_get(k)[1] / reps for k in ("march", "cull_prepass", "cull_audit")}


out = {"frame": f"{W}x{H}, {spp} spp, primary + 45 pairs x 3 wavelengths, sun {SUN[0]}", "blocks": {}, "rows": {}}
for n in (1, 2, 4, 8):
    # ---- dealt by blocks: one launch is the share
    shares = []
    for r in range(n):
        lf = ctx()
        lf.set_block_deal(r, n)
        lf.set_march_culling(2)
        t = timed(lf, lambda: lf.trace_ghosts(spp, 1))
        assert lf.cull_info()["culled"]
        shares.append(t)
        lf.close()
    tot = [t["march"] + t["cull_prepass"] + t["cull_audit"] for t in shares]
    out["blocks"][n] = {"march_ms_per_rank": [t["march"] for t in shares], "prepass_ms_per_rank": [t["cull_prepass"] for t in shares],
                        "audit_ms_per_rank": [t["cull_audit"] for t in shares], "share_ms_per_rank": tot,
                        "slowest_share_ms": max(tot), "fastest_share_ms": min(tot), "spread": max(tot) / min(tot)}
    # ---- dealt by tile rows: slab of the pre-pass + (audit of the whole table + march of the rows) under the complete table
    full = ctx()
    full.set_march_culling(1)
    shares = []
    for r in range(n):
        full.set_row_interleave(r, n)
        t = timed(full, lambda: full.trace_ghosts(spp, 1))                 # (mode 1: the resident table, the march alone)
        if n == 1:
            one = ctx()
            one.set_march_culling(2)
            tp = timed(one, lambda: one.trace_ghosts(spp, 1))
            slab, audit = tp["cull_prepass"], tp["cull_audit"]
            one.close()
        else:
            sl = ctx()
            sl.set_cull_share(r, n)
            slab = timed(sl, lambda: sl.cull_prepare(spp))["cull_prepass"]
            sl.close()
            audit = out["rows"][1]["audit_ms_per_rank"][0]                  # every rank audits the whole table
        shares.append((t["march"], slab, audit))
    full.close()
    tot = [sum(x) for x in shares]
    out["rows"][n] = {"march_ms_per_rank": [x[0] for x in shares], "prepass_slab_ms_per_rank": [x[1] for x in shares],
                      "audit_ms_per_rank": [x[2] for x in shares], "share_ms_per_rank": tot, "slowest_share_ms": max(tot),
                      "fastest_share_ms": min(tot), "spread": max(tot) / min(tot)}
for deal in ("blocks", "rows"):
    base = out[deal][1]["slowest_share_ms"]
    out[f"efficiency_{deal}"] = {n: base / (n * out[deal][n]["slowest_share_ms"]) for n in out[deal]}
out["note"] = ("compute side only, one GPU playing each rank in turn.  Not in it: the all-gather of the finished frame (49.8 MB, "
               "overlapped with the next frame's march), the flare layer (0.13 ms) -- and, dealt by rows, the table's all-gather "
               "(16.7 MB on the critical path between pre-pass and march), which the block deal does not have")
print(json.dumps(out, indent=1))
