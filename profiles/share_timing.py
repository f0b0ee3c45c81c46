#!/usr/bin/env python3
"""How long does ONE rank's share of the bench frame take on one GPU?  (tile rows t % N == 0 of the
1080p / 256 spp frame, N = 1, 2, 4, 8.)  N x share / whole frame is the compute-side strong-scaling
efficiency before the exchange.  Usage (GPU box, repo root): python3 profiles/share_timing.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
from goldenlib import load_texels  # noqa: E402

pkg = g.load_package()
lf = pkg.LensFlare(0)
W, H, spp = 1920, 1080, 256
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, load_texels("pentbig500_14.png"))
lf.set_lens(pkg.load_lens_file("dgauss11.lens"))
lf.set_sun([0.08, 0.05, -1.0], [1.0, 0.9, 0.5], 0.05)
lf.set_ghost_pairs(None, True)
out = {}
for n in (1, 2, 4, 8):
    lf.set_row_interleave(0, n)
    lf.trace_ghosts(spp, 1)
    lf.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        lf.trace_ghosts(spp, 1)
    lf.synchronize()
    out[n] = (time.perf_counter() - t0) / 3 * 1e3
print(json.dumps({"ms_per_share": out, "efficiency": {n: out[1] / (n * out[n]) for n in out}}))
