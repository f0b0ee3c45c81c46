#!/usr/bin/env python3
"""Per-kernel device time of the REFERENCE-FAITHFUL frame (find_sun_pos ->
paraxial ghosts -> starburst DFT -> flare layer -> tonemap) at 1080p and 4K through the C ABI, by the
library's own HIP events (lf_timing_*).  Usage (GPU box, repo root): python3 profiles/flare_frame_timing.py"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
out = {}
for name, (W, H) in {"1080p": (1920, 1080), "4k": (3840, 2160)}.items():
    for jitter in ("counter", "mt19937"):
        # LF_FLARE_JITTER=counter | mt19937: one mode only (per-mode counters: profiles/run_pmc_flare.sh)
        if os.environ.get("LF_FLARE_JITTER", jitter) != jitter:
            continue
        lf = pkg.LensFlare(0)
        lf.timing_enable(True)
        lf.set_frame(W, H)
        lf.set_params(1, 25.0, 1.0)
        lf.set_aperture(pkg.APERTURE_STARBURST, pkg.load_aperture_png("pentbig500_14.png"))
        lf.set_aperture(pkg.APERTURE_GHOST, pkg.load_aperture_png("octagonbokeh.png"))
        hf = 50.0
        vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
        lf.set_camera(np.eye(3), [0, 0, 0], hf, vf)
        if jitter == "counter":
            lf.set_jitter_counter(0x1e45f1a4e)
        else:
            lf.set_jitter_mt19937(5489, None)
        ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
        light = [(2 * 0.62 - 1) * ex * 10, (2 * 0.58 - 1) * ey * 10, -10.0, 1.0, 0.9, 0.5]
        rgba = None
        for rep in range(14):
            if rep == 4:
                lf.timing_reset()
                lf.synchronize()
                t0 = time.perf_counter()
            lf.find_sun_pos([light])
            lf.generate_ghost_buffer()
            lf.render_flare_layer()
            rgba = lf.save_image_rgba()
        lf.synchronize()
        wall = (time.perf_counter() - t0) / 10
        rec = {"wall_ms_per_frame_incl_rgba_readback": wall * 1e3}
        for k in ("frame_setup", "ghost_raster", "dft", "flare_layer", "tonemap"):
            n, ms = lf.timing_get(k)
            rec[k + "_ms"] = ms / max(n, 1) if n else None
        px = W * H
        # compulsory traffic of the flare layer: ghost in, sample + starburst out (24 B each), the
        # jitter table in MT19937 mode (32 draws x 4 B per pixel)
        alg = px * (72 + (128 if jitter == "mt19937" else 0))
        rec["flare_layer_algorithmic_bytes"] = alg
        rec["flare_layer_GBps"] = alg / (rec["flare_layer_ms"] * 1e-3) / 1e9
        out[f"{name}_{jitter}"] = rec
        lf.close()
print(json.dumps(out, indent=1))
