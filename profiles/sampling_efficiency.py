#!/usr/bin/env python3
"""Sampling efficiency of the geometric march on the bench frame (c3: 1080p, double-Gauss, primary + 45
pairs x 3 wavelengths): frame time at EQUAL per-pixel variance for the sampling specifications the C
ABI offers.  events/s (bench.py's metric) says how fast the device marches; this says how much image a
marched ray buys.

  default        every sample aims at the rear element's clear aperture, 4x4 pupil sub-cells per tile
  aimed          primary + front pairs aim at the paraxial exit pupil (lf_aim_at_exit_pupil, margin
                 1.1), rear pairs at the rear element, second launch accumulates (lf_set_ghost_accumulate)
  sub-cells      lf_set_pupil_subcells 0 / 2 / 4: per-pixel variance is the same by construction; what
                 changes is the correlation of the noise inside an 8x8 tile (variance of tile means)

    python profiles/sampling_efficiency.py > gpurun_out/r03_sampling_efficiency.json   (GPU box)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
W, H, SPP = 1920, 1080, 256
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
efl = pkg.paraxial_efl(lens)
sun = bench.sun_direction(lens, efl, W, H)
n, stop = lens["n"], lens["stop"]
allp = [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
PA, PB = [p for p in allp if p[0] < stop], [p for p in allp if p[0] > stop]

lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
KEYS = [0x9000 + k for k in range(6)]


def default_frame(key, spp=SPP):
    lf.set_pupil_target(0.0, 0.0)
    lf.set_ghost_pairs(None, True)
    lf.trace_ghosts(spp, key)


def aimed_frame(key, spp_a, spp_b):
    lf.set_ghost_pairs(PA, True)
    lf.aim_at_exit_pupil(1.1)
    lf.trace_ghosts(spp_a, key)
    lf.set_pupil_target(0.0, 0.0)
    lf.set_ghost_pairs(PB, False)
    lf.set_ghost_accumulate(True)
    lf.trace_ghosts(spp_b, key ^ 0x5555)
    lf.set_ghost_accumulate(False)


def measure(fn, *args):
    """-> ms per frame (steady state), mean image, per-pixel variance estimate, variance of 8x8 tile means"""
    fn(KEYS[0], *args)
    lf.synchronize()
    t0 = time.perf_counter()
    for k in KEYS[:3]:
        fn(k, *args)
    lf.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    imgs = []
    for k in KEYS:
        fn(k, *args)
        imgs.append(lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2))
    a = np.stack(imgs)
    tiles = a.reshape(len(KEYS), H // 8, 8, W // 8, 8).mean(axis=(2, 4))
    return ms, a.mean(0), a.var(0, ddof=1), tiles.var(0, ddof=1)


out = {"frame": f"{W}x{H}, c3 paths, sun at {bench.SUN_NS}", "keys": len(KEYS)}
ms0, m0, v0, tv0 = measure(default_frame)
lit = m0 > 1e-4 * m0.max()
tl = lit.reshape(H // 8, 8, W // 8, 8).all(axis=(1, 3))
V0 = float(v0[lit].sum())
def rel_var(v, m):
    """mean over the lit pixels of variance / mean^2: every lit pixel counts the same (the dim ghosts too)"""
    return float((v[lit] / np.maximum(m[lit], 1e-300) ** 2).mean())


out["default"] = {"spp": SPP, "ms": ms0, "sum_pixel_variance": V0, "mean_relative_variance": rel_var(v0, m0), "lit_pixels": int(lit.sum()),
                  "tile_correlation": float(64.0 * tv0[tl].sum() / v0.reshape(H // 8, 8, W // 8, 8).mean(axis=(1, 3))[tl].sum())}
# aimed sampling: variance at 128 + 128 spp, then the sample counts that match the default's variance
ms1, m1, v1, _ = measure(aimed_frame, 128, 128)
V1 = float(v1[lit].sum())
out["aimed_128"] = {"spp_front": 128, "spp_rear": 128, "ms": ms1, "sum_pixel_variance": V1, "mean_relative_variance": rel_var(v1, m0),
                    "mean_total_ratio_to_default": float(m1.sum() / m0.sum())}
spp_eq = max(16, int(round(128 * V1 / V0 / 4.0)) * 4)
ms2, m2, v2, _ = measure(aimed_frame, spp_eq, spp_eq)
V2 = float(v2[lit].sum())
out["aimed_equal_variance"] = {"spp_front": spp_eq, "spp_rear": spp_eq, "ms": ms2, "sum_pixel_variance": V2, "mean_relative_variance": rel_var(v2, m0),
                               "variance_ratio_to_default": V2 / V0, "speedup_at_equal_variance": ms0 / ms2 * (V0 / V2),
                               "mean_total_ratio_to_default": float(m2.sum() / m0.sum())}
# sub-cells: time, per-pixel variance (equal by construction), noise correlation inside a tile
out["subcells"] = {}
for bits in (0, 2, 3, 4):
    lf.set_pupil_subcells(bits)
    ms, m, v, tv = measure(default_frame)
    out["subcells"][str(bits)] = {
        "cells_per_stratum": (1 << bits) ** 2, "ms": ms, "sum_pixel_variance": float(v[lit].sum()), "mean_relative_variance": rel_var(v, m0),
        "mean_total_ratio_to_default": float(m.sum() / m0.sum()),
        # 1 = the 64 pixels of a tile are independent; 64 = they move together
        "tile_correlation": float(64.0 * tv[tl].sum() / v.reshape(H // 8, 8, W // 8, 8).mean(axis=(1, 3))[tl].sum())}
lf.set_pupil_subcells(pkg.DEFAULT_SUBCELL_BITS)
out["note"] = ("variance = sample variance over independent keys, summed over the lit pixels (mean > 1e-4 of the "
               "peak); tile_correlation = 64 Var(tile mean) / mean pixel variance over fully lit 8x8 tiles")
print(json.dumps(out, indent=1))
lf.close()
