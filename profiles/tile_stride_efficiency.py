#!/usr/bin/env python3
"""What the wave tile's pixel stride (lf_set_tile_stride) does on the bench frame (c3: 1080p, 256 spp,
double-Gauss, primary + 45 pairs x 3 wavelengths): frame time, and where the correlated noise of the shared
pupil sub-cell lands -- tile_correlation = 64 Var(mean of an 8 x 8 block of ADJACENT pixels) / mean pixel
variance over 6 independent keys (1 = independent pixels, 64 = the block moves as one).
    python profiles/tile_stride_efficiency.py > gpurun_out/r04_tile_stride.json      (GPU box)"""
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
lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
lf.set_ghost_pairs(None, True)
KEYS = [0x9000 + k for k in range(6)]


def measure():
    lf.trace_ghosts(SPP, KEYS[0])
    lf.synchronize()
    t0 = time.perf_counter()
    for k in KEYS[:3]:
        lf.trace_ghosts(SPP, k)
    lf.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    imgs = []
    for k in KEYS:
        lf.trace_ghosts(SPP, k)
        imgs.append(lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2))
    a = np.stack(imgs)
    tiles = a.reshape(len(KEYS), H // 8, 8, W // 8, 8).mean(axis=(2, 4))
    return ms, a.mean(0), a.var(0, ddof=1), tiles.var(0, ddof=1)


out = {"frame": f"{W}x{H}, {SPP} spp, c3 paths", "keys": len(KEYS), "variants": {}}
ref = None
VARIANTS = ((1, 2), (8, 2), (2, 2), (4, 2), (8, 3), (8, 4), (8, 5), (8, 6), (8, 8), (4, 4), (1, 4), (1, 0))   # first = rounds 1-3
for stride, bits in VARIANTS:
    lf.set_tile_stride(stride)
    lf.set_pupil_subcells(bits)
    lf.reset_counters()
    ms, m, v, tv = measure()
    cnt = lf.counters()
    if ref is None:
        lit = m > 1e-4 * m.max()
        tl = lit.reshape(H // 8, 8, W // 8, 8).all(axis=(1, 3))
        ref = (m, v)
    out["variants"][f"stride{stride}_subcells{1 << bits}x{1 << bits}"] = {
        "ms": ms,
        "tile_correlation": float(64.0 * tv[tl].sum() / v.reshape(H // 8, 8, W // 8, 8).mean(axis=(1, 3))[tl].sum()),
        "sum_pixel_variance_ratio_to_default": float(v[lit].sum() / ref[1][lit].sum()),
        "mean_total_ratio_to_default": float(m.sum() / ref[0].sum()),
        "rays_reached_scene_fraction": cnt["rays_reached_scene"] / cnt["rays_launched"]}
lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
lf.set_pupil_subcells(pkg.DEFAULT_SUBCELL_BITS)
print(json.dumps(out, indent=1))
lf.close()
