#!/usr/bin/env python3
"""Debug view of the cull table on the bench frame: per path the fraction of (block, cell) entries enabled, and the
16 x 16 cell map of a few (block, path) combinations.  GPU box: python3 profiles/cull_dump.py > gpurun_out/cull_dump.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "256"))
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(bench.sun_direction(lens, pkg.paraxial_efl(lens), W, H), [1.0, 0.9, 0.5], 0.05)
lf.set_march_culling(2)
lf.trace_ghosts(SPP, 1)
t = lf.cull_table()
info = lf.cull_info()
print(info, t.shape)
n, stop = lens["n"], lens["stop"]
paths = [(-1, -1)] + [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
cells = t[..., :-1]
G = info["P"]
for q, (i, j) in enumerate(paths):
    b = (cells >> np.uint64(q)) & np.uint64(1)
    print(f"path {q:2d} ({i:2d},{j:2d}): enabled {b.mean():.4f}  blocks with any {b.any(axis=2).mean():.3f}")
by, bx = int(0.517 * H) // 64, int(0.521 * W) // 64
for (yy, xx) in ((by, bx), (by, bx + 8), (2, 3)):
    for q in (0, 11, 37):
        b = ((cells[yy, xx] >> np.uint64(q)) & np.uint64(1)).reshape(G, G)
        print(f"block ({yy},{xx}) path {q} {paths[q]}: {int(b.sum())} cells")
        for row in b:
            print("   " + "".join("#" if v else "." for v in row))
lf.close()
