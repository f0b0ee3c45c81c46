#!/usr/bin/env python3
"""Which paths lose lit rays to the cull (diagnostic): per path, full vs culled (forced) lit-ray counts and sums."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
H = W * 9 // 16
sun = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.2, -0.1, -1.0]
lf = pkg.LensFlare(0)
lf.set_frame(W, H); lf.set_aperture(pkg.APERTURE_STARBURST, mask); lf.set_lens(lens); lf.set_sun(sun, [1, .9, .5], 0.05)
n, stop = lens["n"], lens["stop"]
paths = [(-1, -1)] + [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
only = [int(v) for v in os.environ.get("ONLY", "").split(",") if v]
for q, (i, j) in enumerate(paths):
    if only and q not in only:
        continue
    lf.set_ghost_pairs([(i, j)], False)
    r = {}
    for mode in (0, 2):
        lf.set_march_culling(mode); lf.reset_counters(); lf.trace_ghosts(64, 0xB10C)
        r[mode] = (lf.counters(), lf.read_buffer(pkg.GHOST_BUFFER))
    miss = r[0][0]["rays_hit_light"] - r[2][0]["rays_hit_light"]
    if miss or (r[0][1] != r[2][1]).any():
        d = np.abs(r[0][1] - r[2][1])
        print(f"path {q} ({i},{j}): missed lit rays {miss} of {r[0][0]['rays_hit_light']}, values differing {(d > 0).sum()}, max diff {d.max():.3e}, frame max {r[0][1].max():.3e}", flush=True)
print("done")
