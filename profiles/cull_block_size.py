#!/usr/bin/env python3
"""How large may a cull block be?  The pre-pass bounds the image of a (64 x 64 pixel block) x (pupil cell) box from 13
rays; the larger the block in millimetres, the less linear the map over it.  Frames of decreasing width on the same
36 mm sensor (block = 64 x 36 / W mm), culled (forced) against the full enumeration: pixels differing.
    LF_CULL_FORCE=1 python3 profiles/cull_block_size.py > gpurun_out/r05_cull_block_size.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
lf = pkg.LensFlare(0)
out = []
for W in [int(v) for v in os.environ.get("WIDTHS", "1920,1280,960,800,640,480,320,192,96").split(",")]:
    H = W * 9 // 16
    spp = 64 if W >= 640 else 256
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    for sun, alpha in (([0.03, 0.02, -1.0], 0.05), ([0.2, -0.1, -1.0], 0.05), ([0.05, 0.3, -1.0], 0.02), ([0.0, 0.0, -1.0], 0.01)):
        lf.set_sun(sun, [1.0, 0.9, 0.5], alpha)
        res = {}
        for mode in (0, 2):
            lf.set_march_culling(mode)
            lf.reset_counters()
            lf.trace_ghosts(spp, 0xB10C)
            res[mode] = (lf.read_buffer(pkg.GHOST_BUFFER), lf.counters(), lf.cull_info()["culled"])
        d = res[0][0] != res[2][0]
        rec = {"W": W, "H": H, "spp": spp, "block_mm": 64 * 36.0 / W, "sun": sun, "alpha": alpha, "culled": res[2][2],
               "values_differing": int(d.sum()), "lit_values": int((res[0][0] > 0).sum()),
               "rel_sum_diff": float(np.abs(res[0][0] - res[2][0]).sum() / max(res[0][0].sum(), 1e-300)),
               "lit_rays_full": res[0][1]["rays_hit_light"], "lit_rays_culled": res[2][1]["rays_hit_light"],
               "started": res[2][1]["rays_launched"] / res[0][1]["rays_launched"]}
        out.append(rec)
        print(rec, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
lf.close()
