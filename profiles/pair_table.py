#!/usr/bin/env python3
"""Where the rays of the bench frame go, PATH BY PATH (c3: 1080p, 256 spp, double Gauss, primary + 45 pairs x 3
wavelengths).  For every path marched ALONE (lf_set_ghost_pairs with one entry): its rays and their fates,
executed events, launch time, what it adds to the image (sum, lit pixels) and how many (wave tile, sample,
wavelength) combinations ended with a lane inside the sun's lobe -- the work a march that knew in advance
where the light is would have to do.  Run on the GPU box:

    python3 profiles/pair_table.py > gpurun_out/r05_pair_table.json
    LF_LIB=lens-flare_amd/build_ab/hist/liblensflare_hip.so LF_MARCH_PRINT_HIST=1 python3 profiles/pair_table.py --hist \
        > gpurun_out/r05_live_hist.json 2> gpurun_out/r05_live_hist.err   (instrumented build: live-lane histograms per row kind)
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
W, H, SPP = 1920, 1080, int(os.environ.get("LF_PAIR_SPP", "256"))
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
efl = pkg.paraxial_efl(lens)
sun = bench.sun_direction(lens, efl, W, H)
n, stop = lens["n"], lens["stop"]
paths = [(-1, -1)] + [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]

lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
KEY = 0x1e45f1a4e


def frame(pairs, primary, bits=None, stride=None):
    lf.set_ghost_pairs(pairs, primary)
    lf.set_pupil_subcells(pkg.DEFAULT_SUBCELL_BITS if bits is None else bits)
    lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE if stride is None else stride)
    lf.trace_ghosts(SPP, KEY)      # warm (tables)
    lf.synchronize()
    lf.reset_counters()
    t0 = time.perf_counter()
    lf.trace_ghosts(SPP, KEY)
    lf.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    c, st = lf.counters(), lf.march_stats()
    return ms, c, st


if "--hist" in sys.argv:
    out = {"frame": f"{W}x{H}, {SPP} spp, c3 paths", "specs": {}}
    for name, bits, stride in (("default_stride8_subcells_64x64", None, None), ("independent_pixels_subcells_1x1", 0, None),
                               ("rounds_1_to_3_stride1_subcells_4x4", 2, 1)):
        print(f"SPEC {name}", file=sys.stderr, flush=True)
        ms, c, st = frame(None, True, bits, stride)
        out["specs"][name] = {"ms": ms, "counters": c, "stats": st}
    print(json.dumps(out, indent=1))
    lf.close()
    sys.exit(0)

rows = []
ms_all, c_all, st_all = frame(None, True)
img_all = lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2)
for q, (i, j) in enumerate(paths):
    if i < 0:
        ms, c, st = frame([(-1, -1)], False)
    else:
        ms, c, st = frame([(i, j)], False)
    img = lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2)
    ev_len = n if i < 0 else n + 2 * (j - i)
    rows.append({"q": q, "i": i, "j": j, "events_per_path": ev_len, "ms_alone": ms,
                 "rays": c["rays_launched"], "clipped_stop": c["rays_clipped_stop"], "vignetted": c["rays_vignetted"],
                 "tir": c["rays_tir"], "reached_scene": c["rays_reached_scene"], "hit_light": c["rays_hit_light"],
                 "executed_events": st["executed_events"], "remarch_rows": st["remarch_rows"],
                 "remarch_lane_events": st["remarch_lane_events"],
                 "image_sum": float(img.sum()), "image_max": float(img.max()), "lit_pixels": int((img > 0).sum()),
                 "share_of_frame_sum": float(img.sum() / max(img_all.sum(), 1e-300))})
    print(f"path {q} ({i},{j}): {ms:.2f} ms, reached {c['rays_reached_scene'] / c['rays_launched']:.3f}, "
          f"lit {c['rays_hit_light'] / c['rays_launched']:.4f}, lit px {rows[-1]['lit_pixels']}", file=sys.stderr, flush=True)
out = {"frame": f"{W}x{H}, {SPP} spp, sun at {bench.SUN_NS}, key {KEY:#x}", "all_paths": {"ms": ms_all, "counters": c_all, "stats": st_all,
       "image_sum": float(img_all.sum()), "lit_pixels": int((img_all > 0).sum())}, "paths": rows,
       "note": "remarch_rows = wave-rows of the weight re-march = sum over (wave tile, sample, wavelength) with a lane inside the lobe "
               "pre-test of the path's length: the rows an oracle-guided march would still have to walk for this path"}
print(json.dumps(out, indent=1))
lf.close()
