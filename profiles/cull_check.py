#!/usr/bin/env python3
"""Path culling on the bench frame (c3: 1080p, 256 spp, double Gauss, primary + 45 pairs x 3 wavelengths) and on
frames with the sun elsewhere: the culled march (lf_cull.hip) against the full enumeration (k_march) -- pixels must
be IDENTICAL, bit for bit -- with what each costs and what the table holds.  GPU box:

    python3 profiles/cull_check.py > gpurun_out/r05_cull_check.json
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
W, H = 1920, 1080
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
efl = pkg.paraxial_efl(lens)

lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
KEY = 0x1e45f1a4e


def run(mode, spp, key=KEY, reps=3):
    lf.set_march_culling(mode)
    lf.trace_ghosts(spp, key)
    lf.synchronize()
    lf.reset_counters()
    lf.timing_reset()
    lf.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        lf.trace_ghosts(spp, key)
    lf.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    lf.timing_enable(False)
    n, ms = lf.timing_get("march")
    nc, msc = lf.timing_get("cull_prepass")
    c, st = lf.counters(), lf.march_stats()
    img = lf.read_buffer(pkg.GHOST_BUFFER)
    return img, {"wall_ms": wall, "march_ms": ms / max(n, 1), "prepass_ms": msc / max(nc, 1), "prepass_launches": nc,
                 "counters": {k: v // reps for k, v in c.items()}, "executed_events": st["executed_events"] // reps}


def table_stats():
    t = lf.cull_table()
    if t is None:
        return None
    cells = t[..., :-1]
    bits = np.unpackbits(cells.view(np.uint8), axis=-1).reshape(cells.shape + (64,)).sum(axis=-1) if cells.size < 5e7 else None
    out = {"shape": list(t.shape), "entries_nonzero": float((cells != 0).mean())}
    if bits is not None:
        out["paths_per_entry_mean"] = float(bits.mean())
        out["paths_per_nonzero_entry_mean"] = float(bits[bits > 0].mean()) if (bits > 0).any() else 0.0
    return out


out = {"frame": f"{W}x{H}", "cases": []}
cases = [("c3 bench sun", bench.sun_direction(lens, efl, W, H), 0.05, 256),
         ("sun off axis", [0.12, 0.08, -1.0], 0.05, 64),
         ("sun near the corner", [0.30, -0.17, -1.0], 0.05, 64),
         ("sun outside the frame", [0.45, 0.1, -1.0], 0.05, 64),
         ("small sun", [0.05, 0.02, -1.0], 0.004, 64),
         ("wide sun", [-0.08, 0.03, -1.0], 0.2, 64),
         ("on axis", [0.0, 0.0, -1.0], 0.05, 16)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for name, sun, alpha, spp in cases:
    lf.set_sun(sun, [1.0, 0.9, 0.5], alpha)
    full, rf = run(0, spp)
    cul, rc = run(2, spp)          # table rebuilt at every launch: the pre-pass is inside the timing
    stats = table_stats()
    cul1, rc1 = run(1, spp)        # table reused
    diff = int((full != cul).sum())
    rec = {"case": name, "sun": sun, "alpha": alpha, "spp": spp, "pixels_differing": diff, "pixels_differing_reused_table": int((full != cul1).sum()),
           "max_abs_diff": float(np.abs(full - cul).max()), "frame_sum": float(full.sum()), "lit_values": int((full > 0).sum()),
           "full": rf, "culled_rebuilt_each_launch": rc, "culled_table_reused": rc1, "table": stats,
           "speedup_march": rf["march_ms"] / max(rc["march_ms"] + rc["prepass_ms"], 1e-9)}
    if diff:
        bad = np.argwhere(full != cul)
        rec["first_differences"] = [[int(v) for v in b] + [float(full[tuple(b)]), float(cul[tuple(b)])] for b in bad[:8]]
        rec["sum_abs_diff_over_sum"] = float(np.abs(full - cul).sum() / max(full.sum(), 1e-300))
    out["cases"].append(rec)
    print(f"{name}: differing {diff}, full {rf['march_ms']:.2f} ms, culled {rc['march_ms']:.2f} + prepass {rc['prepass_ms']:.2f} ms, "
          f"events {rf['executed_events']:.3e} -> {rc['executed_events']:.3e}", file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
lf.close()
