#!/usr/bin/env python3
"""What the boxes NOTHING bounds cost the march: the bench frame's culled march with the final level's
'kept unseen' boxes dropped -- UNSAFE, a measurement of the upper bound only (lf_test_knob cull_disable bits 4 / 5) --
beside the shipped table.  -> gpurun_out/r06_cull_unbounded_cost.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
W, H, spp = 1920, 1080, 256
efl = pkg.paraxial_efl(lens)
sun = [(0.521445 - 0.5) * 36.0 / efl, (0.517156 - 0.5) * 36.0 * H / W / efl, -1.0]
lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
lf.set_ghost_pairs(None, True)
lf.set_march_culling(2)
lf.set_cull_audit(0)
out = {}
only = sys.argv[1:]
for name, knobs in (("shipped", {}), ("every_started_path_alone_round5_march", {"cull_no_prefix": 1}), ("general_kernel_same_rules", {"cull_general_kernel": 1}),
                    ("UNSAFE_too_few_samples_dropped", {"cull_disable": 16}),
                    ("UNSAFE_lost_samples_dropped", {"cull_disable": 32}), ("UNSAFE_both_dropped", {"cull_disable": 48}),
                    ("UNPROVEN_lobe_k_1", {"cull_lobe_k": 1.0}), ("UNPROVEN_lobe_k_1_margin_1", {"cull_lobe_k": 1.0, "cull_margin": 1.0}),
                    ("UNPROVEN_margin_1", {"cull_margin": 1.0})):
    if only and name not in only:
        continue
    lf.test_knob("cull_general_kernel", 0)
    lf.test_knob("cull_no_prefix", 0)
    for k, v in knobs.items():
        lf.test_knob(k, v)
    lf.trace_ghosts(spp, 1)
    lf.synchronize()
    lf.reset_counters()
    lf.timing_reset()
    lf.timing_enable(True)
    t0 = time.perf_counter()
    for i in range(3):
        lf.trace_ghosts(spp, 2 + i)
    lf.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    lf.timing_enable(False)
    c = lf.counters()
    out[name] = {"frame_ms": ms, "march_ms": lf.timing_get("march")[1] / 3, "prepass_ms": lf.timing_get("cull_prepass")[1] / 3,
                 "started_fraction": lf.cull_started_fraction(), "executed_events_per_frame": lf.executed_events() / 3,
                 "lit_rays_per_frame": c["rays_hit_light"] / 3}
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_cull_unbounded_cost.json"), "w"), indent=1)
