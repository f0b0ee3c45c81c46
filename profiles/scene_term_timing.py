#!/usr/bin/env python3
"""Measurement for SURVEY section 8 row f2 (the scene-radiance term, lf_scene.hip): one 1080p frame
of a synthetic scene -- a 96x96 height-field of diffuse triangles (18 432), 64 diffuse spheres, one
emissive sphere, a sun and a point light -- at ns_aa = 16 camera rays per pixel, counter jitter.
Prints the device time per frame and camera rays / s -- with the two delta lights, and with an
area light + the environment listed as sampled lights on top (ns_area_light = 4: 8 more shadow rays
per hit).  (The CPU side of this comparison is the checker's business: tests/ and bench.py.)
Usage (GPU box, repo root): python3 profiles/scene_term_timing.py"""
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
rng = np.random.default_rng(3)
N = 96
xs = np.linspace(-6, 6, N + 1)
zs = np.linspace(-14, -2, N + 1)
hgt = 0.35 * np.sin(xs[:, None] * 1.3) * np.cos(zs[None, :] * 0.9) - 1.5


def P(i, j):
    return [xs[i], hgt[i, j], zs[j]]


tris = []
up = [0.0, 1.0, 0.0]
for i in range(N):
    for j in range(N):
        a, b, c, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
        col = ("d", 0.3 + 0.5 * ((i + j) & 1), 0.6, 0.4)
        tris.append(tuple(a + c + b + up * 3) + col)
        tris.append(tuple(a + d + c + up * 3) + col)
spheres = [(float(rng.uniform(-5, 5)), float(rng.uniform(-1.0, 1.5)), float(rng.uniform(-13, -3)),
            float(rng.uniform(0.15, 0.5)), "d", *[float(v) for v in rng.uniform(0.2, 0.9, 3)])
           for _ in range(64)]
spheres.append((0.0, 3.0, -8.0, 0.6, "e", 5.0, 4.5, 3.0))
sun = np.array([0.3, 1.0, 0.4]); sun /= np.linalg.norm(sun)
lights = [[0.0, *sun, 1.0, 0.95, 0.8], [1.0, 2.0, 2.5, -5.0, 8.0, 8.0, 10.0]]
hf = 50.0
c2w = np.eye(3)
pos = [0.0, 0.5, 2.0]


def vfov(W, H):
    return 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))


lf = pkg.LensFlare(0)
W, H, ns = 1920, 1080, 16
lf.set_frame(W, H)
lf.set_params(ns, 25.0, 1.0)
lf.set_sampling(32, 0.05, 0.01, 100.0)
lf.set_camera(c2w, pos, hf, vfov(W, H))
lf.set_scene(spheres, tris, lights)
lf.set_jitter_counter(7)
lf.render_scene_term(); lf.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    lf.render_scene_term()
lf.synchronize()
gpu_ms = (time.perf_counter() - t0) / 3 * 1e3
out = {"frame": f"{W}x{H}, ns_aa {ns}", "primitives": len(tris) + len(spheres),
       "gpu_ms_per_frame": gpu_ms, "gpu_camera_rays_per_s": W * H * ns / (gpu_ms * 1e-3)}
# the same frame with sampled lights: an area light above the field and a small environment map
rows = [[0.0, 1.0, 0.95, 0.8, *sun] + [0.0] * 9,
        [1.0, 8.0, 8.0, 10.0, 2.0, 2.5, -5.0] + [0.0] * 9,
        [3.0, 6.0, 6.0, 6.0, 0.0, 4.0, -8.0, 0.0, -1.0, 0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 2.0],
        [4.0] + [0.0] * 15]
env = np.ones((16, 32, 3)) * 0.3
env[3:5, 20:23] = 25.0
lf.set_scene_lights(rows)
lf.set_light_samples(4)
lf.set_environment_map(env)
lf.render_scene_term(); lf.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    lf.render_scene_term()
lf.synchronize()
soft_ms = (time.perf_counter() - t0) / 3 * 1e3
out.update(gpu_ms_per_frame_sampled_lights=soft_ms, sampled_lights="area + environment, ns_area_light 4")
print(json.dumps(out))
lf.close()
