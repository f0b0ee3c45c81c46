#!/usr/bin/env python3
"""A/B harness for the scene-radiance kernel (lf_scene.hip): renders fixed frames
through the C ABI, times lf_render_scene_term with the library's own HIP events and writes the scene
buffers to <out>.npz, so that two builds of the library (LF_LIB=<other .so>) can be compared pixel by
pixel:
    python3 profiles/scene_ab.py /tmp/scene_new.npz
    LF_LIB=lens-flare_amd/build_ab/liblensflare_old.so python3 profiles/scene_ab.py /tmp/scene_old.npz
    python3 profiles/scene_ab.py --compare /tmp/scene_old.npz /tmp/scene_new.npz
(the .npz files are ~50 MB: keep them out of gpurun_out/)
Frames: the 18.5 k-primitive height field of profiles/scene_term_timing.py with delta lights (1080p,
ns_aa 16) and with sampled lights on top (area + environment, ns_area_light 4), and maxplanck.dae
(50 801 triangles) seen from its lit side at 1080p, ns_aa 4."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def compare(a_path, b_path):
    a, b = np.load(a_path), np.load(b_path)
    rep = {}
    for k in a.files:
        if k.endswith("_ms"):
            rep[k] = {"a": float(a[k]), "b": float(b[k]), "b_over_a": float(b[k] / a[k])}
            continue
        x, y = a[k], b[k]
        same = (x == y) | (np.isnan(x) & np.isnan(y))
        px = ~same.all(axis=-1)
        d = np.abs(x - y)
        rep[k] = {"pixels": int(px.size), "pixels_differing": int(px.sum()),
                  "max_abs_diff": float(np.nanmax(d)), "max_value": float(np.nanmax(np.abs(x)))}
    print(json.dumps(rep, indent=1))


def vfov(hf, W, H):
    return 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))


def timed(lf, n=3):
    lf.render_scene_term(); lf.synchronize()
    lf.timing_reset()
    for _ in range(n):
        lf.render_scene_term()
    lf.synchronize()
    k, ms = lf.timing_get("scene_term")
    return ms / max(k, 1)


def main(out):
    import __graft_entry__ as g
    pkg = g.load_package()
    rng = np.random.default_rng(3)
    N = 96
    xs = np.linspace(-6, 6, N + 1)
    zs = np.linspace(-14, -2, N + 1)
    hgt = 0.35 * np.sin(xs[:, None] * 1.3) * np.cos(zs[None, :] * 0.9) - 1.5
    P = lambda i, j: [xs[i], hgt[i, j], zs[j]]
    tris, up = [], [0.0, 1.0, 0.0]
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
    res = {}
    lf = pkg.LensFlare(0)
    lf.timing_enable(True)
    W, H, hf = 1920, 1080, 50.0
    lf.set_frame(W, H)
    lf.set_params(16, 25.0, 1.0)
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.set_camera(np.eye(3), [0.0, 0.5, 2.0], hf, vfov(hf, W, H))
    lf.set_scene(spheres, tris, lights)
    lf.set_jitter_counter(7)
    res["field_delta_ms"] = timed(lf)
    res["field_delta"] = lf.read_buffer(pkg.SCENE_BUFFER)
    rows = [[0.0, 1.0, 0.95, 0.8, *sun] + [0.0] * 9,
            [1.0, 8.0, 8.0, 10.0, 2.0, 2.5, -5.0] + [0.0] * 9,
            [3.0, 6.0, 6.0, 6.0, 0.0, 4.0, -8.0, 0.0, -1.0, 0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 2.0],
            [4.0] + [0.0] * 15]
    env = np.ones((16, 32, 3)) * 0.3
    env[3:5, 20:23] = 25.0
    lf.set_scene_lights(rows)
    lf.set_light_samples(4)
    lf.set_environment_map(env)
    res["field_soft_ms"] = timed(lf)
    res["field_soft"] = lf.read_buffer(pkg.SCENE_BUFFER)
    lf.set_direct_hemisphere_sample(True)
    res["field_hemi_ms"] = timed(lf, 1)
    res["field_hemi"] = lf.read_buffer(pkg.SCENE_BUFFER)
    lf.close()

    lf = pkg.LensFlare(0)
    lf.timing_enable(True)
    lf.set_frame(W, H)
    lf.set_params(4, 25.0, 1.0)
    lf.set_sampling(32, 0.05, 0.01, 1000.0)
    camera, suns = lf.load_collada(os.path.join(pkg.DATA, "maxplanck.dae"))
    lo, hi, _ = lf.scene_bounds()
    mid, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    to_sun = np.array(suns[0][:3], float) - mid
    side = np.cross(to_sun, [0.0, 1.0, 0.0])
    view = to_sun / np.linalg.norm(to_sun) + 0.6 * side / np.linalg.norm(side)   # the lit side, obliquely
    pos = mid + view / np.linalg.norm(view) * (0.9 * ext)
    hf2 = 40.0
    c2w = pkg.aim_camera(pos, mid, (0.5, 0.5), hf2, vfov(hf2, W, H))
    lf.set_camera(c2w, pos, hf2, vfov(hf2, W, H))
    lf.set_jitter_counter(11)
    res["maxplanck_ms"] = timed(lf)
    res["maxplanck"] = lf.read_buffer(pkg.SCENE_BUFFER)
    lf.close()
    np.savez_compressed(out, **res)
    print(json.dumps({k: (float(v) if k.endswith("_ms") else [float(np.nanmean(v)), float((v.max(axis=-1) > 0).mean())])
                      for k, v in res.items()}))


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1])
