#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: four random scenes for the scene term, rendered by the REAL
reference (oracle/_ref/ref_dump) -- random spheres (some inside each other, some behind the camera,
some emissive), random triangles with random (non-unit, non-geometric) vertex normals incl. slivers,
a sun in the frame (the flare), a second directional light outside it and one or two point lights, a
randomly turned camera, ns_aa 2..5.  Delta lights only, so the reference's MT19937 stream is
reproducible and parity is exact (tests/test_gpu_scene_term.py, tests/test_oracle_vs_reference.py).
Only runs in the build container; the fixtures it writes (tests/golden/z*.npz) are committed."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

NAMES = ["z40x30_fuzz0", "z40x30_fuzz1", "z36x28_fuzz2", "z44x26_fuzz3"]
# flare-only frames (rows a1-a7): every branch of raytrace_starburst's shaping -- exponent 3 -
# flare_intensity incl. <= 0 (-> 2), = 3 (-> 0) and > 3 (negative), flare radii from half a pixel to
# beyond the frame, suns on the frame's edge and outside it, odd sizes, every aperture family
FLARE = [("q47x31_fuzz0", 47, 31, 2, 0.5, -1.0, "final_apertures/pent2_8.png", "bokeh/octagonbokeh.png",
          [((0.0, 0.5), (1.0, 0.5, 0.2), 10.0)]),
         ("q38x52_fuzz1", 38, 52, 1, 200.0, 0.0, "apertures/pentsmalllottalines.png", "final_apertures/pent_11.png",
          [((0.5, 1.0), (0.3, 0.6, 1.0), 20.0), ((0.25, 0.25), (2.0, 2.0, 2.0), 15.0)]),
         ("q61x33_fuzz2", 61, 33, 3, 8.0, 3.0, "final_apertures/pent3_18.png", "final_apertures/pent4_13.png",
          [((0.9, 0.1), (1.0, 1.0, 1.0), 12.0), ((1.2, 0.4), (5.0, 5.0, 5.0), 12.0)]),
         ("q33x33_fuzz3", 33, 33, 1, 3.0, 5.0, "final_apertures/pent4_15.png", "bokeh/octagonbokeh.png",
          [((0.49, 0.51), (0.7, 0.8, 0.9), 30.0)]),
         ("q52x40_fuzz4", 52, 40, 2, 60.0, 2.999, "final_apertures/pentbig2_500_9.png", "final_apertures/pent2_18.png",
          [((0.13, 0.87), (1.2, 1.1, 1.0), 18.0), ((0.8, 0.2), (0.2, 0.3, 0.4), 18.0),
           ((0.5, 0.5), (0.05, 0.05, 0.05), 18.0)]),
         ("q45x29_fuzz5", 45, 29, 1, 25.0, 0.3, "final_apertures/pentbig4_500_17.png", "final_apertures/pent4_17.png",
          [((0.998, 0.004), (1.0, 0.9, 0.5), 10.0)])]


def scene_for(rng, c2w, pos):
    R, p = np.asarray(c2w), np.asarray(pos, float)

    def w(v):  # camera space -> world
        return (R @ np.asarray(v, float) + p).tolist()

    def mat():
        if rng.random() < 0.2:
            return ["e"] + rng.uniform(0.2, 3.0, 3).tolist()
        return ["d"] + rng.uniform(0.05, 0.95, 3).tolist()

    spheres, tris = [], []
    for _ in range(int(rng.integers(3, 9))):
        c = [rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(-11, 1.5)]   # a few behind the camera
        spheres.append(tuple(w(c)) + (float(rng.uniform(0.15, 1.6)),) + tuple(mat()))
    for _ in range(int(rng.integers(4, 14))):
        a = np.array([rng.uniform(-4, 4), rng.uniform(-2.5, 2.5), rng.uniform(-12, -1.5)])
        size = 10 ** rng.uniform(-1.3, 0.6)
        b, c = a + rng.normal(0, size, 3), a + rng.normal(0, size, 3)
        if rng.random() < 0.25:      # a sliver
            c = a + (b - a) * rng.uniform(0.2, 0.8) + rng.normal(0, 0.01 * size, 3)
        n = []
        for _v in range(3):          # vertex normals: anything (the reference only interpolates them)
            n += (R @ rng.normal(0, 1, 3) * rng.uniform(0.3, 2.0)).tolist()
        tris.append(w(a) + w(b) + w(c) + n + mat())
    # a big floor so that shadows fall on something
    q = [w((-6, -2.2, -1)), w((6, -2.2, -1)), w((6, -2.6, -14)), w((-6, -2.6, -14))]
    up = (R @ np.array([0.0, 1.0, 0.0])).tolist()
    tris.append(q[0] + q[1] + q[2] + up * 3 + ["d", 0.6, 0.6, 0.55])
    tris.append(q[0] + q[2] + q[3] + up * 3 + ["d", 0.6, 0.6, 0.55])
    points = [tuple(w((rng.uniform(-3, 3), rng.uniform(0.5, 3.5), rng.uniform(-9, -1)))) +
              tuple(rng.uniform(1.0, 8.0, 3).tolist()) for _ in range(int(rng.integers(1, 3)))]
    return dict(spheres=spheres, tris=tris, points=points)


def main():
    tmp = tempfile.mkdtemp(prefix="lffuzz")
    for k, name in enumerate(NAMES):
        rng = np.random.default_rng(9100 + k)
        W, H = int(name[1:].split("_")[0].split("x")[0]), int(name.split("_")[0].split("x")[1])
        cam = (float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.3, 0.3)),
               tuple(rng.uniform(-1.5, 1.5, 3).tolist()))
        c2w = mg.rot(cam[0], cam[1])
        hf, vf = mg.fit_fov(50.0, 35.0, W, H)
        # a second sun outside the frame: world-space posLight + radiance as given (raw_lights)
        far = (np.asarray(c2w) @ np.array([rng.uniform(-1, 1), rng.uniform(1.5, 3), rng.uniform(0.5, 2)]) * 30.0
               + np.asarray(cam[2])).tolist()
        c = dict(name=name, W=W, H=H, ns_aa=int(rng.integers(2, 6)), radius=25.0, intensity=1.0,
                 ap="apertures/pentsmall.png", gh="bokeh/octagonbokeh.png",
                 lights=[((float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.2, 0.8))),
                          tuple(rng.uniform(0.4, 1.5, 3).tolist()), 30.0)],
                 raw_lights=[far + rng.uniform(0.2, 0.9, 3).tolist()],
                 cam=cam, visit="tiles", scene=scene_for(rng, c2w, cam[2]))
        mg.run_case(c, tmp)
    for k, (name, W, H, ns_aa, radius, intensity, ap, gh, lights) in enumerate(FLARE):
        rng = np.random.default_rng(9200 + k)
        cam = (float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-0.4, 0.4)), tuple(rng.uniform(-3, 3, 3).tolist()))
        mg.run_case(dict(name=name, W=W, H=H, ns_aa=ns_aa, radius=radius, intensity=intensity, ap=ap, gh=gh,
                         lights=lights, cam=cam, visit="tiles"), tmp)


if __name__ == "__main__":
    main()
