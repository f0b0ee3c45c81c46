#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes tests/golden/geo_f64_rays.json: a few hundred rays through the shipped
prescriptions as the independent float64 tracer (oracle/lf_geo_f64.c) marches them -- start state in,
exit state / fate / event count out, every number a hex float -- so that the float32 march is held to
the float64 one RAY BY RAY (tests/test_geo_rays_vs_f64.py), not only through converged pixels.

    python oracle/make_golden_f64_rays.py

The rays: random sensor point + pupil point pairs for the primary path and for ghost pairs of every
kind (both mirrors in front of the stop, behind it, one on each side, adjacent, far apart), at all
three wavelengths; plus deliberately awkward ones -- aimed at the rim of the rear element and of the
stop (vignetting decided by micrometres), steep ones that end in total internal reflection, rays along
the axis, rays through the thin lens.  The float64 tracer marks a ray FRAGILE when a decision fell
within its tolerance of the boundary; such rays are kept (the test then only asks that float32 agree
with one of the two fates)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def hx(v):
    return float(v).hex()


def main():
    import __graft_entry__ as g
    from oracle import lfo
    from goldenlib import load_texels
    pkg = g.load_package()
    rng = np.random.default_rng(20261004)
    out = {"note": "oracle/make_golden_f64_rays.py; start p (mm, z = the sensor plane), unit start direction d, "
                   "weight 1 -> float64 tracer's dead (0 alive, 1 stop / mask, 2 aperture / miss, 3 total reflection), "
                   "exit point, unit exit direction, weight, completed events, fragile flag",
           "cases": []}
    sets = [("dgauss11.lens", "pentbig500_14.png"), ("thinlens.lens", None)]
    for lens_name, mask_name in sets:
        lens = pkg.load_lens_file(lens_name)
        mask = load_texels(mask_name) if mask_name else np.ones((8, 8), np.float32)
        n, stop = int(lens["n"]), int(lens["stop"])
        zs = lfo.g64_sensor_z(lens)
        h_rear = float(lens["semi_aperture"][-1])
        z_rear = zs - float(lens["thickness"][-1])
        if stop >= 0:
            pairs = [(-1, -1), (0, 1), (0, 4), (1, 3), (2, 4), (3, 4), (6, 7), (7, 10), (6, 10), (9, 10),
                     (0, 6), (2, 8), (4, 6), (1, 10), (3, 9)]
        else:
            pairs = [(-1, -1), (0, 1)]
        rays = []

        def add(p, d, lam, ij, kind):
            d = np.asarray(d, float) / np.linalg.norm(d)
            st, pe, de, w, ne, frag, wpot = lfo.g64_trace_ray_ex(lens, lam, ij[0], ij[1], p, d, 1.0, mask)
            de = de / np.linalg.norm(de)
            rays.append({"kind": kind, "lam": lam, "ij": list(ij), "p": [hx(v) for v in p], "d": [hx(v) for v in d],
                         "dead": int(st), "events": int(ne), "fragile": int(frag),
                         "pe": [hx(v) for v in pe], "de": [hx(v) for v in de], "w": hx(w), "w_pot": hx(wpot)})

        # (a) random sensor point -> random point of the rear element's disc
        for k in range(150 if stop >= 0 else 40):
            X, Y = (rng.random(2) - 0.5) * [float(lens["sensor_width_mm"]), float(lens["sensor_width_mm"]) * 2 / 3]
            r, phi = h_rear * np.sqrt(rng.random()), 2 * np.pi * rng.random()
            q = np.array([r * np.cos(phi), r * np.sin(phi), z_rear])
            p = np.array([X, Y, zs])
            add(p, q - p, int(rng.integers(0, 3)), pairs[k % len(pairs)], "random")
        # (b) aimed at the rim of the rear element: vignetting decided within micrometres
        for k in range(24):
            phi = 2 * np.pi * k / 24
            for dr in (-2e-3, -2e-5, 2e-5, 2e-3):
                q = np.array([(h_rear + dr) * np.cos(phi), (h_rear + dr) * np.sin(phi), z_rear])
                p = np.array([0.3 * np.cos(phi + 1), 0.2 * np.sin(phi + 1), zs])
                add(p, q - p, 1, (-1, -1), "rear rim")
        # (c) the axis and near it
        for eps in (0.0, 1e-9, 1e-6, 1e-3):
            for ij in pairs[:4]:
                add(np.array([eps, -eps, zs]), [0.0, 0.0, -1.0], 1, ij, "axial")
        # (d) steep rays: large sensor offsets aimed across the pupil (total reflection, misses)
        for k in range(40):
            X = (18.0 + 14.0 * rng.random()) * (1 if k % 2 else -1)
            q = np.array([-np.sign(X) * h_rear * rng.random(), h_rear * (rng.random() - 0.5), z_rear])
            p = np.array([X, 3.0 * (rng.random() - 0.5), zs])
            add(p, q - p, int(rng.integers(0, 3)), pairs[k % len(pairs)], "steep")
        # (e) mirrors only: a ray that stays near the axis through every selected pair, all wavelengths
        for ij in pairs[1:]:
            for lam in range(3):
                add(np.array([0.4, -0.3, zs]), np.array([0.05, 0.02, z_rear]) - np.array([0.4, -0.3, zs]), lam, ij, "paraxial ghost")
        out["cases"].append({"lens": lens_name, "mask": mask_name, "rays": rays})
    path = os.path.join(ROOT, "tests", "golden", "geo_f64_rays.json")
    json.dump(out, open(path, "w"), indent=0, separators=(",", ":"))
    tot = sum(len(c["rays"]) for c in out["cases"])
    fates = {}
    for c in out["cases"]:
        for r in c["rays"]:
            fates[r["dead"]] = fates.get(r["dead"], 0) + 1
    print(f"{path}: {tot} rays, fates {fates}, fragile {sum(r['fragile'] > 0 for c in out['cases'] for r in c['rays'])}, "
          f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
