"""The float32 march RAY BY RAY against the independent float64 tracer: the
committed fixture tests/golden/geo_f64_rays.json (oracle/make_golden_f64_rays.py) holds 531 rays through
the double Gauss and the thin lens -- primary path and ghost pairs of every kind, all wavelengths, random
and deliberately awkward (rims, the axis, steep rays that end in total reflection) -- with the float64
tracer's fate, completed events, exit point, exit direction and weight as hex floats.  The float32
oracle (oracle/lf_geo_oracle.c: the recipe the device follows bit for bit) must
reproduce every one of them: the same fate and event count (unless the float64 tracer itself flagged
the ray as within rounding distance of a decision), exit state and weight within float32 accuracy.
CPU only: that the DEVICE equals this float32 oracle bit for bit is what tests/test_gpu_march_parity.py
and tests/test_gpu_lens_camera.py assert."""
import json
import os

import numpy as np

from goldenlib import GOLD, load_texels
from oracle import lfo

FIX = json.load(open(os.path.join(GOLD, "geo_f64_rays.json")))
POS_TOL_MM, DIR_TOL, W_TOL = 5e-5, 6e-6, 5e-5      # measured: 7.6e-6 mm, 1.2e-6, 1.0e-5 (printed below)


def _pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_every_fixture_ray():
    pkg = _pkg()
    worst = dict(pos=0.0, dir=0.0, w=0.0)
    n_alive = n_checked = n_fragile = 0
    fates = {}
    for case in FIX["cases"]:
        lens = pkg.load_lens_file(case["lens"])
        mask = load_texels(case["mask"]) if case["mask"] else np.ones((8, 8), np.float32)
        for r in case["rays"]:
            p = [float.fromhex(v) for v in r["p"]]
            d = [float.fromhex(v) for v in r["d"]]
            st, pe, de, w, ne = lfo.geo_trace_ray(lens, r["lam"], r["ij"][0], r["ij"][1], p, d, 1.0, mask)
            if r["fragile"]:
                n_fragile += 1      # float64 itself called it undecidable at float32 accuracy: either fate
                continue
            n_checked += 1
            fates[r["dead"]] = fates.get(r["dead"], 0) + 1
            assert st == r["dead"], (case["lens"], r["kind"], r["ij"], st, r["dead"])
            assert ne == r["events"], (case["lens"], r["kind"], r["ij"], ne, r["events"])
            if st != 0:
                continue
            n_alive += 1
            pe64 = np.array([float.fromhex(v) for v in r["pe"]])
            de64 = np.array([float.fromhex(v) for v in r["de"]])
            w64 = float.fromhex(r["w"])
            de = np.asarray(de, np.float64)
            de = de / np.linalg.norm(de)
            e_pos, e_dir = np.abs(np.asarray(pe, np.float64) - pe64).max(), np.abs(de - de64).max()
            e_w = abs(float(w) - w64) / w64
            worst = dict(pos=max(worst["pos"], e_pos), dir=max(worst["dir"], e_dir), w=max(worst["w"], e_w))
            assert e_pos <= POS_TOL_MM and e_dir <= DIR_TOL and e_w <= W_TOL, (case["lens"], r["kind"], r["ij"], e_pos, e_dir, e_w)
    print(f"{n_checked} rays checked ({n_fragile} fragile skipped), fates {fates}; {n_alive} alive: worst exit point "
          f"{worst['pos']:.2e} mm, direction {worst['dir']:.2e}, weight {worst['w']:.2e} relative")
    assert n_checked > 500 and n_alive > 120
    assert set(fates) == {0, 1, 2, 3}      # every fate occurs: alive, stop / mask, aperture, total reflection


def test_fixture_is_what_the_tracer_says_today():
    """the float64 tracer still produces the committed numbers (the fixture is not stale)"""
    pkg = _pkg()
    for case in FIX["cases"]:
        lens = pkg.load_lens_file(case["lens"])
        mask = load_texels(case["mask"]) if case["mask"] else np.ones((8, 8), np.float32)
        for r in case["rays"][::7]:
            p = [float.fromhex(v) for v in r["p"]]
            d = [float.fromhex(v) for v in r["d"]]
            st, pe, de, w, ne, frag, wpot = lfo.g64_trace_ray_ex(lens, r["lam"], r["ij"][0], r["ij"][1], p, d, 1.0, mask)
            assert (st, ne, frag) == (r["dead"], r["events"], r["fragile"])
            assert float(w).hex() == r["w"] and [float(v).hex() for v in pe] == r["pe"]
