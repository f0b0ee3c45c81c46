"""Known-answer tests that pin the INDEPENDENT float64 tracer (oracle/lf_geo_f64.c), the second
opinion the GPU march is checked against at the north star's bar (tests/test_gpu_march_f64.py).
Same anchors as tests/test_geo_oracle_kat.py -- closed-form optics, the published Philox vectors,
the small-angle agreement with the reference's own paraxial T/R/L formalism
(pathtracer.cpp:527-537, :588-689) for all 13 reference pairs and 3 colours -- plus a CPU
cross-check against the float32 oracle, with which it shares no code.  CPU only."""
import math

import numpy as np
import pytest

from oracle import lfo
from test_geo_oracle_kat import (PAIRS, _paraxial_matrix, _physical_ghost_matrix, _pkg,
                                 _reference_table_as_geometric_lens)


def _one_surface(radius, n_behind, semi_ap=1e3, stop=-1):
    """A single interface at z = 0 with air in front, as a lens dict."""
    return dict(n=1, stop=stop, radius=np.array([radius], np.float32),
                thickness=np.array([10.0], np.float32), ior=np.array([[n_behind]], np.float32),
                semi_aperture=np.array([semi_ap], np.float32), sensor_width_mm=36.0)


def test_philox_known_answers():
    assert lfo.g64_philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert lfo.g64_philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert lfo.g64_philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                          [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("theta_deg", [0.0, 5.0, 30.0, 60.0, 80.0])
@pytest.mark.parametrize("n2", [1.5, 1.7, 1.33])
def test_snell_and_fresnel_flat_interface(theta_deg, n2):
    """Flat interface, both directions of travel: Snell's law, T = 1 - R, mirror direction."""
    L = lfo.g64_lens(_one_surface(0.0, n2))
    t1 = math.radians(theta_deg)
    for into_glass in (True, False):
        n_in, n_out = (1.0, n2) if into_glass else (float(np.float32(n2)), 1.0)
        if not into_glass:
            n_out, n_in = 1.0, float(np.float32(n2))
        z0, dz = (-1.0, 1.0) if into_glass else (1.0, -1.0)
        d = [math.sin(t1), 0.0, dz * math.cos(t1)]
        s2 = n_in / (float(np.float32(n2)) if into_glass else 1.0) * math.sin(t1)
        st, p, dd, w = lfo.g64_glass_event(L, 0, 0, 0, [0, 0, z0], d)
        if s2 >= 1.0:
            assert st == 3
            st, p, dd, w = lfo.g64_glass_event(L, 0, 0, 1, [0, 0, z0], d)
            assert st == 0 and w == pytest.approx(1.0, abs=1e-12)
            continue
        assert st == 0
        t2 = math.asin(s2)
        assert dd[0] == pytest.approx(math.sin(t2), abs=1e-12)
        assert dd[2] == pytest.approx(dz * math.cos(t2), abs=1e-12)
        assert p[2] == pytest.approx(0.0, abs=1e-12) and p[0] == pytest.approx(math.tan(t1), rel=1e-12)
        na, nb = (1.0, float(np.float32(n2))) if into_glass else (float(np.float32(n2)), 1.0)
        ci, ct = math.cos(t1), math.cos(t2)
        rs = (na * ci - nb * ct) / (na * ci + nb * ct)
        rp = (nb * ci - na * ct) / (nb * ci + na * ct)
        R = 0.5 * (rs * rs + rp * rp)
        assert w == pytest.approx(1.0 - R, rel=1e-12)
        st, p, dr, wr = lfo.g64_glass_event(L, 0, 0, 1, [0, 0, z0], d)
        assert st == 0 and wr == pytest.approx(R, rel=1e-10, abs=1e-15)
        assert dr[0] == pytest.approx(d[0], abs=1e-12) and dr[2] == pytest.approx(-d[2], abs=1e-12)


def test_fresnel_normal_incidence_and_brewster():
    n2 = float(np.float32(1.5168))
    L = lfo.g64_lens(_one_surface(0.0, 1.5168))
    st, _, _, w = lfo.g64_glass_event(L, 0, 0, 1, [0, 0, -1.0], [0, 0, 1.0])
    assert w == pytest.approx(((1 - n2) / (1 + n2)) ** 2, rel=1e-12)
    tb = math.atan(n2)
    st, _, _, w = lfo.g64_glass_event(L, 0, 0, 1, [0, 0, -1.0], [math.sin(tb), 0.0, math.cos(tb)])
    t2 = math.asin(math.sin(tb) / n2)
    rs = (math.cos(tb) - n2 * math.cos(t2)) / (math.cos(tb) + n2 * math.cos(t2))
    assert w == pytest.approx(0.5 * rs * rs, rel=1e-9)   # r_p = 0 at Brewster's angle


@pytest.mark.parametrize("R", [50.0, -80.0, 12.75])
def test_sphere_intersection_sag_and_vignetting(R):
    h = 5.0
    sag = R - math.copysign(math.sqrt(R * R - h * h), R)
    L = lfo.g64_lens(_one_surface(R, 1.0))          # same medium on both sides: the ray goes straight
    for z0, dz in ((-20.0, 1.0), (20.0, -1.0)):
        st, p, d, _ = lfo.g64_glass_event(L, 0, 0, 0, [h, 0, z0], [0, 0, dz])
        assert st == 0 and p[0] == pytest.approx(h, abs=1e-12) and p[2] == pytest.approx(sag, abs=1e-11)
        assert d[2] == pytest.approx(dz, abs=1e-12)
    Ls = lfo.g64_lens(_one_surface(R, 1.0, semi_ap=4.9))
    assert lfo.g64_glass_event(Ls, 0, 0, 0, [h, 0, -20.0], [0, 0, 1.0])[0] == 2   # outside the clear aperture
    assert lfo.g64_glass_event(L, 0, 0, 0, [abs(R) * 1.5, 0, -20.0], [0, 0, 1.0])[0] == 2  # misses the sphere


def test_thin_lens_focal_length():
    lens = _pkg().load_lens_file("thinlens.lens")
    zs = lfo.g64_sensor_z(lens)
    assert zs == pytest.approx(5.0 + 47.54, abs=1e-4)
    for h in (0.05, 0.2, 0.5):
        d = np.array([h, 0.0, -47.54])
        d /= np.linalg.norm(d)
        st, p, dd, w, ne = lfo.g64_trace_ray(lens, 1, -1, -1, [0, 0, zs], d)
        assert st == 0 and ne == 2
        assert abs(dd[0] / dd[2]) < 2e-4 + 3e-3 * h ** 3
        assert 0.90 < w < 0.93


@pytest.mark.parametrize("kind,i,j", PAIRS)
def test_small_angle_limit_matches_paraxial_ghosts(kind, i, j):
    """The same anchor on the reference as the float32 oracle's: a near-axis ray marched backwards
    through ghost pair (i, j) of the reference's own table is mapped back onto the sensor ray by the
    forward paraxial ghost matrix (the reference's own pinned tracer for adjacent pairs)."""
    lens, L = _reference_table_as_geometric_lens()
    zs = lfo.g64_sensor_z(lens)
    for colour in range(3):
        M = _paraxial_matrix(L, kind, i, j, colour) if j == i + 1 else \
            _physical_ghost_matrix(L, i, j, colour)
        for ys, us in ((0.02, 1e-4), (-0.01, 3e-4), (0.0, -2e-4)):
            d = np.array([us, 0.0, -1.0])
            d /= np.linalg.norm(d)
            st, p, dd, w, ne = lfo.g64_trace_ray(lens, colour, i, j, [ys, 0, zs], d)
            assert st == 0 and ne == 9 + 2 * (j - i)
            y_in, u_in = p[0] - dd[0] / dd[2] * p[2], dd[0] / dd[2]
            y_s, u_s = M @ np.array([y_in, u_in])
            slope_s = d[0] / d[2]
            scale_y = max(abs(ys), abs(y_in), 1e-3)
            assert y_s == pytest.approx(ys, abs=2e-3 * scale_y + 2e-5)
            assert u_s == pytest.approx(slope_s, abs=2e-3 * max(abs(slope_s), abs(u_in)) + 2e-6)


@pytest.mark.parametrize("pair", [(-1, -1), (0, 1), (2, 7), (6, 9), (0, 10)])
def test_single_rays_agree_with_the_float32_oracle(pair):
    """Two implementations with nothing in common (float32 vertex-form recipe in optical direction
    cosines vs float64 textbook formulation) follow the same rays: positions to 1e-4 mm, directions to
    1e-5, weights to 5e-5 for EVERY ray (a ray close to the critical angle amplifies float32's 1e-7: the
    worst of 1 060 rays is 2.5e-5) and to 2e-6 on average."""
    lens = _pkg().load_lens_file("dgauss11.lens")
    zs32, zs64 = lfo.geo_z_sensor(lens), lfo.g64_sensor_z(lens)
    assert zs32 == pytest.approx(zs64, abs=1e-4)
    rng = np.random.default_rng(11)
    checked, w_err = 0, []
    for _ in range(400):
        p0 = [rng.uniform(-3, 3), rng.uniform(-2, 2)]
        tgt = rng.uniform(-6, 6, 2)
        d = np.array([tgt[0] - p0[0], tgt[1] - p0[1], -36.106])
        d /= np.linalg.norm(d)
        d32 = d.astype(np.float32)
        d32 /= np.float32(np.linalg.norm(d32.astype(np.float64)))
        lam = int(rng.integers(0, 3))
        s32, q32, e32, w32, n32 = lfo.geo_trace_ray(lens, lam, pair[0], pair[1], [p0[0], p0[1], zs32], d32)
        s64, q64, e64, w64, n64 = lfo.g64_trace_ray(lens, lam, pair[0], pair[1], [p0[0], p0[1], zs64],
                                                    d32.astype(np.float64))
        if s32 != 0 or s64 != 0:
            assert n32 == n64 or abs(n32 - n64) <= 1   # the same fate (up to an edge ray)
            continue
        checked += 1
        assert np.allclose(q32, q64, atol=1e-4) and np.allclose(e32, e64, atol=1e-5)
        assert w32 == pytest.approx(w64, rel=5e-5)
        w_err.append(abs(w32 - w64) / w64)
    assert checked > 20 and np.mean(w_err) < 2e-6


def test_small_frame_agrees_with_the_float32_oracle():
    """A whole (small) frame: float32 oracle vs float64 tracer within 1e-4 relative + the weight of
    the fragile rays, every pixel; counters within the number of fragile rays."""
    pkg = _pkg()
    lens = pkg.load_lens_file("dgauss11.lens")
    from goldenlib import load_texels
    mask = load_texels("pentbig500_14.png")
    W, H, spp, key = 24, 16, 32, 0xBEEF
    sun = dict(sun_dir=[0.03, 0.02, -1.0], sun_radiance=[1.0, 0.9, 0.5], sun_angular_radius=0.05)
    img32, c32 = lfo.geo_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun["sun_dir"],
                               sun["sun_radiance"], sun["sun_angular_radius"])
    img64, frag, c64 = lfo.g64_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun["sun_dir"],
                                     sun["sun_radiance"], sun["sun_angular_radius"])
    assert img64.max() > 0 and c64["rays_launched"] == c32["rays_launched"]
    assert np.all(np.abs(img32 - img64) <= 1e-4 * img64 + 1.05 * frag + 1e-9)
    clean = frag.sum(axis=2) == 0
    assert clean.mean() > 0.5            # most pixels carry no fragile ray: the strict bar applies
    lit = clean & (img64.sum(axis=2) > 1e-6)
    assert lit.sum() > 20
    assert (np.abs(img32 - img64)[lit] / np.maximum(img64[lit], 1e-6)).max() < 1e-4
    n_frag = c64["rays_fragile"]
    assert n_frag < 1e-2 * c64["rays_launched"]
    for name in ("rays_clipped_stop", "rays_vignetted", "rays_tir", "rays_reached_scene", "rays_hit_light"):
        assert abs(c64[name] - c32[name]) <= n_frag, name
    assert abs(c64["surface_events"] - c32["surface_events"]) <= 30 * n_frag
