"""Analytic known-answer tests of the geometric march THROUGH the device entry point
(lf_generate_lens_rays = LensCamera::generate_ray, batched), independent of both CPU oracles: closed
forms of Fresnel / Snell / the lensmaker's equation evaluated in float64 here (SURVEY 8c i-iv).
The formalism is the reference's own where it has one (T / R operators, pathtracer.cpp:527-537); the
geometric march has no reference implementation (camera_lens.cpp:22-30 is a stub)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def lf(pkg):
    ctx = pkg.LensFlare(0)
    ctx.set_frame(64, 64)
    ctx.set_aperture(pkg.APERTURE_STARBURST, np.ones((4, 4), np.float32))   # an open stop
    yield ctx
    ctx.close()


def _geom_norm(lens):
    """start weight of the axial ray: pi h_N^2 / (z_sensor - z_N)^2"""
    return math.pi * float(lens["semi_aperture"][-1]) ** 2 / float(lens["thickness"][-1]) ** 2


def _slab(n, t=20.0, h=500.0):
    """ONE flat interface: air on the scene side, glass of index n from the interface to the sensor."""
    return dict(n=1, stop=-1, radius=np.zeros(1, np.float32), thickness=np.array([t], np.float32),
                ior=np.array([[n]], np.float32), semi_aperture=np.array([h], np.float32), sensor_width_mm=36.0)


def test_normal_incidence_transmission_is_the_product_of_fresnel_factors(pkg, lf):
    """axial ray through all 10 glass interfaces of the double-Gauss: T = prod 1 - ((n1-n2)/(n1+n2))^2"""
    lens = pkg.load_lens_file("dgauss11.lens")
    lf.set_lens(lens)
    for lam in range(3):
        out = lf.generate_lens_rays(lam, [[0.0, 0.0]], [[0.0, 0.0]])[0]
        assert out[7] == 1.0
        assert np.allclose(out[0:2], 0, atol=1e-6) and np.allclose(out[3:6], [0, 0, -1], atol=1e-6)
        T, n_before = 1.0, 1.0
        for k in range(lens["n"]):
            if k == lens["stop"]:
                continue
            n_after = float(lens["ior"][lam, k])
            T *= 1.0 - ((n_before - n_after) / (n_before + n_after)) ** 2
            n_before = n_after
        assert out[6] / _geom_norm(lens) == pytest.approx(T, rel=2e-6)
        assert 0.5 < T < 0.9      # ten uncoated interfaces lose a third of the light


@pytest.mark.parametrize("n", [1.5, 1.7, 2.4])
def test_brewster_angle_and_unpolarised_fresnel(pkg, lf, n):
    """glass -> air at Brewster's angle tan(theta_B) = 1/n: R_p = 0, so the unpolarised transmission
    is 1 - R_s / 2, and the refracted ray is perpendicular to the (would-be) reflected one."""
    t = 20.0
    lf.set_lens(_slab(n, t))
    th_i = math.atan(1.0 / n)
    X = t * math.tan(th_i)                       # the ray is aimed at the rear vertex = the interface's centre
    out = lf.generate_lens_rays(0, [[X, 0.0]], [[0.0, 0.0]])[0]
    assert out[7] == 1.0
    th_t = math.asin(n * math.sin(th_i))
    assert th_i + th_t == pytest.approx(math.pi / 2, abs=1e-12)      # Brewster
    # direction after the interface (travelling -z, x decreasing): Snell
    assert out[3] == pytest.approx(-math.sin(th_t), abs=2e-6) and out[5] == pytest.approx(-math.cos(th_t), abs=2e-6)
    ci, ct = math.cos(th_i), math.cos(th_t)
    rs = (n * ci - ct) / (n * ci + ct)
    rp = (ci - n * ct) / (ci + n * ct)
    assert abs(rp) < 1e-12
    start = _geom_norm(_slab(n, t)) * ci ** 4       # pupil area / distance^2 * cos^4
    assert out[6] / start == pytest.approx(1.0 - 0.5 * rs * rs, rel=5e-6)
    # ... and a general angle: T = 1 - (R_s + R_p) / 2
    for th in (0.1, 0.25, 0.9 * math.asin(1.0 / n)):
        o = lf.generate_lens_rays(0, [[t * math.tan(th), 0.0]], [[0.0, 0.0]])[0]
        tt = math.asin(n * math.sin(th))
        a, b = math.cos(th), math.cos(tt)
        Rs, Rp = ((n * a - b) / (n * a + b)) ** 2, ((a - n * b) / (a + n * b)) ** 2
        assert o[7] == 1.0
        assert o[6] / (_geom_norm(_slab(n, t)) * a ** 4) == pytest.approx(1.0 - 0.5 * (Rs + Rp), rel=2e-5)


def test_total_internal_reflection_kills_the_ray(pkg, lf):
    n, t = 1.5, 20.0
    lf.set_lens(_slab(n, t))
    crit = math.asin(1.0 / n)
    inside = lf.generate_lens_rays(0, [[t * math.tan(crit - 1e-3), 0.0]], [[0.0, 0.0]])[0]
    beyond = lf.generate_lens_rays(0, [[t * math.tan(crit + 1e-3), 0.0]], [[0.0, 0.0]])[0]
    assert inside[7] == 1.0 and inside[6] > 0
    assert beyond[7] == 0.0 and beyond[6] == 0.0
    # grazing exit just inside the critical angle
    assert inside[5] > -0.2 and inside[3] < -0.97


def test_thin_lens_focuses_where_the_lensmakers_equation_says(pkg, lf):
    """1/f = (n-1)(1/R1 - 1/R2 + (n-1) t / (n R1 R2)); the file puts the sensor at the paraxial focus,
    so rays from the axial sensor point leave collimated, and a sensor point at height X leaves at
    angle X / f (through the nodal point)."""
    lens = pkg.load_lens_file("thinlens.lens")
    lf.set_lens(lens)
    n = float(lens["ior"][1, 0])
    R1, R2, d = float(lens["radius"][0]), float(lens["radius"][1]), float(lens["thickness"][0])
    f = 1.0 / ((n - 1) * (1 / R1 - 1 / R2 + (n - 1) * d / (n * R1 * R2)))
    assert pkg.paraxial_efl(lens, 1) == pytest.approx(f, rel=1e-6)
    uv = np.array([[0.02, 0.0], [0.0, 0.03], [-0.03, 0.02], [0.05, 0.05]], np.float32)   # paraxial pupil heights
    out = lf.generate_lens_rays(1, np.zeros((4, 2), np.float32), uv)
    assert np.all(out[:, 7] == 1.0)
    assert np.all(np.abs(out[:, 3:5]) < 2e-5)                  # collimated (spherical aberration ~ h^3)
    for X in (0.2, -0.5, 1.0):
        o = lf.generate_lens_rays(1, [[X, 0.0]], [[0.0, 0.0]])[0]
        assert o[7] == 1.0
        # image point at +X: the ray leaves towards -X ... the scene direction is X / f off the axis
        assert o[3] / -o[5] == pytest.approx(-X / f, rel=3e-3)
    # marginal rays at full aperture focus SHORT of the paraxial focus (under-corrected singlet): from
    # the paraxial focus they leave converging towards the axis
    o = lf.generate_lens_rays(1, [[0.0, 0.0]], [[0.9, 0.0]])[0]
    assert o[7] == 1.0 and o[0] > 0 and o[3] < -1e-4
