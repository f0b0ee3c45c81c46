"""Known-answer tests that pin the GEOMETRIC march oracle (oracle/lf_geo_oracle.c).  The reference
has no geometric lens, so these are the anchors: closed-form optics, the
published Philox test vectors, and -- the strongest one -- agreement of the geometric ghost march
with the reference's OWN paraxial T/R/L matrix formalism (pathtracer.cpp:527-537, :588-689, pinned
bit-exactly by test_oracle_vs_reference.py) in the small-angle limit, for all 13 reference pairs
and 3 colours.  CPU only."""
import math

import numpy as np
import pytest

from oracle import lfo

BIG_H2 = 1e6


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    assert lfo.geo_philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert lfo.geo_philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert lfo.geo_philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                          [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("theta_deg", [0.0, 5.0, 30.0, 60.0, 80.0])
@pytest.mark.parametrize("n1,n2", [(1.0, 1.5), (1.5, 1.0), (1.0, 1.7), (1.33, 1.6)])
def test_snell_and_fresnel_flat_interface(theta_deg, n1, n2):
    """Flat interface: sin(t2) = (n1/n2) sin(t1); T = 1 - R with the unpolarised Fresnel R."""
    t1 = math.radians(theta_deg)
    eta = n1 / n2
    s2 = eta * math.sin(t1)
    d = [math.sin(t1), 0.0, math.cos(t1)]
    st, p, dd, w = lfo.geo_glass_event([0, 0, -1.0], d, 1.0, 0.0, 0.0, BIG_H2, eta, 0, 1)
    if s2 >= 1.0:
        assert st == 3  # total internal reflection
        st, p, dd, w = lfo.geo_glass_event([0, 0, -1.0], d, 1.0, 0.0, 0.0, BIG_H2, eta, 1, 1)
        assert st == 0 and w == pytest.approx(1.0, abs=1e-6)
        return
    assert st == 0
    t2 = math.asin(s2)
    assert dd[0] == pytest.approx(math.sin(t2), abs=2e-6)
    assert dd[2] == pytest.approx(math.cos(t2), abs=2e-6)
    assert p[2] == pytest.approx(0.0, abs=1e-6) and p[0] == pytest.approx(math.tan(t1), rel=1e-5)
    ci, ct = math.cos(t1), math.cos(t2)
    rs = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct)
    rp = (n2 * ci - n1 * ct) / (n2 * ci + n1 * ct)
    R = 0.5 * (rs * rs + rp * rp)
    assert w == pytest.approx(1.0 - R, rel=3e-6)
    st, p, dr, wr = lfo.geo_glass_event([0, 0, -1.0], d, 1.0, 0.0, 0.0, BIG_H2, eta, 1, 1)
    assert st == 0 and wr == pytest.approx(R, rel=2e-5, abs=1e-9)
    assert dr[0] == pytest.approx(d[0], abs=1e-6) and dr[2] == pytest.approx(-d[2], abs=1e-6)


def test_fresnel_normal_incidence_and_brewster():
    n1, n2 = 1.0, 1.5168
    st, _, _, w = lfo.geo_glass_event([0, 0, -1.0], [0, 0, 1.0], 1.0, 0.0, 0.0, BIG_H2, n1 / n2, 1, 1)
    assert w == pytest.approx(((n1 - n2) / (n1 + n2)) ** 2, rel=1e-5)
    tb = math.atan(n2 / n1)  # Brewster: rp = 0 -> R = rs^2 / 2
    d = [math.sin(tb), 0.0, math.cos(tb)]
    st, _, _, w = lfo.geo_glass_event([0, 0, -1.0], d, 1.0, 0.0, 0.0, BIG_H2, n1 / n2, 1, 1)
    t2 = math.asin(math.sin(tb) * n1 / n2)
    rs = (n1 * math.cos(tb) - n2 * math.cos(t2)) / (n1 * math.cos(tb) + n2 * math.cos(t2))
    assert w == pytest.approx(0.5 * rs * rs, rel=1e-4)


@pytest.mark.parametrize("R", [50.0, -80.0, 12.75])
def test_sphere_intersection_sag_and_vignetting(R):
    """A ray parallel to the axis at height h meets the sphere at the sag z = R - sign(R) sqrt(R^2-h^2);
    outside the clear aperture it is vignetted; backwards-travelling rays use the other root sign."""
    h = 5.0
    sag = R - math.copysign(math.sqrt(R * R - h * h), R)
    for fwd, z0, dz in ((1, -20.0, 1.0), (0, 20.0, -1.0)):
        st, p, _, _ = lfo.geo_glass_event([h, 0, z0], [0, 0, dz], 1.0, 0.0, 1.0 / R, BIG_H2, 1.0, 0, fwd)
        assert st == 0
        assert p[0] == pytest.approx(h, abs=1e-6) and p[2] == pytest.approx(sag, abs=2e-5)
    st, _, _, _ = lfo.geo_glass_event([h, 0, -20.0], [0, 0, 1.0], 1.0, 0.0, 1.0 / R, 4.9 ** 2, 1.0, 0, 1)
    assert st == 2
    st, _, _, _ = lfo.geo_glass_event([abs(R) * 1.5, 0, -20.0], [0, 0, 1.0], 1.0, 0.0, 1.0 / R, BIG_H2, 1.0, 0, 1)
    assert st == 2  # misses the sphere altogether


def _pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_thin_lens_focal_length():
    """C1 lens: a fan of rays leaving the on-axis sensor point at the paraxial back focal distance
    leaves the lens parallel to the axis (up to third-order spherical aberration ~ h^3)."""
    lens = _pkg().load_lens_file("thinlens.lens")
    zs = lfo.geo_z_sensor(lens)
    assert zs == pytest.approx(5.0 + 47.54, abs=1e-4)
    for h in (0.05, 0.2, 0.5):
        d = np.array([h, 0.0, -47.54])
        d /= np.linalg.norm(d)
        st, p, dd, w, ne = lfo.geo_trace_ray(lens, 1, -1, -1, [0, 0, zs], d)
        assert st == 0 and ne == 2
        slope = dd[0] / dd[2]
        assert abs(slope) < 2e-4 + 3e-3 * h ** 3   # collimated
        assert 0.90 < w < 0.93                       # two uncoated surfaces: (1 - 0.042)^2


def _reference_table_as_geometric_lens():
    """The reference's hard-coded 9-interface prescription (pathtracer.cpp:541-556) as a geometric
    lens: radius = 1/curvature, interface 5 is the stop, 6 a flat glass face."""
    L = lfo.default_lens()
    n = L.n
    radius = [0.0 if L.curvature[k] == 0 else 1.0 / L.curvature[k] for k in range(n)]
    ior = np.array([[L.ior[c][k] for k in range(n)] for c in range(3)], np.float32)
    return dict(n=n, stop=L.stop, radius=np.array(radius, np.float32),
                thickness=np.array([L.thickness[k] for k in range(n)], np.float32), ior=ior,
                semi_aperture=np.full(n, 30.0, np.float32), sensor_width_mm=36.0), L


def _paraxial_matrix(L, kind, i, j, colour):
    """The reference's own forward ghost matrix, recovered from its (pinned) linear tracer by
    probing with two tiny rays that never trigger the aperture recast."""
    e = 1e-3
    a = np.array(lfo.trace(L, kind, e, 0.0, i, j, colour)) / e
    b = np.array(lfo.trace(L, kind, 0.0, e, i, j, colour)) / e
    return np.column_stack([a, b])


PAIRS = [("before", i, j) for i in range(5) for j in range(i + 1, 5)] + \
        [("after", i, j) for i in range(6, 9) for j in range(i + 1, 9)]


def _physical_ghost_matrix(L, i, j, colour):
    """Textbook paraxial ghost matrix in the unfolded convention, built from the reference's own
    T / R / L operators (pathtracer.cpp:527-537).  It differs from the reference's
    trace_ray_auto_* in ONE place: on the backward leg the reference applies invert2x2(R_k)
    (pathtracer.cpp:608-609, :675-676), i.e. it retraces the refraction in time, which flips the
    sign of the surface power c (n1-n2); a ray that really travels backwards meets the mirrored
    interface, whose unfolded matrix is R(-c, n_k, n_{k-1}).  With no interface between the two
    mirrors (j = i+1) the two formalisms coincide."""
    def T(d):
        return np.array([[1.0, d], [0.0, 1.0]])

    def R(c, n1, n2):
        return np.array([[1.0, 0.0], [c * (n1 - n2) / n2, n1 / n2]])

    def Lm(c):
        return np.array([[1.0, 0.0], [2 * c, 1.0]])

    n = [1.0] + [L.ior[colour][k] for k in range(L.n)]   # n[k] = index in front of interface k
    c = [L.curvature[k] for k in range(L.n)]
    t = [L.thickness[k] for k in range(L.n)]

    def fwd(k):
        return T(t[k]) if k == L.stop else T(t[k]) @ R(c[k], n[k], n[k + 1])

    M = np.eye(2)
    for k in range(j):
        M = fwd(k) @ M
    M = Lm(c[j]) @ M
    for k in range(j - 1, i, -1):
        back = np.eye(2) if k == L.stop else R(-c[k], n[k + 1], n[k])
        M = back @ T(t[k]) @ M
    M = T(t[i]) @ np.linalg.inv(Lm(c[i])) @ T(t[i]) @ M
    for k in range(i + 1, L.n):
        M = fwd(k) @ M
    return M


@pytest.mark.parametrize("kind,i,j", PAIRS)
def test_reference_formalism_where_it_is_physical(kind, i, j):
    """Adjacent pairs (no backward refraction): the reference's pinned tracer IS the physical
    matrix -> the geometric march is anchored on the reference there.  Non-adjacent pairs: the
    reference's invert2x2(R) backward leg is not what light does (see _physical_ghost_matrix);
    the difference is asserted so the finding stays documented."""
    _, L = _reference_table_as_geometric_lens()
    for colour in range(3):
        ref = _paraxial_matrix(L, kind, i, j, colour)
        phys = _physical_ghost_matrix(L, i, j, colour)
        if j == i + 1:
            np.testing.assert_allclose(ref, phys, rtol=2e-5, atol=1e-6)
        else:
            assert np.abs(ref - phys).max() > 1e-3


@pytest.mark.parametrize("kind,i,j", PAIRS)
def test_small_angle_limit_matches_paraxial_ghosts(kind, i, j):
    """March a near-axis ray BACKWARDS through ghost pair (i, j) of the reference's own table; by
    reversibility the forward paraxial ghost matrix applied to the emerging ray must give back the
    sensor ray.  Confirms sequence order, reflection/refraction signs and index bookkeeping.  For
    adjacent pairs the matrix used is the reference's own (pinned) tracer."""
    lens, L = _reference_table_as_geometric_lens()
    zs = lfo.geo_z_sensor(lens)
    for colour in range(3):
        M = _paraxial_matrix(L, kind, i, j, colour) if j == i + 1 else \
            _physical_ghost_matrix(L, i, j, colour)
        for ys, us in ((0.02, 1e-4), (-0.01, 3e-4), (0.0, -2e-4)):
            d = np.array([us, 0.0, -1.0])   # travelling -z with dy/dz = -us ... see below
            d /= np.linalg.norm(d)
            st, p, dd, w, ne = lfo.geo_trace_ray(lens, colour, i, j, [ys, 0, zs], d)
            assert st == 0 and ne == 9 + 2 * (j - i)
            # time-reversed: the light ENTERS along the same line: height at interface 0, slope dx/dz
            y_in, u_in = p[0] - dd[0] / dd[2] * p[2], dd[0] / dd[2]
            y_s, u_s = M @ np.array([y_in, u_in])
            slope_s = d[0] / d[2]
            scale_y = max(abs(ys), abs(y_in), 1e-3)
            assert y_s == pytest.approx(ys, abs=2e-3 * scale_y + 2e-5)
            assert u_s == pytest.approx(slope_s, abs=2e-3 * max(abs(slope_s), abs(u_in)) + 2e-6)


def test_primary_path_matches_paraxial_product():
    """Primary (no reflection) path against the plain product of the reference's T and R matrices."""
    lens, L = _reference_table_as_geometric_lens()
    zs = lfo.geo_z_sensor(lens)
    for colour in range(3):
        M = np.eye(2)
        prev = 1.0
        for k in range(L.n):
            c, n2, t = L.curvature[k], L.ior[colour][k], L.thickness[k]
            Rm = np.eye(2) if k == L.stop else np.array([[1, 0], [c * (prev - n2) / n2, prev / n2]])
            M = np.array([[1, t], [0, 1]]) @ Rm @ M
            if k != L.stop:
                prev = n2
        d = np.array([2e-4, 0.0, -1.0])
        d /= np.linalg.norm(d)
        st, p, dd, w, ne = lfo.geo_trace_ray(lens, colour, -1, -1, [0.03, 0, zs], d)
        assert st == 0 and ne == 9
        y_in, u_in = p[0] - dd[0] / dd[2] * p[2], dd[0] / dd[2]
        y_s, u_s = M @ np.array([y_in, u_in])
        assert y_s == pytest.approx(0.03, abs=1e-4)
        assert u_s == pytest.approx(d[0] / d[2], abs=2e-6)
