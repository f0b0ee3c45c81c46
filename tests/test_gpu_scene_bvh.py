"""The device BVH of the scene term (lf_scene.hip) only culls: its float boxes, nearest-first order,
split rule and leaf size must never change WHICH primitive test accepts a ray -- those tests are the
reference's own double-precision expressions (scene/sphere.cpp:11-111, scene/triangle.cpp:25-112,
closest hit by shrinking max_t as in scene/bvh.cpp:201-222).

(1) lf_scene_trace_ray against a brute force over every primitive that evaluates the same expressions
    in numpy (same operation order, so t must agree bit for bit), on random rays and on the rays a
    conservative float box test gets wrong first: axis-parallel ones, origins on box planes and far
    from the scene, direction components that are zero, denormal or tiny, windows [min_t, max_t] that
    cut the scene.
(2) the same frame through every tree the builder can make (SAH / median, leaves of 1, 2, 4): equal
    pixel for pixel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(u, v):
    return np.stack([u[..., 1] * v[..., 2] - u[..., 2] * v[..., 1], u[..., 2] * v[..., 0] - u[..., 0] * v[..., 2],
                     u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]], -1)


def brute_force(spheres, tris, o, d, min_t, max_t):
    """closest accepted t over all primitives (inf = no hit) + the index of the primitive; None when
    some triangle accepts the ray with a NaN / infinite t (a ray inside the plane of a triangle it is
    exactly parallel to: 0 * inf passes every comparison of the reference's test, and what the
    reference then returns depends on its own visiting order -- not a case any tree can be held to)"""
    best, who = np.inf, -1
    with np.errstate(all="ignore"):
        if len(tris):
            p0, p1, p2 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
            e1, e2, s = p1 - p0, p2 - p0, o - p0
            s1, s2 = _cross(np.broadcast_to(d, e2.shape), e2), _cross(s, e1)
            rc = 1.0 / _dot(s1, e1)
            tt, b1, b2 = _dot(s2, e2) * rc, _dot(s1, s) * rc, _dot(s2, np.broadcast_to(d, s2.shape)) * rc
            ok = ~((tt < min_t) | (tt > max_t)) & ~((b1 < 0) | (b1 > 1)) & ~((b2 < 0) | (b2 > 1)) & ~(b1 + b2 > 1)
            if (ok & ~np.isfinite(tt)).any():
                return None, -1
            if ok.any():
                k = np.where(ok, tt, np.inf).argmin()
                best, who = tt[k], len(spheres) + k
        for i, sp in enumerate(spheres):
            c, r2 = sp[:3], sp[3] * sp[3]
            oc = o - c
            a, b, cc = _dot(d, d), 2 * _dot(oc, d), _dot(oc, oc) - r2
            if b * b < 4.0 * a * cc or np.isnan(b * b - 4.0 * a * cc):
                continue
            if b * b == 4.0 * a * cc:
                t1 = (-b) / (2.0 * a)
                if t1 < min_t or t1 > max_t:
                    continue
            else:
                q = np.sqrt(b * b - 4.0 * a * cc)
                r1, r2_ = (-b - q) / (2.0 * a), (-b + q) / (2.0 * a)
                lo, hi = min(r1, r2_), max(r1, r2_)
                if lo > max_t or hi < min_t:
                    continue
                if lo < min_t:
                    if hi > max_t:
                        continue
                    t1 = hi
                else:
                    t1 = lo
            if t1 <= best:
                best, who = t1, i
    return best, who


def make_soup(rng, n_tris, n_spheres, scale=1.0, offset=(0.0, 0.0, 0.0)):
    cen = rng.uniform(-4, 4, (n_tris, 1, 3))
    tris = (cen + rng.normal(0, 0.25, (n_tris, 3, 3))).reshape(n_tris, 9) * scale + np.tile(offset, 3)
    # some axis-aligned, grid-snapped triangles: box planes that rays can hit exactly
    k = n_tris // 5
    off3 = np.tile(offset, 3)
    big = (cen[:k] + rng.normal(0, 0.8, (k, 3, 3))).reshape(k, 9)
    snapped = np.round(big * 2) / 2
    e1, e2 = snapped[:, 3:6] - snapped[:, 0:3], snapped[:, 6:9] - snapped[:, 0:3]
    flat = (np.abs(_cross(e1, e2)).sum(-1) == 0)          # (collinear / coincident vertices accept EVERY ray
    snapped[flat] = big[flat]                             #  with t = NaN in the reference's test: left out)
    tris[:k] = snapped * scale + off3
    nrm = rng.normal(0, 1, (n_tris, 3, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    sph = np.concatenate([rng.uniform(-4, 4, (n_spheres, 3)) * scale + offset,
                          rng.uniform(0.05, 0.6, (n_spheres, 1)) * scale], 1)
    sph[::7, 3] *= -1.0     # (Sphere::test only reads r^2: a negative radius is still a sphere)
    return sph, tris, nrm.reshape(n_tris, 9)


def rays_for(rng, sph, tris, n_random):
    lo = min(tris.reshape(-1, 3).min(0).min(), -1.0)
    hi = max(tris.reshape(-1, 3).max(0).max(), 1.0)
    ext = hi - lo
    rays = []
    for _ in range(n_random):
        o = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, 3)
        d = rng.normal(0, 1, 3)
        rays.append((o, d / np.linalg.norm(d), 0.0, np.inf))
    verts = tris.reshape(-1, 3)
    for i in range(120):
        v = verts[rng.integers(len(verts))]
        a = i % 3
        d = np.zeros(3); d[a] = 1.0 if i & 1 else -1.0                     # axis-parallel through a vertex
        o = v - d * rng.uniform(1, 20) * ext * 0.1
        rays.append((o, d, 0.0, np.inf))
        d2 = d.copy(); d2[(a + 1) % 3] = [0.0, 5e-324, 1e-310, 1e-40, -1e-30, 1e-17][i % 6]   # zero / denormal / tiny
        rays.append((o, d2, 0.0, np.inf))
        c = tris[rng.integers(len(tris))].reshape(3, 3).mean(0)            # at a triangle's centroid from far away
        o3 = c + rng.normal(0, 1, 3) * 1e4 * ext
        d3 = c - o3
        rays.append((o3, d3 / np.linalg.norm(d3), 0.0, np.inf))
        o4 = rng.uniform(lo, hi, 3); d4 = rng.normal(0, 1, 3)              # windows that cut the scene
        t0 = rng.uniform(0, ext)
        rays.append((o4, d4, t0, t0 + rng.uniform(0, ext)))
        o5 = np.round(rng.uniform(lo, hi, 3) * 2) / 2                      # origin on the planes of the snapped boxes
        rays.append((o5, np.round(rng.normal(0, 1, 3) * 2) / 2 + [0.0, 0.0, 0.5], 0.0, np.inf))
    for s in sph[:20]:
        d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
        rays.append((s[:3] - d * 3 * s[3], d, 0.0, np.inf))                 # through a sphere's centre
        rays.append((s[:3], d, 0.0, np.inf))                                # from inside
    return rays


@pytest.mark.parametrize("seed,scale,offset", [(1, 1.0, (0.0, 0.0, 0.0)), (2, 1e-3, (0.0, 0.0, 0.0)),
                                                 (3, 1.0, (1e5, -2e5, 3e5)), (4, 300.0, (0.0, 10.0, 0.0))])
def test_trace_ray_equals_brute_force(pkg, seed, scale, offset):
    rng = np.random.default_rng(seed)
    sph, tris, nrm = make_soup(rng, 1500, 40, scale, np.array(offset))
    lf = pkg.LensFlare(0)
    try:
        lf.set_scene([tuple(s) + ("d", 0.5, 0.5, 0.5) for s in sph],
                     [tuple(t) + tuple(n) + ("d", 0.5, 0.5, 0.5) for t, n in zip(tris, nrm)], [])
        hits = skipped = 0
        for o, d, t0, t1 in rays_for(rng, sph, tris, 250):
            got = lf.scene_trace_ray(o, d, t0, t1)
            want_t, who = brute_force(sph, tris, np.asarray(o, float), np.asarray(d, float), t0, t1)
            if want_t is None:
                skipped += 1
                continue
            assert got["hit"] == np.isfinite(want_t), (o, d, t0, t1, got, want_t)
            if got["hit"]:
                hits += 1
                assert got["t"] == want_t, (o, d, got["t"], want_t)      # the same expressions: the same bits
                assert abs(np.linalg.norm(got["n"]) - 1) < 1e-12
        assert hits > 150 and skipped < 200, (hits, skipped)
    finally:
        lf.close()


def test_every_tree_renders_the_same_frame(pkg):
    rng = np.random.default_rng(9)
    sph, tris, nrm = make_soup(rng, 3000, 30)
    tris[:, 2::3] -= 9.0
    sph[:, 2] -= 9.0
    spheres = [tuple(s) + ("d", 0.6, 0.5, 0.4) for s in sph]
    triangles = [tuple(t) + tuple(n) + ("d", 0.4, 0.6, 0.5) for t, n in zip(tris, nrm)]
    lights = [[0.0, 0.3, 0.9, 0.3, 1.0, 1.0, 1.0], [1.0, 0.0, 3.0, -4.0, 20.0, 20.0, 20.0]]
    frames = {}
    for split in ("sah", "median"):
        for leaf in ("1", "2", "4"):
            lf = pkg.LensFlare(0)
            lf.test_knob("bvh_median", split == "median")       # (the tree of the next set_scene)
            lf.test_knob("bvh_leaf", int(leaf))
            lf.set_frame(160, 96)
            lf.set_params(4, 25.0, 1.0)
            lf.set_camera(np.eye(3), [0.0, 0.0, 2.0], 60.0, 38.0)
            lf.set_scene(spheres, triangles, lights)
            lf.set_jitter_counter(5)
            lf.render_scene_term()
            frames[(split, leaf)] = lf.read_buffer(pkg.SCENE_BUFFER)
            lf.close()
    ref = frames[("sah", "2")]
    assert (ref.max(axis=-1) > 0).mean() > 0.15
    for k, f in frames.items():
        assert np.array_equal(f, ref), k


def test_bands_and_row_interleave_tile_the_frame(pkg):
    """The kernel maps a wave to an 8 x 8 pixel tile: bands whose edges cut tiles, frames whose sizes are
    no multiples of 8 / 32, and the multi-GPU tile-row deal must each produce exactly the pixels of the
    one-launch frame (the jitter and light samples are keyed by the pixel, not by the launch)."""
    rng = np.random.default_rng(21)
    sph, tris, nrm = make_soup(rng, 800, 20)
    tris[:, 2::3] -= 9.0
    sph[:, 2] -= 9.0
    spheres = [tuple(s) + ("d", 0.6, 0.5, 0.4) for s in sph]
    triangles = [tuple(t) + tuple(n) + ("d", 0.4, 0.6, 0.5) for t, n in zip(tris, nrm)]
    lights = [[0.0, 0.3, 0.9, 0.3, 1.0, 1.0, 1.0]]
    W, H = 77, 45

    def ctx():
        lf = pkg.LensFlare(0)
        lf.set_frame(W, H)
        lf.set_params(3, 25.0, 1.0)
        lf.set_camera(np.eye(3), [0.0, 0.0, 2.0], 60.0, 38.0)
        lf.set_scene(spheres, triangles, lights)
        lf.set_jitter_counter(5)
        return lf

    lf = ctx()
    lf.render_scene_term()
    whole = lf.read_buffer(pkg.SCENE_BUFFER)
    lf.close()
    assert (whole.max(axis=-1) > 0).mean() > 0.1

    lf = ctx()
    for y0, y1 in ((0, 13), (13, 14), (14, 37), (37, H)):
        lf.set_band(y0, y1)
        lf.render_scene_term()
    assert np.array_equal(lf.read_buffer(pkg.SCENE_BUFFER), whole)
    lf.close()

    got = np.zeros_like(whole)
    for phase in range(3):
        lf = ctx()
        lf.set_row_interleave(phase, 3)
        lf.render_scene_term()
        part = lf.read_buffer(pkg.SCENE_BUFFER)
        rows = [y for y in range(H) if (y >> 3) % 3 == phase]
        got[rows] = part[rows]
        others = [y for y in range(H) if (y >> 3) % 3 != phase]
        assert not part[others].any()          # nobody else's tile rows are touched
        lf.close()
    assert np.array_equal(got, whole)
