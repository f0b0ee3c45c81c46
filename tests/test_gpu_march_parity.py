"""GPU parity of the geometric march (lf_trace_ghosts) against the CPU oracle
(oracle/lf_geo_oracle.c, PARITY UNPINNED: there is no reference implementation of this path).
Both follow the same float32 arithmetic contract, so pixels AND event counters are compared bit for
bit; at full size the test falls back to size-independent properties.

The one operation of the contract that is not IEEE is the square root: the device uses v_sqrt_f32
(1 ulp).  The `lf` fixture measures its deviation from the correctly rounded root for all 2^24
(exponent parity, significand) patterns with lf_native_sqrt and installs it in the oracle, which
then reproduces the instruction exactly (test_native_sqrt_* check the premises)."""
import numpy as np
import pytest

from goldenlib import ray_budget, load_texels
from oracle import lfo

pytestmark = pytest.mark.gpu

SUN = dict(direction=[0.03, 0.02, -1.0], radiance=[1.0, 0.9, 0.5], angular_radius=0.05)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def lf(pkg):
    ctx = pkg.LensFlare(0)
    lfo.geo_follow_device(ctx)
    yield ctx
    lfo.geo_follow_device(None)
    ctx.close()


def test_native_sqrt_is_one_ulp_and_scale_invariant(lf):
    """Premises of the oracle's sqrt emulation: |v_sqrt_f32 - correctly rounded| <= 1 ulp, and the
    deviation depends only on (exponent parity, significand) -- checked on 2^20 random significands
    at exponents from 2^-60 to 2^+60 (the march feeds it 0, NaN or values >= 2^-54)."""
    table = lfo.sqrt_deviation_table(lf.native_sqrt)
    assert set(np.unique(table)) <= {-1, 0, 1}
    assert (table != 0).mean() < 0.5          # mostly the correctly rounded root ...
    assert (table != 0).any()                 # ... but not always: the emulation is needed
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 1 << 24, 1 << 20, dtype=np.uint32)
    for e2 in (-30, -13, -1, 0, 7, 30):       # x = pattern * 4^e2
        bits = (idx | np.uint32(0x3F000000)).astype(np.int64) + ((2 * e2) << 23)
        x = bits.astype(np.uint32).view(np.float32)
        hw = lf.native_sqrt(x).view(np.int32)
        want = np.sqrt(x).view(np.int32) + table[idx]
        assert np.array_equal(hw, want), e2
    # and the oracle's own sqrt follows the table
    x = (idx[:2000] | np.uint32(0x3F000000)).view(np.float32)
    assert np.array_equal(lfo.geo_sqrt(x).view(np.int32), lf.native_sqrt(x).view(np.int32))
    # special values: 0 -> 0, negative -> NaN, NaN -> NaN
    sp = lf.native_sqrt(np.array([0.0, -1.0, np.nan], np.float32))
    assert sp[0] == 0.0 and np.isnan(sp[1]) and np.isnan(sp[2])


def test_native_rcp_is_one_ulp_and_scale_invariant(lf):
    """Premises of the oracle's reciprocal emulation (round 4: the march's per-event / per-sample divisions
    are multiplications by v_rcp_f32): |v_rcp_f32 - correctly rounded 1/x| <= 1 ulp, and the deviation
    depends only on the significand -- checked on 2^20 random significands at exponents 2^-40 .. 2^+40, both
    signs (the march feeds it direction cosines and lengths of order 1e-2 .. 1e2)."""
    table = lfo.rcp_deviation_table(lf.native_rcp)
    assert set(np.unique(table)) <= {-1, 0, 1}
    assert (table != 0).any()                 # not the correctly rounded reciprocal: the emulation is needed
    rng = np.random.default_rng(6)
    idx = rng.integers(0, 1 << 23, 1 << 20, dtype=np.uint32)
    for e in (-40, -7, -1, 0, 1, 6, 40):
        for sign in (0, 1):
            bits = (idx | np.uint32(0x3F800000)).astype(np.int64) + (e << 23) + (sign << 31)
            x = bits.astype(np.uint32).view(np.float32)
            hw = lf.native_rcp(x).view(np.int32)
            want = (np.float32(1.0) / x).view(np.int32) + table[idx]
            assert np.array_equal(hw, want), (e, sign)
    x = ((idx[:2000] | np.uint32(0x3F800000)).view(np.float32) * np.float32(-0.37)).astype(np.float32)
    lfo.geo_set_rcp_table(None)
    try:
        assert np.array_equal(lfo.geo_rcp(x), (np.float32(1.0) / x).astype(np.float32))   # no table: correctly rounded
    finally:
        lfo.geo_set_rcp_table(table)       # (the module's oracle keeps following the device)
    assert np.array_equal(lfo.geo_rcp(x).view(np.int32), lf.native_rcp(x).view(np.int32))


def _run(pkg, lf, lens, W, H, spp, key, mask, pairs=None, primary=True, band=None, sun=SUN):
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(sun["direction"], sun["radiance"], sun["angular_radius"])
    lf.set_ghost_pairs(pairs, primary)
    if band:
        lf.set_band(*band)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    y0, y1 = band if band else (0, H)
    g = lf.read_buffer(pkg.GHOST_BUFFER)
    og, ocnt = lfo.geo_trace(lens, W, H, y0, y1, spp, key, pairs, primary, mask, sun["direction"],
                             sun["radiance"], sun["angular_radius"])
    return g, lf.counters(), og, ocnt


@pytest.mark.parametrize("W,H,spp", [(48, 32, 16), (33, 17, 5), (8, 4, 300), (16, 8, 256), (20, 6, 1)])
def test_dgauss_all_pairs_bit_exact(pkg, lf, W, H, spp):
    """Double-Gauss, primary + all 45 glass pairs, 3 wavelengths, pentagon mask."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    g, cnt, og, ocnt = _run(pkg, lf, lens, W, H, spp, 0xC0FFEE + spp, mask)
    assert ray_budget(lf, cnt, W * H * spp * 3 * 46)
    assert cnt == ocnt
    assert np.array_equal(g, og)
    assert cnt["rays_hit_light"] > 0 and og.max() > 0  # the test is not vacuous


@pytest.mark.parametrize("radius", [0.0008, 0.004, 0.3, 1.4])
def test_sun_lobes_from_sub_milliradian_to_wide(pkg, lf, radius):
    """The walk selects candidate rays with a host-side threshold on d.s (conservative against the
    float test the oracle applies, lf_march.hip lfk_march); the lobe factor itself decides.  Pixels and
    counters must stay those of the oracle from a 0.8 mrad sun to one of 1.4 rad."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    sun = dict(direction=[0.01, -0.015, -1.0], radiance=[1.0, 0.9, 0.5], angular_radius=radius)
    W, H, spp = (96, 64, 64) if radius < 0.01 else (40, 24, 16)
    g, cnt, og, ocnt = _run(pkg, lf, lens, W, H, spp, 0xBEEF, mask, sun=sun)
    assert cnt == ocnt
    assert np.array_equal(g, og)
    assert cnt["rays_hit_light"] > 0 and og.max() > 0


def test_thin_lens_config_c1(pkg, lf):
    """BASELINE.json configs[0]: single thin lens (2 spherical surfaces), 256x256, 1 spp, one
    light; no stop, one ghost pair (0,1) + primary.  GPU == CPU oracle bit for bit; plus 4 spp."""
    lens = pkg.load_lens_file("thinlens.lens")
    mask = np.ones((8, 8), np.float32)
    sun = dict(direction=[0.0, 0.0, -1.0], radiance=[1, 1, 1], angular_radius=0.1)
    for spp in (1, 4):
        g, cnt, og, ocnt = _run(pkg, lf, lens, 256, 256, spp, 99, mask, sun=sun)
        assert cnt == ocnt and np.array_equal(g, og)
        assert ray_budget(lf, cnt, 256 * 256 * spp * 3 * 2)
        assert og.max() > 0


def test_eight_wavelengths_config_c5_subset(pkg, lf):
    """BASELINE.json configs[4] asks for 8 wavelengths: the march takes any n_lambda <= 8 with an
    RGB weight per wavelength (lf_set_lambda_rgb).  8 indices by the 2-term Cauchy fit through the lens
    file's three columns (the committed dgauss11_8lambda.lens); GPU == oracle bit for bit, and with the 3
    original columns + identity weights inside 8 the 3-wavelength result is reproduced exactly."""
    lens8 = pkg.load_lens_file("dgauss11_8lambda.lens")   # 2-term Cauchy fit through C, d, F
    w8, _ = pkg.spectral_weights(lens8["lambda_nm"])
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 32, 16, 9
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens8)
    lf.set_lambda_rgb(w8)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, 77)
    g = lf.read_buffer(pkg.GHOST_BUFFER)
    og, ocnt = lfo.geo_trace(lens8, W, H, 0, H, spp, 77, None, True, mask, SUN["direction"],
                             SUN["radiance"], SUN["angular_radius"], lambda_rgb=w8)
    assert lf.counters() == ocnt and ray_budget(lf, ocnt, W * H * spp * 8 * 46)
    assert np.array_equal(g, og) and og.max() > 0
    # Lambda = 3 inside Lambda = 8 (SURVEY 8c KAT v): eight columns of which the first three are the
    # file's C, d, F columns with the identity weights and the other five carry no weight reproduce the
    # 3-wavelength frame exactly (the wavelengths of a sample share its start ray and add as integers)
    lens3 = pkg.load_lens_file("dgauss11.lens")
    cols = np.concatenate([lens3["ior"], np.repeat(lens3["ior"][1:2], 5, axis=0)]).astype(np.float32)
    w38 = np.zeros((8, 3), np.float32)
    w38[:3] = np.eye(3, dtype=np.float32)
    lf.set_lens(dict(lens3, ior=cols))
    lf.set_lambda_rgb(w38)
    lf.trace_ghosts(spp, 77)
    g38 = lf.read_buffer(pkg.GHOST_BUFFER)
    lf.set_lens(lens3)
    lf.trace_ghosts(spp, 77)
    assert np.array_equal(g38, lf.read_buffer(pkg.GHOST_BUFFER)) and g38.max() > 0


@pytest.mark.parametrize("n_lambda", [1, 2, 4, 5, 6, 7])
def test_every_wavelength_count_groups_correctly(pkg, lf, n_lambda):
    """The wavelengths walk the path tree in groups of up to three per lane (1, 2, 3, 2+2, 3+2, 3+3,
    3+3+1, 3+3+2): every count, short last group included, bit for bit against the oracle -- which
    marches each wavelength of each path on its own."""
    lens3 = pkg.load_lens_file("dgauss11.lens")
    lens, w, _ = pkg.spectral_lens(lens3, n_lambda)     # Cauchy-fitted indices at n_lambda wavelengths
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 24, 16, 9
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_lambda_rgb(w)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, 1000 + n_lambda)
    g = lf.read_buffer(pkg.GHOST_BUFFER)
    og, ocnt = lfo.geo_trace(lens, W, H, 0, H, spp, 1000 + n_lambda, None, True, mask, SUN["direction"],
                             SUN["radiance"], SUN["angular_radius"], lambda_rgb=w)
    assert lf.counters() == ocnt and ray_budget(lf, ocnt, W * H * spp * n_lambda * 46)
    assert np.array_equal(g, og) and og.max() > 0


def _random_lens(rng):
    """A random stack of singlets / cemented groups with a stop somewhere (or none): nothing that
    focuses, everything the event arithmetic and the path-tree builder can meet -- flats, weak and
    steep curvatures of both signs, total reflection, rays that miss the sphere, a stop next to the
    sensor or in front of everything."""
    n_glass = int(rng.integers(1, 4))
    rows = []   # radius, thickness, [ior x3], semi_aperture
    for _ in range(n_glass):
        m = int(rng.integers(2, 4))          # surfaces of this group (2 = singlet, 3 = cemented pair)
        for k in range(m):
            kind = rng.random()
            rad = 0.0 if kind < 0.15 else float(rng.choice([-1, 1]) * (12.0 if kind < 0.3 else 1.0) *
                                                 rng.uniform(14.0, 120.0) ** (1.6 if kind > 0.85 else 1.0))
            nd = float(rng.uniform(1.45, 1.85)) if k < m - 1 else 1.0
            disp = (nd - 1.0) / float(rng.uniform(25.0, 65.0))
            rows.append([rad, float(rng.uniform(0.4, 7.0)), nd - 0.3 * disp, nd, nd + 0.7 * disp,
                         float(rng.uniform(6.0, 14.0))])
    stop = int(rng.integers(-1, len(rows) + 1))
    stop = -1 if stop < 0 else stop
    if stop >= 0:
        # (the stop sits in air: insert it in front of a group's first surface, or behind the last)
        firsts = [0] + [i + 1 for i, r in enumerate(rows) if r[3] == 1.0]
        at = firsts[stop % len(firsts)]
        rows.insert(at, [0.0, float(rng.uniform(0.5, 5.0)), 1.0, 1.0, 1.0, float(rng.uniform(4.0, 9.0))])
        stop = at
    rows[-1][1] = float(rng.uniform(25.0, 60.0))
    a = np.array(rows, np.float64)
    return dict(n=len(rows), stop=stop, radius=a[:, 0].astype(np.float32), thickness=a[:, 1].astype(np.float32),
                ior=a[:, 2:5].T.astype(np.float32).copy(), semi_aperture=a[:, 5].astype(np.float32),
                sensor_width_mm=36.0)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_prescriptions_bit_exact(pkg, lf, seed):
    rng = np.random.default_rng(4242 + seed)
    lens = _random_lens(rng)
    mask = load_texels("pentbiglines.png") if seed % 2 else np.ones((8, 8), np.float32)
    sun = dict(direction=[float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.1, 0.1)), -1.0],
               radiance=[1.0, 0.8, 0.6], angular_radius=0.2)
    W, H, spp = 24, 16, 16
    try:
        g, cnt, og, ocnt = _run(pkg, lf, lens, W, H, spp, 31 + seed, mask, sun=sun)
    except pkg.LensFlareError as e:
        # a prescription the library refuses is refused with a message, not marched wrongly
        assert str(e)
        return
    n_pairs = sum(1 for i in range(lens["n"]) for j in range(i + 1, lens["n"])
                  if i != lens["stop"] and j != lens["stop"])
    assert ray_budget(lf, cnt, W * H * spp * 3 * (n_pairs + 1))
    assert cnt == ocnt
    assert np.array_equal(g, og)


def test_reference_pair_subset_and_no_primary(pkg, lf):
    """The reference's own enumeration (pathtracer.cpp:735-762): pairs on one side of the stop."""
    lens = pkg.load_lens_file("dgauss11.lens")
    stop = lens["stop"]
    pairs = [(i, j) for i in range(stop) for j in range(i + 1, stop)] + \
            [(i, j) for i in range(stop + 1, lens["n"]) for j in range(i + 1, lens["n"])]
    mask = load_texels("pentbiglines.png")
    g, cnt, og, ocnt = _run(pkg, lf, lens, 40, 24, 8, 5, mask, pairs=pairs, primary=False)
    assert ray_budget(lf, cnt, 40 * 24 * 8 * 3 * len(pairs))
    assert cnt == ocnt and np.array_equal(g, og)


def test_path_tree_odd_pair_lists(pkg, lf):
    """The device walks the selected paths as one tree (shared legs once); the oracle marches every
    path on its own.  Lists that stress the tree builder: unsorted, duplicated pairs, a single pair,
    the primary path alone, pairs that share only i or only j."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    cases = [
        ([(3, 8), (0, 2), (3, 4), (0, 10), (3, 8), (7, 9), (0, 2)], True),   # unsorted + duplicates
        ([(2, 9)], False),                                                   # one pair, no primary
        ([(-1, -1)], False),                                                 # the primary path alone
        ([(0, 10), (1, 10), (2, 10), (9, 10)], True),                        # same j
        ([(4, 6), (4, 7), (4, 10)], False),                                  # same i, across the stop
    ]
    for k, (pairs, primary) in enumerate(cases):
        g, cnt, og, ocnt = _run(pkg, lf, lens, 24, 16, 6, 100 + k, mask, pairs=pairs, primary=primary)
        assert cnt == ocnt, (k, cnt, ocnt)
        assert np.array_equal(g, og), k
        assert ray_budget(lf, cnt, 24 * 16 * 6 * 3 * (len(pairs) + int(primary)))
        assert 0 < lf.executed_events() <= cnt["surface_events"]


def test_flat_glass_surfaces(pkg, lf):
    """Plano-convex elements around the stop: refraction and mirror events at FLAT glass use the
    quotient form of the intersection (the curved form multiplies by R = 1/c)."""
    n = np.array([[1.60, 1.0, 1.0, 1.55, 1.0],
                  [1.61, 1.0, 1.0, 1.56, 1.0],
                  [1.62, 1.0, 1.0, 1.57, 1.0]], np.float32)
    lens = dict(n=5, stop=2, radius=np.array([45.0, 0.0, 0.0, 0.0, -38.0], np.float32),
                thickness=np.array([6.0, 4.0, 4.0, 5.0, 30.0], np.float32), ior=n,
                semi_aperture=np.array([14.0, 14.0, 6.0, 13.0, 13.0], np.float32), sensor_width_mm=24.0)
    mask = load_texels("pentbig500_14.png")
    sun = dict(SUN, direction=[0.02, 0.01, -1.0])
    g, cnt, og, ocnt = _run(pkg, lf, lens, 40, 24, 16, 31, mask, sun=sun)
    assert cnt == ocnt and np.array_equal(g, og)
    assert ray_budget(lf, cnt, 40 * 24 * 16 * 3 * 7)     # primary + C(4, 2) pairs
    assert cnt["rays_reached_scene"] > 0 and cnt["surface_events"] > 0


def test_band_sharding(pkg, lf):
    """Rows [y0,y1) only: identical to the same rows of the full frame (the multi-GPU shard)."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    g, cnt, og, ocnt = _run(pkg, lf, lens, 32, 24, 8, 11, mask, band=(5, 17))
    assert cnt == ocnt
    assert np.array_equal(g[5:17], og[5:17])
    full, _ = lfo.geo_trace(lens, 32, 24, 0, 24, 8, 11, None, True, mask, SUN["direction"],
                            SUN["radiance"], SUN["angular_radius"])
    assert np.array_equal(g[5:17], full[5:17])


def test_full_size_properties(pkg, lf):
    """1080p, 4 spp, all pairs: too big for the oracle -> properties.
    (a) counters are conserved: launched = clipped + vignetted + tir + reached_scene;
    (b) executed events never exceed the nominal sum_pairs (N + 2(j-i)) = 875 + 11 per sample-lambda;
    (c) determinism: same key -> identical bits; (d) linearity in the sun's radiance (x2, exact
    in fixed point up to one ulp of the 2^-36 grid per contributing ray);
    (e) a 64x36 crop of rows equals the oracle on those rows."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 1920, 1080, 64   # BASELINE.json configs[1]: 1080p, 64 spp
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, 42)
    a = lf.read_buffer(pkg.GHOST_BUFFER)
    c = lf.counters()
    assert ray_budget(lf, c, W * H * spp * 3 * 46)
    assert c["rays_launched"] == (c["rays_clipped_stop"] + c["rays_vignetted"] + c["rays_tir"] +
                                  c["rays_reached_scene"])
    assert c["surface_events"] <= W * H * spp * 3 * (875 + 11)
    assert c["rays_hit_light"] <= c["rays_reached_scene"]
    lf.trace_ghosts(spp, 42)
    assert np.array_equal(a, lf.read_buffer(pkg.GHOST_BUFFER))
    lf.set_sun(SUN["direction"], [2 * v for v in SUN["radiance"]], SUN["angular_radius"])
    lf.trace_ghosts(spp, 42)
    b = lf.read_buffer(pkg.GHOST_BUFFER)
    tol = 46 * 3 * 2.0 ** -36  # one grid step per contributing ray of a sample
    assert np.all(np.abs(b - 2 * a) <= tol)
    rows = (500, 502)
    og, _ = lfo.geo_trace(lens, W, H, rows[0], rows[1], spp, 42, None, True, mask, SUN["direction"],
                          SUN["radiance"], SUN["angular_radius"])
    assert np.array_equal(a[rows[0]:rows[1]], og[rows[0]:rows[1]])


@pytest.mark.parametrize("W,H,spp", [(1920, 1080, 2), (3840, 2160, 1)])
def test_whole_frame_bit_exact_at_low_spp(pkg, lf, W, H, spp):
    """The full BASELINE frame geometries (configs[1-2]: 1080p, configs[3-4]: 4K) -- primary + 45
    pairs, 3 wavelengths -- at 2 / 1 spp are 1.1e9 rays: a few seconds for the oracle on all host
    cores.  Every pixel and every counter of the whole frame, bit for bit (the device walks the
    path tree, the oracle each path)."""
    import os
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, 2024)
    g = lf.read_buffer(pkg.GHOST_BUFFER)
    cnt = lf.counters()
    og, ocnt = lfo.geo_trace(lens, W, H, 0, H, spp, 2024, None, True, mask, SUN["direction"],
                             SUN["radiance"], SUN["angular_radius"], n_threads=os.cpu_count() or 8)
    assert cnt == ocnt
    assert np.array_equal(g, og)
    assert ray_budget(lf, cnt, W * H * spp * 3 * 46) and cnt["rays_hit_light"] > 10 ** 6
    assert (g.max(axis=-1) > 0).mean() > 0.01     # the ghosts cover a visible part of the frame


def test_lens_camera_generate_ray(pkg, lf):
    """lf_generate_lens_rays (= LensCamera::generate_ray, batched): the primary path sensor ->
    scene.  (a) every alive ray equals the CPU oracle's march of the same start ray (the oracle is
    given the double-precision pupil direction, so the comparison is to 2e-5, not bitwise);
    (b) thin lens: a fan from the on-axis point at the paraxial focus leaves collimated;
    (c) directions are unit vectors, weights in (0, 1)."""
    rng = np.random.RandomState(3)
    n = 4096
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf.set_frame(64, 64)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    xy = (rng.rand(n, 2).astype(np.float32) - 0.5) * np.float32([36.0, 24.0])
    uv = (rng.rand(n, 2).astype(np.float32) * 2 - 1)
    out = lf.generate_lens_rays(1, xy, uv)
    alive = out[:, 7] > 0
    assert 0.1 < alive.mean() < 0.6                       # the pentagon stop clips most samples
    d = out[alive, 3:6].astype(np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=2e-6) and (d[:, 2] < 0).all()
    assert ((out[alive, 6] > 0) & (out[alive, 6] < 1)).all()
    zs = lfo.geo_z_sensor(lens)
    h, zp = float(lens["semi_aperture"][-1]), zs - float(lens["thickness"][-1])
    checked = 0
    for i in np.flatnonzero(alive)[:200]:
        a, b = float(uv[i, 0]), float(uv[i, 1])   # Shirley-Chiu concentric map in double
        if abs(a) > abs(b):
            r, phi = a, (np.pi / 4) * (b / a)
        else:
            r, phi = b, np.pi / 2 - (np.pi / 4) * (a / b)
        q = np.array([h * r * np.cos(phi), h * r * np.sin(phi), zp])
        p0 = np.array([float(xy[i, 0]), float(xy[i, 1]), zs])
        dd = (q - p0) / np.linalg.norm(q - p0)
        st, p, de, w, ne = lfo.geo_trace_ray(lens, 1, -1, -1, p0, dd, mask=mask)
        if st != 0:
            continue   # the oracle's double-precision start ray grazes an edge the float one clears
        # oracle weight starts at 1; the device folds in the pupil geometry factor
        geom = np.pi * h * h / (zs - zp) ** 2 * dd[2] ** 4
        assert np.allclose(out[i, 0:3], p, atol=3e-4) and np.allclose(out[i, 3:6], de, atol=2e-5)
        assert abs(out[i, 6] - w * geom) <= 2e-4 * w * geom
        checked += 1
    assert checked > 100
    thin = pkg.load_lens_file("thinlens.lens")
    lf.set_lens(thin)
    lf.set_aperture(pkg.APERTURE_STARBURST, np.ones((8, 8), np.float32))
    uv2 = (rng.rand(256, 2).astype(np.float32) * 2 - 1) * 0.05   # a narrow fan: paraxial
    o2 = lf.generate_lens_rays(1, np.zeros((256, 2), np.float32), uv2)
    assert (o2[:, 7] > 0).all()
    assert np.abs(o2[:, 3:5]).max() < 2e-4                 # collimated along -z


def test_4k_frame_properties(pkg, lf):
    """BASELINE.json configs[3] size (3840x2160; its dragon.dae is absent from the reference, so the
    synthetic sun stands in): counters conserved, and 8 interleave phases (the 8-GPU deal) add up to
    the single-launch frame exactly."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 3840, 2160, 2
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, 9)
    full = lf.read_buffer(pkg.GHOST_BUFFER)
    c = lf.counters()
    assert ray_budget(lf, c, W * H * spp * 3 * 46)
    assert c["rays_launched"] == (c["rays_clipped_stop"] + c["rays_vignetted"] + c["rays_tir"] +
                                  c["rays_reached_scene"])
    lf.set_frame(W, H)          # fresh (zeroed) buffers
    lf.set_lens(lens)
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    for phase in range(8):
        lf.set_row_interleave(phase, 8)
        lf.trace_ghosts(spp, 9)
    lf.set_row_interleave(0, 1)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), full)
    assert lf.counters() == c


def test_tile_row_interleave_reassembles_frame(pkg, lf):
    """lf_set_row_interleave (the N-GPU deal of 8-row tile rows): phases 0..2 of period 3, rendered
    one after the other into the same buffer, give exactly the single-launch frame; counters add."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 40, 52, 9   # 7 tile rows, the last one partial
    g, cnt, og, ocnt = _run(pkg, lf, lens, W, H, spp, 3, mask)
    assert np.array_equal(g, og) and cnt == ocnt
    lf.set_frame(W, H)
    lf.set_lens(lens)
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    for phase in range(3):
        lf.set_row_interleave(phase, 3)
        lf.trace_ghosts(spp, 3)
    lf.set_row_interleave(0, 1)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), og)
    assert lf.counters() == ocnt


def test_reprogram_without_readback(pkg, lf):
    """trace -> lf_set_ghost_pairs (same table capacity) -> trace with NO read-back in between:
    the context's stream is non-blocking, so the second program upload must wait for the first
    march (otherwise program and jump table change under a live kernel).  The second launch is a
    long one (so it is certainly still running when the third set-up arrives) and the final frame
    must be the oracle's frame for the last pair set."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 64, 32, 16
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(SUN["direction"], SUN["radiance"], SUN["angular_radius"])
    stop = lens["stop"]
    before = [(i, j) for i in range(stop) for j in range(i + 1, stop)]
    after = [(i, j) for i in range(stop + 1, lens["n"]) for j in range(i + 1, lens["n"])]
    lf.set_ghost_pairs(None, True)          # the big table first: capacity for everything below
    lf.trace_ghosts(spp, 1)
    for k in range(6):                      # alternate programs of different shapes, no read-back
        lf.set_ghost_pairs(before if k % 2 == 0 else after, k % 3 == 0)
        lf.trace_ghosts(256 if k == 4 else spp, 100 + k)
    lf.reset_counters()
    lf.set_ghost_pairs(after, True)
    lf.trace_ghosts(spp, 321)
    g = lf.read_buffer(pkg.GHOST_BUFFER)
    og, ocnt = lfo.geo_trace(lens, W, H, 0, H, spp, 321, after, True, mask, SUN["direction"],
                             SUN["radiance"], SUN["angular_radius"])
    assert lf.counters() == ocnt
    assert np.array_equal(g, og)


def test_empty_band_and_tonemap_rows(pkg, lf):
    """lf_set_band(y, y) is a no-op for every launcher (a 0-block launch would be a HIP error), and
    lf_write_to_framebuffer of a tile outside the current band tonemaps those rows instead of
    returning stale zeros."""
    from goldenlib import Case
    case = Case("f64x48_pentbiglines")
    m = case.meta
    lf.set_frame(case.W, case.H)
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.find_sun_pos(m["lights"])
    lf.set_jitter_mt19937(5489, None)
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    full = lf.write_to_framebuffer(0, 0, case.W, case.H)
    assert np.array_equal(full, case.rgba)
    lf.set_band(16, 16)                      # empty band: every launcher returns LF_OK
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    lf.set_band(8, 16)
    lf.render_flare_layer()                  # invalidates the tonemapped copy
    tile = lf.write_to_framebuffer(0, 32, 32, 48)   # rows outside the band
    assert np.array_equal(tile, case.rgba[32:48, 0:32])
    lf.set_band(0, case.H)


@pytest.mark.parametrize("stride,W,H", [(1, 100, 24), (1, 64, 16), (2, 50, 12), (4, 33, 9)])
def test_tile_stride_bit_exact(pkg, lf, stride, W, H):
    """lf_set_tile_stride: the lanes of a wave take pixels `stride` apart in x (round 4 default: 8 -- the
    shared pupil sub-cell then correlates pixels 8 apart instead of neighbours; every other test of this
    file runs that default).  A sampling-specification parameter like
    the sub-cells: pixels and counters stay those of the oracle, which derives the sub-cell of a pixel from
    the same tile numbering -- frame widths that are no multiple of the block included."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf.set_tile_stride(stride)
    lfo.set_tile_stride(stride)
    try:
        g, cnt, og, ocnt = _run(pkg, lf, lens, W, H, 16, 0x57D + stride, mask)
        assert cnt == ocnt and ray_budget(lf, cnt, W * H * 16 * 3 * 46)
        assert np.array_equal(g, og) and og.max() > 0
        # not the default specification (columns 8 apart): the same key draws other sub-cells there
        lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
        lf.trace_ghosts(16, 0x57D + stride)
        assert not np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), g)
        # a band of tile rows under the interleave of a multi-GPU deal
        lf.set_tile_stride(stride)
        lf.set_row_interleave(1, 2)
        lf.clear_ghost_buffer()
        lf.trace_ghosts(16, 0x57D + stride)
        part = lf.read_buffer(pkg.GHOST_BUFFER)
        own = (np.arange(H) // 8) % 2 == 1
        assert np.array_equal(part[own], og[own]) and not part[~own].any()
    finally:
        lf.set_row_interleave(0, 1)
        lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
        lfo.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)

