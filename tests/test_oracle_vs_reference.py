"""Pins the CPU oracle (oracle/lf_oracle.c) to the REAL reference: every check compares the
oracle against fixtures that oracle/make_golden.py produced by running the reference's own code
(oracle/_ref/ref_dump).  CPU only."""
import os

import numpy as np
import pytest

from goldenlib import GOLD, Case, FRAME_CASES, aperture_stats_golden, load_red, load_texels
from oracle import lfo

# reference and oracle run the same libm in the same container: doubles agree to rounding noise
TOL = 1e-12


def test_aperture_texture_matches_reference():
    """CameraApertureTexture::init (camera.h:26-83): texels, bbox and total_value, all 17 PNGs."""
    import hashlib
    gold = aperture_stats_golden()
    assert len(gold) == 17
    for name, g in gold.items():
        tex, st = lfo.aperture_from_red(load_red(name))
        assert hashlib.sha256(tex.tobytes()).hexdigest() == g["sha256_f32"], name
        assert np.array_equal(tex, load_texels(name))
        assert (st.width, st.height) == (g["width"], g["height"])
        assert (st.min_x, st.min_y, st.max_x, st.max_y) == (g["min_x"], g["min_y"], g["max_x"], g["max_y"]), name
        assert st.total_value == float.fromhex(g["total_value"]), name


def test_paraxial_trace_bit_exact():
    """trace_ray_auto_before/after (pathtracer.cpp:588-689): 7 angles x 3 colours x 13 pairs x 2 rays."""
    L = lfo.default_lens()
    n = 0
    for line in open(os.path.join(GOLD, "paraxial_trace.txt")):
        kind, th, c, i, j, r, x, y = line.split()
        got = lfo.trace(L, kind, float.fromhex(r), float.fromhex(th), int(i), int(j), int(c))
        assert got == (float.fromhex(x), float.fromhex(y)), line
        n += 1
    assert n == 7 * 3 * 13 * 2


def test_convert_coordinate_bit_exact():
    for line in open(os.path.join(GOLD, "convert_coordinate.txt")):
        length, yflag, p, v = line.split()
        assert lfo.convert_coordinate(int(p), int(length), int(yflag)) == float.fromhex(v)


def _frame(case):
    m = case.meta
    f = lfo.make_frame(case.W, case.H, ns_aa=m["ns_aa"], flare_radius=m["flare_radius"],
                       flare_intensity=m["flare_intensity"])
    lfo.find_sun_pos(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"], m["lights"], f)
    return f


@pytest.mark.parametrize("name", FRAME_CASES + ["f64x64_no_sun"])
def test_find_sun_pos_bit_exact(name):
    """find_sun_pos + analyze_world_coord (pathtracer.cpp:32-64, camera.cpp:245-273)."""
    case = Case(name)
    f = _frame(case)
    assert f.n_flares == case.meta["n_flares"]
    for k, fl in enumerate(case.flares):
        assert (f.flare_origin[k][0], f.flare_origin[k][1]) == (fl[0], fl[1])
        assert tuple(f.flare_radiance[k]) == tuple(fl[2:])
    if f.n_flares:
        assert tuple(f.axis_ray) == tuple(case.meta["axis_ray"])
        assert f.angle_to_sun == case.meta["angle_to_sun"]


@pytest.mark.parametrize("name", FRAME_CASES + ["f64x64_no_sun"])
def test_ghost_buffer_bit_exact(name):
    """generate_ghost_buffer (pathtracer.cpp:714-817): 39 textured quads, every pixel bit-exact."""
    case = Case(name)
    f = _frame(case)
    ghost = lfo.ghost_buffer(lfo.default_lens(), f, load_texels(case.meta["ghost_aperture"]))
    assert np.array_equal(ghost, case.ghost)


@pytest.mark.parametrize("name", FRAME_CASES)
def test_raytrace_pixel_matches_reference(name):
    """raytrace_pixel (pathtracer.cpp:819-899) = ghost + starburst + falloff, in visit order."""
    case = Case(name)
    f = _frame(case)
    tex, st = lfo.aperture_from_red(load_red(case.meta["aperture"]))
    order = case.order if case.order is not None else lfo.tile_order(case.W, case.H)
    got = lfo.render_pixels(f, tex, st, case.ghost, order, n_threads=8)
    if case.order is not None:
        # a repeated pixel keeps its last visit, like the reference buffer
        got_o, ref_o = got.reshape(-1, 3)[order], case.sample_at_order
    else:
        got_o, ref_o = got, case.sample
    err = np.abs(got_o - ref_o) / np.abs(ref_o)
    assert err.max() <= TOL, (name, err.max())
    rgba = lfo.to_color(got)
    ref_rgba = case.rgba_at_order if case.order is not None else case.rgba
    got_rgba = rgba.reshape(-1)[order] if case.order is not None else rgba
    assert np.array_equal(got_rgba, ref_rgba)


def test_spectral_starburst_reduces_to_the_reference_formula():
    """Row f4 has no reference counterpart (parity unpinned), but one wavelength with scale 1 and
    weight (1,1,1) must BE the reference's starburst: compare the spectral restatement with
    lfo_starburst_pixel, which the golden frames above pin to the real reference."""
    case = Case("f64x48_pentbiglines")
    f = _frame(case)
    tex, st = lfo.aperture_from_red(load_red(case.meta["aperture"]))
    rng = np.random.default_rng(1)
    for x, y in zip(rng.integers(0, case.W, 24), rng.integers(0, case.H, 24)):
        ref, _ = lfo.starburst_pixel(f, tex, st, int(x), int(y))
        got = lfo.starburst_pixel_spectral(f, tex, st, int(x), int(y), [1.0], [[1.0, 1.0, 1.0]])
        assert np.all(np.abs(got - ref) <= 1e-9 * np.abs(ref)), (x, y, got, ref)
    # and it is linear in the weights / additive over wavelengths
    a = lfo.starburst_pixel_spectral(f, tex, st, 40, 20, [0.9], [[1.0, 0.0, 0.5]])
    b = lfo.starburst_pixel_spectral(f, tex, st, 40, 20, [1.15], [[0.0, 2.0, 0.5]])
    ab = lfo.starburst_pixel_spectral(f, tex, st, 40, 20, [0.9, 1.15], [[1.0, 0.0, 0.5], [0.0, 2.0, 0.5]])
    assert np.allclose(ab, a + b, rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", ["s96x64_spheres", "s80x60_tris_rotcam", "z40x30_fuzz0", "z40x30_fuzz1",
                                  "z36x28_fuzz2", "z44x26_fuzz3"])
def test_scene_term_oracle_matches_reference(name):
    """Row f2: the sample loop of raytrace_pixel with real geometry (oracle/lf_scene_oracle.c):
    spheres, triangles with interpolated normals, emission, sun + point light, shadow rays --
    against frames rendered by the real reference; RGBA8 byte-exact."""
    case = Case(name)
    m = case.meta
    f = _frame(case)
    tex, st = lfo.aperture_from_red(load_red(m["aperture"]))
    order = lfo.tile_order(case.W, case.H)
    lights = []
    for l in m["lights"]:   # DirectionalLight: dirToLight = unit(posLight) (src/scene/light.cpp:11-24)
        p = np.array(l[:3])
        lights.append([0.0] + (p / np.sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2])).tolist() + list(l[3:6]))
    lights += [[1.0] + list(p) for p in m["scene"]["points"]]
    spheres = [(1e4, 1e4, 1e4, 1.0, "d", 0.5, 0.5, 0.5)] + [tuple(s) for s in m["scene"]["spheres"]]
    scene = lfo.scene_term(case.W, case.H, m["ns_aa"], m["c2w"], m["cam_pos"], m["hFov"], m["vFov"],
                           spheres, [tuple(t) for t in m["scene"]["tris"]], lights, order)
    assert (scene.max(axis=-1) > 0.05).mean() > (0.3 if name[0] == "s" else 0.1)   # not vacuous
    lfo.set_scene_term(scene)
    try:
        got = lfo.render_pixels(f, tex, st, case.ghost, order, n_threads=8)
    finally:
        lfo.set_scene_term(None)
    err = np.abs(got - case.sample) / np.abs(case.sample)
    assert err.max() <= TOL, err.max()
    assert np.array_equal(lfo.to_color(got), case.rgba)


def test_mt19937_known_answer():
    """std::mt19937 default seed: the 10000th output is 4123659995 (C++11 [rand.predef])."""
    raw = lfo.mt19937_raw(5489, 9999, 1)
    assert int(raw[0]) == 4123659995
