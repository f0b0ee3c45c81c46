"""GPU parity of the scene-radiance term (the sample loop of raytrace_pixel,
pathtracer.cpp:831-875, with BVH closest hit, diffuse direct lighting, shadow rays) against frames
rendered by the REAL reference (tests/golden/s*.npz, generated through oracle/_ref/ref_dump from
programmatic scenes: spheres, triangles with interpolated normals, emissive surfaces, a directional
sun and a point light).  The full sensor pixel = (scene + ghost) + starburst must agree within the
north star's 1e-4 relative (asserted: 1e-9)."""
import numpy as np
import pytest

from goldenlib import Case, load_texels

pytestmark = pytest.mark.gpu
# (z*: random scenes -- spheres inside each other and behind the camera, sliver triangles, arbitrary vertex
# normals, two suns, point lights, turned cameras -- rendered by the real reference, oracle/make_golden_fuzz.py)
SCENE_CASES = ["s96x64_spheres", "s80x60_tris_rotcam", "z40x30_fuzz0", "z40x30_fuzz1", "z36x28_fuzz2",
               "z44x26_fuzz3"]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def lf(pkg):
    ctx = pkg.LensFlare(0)
    yield ctx
    ctx.close()


def scene_lights(case):
    """scene->lights in the reference's order: the DirectionalLights (dirToLight = unit(posLight),
    src/scene/light.cpp:11-24), then the point lights."""
    out = []
    for l in case.meta["lights"]:
        p = np.array(l[:3])
        out.append([0.0] + (p / np.sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2])).tolist() + list(l[3:6]))
    for p in case.meta["scene"]["points"]:
        out.append([1.0] + list(p))
    return out


def _render(pkg, lf, case, counter=False):
    m = case.meta
    lf.set_frame(case.W, case.H)
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.set_paraxial_lens()
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.set_flares(np.zeros((0, 2)), np.zeros((0, 3)), (0.0, 0.0), 0.0)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.find_sun_pos(m["lights"])
    sc = m["scene"]
    # the driver always adds one far-away diffuse sphere first (oracle/ref_driver.cpp)
    spheres = [(1e4, 1e4, 1e4, 1.0, "d", 0.5, 0.5, 0.5)] + [tuple(s) for s in sc["spheres"]]
    lf.set_scene(spheres, [tuple(t) for t in sc["tris"]], scene_lights(case))
    if counter:
        lf.set_jitter_counter(99)
    else:
        lf.set_jitter_mt19937(5489, None)
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    return lf.read_buffer(pkg.SAMPLE_BUFFER)


@pytest.mark.parametrize("name", SCENE_CASES)
def test_scene_term_matches_reference(pkg, lf, name):
    case = Case(name)
    got = _render(pkg, lf, case)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), case.ghost)
    err = np.abs(got - case.sample) / np.abs(case.sample)
    assert err.max() <= 1e-4
    assert err.max() <= 1e-9, err.max()
    # the scene term alone is a substantial part of these frames (the test is not vacuous)
    star = lf.read_buffer(pkg.STARBURST_BUFFER)
    scene = case.sample - case.ghost - star
    assert (scene.max(axis=-1) > 0.05).mean() > (0.3 if name[0] == "s" else 0.1)
    assert np.array_equal(lf.write_to_framebuffer(0, 0, case.W, case.H), case.rgba)


def test_collada_file_end_to_end(pkg, lf):
    """Row f3 + f2 + the flare path together: lf_load_collada(pyramid.dae) -- 138 triangles with the
    reference's half-edge vertex normals, two point lights and the sun -- rendered on the device
    against the frame the REAL reference renders from what its own COLLADA loader made of the same
    file (tests/golden/c96x72_pyramid_dae.npz, oracle/make_golden_collada.py)."""
    import os
    case = Case("c96x72_pyramid_dae")
    m = case.meta
    lf.set_frame(case.W, case.H)
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.set_paraxial_lens()
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.set_flares(np.zeros((0, 2)), np.zeros((0, 3)), (0.0, 0.0), 0.0)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    dae = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collada", "pyramid.dae")
    cam, suns = lf.load_collada(dae)
    assert cam is not None and abs(cam["hfov"] - 39.5978) < 1e-3
    assert suns == m["lights"]              # the sun's "position" and radiance, bit for bit
    lf.find_sun_pos(suns)
    lf.set_jitter_mt19937(5489, None)
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    got = lf.read_buffer(pkg.SAMPLE_BUFFER)
    err = np.abs(got - case.sample) / np.abs(case.sample)
    assert err.max() <= 1e-9, err.max()
    scene = case.sample - case.ghost - lf.read_buffer(pkg.STARBURST_BUFFER)
    assert (scene.max(axis=-1) > 0.02).mean() > 0.25     # the pyramids fill a good part of the frame
    assert np.array_equal(lf.write_to_framebuffer(0, 0, case.W, case.H), case.rgba)
    # glass BSDFs are unfilled stubs in the reference (f() = 0): black occluders, loaded as such
    cam2, _ = lf.load_collada(os.path.join(os.path.dirname(dae), "CBgems.dae"))
    assert cam2 is not None


def test_counter_jitter_converges_to_the_same_image(pkg, lf):
    """Order-free Philox pixel jitter: same estimator, different sub-pixel positions -> interior
    pixels (away from silhouettes) agree closely with the MT19937 frame."""
    case = Case("s96x64_spheres")
    a = _render(pkg, lf, case, counter=True)
    rel = np.abs(a - case.sample) / np.abs(case.sample)
    assert np.median(rel) < 0.02
    b = _render(pkg, lf, case, counter=True)
    assert np.array_equal(a, b)


def test_unsupported_inputs_fail_loudly(pkg, lf):
    lf.set_frame(16, 16)
    with pytest.raises(pkg.LensFlareError):
        lf.set_scene([(0, 0, -3, 1, "d", 1, 1, 1)], [], [[2.0, 0, 1, 0, 1, 1, 1]])  # area light: refused
    with pytest.raises(pkg.LensFlareError):  # a NaN vertex has no place in a tree of boxes
        lf.set_scene([], [(0, 0, -3, 1, 0, -3, float("nan"), 1, -3) + (0, 0, 1) * 3 + ("d", 1, 1, 1)], [])
    lf.set_scene([(0, 0, -3, 1, "d", 1, 1, 1)], [], [[1.0, 0, 3, 0, 1, 1, 1]])
    lf.set_camera(np.eye(3), [0, 0, 0], 50, 50)
    lf.set_params(40, 25.0, 1.0)           # ns_aa >= samplesPerBatch ...
    lf.set_jitter_mt19937(5489, None)
    with pytest.raises(pkg.LensFlareError):  # ... cannot be reproduced with the sequential RNG
        lf.render_scene_term()
    lf.set_jitter_counter(1)
    lf.render_scene_term()                  # fine with the counter RNG (adaptive early-out active)


def test_scene_buffer_reads_back(pkg):
    """lf_read_tile(which = 3): the scene term as the host left it (lf_set_scene_term) or as
    lf_render_scene_term computed it; LF_ERR_STATE while there is none."""
    lf = pkg.LensFlare(0)
    try:
        lf.set_frame(40, 24)
        with pytest.raises(pkg.LensFlareError) as e:
            lf.read_buffer(pkg.SCENE_BUFFER)
        assert "LF_ERR_STATE" in str(e.value) or e.value.status != 0
        rgb = np.random.default_rng(0).uniform(0, 2, (24, 40, 3))
        lf.set_scene_term(rgb)
        assert np.array_equal(lf.read_buffer(pkg.SCENE_BUFFER), rgb)
        assert np.array_equal(lf.read_tile(pkg.SCENE_BUFFER, 3, 5, 11, 9), rgb[5:9, 3:11])
        lf.set_params(2, 25.0, 1.0)
        lf.set_camera(np.eye(3), [0, 0, 0], 60.0, 38.0)
        lf.set_scene([(0, 0, -3, 1, "e", 1.0, 0.5, 0.25)], [], [])
        lf.set_jitter_counter(1)
        lf.render_scene_term()
        got = lf.read_buffer(pkg.SCENE_BUFFER)
        # (the reference divides the sum of ns_aa samples by the loop variable, ns_aa + 1: pathtracer.cpp:875)
        assert np.allclose(got[12, 20], np.array([1.0, 0.5, 0.25]) * 2 / 3) and np.all(got[0, 0] == 0)
    finally:
        lf.close()
