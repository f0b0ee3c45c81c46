"""BASELINE.json's configurations as whole frames through the C ABI ("configs only
exercised in pieces"): C5 = 8-wavelength march + spectral starburst composed in ONE frame (small
size against the oracles, 4K x 1 spp and 4K x 1024 spp by properties); C3 at its full 1080p x 256 spp
(properties + a row crop bit for bit against the oracle); C4 at its full 4K x 256 spp with the scene imaged
through the lens (properties + the sun's tile row bit for bit against the oracle)."""
import math

import numpy as np
import pytest

from goldenlib import ray_budget, load_red, load_texels
from oracle import lfo

pytestmark = pytest.mark.gpu
SUN_NS = (0.521445, 0.517156)


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


def _lens8(pkg):
    """C5's prescription: the committed 8-column file (2-term Cauchy fit)"""
    lens8 = pkg.load_lens_file("dgauss11_8lambda.lens")
    w8, scale = pkg.spectral_weights(lens8["lambda_nm"])
    return lens8, w8, scale


def _c5_setup(pkg, lf, W, H, mask_name="pentbig500_14.png"):
    lens8, w8, scale = _lens8(pkg)
    efl = pkg.paraxial_efl(lens8)
    hf = 2 * math.degrees(math.atan(0.5 * lens8["sensor_width_mm"] / efl))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_frame(W, H)
    lf.set_params(1, 25.0, 1.0)
    mask = load_texels(mask_name)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens8)
    lf.set_lambda_rgb(w8)
    lf.set_ghost_pairs(None, True)
    lf.set_starburst_spectrum(scale, w8)
    lf.set_scene_term(None)
    lf.set_jitter_counter(0x1e45f1a4e)
    lf.set_camera(np.eye(3), [0, 0, 0], hf, vf)
    ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
    light = [(2 * SUN_NS[0] - 1) * ex * 10, (2 * SUN_NS[1] - 1) * ey * 10, -10.0, 1.0, 0.9, 0.5]
    return lens8, w8, scale, mask, efl, hf, vf, light


def test_c5_one_frame_small(pkg, lf):
    """C5 at 96x54, 16 spp: ghost buffer = the 8-wavelength march, bit for bit against the oracle;
    starburst = the per-wavelength pattern against the oracle's restatement; sensor = their sum."""
    W, H, spp, key = 96, 54, 16, 0xC5
    lens8, w8, scale, mask, efl, hf, vf, light = _c5_setup(pkg, lf, W, H)
    lf.find_sun_pos([light])
    lf.set_sun_from_flares(0, efl, 0.05)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()
    ghost, star, sample = (lf.read_buffer(b) for b in (pkg.GHOST_BUFFER, pkg.STARBURST_BUFFER, pkg.SAMPLE_BUFFER))
    cnt = lf.counters()
    sun = [(SUN_NS[0] - 0.5) * lens8["sensor_width_mm"] / efl, (SUN_NS[1] - 0.5) * lens8["sensor_width_mm"] * H / W / efl, -1.0]
    og, ocnt = lfo.geo_trace(lens8, W, H, 0, H, spp, key, None, True, mask, sun, [1.0, 0.9, 0.5], 0.05, lambda_rgb=w8)
    assert cnt == ocnt and ray_budget(lf, cnt, W * H * spp * 8 * 46)
    assert np.array_equal(ghost, og) and og.max() > 0
    assert np.array_equal(sample, ghost + star)         # (scene + ghost) + starburst, no scene term
    # the spectral starburst against the oracle (falloff included in both: compare the spectral part)
    f = lfo.make_frame(W, H, ns_aa=1, flare_radius=25.0, flare_intensity=1.0)
    lfo.find_sun_pos(np.eye(3), [0, 0, 0], hf, vf, [light], f)
    tex, st = lfo.aperture_from_red(load_red("pentbig500_14.png"))
    lf.set_starburst_spectrum(None)
    lf.render_flare_layer()
    mono = lf.read_buffer(pkg.STARBURST_BUFFER)
    rng = np.random.default_rng(5)
    for x, y in zip(rng.integers(0, W, 60), rng.integers(0, H, 60)):
        o_mono, _ = lfo.starburst_pixel(f, tex, st, int(x), int(y))
        o_spec = lfo.starburst_pixel_spectral(f, tex, st, int(x), int(y), scale, w8)
        tol = 1e-9 * np.maximum(np.abs(o_spec), np.abs(o_mono)) + 1e-13 * np.abs(mono[y, x])
        assert np.all(np.abs((star[y, x] - mono[y, x]) - (o_spec - o_mono)) <= tol), (x, y)


@pytest.mark.parametrize("spp", [1, 1024])
def test_c5_whole_4k_frame_properties(pkg, lf, spp):
    """C5's frame size with 8 wavelengths + spectral starburst, at 1 spp and at the configuration's full
    1024 spp: exact ray budget, every ray accounted for, sensor = ghost + starburst on a window, identical
    when rendered twice, and (1 spp) the first 16 rows bit for bit against the oracle -- at 1024 spp the
    oracle's share is the tile row of tests/test_gpu_march_f64.py and the 8-rank form of test_gpu_multi.py."""
    W, H, key = 3840, 2160, 0x5C
    lens8, w8, scale, mask, efl, hf, vf, light = _c5_setup(pkg, lf, W, H)
    lf.find_sun_pos([light])
    lf.set_sun_from_flares(0, efl, 0.05)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()
    cnt = lf.counters()
    assert ray_budget(lf, cnt, W * H * spp * 8 * 46)
    assert cnt["rays_launched"] == cnt["rays_clipped_stop"] + cnt["rays_vignetted"] + cnt["rays_tir"] + cnt["rays_reached_scene"]
    assert cnt["rays_hit_light"] > 0
    x0, y0 = int(SUN_NS[0] * W) - 256, int(SUN_NS[1] * H) - 128
    g1, s1, a1 = (lf.read_tile(b, x0, y0, x0 + 512, y0 + 256) for b in (pkg.GHOST_BUFFER, pkg.STARBURST_BUFFER, pkg.SAMPLE_BUFFER))
    assert np.array_equal(a1, g1 + s1) and g1.max() > 0 and s1.max() > 0
    top = lf.read_tile(pkg.GHOST_BUFFER, 0, 0, W, 16)
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()
    assert np.array_equal(lf.read_tile(pkg.SAMPLE_BUFFER, x0, y0, x0 + 512, y0 + 256), a1)
    if spp > 1:
        # the configuration's full sample count is a culled launch: the WHOLE 3840 x 2160 x 1024 spp x 8-wavelength
        # frame against every path of every sample through the path tree (3.1e12 rays) -- identical, bit for bit
        assert lf.cull_info()["culled"]
        whole = lf.read_buffer(pkg.GHOST_BUFFER)
        lf.set_march_culling(0)
        try:
            lf.reset_counters()
            lf.trace_ghosts(spp, key)
            full = lf.counters()
            assert full["rays_launched"] == W * H * spp * 8 * 46 and full["rays_hit_light"] == cnt["rays_hit_light"]
            assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), whole)
            print(f"C5 at 1024 spp: the cull starts {cnt['rays_launched'] / full['rays_launched']:.4f} of the rays")
        finally:
            lf.set_march_culling(1)
        return
    sun = [(SUN_NS[0] - 0.5) * lens8["sensor_width_mm"] / efl, (SUN_NS[1] - 0.5) * lens8["sensor_width_mm"] * H / W / efl, -1.0]
    og, _ = lfo.geo_trace(lens8, W, H, 0, 16, spp, key, None, True, mask, sun, [1.0, 0.9, 0.5], 0.05,
                          n_threads=16, lambda_rgb=w8)
    assert np.array_equal(top, og[:16])


def test_c3_full_spp_frame(pkg, lf):
    """The benchmark frame itself (1080p, 256 spp, primary + 45 pairs x 3 wavelengths): exact ray
    budget, every ray accounted for, the executed-event counter below the per-path one by the
    shared-leg factor, identical when rendered twice, and an 8-row tile row around the sun bit for bit
    against the oracle (5e8 rays on the host's cores)."""
    W, H, spp, key = 1920, 1080, 256, 0x1e45f1a4e
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    efl = pkg.paraxial_efl(lens)
    sun = [(SUN_NS[0] - 0.5) * lens["sensor_width_mm"] / efl, (SUN_NS[1] - 0.5) * lens["sensor_width_mm"] * H / W / efl, -1.0]
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
    lf.set_ghost_pairs(None, True)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    cnt = lf.counters()
    executed = lf.executed_events()
    assert lf.cull_info()["culled"] and ray_budget(lf, cnt, W * H * spp * 3 * 46)
    assert cnt["rays_launched"] == cnt["rays_clipped_stop"] + cnt["rays_vignetted"] + cnt["rays_tir"] + cnt["rays_reached_scene"]
    # the culled march: the started paths of a sample share their common leg from the sensor (round 6: march_started_set)
    assert 0.4 * cnt["surface_events"] < executed < 0.9 * cnt["surface_events"]
    print(f"c3: {cnt['surface_events']:.4g} events of started paths, {executed:.4g} executed (the common leg once per sample)")
    whole = lf.read_buffer(pkg.GHOST_BUFFER)
    # ... and the same frame with EVERY path of every sample marched (the path tree, rounds 1-4): the whole
    # 1920 x 1080 x 256 spp frame is identical, bit for bit -- what the cull skipped added nothing
    lf.set_march_culling(0)
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    full_cnt, full_exec = lf.counters(), lf.executed_events()
    assert full_cnt["rays_launched"] == W * H * spp * 3 * 46 and full_cnt["rays_hit_light"] == cnt["rays_hit_light"]
    assert 3.0 < full_cnt["surface_events"] / full_exec < 5.0      # the path tree computes shared legs once
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), whole)
    assert cnt["rays_launched"] < 0.09 * full_cnt["rays_launched"]
    lf.set_march_culling(1)
    lf.trace_ghosts(spp, key)
    y0 = (int(SUN_NS[1] * H) // 8) * 8
    band = lf.read_tile(pkg.GHOST_BUFFER, 0, y0, W, y0 + 8)
    lf.trace_ghosts(spp, key)
    assert np.array_equal(lf.read_tile(pkg.GHOST_BUFFER, 0, y0, W, y0 + 8), band)
    og, _ = lfo.geo_trace(lens, W, H, y0, y0 + 8, spp, key, None, True, mask, sun, [1.0, 0.9, 0.5], 0.05,
                          n_threads=16)
    assert np.array_equal(band, og[y0:y0 + 8]) and band.max() > 0


def test_c4_full_size_frame_through_the_lens(pkg, lf):
    """BASELINE.json configs[3] on one GPU at its full size -- 3840 x 2160, 256 spp, a .dae scene with its own
    DirectionalLight (maxplanck.dae, 50 801 triangles: dragon.dae is absent from the reference checkout) --
    as ONE frame in which every sensor sample's primary path images the scene through the prescription and
    its ghost paths collect the sun (round 4): the pieces add up (sensor = (scene + ghost) + starburst), every
    lens sample and every ray of the march is accounted for, a second rendering is identical, the sun's image
    sits on the starburst's origin, and the mesh is in the way of the camera's rays."""
    import os
    W, H, spp, key = 3840, 2160, 256, 0x1e45f1a4e
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    efl = pkg.paraxial_efl(lens)
    hf = 2 * math.degrees(math.atan(0.5 * lens["sensor_width_mm"] / efl))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens)
    lf.set_ghost_pairs(None, True)
    lf.set_starburst_spectrum(None)
    lf.set_jitter_counter(key)
    camera, suns = lf.load_collada(os.path.join(pkg.DATA, "maxplanck.dae"))
    light = suns[0]
    lo, hi, n_prims = lf.scene_bounds()
    assert n_prims == 50801
    mid, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    to_sun = np.array(light[:3], float) - mid
    pos = mid - to_sun / np.linalg.norm(to_sun) * (1.5 * ext)        # behind the bust, looking at the sun
    lf.set_camera(pkg.aim_camera(pos, light[:3], SUN_NS, hf, vf), pos, hf, vf)
    lf.set_params(spp, 25.0, 1.0)
    lf.set_sampling(32, 0.05, 0.01, 100.0)                          # the reference's adaptive early-out
    lf.set_lens_camera(1, 0.001, 0.0)

    def frame():
        lf.reset_counters()
        lf.reset_scene_counters()
        lf.find_sun_pos([light])
        lf.set_sun_from_flares(0, efl, 0.05)
        lf.render_scene_term()
        lf.trace_ghosts(spp, key)
        lf.render_flare_layer()

    frame()
    cnt, sc = lf.counters(), lf.scene_counters()
    assert ray_budget(lf, cnt, W * H * spp * 3 * 46)
    assert cnt["rays_launched"] == cnt["rays_clipped_stop"] + cnt["rays_vignetted"] + cnt["rays_tir"] + cnt["rays_reached_scene"]
    # the lens camera: between one batch (the adaptive test stops a dark or a flat pixel after 32 samples) and
    # all 256 samples per pixel were marched; about a quarter leaves the pentagon-stopped lens; every one that
    # does is a BVH query, a hit adds a shadow ray towards the sun
    assert W * H * 32 <= sc["lens_samples"] <= W * H * spp
    assert 0.2 < sc["lens_left"] / sc["lens_samples"] < 0.32
    assert sc["lens_left"] <= sc["rays"] <= 4 * sc["lens_left"] and sc["isects"] > 0
    x0, y0 = int(SUN_NS[0] * W) - 256, int(SUN_NS[1] * H) - 128
    s1, g1, t1, a1 = (lf.read_tile(b, x0, y0, x0 + 512, y0 + 256) for b in
                      (pkg.SCENE_BUFFER, pkg.GHOST_BUFFER, pkg.STARBURST_BUFFER, pkg.SAMPLE_BUFFER))
    assert np.array_equal(a1, (s1 + g1) + t1) and g1.max() > 0 and t1.max() > 0
    # the bust is in the picture: camera rays hit it (every hit casts a shadow ray towards the sun) -- seen from
    # behind it is in its own shadow, so its pixels are dark like the sky; only its rim may catch light
    assert sc["rays"] > 1.05 * sc["lens_left"]
    # the sun's image (primary path) is where the starburst is built
    fl = lf.get_flares()
    ghost_row = lf.read_tile(pkg.GHOST_BUFFER, 0, y0, W, y0 + 256).sum(axis=-1)
    ys, xs = np.nonzero(ghost_row > 0.5 * ghost_row.max())
    assert abs(xs.mean() - fl["origins"][0][0] * W) < 6 and abs(ys.mean() + y0 - fl["origins"][0][1] * H) < 6
    # ... and the tile row of the 4K frame that holds it, at the configuration's full 256 spp, against the float32
    # oracle bit for bit (the sun = what the hand-over made of the scene's own DirectionalLight)
    nx, ny = fl["origins"][0]
    sun = [(nx - 0.5) * lens["sensor_width_mm"] / efl, (ny - 0.5) * lens["sensor_width_mm"] * H / W / efl, -1.0]
    yt = (int(ny * H) // 8) * 8
    band = lf.read_tile(pkg.GHOST_BUFFER, 0, yt, W, yt + 8)
    og, _ = lfo.geo_trace(lens, W, H, yt, yt + 8, spp, key, None, True, mask, sun, list(fl["radiance"][0]), 0.05,
                          n_threads=16)
    assert band.max() > 0 and np.array_equal(band, og[yt:yt + 8])
    frame()
    assert np.array_equal(lf.read_tile(pkg.SAMPLE_BUFFER, x0, y0, x0 + 512, y0 + 256), a1)
    assert lf.scene_counters() == sc and lf.counters() == cnt
    lf.set_lens_camera(0)
    lf.set_params(1, 25.0, 1.0)
