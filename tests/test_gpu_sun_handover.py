"""Sun hand-over from the scene to the march (lf_set_sun_from_flares): the in-frame
DirectionalLight that lf_find_sun_pos projects (pathtracer.cpp:32-64, camera.cpp:245-273) becomes
the lens-space light of the geometric march, so that ghosts and starburst agree about where the sun
is.  No reference counterpart for the march side; the screen-space side is pinned by the golden
frames (test_gpu_flare_parity.py)."""
import math
import os

import numpy as np
import pytest

from goldenlib import ray_budget, load_texels

SUN_NS = (0.521445, 0.517156)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_paraxial_focal_lengths(pkg):
    """lf_paraxial_efl (host arithmetic, no device): the bench lens is the 50 mm double-Gauss its
    file says it is, and the thin lens obeys the lensmaker's equation."""
    dg = pkg.load_lens_file("dgauss11.lens")
    assert pkg.paraxial_efl(dg, 1) == pytest.approx(50.358, abs=2e-3)
    assert pkg.paraxial_efl(dg, 0) > pkg.paraxial_efl(dg, 1) > pkg.paraxial_efl(dg, 2)   # normal dispersion
    tl = pkg.load_lens_file("thinlens.lens")
    n = float(tl["ior"][1, 0])
    R1, R2, d = float(tl["radius"][0]), float(tl["radius"][1]), float(tl["thickness"][0])
    f = 1.0 / ((n - 1) * (1 / R1 - 1 / R2 + (n - 1) * d / (n * R1 * R2)))
    assert pkg.paraxial_efl(tl, 1) == pytest.approx(f, rel=1e-6)


def _setup(pkg, lf, W, H, lens, pos, sun_point):
    efl = pkg.paraxial_efl(lens)
    hf = 2 * math.degrees(math.atan(0.5 * lens["sensor_width_mm"] / efl))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_frame(W, H)
    lf.set_params(1, 25.0, 1.0)
    mask = pkg.load_aperture_png("pentbig500_14.png")
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels("octagonbokeh.png"))
    lf.set_lens(lens)
    lf.set_camera(pkg.aim_camera(pos, sun_point, SUN_NS, hf, vf), pos, hf, vf)
    lf.set_jitter_counter(7)
    return efl


@pytest.mark.gpu
def test_ghost_image_of_the_sun_sits_on_the_starburst_origin(pkg):
    """The primary (no reflection) path images the sun: its centroid in the ghost buffer must be the
    flare origin the starburst is built around (ceil(ns * (W, H)), pathtracer.cpp:921-922)."""
    lf = pkg.LensFlare(0)
    W, H = 384, 216
    lens = pkg.load_lens_file("dgauss11.lens")
    _setup(pkg, lf, W, H, lens, [0.0, 0.0, 0.0], [3.0, 2.0, -40.0])
    lf.find_sun_pos([[3.0, 2.0, -40.0, 1.0, 0.9, 0.5]])
    fl = lf.get_flares()
    assert fl["n"] == 1
    assert fl["origins"][0] == pytest.approx(SUN_NS, abs=1e-9)
    lf.set_sun_from_flares(0, 0.0, 0.01)
    lf.set_ghost_pairs([(-1, -1)], False)                 # the primary path alone
    lf.trace_ghosts(64, 11)
    g = lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2)
    assert g.max() > 0
    ys, xs = np.mgrid[0:H, 0:W]
    cx, cy = (g * (xs + 0.5)).sum() / g.sum(), (g * (ys + 0.5)).sum() / g.sum()
    ox, oy = math.ceil(SUN_NS[0] * W), math.ceil(SUN_NS[1] * H)
    assert abs(cx - ox) < 1.5 and abs(cy - oy) < 1.5, (cx, cy, ox, oy)
    # ... and that is where the starburst peaks
    lf.set_ghost_pairs(None, True)
    lf.trace_ghosts(4, 11)
    lf.render_flare_layer()
    star = lf.read_buffer(pkg.STARBURST_BUFFER).sum(axis=2)
    py, px = np.unravel_index(np.argmax(star), star.shape)
    assert abs(px - ox) <= 1 and abs(py - oy) <= 1
    # the radiance the march uses is the flare's
    assert fl["radiance"][0] == pytest.approx([1.0, 0.9, 0.5])
    lf.close()


@pytest.mark.gpu
def test_c4_shaped_frame_scene_light_feeds_the_march(pkg):
    """BASELINE configs[3] in one piece, on one GPU at low spp: a .dae scene (the reference's own
    sun scene dae/pyramid.dae) -> its DirectionalLight -> lf_find_sun_pos -> lf_set_sun_from_flares
    -> scene term + geometric ghosts at 4K -> composed frame.  Checked by properties: the flare is
    where the light projects, the ray budget is exact, composition is scene + ghost + starburst, and
    moving the camera moves ghosts and starburst together."""
    lf = pkg.LensFlare(0)
    W, H, spp = 3840, 2160, 1
    lens = pkg.load_lens_file("dgauss11.lens")
    camera, suns = lf.load_collada(os.path.join(pkg.DATA, "pyramid.dae"))
    assert len(suns) >= 1
    pos = np.array(camera["pos"] if camera else [0, 0, 0], float)
    _setup(pkg, lf, W, H, lens, pos, suns[0][:3])
    lf.find_sun_pos(suns)
    fl = lf.get_flares()
    assert fl["n"] >= 1 and fl["origins"][0] == pytest.approx(SUN_NS, abs=1e-6)
    lf.set_sun_from_flares(0, 0.0, 0.005)    # a 0.005 rad lobe: the sun's image is ~27 px in radius at 4K
    lf.set_ghost_pairs(None, True)
    lf.render_scene_term()
    lf.reset_counters()
    lf.trace_ghosts(spp, 5)
    lf.render_flare_layer()
    cnt = lf.counters()
    assert ray_budget(lf, cnt, W * H * spp * 3 * 46)
    assert cnt["rays_launched"] == cnt["rays_clipped_stop"] + cnt["rays_vignetted"] + cnt["rays_tir"] + \
        cnt["rays_reached_scene"]
    assert cnt["rays_hit_light"] > 0
    ox, oy = math.ceil(SUN_NS[0] * W), math.ceil(SUN_NS[1] * H)
    x0, y0 = ox - 200, oy - 120
    ghost = lf.read_tile(pkg.GHOST_BUFFER, x0, y0, x0 + 400, y0 + 240)
    star = lf.read_tile(pkg.STARBURST_BUFFER, x0, y0, x0 + 400, y0 + 240)
    samp = lf.read_tile(pkg.SAMPLE_BUFFER, x0, y0, x0 + 400, y0 + 240)
    assert ghost.max() > 0 and star.max() > 0
    scene = samp - ghost - star                      # >= 0 and bounded: the scene term of those pixels
    assert scene.min() > -1e-9 * max(1.0, samp.max())
    # the ghost energy of the window (dominated by the primary image of the sun; the pair ghosts are
    # ~1e-3 of it) is centred on the flare origin
    g2 = ghost.sum(axis=2)
    ys, xs = np.mgrid[0:g2.shape[0], 0:g2.shape[1]]
    gx, gy = (g2 * (xs + 0.5)).sum() / g2.sum(), (g2 * (ys + 0.5)).sum() / g2.sum()
    assert abs(gx + x0 - ox) < 10 and abs(gy + y0 - oy) < 10, (gx + x0, gy + y0, ox, oy)
    lf.close()


@pytest.mark.gpu
def test_refocused_lens_still_puts_the_sun_where_the_starburst_is(pkg):
    """lf_focus_lens moves the sensor out of the focal plane: a distant point's image then sits where its CHIEF ray
    lands -- focal length + sensor shift x the chief ray's exit slope, not the focal length alone.
    lf_set_sun_from_flares (efl <= 0) uses that scale, so the sun's (now defocused) image stays centred on the
    flare origin the starburst is built around; with the plain focal length it would sit 8 % further out."""
    lf = pkg.LensFlare(0)
    W, H = 384, 216
    lens = pkg.load_lens_file("dgauss11.lens")
    sun_ns = (0.78, 0.70)                                  # well off the axis: a scale error shows
    efl = pkg.paraxial_efl(lens)
    hf = 2 * math.degrees(math.atan(0.5 * lens["sensor_width_mm"] / efl))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_frame(W, H)
    lf.set_params(1, 25.0, 1.0)
    lf.set_aperture(pkg.APERTURE_STARBURST, pkg.load_aperture_png("pentbig500_14.png"))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels("octagonbokeh.png"))
    lf.set_lens(lens)
    pos, sun_point = [0.0, 0.0, 0.0], [3.0, 2.0, -40.0]
    lf.set_camera(pkg.aim_camera(pos, sun_point, sun_ns, hf, vf), pos, hf, vf)
    lf.set_jitter_counter(7)
    lf.set_ghost_pairs([(-1, -1)], False)                  # the primary path alone: the sun's image
    lf.find_sun_pos([sun_point + [1.0, 0.9, 0.5]])
    fl = lf.get_flares()
    assert fl["n"] == 1

    def centroid(efl_arg):
        lf.set_sun_from_flares(0, efl_arg, 0.02)
        lf.trace_ghosts(256, 11)
        g = lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=-1)
        ys, xs = np.nonzero(g > 0.2 * g.max())
        w = g[ys, xs]
        return np.array([(xs * w).sum() / w.sum() + 0.5, (ys * w).sum() / w.sum() + 0.5])

    want = np.array([fl["origins"][0][0] * W, fl["origins"][0][1] * H])
    assert np.abs(centroid(0.0) - want).max() < 1.5        # in the focal plane: as before
    shift = lf.focus_lens(500.0) - float(lens["thickness"][-1])
    assert 4.0 < shift < 8.0                               # focusing at half a metre moves the sensor ~5.7 mm back
    auto = centroid(0.0)
    plain = centroid(efl)                                  # the focal length alone: what rounds 2-3 would have used
    assert np.abs(auto - want).max() < 2.5, (auto, want)
    assert np.linalg.norm(plain - want) > 3 * np.linalg.norm(auto - want) and np.linalg.norm(plain - want) > 6.0
    lf.close()
