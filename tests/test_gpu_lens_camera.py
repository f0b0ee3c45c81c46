"""The lens camera of the scene term (round 4; lf_set_lens_camera): the scene imaged THROUGH the
prescription -- the north star's "for each sensor sample, march a ray through the lens ... accumulate
radiance into the sensor buffer" at the call site the reference has for it, camera->generate_ray in
the sample loop of raytrace_pixel (pathtracer.cpp:841-850; its own lens camera is a stub,
camera_lens.cpp:22-30).

PARITY STATUS: unpinned by construction (no reference lens exists).  What anchors it:
  * the reference's own pinhole frames: with the stop closed to a pinhole the lens frame, divided by
    the analytic cos^4 law, converges to frames the REAL reference rendered (tests/golden/s96x64_spheres,
    c96x72_pyramid_dae) -- same pixel <-> direction mapping as find_sun_pos / lf_set_sun_from_flares;
  * an analytic known answer: a point source off the focal plane renders a blur disc of the
    circle-of-confusion diameter the thin-lens equation predicts;
  * the float32 oracle (bit-exact exit rays => pixels to 1e-9) and the independent float64 tracer
    (1e-4 + a per-pixel bound on the fragile rays' weight), composed with the pinned scene oracle.
"""
import math

import numpy as np
import pytest

from goldenlib import Case, load_texels
from oracle import lfo

pytestmark = pytest.mark.gpu
KEY = 0x1e45f1a4e


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture()
def lf(pkg):
    ctx = pkg.LensFlare(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def sqrt_table(pkg):
    """the float32 oracle follows the device's v_sqrt_f32 and v_rcp_f32 through their measured deviation tables"""
    ctx = pkg.LensFlare(0)
    t = lfo.geo_follow_device(ctx)
    lfo.geo_follow_device(None)
    ctx.close()
    return t


SPHERES = [(0, -101.0, -6, 100.0, "d", 0.6, 0.6, 0.55), (-0.9, -0.4, -5, 0.6, "d", 0.8, 0.2, 0.2),
           (0.7, -0.55, -4.2, 0.45, "d", 0.2, 0.7, 0.3), (0.1, 0.35, -6.5, 0.5, "e", 1.5, 1.2, 0.4),
           (1.6, 0.2, -7.0, 0.8, "d", 0.3, 0.3, 0.9)]
TRIS = [(-2.0, -1.0, -8.0, 2.0, -1.0, -8.0, 0.0, 2.0, -9.0, 0, 0, 1, 0, 0, 1, 0, 0, 1, "d", 0.7, 0.7, 0.2)]
LIGHTS = [[0.0, 0.3, 0.8, 0.52, 2.0, 1.8, 1.5], [1.0, -1.0, 2.0, -3.0, 4.0, 4.0, 5.0]]


def unit_lights(lights):
    out = []
    for l in lights:
        if l[0] == 0.0:
            p = np.array(l[1:4])
            out.append([0.0] + (p / np.sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2])).tolist() + list(l[4:7]))
        else:
            out.append(list(l))
    return out


def look_at(pos, target):
    """c2w (row-major 3x3) of a camera at pos looking at target, y up: columns = right, up, back"""
    f = np.array(target, float) - np.array(pos, float)
    f /= np.linalg.norm(f)
    r = np.cross(f, [0, 1, 0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    return np.stack([r, u, -f], axis=1).reshape(-1)


def setup_scene_frame(pkg, lf, lens, mask, W, H, ns_aa, c2w, pos, spheres=SPHERES, tris=TRIS, lights=LIGHTS):
    efl = pkg.paraxial_efl(lens)
    hf = 2 * math.degrees(math.atan(0.5 * float(lens["sensor_width_mm"]) / efl))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_frame(W, H)
    lf.set_params(ns_aa, 25.0, 1.0)
    lf.set_sampling(1 << 20, 0.05, 0.01, 100.0)      # no adaptive early-out
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens)
    lf.set_camera(c2w, pos, hf, vf)
    lf.set_scene(list(spheres), list(tris), unit_lights(lights))
    lf.set_jitter_counter(KEY)
    return hf, vf


def compose(lens, mask, W, H, ns, c2w, pos, wpm, z_ref, exposure, samples, weight_col, spheres, tris, lights):
    """pixels = sum_s exposure * weight_s * L(exit ray_s) / (ns + 1): the device's composition with
    the scene oracle's radiance (lf_scene_oracle.c, pinned by the reference's frames)."""
    n_pix = samples.shape[0]
    o, d = lfo.lens_exit_to_world(samples[..., 0:3], samples[..., 3:6], c2w, pos, wpm, z_ref)
    rays = np.concatenate([o, d, np.full(o.shape[:-1] + (1,), 0.01), np.full(o.shape[:-1] + (1,), 100.0)], -1)
    alive = samples[..., weight_col] > 0
    L = np.zeros(o.shape)
    L[alive] = lfo.scene_radiance_rays(spheres, tris, unit_lights(lights), rays[alive])
    w = samples[..., weight_col].astype(np.float64) * exposure
    return (L * w[..., None]).sum(axis=1) / (ns + 1), L, alive


def test_lens_frame_equals_the_float32_oracle(pkg, lf, sqrt_table):
    """Every pixel of a lens-imaged frame (double Gauss, pentagon stop, 9 samples per pixel) = the
    float32 oracle's primary paths (bit-exact exit rays) carried into the scene and shaded by the scene
    oracle, to 1e-9; calibration and entrance pupil as the host computes them; both modes."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns, wpm = 48, 32, 9, 0.004
    pos = [0.3, 0.2, 1.0]
    c2w = look_at(pos, [0.0, -0.2, -5.5])
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.reset_scene_counters()
    lf.render_scene_term()
    got = lf.read_buffer(pkg.SCENE_BUFFER)
    info = lf.lens_camera()
    cnt = lf.scene_counters()
    lfo.geo_follow_device(sqrt_table)
    try:
        exposure = lfo.lens_exposure(lens, W, mask)
        z_ref, _ = pkg.paraxial_entrance_pupil(lens)
        assert info["mode"] == 1 and info["world_per_mm"] == wpm
        assert abs(info["exposure"] - exposure) <= 1e-12 * exposure
        assert info["entrance_pupil_z_mm"] == z_ref
        smp = lfo.geo_lens_samples(lens, W, H, ns, KEY, 1, np.arange(W * H), mask)
        want, L, alive = compose(lens, mask, W, H, ns, c2w, pos, wpm, z_ref, exposure, smp, 6, SPHERES, TRIS, LIGHTS)
        want = want.reshape(H, W, 3)
        assert cnt["lens_samples"] == W * H * ns and cnt["lens_left"] == int(alive.sum())
        assert 0.1 < alive.mean() < 0.5                  # the pentagon clips most of the rear element's disc
        assert (want.max(axis=-1) > 0.02).mean() > 0.5   # not a dark frame
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
        assert err.max() <= 1e-9, err.max()
        # mode 2: one ray per wavelength, channel c = wavelength c (3 wavelengths: the identity weights)
        lf.set_lens_camera(2, wpm, 0.0)
        lf.render_scene_term()
        got2 = lf.read_buffer(pkg.SCENE_BUFFER)
        want2 = np.zeros((H * W, 3))
        for l in range(3):
            sl = lfo.geo_lens_samples(lens, W, H, ns, KEY, l, np.arange(W * H), mask)
            wl, _, _ = compose(lens, mask, W, H, ns, c2w, pos, wpm, z_ref, exposure, sl, 6, SPHERES, TRIS, LIGHTS)
            want2[:, l] = wl[:, l]
        want2 = want2.reshape(H, W, 3)
        err2 = np.abs(got2 - want2) / np.maximum(np.abs(want2), 1e-12)
        assert err2.max() <= 1e-9, err2.max()
        assert np.abs(got2 - got).max() > 1e-6           # dispersion is visible
        assert np.array_equal(got2[..., 1], got[..., 1])  # the reference wavelength IS wavelength 1
    finally:
        lfo.geo_follow_device(None)
    # a fixed exposure is taken as given
    lf.set_lens_camera(1, wpm, 2.0 * exposure)
    lf.render_scene_term()
    assert np.allclose(lf.read_buffer(pkg.SCENE_BUFFER), 2.0 * got, rtol=1e-12, atol=0)
    # back to the pinhole: the reference's camera again
    lf.set_lens_camera(0)
    lf.render_scene_term()
    pin = lf.read_buffer(pkg.SCENE_BUFFER)
    assert np.abs(pin - got).max() > 1e-3


def test_lens_frame_against_the_independent_float64_tracer(pkg, lf):
    """The same frame by the float64 textbook tracer (oracle/lf_geo_f64.c: no shared recipe, no sqrt
    table): every pixel within 1e-4 relative + the summed potential weight of its fragile samples."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns, wpm = 48, 32, 16, 0.004
    pos = [0.3, 0.2, 1.0]
    c2w = look_at(pos, [0.0, -0.2, -5.5])
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    got = lf.read_buffer(pkg.SCENE_BUFFER).reshape(-1, 3)
    info = lf.lens_camera()
    smp = lfo.g64_lens_samples(lens, W, H, ns, KEY, 1, np.arange(W * H), mask)
    want, L, alive = compose(lens, mask, W, H, ns, c2w, pos, wpm, info["entrance_pupil_z_mm"], info["exposure"],
                             smp, 6, SPHERES, TRIS, LIGHTS)
    # a fragile sample may go either way in float32: it can add or remove at most its potential weight
    # times the brightest radiance any ray of the frame found
    frag = (smp[..., 8] > 0) * smp[..., 7] * info["exposure"] * L.max() / (ns + 1)
    allow = frag.sum(axis=1)[:, None]
    dev = np.abs(got - want)
    # the lens part of a sample (exit point, direction, weight) agrees to ~1e-5 (tests/test_lens_camera_cpu.py,
    # tests/test_geo_rays_vs_f64.py); the SCENE can amplify the last bits of a direction where radiance changes
    # quickly (a terminator, a silhouette), so: 1e-4 (+ allowance) on at least 99.5 % of the values -- measured:
    # all of them -- and 2e-3 on every one
    assert (dev <= 2e-3 * np.abs(want) + allow + 1e-13).all(), (dev - 2e-3 * np.abs(want) - allow).max()
    assert (dev <= 1e-4 * np.abs(want) + allow + 1e-13).mean() > 0.995
    plain = dev <= 1e-4 * np.abs(want) + 1e-13
    assert plain.mean() > 0.99, plain.mean()             # the allowance is the exception
    lit = want > 1e-3
    print(f"lens frame vs float64: {int(lit.sum())} lit values, median rel dev "
          f"{np.median(dev[lit] / want[lit]):.2e}, max {np.max(dev[lit & plain] / want[lit & plain]):.2e}, "
          f"{int((~plain).sum())} values inside the fragile allowance")


def pinhole_mask(n=501, radius_texels=25):
    yy, xx = np.mgrid[0:n, 0:n]
    return (((xx - n // 2) ** 2 + (yy - n // 2) ** 2) <= radius_texels ** 2).astype(np.float32)


@pytest.mark.parametrize("name", ["s96x64_spheres", "c96x72_pyramid_dae"])
def test_closed_stop_converges_to_the_reference_pinhole_frames(pkg, lf, name):
    """Stop closed to 10 % of its housing (f/21): every sample crosses the lens through (almost) one point,
    the entrance pupil's centre = the camera position -- a pinhole.  Flat-fielded (divided by the frame
    the same lens, samples and exposure give of a uniformly bright enclosure: its relative illumination,
    vignetting and transmission) the lens frame must be the frame the REAL reference rendered with its
    pinhole camera (golden fixture) -- the same pixel <-> direction mapping as find_sun_pos and
    lf_set_sun_from_flares use -- up to the lens' distortion (< 0.3 % of the field) at silhouettes and
    the pixel jitter of both; inside the field this double Gauss passes."""
    from test_gpu_scene_term import scene_lights
    case = Case(name)
    m = case.meta
    lens = pkg.load_lens_file("dgauss11.lens")
    efl = pkg.paraxial_efl(lens)
    # the sensor that gives this prescription the golden frame's field of view
    lens["sensor_width_mm"] = np.float32(2.0 * efl * math.tan(math.radians(m["hFov"]) / 2))
    mask = pinhole_mask()
    W, H, ns = case.W, case.H, 64
    sc = m["scene"]
    spheres = [(1e4, 1e4, 1e4, 1.0, "d", 0.5, 0.5, 0.5)] + [tuple(s) for s in sc["spheres"]]
    lf.set_frame(W, H)
    lf.set_params(ns, 25.0, 1.0)
    lf.set_sampling(1 << 20, 0.05, 0.01, 100.0)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens)
    # the samples aim at the image of the open part of the stop, with room for the pupil's aberration off
    # the axis (a disc that does not cover the real pupil of a field point would bias its pixels dark)
    lf.aim_at_exit_pupil(1.6)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.set_jitter_counter(KEY)
    lf.set_lens_camera(1, 1e-4, 0.0)   # a 10 cm lens in a scene of ~10 units: 1 unit = 10 m
    # the flat field: the camera inside a uniformly bright sphere
    lf.set_scene([tuple(m["cam_pos"]) + (50.0, "e", 1.0, 1.0, 1.0)], [], [])
    lf.reset_scene_counters()
    lf.render_scene_term()
    flat_field = lf.read_buffer(pkg.SCENE_BUFFER)[..., 0] * (ns + 1) / ns
    cnt = lf.scene_counters()
    lf.set_scene(spheres, [tuple(t) for t in sc["tris"]], scene_lights(case))
    lf.render_scene_term()
    lens_frame = lf.read_buffer(pkg.SCENE_BUFFER) * (ns + 1) / ns   # undo the reference's 1 / (ns_aa + 1)
    lf.set_lens_camera(0)
    lf.render_scene_term()
    pin = lf.read_buffer(pkg.SCENE_BUFFER) * (ns + 1) / ns          # the device's pinhole, same sample count
    # field angle of a pixel centre, as Camera::generate_ray maps it
    ex, ey = math.tan(math.radians(m["hFov"]) / 2), math.tan(math.radians(m["vFov"]) / 2)
    xs = (np.arange(W) + 0.5) / W * 2 - 1
    ys = (np.arange(H) + 0.5) / H * 2 - 1
    tan2 = (xs[None, :] * ex) ** 2 + (ys[:, None] * ey) ** 2
    cos4 = 1.0 / (1.0 + tan2) ** 2
    # where the lens passes the field at all (the double Gauss was drawn for a 36 mm sensor)
    passes = cnt["lens_left"] / cnt["lens_samples"]
    assert passes > 0.12, passes
    inside = np.sqrt(tan2) < math.tan(math.radians(19.0))
    # the relative illumination: 1 on the axis by calibration; off the axis it follows cos^4 of the IMAGE-side
    # chief-ray angle (the exit pupil sits farther from the sensor than the focal length: a little brighter
    # than cos^4 of the field angle).  Per pixel it carries the noise of 64 pass / block decisions, which
    # the flat-fielding below cancels sample by sample; here: means over rings of the field.
    centre = flat_field[H // 2 - 6:H // 2 + 6, W // 2 - 6:W // 2 + 6].mean()
    assert abs(centre - 1.0) < 0.04, centre
    ring = np.minimum((np.sqrt(tan2) / math.tan(math.radians(19.0)) * 5).astype(int), 5)
    rel = np.array([(flat_field[ring == k] / cos4[ring == k]).mean() for k in range(5)])
    print(f"{name}: relative illumination / cos^4(field angle), rings out to 19 deg: {np.round(rel, 3)}; "
          f"{passes:.2f} of the samples leave the lens")
    assert (rel > 0.93).all() and (rel < 1.25).all() and rel[4] > rel[0]
    flat = lens_frame / np.maximum(flat_field, 1e-6)[..., None]
    # smooth pixels only: at silhouettes a 1-pixel shift (distortion, jitter) is a large difference
    lum = pin.sum(axis=-1)
    gy, gx = np.gradient(lum)
    smooth = inside & (np.hypot(gx, gy) < 0.05 * (lum + 0.05)) & (lum > 0.05)
    assert smooth.mean() > 0.08, smooth.mean()
    ratio = flat[smooth].sum(axis=-1) / lum[smooth]
    print(f"{name}: {int(smooth.sum())} smooth pixels in the passed field, flat-fielded lens / pinhole: median "
          f"{np.median(ratio):.4f}, 5..95 % {np.percentile(ratio, 5):.4f} .. {np.percentile(ratio, 95):.4f}")
    assert abs(np.median(ratio) - 1.0) < 0.01
    assert np.percentile(ratio, 5) > 0.97 and np.percentile(ratio, 95) < 1.03
    # ... and the device's pinhole frame is the reference's (the golden, rendered by the real reference
    # with ns_aa = 2 or 3 and its own jitter): so the lens frame converges to the reference's frame
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_jitter_mt19937(5489, None)
    lf.set_paraxial_lens()
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.find_sun_pos(m["lights"])
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    full = lf.read_buffer(pkg.SAMPLE_BUFFER)
    assert (np.abs(full - case.sample) <= 1e-9 * np.abs(case.sample)).all()   # pinned, as test_gpu_scene_term
    ref_scene = lf.read_buffer(pkg.SCENE_BUFFER) * (m["ns_aa"] + 1) / m["ns_aa"]
    r2 = flat[smooth].sum(axis=-1) / np.maximum(ref_scene[smooth].sum(axis=-1), 1e-9)
    print(f"{name}: flat-fielded lens / the reference's own frame: median {np.median(r2):.4f}")
    assert abs(np.median(r2) - 1.0) < 0.02, np.median(r2)


def paraxial_ray(lens, lam, y, u):
    """(height, angle) behind the last interface of a paraxial ray that meets the first vertex at height
    y with angle u (the reference's T / R operators, pathtracer.cpp:527-533)."""
    n1 = 1.0
    for k in range(int(lens["n"])):
        if k != int(lens["stop"]):
            R = float(lens["radius"][k])
            c = 0.0 if R == 0 else 1.0 / R
            n2 = float(lens["ior"][lam][k])
            u = c * (n1 - n2) / n2 * y + n1 / n2 * u
            n1 = n2
        if k + 1 < int(lens["n"]):
            y = y + float(lens["thickness"][k]) * u
    return y, u


def test_blur_disc_has_the_thin_lens_circle_of_confusion(pkg, lf):
    """Known answer: a small emitter at distance d_o, the lens focused at d_f != d_o.  The geometric
    blur disc on the sensor is where the marginal rays of the emitter's cone land: paraxially a disc of
    diameter c = 2 |y + u v_f| for the ray through the rim of the aperture (y, u behind the lens, v_f the
    sensor distance) -- for a thin lens the textbook A |v_o - v_f| / v_o.  A slow singlet (f/10), so that
    spherical aberration (~ h^2) stays at the per cent level of the defocus."""
    lens = pkg.load_lens_file("thinlens.lens")
    lens["semi_aperture"] = np.array([2.5, 2.5], np.float32)
    mask = np.ones((8, 8), np.float32)
    W, H, ns, wpm = 200, 200, 1024, 0.001           # 1 unit = 1 m
    lens["sensor_width_mm"] = np.float32(4.0)       # 0.02 mm pixels
    pos, c2w = [0.0, 0.0, 0.0], np.eye(3).reshape(-1)
    d_o, d_f, r_e = 0.3, 3.0, 0.00025               # metres in front of the camera; the emitter's radius
    emitter = [(0.0, 0.0, -d_o, r_e, "e", 5.0, 5.0, 5.0)]
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos, spheres=emitter, tris=[], lights=[])
    # no stop: the camera position is the front vertex, object distances count from there
    v_o = lf.focus_lens(d_o * 1000.0)               # where the emitter is imaged
    v_f = lf.focus_lens(d_f * 1000.0)               # where the sensor is
    assert v_o > v_f > 0
    y0, u0 = paraxial_ray(lens, 1, d_o * 1000.0 * 1e-3, 1e-3)
    assert abs(-y0 / u0 - v_o) < 1e-3 * v_o          # the helper and lf_focus_lens agree about the image
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    img = lf.read_buffer(pkg.SCENE_BUFFER).sum(axis=-1)
    lit = img > 0.04 * img.max()
    pitch = 4.0 / W
    diameter = 2.0 * math.sqrt(lit.sum() / math.pi) * pitch
    h = 2.5
    ym, um = paraxial_ray(lens, 1, h, h / (d_o * 1000.0))    # the marginal ray of the emitter's cone
    want = 2.0 * abs(ym + um * v_f)
    own = 2.0 * (r_e * 1000.0) * v_f / (d_o * 1000.0)        # the emitter's own image, convolved in
    print(f"blur disc: measured {diameter:.4f} mm, circle of confusion {want:.4f} mm (+ emitter image {own:.4f} mm), "
          f"thin-lens A |v_o - v_f| / v_o = {2 * h * (v_o - v_f) / v_o:.4f}, v_o {v_o:.3f} v_f {v_f:.3f}")
    assert abs(2 * h * (v_o - v_f) / v_o - want) < 0.03 * want   # the textbook form, up to the lens' thickness
    assert want - 0.03 * want < diameter < want + own + 0.03 * want, (diameter, want, own)
    # centred on the axis, round
    ys, xs = np.nonzero(lit)
    assert abs(xs.mean() - (W - 1) / 2) < 1.0 and abs(ys.mean() - (H - 1) / 2) < 1.0
    assert abs((xs.max() - xs.min()) - (ys.max() - ys.min())) <= 2
    # in focus the same emitter is a few pixels
    lf.focus_lens(d_o * 1000.0)
    lf.render_scene_term()
    sharp = lf.read_buffer(pkg.SCENE_BUFFER).sum(axis=-1)
    assert (sharp > 0.04 * sharp.max()).sum() < 0.05 * lit.sum()


def test_close_focus_counts_from_the_entrance_pupil(pkg, lf):
    """Camera::focalDistance is measured from the camera position, which the lens camera puts at the centre of
    the entrance pupil -- 19.95 mm BEHIND the double Gauss's first vertex, from which lf_focus_lens measures.  An emitter
    5 focal lengths away: focused with lf_focus_lens_from_pupil (what the drop-in passes focalDistance to) it is a few
    pixels; focused to the same number counted from the first vertex the sensor sits more than a millimetre off and the
    disc is several times larger.  (At 10 focal lengths the 0.24 mm between the two sensor positions is what this
    prescription's spherical aberration moves the best focus by, towards the lens too: both spots measure 4 x 4 pixels.)"""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns, wpm = 200, 200, 1024, 0.001
    lens["sensor_width_mm"] = np.float32(4.0)       # 0.02 mm pixels around the axis
    d_o, r_e = 0.25, 0.0001
    emitter = [(0.0, 0.0, -d_o, r_e, "e", 5.0, 5.0, 5.0)]
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, np.eye(3).reshape(-1), [0.0, 0.0, 0.0], spheres=emitter, tris=[], lights=[])
    z_ep, _ = pkg.paraxial_entrance_pupil(lens)
    assert 15.0 < z_ep < 25.0
    v_pupil = lf.focus_lens_from_pupil(d_o * 1000.0)
    assert v_pupil == pytest.approx(lf.focus_lens(d_o * 1000.0 - z_ep), rel=1e-6)
    lf.focus_lens_from_pupil(d_o * 1000.0)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    sharp = lf.read_buffer(pkg.SCENE_BUFFER).sum(axis=-1)
    v_vertex = lf.focus_lens(d_o * 1000.0)          # the same number, counted from the first vertex
    assert v_pupil - v_vertex > 1.0                 # the sensor is more than a millimetre off
    lf.render_scene_term()
    soft = lf.read_buffer(pkg.SCENE_BUFFER).sum(axis=-1)
    a_sharp, a_soft = (sharp > 0.04 * sharp.max()).sum(), (soft > 0.04 * soft.max()).sum()
    print(f"close focus at {d_o} m: entrance pupil {z_ep:.2f} mm behind the first vertex, sensor {v_pupil:.3f} vs {v_vertex:.3f} mm, "
          f"image area {a_sharp} px focused from the pupil, {a_soft} px focused from the vertex")
    assert sharp.max() > 0 and a_sharp < 0.25 * a_soft
    lf.set_lens_camera(0)


def test_lens_camera_preconditions(pkg, lf):
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    lf.set_frame(32, 16)
    lf.set_params(2, 25.0, 1.0)
    lf.set_camera(np.eye(3).reshape(-1), [0, 0, 0], 40.0, 30.0)
    lf.set_scene(SPHERES, TRIS, unit_lights(LIGHTS))
    lf.set_lens_camera(1, 0.001, 0.0)
    with pytest.raises(pkg.LensFlareError, match="prescription"):
        lf.set_jitter_counter(1); lf.render_scene_term()
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_jitter_mt19937(5489, None)
    with pytest.raises(pkg.LensFlareError, match="counter RNG"):
        lf.render_scene_term()
    lf.set_jitter_counter(1)
    lf.render_scene_term()
    with pytest.raises(pkg.LensFlareError):
        lf.set_lens_camera(1, 0.0, 0.0)
    with pytest.raises(pkg.LensFlareError):
        lf.set_lens_camera(3, 0.001, 0.0)
    # a stop that lets nothing through cannot be calibrated
    lf.set_aperture(pkg.APERTURE_STARBURST, np.zeros((16, 16), np.float32))
    with pytest.raises(pkg.LensFlareError, match="no on-axis sample"):
        lf.render_scene_term()


@pytest.mark.parametrize("lens_name,W,H,ns", [("dgauss11.lens", 50, 27, 1), ("dgauss11.lens", 33, 20, 5),
                                                ("thinlens.lens", 40, 18, 7)])
def test_odd_frames_sample_counts_and_the_multi_gpu_deal(pkg, lf, sqrt_table, lens_name, W, H, ns):
    """Frame sizes that are no multiple of the wave tile, sample counts that are no square (part of the
    samples is unstratified), a lens without a stop -- and the tile-row deal of a multi-GPU frame: a context
    that owns every second tile row renders exactly those rows of the same frame."""
    lens = pkg.load_lens_file(lens_name)
    mask = load_texels("pentbig500_14.png") if int(lens["stop"]) >= 0 else np.ones((8, 8), np.float32)
    wpm = 0.003
    pos = [0.2, 0.1, 0.8]
    c2w = look_at(pos, [0.0, -0.2, -5.5])
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    got = lf.read_buffer(pkg.SCENE_BUFFER)
    info = lf.lens_camera()
    lfo.geo_follow_device(sqrt_table)
    try:
        lam = int(np.asarray(lens["ior"]).shape[0]) // 2
        smp = lfo.geo_lens_samples(lens, W, H, ns, KEY, lam, np.arange(W * H), mask)
        want, _, alive = compose(lens, mask, W, H, ns, c2w, pos, wpm, info["entrance_pupil_z_mm"], info["exposure"],
                                 smp, 6, SPHERES, TRIS, LIGHTS)
    finally:
        lfo.geo_follow_device(None)
    want = want.reshape(H, W, 3)
    assert alive.any() and (want.max(axis=-1) > 0.01).mean() > 0.1
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
    assert err.max() <= 1e-9, err.max()
    # rank 1 of 2: only its tile rows are rendered, and they are the same pixels
    lf.set_scene_term(np.zeros((H, W, 3)))            # (clear what the whole-frame launch left)
    lf.set_row_interleave(1, 2)
    lf.render_scene_term()
    part = lf.read_buffer(pkg.SCENE_BUFFER)
    lf.set_row_interleave(0, 1)
    own = (np.arange(H) // 8) % 2 == 1
    assert np.array_equal(part[own], got[own]) and not part[~own].any()


def test_aiming_the_lens_camera_at_the_exit_pupil(pkg, lf, sqrt_table):
    """lf_set_lens_camera_aim: the lens camera's samples aim at the paraxial image of the stop's open part
    (x 1.3) instead of the march's disc.  Same estimator for the primary path -- pixels = the float32 oracle
    under that disc to 1e-9, the flat field is 1 on the axis again (the calibration follows), the scene's total
    agrees with the default disc's within the Monte-Carlo error -- with more of the samples leaving the lens
    (1.2-1.8x under a pentagon stop); the march's own disc is not touched."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns, wpm = 48, 32, 16, 0.004
    pos = [0.3, 0.2, 1.0]
    c2w = look_at(pos, [0.0, -0.2, -5.5])
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos)
    disc = lf.aim_at_exit_pupil(1.3)                   # the numbers lf_set_lens_camera_aim(1.3) derives ...
    lf.set_pupil_target(0.0, 0.0)                      # ... while the march keeps its default disc
    default_disc = lf.pupil_target()
    lf.set_lens_camera(1, wpm, 0.0)
    lf.reset_scene_counters()
    lf.render_scene_term()
    base = lf.read_buffer(pkg.SCENE_BUFFER)
    base_cnt = lf.scene_counters()
    lf.set_lens_camera_aim(1.3)
    lf.reset_scene_counters()
    lf.render_scene_term()
    got = lf.read_buffer(pkg.SCENE_BUFFER)
    cnt, info = lf.scene_counters(), lf.lens_camera()
    assert lf.pupil_target() == default_disc
    # (a pentagon fills 76 % of its circumscribed circle, the margin takes 1 / 1.3^2 of that, vignetting the rest)
    assert cnt["lens_left"] > 1.2 * base_cnt["lens_left"]
    lfo.geo_follow_device(sqrt_table)
    lfo.set_pupil_target(disc["radius_mm"], disc["z_mm"])
    try:
        exposure = lfo.lens_exposure(lens, W, mask)
        assert abs(info["exposure"] - exposure) <= 1e-12 * exposure
        smp = lfo.geo_lens_samples(lens, W, H, ns, KEY, 1, np.arange(W * H), mask)
        want, _, _ = compose(lens, mask, W, H, ns, c2w, pos, wpm, info["entrance_pupil_z_mm"], exposure, smp, 6,
                             SPHERES, TRIS, LIGHTS)
    finally:
        lfo.set_pupil_target(0.0, 0.0)
        lfo.geo_follow_device(None)
    want = want.reshape(H, W, 3)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
    assert err.max() <= 1e-9, err.max()
    # the same image: totals over the frame (16 samples per pixel, a quarter of them alive in `base`)
    assert got.sum() == pytest.approx(base.sum(), rel=0.05)
    lf.set_lens_camera_aim(0.0)
    lf.render_scene_term()
    assert np.array_equal(lf.read_buffer(pkg.SCENE_BUFFER), base)


@pytest.mark.parametrize("what", ["delta_one_lambda", "adaptive_batches", "per_wavelength", "area_light", "odd_rows"])
def test_compacted_scene_rays_equal_the_per_lane_kernel(pkg, lf, what):
    """Round 5: k_scene_lens queues the samples that LEFT the lens and walks the tree with full waves; every pixel
    must see the additions of k_scene_term<.., true> (one traversal per lane's own sample, lf_test_knob scene_compact 0) in
    the same order: frames and counters bit for bit -- with and without the adaptive early-out (lanes leaving at
    batch ends), one ray per wavelength (several entries per sample), sampled lights (Philox counters of the
    OWNER's pixel and sample), frame edges and the multi-GPU row deal."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns = (77, 45, 9) if what == "odd_rows" else (96, 64, 64)
    pos = [0.2, 0.1, 0.8]
    c2w = look_at(pos, [0.0, -0.2, -5.5])
    setup_scene_frame(pkg, lf, lens, mask, W, H, ns, c2w, pos)
    if what in ("adaptive_batches", "area_light"):
        lf.set_sampling(8, 0.25, 0.01, 100.0)
    if what == "area_light":
        # {type, rgb, v0..v3}: the sun as a DirectionalLight + an AreaLight (position, direction, dim_x, dim_y) above
        lf.set_scene_lights([[0.0, 2.0, 1.8, 1.5, 0.3 / 1.0, 0.8, 0.52] + [0.0] * 9,
                             [3.0, 6.0, 6.0, 5.0, 0.0, 2.5, -5.0, 0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
        lf.set_light_samples(4)
    lf.set_lens_camera(2 if what == "per_wavelength" else 1, 0.003, 0.0)
    if what == "odd_rows":
        lf.set_row_interleave(1, 2)
    frames, counters = [], []
    try:
        for compact, strided in ((1, 1), (0, 0), (1, 0)):      # (the wave's pixels: the march's strided tile / 8 x 8 adjacent)
            lf.test_knob("scene_compact", compact)
            lf.test_knob("scene_lens_strided", strided)
            lf.set_scene_term(np.zeros((H, W, 3)))
            lf.reset_scene_counters()
            lf.render_scene_term()
            frames.append(lf.read_buffer(pkg.SCENE_BUFFER))
            counters.append(lf.scene_counters())
    finally:
        lf.test_knob("scene_compact", -1)
        lf.test_knob("scene_lens_strided", -1)
    assert (frames[0].max(axis=-1) > 1e-3).mean() > 0.1
    for k in (1, 2):
        assert np.array_equal(frames[0], frames[k]), (k, np.abs(frames[0] - frames[k]).max())
        assert counters[0] == counters[k], (k, counters)
    if what in ("adaptive_batches", "area_light"):
        full = H * W * ns * (1 if what != "per_wavelength" else 3)
        assert counters[0]["lens_samples"] < 0.9 * full        # (some pixels did leave early)
