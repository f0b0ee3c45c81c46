"""CPU checks behind the lens camera of the scene term (round 4; lf_set_lens_camera): the host arithmetic that
needs no device, and the two tracers against each other on the primary path -- so that the `not gpu` run already
holds what the GPU tests then compare the device with."""
import math

import numpy as np

from goldenlib import load_texels
from oracle import lfo


def _pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_entrance_pupil_is_where_the_chief_rays_cross_the_axis():
    """lf_paraxial_entrance_pupil (the reference's T / R operators through the front group) against the float32
    march itself: rays from sensor points near the axis aimed at the centre of the paraxial EXIT pupil pass the
    stop's centre and leave the front element along lines that cross the axis at the entrance pupil; their slope
    is the sensor height over the focal length (the pixel <-> direction mapping find_sun_pos assumes)."""
    pkg = _pkg()
    lens = pkg.load_lens_file("dgauss11.lens")
    z_ep, m_ep = pkg.paraxial_entrance_pupil(lens)
    z_xp, m_xp = pkg.paraxial_exit_pupil(lens)
    efl = pkg.paraxial_efl(lens)
    assert 15.0 < z_ep < 25.0 and 1.2 < m_ep < 1.7          # inside the lens, magnified: a double Gauss
    mask = np.ones((8, 8), np.float32)
    lfo.set_pupil_target(0.01, z_xp)
    try:
        for X in (0.05, 0.5, 2.0):
            o = lfo.geo_lens_rays(lens, 64, 1, [[X, 0.0]], [[0.0, 0.0]], mask)[0]
            assert o[7] == 1.0
            z_cross = o[2] - o[0] * o[5] / o[3]             # where the exit ray's line meets the axis
            assert abs(z_cross - z_ep) < 0.05 * (1 + X), (X, z_cross, z_ep)
            assert abs(o[3] / -o[5] - (-X / efl)) < 2e-4 * (1 + X)
    finally:
        lfo.set_pupil_target(0.0, 0.0)
    # a stop in front of a thin lens: the entrance pupil is the stop itself (nothing in front of it images it)
    thin = pkg.load_lens_file("thinlens.lens")
    front_stop = dict(n=3, stop=0, radius=np.array([0.0, 50.0, -50.0], np.float32),
                      thickness=np.array([4.0, 5.0, 47.54], np.float32),
                      ior=np.stack([np.array([1.0, thin["ior"][l, 0], 1.0], np.float32) for l in range(3)]),
                      semi_aperture=np.array([5.0, 10.0, 10.0], np.float32), sensor_width_mm=36.0)
    z, m = pkg.paraxial_entrance_pupil(front_stop)
    assert z == 0.0 and m == 1.0
    # a stop BEHIND a single refracting surface (R = 50, n = 1.5, 10 mm of glass): the classic apparent-depth
    # formula for the surface's image of an axial point at depth d: s' = d / (n - (n - 1) d / R), m = n s' / d ... in air
    n, R, d = 1.5, 50.0, 10.0
    rear_stop = dict(n=2, stop=1, radius=np.array([R, 0.0], np.float32), thickness=np.array([d, 30.0], np.float32),
                     ior=np.array([[n, 1.0]] * 3, np.float32), semi_aperture=np.array([12.0, 6.0], np.float32),
                     sensor_width_mm=36.0)
    z, m = pkg.paraxial_entrance_pupil(rear_stop)
    # refraction at one surface, object in the glass at distance d: n / d + 1 / s' = (n - 1) / R (distances positive
    # on their own sides) -> virtual image at depth |s'| behind the vertex
    s_img = 1.0 / ((n - 1) / R - n / d)
    assert s_img < 0 and abs(z - (-s_img)) < 1e-9
    assert abs(m - (n * (-s_img) / d)) < 1e-9


def test_both_tracers_agree_on_the_lens_cameras_samples():
    """geo_lens_samples (float32, the device's recipe) against g64_lens_samples (float64, textbook): the same
    sample of the same pixel leaves the front element at the same place in the same direction with the same
    weight; fates differ only where the float64 tracer itself flagged the sample as fragile."""
    pkg = _pkg()
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, ns, key = 40, 24, 16, 0x1e45f1a4e
    px = np.arange(W * H)
    a = lfo.geo_lens_samples(lens, W, H, ns, key, 1, px, mask)
    b = lfo.g64_lens_samples(lens, W, H, ns, key, 1, px, mask)
    alive32, alive64, fragile = a[..., 7] > 0, b[..., 9] == 0, b[..., 8] > 0
    assert 0.15 < alive64.mean() < 0.4
    assert np.all((alive32 == alive64) | fragile)
    both = alive32 & alive64
    assert np.abs(a[both][:, 0:3] - b[both][:, 0:3]).max() < 5e-5          # mm on the front element
    assert np.abs(a[both][:, 3:6] - b[both][:, 3:6]).max() < 6e-6
    assert np.abs(a[both][:, 6] / b[both][:, 6] - 1.0).max() < 2e-5
    # the exposure calibration of the two: the mean on-axis weight over the 64 x 64 pupil grid
    e32 = lfo.lens_exposure(lens, W, mask)
    assert 100.0 < e32 < 400.0
    # carried into a scene, the two give the same pixels (1e-4 + the fragile samples' potential weight)
    spheres = [(0, -101.0, -6, 100.0, "d", 0.6, 0.6, 0.55), (-0.9, -0.4, -5, 0.6, "d", 0.8, 0.2, 0.2),
               (0.1, 0.35, -6.5, 0.5, "e", 1.5, 1.2, 0.4)]
    lights = [[0.0, 0.3, 0.8, 0.52, 2.0, 1.8, 1.5]]
    lights[0][1:4] = (np.array(lights[0][1:4]) / np.linalg.norm(lights[0][1:4])).tolist()
    z_ep, _ = pkg.paraxial_entrance_pupil(lens)

    def frame(smp, wcol):
        o, d = lfo.lens_exit_to_world(smp[..., 0:3], smp[..., 3:6], np.eye(3), [0.0, 0.0, 0.0], 0.004, z_ep)
        rays = np.concatenate([o, d, np.full(o.shape[:-1] + (1,), 0.01), np.full(o.shape[:-1] + (1,), 100.0)], -1)
        ok = smp[..., wcol] > 0
        L = np.zeros(o.shape)
        L[ok] = lfo.scene_radiance_rays(spheres, [], lights, rays[ok])
        return (L * (smp[..., wcol].astype(np.float64) * e32)[..., None]).sum(axis=1) / (ns + 1), L

    f32, L32 = frame(a, 6)
    f64, L64 = frame(b, 6)
    allow = ((b[..., 8] > 0) * b[..., 7] * e32 * max(L32.max(), L64.max()) / (ns + 1)).sum(axis=1)[:, None]
    # (the SCENE amplifies the last bits of an exit direction where radiance changes quickly -- a shading
    # terminator, a silhouette: the lens part is held to 1e-5 above; composed pixels to 1e-4 on 99 % and 2e-3 on all)
    dev = np.abs(f32 - f64)
    assert np.all(dev <= 2e-3 * np.abs(f64) + allow + 1e-13)
    assert (dev <= 1e-4 * np.abs(f64) + allow + 1e-13).mean() > 0.99
    assert (f64.max(axis=-1) > 0.01).mean() > 0.3


def test_image_scale_is_the_focal_length_in_the_focal_plane_and_follows_the_sensor():
    """lf_paraxial_image_scale (what lf_set_sun_from_flares(efl_mm <= 0) divides by): equal to the focal length when
    the last thickness is the paraxial back focal distance; with the sensor moved by s it changes by s x the chief
    ray's exit slope = s x efl / (distance exit pupil -> focal plane) ... checked through the exit pupil."""
    pkg = _pkg()
    lens = pkg.load_lens_file("dgauss11.lens")
    efl = pkg.paraxial_efl(lens)
    scale = pkg.paraxial_image_scale(lens)
    assert abs(scale / efl - 1.0) < 2e-3                   # the file's sensor sits (nearly) in the focal plane
    z_xp, _ = pkg.paraxial_exit_pupil(lens)                # prescription coordinates: z = 0 at the first vertex
    to_sensor = float(np.sum(lens["thickness"], dtype=np.float64)) - z_xp
    moved = dict(lens, thickness=lens["thickness"].copy())
    moved["thickness"][-1] += 1.5
    scale_m = pkg.paraxial_image_scale(moved)
    # a chief ray leaves the exit pupil's centre: similar triangles over the distances pupil -> sensor
    assert scale_m > scale
    assert abs(scale_m / scale - (to_sensor + 1.5) / to_sensor) < 1e-6
    # a thin lens with the stop at the lens: the chief ray goes straight through the vertex
    tl = pkg.load_lens_file("thinlens.lens")
    s = pkg.paraxial_image_scale(tl)
    assert abs(s / pkg.paraxial_efl(tl) - 1.0) < 0.05 and s > 0
