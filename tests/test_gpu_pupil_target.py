"""Exit-pupil-aimed sampling (lf_set_pupil_target / lf_aim_at_exit_pupil, an OPT-IN part of the sampling
specification; no reference counterpart -- the reference has no lens to aim through).  By default every
sensor sample aims at the rear element's whole clear aperture, which is valid for every path but spends
most samples on rays that never pass the stop.  Aiming at the image of the stop's open part instead is
an unbiased estimator for the paths whose first crossing of the stop precedes their first reflection
(primary + pairs (i, j) with i in front of the stop); the pairs with both mirrors behind the stop keep
the default and are added by a second launch (lf_set_ghost_accumulate).  Checked here: the host
arithmetic; device = float32 oracle bit for bit and = float64 tracer within 1e-4 under a target; the
two-launch frame converges to the default frame; and the waste it removes."""
import numpy as np
import pytest

from goldenlib import load_texels
from oracle import lfo


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _abcd_exit_pupil(lens, lam):
    """independent of the library: numpy ABCD from the stop plane to the rear vertex"""
    n, stop = lens["n"], lens["stop"]
    M = np.eye(2)
    medium = float(lens["ior"][lam, stop - 1]) if stop > 0 else 1.0
    for k in range(stop, n):
        if k > stop:
            c = 0.0 if lens["radius"][k] == 0 else 1.0 / float(lens["radius"][k])
            n2 = float(lens["ior"][lam, k])
            M = np.array([[1, 0], [c * (medium - n2) / n2, medium / n2]]) @ M
            medium = n2
        if k + 1 < n:
            M = np.array([[1, float(lens["thickness"][k])], [0, 1]]) @ M
    l = -M[0, 1] / M[1, 1]
    z_rear = float(np.sum(lens["thickness"][:-1].astype(np.float64)))
    return z_rear + l, M[0, 0] + l * M[1, 0]


def test_paraxial_exit_pupil_host_arithmetic(pkg):
    lens = pkg.load_lens_file("dgauss11.lens")
    for lam in range(3):
        z, m = pkg.paraxial_exit_pupil(lens, lam)
        z2, m2 = _abcd_exit_pupil(lens, lam)
        assert z == pytest.approx(z2, rel=1e-6) and m == pytest.approx(m2, rel=1e-6)
    z, m = pkg.paraxial_exit_pupil(lens)
    z_sensor = float(lens["thickness"].astype(np.float64).sum())
    assert 10 < z < 20 and 1.4 < m < 1.7 and z < z_sensor       # a magnified virtual image inside the lens
    # a stop in front of one thin positive lens (f = 50 mm at distance 20 mm): 1/l' = 1/l + 1/f with
    # l = -20 -> l' = -33.3 mm (virtual, in front of the lens), m = l'/l = 1.667
    nn, R = 1.5, 50.0
    one = dict(n=3, stop=0, radius=np.array([0, R, -R], np.float32), thickness=np.array([20.0, 1e-4, 49.99], np.float32),
               ior=np.array([[1.0, nn, 1.0]], np.float32), semi_aperture=np.array([5, 10, 10], np.float32), sensor_width_mm=36.0)
    f = 1.0 / ((nn - 1) * (2 / R))
    lp = 1.0 / (1.0 / -20.0 + 1.0 / f)
    z, m = pkg.paraxial_exit_pupil(one, 0)
    assert z - 20.0 == pytest.approx(lp, rel=1e-3) and m == pytest.approx(lp / -20.0, rel=1e-3)


def _class_pairs(lens):
    """A: primary + pairs whose first mirror lies in front of the stop; B: both mirrors behind it"""
    n, stop = lens["n"], lens["stop"]
    allp = [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
    return [p for p in allp if p[0] < stop], [p for p in allp if p[0] > stop]


@pytest.fixture(scope="module")
def lf(pkg):
    ctx = pkg.LensFlare(0)
    yield ctx
    ctx.close()


@pytest.mark.gpu
def test_targeted_march_equals_both_oracles(pkg, lf):
    """under a pupil target: pixels and counters = the float32 oracle bit for bit, and within 1e-4 of
    the independent float64 tracer (same bar as tests/test_gpu_march_f64.py)"""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp, key = 64, 48, 64, 0xA1A1
    sun, rad, alpha = [0.03, 0.02, -1.0], [1.0, 0.9, 0.5], 0.05
    pa, _ = _class_pairs(lens)
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(sun, rad, alpha)
    lf.set_ghost_pairs(pa, True)
    tgt = lf.aim_at_exit_pupil(1.1)
    base_h = float(lens["semi_aperture"][-1])
    assert 0 < tgt["radius_mm"] < 12 and tgt["z_mm"] < tgt["z_sensor_mm"]
    # the solid angle sampled shrinks by more than half
    z_rear = tgt["z_sensor_mm"] - float(lens["thickness"][-1])
    ratio = (tgt["radius_mm"] / (tgt["z_sensor_mm"] - tgt["z_mm"])) ** 2 / (base_h / (tgt["z_sensor_mm"] - z_rear)) ** 2
    assert ratio < 0.65, ratio
    lf.reset_counters()
    lf.trace_ghosts(spp, key)
    img, cnt = lf.read_buffer(pkg.GHOST_BUFFER), lf.counters()
    lfo.set_pupil_target(tgt["radius_mm"], tgt["z_mm"])
    lfo.geo_follow_device(lf)
    try:
        og, ocnt = lfo.geo_trace(lens, W, H, 0, H, spp, key, pa, True, mask, sun, rad, alpha)
        lfo.geo_follow_device(None)
        ref, frag, c64 = lfo.g64_trace(lens, W, H, 0, H, spp, key, pa, True, mask, sun, rad, alpha, n_threads=16, cull=lf.cull_table_and_block())
    finally:
        lfo.geo_follow_device(None)
        lfo.set_pupil_target(0.0, 0.0)
        lf.set_pupil_target(0.0, 0.0)
    assert cnt == ocnt and np.array_equal(img, og) and og.max() > 0
    lit = ref >= 2e-5
    assert lit.sum() > 100
    assert np.all(np.abs(img - ref)[lit] <= 1e-4 * ref[lit] + 1.05 * frag[lit])
    assert c64["rays_launched"] == cnt["rays_launched"]


@pytest.mark.gpu
def test_two_launch_frame_converges_to_the_default_frame_and_wastes_less(pkg, lf):
    """frame = [primary + front pairs, aimed at the exit pupil] + [rear pairs, aimed at the rear
    element]: the same image as the default sampling (means over independent keys agree within their
    Monte-Carlo error, over the whole frame incl. its corners), with far fewer rays lost before the stop."""
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = load_texels("pentbig500_14.png")
    W, H, spp = 96, 64, 256
    sun, rad, alpha = [0.12, 0.08, -1.0], [1.0, 0.9, 0.5], 0.08     # a sun well off the axis
    pa, pb = _class_pairs(lens)
    keys = [0x100 + k for k in range(12)]
    lf.set_frame(W, H)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)
    lf.set_lens(lens)
    lf.set_sun(sun, rad, alpha)

    def default_frame(key):
        lf.set_pupil_target(0.0, 0.0)
        lf.set_ghost_pairs(None, True)
        lf.reset_counters()
        lf.trace_ghosts(spp, key)
        return lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2), lf.counters()

    def targeted_frame(key):
        lf.set_ghost_pairs(pa, True)
        lf.aim_at_exit_pupil(1.1)
        lf.reset_counters()
        lf.trace_ghosts(spp, key)
        ca = lf.counters()
        lf.set_pupil_target(0.0, 0.0)
        lf.set_ghost_pairs(pb, False)
        lf.set_ghost_accumulate(True)
        lf.trace_ghosts(spp, key ^ 0x5555)
        lf.set_ghost_accumulate(False)
        return lf.read_buffer(pkg.GHOST_BUFFER).sum(axis=2), ca

    d = np.stack([default_frame(k)[0] for k in keys])
    t = np.stack([targeted_frame(k)[0] for k in keys])
    _, cd = default_frame(keys[0])
    _, ct = targeted_frame(keys[0])
    lf.set_ghost_pairs(None, True)
    md, mt = d.mean(0), t.mean(0)
    se = np.sqrt(d.var(0, ddof=1) / len(keys) + t.var(0, ddof=1) / len(keys))
    lit = md > 1e-3 * md.max()
    assert lit.sum() > 500
    z = (mt - md)[lit] / np.maximum(se[lit], 1e-300)
    assert abs(z.mean()) < 0.25, z.mean()                 # no offset
    assert 0.7 < z.std() < 1.4, z.std()                   # differences are Monte-Carlo noise, nothing else
    # (12 keys: z follows a t distribution with ~20 degrees of freedom, and the sub-cell sharing makes a tile's
    # pixels move together: a handful of 3000 pixels beyond 4.5 is within that)
    assert (np.abs(z) < 4.5).mean() > 0.997 and np.abs(z).max() < 8.0
    # the frame totals: key-to-key spread of the totals themselves (pixels of a tile are correlated)
    sd, st = d.sum(axis=(1, 2)), t.sum(axis=(1, 2))
    zt = (st.mean() - sd.mean()) / np.sqrt(sd.var(ddof=1) / len(keys) + st.var(ddof=1) / len(keys))
    print(f"frame totals: default {sd.mean():.5f} +- {sd.std(ddof=1) / np.sqrt(len(keys)):.5f}, aimed {st.mean():.5f} +- "
          f"{st.std(ddof=1) / np.sqrt(len(keys)):.5f} (z = {zt:.2f})")
    assert abs(zt) < 4.0 and st.mean() == pytest.approx(sd.mean(), rel=2e-2)
    # per-pixel variance at equal spp drops (each sample is aimed where light can pass)
    vr = t.var(0, ddof=1)[lit].sum() / d.var(0, ddof=1)[lit].sum()
    assert vr < 0.85, vr
    # rays lost before they get anywhere (mask / housing / vignetting), per ray launched: default vs aimed
    lost_d = (cd["rays_clipped_stop"] + cd["rays_vignetted"]) / cd["rays_launched"]
    lost_t = (ct["rays_clipped_stop"] + ct["rays_vignetted"]) / ct["rays_launched"]
    reach_d = cd["rays_reached_scene"] / cd["rays_launched"]
    reach_t = ct["rays_reached_scene"] / ct["rays_launched"]
    print(f"lost before the scene: default {lost_d:.3f}, aimed {lost_t:.3f}; reach the scene: {reach_d:.3f} -> {reach_t:.3f}; "
          f"variance ratio at equal spp {vr:.3f}")
    assert reach_t > 1.5 * reach_d
