"""Range contract of the march's accumulators (include/lensflare.h, "RANGE CONTRACT"; round 5).  A pixel
channel is a sum of unsigned 64-bit fixed-point contributions: negative or non-finite radiances / spectral
weights are refused, and the grid's exponent is chosen per launch so that no sum can wrap -- an HDR sun is this
path's use case: on the fixed 2^-36 grid of rounds 1-4 a pixel channel wrapped once sum v >= 2^28, which the sun's
own image on the test frame below reaches at a radiance of 6.5e7 (1024 spp x 8 wavelengths).  No reference counterpart (the reference's ghosts are f64 sums,
src/pathtracer/pathtracer.cpp:305-410)."""
import numpy as np
import pytest

from goldenlib import load_texels
from oracle import lfo


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_fix_bits_rule_of_the_oracle(pkg):
    """CPU: the exponent rule itself (the device's lf_march_fix_bits is the same double arithmetic)."""
    import ctypes as C
    lens = pkg.load_lens_file("dgauss11.lens")
    lib = lfo.lib()
    lib.geo_fix_bits.restype = C.c_int

    def bits(rad, n_paths, spp, gn=0.3):
        L = lfo.geo_lens(lens, (0, 0, -1), (rad, rad, rad), 0.05)
        return lib.geo_fix_bits(C.byref(L), C.c_float(gn), n_paths, spp)

    assert bits(1.0, 46, 256) == 36 and bits(10.0, 46, 1024) == 36       # every frame of rounds 1-4 keeps its grid
    b = bits(1e7, 46, 1024)
    assert b < 36
    worst = 1024 * 46 * float(np.float32(0.3)) * 1e7
    assert worst * 2.0 ** b < 2.0 ** 62 <= worst * 2.0 ** (b + 1)         # the largest exponent below the bound
    assert bits(2.0 ** 40, 46, 1024) == bits(2.0 ** 30, 46, 1024) - 10    # scale-covariant
    assert bits(0.0, 46, 256) == 36


@pytest.mark.gpu
def test_hdr_sun_does_not_wrap_and_bad_inputs_are_refused(pkg):
    lens = pkg.load_lens_file("dgauss11_8lambda.lens")
    w8, _scale = pkg.spectral_weights(lens["lambda_nm"])
    mask = load_texels("pentbig500_14.png")
    W, H, spp, key = 32, 16, 1024, 0xBEE5
    sun = [0.004, 0.003, -1.0]          # its image lands inside the small frame
    lf = pkg.LensFlare(0)
    try:
        lf.set_frame(W, H)
        lf.set_aperture(pkg.APERTURE_STARBURST, mask)
        lf.set_lens(lens)
        lf.set_lambda_rgb(w8)

        def frame(rad):
            lf.set_sun(sun, [rad, 0.9 * rad, 0.5 * rad], 0.05)
            lf.reset_counters()
            lf.trace_ghosts(spp, key)
            return lf.read_buffer(pkg.GHOST_BUFFER), lf.counters(), lf.march_fix_bits()

        one, c1, b1 = frame(1.0)
        big, c7, b7 = frame(1e9)
        assert b1 == 36 and b7 < 36 and c1 == c7
        assert one.max() > 0
        # the old grid WOULD have wrapped: the brightest channel's sum exceeds 2^28 (2^64 / 2^36)
        assert big.max() * spp >= 2.0 ** 28
        # ... and the new one did not: 1e9 x the radiance-1 frame (float rounding of each contribution's
        # radiance product, 2^-24 relative, + one grid step per contribution)
        lit = one > 0
        step = 2.0 ** -b7
        n_contrib = spp * 46 * 8
        assert np.all(np.abs(big - 1e9 * one)[lit] <= 2e-7 * 1e9 * one[lit] + step * n_contrib / spp + 1e9 * 2.0 ** -36 * n_contrib / spp)
        assert np.all(big[~lit] <= step * n_contrib / spp)
        # the oracle follows the exponent: bit for bit at 1e9
        lfo.geo_follow_device(lf)
        try:
            og, oc = lfo.geo_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun, [1e9, 0.9e9, 0.5e9], 0.05,
                                   n_threads=16, lambda_rgb=w8)
        finally:
            lfo.geo_follow_device(None)
        assert oc == c7 and np.array_equal(og, big)
        # scale covariance beyond the default grid: 2^10 x the radiance = 2^10 x the frame, bit for bit
        a, _, ba = frame(2.0 ** 30)
        b, _, bb = frame(2.0 ** 40)
        assert ba - bb == 10 and np.array_equal(b, a * 1024.0) and a.max() > 0
        # a frame composed of two launches (lf_set_ghost_accumulate) on an HDR grid: what the buffer holds is
        # brought to the launch's grid and back by exact powers of two -- the sum of the two frames, bit for bit
        lf.set_sun(sun, [2.0 ** 30, 2.0 ** 29, 2.0 ** 28], 0.05)
        n, stop = lens["n"], lens["stop"]
        allp = [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
        front, rear = [p for p in allp if p[0] < stop], [p for p in allp if p[0] > stop]
        lf.set_ghost_pairs(front, True)
        lf.trace_ghosts(spp, key)
        fa = lf.read_buffer(pkg.GHOST_BUFFER)
        lf.set_ghost_pairs(rear, False)
        lf.trace_ghosts(spp, key)
        fb = lf.read_buffer(pkg.GHOST_BUFFER)
        assert lf.march_fix_bits() < 36 and fa.max() > 0 and fb.max() > 0
        lf.set_ghost_pairs(front, True)
        lf.trace_ghosts(spp, key)
        lf.set_ghost_pairs(rear, False)
        lf.set_ghost_accumulate(True)
        lf.trace_ghosts(spp, key)
        lf.set_ghost_accumulate(False)
        assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), fa + fb)
        lf.set_ghost_pairs(None, True)
        # refused inputs leave the context usable
        for bad in ([np.nan, 1, 1], [1, -1e-3, 1], [1, 1, np.inf], [-0.0 - 1.0, 0, 0]):
            with pytest.raises(pkg.LensFlareError) as e:
                lf.set_sun(sun, bad, 0.05)
            assert e.value.status == 1 and "radiance" in str(e.value)      # LF_ERR_INVALID
        with pytest.raises(pkg.LensFlareError):
            lf.set_sun([np.nan, 0, -1], [1, 1, 1], 0.05)
        wbad = np.array(w8, np.float32).copy()
        wbad[3, 1] = -0.01
        with pytest.raises(pkg.LensFlareError) as e:
            lf.set_lambda_rgb(wbad)
        assert e.value.status == 1
        wbad[3, 1] = np.nan
        with pytest.raises(pkg.LensFlareError):
            lf.set_lambda_rgb(wbad)
        again, _, _ = frame(1.0)
        assert np.array_equal(again, one)
    finally:
        lf.close()
