"""Error behaviour of the prescription hand-over (lf_set_lens, lf_load_lens_file): malformed input is
refused with LF_ERR_INVALID and a message, the context stays usable -- never a crash, never a
silently different lens."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_set_lens_refuses_malformed_prescriptions(pkg):
    lens = pkg.load_lens_file("dgauss11.lens")
    lf = pkg.LensFlare(0)
    try:
        with pytest.raises(pkg.LensFlareError):        # before lf_set_frame
            lf.set_lens(lens)
        lf.set_frame(64, 48)
        lf.set_lens(lens)                              # the good one
        n_ok = lf.lens_info()["n"]

        def broken(**kw):
            b = copy.deepcopy(lens)
            for k, v in kw.items():
                a = np.array(b[k], np.float32)
                idx, val = v
                a[idx] = val
                b[k] = a
            return b

        stop = lens["stop"]
        glass = 0 if stop != 0 else 1
        cases = [broken(radius=(glass, np.nan)), broken(radius=(glass, np.inf)), broken(thickness=(glass, np.nan)),
                 broken(thickness=(glass, -np.inf)), broken(semi_aperture=(glass, 0.0)),
                 broken(semi_aperture=(glass, -1.0)), broken(semi_aperture=(glass, np.nan)),
                 broken(semi_aperture=(glass, np.inf)), broken(ior=((0, glass), 0.9)),
                 broken(ior=((0, glass), np.nan)), broken(ior=((0, glass), np.inf)),
                 broken(radius=(stop, 10.0))]
        for b in cases:
            with pytest.raises(pkg.LensFlareError) as e:
                lf.set_lens(b)
            assert "lens" in str(e.value)
        b = copy.deepcopy(lens); b["sensor_width_mm"] = 0.0
        with pytest.raises(pkg.LensFlareError):
            lf.set_lens(b)
        b = copy.deepcopy(lens); b["stop"] = lens["n"]
        with pytest.raises(pkg.LensFlareError):
            lf.set_lens(b)
        # the refused calls left the last good lens in place
        assert lf.lens_info()["n"] == n_ok
    finally:
        lf.close()


@pytest.mark.parametrize("text", [
    "",                                              # no surface at all
    "# only a comment\n",
    "35.0 2.0 1.5\n",                                # a row needs radius thickness n semi_aperture
    "35.0 2.0 1.5 10.0\n-35.0 20.0 1.0\n",          # rows of different length
    "35.0 2.0 abc 10.0\n",                           # not a number
    "35.0 2.0 1.5 10.0 extra\n",
    "sensor_width_mm\n35.0 2.0 1.5 10.0\n",          # keyword without its value
    "sensor_width_mm 36 24\n35.0 2.0 1.5 10.0\n",
    "35.0 2.0 0.5 10.0\n",                           # index below 1
    "nan 2.0 1.5 10.0\n",
    "35.0 2.0 1.5 -3\n",
    "\n".join("35.0 1.0 1.5 10.0" for _ in range(400)) + "\n",   # more surfaces than LF_MAX_SURFACES
    "35.0 2.0 " + " ".join(["1.5"] * 40) + " 10.0\n",            # more wavelengths than LF_MAX_LAMBDA
])
def test_lens_file_parser_refuses_malformed_files(pkg, tmp_path, text):
    lf = pkg.LensFlare(0)
    try:
        lf.set_frame(32, 24)
        f = tmp_path / "bad.lens"
        f.write_text(text)
        with pytest.raises(pkg.LensFlareError):
            lf.load_lens_file(str(f))
        with pytest.raises(pkg.LensFlareError):
            lf.load_lens_file(str(tmp_path / "missing.lens"))
        lf.load_lens_file("thinlens.lens")           # the context still works
        assert lf.lens_info()["n"] >= 1
    finally:
        lf.close()


def test_a_new_lens_drops_the_pupil_target_and_reduced_discs_refuse_rear_pairs(pkg):
    """a pupil target is a property of the prescription it was computed for.
    lf_set_lens / lf_load_lens_file start from the default disc again; and a reduced disc with a pair whose
    mirrors both sit behind the stop is refused instead of rendered biased."""
    from goldenlib import load_texels
    lf = pkg.LensFlare(0)
    try:
        lens = pkg.load_lens_file("dgauss11.lens")
        lf.set_frame(32, 16)
        lf.set_aperture(pkg.APERTURE_STARBURST, load_texels("pentbig500_14.png"))
        lf.set_lens(lens)
        default = lf.pupil_target()
        aimed = lf.aim_at_exit_pupil(1.2)
        assert aimed["radius_mm"] != default["radius_mm"] and aimed["z_mm"] != default["z_mm"]
        lf.set_sun([0.03, 0.02, -1.0], [1.0, 0.9, 0.5], 0.05)
        # the default pair set holds the 10 pairs behind the stop: refused under a reduced disc
        with pytest.raises(pkg.LensFlareError, match="behind the stop"):
            lf.trace_ghosts(4, 1)
        stop = lens["stop"]
        front = [(i, j) for i in range(lens["n"]) for j in range(i + 1, lens["n"]) if i != stop and j != stop and i < stop]
        lf.set_ghost_pairs(front, True)
        lf.trace_ghosts(4, 1)                          # pairs with a mirror in front of the stop: fine
        # another prescription: the disc of the old one is gone
        thin = pkg.load_lens_file("thinlens.lens")
        lf.set_lens(thin)
        t = lf.pupil_target()
        assert t["radius_mm"] == float(thin["semi_aperture"][-1])
        assert abs(t["z_mm"] - float(thin["thickness"][0])) < 1e-6
        lf.set_lens(lens)
        assert lf.pupil_target() == default
        lf.trace_ghosts(4, 1)                          # default disc + default pairs
    finally:
        lf.close()


def test_flare_arithmetic_modes_agree_to_a_few_ulp(pkg):
    """the flare layer keeps the reference's own pow() calls selectable (and uses them in
    MT19937 parity mode); the fast forms differ from them by a few ulp, stated in include/lensflare.h."""
    from goldenlib import Case, load_texels
    case = Case("f64x48_pentbiglines")
    m = case.meta
    lf = pkg.LensFlare(0)
    try:
        lf.set_frame(case.W, case.H)
        lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
        lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
        lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
        lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
        lf.find_sun_pos(m["lights"])
        lf.generate_ghost_buffer()
        lf.set_jitter_counter(5)
        frames = {}
        for mode in (1, 2, 0):
            lf.set_flare_arithmetic(mode)
            lf.render_flare_layer()
            frames[mode] = lf.read_buffer(pkg.STARBURST_BUFFER)
        assert np.array_equal(frames[0], frames[2])                 # counter RNG: auto = fast
        rel = np.abs(frames[1] - frames[2]) / np.abs(frames[1])
        assert 0 < rel.max() < 4e-15                                # a few ulp, not more -- and not identical
        lf.set_jitter_mt19937(5489, None)
        lf.render_flare_layer()
        auto_mt = lf.read_buffer(pkg.STARBURST_BUFFER)
        lf.set_flare_arithmetic(1)
        lf.render_flare_layer()
        assert np.array_equal(auto_mt, lf.read_buffer(pkg.STARBURST_BUFFER))   # parity mode: auto = exact
    finally:
        lf.close()
