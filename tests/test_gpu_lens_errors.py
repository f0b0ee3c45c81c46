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
