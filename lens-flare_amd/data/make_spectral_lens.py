#!/usr/bin/env python3
"""Writes dgauss11_8lambda.lens: dgauss11.lens with 8 index columns by the 2-term Cauchy fit
n(lambda) = A + B / lambda^2 through each glass's three tabulated indices (C, d, F lines), wavelengths
equally spaced from the C to the F line (configuration C5; lens_flare_amd.spectral_lens).
    python lens-flare_amd/data/make_spectral_lens.py
tests/test_spectral_lens_cpu.py checks that the committed file is what this script writes."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def render(pkg):
    lens3 = pkg.load_lens_file("dgauss11.lens")
    lens, _, _ = pkg.spectral_lens(lens3, 8)
    lam = lens["lambda_nm"]
    out = ["# dgauss11.lens with 8 index columns: n(lambda) = A + B / lambda^2 fitted (least squares) through the",
           "# C, d and F indices of every glass -- SURVEY 8d's dispersion law for configuration C5.  Written by",
           "# make_spectral_lens.py; row: radius[mm] thickness[mm] n_1 .. n_8 semi_aperture[mm].",
           "sensor_width_mm %.1f" % lens["sensor_width_mm"],
           "lambda_nm " + " ".join("%.4f" % v for v in lam)]
    for k in range(lens["n"]):
        is_stop = k == lens["stop"]
        idx = ["0" if is_stop else ("1" if float(v) == 1.0 else "%.6f" % float(v)) for v in lens["ior"][:, k]]
        out.append("%-9s %-7s %s %s" % (("%g" % lens["radius"][k]), ("%g" % lens["thickness"][k]), " ".join(idx),
                                        "%g" % lens["semi_aperture"][k]))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    import __graft_entry__ as g
    open(os.path.join(HERE, "dgauss11_8lambda.lens"), "w").write(render(g.load_package()))
