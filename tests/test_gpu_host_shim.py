"""The C++ host mirror of the reference's PathTracer surface (lens-flare_amd/host/): shim_demo
replays RaytracedRenderer::start_raytracing + the tile workers (raytraced_renderer.cpp:300-354,
:622-647) through lfamd::PathTracer with 4 worker threads, and its buffers must equal what the REAL
reference produced for the same inputs (tests/golden/)."""
import os
import subprocess

import numpy as np
import pytest

from goldenlib import Case, load_texels

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "lens-flare_amd", "host", "shim_demo")


@pytest.mark.parametrize("name", ["f64x48_pentbiglines", "f97x65_odd_rotcam", "f80x50_two_suns",
                                  "f64x64_no_sun"])
def test_cpp_shim_matches_reference(name, tmp_path):
    assert os.path.exists(DEMO), "run __graft_entry__.build() first"
    case = Case(name)
    m = case.meta
    ap, gh = load_texels(m["aperture"]), load_texels(m["ghost_aperture"])
    ap.tofile(tmp_path / "ap.f32")
    gh.tofile(tmp_path / "gh.f32")
    with open(tmp_path / "case.txt", "w") as f:
        vals = [case.W, case.H, m["ns_aa"], repr(float(m["flare_radius"])), repr(float(m["flare_intensity"])),
                repr(m["hFov"]), repr(m["vFov"])] + [repr(v) for v in m["cam_pos"]] + \
               [repr(v) for v in m["c2w"]] + [ap.shape[1], ap.shape[0], gh.shape[1], gh.shape[0],
                                              len(m["lights"])]
        f.write(" ".join(map(str, vals)) + "\n")
        for l in m["lights"]:
            f.write(" ".join(repr(float(v)) for v in l) + "\n")
    out = str(tmp_path / "o")
    r = subprocess.run([DEMO, str(tmp_path / "case.txt"), str(tmp_path / "ap.f32"),
                        str(tmp_path / "gh.f32"), out, "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ghost = np.fromfile(out + ".ghost.f64", np.float64).reshape(case.H, case.W, 3)
    assert np.array_equal(ghost, case.ghost)
    flares = open(out + ".flares.txt").read().split("\n")
    assert int(flares[0].split()[0]) == m["n_flares"]
    sample = np.fromfile(out + ".sample.f64", np.float64).reshape(case.H, case.W, 3)
    if m["n_flares"] == 0:
        assert not sample.any()
        return
    err = np.abs(sample - case.sample) / np.abs(case.sample)
    assert err.max() <= 1e-9
    rgba = np.fromfile(out + ".rgba.u32", np.uint32).reshape(case.H, case.W)
    assert np.array_equal(rgba, case.rgba)
