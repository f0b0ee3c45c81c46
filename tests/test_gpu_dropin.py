"""The drop-in, proven: oracle/_ref/ref_dump_amd is the REFERENCE's own objects, unmodified, with ONE
translation unit replaced -- src/pathtracer/pathtracer.cpp by lens-flare_amd/host/pathtracer_amd.cpp,
which implements the reference's own `class CGL::PathTracer` (compiled against its unchanged header,
src/pathtracer/pathtracer.h:25-143) on the MI355X through the C ABI.  The same driver
(oracle/ref_driver.cpp) replays RaytracedRenderer::start_raytracing + raytrace_tile
(raytraced_renderer.cpp:300-311, :622-647) on it, and the buffers it dumps must equal the golden
frames the all-reference build of the very same driver produced (tests/golden/, oracle/make_golden.py).

The binary is built in the build container only (oracle/Makefile `dropin`; it needs the reference
checkout) and travels to the GPU box with the snapshot."""
import math
import os
import subprocess

import numpy as np
import pytest

from goldenlib import GOLD, Case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_dump_amd")

CASES = ["f64x48_pentbiglines", "f97x65_odd_rotcam", "f80x50_two_suns", "f96x64_naive_rgba",
         "s96x64_spheres", "s80x60_tris_rotcam", "c96x72_pyramid_dae", "z40x30_fuzz1", "z44x26_fuzz3",
         "q47x31_fuzz0", "q38x52_fuzz1", "q52x40_fuzz4"]


def _write_inputs(case, tmp):
    m = case.meta
    cam = tmp / "cam.txt"
    sd = case.H / (2 * math.tan(math.radians(m["vFov"]) / 2))
    with open(cam, "w") as f:     # Camera::load_settings (camera.cpp:228-242)
        f.write(f"{m['hFov']!r} {m['vFov']!r} {case.W / case.H!r} 0.01 100\n")
        f.write(" ".join(repr(float(v)) for v in m["cam_pos"]) + " 0 0 0\n1.5 0.7 5 0.5 100\n")
        f.write(" ".join(repr(float(v)) for v in m["c2w"]) + f"\n{case.W} {case.H} {sd!r}\n4.7 0\n")
    spec = ";".join(",".join(repr(float(v)) for v in l) for l in m["lights"])
    args = [str(cam), str(case.W), str(case.H), str(m["ns_aa"]), repr(float(m["flare_radius"])),
            repr(float(m["flare_intensity"])), os.path.join(GOLD, "apertures", m["aperture"]),
            os.path.join(GOLD, "apertures", m["ghost_aperture"]), spec, "tiles", str(tmp / "o")]
    sc = m.get("scene")
    if sc:
        sfile = tmp / "scene.txt"
        with open(sfile, "w") as f:
            num = lambda v: v if isinstance(v, str) else repr(float(v))  # noqa: E731
            for s in sc["spheres"]:
                f.write("sphere " + " ".join(num(v) for v in s) + "\n")
            for t in sc["tris"]:
                f.write("tri " + " ".join(num(v) for v in t) + "\n")
            for p in sc["points"]:
                f.write("point " + " ".join(num(v) for v in p) + "\n")
        args.append(str(sfile))
    return args


@pytest.mark.parametrize("name", CASES)
def test_reference_binary_with_pathtracer_replaced(name, tmp_path):
    assert os.path.exists(BIN), "oracle/_ref/ref_dump_amd is missing: make -C oracle dropin (build container)"
    case = Case(name)
    r = subprocess.run([BIN, "frame"] + _write_inputs(case, tmp_path), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = str(tmp_path / "o")
    meta = dict(l.split(None, 1) for l in open(out + ".meta.txt") if not l.startswith("flare "))
    assert int(meta["n_flares"]) == case.meta["n_flares"]
    assert [float.fromhex(v) for v in meta["axis_ray"].split()] == case.meta["axis_ray"]
    assert float.fromhex(meta["angle_to_sun"].strip()) == case.meta["angle_to_sun"]
    ghost = np.fromfile(out + ".ghost.f64", np.float64).reshape(case.H, case.W, 3)
    assert np.array_equal(ghost, case.ghost)            # the ghost buffer: bit for bit
    sample = np.fromfile(out + ".sample.f64", np.float64).reshape(case.H, case.W, 3)
    err = np.abs(sample - case.sample) / np.abs(case.sample)
    assert err.max() <= 1e-9, err.max()                 # north star: 1e-4
    rgba = np.fromfile(out + ".rgba.u32", np.uint32).reshape(case.H, case.W)
    assert np.array_equal(rgba, case.rgba)              # the framebuffer: byte for byte
    from oracle import lfo
    order = np.fromfile(out + ".order.u32", np.uint32)   # the driver visited the pixels like the reference's tiles
    assert np.array_equal(order, lfo.tile_order(case.W, case.H))
