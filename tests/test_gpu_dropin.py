"""The drop-in, proven: oracle/_ref/ref_dump_amd is the REFERENCE's own objects, unmodified, with ONE
translation unit replaced -- src/pathtracer/pathtracer.cpp by lens-flare_amd/host/pathtracer_amd.cpp,
which implements the reference's own `class CGL::PathTracer` (compiled against its unchanged header,
src/pathtracer/pathtracer.h:25-143) on the MI355X through the C ABI.  The same driver
(oracle/ref_driver.cpp) replays RaytracedRenderer::start_raytracing + raytrace_tile
(raytraced_renderer.cpp:300-311, :622-647) on it, and the buffers it dumps must equal the golden
frames the all-reference build of the very same driver produced (tests/golden/, oracle/make_golden.py).

Round 3 completes the replacement: pathtracer_amd.cpp defines EVERY member the header declares
(tests/test_dropin_symbols.py checks the link), so this file also runs
  * oracle/_ref/ref_app_amd -- the reference's own render controller, RaytracedRenderer compiled from
    raytraced_renderer.cpp as it is (tile queue, worker threads, save_image, autofocus), on the
    replaced PathTracer -- against the PNG the all-reference ref_app wrote (tests/golden/app_*.json);
  * every other public member one by one (`ref_dump members`, tests/golden/members_*.json);
  * the geometric march -- the kernel bench.py measures -- THROUGH the reference's surface, selected
    by LF_LENS_FILE in the environment of the unchanged host or by handing the renderer a
    CGL::LensCamera, bit for bit against the float32 oracle;
  * CGL::LensCamera::generate_ray(s) against lf_generate_lens_rays.

The binaries are built in the build container only (oracle/Makefile `dropin`, `app`; they need the
reference checkout) and travel to the GPU box with the snapshot."""
import base64
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from goldenlib import GOLD, Case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_dump_amd")
APP = os.path.join(ROOT, "oracle", "_ref", "ref_app_amd")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CASES = ["f64x48_pentbiglines", "f97x65_odd_rotcam", "f80x50_two_suns", "f96x64_naive_rgba",
         "s96x64_spheres", "s80x60_tris_rotcam", "c96x72_pyramid_dae", "z40x30_fuzz1", "z44x26_fuzz3",
         "q47x31_fuzz0", "q38x52_fuzz1", "q52x40_fuzz4"]


def _write_inputs(case, tmp, focus="4.7 0"):
    m = case.meta
    cam = tmp / "cam.txt"
    sd = case.H / (2 * math.tan(math.radians(m["vFov"]) / 2))
    with open(cam, "w") as f:     # Camera::load_settings (camera.cpp:228-242)
        f.write(f"{m['hFov']!r} {m['vFov']!r} {case.W / case.H!r} 0.01 100\n")
        f.write(" ".join(repr(float(v)) for v in m["cam_pos"]) + " 0 0 0\n1.5 0.7 5 0.5 100\n")
        f.write(" ".join(repr(float(v)) for v in m["c2w"]) + f"\n{case.W} {case.H} {sd!r}\n{focus}\n")   # focalDistance lensRadius
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
    assert int(meta["host_glue_calls"]) == 0            # nothing on the frame path is host arithmetic
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


# ---- the reference's own render controller on the replaced PathTracer -------------------------------
def _golden_json(prefix):
    return sorted(f for f in os.listdir(GOLD) if f.startswith(prefix) and f.endswith(".json"))


@pytest.mark.parametrize("fixture", _golden_json("app_"))
@pytest.mark.parametrize("threads", [1, 5])
def test_reference_render_controller_with_pathtracer_replaced(fixture, threads, tmp_path):
    """RaytracedRenderer::render_to_file (start_raytracing -> tile queue -> N worker threads calling
    raytrace_pixel / write_to_framebuffer -> save_image) and ::autofocus, compiled from the
    reference's raytraced_renderer.cpp unchanged: the PNG on disk, the sampling-rate PNG and the focal
    distance equal what the all-reference build produced.  With 5 workers too: unlike the reference
    (shared generator), the replacement does not depend on the schedule."""
    import make_golden_app as mga
    assert os.path.exists(APP), "oracle/_ref/ref_app_amd is missing: make -C oracle app (build container)"
    rec = json.load(open(os.path.join(GOLD, fixture)))
    out = str(tmp_path / "out.png")
    args = mga.app_args(dict(rec, threads=threads), str(tmp_path), out)
    r = subprocess.run([APP] + args, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    from PIL import Image
    import io
    got = np.asarray(Image.open(out).convert("RGBA"))
    want = np.asarray(Image.open(io.BytesIO(base64.b64decode(rec["png"]))).convert("RGBA"))
    assert got.shape == want.shape == (rec["H"], rec["W"], 4)
    assert np.array_equal(got, want)                                      # every byte of every pixel
    assert open(out, "rb").read() == base64.b64decode(rec["png"])         # ... and of the file
    assert open(out[:-4] + "_rate.png", "rb").read() == base64.b64decode(rec["rate_png"])
    focal = [float.fromhex(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("FOCAL ")]
    assert focal and focal[0] == pytest.approx(float.fromhex(rec["focal"]), rel=1e-12)


def _numbers(line):
    out = []
    for t in line.split()[1:]:
        out.append(float.fromhex(t) if ("0x" in t or t in ("inf", "-inf", "nan")) else float(t))
    return np.array(out)


@pytest.mark.parametrize("fixture", _golden_json("members_"))
def test_every_other_public_member(fixture, tmp_path):
    """calculate_irradiance_falloff, raytrace_starburst, shift_vertex, compute_phase, draw_ghost,
    rasterize_textured_triangle, fill_textured_pixel, est_radiance_global_illumination, zero / one
    bounce, the importance estimator, at_least_one_bounce_radiance, autofocus: called one by one on
    the replacement, compared with what the reference's own members returned for the same calls."""
    import make_golden_app as mga
    rec = json.load(open(os.path.join(GOLD, fixture)))
    out = str(tmp_path / "members.txt")
    r = subprocess.run([BIN] + mga.member_args(rec, str(tmp_path), out), capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    got = open(out).read().splitlines()
    glue = [l for l in got if l.startswith("host_glue_calls")]
    got = [l for l in got if not l.startswith("host_glue_calls")]
    want = rec["lines"]
    assert [l.split()[0] for l in got] == [l.split()[0] for l in want]
    exact = ("n_flares", "ghost_frame", "ghost_after_draw_red", "ghost_after_draw_blue", "ghost_after_triangle",
             "ghost_after_pixels", "ghost_px_14_10", "zero_bounce", "miss")
    seen = set()
    for g, w in zip(got, want):
        tag = g.split()[0]
        seen.add(tag)
        a, b = _numbers(g), _numbers(w)
        if tag in exact:      # float rasteriser + double accumulation in the reference's order: bit for bit
            assert np.array_equal(a, b), (g, w)
        elif tag == "compute_phase":   # cos / sin of ~1e2 rad: absolute
            assert np.allclose(a, b, rtol=0, atol=1e-12), (g, w)
        else:
            assert np.allclose(a, b, rtol=1e-9, atol=0), (g, w)
    assert {"falloff", "starburst", "shift_vertex", "draw", "est_radiance", "hit", "one_bounce", "importance",
            "at_least_one", "autofocus"} <= {t.replace("ghost_after_draw_red", "draw") for t in seen}
    # the one member that is host glue (dead code in the reference) ran here, 5 hits x once
    assert glue and int(glue[0].split()[1]) >= 1


# ---- the geometric march through the reference's surface --------------------------------------------
def _geometric_frame(tmp_path, env_extra, W=64, H=48):
    case = Case("f64x48_pentbiglines")
    args = _write_inputs(case, tmp_path)
    args[6] = os.path.join(GOLD, "apertures", "pentbig500_14.png")     # the stop's mask = the -x aperture
    env = dict(os.environ, **env_extra)
    r = subprocess.run([BIN, "frame"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = str(tmp_path / "o")
    flares = [[float.fromhex(v) for v in l.split()[1:]] for l in open(out + ".meta.txt") if l.startswith("flare ")]
    ghost = np.fromfile(out + ".ghost.f64", np.float64).reshape(H, W, 3)
    sample = np.fromfile(out + ".sample.f64", np.float64).reshape(H, W, 3)
    return case, flares, ghost, sample


@pytest.mark.parametrize("route", ["LF_LENS_FILE", "LensCamera"])
def test_geometric_march_through_the_reference_surface(route, tmp_path):
    """The kernel the bench line measures, reached from the reference's own call sequence
    (find_sun_pos(); generate_ghost_buffer(); raytrace_pixel ...): ghost_buffer as the host reads it
    equals the float32 oracle's march bit for bit, and sampleBuffer = ghost + starburst."""
    import __graft_entry__ as g
    from goldenlib import load_texels
    from oracle import lfo
    pkg = g.load_package()
    lens_path = os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens")
    spp, key, radius, W, H = 16, 0x51a7, 0.04, 64, 48
    if route == "LF_LENS_FILE":
        env = dict(LF_LENS_FILE=lens_path, LF_GEOMETRIC_SPP=str(spp), LF_GEOMETRIC_KEY=hex(key),
                   LF_SUN_ANGULAR_RADIUS=repr(radius))
    else:
        env = dict(REF_LENS_CAMERA=lens_path, REF_LENS_SPP=str(spp), REF_LENS_SUN_RADIUS=repr(radius),
                   LF_GEOMETRIC_KEY=hex(key))
    case, flares, ghost, sample = _geometric_frame(tmp_path, env)
    assert len(flares) == 1
    lens = pkg.load_lens_file("dgauss11.lens")
    efl = pkg.paraxial_image_scale(lens)      # what lf_set_sun_from_flares(efl_mm <= 0) divides by
    assert efl == pytest.approx(pkg.paraxial_efl(lens), rel=2e-3)
    nx, ny = flares[0][0], flares[0][1]
    sun = [(nx - 0.5) * lens["sensor_width_mm"] / efl, (ny - 0.5) * lens["sensor_width_mm"] * H / W / efl, -1.0]
    mask = load_texels("pentbig500_14.png")
    lf = pkg.LensFlare(0)
    lfo.geo_follow_device(lf)
    try:
        og, _ = lfo.geo_trace(lens, W, H, 0, H, spp, key, None, True, mask, sun, flares[0][2:5], radius)
    finally:
        lfo.geo_follow_device(None)
        lf.close()
    assert og.max() > 0
    assert np.array_equal(ghost, og)
    # the paraxial quads are NOT what filled the buffer
    assert not np.array_equal(ghost, case.ghost)
    # raytrace_pixel composed it: sample - ghost = the frame's starburst (+ scene term), which is the
    # same whichever way the ghosts were made (the same inputs through the paraxial path)
    # (with a lens selected the scene is imaged through it, which samples with the march's counter RNG: the
    # falloff's jitter of the comparison frame must come from the same key)
    (tmp_path / "p").mkdir()
    _, _, ghost_p, sample_p = _geometric_frame(tmp_path / "p", {"LF_COUNTER_JITTER": hex(key)})
    assert np.array_equal(ghost_p[ghost_p != 0] > 0, np.ones((ghost_p != 0).sum(), bool)) and ghost_p.max() > 0
    assert np.allclose(sample - ghost, sample_p - ghost_p, rtol=1e-9, atol=1e-12 * np.abs(sample_p).max())


def test_geometric_march_without_a_sun_clears_the_ghost_buffer(tmp_path):
    """generate_ghost_buffer's early return (pathtracer.cpp:724-726): no light in the frame, no ghosts."""
    case = Case("f64x64_no_sun")
    args = _write_inputs(case, tmp_path)
    env = dict(os.environ, LF_LENS_FILE=os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens"), LF_GEOMETRIC_SPP="4")
    r = subprocess.run([BIN, "frame"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ghost = np.fromfile(str(tmp_path / "o") + ".ghost.f64", np.float64)
    assert ghost.size == case.W * case.H * 3 and not ghost.any()


def test_lens_camera_generate_rays(tmp_path):
    """CGL::LensCamera (a Camera subclass compiled against the reference's camera.h): its rays are
    lf_generate_lens_rays' exit rays carried into world space by the camera's c2w and position."""
    import __graft_entry__ as g
    from goldenlib import load_texels
    pkg = g.load_package()
    case = Case("f97x65_odd_rotcam")
    args = _write_inputs(case, tmp_path)
    cam = args[0]
    rng = np.random.default_rng(11)
    smp = rng.random((64, 4))
    smp[0] = [0.5, 0.5, 0.5, 0.5]                     # the chief ray of the centre pixel
    np.savetxt(tmp_path / "samples.txt", smp, fmt="%.17g")
    lens_path = os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens")
    mask_png = os.path.join(GOLD, "apertures", "pentbig500_14.png")
    out = tmp_path / "rays.txt"
    r = subprocess.run([BIN, "lensrays", lens_path, cam, mask_png, str(tmp_path / "samples.txt"), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(out).read().splitlines()
    rows = np.array([[float.fromhex(v) if "0x" in v else float(v) for v in l.split()] for l in lines[:64]])
    lens = pkg.load_lens_file("dgauss11.lens")
    lf = pkg.LensFlare(0)
    lf.set_frame(64, 64)
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels("pentbig500_14.png"))
    lf.set_lens(lens)
    sw = lens["sensor_width_mm"]
    sh = sw / (case.W / case.H)
    xy = np.stack([-(smp[:, 0] - 0.5) * sw, -(smp[:, 1] - 0.5) * sh], 1).astype(np.float32)
    uv = (2.0 * smp[:, 2:4] - 1.0).astype(np.float32)
    dev = lf.generate_lens_rays(1, xy, uv)
    lf.set_lens_camera(1, 0.001, 1.0)
    z_ep = lf.lens_camera()["entrance_pupil_z_mm"]   # what sits at the camera position
    lf.close()
    c2w = np.array(case.meta["c2w"], float).reshape(3, 3)
    pos = np.array(case.meta["cam_pos"], float)
    alive = dev[:, 7] != 0
    assert 8 < alive.sum() < 64                       # some samples are blocked by the pentagon / vignetted
    assert np.array_equal(rows[:, 7] != 0, alive)
    front = (dev[:, 0:3].astype(float) - [0.0, 0.0, z_ep]) * 0.001     # LensCamera::world_per_mm's default
    assert np.allclose(rows[:, 0:3], pos + front @ c2w.T, rtol=1e-12, atol=1e-12)
    assert np.allclose(rows[:, 3:6], dev[:, 3:6].astype(float) @ c2w.T, rtol=1e-12, atol=1e-12)
    assert np.array_equal(rows[:, 6], dev[:, 6].astype(float))
    assert np.all(rows[:, 8] == 0.01) and np.all(rows[:, 9] == 100.0)     # Camera's clip range
    # the centre pixel's chief ray leaves along the camera's -z axis
    assert alive[0] and np.allclose(rows[0, 3:6], -c2w[:, 2], atol=1e-6)
    single = [float.fromhex(v) if "0x" in v else float(v) for v in lines[64].split()[1:]]
    assert np.array_equal(np.array(single[:7]), rows[0, :7]) and single[7] == 1
    drawn = lines[65].split()
    assert drawn[0] == "drawn" and int(drawn[7]) == 1    # a random pupil point that passes is found


def test_sun_outside_the_frame_renders_the_scene_without_a_flare(tmp_path):
    """teapot.dae from the camera the file declares (3 units away): the sun's image falls above the frame,
    find_sun_pos keeps no flare -- and the REFERENCE then reads flare_origins[0] of an empty vector
    (pathtracer.cpp:918, :968) and dies with SIGSEGV on this very command line (seen again
    while making the fixtures).  The replacement must render the scene term, without a flare, and exit 0."""
    import make_golden_app as mga
    assert os.path.exists(APP), "oracle/_ref/ref_app_amd is missing: make -C oracle app (build container)"
    case = dict(name="teapot_nosun", dae="teapot.dae", cam=mga.cam_meshedit(3.0), W=160, H=90, ns_aa=1, threads=3,
                ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(80, 45))
    out = str(tmp_path / "out.png")
    r = subprocess.run([APP] + mga.app_args(case, str(tmp_path), out), capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    from PIL import Image
    img = np.asarray(Image.open(out).convert("RGB")).astype(int)
    lit = img.max(axis=-1) > 0
    assert 0.1 < lit.mean() < 0.9          # the teapot in front of a black background: no falloff glow, no starburst
    assert img[0, 0].max() == 0 and img[-1, -1].max() == 0


# ---- round 4: the scene imaged through the lens, and the reference's own throughput log ---------------
def test_scene_through_the_lens_from_the_reference_surface(tmp_path):
    """A LensCamera handed to the renderer: the sample loop behind raytrace_pixel marches every sensor
    sample's primary path through the prescription (lf_set_lens_camera) -- sampleBuffer as the host reads
    it equals the frame the C ABI renders when driven directly with the same settings, and differs from
    the pinhole frame (REF_LENS_IMAGE_SCENE=0) by what a lens does: vignetting towards the corners."""
    import __graft_entry__ as g
    from goldenlib import load_texels
    from test_gpu_scene_term import scene_lights
    pkg = g.load_package()
    case = Case("s96x64_spheres")
    m = case.meta
    lens_path = os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens")
    spp, key, radius, wpm = 4, 0x51a7, 0.04, 0.002

    def run(extra, sub, focus="4.7 0"):
        (tmp_path / sub).mkdir()
        args = _write_inputs(case, tmp_path / sub, focus)
        args[6] = os.path.join(GOLD, "apertures", "pentbig500_14.png")
        env = dict(os.environ, REF_LENS_CAMERA=lens_path, REF_LENS_SPP=str(spp), REF_LENS_SUN_RADIUS=repr(radius),
                   LF_GEOMETRIC_KEY=hex(key), REF_LENS_WORLD_PER_MM=repr(wpm), **extra)
        r = subprocess.run([BIN, "frame"] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out = str(tmp_path / sub / "o")
        return (np.fromfile(out + ".sample.f64", np.float64).reshape(case.H, case.W, 3),
                np.fromfile(out + ".ghost.f64", np.float64).reshape(case.H, case.W, 3), r.stdout)

    sample, ghost, stdout = run({}, "lens")
    assert "Lens camera marched" in stdout and "Lens march executed" in stdout
    # the same frame through the C ABI directly
    lf = pkg.LensFlare(0)
    lf.set_frame(case.W, case.H)
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels("pentbig500_14.png"))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.load_lens_file(lens_path)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.find_sun_pos(m["lights"])
    sc = m["scene"]
    spheres = [(1e4, 1e4, 1e4, 1.0, "d", 0.5, 0.5, 0.5)] + [tuple(s) for s in sc["spheres"]]
    lf.set_scene(spheres, [tuple(t) for t in sc["tris"]], scene_lights(case))
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_jitter_counter(key)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    lf.set_sun_from_flares(0, 0.0, radius)
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()
    want = lf.read_buffer(pkg.SAMPLE_BUFFER)
    scene_lens = lf.read_buffer(pkg.SCENE_BUFFER)
    assert np.array_equal(ghost, lf.read_buffer(pkg.GHOST_BUFFER))
    lf.close()
    # (the host hands materials over as the value of BSDF::f, the direct call as a reflectance the device
    # divides by pi itself: the last bit of a shaded value may differ, nothing else)
    assert np.allclose(sample, want, rtol=1e-12, atol=0)
    # Camera::lensRadius > 0 (the reference's switch for its thin-lens stub, the -b flag): the focus follows
    # Camera::focalDistance, counted from the camera position = the entrance pupil's centre -- the frame of a lens
    # refocused at 4.2 units (lf_focus_lens_from_pupil), ghosts included
    sample_f, ghost_f, _ = run({}, "focused", focus="4.2 0.01")
    lf = pkg.LensFlare(0)
    lf.set_frame(case.W, case.H)
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels("pentbig500_14.png"))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.load_lens_file(lens_path)
    sensor_mm = lf.focus_lens_from_pupil(4.2 / wpm)
    assert sensor_mm > 36.2                                   # the file's back focal distance is 36.106 (infinity)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.find_sun_pos(m["lights"])
    lf.set_scene(spheres, [tuple(t) for t in sc["tris"]], scene_lights(case))
    lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
    lf.set_jitter_counter(key)
    lf.set_lens_camera(1, wpm, 0.0)
    lf.render_scene_term()
    lf.set_sun_from_flares(0, 0.0, radius)
    lf.trace_ghosts(spp, key)
    lf.render_flare_layer()
    assert np.array_equal(ghost_f, lf.read_buffer(pkg.GHOST_BUFFER)) and not np.array_equal(ghost_f, ghost)
    assert np.allclose(sample_f, lf.read_buffer(pkg.SAMPLE_BUFFER), rtol=1e-12, atol=0)
    lf.close()
    # the pinhole scene under the same ghosts
    sample_p, ghost_p, _ = run({"REF_LENS_IMAGE_SCENE": "0", "LF_COUNTER_JITTER": hex(key)}, "pinhole")
    assert np.array_equal(ghost_p, ghost)
    scene_pin = sample_p - (sample - scene_lens)
    lum_l, lum_p = scene_lens.sum(axis=-1), scene_pin.sum(axis=-1)
    centre = (slice(case.H // 2 - 8, case.H // 2 + 8), slice(case.W // 2 - 12, case.W // 2 + 12))
    corner = (slice(0, 8), slice(0, 12))
    # (3 samples per pixel of which a quarter passes the pentagon: a coarse check of the exposure only)
    assert 0.6 < lum_l[centre].sum() / lum_p[centre].sum() < 1.6      # calibrated: the same exposure on the axis
    assert lum_l[corner].sum() < 0.9 * lum_p[corner].sum()            # a 36 mm lens on a 47 mm sensor vignettes


def test_the_reference_log_line_reports_the_device_work(tmp_path):
    """RaytracedRenderer's end-of-frame log (raytraced_renderer.cpp:706-709: BVH rays, rays per second,
    intersection tests per ray) is fed from the scene kernel's counters: real numbers, not '0 rays, nan'."""
    import make_golden_app as mga
    rec = json.load(open(os.path.join(GOLD, "app_pyramid_96x72.json")))
    out = str(tmp_path / "out.png")
    args = mga.app_args(dict(rec, threads=3), str(tmp_path), out)
    r = subprocess.run([APP] + args, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    rays = [int(v) for v in re.findall(r"BVH traced (\d+) rays", r.stdout)]
    per = [float(v) for v in re.findall(r"Averaged ([0-9.eE+-]+|nan|-nan) intersection tests per ray", r.stdout)
           if "nan" not in v]
    speed = [float(v) for v in re.findall(r"Average speed ([0-9.eE+-]+) million rays per second", r.stdout)]
    assert rays and per and speed, r.stdout[-1500:]
    W, H, ns = rec["W"], rec["H"], rec["ns_aa"]
    # at least one camera ray per sample; a hit adds one shadow ray per light
    assert rays[0] >= W * H * ns and rays[0] <= W * H * ns * 8
    assert 0.0 < per[0] < 64.0 and speed[0] > 0.0
    # ... and they ARE the device's counters: what the reference's log line prints = what the scene kernel counted
    dev = re.findall(r"Scene kernel traced (\d+) rays with (\d+) primitive tests", r.stdout)
    assert dev and int(dev[0][0]) == rays[0]
    assert abs(int(dev[0][1]) / int(dev[0][0]) - per[0]) <= 1e-5 * per[0] + 1e-6
    # ... and with a lens selected by LF_LENS_FILE the march's own line appears beside it
    env = dict(os.environ, LF_LENS_FILE=os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens"), LF_GEOMETRIC_SPP="8")
    r2 = subprocess.run([APP] + args, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    ev = re.findall(r"Lens march executed (\d+) ray-surface intersections in ([0-9.]+) ms", r2.stdout)
    assert ev and int(ev[0][0]) > W * H * 8 and float(ev[0][1]) > 0.0
    cam = re.findall(r"Lens camera marched (\d+) sensor samples, (\d+) left the front element", r2.stdout)
    assert cam and int(cam[0][0]) == W * H * ns and 0 < int(cam[0][1]) < int(cam[0][0])
