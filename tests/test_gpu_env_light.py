"""Environment light and hemisphere sampling of the scene term on the device:
EnvironmentLight::sample_dir for camera rays that hit nothing (pathtracer.cpp:291-292,
environment_light.cpp:173-182) -- deterministic, EXACT against the real reference; the
importance-sampled EnvironmentLight::sample_L (:140-171) and estimate_direct_lighting_hemisphere
(pathtracer.cpp:86-138, the -H flag) -- drawn from the reference's shared generator in hit order, so
STATISTICAL parity against two reference renders, as in test_gpu_area_lights.py
(fixtures: oracle/make_golden_env.py)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from goldenlib import GOLD, load_texels
from test_gpu_area_lights import _inside_reference_spread, _light_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _setup(pkg, m, ns_aa):
    lf = pkg.LensFlare(0)
    lf.set_frame(m["W"], m["H"])
    lf.set_params(ns_aa, m["flare_radius"], m["flare_intensity"])
    lf.set_sampling(32, 0.05, 0.01, 100.0)
    lf.set_paraxial_lens()
    lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
    lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
    lf.set_flares(np.zeros((0, 2)), np.zeros((0, 3)), (0.0, 0.0), 0.0)
    lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
    lf.find_sun_pos(m["lights"])
    sc = m["scene"]
    spheres = [(1e4, 1e4, 1e4, 1.0, "d", 0.5, 0.5, 0.5)] + [tuple(s) for s in sc["spheres"]]
    lf.set_scene(spheres, [tuple(t) for t in sc["tris"]], [])
    return lf


def _sun_rows(m):
    rows = []
    for l in m["lights"]:
        p = np.array(l[:3])
        d = p / np.sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2])
        rows.append([0.0] + list(l[3:6]) + d.tolist() + [0.0] * 9)
    return rows


def test_environment_map_behind_the_scene_is_exact(pkg):
    """envLight set, not listed as a light: hits are lit by the sun and a point light (no random
    draws), every other camera ray returns the bilinear map value -- the reference's MT19937 pixel
    jitter, the whole frame to 1e-9 of the frame the reference rendered."""
    z, m = _load("e48x36_envmap_miss")
    lf = _setup(pkg, m, m["ns_aa"])
    rows = _sun_rows(m) + [[1.0] + list(p[3:6]) + list(p[0:3]) + [0.0] * 9 for p in m["scene"]["points"]]
    lf.set_scene_lights(rows)
    lf.set_environment_map(z["env"])
    lf.set_jitter_mt19937(5489, None)
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    got = lf.read_buffer(pkg.SAMPLE_BUFFER)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), z["ghost"])
    ref = z["sample"]
    err = np.abs(got - ref) / np.abs(ref)
    assert err.max() <= 1e-9, err.max()
    # the map is what most of the frame shows: without it those pixels are far off
    lf.set_environment_map(None)
    lf.render_scene_term()
    lf.render_flare_layer()
    bare = lf.read_buffer(pkg.SAMPLE_BUFFER)
    assert (np.abs(bare - ref).max(axis=-1) > 0.1).mean() > 0.3
    # a listed environment light without a map is a state error, and a map without light is refused
    lf.set_scene_lights(rows + [[4.0] + [0.0] * 15])
    lf.set_jitter_counter(1)
    with pytest.raises(pkg.LensFlareError):
        lf.render_scene_term()
    with pytest.raises(pkg.LensFlareError):
        lf.set_environment_map(np.zeros((4, 8, 3)))
    with pytest.raises(pkg.LensFlareError):
        lf.set_environment_map(np.ones((1, 8, 3)))
    lf.close()


def test_environment_light_importance_sampling_within_the_references_spread(pkg):
    z, m = _load("e48x36_envlight")
    lf = _setup(pkg, m, m["ns_aa_a"])
    lf.set_scene_lights(_sun_rows(m) + [[4.0] + [0.0] * 15])
    lf.set_light_samples(m["ns_area_light"])
    lf.set_environment_map(z["env"])
    lf.set_jitter_mt19937(5489, None)
    with pytest.raises(pkg.LensFlareError):     # sampled light: the reference's stream cannot be reproduced
        lf.render_scene_term()
    lf.set_jitter_counter(77)
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    got = lf.read_buffer(pkg.SAMPLE_BUFFER)
    assert np.array_equal(lf.read_buffer(pkg.GHOST_BUFFER), z["ghost"])
    _inside_reference_spread(got, z["sample_a"], z["sample_b"])
    # ... and the sampled part matters: with the environment only behind the scene (not a light) the
    # floor and the spheres lose its light and the frame leaves the spread
    lf.set_scene_lights(_sun_rows(m))
    lf.render_scene_term()
    lf.render_flare_layer()
    dark = lf.read_buffer(pkg.SAMPLE_BUFFER)
    ref = 0.5 * (z["sample_a"] + z["sample_b"])
    assert (np.abs(dark - ref).sum(axis=-1) > 0.1 * ref.sum(axis=-1)).mean() > 0.2
    lf.close()


def test_hemisphere_sampling_within_the_references_spread(pkg):
    z, m = _load("s48x36_cbspheres_hsample")
    lf = _setup(pkg, m, m["ns_aa_a"])
    lf.set_scene_lights(_light_rows(m))
    lf.set_light_samples(m["ns_area_light"])
    lf.set_direct_hemisphere_sample(True)
    lf.set_jitter_mt19937(5489, None)
    with pytest.raises(pkg.LensFlareError):
        lf.render_scene_term()
    lf.set_jitter_counter(5)
    lf.render_scene_term()
    lf.generate_ghost_buffer()
    lf.render_flare_layer()
    got = lf.read_buffer(pkg.SAMPLE_BUFFER)
    _inside_reference_spread(got, z["sample_a"], z["sample_b"], max_rel_spread=0.08)
    # light sampling (the default) gives a visibly different frame of the same scene: the area light's
    # one-sidedness and the sun are not seen by the hemisphere estimator (pathtracer.cpp:91-92)
    lf.set_direct_hemisphere_sample(False)
    lf.render_scene_term()
    lf.render_flare_layer()
    other = lf.read_buffer(pkg.SAMPLE_BUFFER)
    ref = 0.5 * (z["sample_a"] + z["sample_b"])
    assert (np.abs(other - ref).sum(axis=-1) > 0.1 * ref.sum(axis=-1)).mean() > 0.2
    lf.close()


def test_drop_in_binary_forwards_environment_and_hemisphere_flag(pkg, tmp_path):
    """The reference's own objects with pathtracer.o replaced (oracle/_ref/ref_dump_amd): the
    drop-in reads PathTracer::envLight (its private map through the explicit-instantiation accessor),
    finds it in scene->lights and forwards direct_hemisphere_sample."""
    binary = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "ref_dump_amd")
    assert os.path.exists(binary), "oracle/_ref/ref_dump_amd is missing: make -C oracle dropin (build container)"
    num = lambda v: v if isinstance(v, str) else repr(float(v))  # noqa: E731
    for name, hemisphere in (("e48x36_envlight", False), ("s48x36_cbspheres_hsample", True)):
        z, m = _load(name)
        W, H = m["W"], m["H"]
        d = tmp_path / name
        d.mkdir()
        sd = H / (2 * math.tan(math.radians(m["vFov"]) / 2))
        with open(d / "cam.txt", "w") as f:
            f.write(f"{m['hFov']!r} {m['vFov']!r} {W / H!r} 0.01 100\n")
            f.write(" ".join(repr(float(v)) for v in m["cam_pos"]) + " 0 0 0\n1.5 0.7 5 0.5 100\n")
            f.write(" ".join(repr(float(v)) for v in m["c2w"]) + f"\n{W} {H} {sd!r}\n4.7 0\n")
        with open(d / "scene.txt", "w") as f:
            for s in m["scene"]["spheres"]:
                f.write("sphere " + " ".join(num(v) for v in s) + "\n")
            for t in m["scene"]["tris"]:
                f.write("tri " + " ".join(num(v) for v in t) + "\n")
            for a in m["scene"].get("area", []):
                f.write("area " + " ".join(num(v) for v in a) + "\n")
            if "env" in z.files:
                z["env"].tofile(str(d / "env.f64"))
                f.write(f"env {z['env'].shape[1]} {z['env'].shape[0]} {d / 'env.f64'} light\n")
        spec = ";".join(",".join(repr(float(v)) for v in l) for l in m["lights"])
        env = dict(os.environ, LF_COUNTER_JITTER="1", REF_NS_AREA_LIGHT=str(m["ns_area_light"]))
        if hemisphere:
            env["REF_HEMISPHERE"] = "1"
        r = subprocess.run([binary, "frame", str(d / "cam.txt"), str(W), str(H), str(m["ns_aa_a"]), "25.0", "1.0",
                            os.path.join(GOLD, "apertures", m["aperture"]),
                            os.path.join(GOLD, "apertures", m["ghost_aperture"]),
                            spec, "tiles", str(d / "o"), str(d / "scene.txt")], env=env, cwd=str(d),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.fromfile(str(d / "o") + ".sample.f64").reshape(H, W, 3)
        assert np.array_equal(np.fromfile(str(d / "o") + ".ghost.f64").reshape(H, W, 3), z["ghost"])
        _inside_reference_spread(got, z["sample_a"], z["sample_b"], max_rel_spread=0.08 if hemisphere else 0.03)
