#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: reference frames for the environment light and for hemisphere sampling of
the scene term (scene/environment_light.cpp, pathtracer.cpp:86-138, :291-292).

Three frames, each rendered by the REAL reference (oracle/_ref/ref_dump):

  e48x36_envmap_miss      EXACT.  PathTracer::envLight set but not listed as a light; a sphere and a
                          floor lit by the sun and a point light (delta lights: no random draws), all
                          other camera rays read EnvironmentLight::sample_dir.  One run, ns_aa = 4.
  e48x36_envlight         STATISTICAL.  The same geometry, the environment ALSO in scene->lights
                          (as RaytracedRenderer::set_scene does): importance-sampled sample_L with
                          ns_area_light = 4.  Two runs (ns_aa 256 / 255, the second after 100003 discarded
                          draws: the runs share no random number).
  s48x36_cbspheres_hsample  STATISTICAL.  The Cornell box of make_golden_area.py under
                          direct_hemisphere_sample (the -H flag).  Two runs.

The environment map is synthetic (no .exr ships with the reference): a sky gradient, a warm ground
and one bright patch, 32 x 16 texels; it is stored inside the fixtures.  Only runs in the build
container; the fixtures it writes are committed."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_collada as mc  # noqa: E402
from make_golden_area import parse_soft_lights  # noqa: E402


def env_map(w=32, h=16):
    j, i = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    theta = (j + 0.5) / h            # 0 = +y (up) ... 1 = -y
    sky = np.stack([0.25 + 0.3 * theta, 0.35 + 0.3 * theta, 0.7 - 0.2 * theta], -1)
    ground = np.stack([0.30 + 0 * theta, 0.22 + 0 * theta, 0.12 + 0 * theta], -1)
    m = np.where((theta < 0.5)[..., None], sky, ground)
    m = m * (1.0 + 0.25 * np.sin(2 * np.pi * i / w * 3)[..., None])
    patch = np.exp(-(((i - 0.70 * w) / 1.6) ** 2 + ((j - 0.28 * h) / 1.3) ** 2))
    m = m + patch[..., None] * np.array([40.0, 34.0, 22.0])
    return np.ascontiguousarray(m, np.float64)


def render(name, W, H, ns_aa, ns_area, cam, lights, scene, tmp, area=(), env=None, env_as_light=False,
           hemisphere=False, burn=0):
    hf, vf = mg.fit_fov(50.0, 35.0, W, H)
    yaw, pitch, pos = cam
    c2w = mg.rot(yaw, pitch)
    camf = os.path.join(tmp, name + ".cam")
    mg.write_cam(camf, hf, vf, W, H, pos, c2w)
    L = [mg.light_for(ns, c2w, pos, hf, vf, dist) + list(rad) for ns, rad, dist in lights]
    spec = ";".join(",".join(repr(float(v)) for v in l) for l in L)
    sfile = os.path.join(tmp, name + ".scene")
    num = lambda v: v if isinstance(v, str) else repr(float(v))  # noqa: E731
    with open(sfile, "w") as f:
        for s in scene["spheres"]:
            f.write("sphere " + " ".join(num(v) for v in s) + "\n")
        for t in scene["tris"]:
            f.write("tri " + " ".join(num(v) for v in t) + "\n")
        for p in scene.get("points", []):
            f.write("point " + " ".join(num(v) for v in p) + "\n")
        for a in area:
            f.write("area " + " ".join(num(v) for v in a) + "\n")
        if env is not None:
            ef = os.path.join(tmp, name + ".env.f64")
            env.tofile(ef)
            f.write(f"env {env.shape[1]} {env.shape[0]} {ef} {'light' if env_as_light else 'map'}\n")
    out = os.path.join(tmp, name + f"_{ns_aa}")
    e = dict(os.environ, REF_NS_AREA_LIGHT=str(ns_area))
    if hemisphere:
        e["REF_HEMISPHERE"] = "1"
    if burn:
        e["REF_MT_BURN"] = str(burn)   # the second run shares no draw with the first
    subprocess.run([mg.DUMP, "frame", camf, str(W), str(H), str(ns_aa), "25.0", "1.0",
                    os.path.join(mg.REF, "apertures/pentsmall.png"), os.path.join(mg.REF, "bokeh/octagonbokeh.png"),
                    spec, "tiles", out, sfile], check=True, env=e, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=1800, cwd=tmp)   # (probability_debug.png lands in tmp)
    sample = np.fromfile(out + ".sample.f64").reshape(H, W, 3)
    ghost = np.fromfile(out + ".ghost.f64").reshape(H, W, 3)
    meta = dict(W=W, H=H, hFov=hf, vFov=vf, c2w=c2w.reshape(9).tolist(), cam_pos=list(pos), lights=L)
    return sample, ghost, meta


def flat(v):
    return [list(x) for x in v]


def main():
    tmp = tempfile.mkdtemp(prefix="lfenv")
    W, H, ns_area = 48, 36, 4
    env = env_map()
    # a floor (two triangles, normals up) and two spheres in front of the camera
    up = [0.0, 1.0, 0.0] * 3
    floor = [([-3.0, 0.0, 3.0, 3.0, 0.0, 3.0, 3.0, 0.0, -3.0] + up + ["d", 0.55, 0.5, 0.45]),
             ([-3.0, 0.0, 3.0, 3.0, 0.0, -3.0, -3.0, 0.0, -3.0] + up + ["d", 0.55, 0.5, 0.45])]
    spheres = [(0.45, 0.35, 0.1, 0.35, "d", 0.8, 0.3, 0.3), (-0.5, 0.25, -0.2, 0.25, "d", 0.3, 0.5, 0.8)]
    cam = (0.0, -0.12, (0.0, 0.7, 3.2))
    sun = [((0.70, 0.82), (0.9, 0.85, 0.7), 30.0)]
    scene = dict(spheres=spheres, tris=floor, points=[(1.5, 2.0, 1.0, 3.0, 3.0, 3.0)])

    name = "e48x36_envmap_miss"
    s, ghost, meta = render(name, W, H, 4, 1, cam, sun, scene, tmp, env=env, env_as_light=False)
    meta.update(name=name, ns_aa=4, flare_radius=25.0, flare_intensity=1.0, aperture="pentsmall.png",
                ghost_aperture="octagonbokeh.png",
                scene=dict(spheres=flat(spheres), tris=flat(floor), points=flat(scene["points"])))
    np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), sample=s, ghost=ghost, env=env,
                        meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    print(name, "mean", s.mean(), flush=True)

    name = "e48x36_envlight"
    scene2 = dict(spheres=spheres, tris=floor)
    sa, ghost, meta = render(name, W, H, 256, ns_area, cam, sun, scene2, tmp, env=env, env_as_light=True)
    sb, _, _ = render(name, W, H, 255, ns_area, cam, sun, scene2, tmp, env=env, env_as_light=True, burn=100003)
    meta.update(name=name, ns_aa_a=256, ns_aa_b=255, ns_area_light=ns_area, flare_radius=25.0, flare_intensity=1.0,
                aperture="pentsmall.png", ghost_aperture="octagonbokeh.png",
                scene=dict(spheres=flat(spheres), tris=flat(floor)))
    np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), sample_a=sa, sample_b=sb, ghost=ghost, env=env,
                        meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    print(name, "mean", sa.mean(), "rel spread", np.abs(sa - sb).mean() / sa.mean(), flush=True)

    name = "s48x36_cbspheres_hsample"
    dump = gzip.open(os.path.join(mg.GOLD, "collada", "CBspheres_lambertian.dump.txt.gz")).read()
    cs, ct, _, _ = mc.parse_dump(dump)
    area, _ = parse_soft_lights(dump)
    cscene = dict(spheres=list(cs), tris=ct)
    ccam = (0.0, 0.0, (0.0, 0.75, 3.4))
    csun = [((0.80, 0.86), (0.5, 0.5, 0.4), 30.0)]
    sa, ghost, meta = render(name, W, H, 256, ns_area, ccam, csun, cscene, tmp, area=area, hemisphere=True)
    sb, _, _ = render(name, W, H, 255, ns_area, ccam, csun, cscene, tmp, area=area, hemisphere=True, burn=100003)
    meta.update(name=name, ns_aa_a=256, ns_aa_b=255, ns_area_light=ns_area, flare_radius=25.0, flare_intensity=1.0,
                aperture="pentsmall.png", ghost_aperture="octagonbokeh.png",
                scene=dict(spheres=flat(cs), tris=flat(ct), area=area, hemi=[]))
    np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), sample_a=sa, sample_b=sb, ghost=ghost,
                        meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    print(name, "mean", sa.mean(), "rel spread", np.abs(sa - sb).mean() / sa.mean(), flush=True)


if __name__ == "__main__":
    main()
