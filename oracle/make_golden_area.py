#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: reference frames for the SAMPLED lights of the scene term (SURVEY 8 row f2:
AreaLight and InfiniteHemisphereLight through estimate_direct_lighting_importance,
pathtracer.cpp:143-213, light.cpp:35-48, :82-101).

The reference draws these samples from its shared std::mt19937 in hit order, so no device schedule
can reproduce its stream: parity is statistical.  For each scene the REAL reference (oracle/_ref/
ref_dump) renders the frame TWICE (ns_aa 256 and 255, the second run after 100003 discarded draws:
the adaptive early-out otherwise stops both runs at the same sample count pixel after pixel and their
streams never part), which gives the reference's own Monte-Carlo spread per pixel;
tests/test_gpu_area_lights.py requires the device frame to sit inside that spread.

Scene: the reference's own Cornell box dae/sky/CBspheres_lambertian.dae as its loader flattens it
(tests/golden/collada/CBspheres_lambertian.dump.txt.gz, bit-identical to the reference's dump) with
its area light, plus a sun in the frame (the reference dereferences flare_origins[0] unconditionally,
pathtracer.cpp:918).  Only runs in the build container; the fixtures it writes are committed."""
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


def parse_soft_lights(text):
    area, hemi = [], []
    for line in text.decode().splitlines():
        t = line.split()
        f = lambda k, n=3: [float.fromhex(v) for v in t[t.index(k) + 1:t.index(k) + 1 + n]]  # noqa: E731
        if t[:2] == ["light", "area"]:
            area.append(f("pos") + f("dir") + f("dim_x") + f("dim_y") + f("rad"))
        elif t[:2] == ["light", "hemisphere"]:
            hemi.append(f("rad"))
    return area, hemi


def render(name, W, H, ns_aa, ns_area, cam, lights, scene, area, hemi, tmp, burn=0):
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
        for a in area:
            f.write("area " + " ".join(num(v) for v in a) + "\n")
        for h in hemi:
            f.write("hemi " + " ".join(num(v) for v in h) + "\n")
    out = os.path.join(tmp, name + f"_{ns_aa}")
    env = dict(os.environ, REF_NS_AREA_LIGHT=str(ns_area))
    if burn:
        env["REF_MT_BURN"] = str(burn)   # oracle/ref_driver.cpp: discard that many draws first
    subprocess.run([mg.DUMP, "frame", camf, str(W), str(H), str(ns_aa), "25.0", "1.0",
                    os.path.join(mg.REF, "apertures/pentsmall.png"), os.path.join(mg.REF, "bokeh/octagonbokeh.png"),
                    spec, "tiles", out, sfile], check=True, env=env, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=1200)
    sample = np.fromfile(out + ".sample.f64").reshape(H, W, 3)
    ghost = np.fromfile(out + ".ghost.f64").reshape(H, W, 3)
    meta = dict(W=W, H=H, hFov=hf, vFov=vf, c2w=c2w.reshape(9).tolist(), cam_pos=list(pos), lights=L)
    return sample, ghost, meta


def main():
    tmp = tempfile.mkdtemp(prefix="lfarea")
    dump = gzip.open(os.path.join(mg.GOLD, "collada", "CBspheres_lambertian.dump.txt.gz")).read()
    spheres, tris, _, _ = mc.parse_dump(dump)
    area, _ = parse_soft_lights(dump)
    assert len(area) == 1 and len(tris) == 12 and len(spheres) == 2
    # ... plus one sphere with one of the reference's unfilled BSDFs (MirrorBSDF: f() = 0, a black
    # occluder under the direct-lighting integrator), which the device renders as exactly that
    spheres = list(spheres) + [(0.0, 0.22, 0.55, 0.22, "m", 0.9, 0.9, 0.9)]
    scene = dict(spheres=spheres, tris=tris)
    W, H, ns_area = 48, 36, 4
    cam = (0.0, 0.0, (0.0, 0.75, 3.4))
    sun = [((0.80, 0.86), (0.5, 0.5, 0.4), 30.0)]
    for name, a, h in (("a48x36_cbspheres_area", area, []),
                       ("h48x36_cbspheres_hemisphere", [], [[0.6, 0.7, 0.9]])):
        sa, ghost, meta = render(name, W, H, 256, ns_area, cam, sun, scene, a, h, tmp)
        sb, _, _ = render(name, W, H, 255, ns_area, cam, sun, scene, a, h, tmp, burn=100003)
        meta.update(name=name, ns_aa_a=256, ns_aa_b=255, ns_area_light=ns_area, flare_radius=25.0, flare_intensity=1.0,
                    aperture="pentsmall.png", ghost_aperture="octagonbokeh.png",
                    scene=dict(spheres=[list(s) for s in spheres], tris=[list(t) for t in tris], area=a, hemi=h))
        np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), sample_a=sa, sample_b=sb, ghost=ghost,
                            meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
        print(name, "mean", sa.mean(), "rel spread", np.abs(sa - sb).mean() / sa.mean(), flush=True)


if __name__ == "__main__":
    main()
