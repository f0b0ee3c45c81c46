#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generate tests/golden/ from the REAL reference hot path.

Runs oracle/_ref/ref_dump (built by `make -C oracle ref` from the reference's own sources under
/root/reference) and packs its raw dumps into small fixtures.  Only runs in the build container
(the reference does not travel); the fixtures it writes are committed.

    python oracle/make_golden.py            # all cases (~2 min on one core)

Fixtures are DATA: inputs (camera, lights, aperture PNG assets) and the reference's outputs.
"""
import hashlib
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("LF_REFERENCE", "/root/reference")
DUMP = os.path.join(HERE, "_ref", "ref_dump")
GOLD = os.path.join(ROOT, "tests", "golden")

PNGS = ["apertures/naive.png", "apertures/pentbiglines.png", "apertures/pentsmall.png",
        "apertures/pentsmalllines.png", "apertures/pentsmalllottalines.png",
        "bokeh/octagonbokeh.png"] + ["final_apertures/" + n for n in (
            "pent2_18.png", "pent2_8.png", "pent3_18.png", "pent4_10.png", "pent4_13.png",
            "pent4_15.png", "pent4_17.png", "pent_11.png", "pentbig2_500_9.png",
            "pentbig4_500_17.png", "pentbig500_14.png")]


def fit_fov(hf, vf, W, H):
    """Camera::configure (src/pathtracer/camera.cpp:69-88)."""
    ar1 = math.tan(math.radians(hf) / 2) / math.tan(math.radians(vf) / 2)
    ar = W / H
    if ar1 < ar:
        hf = 2 * math.degrees(math.atan(math.tan(math.radians(vf) / 2) * ar))
    elif ar1 > ar:
        vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) / ar))
    return hf, vf


def rot(yaw, pitch):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return ry @ rx


def light_for(ns, c2w, pos, hf, vf, dist):
    """World position that analyze_world_coord projects to normalised screen coords `ns`."""
    ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
    pc = np.array([(2 * ns[0] - 1) * ex * dist, (2 * ns[1] - 1) * ey * dist, -dist])
    return (np.asarray(c2w) @ pc + np.asarray(pos)).tolist()


def write_cam(path, hf, vf, W, H, pos, c2w):
    sd = H / (2 * math.tan(math.radians(vf) / 2))
    with open(path, "w") as f:
        f.write(f"{hf!r} {vf!r} {W / H!r} 0.01 100\n")
        f.write(" ".join(repr(float(v)) for v in pos) + " 0 0 0\n")
        f.write("1.5 0.7 5 0.5 100\n")
        f.write(" ".join(repr(float(v)) for v in np.asarray(c2w).reshape(9)) + "\n")
        f.write(f"{W} {H} {sd!r}\n4.7 0\n")


def parse_meta(path):
    m = {"flares": []}
    for line in open(path):
        t = line.split()
        if t[0] == "flare":
            m["flares"].append([float.fromhex(v) for v in t[1:]])
        elif t[0] in ("axis_ray",):
            m["axis_ray"] = [float.fromhex(v) for v in t[1:]]
        elif t[0] == "angle_to_sun":
            m["angle_to_sun"] = float.fromhex(t[1])
        else:
            m[t[0]] = int(t[1])
    return m


CASES = [
    # name, W, H, ns_aa, radius, intensity, aperture, ghost, [(ns, radiance, dist)], cam(yaw,pitch,pos), visit
    dict(name="f64x48_pentbiglines", W=64, H=48, ns_aa=1, radius=25, intensity=1,
         ap="apertures/pentbiglines.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.521445, 0.517156), (1, 0.9, 0.5), 10)], cam=(0, 0, (0, 0, 0)), visit="tiles"),
    dict(name="f97x65_odd_rotcam", W=97, H=65, ns_aa=2, radius=20, intensity=0.5,
         ap="apertures/pentsmalllines.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.25, 0.7), (0.8, 0.8, 1.0), 25)], cam=(0.6, -0.25, (1.5, -2, 3)), visit="tiles"),
    dict(name="f96x64_naive_rgba", W=96, H=64, ns_aa=1, radius=25, intensity=3.5,
         ap="apertures/naive.png", gh="final_apertures/pent4_10.png",
         lights=[((0.61, 0.33), (2, 1.5, 1), 40)], cam=(-1.1, 0.3, (0, 1, 0)), visit="tiles"),
    dict(name="f80x50_two_suns", W=80, H=50, ns_aa=1, radius=12, intensity=1,
         ap="apertures/pentsmall.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.3, 0.4), (1, 0.2, 0.1), 10), ((1.4, 0.5), (9, 9, 9), 10),
                 ((0.7, 0.6), (0.1, 0.5, 1.0), 30)], cam=(0.2, 0.1, (0, 0, 0)), visit="tiles"),
    dict(name="f64x64_no_sun", W=64, H=64, ns_aa=1, radius=25, intensity=1,
         ap="apertures/pentsmall.png", gh="bokeh/octagonbokeh.png",
         lights=[((1.63, 0.5), (1, 1, 1), 10)], cam=(0, 0, (0, 0, 0)), visit="tiles"),
    dict(name="f256_pentbiglines", W=256, H=256, ns_aa=1, radius=25, intensity=1,
         ap="apertures/pentbiglines.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.519978, 0.517027), (1, 0.9, 0.5), 12.9)], cam=(0.85, -0.08, (3.7, 1.4, 3.3)),
         visit="tiles"),
    dict(name="f1080p_pentbig500_14_scatter", W=1920, H=1080, ns_aa=1, radius=25, intensity=1,
         ap="final_apertures/pentbig500_14.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.521445, 0.517156), (1, 0.9, 0.5), 10)], cam=(0, 0, (0, 0, 0)), visit="list"),
    dict(name="f4k_pentbiglines_scatter", W=3840, H=2160, ns_aa=1, radius=25, intensity=1,
         ap="apertures/pentbiglines.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.32, 0.71), (1, 1, 1), 10)], cam=(0, 0, (0, 0, 0)), visit="list"),
    # ---- scene-radiance term: geometry in front of the camera, delta lights ----
    dict(name="s96x64_spheres", W=96, H=64, ns_aa=3, radius=25, intensity=1,
         ap="apertures/pentsmall.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.62, 0.71), (2.0, 1.8, 1.5), 30)], cam=(0, 0, (0, 0, 0)), visit="tiles",
         scene=dict(
             spheres=[(0, -101.0, -6, 100.0, "d", 0.6, 0.6, 0.55), (-0.9, -0.4, -5, 0.6, "d", 0.8, 0.2, 0.2),
                      (0.7, -0.55, -4.2, 0.45, "d", 0.2, 0.7, 0.3), (0.1, 0.35, -6.5, 0.5, "e", 1.5, 1.2, 0.4),
                      (1.6, 0.2, -7.0, 0.9, "d", 0.3, 0.3, 0.9)],
             tris=[], points=[(-2.0, 3.0, -3.0, 6.0, 6.0, 5.0)])),
    dict(name="s80x60_tris_rotcam", W=80, H=60, ns_aa=2, radius=20, intensity=1,
         ap="apertures/pentsmall.png", gh="bokeh/octagonbokeh.png",
         lights=[((0.3, 0.8), (1.5, 1.5, 1.5), 40)], cam=(0.5, -0.2, (2.0, 1.0, 1.5)), visit="tiles",
         scene="tris"),
]


def tri_scene(c2w, pos):
    """A floor quad (2 triangles, vertex normals tilted so interpolation matters), a tilted
    triangle, an emissive triangle and two spheres, placed in front of the (rotated) camera."""
    R, p = np.asarray(c2w), np.asarray(pos, float)

    def w(v):  # camera space -> world
        return (R @ np.asarray(v, float) + p).tolist()

    def n(v):
        v = R @ np.asarray(v, float)
        return (v / np.linalg.norm(v)).tolist()

    q = [w((-3, -1, -2)), w((3, -1, -2)), w((3, -1.4, -9)), w((-3, -1.4, -9))]
    qn = [n((0.1, 1, 0.05)), n((-0.1, 1, 0.05)), n((-0.1, 1, -0.1)), n((0.1, 1, -0.1))]
    tris = [q[0] + q[1] + q[2] + qn[0] + qn[1] + qn[2] + ["d", 0.7, 0.7, 0.7],
            q[0] + q[2] + q[3] + qn[0] + qn[2] + qn[3] + ["d", 0.7, 0.7, 0.7]]
    t = [w((-1.5, -0.9, -5)), w((0.2, -0.8, -4.5)), w((-0.8, 1.0, -5.5))]
    tn = n(np.cross(np.subtract((0.2, -0.8, -4.5), (-1.5, -0.9, -5)), np.subtract((-0.8, 1.0, -5.5), (-1.5, -0.9, -5))))
    tris.append(t[0] + t[1] + t[2] + tn + tn + tn + ["d", 0.2, 0.4, 0.9])
    e = [w((1.0, 0.8, -6)), w((1.8, 0.9, -6.2)), w((1.3, 1.6, -6.1))]
    tris.append(e[0] + e[1] + e[2] + n((0, 0, 1)) * 3 + ["e", 0.5, 2.0, 0.5])
    spheres = [tuple(w((1.0, -0.6, -4.0))) + (0.5, "d", 0.9, 0.5, 0.1),
               tuple(w((-0.3, -0.2, -7.5))) + (0.8, "d", 0.5, 0.5, 0.5)]
    points = [tuple(w((0.5, 2.5, -3.0))) + (5.0, 4.0, 3.0)]
    return dict(spheres=spheres, tris=tris, points=points)


def run_case(c, tmp):
    W, H = c["W"], c["H"]
    hf, vf = fit_fov(50.0, 35.0, W, H)
    yaw, pitch, pos = c["cam"]
    c2w = rot(yaw, pitch)
    cam = os.path.join(tmp, c["name"] + ".cam")
    write_cam(cam, hf, vf, W, H, pos, c2w)
    lights = []
    for ns, rad, dist in c["lights"]:
        lights.append(light_for(ns, c2w, pos, hf, vf, dist) + list(rad))
    lights += [list(l) for l in c.get("raw_lights", [])]   # world-space posLight + radiance as given
    spec = ";".join(",".join(repr(float(v)) for v in l) for l in lights)
    visit = "tiles"
    order_xy = None
    if c["visit"] == "list":
        rng = np.random.RandomState(12345)
        n = 384
        xs = rng.randint(0, W, n)
        ys = rng.randint(0, H, n)
        # make sure the flare centre, its neighbourhood and the far field are all hit
        fx, fy = int(math.ceil(c["lights"][0][0][0] * W)), int(math.ceil(c["lights"][0][0][1] * H))
        near = [(fx, fy), (fx - 1, fy), (fx + 3, fy - 2), (fx + 20, fy + 10), (fx - 24, fy),
                (fx + 200, fy - 100), (fx - 249, fy), (fx - 251, fy), (0, 0), (W - 1, H - 1),
                (W // 2, H // 2), (W // 2 - 1, H // 2 - 1), (W // 2, 0), (0, H // 2)]
        for k, (x, y) in enumerate(near):
            xs[k], ys[k] = min(max(x, 0), W - 1), min(max(y, 0), H - 1)
        order_xy = np.stack([xs, ys], 1)
        lst = os.path.join(tmp, c["name"] + ".list")
        np.savetxt(lst, order_xy, fmt="%d")
        visit = "list:" + lst
    out = os.path.join(tmp, c["name"])
    cmd = [DUMP, "frame", cam, str(W), str(H), str(c["ns_aa"]), repr(float(c["radius"])),
           repr(float(c["intensity"])), os.path.join(REF, c["ap"]), os.path.join(REF, c["gh"]),
           spec, visit, out]
    scene = c.get("scene")
    if scene == "tris":
        scene = tri_scene(c2w, pos)
    if scene:
        sfile = os.path.join(tmp, c["name"] + ".scene")
        with open(sfile, "w") as f:
            for s in scene["spheres"]:
                f.write("sphere " + " ".join(v if isinstance(v, str) else repr(float(v)) for v in s) + "\n")
            for t in scene["tris"]:
                f.write("tri " + " ".join(v if isinstance(v, str) else repr(float(v)) for v in t) + "\n")
            for p in scene["points"]:
                f.write("point " + " ".join(repr(float(v)) for v in p) + "\n")
        cmd.append(sfile)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    meta = parse_meta(out + ".meta.txt")
    if scene:
        # what the device needs for the scene term: geometry, materials and, per light, what
        # SceneLight::sample_L uses (DirectionalLight: dirToLight = unit(posLight), src/scene/light.cpp:11-24)
        meta["scene"] = dict(spheres=[list(s) for s in scene["spheres"]],
                             tris=[list(t) for t in scene["tris"]],
                             points=[list(p) for p in scene["points"]])
    meta.update(dict(name=c["name"], flare_radius=c["radius"], flare_intensity=c["intensity"],
                     aperture=os.path.basename(c["ap"]), ghost_aperture=os.path.basename(c["gh"]),
                     hFov=hf, vFov=vf, cam_pos=list(map(float, pos)),
                     c2w=np.asarray(c2w).reshape(9).tolist(), lights=lights, visit=c["visit"]))
    arrays = {}
    ghost = np.fromfile(out + ".ghost.f64", np.float64).reshape(H, W, 3)
    nz = np.flatnonzero(ghost.reshape(-1))
    arrays["ghost_idx"] = nz.astype(np.uint32)
    arrays["ghost_val"] = ghost.reshape(-1)[nz]
    if meta["n_flares"] > 0:
        sample = np.fromfile(out + ".sample.f64", np.float64).reshape(H, W, 3)
        rgba = np.fromfile(out + ".rgba.u32", np.uint32).reshape(H, W)
        order = np.fromfile(out + ".order.u32", np.uint32)
        if c["visit"] == "list":
            arrays["order"] = order
            # a pixel visited twice keeps the value of its LAST visit
            arrays["sample_at_order"] = sample.reshape(-1, 3)[order]
            arrays["rgba_at_order"] = rgba.reshape(-1)[order]
        else:
            arrays["sample"] = sample
            arrays["rgba"] = rgba
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, c["name"] + ".npz"), **arrays)
    print("case", c["name"], "n_flares", meta["n_flares"], "ghost nz", len(nz), flush=True)


def main():
    if not os.path.exists(DUMP):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    os.makedirs(os.path.join(GOLD, "apertures"), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="lfgold")
    # G7: aperture statistics + texel checksums; the PNG assets themselves are data fixtures
    stats = {}
    for rel in PNGS:
        name = os.path.basename(rel)
        shutil.copyfile(os.path.join(REF, rel), os.path.join(GOLD, "apertures", name))
        os.chmod(os.path.join(GOLD, "apertures", name), 0o644)
        raw = os.path.join(tmp, name + ".f32")
        r = subprocess.run([DUMP, "aperture", os.path.join(REF, rel), raw], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stderr.splitlines() if l.startswith("APERTURE")][0].split()
        tex = np.fromfile(raw, np.float32)
        stats[name] = dict(width=int(line[1]), height=int(line[2]), min_x=int(line[3]),
                           min_y=int(line[4]), max_x=int(line[5]), max_y=int(line[6]),
                           total_value=line[7], sha256_f32=hashlib.sha256(tex.tobytes()).hexdigest())
    json.dump(stats, open(os.path.join(GOLD, "apertures.json"), "w"), indent=1, sort_keys=True)
    subprocess.check_call([DUMP, "trace", os.path.join(GOLD, "paraxial_trace.txt")])
    subprocess.check_call([DUMP, "convert", os.path.join(GOLD, "convert_coordinate.txt")])
    only = sys.argv[1:]
    for c in CASES:
        if only and c["name"] not in only:
            continue
        run_case(c, tmp)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
