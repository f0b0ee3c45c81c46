#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden fixtures for SURVEY section 8 row f3 (COLLADA -> flat scene).

Runs the REAL reference (oracle/_ref/ref_dump, built by `make -C oracle ref` from the sources under
/root/reference: its ColladaParser, GLScene::Mesh/..Light and SceneObjects classes) on scene files the
reference ships and commits
  tests/golden/collada/<scene>.dae            the input (a data file of the reference, copied)
  tests/golden/collada/<scene>.dump.txt.gz    every triangle / normal / light / camera / material it
                                              ends up with, as hex floats (ref_driver.cpp `collada`)
  tests/golden/collada/sha256.json            sha256 of that dump for ALL shipped scenes (the big
                                              ones are checked in this container only)
  tests/golden/c96x72_pyramid_dae.npz         an end-to-end frame: the reference renders the flat
                                              scene of pyramid.dae (its sun makes the flare)
Only runs in the build container (needs /root/reference)."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF
DUMP = mg.DUMP
OUT = os.path.join(mg.GOLD, "collada")
SMALL = ["pyramid.dae", "sky/CBspheres_lambertian.dae", "sky/CBempty.dae", "sky/CBgems.dae"]


def ref_dump(path, tmp):
    out = os.path.join(tmp, "dump.txt")
    subprocess.run([DUMP, "collada", path, out], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=600)
    return open(out, "rb").read()


def parse_dump(text):
    """-> spheres, tris (ref_driver scene-file tuples), directional lights, point lights"""
    spheres, tris, suns, points = [], [], [], []
    mat = None
    for line in text.decode().splitlines():
        t = line.split()
        f = lambda k, n=3: [float.fromhex(v) for v in t[t.index(k) + 1:t.index(k) + 1 + n]]  # noqa: E731
        bs = lambda: ("e" if t[t.index("bsdf") + 1] == "emission" else "d", *f("rgb"))        # noqa: E731
        if t[0] == "mesh":
            assert t[t.index("bsdf") + 1] in ("diffuse", "emission")
            mat = bs()
        elif t[0] == "tri":
            tris.append(tuple(f("p1") + f("p2") + f("p3") + f("n1") + f("n2") + f("n3")) + mat)
        elif t[0] == "sphere":
            spheres.append(tuple(f("o") + f("r", 1)) + bs())
        elif t[0] == "light" and t[1] == "directional":
            suns.append(f("pos_light") + f("rad"))
        elif t[0] == "light" and t[1] == "point":
            points.append(tuple(f("pos") + f("rad")))
    return spheres, tris, suns, points


def main():
    if not os.path.exists(DUMP):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="lfcollada")
    sha = {}
    for root, _, files in os.walk(os.path.join(REF, "dae")):
        for fn in sorted(files):
            if fn.endswith(".dae"):
                rel = os.path.relpath(os.path.join(root, fn), os.path.join(REF, "dae"))
                d = ref_dump(os.path.join(root, fn), tmp)
                sha[rel] = hashlib.sha256(d).hexdigest()
                if rel in SMALL:
                    base = os.path.basename(rel)
                    shutil.copyfile(os.path.join(root, fn), os.path.join(OUT, base))
                    os.chmod(os.path.join(OUT, base), 0o644)
                    with gzip.GzipFile(os.path.join(OUT, base[:-4] + ".dump.txt.gz"), "wb", mtime=0) as g:
                        g.write(d)
                print("scene", rel, len(d), "bytes", flush=True)
    json.dump(sha, open(os.path.join(OUT, "sha256.json"), "w"), indent=1, sort_keys=True)
    # end to end: the reference renders what its loader made of pyramid.dae
    spheres, tris, suns, points = parse_dump(ref_dump(os.path.join(REF, "dae", "pyramid.dae"), tmp))
    assert suns and tris
    case = dict(name="c96x72_pyramid_dae", W=96, H=72, ns_aa=2, radius=25, intensity=1,
                ap="apertures/pentbiglines.png", gh="bokeh/octagonbokeh.png",
                lights=[], raw_lights=suns, cam=(0.85, -0.08, (3.7, 1.4, 3.3)), visit="tiles",
                scene=dict(spheres=spheres, tris=tris, points=points))
    mg.run_case(case, tmp)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
