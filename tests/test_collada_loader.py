"""COLLADA -> flat scene (lens-flare_amd/host/lf_collada.cpp).

The golden dumps under tests/golden/collada/ were produced by the REAL reference
(oracle/make_golden_collada.py: its ColladaParser, GLScene::Mesh / HalfedgeMesh, the *Light classes
and SceneObjects, driven by oracle/ref_driver.cpp `collada`).  The loader has to reproduce them to
the last bit: every triangle (the reference's rotated vertex order), every area-weighted vertex
normal (including Vertex::computeNormal's boundary walk), the sun's "position", point lights,
cameras, sphere radii (float x double) and materials.  No GPU needed: host code only."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden", "collada")
TOOL = os.path.join(ROOT, "lens-flare_amd", "host", "collada_dump")
SCENES = ["pyramid", "CBspheres_lambertian", "CBempty", "CBgems"]


@pytest.fixture(scope="module")
def tool():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(TOOL)
    return TOOL


def dump(tool, path):
    r = subprocess.run([tool, path], capture_output=True)
    return r.returncode, r.stdout, r.stderr.decode()


@pytest.mark.parametrize("name", SCENES)
def test_flat_scene_is_bit_identical_to_the_reference(tool, name):
    rc, out, err = dump(tool, os.path.join(GOLD, name + ".dae"))
    assert rc == 0, err
    want = gzip.open(os.path.join(GOLD, name + ".dump.txt.gz")).read()
    assert out == want
    assert out.count(b"\ntri ") + out.startswith(b"tri ") > 0 or b"sphere" in out   # not vacuous


def test_every_scene_the_reference_ships(tool):
    """All 21 .dae files of the reference (up to 51k triangles), by the sha256 of the dump; they
    are too big to commit, so this part needs the reference tree (build container only)."""
    ref = os.environ.get("LF_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "dae")):
        pytest.skip("reference tree not present")
    sha = json.load(open(os.path.join(GOLD, "sha256.json")))
    assert len(sha) >= 20
    for rel, want in sorted(sha.items()):
        if os.path.getsize(os.path.join(ref, "dae", rel)) > 2_500_000:
            continue   # the four largest take a minute each; the other 17 cover every code path
        rc, out, err = dump(tool, os.path.join(ref, "dae", rel))
        assert rc == 0, (rel, err)
        assert hashlib.sha256(out).hexdigest() == want, rel


def _write(tmp_path, name, body):
    p = tmp_path / name
    p.write_text(body)
    return str(p)


COLLADA = """<?xml version="1.0"?>
<COLLADA><asset><up_axis>Y_UP</up_axis></asset>
<library_geometries><geometry id="g" name="g"><mesh>
<source id="p"><float_array id="pa" count="%d">%s</float_array></source>
<vertices id="v"><input semantic="POSITION" source="#p"/></vertices>
<triangles count="%d"><input semantic="VERTEX" source="#v" offset="0"/><p>%s</p></triangles>
</mesh></geometry></library_geometries>
<library_visual_scenes><visual_scene id="s"><node id="n" name="n">%s<instance_geometry url="#g"/></node></visual_scene></library_visual_scenes>
<scene><instance_visual_scene url="#s"/></scene></COLLADA>"""
QUAD = (12, "0 0 0 1 0 0 1 1 0 0 1 0", 2)


def test_refuses_what_the_reference_cannot_load(tool, tmp_path):
    """<rotate>/<translate>/<scale> multiply an uninitialised matrix in the reference
    (collada.cpp:262-318); inconsistent orientation / non-manifold meshes make it exit()."""
    ok = _write(tmp_path, "ok.dae", COLLADA % (QUAD[0], QUAD[1], QUAD[2], "0 1 2 0 2 3", "<matrix>1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1</matrix>"))
    rc, out, err = dump(tool, ok)
    assert rc == 0 and out.count(b"tri ") == 2
    rot = _write(tmp_path, "rot.dae", COLLADA % (QUAD[0], QUAD[1], QUAD[2], "0 1 2 0 2 3", '<rotate sid="rotationZ">0 0 1 90</rotate>'))
    rc, out, err = dump(tool, rot)
    assert rc != 0 and "no defined result in the reference" in err
    flipped = _write(tmp_path, "flip.dae", COLLADA % (QUAD[0], QUAD[1], QUAD[2], "0 1 2 0 3 2", ""))
    rc, out, err = dump(tool, flipped)   # the edge (0,2)... is traversed twice in the same direction
    assert rc != 0 and "oriented" in err
    rc, out, err = dump(tool, _write(tmp_path, "bad.dae", "<COLLADA><asset></asset>"))
    assert rc != 0
    rc, out, err = dump(tool, str(tmp_path / "missing.dae"))
    assert rc != 0 and "cannot open" in err


def test_boundary_vertex_normals_follow_the_reference_walk(tool, tmp_path):
    """A single counter-clockwise triangle in the z = 0 plane: all three vertices are on the
    boundary, and the reference's boundary branch (h = h->next()->twin(), halfEdgeMesh.h:498-504)
    walks the boundary loop as well -- the result is the FLIPPED face normal (0, 0, -1).  (Known
    answer taken from the real reference: `ref_dump collada` on this very file.)"""
    one = _write(tmp_path, "one.dae", COLLADA % (9, "0 0 0 1 0 0 0 1 0", 1, "0 1 2", ""))
    rc, out, err = dump(tool, one)
    assert rc == 0, err
    line = [l for l in out.decode().splitlines() if l.startswith("tri")][0].split()
    n1 = [float.fromhex(v) for v in line[line.index("n1") + 1:line.index("n1") + 4]]
    assert n1 == [0.0, 0.0, -1.0]
    # vertex order of the triangle: (p[2], p[0], p[1])
    p1 = [float.fromhex(v) for v in line[line.index("p1") + 1:line.index("p1") + 4]]
    assert p1 == [0.0, 1.0, 0.0]
