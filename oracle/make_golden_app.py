#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden fixtures for the completed drop-in (round 3).

Both come from binaries made ONLY of the reference's own objects (oracle/Makefile `ref`, `app`):

  tests/golden/app_<case>.json (+ .png, _rate.png)
      oracle/_ref/ref_app = the reference's RaytracedRenderer (raytraced_renderer.cpp compiled as it
      is) driven headless the way main.cpp + Application drive it: COLLADA file -> render_to_file ->
      the PNG lodepng wrote, the sampling-rate PNG beside it, and the focal distance
      RaytracedRenderer::autofocus leaves in the camera.
  tests/golden/members_<case>.json
      oracle/_ref/ref_dump members = every other public member of PathTracer called one by one
      (ghost / starburst helpers, single-ray integrator queries, autofocus), results as hex floats.

tests/test_gpu_dropin.py runs the SAME commands on the binaries whose pathtracer.o is replaced by
lens-flare_amd/host/pathtracer_amd.cpp.  Only runs in the build container (needs /root/reference)."""
import base64
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

APP = os.path.join(HERE, "_ref", "ref_app")

# Camera::load_settings files (camera.cpp:228-242): the one of SURVEY appendix D (pyramid.dae's sun at
# normalised (0.519978, 0.517027)) and a second view that leaves the sun near the frame's corner
CAM_D = """40.0 40.0 1 0.01 100
0.0 1.0 0.0 3.7355586634545626 1.418441757928949 3.2970453389528775
1.5 0.7 5 0.5 100
-0.6617304335094221 -0.06274465907959714 -0.7471117326909125 0.0 0.9964919767910092 -0.08368835158578981 0.7497418444821072 -0.05537912917455362 -0.6594090677905755
{W} {H} 351.7
4.7 0
"""

# the camera maxplanck.dae itself declares (5 units in front of the bust, looking down -z; its sun stands
# behind that camera: a frame without a flare, all scene term -- 50 801 triangles through the
# reference's own BVH on one side and through the device's on the other)
CAM_MAXPLANCK = """39.5978 22.8952 1.7777777778 0.01 10000
0.0 0.0 5.0 0.0 0.0 0.0
0.0 0.0 5.0 0.5 100
1.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 1.0
{W} {H} 300.0
4.7 0
"""

def cam_meshedit(z):
    """the default camera the meshedit scenes declare: on the z axis, looking at the origin"""
    sgn = 1.0 if z > 0 else -1.0
    return ("39.5978 22.8952 1.7777777778 0.01 10000\n0.0 0.0 %r 0.0 0.0 0.0\n0.0 0.0 %r 0.5 100\n"
            "%r 0.0 0.0 0.0 1.0 0.0 0.0 0.0 %r\n{W} {H} 300.0\n4.7 0\n" % (z, abs(z), sgn, sgn))


APP_CASES = [
    # a frame that is no multiple of the 32-pixel tile, 2 camera rays per pixel, one worker (the only
    # configuration in which the reference itself is reproducible: its workers share one generator)
    dict(name="pyramid_96x72", dae="pyramid.dae", cam=CAM_D, W=96, H=72, ns_aa=2, threads=1,
         ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(48, 20)),
    dict(name="pyramid_130x70", dae="pyramid.dae", cam=CAM_D, W=130, H=70, ns_aa=1, threads=1,
         ap="pentsmalllines.png", gh="pent4_10.png", radius=12, intensity=2.5, autofocus=(10, 60)),
    dict(name="maxplanck_192x108", dae="maxplanck.dae", dae_dir="data", cam=CAM_MAXPLANCK, W=192, H=108, ns_aa=1,
         threads=1, ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(96, 54)),
    # two more of the meshes the reference ships (dae/meshedit), their own cameras and sun
    dict(name="teapot_160x90", dae="teapot.dae", cam=cam_meshedit(5.0), W=160, H=90, ns_aa=2, threads=1,
         ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(80, 45)),
    dict(name="cow_128x72", dae="cow.dae", cam=cam_meshedit(5.0), W=128, H=72, ns_aa=1, threads=1,
         ap="pentsmalllines.png", gh="pent4_10.png", radius=12, intensity=2.5, autofocus=(64, 36)),
] + ([
    # one-off sweep (LF_GOLDEN_EXTRA=1, the .dae copied next to the others for the occasion; not committed:
    # profiles/r03_mesh_sweep.log)
    dict(name="beetle_160x90", dae="beetle.dae", cam=cam_meshedit(5.0), W=160, H=90, ns_aa=1, threads=1,
         ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(80, 45)),
    dict(name="peter_160x90", dae="peter.dae", cam=cam_meshedit(-5.0), W=160, H=90, ns_aa=1, threads=1,
         ap="pentbiglines.png", gh="octagonbokeh.png", radius=25, intensity=1, autofocus=(80, 45)),
] if os.environ.get("LF_GOLDEN_EXTRA") == "1" else [])

MEMBER_CASES = [
    dict(name="spheres_96x72", cam=CAM_D, W=96, H=72, ap="pentbiglines.png", gh="octagonbokeh.png",
         light="9.39094,2.22422,8.5358,1,0.944,0.544",
         scene="sphere 3.4 1.3 3.6 0.8 d 0.7 0.5 0.3\nsphere 4.2 2.3 3.0 0.4 e 2.0 1.5 1.0\n"
               "sphere 0.0 -100.0 0.0 100.0 d 0.4 0.6 0.4\npoint 2.0 4.0 2.0 8.0 7.0 6.0\n"),
]


def png(name):
    return os.path.join(mg.GOLD, "apertures", name)


def app_args(case, tmp, out):
    """argv of ref_app / ref_app_amd for a case (shared with the test)."""
    cam = os.path.join(tmp, "cam.txt")
    with open(cam, "w") as f:
        f.write(case["cam"].format(W=case["W"], H=case["H"]))
    dae = (os.path.join(os.path.dirname(HERE), "lens-flare_amd", "data", case["dae"]) if case.get("dae_dir") == "data"
           else os.path.join(mg.GOLD, "collada", case["dae"]))
    return [dae, cam, str(case["W"]), str(case["H"]), str(case["ns_aa"]),
            str(case["threads"]), png(case["ap"]), png(case["gh"]), repr(float(case["radius"])),
            repr(float(case["intensity"])), out] + [repr(float(v)) for v in case["autofocus"]]


def member_args(case, tmp, out):
    cam, scene = os.path.join(tmp, "cam.txt"), os.path.join(tmp, "scene.txt")
    with open(cam, "w") as f:
        f.write(case["cam"].format(W=case["W"], H=case["H"]))
    with open(scene, "w") as f:
        f.write(case["scene"])
    return ["members", cam, str(case["W"]), str(case["H"]), png(case["ap"]), png(case["gh"]), case["light"], scene, out]


def main():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref", "app"])
    for case in APP_CASES:
        tmp = tempfile.mkdtemp(prefix="lfapp")
        out = os.path.join(tmp, "out.png")
        r = subprocess.run([APP] + app_args(case, tmp, out), capture_output=True, text=True, timeout=3600, cwd=tmp)
        assert r.returncode == 0, r.stderr[-2000:]
        focal = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("FOCAL ")]
        rec = dict(case, focal=focal[0],
                   png=base64.b64encode(open(out, "rb").read()).decode(),
                   rate_png=base64.b64encode(open(out[:-4] + "_rate.png", "rb").read()).decode())
        json.dump(rec, open(os.path.join(mg.GOLD, f"app_{case['name']}.json"), "w"), indent=1)
        print("app", case["name"], len(rec["png"]), "b64 bytes, focal", focal[0], flush=True)
    for case in MEMBER_CASES:
        tmp = tempfile.mkdtemp(prefix="lfmem")
        out = os.path.join(tmp, "members.txt")
        r = subprocess.run([mg.DUMP] + member_args(case, tmp, out), capture_output=True, text=True, timeout=3600, cwd=tmp)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = dict(case, lines=open(out).read().splitlines())
        json.dump(rec, open(os.path.join(mg.GOLD, f"members_{case['name']}.json"), "w"), indent=1)
        print("members", case["name"], len(rec["lines"]), "lines", flush=True)


if __name__ == "__main__":
    main()
