#!/usr/bin/env python3
"""What a user of the reference's own host application sees: oracle/_ref/ref_app_amd -- the reference's
RaytracedRenderer (raytraced_renderer.cpp compiled as it is: tile queue, worker threads, save_image)
on the replaced PathTracer -- renders dae/pyramid.dae at 1920x1080 to a PNG, wall clock of the whole
process, (a) with the reference's paraxial ghosts, (b) with the geometric march selected by
LF_LENS_FILE at 256 samples per pixel.  Usage (GPU box, repo root): python3 profiles/app_frame_timing.py"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden_app as mga  # noqa: E402

APP = os.path.join(ROOT, "oracle", "_ref", "ref_app_amd")
case = dict(mga.APP_CASES[0], W=1920, H=1080, ns_aa=1, threads=8, ap="pentbig500_14.png")
out = {"binary": "oracle/_ref/ref_app_amd", "frame": "1920x1080, dae/pyramid.dae, ns_aa 1, 8 worker threads"}
for name, env in (("paraxial_ghosts", {}),
                  ("geometric_march_256spp", {"LF_LENS_FILE": os.path.join(ROOT, "lens-flare_amd", "data", "dgauss11.lens"),
                                              "LF_GEOMETRIC_SPP": "256"})):
    tmp = tempfile.mkdtemp(prefix="lfapp")
    png = os.path.join(tmp, "out.png")
    args = mga.app_args(case, tmp, png)[:11]   # no autofocus call
    t0 = time.time()
    r = subprocess.run([APP] + args, capture_output=True, text=True, env=dict(os.environ, **env), cwd=tmp, timeout=900)
    dt = time.time() - t0
    render = [l for l in r.stdout.replace("\r", "\n").splitlines() if "100%!" in l]
    out[name] = {"wall_s_whole_process": dt, "returncode": r.returncode, "png_bytes": os.path.getsize(png) if os.path.exists(png) else 0,
                 "renderer_log": render[-1].strip() if render else None}
print(json.dumps(out, indent=1))
