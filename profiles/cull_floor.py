#!/usr/bin/env python3
"""Floor of a table-driven path cull on the bench frame (instrumented build -DLF_MARCH_LIT_MAP, never shipped): the
full march of c3 under N keys records which (sensor block, pupil cell at resolution P, path) combinations ever end
inside the sun's lobe; the dump (stderr, LIT_MAP lines) gives the fraction of (entry, path) bits that are needed at
each granularity -- what no pre-pass at that granularity can go below.
    LF_LIB=lens-flare_amd/build_ab/litmap/liblensflare_hip.so LF_LIT_MAP=1 python3 profiles/cull_floor.py 2> gpurun_out/r05_cull_floor.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package()
W, H, SPP = 1920, 1080, 256
lens = pkg.load_lens_file("dgauss11.lens")
mask = pkg.load_aperture_png("pentbig500_14.png")
lf = pkg.LensFlare(0)
lf.set_frame(W, H)
lf.set_aperture(pkg.APERTURE_STARBURST, mask)
lf.set_lens(lens)
lf.set_sun(bench.sun_direction(lens, pkg.paraxial_efl(lens), W, H), [1.0, 0.9, 0.5], 0.05)
lf.set_march_culling(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for k in range(N):
    if k == N - 1:
        os.environ["LF_LIT_MAP_DUMP"] = "1"
    lf.trace_ghosts(SPP, 0x7000 + k)
    lf.synchronize()
    print(f"key {k} done", file=sys.stderr, flush=True)
os.environ["LF_LIT_MAP_DUMP"] = "1"
lf.trace_ghosts(1, 1)      # (the dump happens before a launch: one more, tiny)
lf.synchronize()
lf.close()
