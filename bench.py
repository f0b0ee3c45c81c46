#!/usr/bin/env python3
"""bench.py -- headline benchmark of the lens-flare hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one its metric is quoted on): Double-Gauss 11-interface
prescription (lens-flare_amd/data/dgauss11.lens), primary path + all 45 ghost pairs, 3 wavelengths,
pentagon aperture mask (final_apertures/pentbig500_14.png), 1920x1080, 256 sensor samples/pixel,
one synthetic sun at normalised screen position (0.521445, 0.517156).

One step = one full frame of the hot path: find_sun_pos -> geometric ghost march (the dominant
kernel) -> starburst/falloff/compose flare layer; with N > 1 the 8-row sensor tile rows are dealt
round-robin to the ranks (strong scaling: the frame is fixed) and the finished tile rows are
exchanged with in-place RCCL all-gathers over xGMI.  `value` = executed ray-surface events of the whole
job per second (device counters, not an upper bound).  Inputs are resident in HBM before the timed
region starts.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 1024 SIMD-32 x 2.4 GHz (one wave64 VALU op / 2 clk)
SUN_NS = (0.521445, 0.517156)
TILE_ROWS = 8   # the march kernel's sensor tile is 8x8 pixels


def load_mask():
    from goldenlib import load_texels
    return load_texels("pentbig500_14.png")


def sun_direction(lens, W, H):
    """Lens-space direction towards a sun that a pinhole of the lens' focal length images at
    normalised screen position SUN_NS (same convention as Camera::analyze_world_coord)."""
    efl = 50.358
    ex = 0.5 * lens["sensor_width_mm"] / efl
    ey = ex * H / W
    return [(2 * SUN_NS[0] - 1) * ex, (2 * SUN_NS[1] - 1) * ey, -1.0]


class DevView:
    """Expose a device allocation of the library to torch (zero copy) for the RCCL gather."""

    def __init__(self, ptr, n_doubles):
        self.__cuda_array_interface__ = {"shape": (n_doubles,), "typestr": "<f8",
                                         "data": (ptr, False), "version": 2}


def cpu_baseline(lens, mask, sun, W, H, target_s):
    """The CPU oracle (kind 'port': oracle/lf_geo_oracle.c) timed on this host's cores on a bounded
    sample of the same workload: a band of rows of the same frame at reduced spp."""
    from oracle import lfo
    cores = min(os.cpu_count() or 1, 64)
    rows = (H // 2 - 32, H // 2 + 32)
    t0 = time.time()
    _, c = lfo.geo_trace(lens, W, H, rows[0], rows[0] + 4, 1, 1, None, True, mask, sun,
                         [1.0, 0.9, 0.5], 0.05, n_threads=cores)
    dt = max(time.time() - t0, 1e-3)
    rate = c["surface_events"] / dt
    per_sample = c["surface_events"] / (4 * W)
    spp = int(max(1, min(64, target_s * rate / (per_sample * (rows[1] - rows[0]) * W))))
    t0 = time.time()
    _, c = lfo.geo_trace(lens, W, H, rows[0], rows[1], spp, 1, None, True, mask, sun,
                         [1.0, 0.9, 0.5], 0.05, n_threads=cores)
    dt = time.time() - t0
    return {"value": c["surface_events"] / dt / 1e6, "unit": "Mray-surface-intersections/s",
            "cores": cores, "kind": "port",
            "sample": f"rows {rows[0]}..{rows[1]} of the {W}x{H} frame, {spp} of the spp, "
                      f"46 paths x 3 wavelengths, {c['surface_events']} events in {dt:.1f} s "
                      f"(oracle/lf_geo_oracle.c, OpenMP)"}


def reference_flare_path(pkg):
    """Side datum (never `value`): the REAL reference's own CPU renderer (oracle/_ref/ref_dump, the
    reference hot path compiled from its own sources, 1 thread) on the golden 64x48 frame, next to
    the same frame through the C ABI on the GPU.  None when the prebuilt binary is absent."""
    import subprocess
    import tempfile
    from goldenlib import GOLD, Case, load_texels
    dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(dump):
        return None
    try:
        case = Case("f64x48_pentbiglines")
        m = case.meta
        tmp = tempfile.mkdtemp(prefix="lfref")
        cam = os.path.join(tmp, "cam.txt")
        sd = case.H / (2 * math.tan(math.radians(m["vFov"]) / 2))
        with open(cam, "w") as f:
            f.write(f"{m['hFov']!r} {m['vFov']!r} {case.W / case.H!r} 0.01 100\n")
            f.write(" ".join(repr(float(v)) for v in m["cam_pos"]) + " 0 0 0\n1.5 0.7 5 0.5 100\n")
            f.write(" ".join(repr(float(v)) for v in m["c2w"]) + f"\n{case.W} {case.H} {sd!r}\n4.7 0\n")
        spec = ";".join(",".join(repr(float(v)) for v in l) for l in m["lights"])
        ap_png = os.path.join(GOLD, "apertures", m["aperture"])
        gh_png = os.path.join(GOLD, "apertures", m["ghost_aperture"])
        t0 = time.time()
        subprocess.run([dump, "frame", cam, str(case.W), str(case.H), str(m["ns_aa"]),
                        repr(float(m["flare_radius"])), repr(float(m["flare_intensity"])), ap_png,
                        gh_png, spec, "tiles", os.path.join(tmp, "o")], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_ref = time.time() - t0
        got = np.fromfile(os.path.join(tmp, "o.sample.f64")).reshape(case.H, case.W, 3)
        lf = pkg.LensFlare(0)
        lf.set_frame(case.W, case.H)
        lf.set_params(m["ns_aa"], m["flare_radius"], m["flare_intensity"])
        lf.set_aperture(pkg.APERTURE_STARBURST, load_texels(m["aperture"]))
        lf.set_aperture(pkg.APERTURE_GHOST, load_texels(m["ghost_aperture"]))
        lf.set_camera(m["c2w"], m["cam_pos"], m["hFov"], m["vFov"])
        lf.set_jitter_mt19937(5489, None)
        lf.synchronize()
        t0 = time.perf_counter()
        lf.find_sun_pos(m["lights"])
        lf.generate_ghost_buffer()
        lf.render_flare_layer()
        lf.synchronize()
        t_gpu = time.perf_counter() - t0
        dev = lf.read_buffer(pkg.SAMPLE_BUFFER)
        lf.close()
        return {"frame": "64x48, apertures/pentbiglines.png (80x78 bbox), 1 sun, ns_aa 1",
                "reference_cpu_s": t_ref, "cores": 1, "kind": "reference",
                "gpu_s": t_gpu, "max_rel_diff": float((np.abs(dev - got) / np.abs(got)).max()),
                "note": "reference time includes its PNG decode; its cost is the per-pixel direct "
                        "DFT (pathtracer.cpp:947-974), the device path gathers from one DFT of the aperture"}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # LF_BENCH_BACKEND=gloo is a REHEARSAL mode for a 1-GPU box (several ranks share GPU 0 and the
    # tile-row exchange is staged through host memory); the real run uses nccl (= RCCL over xGMI).
    backend = os.environ.get("LF_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    W, H, spp = args.width, args.height, args.spp
    lens = pkg.load_lens_file(os.environ.get("LF_BENCH_LENS", "dgauss11.lens"))  # env: experiments only
    mask = load_mask()
    sun = sun_direction(lens, W, H)

    cpu, ref_path = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(lens, mask, sun, W, H, args.cpu_seconds)
        ref_path = reference_flare_path(pkg)

    lf = pkg.LensFlare(local)
    lf.set_frame(W, H)
    lf.set_params(1, 25.0, 1.0)
    lf.set_aperture(pkg.APERTURE_STARBURST, mask)       # stop mask + starburst spectrum (set-up)
    lf.set_aperture(pkg.APERTURE_GHOST, mask)
    lf.set_lens(lens)
    lf.set_sun(sun, [1.0, 0.9, 0.5], 0.05)
    lf.set_ghost_pairs(None, True)
    lf.set_jitter_counter(0x1e45f1a4e)
    # camera looking down -z from the origin; the sun where analyze_world_coord puts SUN_NS
    hf = 2 * math.degrees(math.atan(0.5 * lens["sensor_width_mm"] / 50.358))
    vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
    lf.set_camera(np.eye(3), [0, 0, 0], hf, vf)
    ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
    lights = [[(2 * SUN_NS[0] - 1) * ex * 10, (2 * SUN_NS[1] - 1) * ey * 10, -10.0, 1.0, 0.9, 0.5]]

    # sensor tile rows (8 rows each) are dealt round-robin: tile row t belongs to rank t % world.
    # One march launch per frame covers all of this rank's tile rows.
    from lens_flare_amd import sharding
    my_trows = len(sharding.my_tile_rows(H, rank, world))
    lf.set_band(0, H)
    lf.set_row_interleave(rank, world)
    frame_t, scratch = None, {}
    if world > 1:
        ptr, nbytes = lf.device_buffer(pkg.SAMPLE_BUFFER)
        frame_t = torch.as_tensor(DevView(ptr, nbytes // 8), device=f"cuda:{local}")

    def one_frame():
        lf.find_sun_pos(lights)
        lf.trace_ghosts(spp, 0x1e45f1a4e)
        lf.render_flare_layer()
        if world > 1:
            # the real exchange step: every rank ends up with the whole frame -- ONE all-gather per
            # frame (lens_flare_amd/sharding.py: pack this rank's tile rows, gather, unpack); the
            # buffer is padded to 64 rows so the last group of tile rows never runs past the end.
            lf.synchronize()
            if backend == "nccl":
                sharding.gather_frame(frame_t, W, H, rank, world, dist, scratch=scratch)
            else:
                host = frame_t.cpu()
                sharding.gather_frame(host, W, H, rank, world, dist)
                frame_t.copy_(host)
            torch.cuda.synchronize()   # the next frame rewrites these rows on the library's stream

    def barrier():
        lf.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_frame()
    barrier()
    lf.reset_counters()
    lf.timing_reset()
    lf.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_frame()
    barrier()
    dt = time.perf_counter() - t0
    lf.timing_enable(False)

    cnt = lf.counters()
    n_launch, march_ms = lf.timing_get("march")
    ev = torch.tensor([float(cnt["surface_events"]), float(cnt["rays_launched"]),
                       float(lf.executed_events()), dt],
                      dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
    if world > 1:
        tot = ev.clone()
        dist.all_reduce(tot[:3], op=dist.ReduceOp.SUM)
        mx = ev[3:].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        events, rays, executed, dt = float(tot[0]), float(tot[1]), float(tot[2]), float(mx[0])
    else:
        events, rays, executed = float(ev[0]), float(ev[1]), float(ev[2])

    if rank == 0:
        # roofline of the dominant kernel (the march): algorithmic HBM bytes per launch =
        # framebuffer rows it writes (f64 RGB) + the aperture mask + the lens/pair tables it reads
        rows_per_launch = min(my_trows * TILE_ROWS, H)
        alg_bytes = rows_per_launch * W * 24 + mask.size * 4 + 4096
        avg_ms = march_ms / max(n_launch, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, slots = None, None
        tf = os.path.join(ROOT, "profiles", "r01_march_pmc.json")
        if os.path.exists(tf):
            try:
                pmc = json.load(open(tf))
                slots = pmc.get("valu_lane_slots_per_event")
                # PMC traffic is per launch of the profiled 1-GPU run (whole frame per launch)
                traffic = pmc.get("hbm_bytes_per_launch") if world == 1 else None
            except Exception:
                traffic = None
        # the VALU accounting is on the events the device computes (shared legs once), not on the
        # per-path count that `value` reports
        ev_per_launch = executed / max(1, n_launch * world)
        ev_rate_gpu = ev_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        valu = {"bound": "valu", "unit": "wave-instr/s", "peak": VALU_PEAK_LANEOPS / 64.0,
                "practical_peak": 1.05e12, "executed_events_per_s_per_gpu": ev_rate_gpu,
                "note": "peak = 1024 SIMD-32 x 2.4 GHz / 2 clk per wave64 op; practical_peak = "
                        "profiles/microbench/valu_issue.hip (8 waves/SIMD, independent v_fma_f32); "
                        "instructions per event from the rocprofv3 PMC pass in profiles/"}
        if slots:
            valu["lane_slots_per_event"] = slots
            valu["achieved"] = ev_rate_gpu * slots / 64.0
            valu["frac"] = valu["achieved"] / valu["peak"]
            valu["frac_of_practical"] = valu["achieved"] / valu["practical_peak"]
        out = {
            "metric": "Mray-surface-intersections/s + frame time, 1080p 256spp double-Gauss",
            "value": events / dt / 1e6,
            "unit": "Mray-surface-intersections/s",
            # the same frames counted by the intersections the device actually computes (legs
            # shared by the paths of one sample are computed once), see config.note
            "value_computed_only": executed / dt / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"double-Gauss 11 interfaces (dgauss11.lens), primary + 45 ghost "
                                   f"pairs x 3 wavelengths, pentagon mask pentbig500_14, {W}x{H}, "
                                   f"{spp} spp, one sun (BASELINE.json configs[2])",
                       "parallelism": f"{world} GPU(s), 8-row sensor tile rows dealt round-robin"
                                      + (", one RCCL all_gather per frame"
                                         if world > 1 else ""),
                       "rays_per_frame": rays / args.steps,
                       "events_per_frame": events / args.steps,
                       "events_executed_per_frame": executed / args.steps,
                       "note": "value counts every path's ray-surface intersections on their own "
                               "(what the per-path CPU oracle counts and does); the device computes "
                               "the legs that the paths of one sample share once "
                               "(events_executed_per_frame)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_march", "launches": n_launch, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "register-resident march: compulsory traffic is O(frame); the "
                                 "binding resource is FP32 VALU issue, see valu"},
            "valu": valu,
            "cpu_baseline": cpu,
            "reference_flare_path": ref_path,
        }
        print(json.dumps(out), flush=True)
    lf.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
