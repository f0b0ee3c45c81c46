#!/usr/bin/env python3
"""bench.py -- headline benchmark of the lens-flare hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4_1gpu|c5_1gpu]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself, as child
processes, before it touches the GPU: lens_flare_amd.sharding.self_launch.)

Default workload = BASELINE.json configs[2], the one its metric is quoted on ("c3"): Double-Gauss
11-interface prescription (lens-flare_amd/data/dgauss11.lens), primary path + all 45 ghost pairs,
3 wavelengths, pentagon aperture mask (final_apertures/pentbig500_14.png), 1920x1080, 256 sensor
samples per pixel, one synthetic sun at normalised screen position (0.521445, 0.517156).  The other
single-GPU configurations of BASELINE.json are selectable with --config (see CONFIGS).

One step = one full frame of the hot path: find_sun_pos -> [scene term] -> geometric ghost march
(the dominant kernel) -> starburst / falloff / compose flare layer; with N > 1 the 8-row sensor
tile rows are dealt round-robin to the ranks (strong scaling: the frame is fixed) and the finished
tile rows are exchanged with ONE all-gather per frame (RCCL over xGMI).

`value` = ray-surface intersections the device EXECUTED per second, whole job (device counter
lf_get_executed_events; the reference's own metric also counts rays actually traced,
raytraced_renderer.cpp:706-709).  The paths of one sensor sample share legs, which the march
computes once; counting every path on its own (what a per-path tracer such as the CPU baseline
does) gives `value_logical`, 3.9x larger on c3 -- an algorithmic saving, reported as such, never as
throughput.  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CLOCK_HZ = 2.4e9
VALU_PEAK = 256 * 4 * CLOCK_HZ / 2.0      # wave64 instructions / s: 1024 SIMD-32, 2 clk each
VALU_PRACTICAL = 1.05e12                  # profiles/microbench/valu_issue.hip (independent v_fma_f32)
# What one SIMD spends per wave64 instruction when 8 waves issue (profiles/microbench/valu_issue_result_3.txt):
# a dependent v_fma_f32 chain sustains 1.06e12 instr/s on the 1024 SIMDs; a v_sqrt_f32 INSIDE such a chain
# (7 fma + 1 root: 6.78e11; the march's event mix of 25 + 2: 7.88e11, wherever the roots are placed) costs
# 5.4 ns = 13 clocks -- the quarter rate (8.1 clocks back to back) plus the switch in and out of the
# transcendental pipe -- and nothing overlaps it.
T_VALU_NS = 1024.0 / 1.06e12 * 1e9        # 0.966 ns per SIMD
T_TRANS_NS = 5.4
SCALAR_PEAK = 256 * CLOCK_HZ              # one scalar unit per CU, one instruction per clock
FP32_PEAK_TFLOPS = 256 * 4 * 32 * 2 * CLOCK_HZ / 1e12   # 157.3: 32 lanes x FMA per clock and SIMD (dense vector FP32)
# arithmetic of ONE executed event as lf_march.hip's surface_event<false> codes it (a refraction at a curved
# interface, the common row): 15 FMA + 8 multiplies / adds (38 flop) + 2 compares + 2 square roots = the
# kernel's 27 vector instructions; the Fresnel weight of surface_event<true> adds 20 flop (17 instructions)
# where it is evaluated.  SURVEY 8d guessed ~80 flop + 2 sqrt + 2 rcp for an event WITH its weight.
FLOP_PER_EVENT = 38.0
FLOP_PER_FRESNEL = 20.0
SUN_NS = (0.521445, 0.517156)
ABLATION_FILE = os.path.join(ROOT, "profiles", "r04_all_weights_ablation.json")


def every_event_weighted():
    """The like-for-like figure for SURVEY 8d's unit (intersect + refract/reflect + Fresnel on EVERY event), as
    recorded by the ablation build (never shipped: its pixels are wrong); None if the record is absent."""
    try:
        with open(ABLATION_FILE) as f:
            rec = json.load(f)
        return {"recorded_in": os.path.relpath(ABLATION_FILE, ROOT), "config": "c3",
                "ms_per_frame": rec["all_weights"]["ms_per_frame"],
                "executed_events_per_s": rec["all_weights"]["executed_events_per_s"],
                "time_over_shipped": rec["ratio_time"],
                "note": "not measured by this run: -DLF_MARCH_ALL_WEIGHTS timing build, same box as its shipped leg"}
    except Exception:  # noqa: BLE001
        return None
TILE_ROWS = 8   # the march kernel's sensor tile is 8x8 pixels
PMC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r06_march_pmc.json", "r05_march_pmc.json", "r04_march_pmc.json", "r03_march_pmc.json", "r02_march_pmc.json")]   # newest first
R04_C3_MS = 105.22        # BENCH_r04.json: the c3 frame of round 4 (every sample marches every path), driver-timed

# BASELINE.json configs[1..4]; configs[3] / [4] are 8-GPU jobs there, here one GPU's whole frame
CONFIGS = {
    "c2": dict(W=1920, H=1080, spp=64, pairs="reference", n_lambda=3, scene=None, spectral=False,
               text="BASELINE.json configs[1]: double-Gauss 11 interfaces, pentagon mask, 1080p 64 spp, "
                    "primary + the reference's pair enumeration (both mirrors on one side of the stop, "
                    "pathtracer.cpp:735-762: 20 pairs) x 3 wavelengths"),
    "c3": dict(W=1920, H=1080, spp=256, pairs="all", n_lambda=3, scene=None, spectral=False,
               text="BASELINE.json configs[2]: double-Gauss 11 interfaces (dgauss11.lens), primary + 45 "
                    "ghost pairs x 3 wavelengths, pentagon mask pentbig500_14, 1920x1080, 256 spp, one sun"),
    "c4_1gpu": dict(W=3840, H=2160, spp=256, pairs="all", n_lambda=3, scene="pyramid.dae", spectral=False,
                    text="BASELINE.json configs[3] on ONE GPU: 4K 256 spp, the sun is the scene's "
                         "DirectionalLight (dae/dragon.dae is absent from the reference checkout: "
                         "dae/pyramid.dae, the project's own sun scene); every sensor sample's primary path "
                         "images the SCENE through the prescription (lf_set_lens_camera: BVH, direct lighting, "
                         "weighted by the transmitted fraction) and its ghost paths collect the sun (sun handed to "
                         "the march by lf_set_sun_from_flares)"),
    "c4_maxplanck_1gpu": dict(W=3840, H=2160, spp=256, pairs="all", n_lambda=3, scene="maxplanck.dae", spectral=False,
                              behind_mesh=True,
                              text="configs[3]'s shape on the LARGEST scene file the reference ships that loads "
                                   "(dae/meshedit/maxplanck.dae: 50 801 triangles, its own DirectionalLight is the sun): "
                                   "4K 256 spp on ONE GPU, camera behind the mesh looking at the sun, the scene imaged "
                                   "through the prescription (lf_set_lens_camera), sun handed to the march by "
                                   "lf_set_sun_from_flares"),
    "c5_1gpu": dict(W=3840, H=2160, spp=1024, pairs="all", n_lambda=8, scene=None, spectral=True,
                    text="BASELINE.json configs[4] on ONE GPU: 8 wavelengths (dgauss11_8lambda.lens: 2-term Cauchy "
                         "fit through each glass's C, d, F indices) + spectral starburst, 4K 1024 spp"),
}


def sun_direction(lens, efl, W, H):
    """Lens-space direction towards a sun that a lens of focal length efl images at normalised
    screen position SUN_NS (what lf_set_sun_from_flares computes from a flare)."""
    sw = lens["sensor_width_mm"]
    return [(SUN_NS[0] - 0.5) * sw / efl, (SUN_NS[1] - 0.5) * sw * H / W / efl, -1.0]


def lens_8_lambda(pkg):
    """C5's prescription: the committed 8-column file (Cauchy fit through the C, d, F indices of every
    glass; lens-flare_amd/data/make_spectral_lens.py), tent weights of the wavelengths onto R,
    G, B and the starburst's scale lambda_d / lambda."""
    lens = pkg.load_lens_file("dgauss11_8lambda.lens")
    w8, scale = pkg.spectral_weights(lens["lambda_nm"])
    return lens, w8, scale


def pair_list(lens, kind):
    if kind == "all":
        return None
    stop = lens["stop"]
    return [(i, j) for i in range(stop) for j in range(i + 1, stop)] + \
           [(i, j) for i in range(stop + 1, lens["n"]) for j in range(i + 1, lens["n"])]


class DevView:
    """Expose a device allocation of the library to torch (zero copy) for the RCCL gather."""

    def __init__(self, ptr, n_doubles):
        self.__cuda_array_interface__ = {"shape": (n_doubles,), "typestr": "<f8",
                                         "data": (ptr, False), "version": 2}


class DevViewI64(DevView):
    """... the cull table's u64 masks, as the int64 torch can move"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def source_sha():
    """Identity of the shipped march kernel: the PMC-derived figures are only quoted as this
    binary's when the profile in profiles/ was taken from the same sources and compile flags."""
    h = hashlib.sha256()
    for f in ("lens-flare_amd/csrc/lf_march.hip", "lens-flare_amd/csrc/lf_cull.hip", "lens-flare_amd/csrc/lf_march_common.h",
              "lens-flare_amd/csrc/lf_march_events.h", "lens-flare_amd/csrc/lf_internal.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    mk = open(os.path.join(ROOT, "lens-flare_amd", "Makefile")).read()
    h.update(mk[mk.index("FLAGS  :="):mk.index("SRCS   :=")].encode())
    return h.hexdigest()[:16]


def host_cpu_info():
    """What the host offers: logical CPUs, the affinity mask, a cgroup CPU quota if there is one, and the
    topology lscpu reports (threads per core, cores, sockets, NUMA nodes)."""
    import subprocess
    info = {"logical": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        info["affinity"] = info["logical"]
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cgroup_quota_cpus"] = None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001
        info["cgroup_quota_cpus"] = None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        info["threads_per_core"] = int(kv.get("Thread(s) per core", "1"))
        info["cores_per_socket"] = int(kv.get("Core(s) per socket", "0"))
        info["sockets"] = int(kv.get("Socket(s)", "1"))
        info["numa_nodes"] = int(kv.get("NUMA node(s)", "1"))
        info["model"] = kv.get("Model name", "")
    except Exception:  # noqa: BLE001
        pass
    phys = info.get("cores_per_socket", 0) * info.get("sockets", 1)
    info["physical_cores"] = phys if phys > 0 else max(1, info["logical"] // max(1, info.get("threads_per_core", 1)))
    return info


def cpu_leg(config, W, H, rows, spp):
    """One timed leg of the CPU baseline, run in a CHILD process (bench.py --cpu-leg ...) whose environment pins the
    OpenMP team (OMP_NUM_THREADS / OMP_PROC_BIND=close / OMP_PLACES=cores are read when libgomp loads): the oracle's
    march of rows [rows[0], rows[1]) of the config's frame at spp samples, every path on its own.  Never touches the GPU."""
    import __graft_entry__ as g
    pkg = g.load_package()
    from oracle import lfo
    cfg = CONFIGS[config]
    lens = pkg.load_lens_file("dgauss11.lens")
    mask = pkg.load_aperture_png("pentbig500_14.png")
    lambda_rgb = None
    if cfg["n_lambda"] == 8:
        lens, lambda_rgb, _ = lens_8_lambda(pkg)
    sun = sun_direction(lens, pkg.paraxial_efl(lens), W, H)
    pairs = pair_list(lens, cfg["pairs"])
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    t0 = time.time()
    _, c = lfo.geo_trace(lens, W, H, rows[0], rows[1], spp, 1, pairs, True, mask, sun, [1.0, 0.9, 0.5], 0.05,
                         n_threads=threads, lambda_rgb=lambda_rgb, cull=None)
    print(json.dumps({"events": c["surface_events"], "seconds": max(time.time() - t0, 1e-6), "threads": threads}))


def cpu_baseline(config, W, H, target_s):
    """The CPU oracle (kind 'port': oracle/lf_geo_oracle.c, one path at a time like any per-path tracer -- the
    reference has no geometric lens to time) on this host's cores, on a bounded sample of the same workload: a band
    of rows of the same frame at reduced spp.  Every leg is a child process with a PINNED OpenMP team
    (OMP_PROC_BIND=close, OMP_PLACES=cores) that runs >= 3 s; `value` = the team of all physical cores the
    process may use (affinity mask, cgroup quota), `value_t1` = one thread, the sweep is side data."""
    import subprocess
    info = host_cpu_info()
    usable = min(info["affinity"], info["logical"])
    if info.get("cgroup_quota_cpus"):
        usable = max(1, min(usable, int(info["cgroup_quota_cpus"])))
    full_team = max(1, min(usable, info["physical_cores"]))
    rows = (H // 2 - 32, H // 2 + 32)

    def leg(threads, rows_, spp):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_DYNAMIC="false")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", config, "--width", str(W), "--height", str(H),
                            "--cpu-leg", json.dumps({"rows": list(rows_), "spp": spp})], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            raise RuntimeError(f"cpu leg failed: {p.stderr[-400:]}")
        r = json.loads(line[-1])
        return r["events"], r["seconds"]

    # calibration: one row at 1 spp on one thread -> intersections per second and per (row, sample)
    ev_c, dt_c = leg(1, (rows[0], rows[0] + 1), 1)
    rate1, per_row_sample = ev_c / dt_c, float(ev_c)
    leg_s = max(4.5, target_s / 6.0)      # (the calibration row runs a little faster than a sustained leg: >= 3 s in practice)
    teams = sorted({t for t in (1, 4, 16, 32, 64, 128, full_team, usable) if 1 <= t <= usable})
    sweep, secs = {}, {}
    eff = 1.0        # scaling efficiency seen so far, to size the next leg
    for t in teams:
        budget_rs = leg_s * rate1 * t * eff / per_row_sample          # (row, sample) units this leg can afford
        n_rows = int(max(1, min(rows[1] - rows[0], budget_rs)))
        spp = int(max(1, min(256, budget_rs / n_rows)))
        ev, dt = leg(t, (rows[0], rows[0] + n_rows), spp)
        sweep[t], secs[t] = ev / dt / 1e6, dt
        eff = max(0.05, min(1.0, (ev / dt) / (rate1 * t)))
    upto = [t for t in teams if t <= info["physical_cores"]]
    monotonic = all(sweep[b] >= 0.9 * sweep[a] for a, b in zip(upto, upto[1:]))
    cause = None
    if not monotonic or any(sweep[t] < 0.9 * sweep[max(upto)] for t in teams if t > info["physical_cores"]):
        cause = (f"lscpu: {info.get('sockets')} socket(s) x {info.get('cores_per_socket')} cores x {info.get('threads_per_core')} "
                 f"threads, {info.get('numa_nodes')} NUMA node(s); the process may use {usable} CPUs"
                 + (f" under a cgroup quota of {info['cgroup_quota_cpus']:.1f}" if info.get("cgroup_quota_cpus") else "")
                 + ": teams beyond the physical cores share cores (SMT) and span NUMA domains")
    return {"value": sweep[full_team], "unit": "Mray-surface-intersections/s", "cores": full_team,
            "host_cores": info["logical"], "physical_cores": info["physical_cores"], "usable_cpus": usable, "kind": "port",
            "value_t1": sweep[1], "cores_t1": 1, "pinning": "OMP_PROC_BIND=close OMP_PLACES=cores, one child process per leg",
            "thread_sweep_M_per_s": {str(k): v for k, v in sweep.items()},
            "thread_sweep_seconds": {str(k): v for k, v in secs.items()},
            "monotonic_up_to_physical_cores_within_10pct": monotonic, "stated_cause": cause, "host": info,
            "sample": f"rows {rows[0]}.. of the {W}x{H} frame at reduced spp, every path marched on its own, each leg sized to "
                      f">= {leg_s:.0f} s on its team ({full_team} threads: {secs[full_team]:.1f} s; one thread: {secs[1]:.1f} s); "
                      f"oracle/lf_geo_oracle.c; the reference has no geometric lens to time"}


def reference_flare_path(pkg, budget_s, gpu_leg=True):
    """gpu_leg=False: the CPU runs only (child processes of the reference's binary: started BEFORE this process
    touches the GPU); reference_flare_path_gpu adds the device's frame later.
    Side data (never `value`): the REAL reference's own CPU renderer (oracle/_ref/ref_dump, the
    reference hot path compiled from its own sources) per BASELINE.md section 3 -- the 1080p frame with
    final_apertures/pentbig500_14.png (340x350 bbox: a crop, the whole frame is ~4 core-hours) and
    with apertures/pentbiglines.png (80x78 bbox), on 1 thread and on all host cores (the reference's
    tile workers = independent pixel lists) -- next to the same frame through the C ABI on the GPU.
    None when the prebuilt binary is absent."""
    import subprocess
    import tempfile
    dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(dump):
        return None
    out = {"kind": "reference", "binary": "oracle/_ref/ref_dump (reference TUs, g++ -O3 -mavx2)"}
    try:
        W, H = 1920, 1080
        tmp = tempfile.mkdtemp(prefix="lfref")
        hf, vf = 50.0, 2 * math.degrees(math.atan(math.tan(math.radians(25.0)) * H / W))
        ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
        light = [(2 * SUN_NS[0] - 1) * ex * 10, (2 * SUN_NS[1] - 1) * ey * 10, -10.0, 1.0, 0.9, 0.5]
        cam = os.path.join(tmp, "cam.txt")
        sd = H / (2 * math.tan(math.radians(vf) / 2))
        with open(cam, "w") as f:   # Camera::load_settings (camera.cpp:228-242)
            f.write(f"{hf!r} {vf!r} {W / H!r} 0.01 100\n0 0 0 0 0 -1\n1.5 0.7 5 0.5 100\n")
            f.write("1 0 0 0 1 0 0 0 1\n" + f"{W} {H} {sd!r}\n4.7 0\n")
        spec = ",".join(repr(float(v)) for v in light)
        gh_png = os.path.join(pkg.DATA, "octagonbokeh.png")
        cores = min(os.cpu_count() or 1, 64)
        cx, cy = int(SUN_NS[0] * W), int(SUN_NS[1] * H)

        def run(ap_name, crop, threads):
            """crop x crop pixels around the sun, dealt to `threads` processes; -> seconds, pixels."""
            ap_png = os.path.join(pkg.DATA, ap_name)
            x0, y0 = max(0, cx - crop // 2), max(0, cy - crop // 2)
            pix = [(x, y) for y in range(y0, min(H, y0 + crop)) for x in range(x0, min(W, x0 + crop))]
            procs = []
            t0 = time.time()
            for t in range(threads):
                lst = os.path.join(tmp, f"list{t}.txt")
                with open(lst, "w") as f:
                    f.write("\n".join(f"{x} {y}" for x, y in pix[t::threads]))
                procs.append(subprocess.Popen(
                    [dump, "frame", cam, str(W), str(H), "1", "25.0", "1.0", ap_png, gh_png, spec,
                     "list:" + lst, os.path.join(tmp, f"o{t}")], stdout=subprocess.DEVNULL,
                    stderr=subprocess.DEVNULL))
            for p in procs:
                if p.wait() != 0:
                    raise RuntimeError("ref_dump failed")
            return time.time() - t0, len(pix)

        # size the crops to the budget from the reference's measured ~1.7e7 DFT terms / s / core
        # 
        cases = []
        for ap_name, bbox in (("pentbig500_14.png", 340 * 350), ("pentbiglines.png", 80 * 78)):
            per_px = bbox / 1.7e7
            c1 = int(max(8, min(256, math.sqrt(budget_s / 4 / per_px))))
            cn = int(max(8, min(1024, math.sqrt(budget_s / 4 * cores / per_px))))
            t1, n1 = run(ap_name, c1, 1)
            tn, nn = run(ap_name, cn, cores)
            cases.append({"aperture": ap_name, "dft_terms_per_pixel": bbox,
                          "t1": {"threads": 1, "pixels": n1, "seconds": t1, "terms_per_s": n1 * bbox / t1,
                                 "whole_1080p_frame_s_extrapolated": t1 * W * H / n1},
                          "tN": {"threads": cores, "pixels": nn, "seconds": tn, "terms_per_s": nn * bbox / tn,
                                 "whole_1080p_frame_s_extrapolated": tn * W * H / nn}})
        out["frames"] = cases
        out["_gpu_args"] = (W, H, hf, vf, light)
        if gpu_leg:
            reference_flare_path_gpu(pkg, out)
        return out
    except Exception as e:  # noqa: BLE001
        out["error"] = str(e)
        return out


def reference_flare_path_gpu(pkg, out):
    """the same 1080p frame (paraxial ghosts + starburst + falloff + compose) through the C ABI"""
    if not out or "_gpu_args" not in out:
        return out
    W, H, hf, vf, light = out.pop("_gpu_args")
    try:
        lf = pkg.LensFlare(0)
        lf.set_frame(W, H)
        lf.set_params(1, 25.0, 1.0)
        lf.set_aperture(pkg.APERTURE_STARBURST, pkg.load_aperture_png("pentbig500_14.png"))
        lf.set_aperture(pkg.APERTURE_GHOST, pkg.load_aperture_png("octagonbokeh.png"))
        lf.set_camera(np.eye(3), [0, 0, 0], hf, vf)
        lf.set_jitter_counter(0x1e45f1a4e)
        for _ in range(2):
            lf.synchronize()
            t0 = time.perf_counter()
            lf.find_sun_pos([light])
            lf.generate_ghost_buffer()
            lf.render_flare_layer()
            lf.synchronize()
            t_gpu = time.perf_counter() - t0
        lf.close()
        out["gpu_whole_1080p_frame_s"] = t_gpu
        out["note"] = ("the reference re-sums the aperture DFT for every pixel (pathtracer.cpp:947-974); "
                       "the device gathers from one DFT of the aperture (DESIGN.md section 4); parity of the "
                       "two is pinned by tests/test_gpu_flare_parity.py")
        return out
    except Exception as e:  # noqa: BLE001
        out["error"] = str(e)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--ref-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)   # internal: one leg of cpu_baseline, in a child process
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    W, H, spp = args.width or cfg["W"], args.height or cfg["H"], args.spp or cfg["spp"]
    if args.cpu_leg:
        spec = json.loads(args.cpu_leg)
        cpu_leg(args.config, W, H, spec["rows"], spec["spp"])
        return

    import __graft_entry__ as g
    pkg = g.load_package()          # (ctypes + numpy: nothing here touches the GPU)
    from lens_flare_amd import sharding
    if sharding.needs_self_launch(args.gpus, os.environ):
        # `python3 bench.py --gpus N` without torchrun: this process starts the N ranks as children (one per GPU,
        # the same arguments), relays rank 0's JSON line and exits with their verdict.  It has not imported
        # torch.cuda nor made any HIP call, and it never execs.
        raise SystemExit(sharding.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    # The CPU legs (the oracle's pinned child processes, the real reference's binary) run BEFORE this process makes its
    # first GPU call: fresh children of a process without HIP / ROCr / gloo threads competing with their pinned teams --
    # and never from under a profiler's preload (rocprofv3 initialises the GPU in every process it preloads into).
    cpu, ref_path = None, None
    profiled = any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rank == 0 and world == 1 and not args.no_cpu and not profiled:
        cpu = cpu_baseline(args.config, W, H, args.cpu_seconds)
        ref_path = reference_flare_path(pkg, args.ref_seconds, gpu_leg=False)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # Control plane (rendezvous, the RCCL id, barriers, the final sums): torch.distributed over gloo.
    # Data plane: the C ABI's own RCCL communicator (lf_comm_*, one all-gather of finished tile rows
    # per frame over xGMI).  LF_BENCH_GATHER=torch selects the round-1 exchange (torch.distributed
    # nccl all_gather_into_tensor on the library's buffers) instead; it is also the fallback, on every
    # rank together, should the C ABI's communicator fail to initialise.
    # LF_BENCH_REHEARSAL=1: several ranks share GPU 0 on a one-GPU box (no RCCL communicator is
    # possible there: the exchange is staged through host memory over gloo).
    # LF_BENCH_REHEARSAL=rccl: the same sharing, but the C ABI's RCCL bring-up is ATTEMPTED on the shared GPU
    # (agree -> id over gloo -> ncclCommInitRank under the deadline -> first exchange -> common verdict); RCCL
    # is expected to refuse two ranks on one device, which then exercises the real failure path (every rank
    # aborts together, the exchange falls back to the host staging) on real hardware.
    rehearsal_rccl = os.environ.get("LF_BENCH_REHEARSAL") == "rccl"
    rehearsal = os.environ.get("LF_BENCH_REHEARSAL") == "1" or rehearsal_rccl
    # LF_BENCH_SOLO_COMM=1 (tests): ONE rank goes through everything N ranks go through -- gloo control plane, the
    # C ABI's communicator (RCCL forms one of a single rank), the first exchange under its deadline, the shared cull
    # table's all-gather and the frame's -- with the test knob comm_force_exchange making the library run its collectives even
    # so.  What a one-GPU box can rehearse of `bench.py --gpus N` beyond the fallbacks of LF_BENCH_REHEARSAL.
    solo = world == 1 and os.environ.get("LF_BENCH_SOLO_COMM") == "1"
    if solo:
        pkg.test_knob_default("comm_force_exchange", 1)     # (the library's collectives run with one rank as well)
    multi = world > 1 or solo
    gather_mode = "none" if not multi else ("cabi" if rehearsal_rccl else "host" if rehearsal
                                            else os.environ.get("LF_BENCH_GATHER", "cabi"))
    local = local % max(1, torch.cuda.device_count()) if rehearsal else local
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU (this node shows {torch.cuda.device_count()}); "
                         "LF_BENCH_REHEARSAL=1 lets several ranks share GPU 0")
    torch.cuda.set_device(local)
    nccl_group = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if solo:
            os.environ.setdefault("MASTER_PORT", str(sharding.free_port()))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # (the control plane's own deadline: a rank that died takes its peers' next gloo collective down
        # with an error after 10 minutes instead of the default half hour)
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        if gather_mode == "torch":
            nccl_group = dist.new_group(backend="nccl")

    lens = pkg.load_lens_file(os.environ.get("LF_BENCH_LENS", "dgauss11.lens"))  # env: experiments only
    mask = pkg.load_aperture_png("pentbig500_14.png")
    lambda_rgb, star_scale = None, None
    if cfg["n_lambda"] == 8:
        lens, lambda_rgb, star_scale = lens_8_lambda(pkg)
    efl = pkg.paraxial_efl(lens)
    sun = sun_direction(lens, efl, W, H)
    pairs = pair_list(lens, cfg["pairs"])

    if ref_path is not None:
        reference_flare_path_gpu(pkg, ref_path)        # (the device's frame beside the reference's CPU runs above)

    cull_mode = int(os.environ.get("LF_BENCH_CULL", "2"))

    def build_context():
        """-> (context, lights): everything the frame needs set on a FRESH context (also what a bring-up whose blocking call
        never returned falls back to: the abandoned helper thread still stands in the old context, which is then left to it)"""
        lf = pkg.LensFlare(local)
        lf.set_frame(W, H)
        lf.set_params(1, 25.0, 1.0)
        lf.set_aperture(pkg.APERTURE_STARBURST, mask)       # stop mask + starburst spectrum (set-up)
        lf.set_aperture(pkg.APERTURE_GHOST, mask)
        lf.set_lens(lens)
        if lambda_rgb is not None:
            lf.set_lambda_rgb(lambda_rgb)
        lf.set_ghost_pairs(pairs, True)
        if os.environ.get("LF_BENCH_TILE_STRIDE"):      # experiments only (profiles/r04_march_variants.txt)
            lf.set_tile_stride(int(os.environ["LF_BENCH_TILE_STRIDE"]))
        if os.environ.get("LF_BENCH_SUBCELL_BITS"):
            lf.set_pupil_subcells(int(os.environ["LF_BENCH_SUBCELL_BITS"]))
        lf.set_jitter_counter(0x1e45f1a4e)
        # Path culling (lf_cull.hip): the march starts only the paths a pre-pass found able to carry light from the sun to
        # the tile through the sample's pupil cell; ghost_buffer is the full enumeration's, bit for bit.  Mode 2 = the
        # table is REBUILT AT EVERY FRAME, so that the timed frame holds the whole cost (a host that renders the same
        # sun again would take mode 1 and reuse it).  LF_BENCH_CULL=0: every sample marches every path (rounds 1-4).
        lf.set_march_culling(cull_mode)
        if cfg["spectral"]:
            lf.set_starburst_spectrum(star_scale, lambda_rgb)
        # the camera has the lens' own field of view, so that the pinhole projection of find_sun_pos and
        # the lens agree about where a direction lands on the sensor
        hf = 2 * math.degrees(math.atan(0.5 * lens["sensor_width_mm"] / efl))
        vf = 2 * math.degrees(math.atan(math.tan(math.radians(hf) / 2) * H / W))
        if cfg["scene"]:
            # C4: the scene file's own sun; camera looking at it so that it projects to SUN_NS
            camera, suns = lf.load_collada(os.path.join(pkg.DATA, cfg["scene"]))
            lights = suns[:1]
            pos = np.array(camera["pos"], float) if camera else np.zeros(3)
            if cfg.get("behind_mesh"):
                # the mesh between the camera and the sun: it fills the middle of the frame (a scene-heavy
                # frame), the sun's screen position does not depend on what lies in between
                lo, hi, _ = lf.scene_bounds()
                mid, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
                to_sun = np.array(lights[0][:3], float) - mid
                pos = mid - to_sun / np.linalg.norm(to_sun) * (1.5 * ext)
            c2w = pkg.aim_camera(pos, lights[0][:3], SUN_NS, hf, vf)
            lf.set_camera(c2w, pos, hf, vf)
            # the scene through the lens: one primary path per sensor sample, as many samples as the march
            # takes (LF_BENCH_SCENE_SPP / LF_BENCH_PINHOLE_SCENE: experiments), 1 scene unit = 1 m
            scene_spp = int(os.environ.get("LF_BENCH_SCENE_SPP", spp))
            lf.set_params(scene_spp, 25.0, 1.0)
            if not os.environ.get("LF_BENCH_PINHOLE_SCENE"):
                lf.set_lens_camera(1, 0.001, 0.0)
        else:
            lf.set_camera(np.eye(3), [0, 0, 0], hf, vf)
            ex, ey = math.tan(math.radians(hf) / 2), math.tan(math.radians(vf) / 2)
            lights = [[(2 * SUN_NS[0] - 1) * ex * 10, (2 * SUN_NS[1] - 1) * ey * 10, -10.0, 1.0, 0.9, 0.5]]
        lf.set_band(0, H)
        return lf, lights

    lf, lights = build_context()

    # The frame is dealt by BLOCKS of 64 x 64 pixels, round-robin (block b belongs to rank b % world: lf_set_block_deal,
    # round 6): the block is the path cull's, so a rank builds, audits and reads only its own rows of the cull table --
    # pre-pass, audit and march shrink with the number of ranks and nothing but finished blocks is exchanged.
    # LF_BENCH_DEAL=rows: rounds 1-5's deal by 8-row tile rows (tile row t belongs to rank t % world) with the cull table
    # shared by a second collective.  One march launch per frame covers all of this rank's share.
    deal = os.environ.get("LF_BENCH_DEAL", "blocks")
    if deal not in ("blocks", "rows"):
        raise SystemExit("LF_BENCH_DEAL: blocks or rows")
    my_trows = len(sharding.my_tile_rows(H, rank, world))
    my_blocks = len(sharding.my_blocks(W, H, rank, world))
    def set_deal():
        if deal == "blocks":
            lf.set_block_deal(rank, world)
        else:
            lf.set_row_interleave(rank, world)

    set_deal()
    frame_t, scratch, gather_note = None, {}, None
    if gather_mode == "cabi":
        # Bring-up of the C ABI's RCCL communicator such that no rank can hang behind a peer that failed
        # alone (lens_flare_amd.sharding.agree / first_exchange; tests/test_sharding_gloo.py injects the
        # failures): (1) every rank says over gloo whether it CAN (librccl loads, its context is ready)
        # BEFORE any blocking RCCL call; (2) rank 0 makes the id, gloo carries it, every rank attaches;
        # (3) the first exchange -- which also sets up the communicator's channels, tens to hundreds of
        # ms that are set-up like the aperture spectrum, not part of a frame -- is waited for with a
        # deadline.  Any verdict is taken by all ranks together; after a failed step every rank aborts
        # its communicator (lf_comm_abort: the streams drain) and all fall back to the
        # torch.distributed nccl exchange, saying so in the JSON line.
        ok, why = True, ""
        try:
            if not pkg.comm_available():
                ok, why = False, "librccl.so.1 could not be loaded"
            if os.environ.get("LF_BENCH_EXCHANGE") == "f32":
                lf.comm_set_exchange_precision(32)
        except Exception as e:  # noqa: BLE001
            ok, why = False, str(e)
        all_ok, bad = sharding.agree(dist, ok, why)
        if all_ok:
            box = [None]
            try:
                if rank == 0:
                    box = [pkg.comm_unique_id()]
            except Exception as e:  # noqa: BLE001
                why = str(e)
            dist.broadcast_object_list(box, src=0)
            ok = box[0] is not None
            all_ok, bad = sharding.agree(dist, ok, why if rank == 0 else "")
        if all_ok:
            ok, why = True, ""
            # ncclCommInitRank blocks on the host until every peer has called it: under the same deadline
            # as the first exchange (a helper thread that is abandoned, never a re-exec)
            # (on expiry the context is told that the blocked call must publish nothing when it comes back: lf_comm_poison)
            done, err = sharding.call_with_deadline(lambda: lf.comm_init_rank(world, rank, box[0]),
                                                    float(os.environ.get("LF_BENCH_COMM_TIMEOUT", "120")),
                                                    on_expire=lf.comm_poison)
            if not done:
                ok, why = False, "ncclCommInitRank did not return within the deadline (a peer never joined)"
            elif err is not None:
                ok, why = False, str(err)
            all_ok, bad = sharding.agree(dist, ok, why)
        if all_ok:
            set_deal()          # (lf_comm_init_rank deals by tile rows, the C ABI's default: the exchange's unit follows the deal)
            all_ok, bad = sharding.first_exchange(
                dist, lambda: lf.comm_gather(pkg.SAMPLE_BUFFER), lf.comm_test,
                timeout_s=float(os.environ.get("LF_BENCH_COMM_TIMEOUT", "120")), on_expire=lf.comm_poison)
        if not all_ok:
            try:
                lf.comm_abort()
            except Exception:  # noqa: BLE001
                pass
            if sharding.expired(bad):
                # some rank's blocking call never came back: a helper thread still stands in that context (poisoned: it
                # publishes nothing) -- every rank goes on with a fresh one, the old one is left to it, never freed
                lf, lights = build_context()
            if rehearsal:
                gather_mode = "host"
                gather_note = "C-ABI RCCL exchange unavailable (" + "; ".join(bad) + "): exchange staged through the host (rehearsal)"
            else:
                gather_mode = "torch"
                gather_note = "C-ABI RCCL exchange unavailable (" + "; ".join(bad) + "): torch.distributed nccl exchange"
                nccl_group = dist.new_group(backend="nccl")
            set_deal()
    # The cull pre-pass is shared between the ranks (each builds 1 / world of the table, one all-gather completes it):
    # through the C ABI's communicator where that stands, through the exchange the frame itself falls back to otherwise.
    # LF_BENCH_CULL_SHARE=0: every rank builds the whole table (A/B).
    cull_share = None
    if multi and cull_mode != 0 and deal == "blocks":
        cull_share = "not needed: the frame is dealt by cull blocks, every rank builds the rows it reads"
    elif multi and cull_mode != 0 and os.environ.get("LF_BENCH_CULL_SHARE", "1") != "0":
        if gather_mode == "cabi":
            lf.comm_share_cull(True)
            cull_share = "rccl (C ABI)"
        elif gather_mode in ("torch", "host"):
            lf.set_cull_share(rank, world)
            cull_share = "torch.distributed nccl" if gather_mode == "torch" else "host (rehearsal)"
    if gather_mode in ("torch", "host"):
        ptr, nbytes = lf.device_buffer(pkg.SAMPLE_BUFFER)
        frame_t = torch.as_tensor(DevView(ptr, nbytes // 8), device=f"cuda:{local}")

    class GroupDist:   # sharding.gather_frame only needs all_gather_into_tensor
        @staticmethod
        def all_gather_into_tensor(out, inp):
            dist.all_gather_into_tensor(out, inp, group=nccl_group)

    host_exchange = [0.0]   # seconds spent in the torch / host exchange (the C ABI's is timed on its stream)

    def one_frame():
        lf.find_sun_pos(lights)
        lf.set_sun_from_flares(0, efl, 0.05)   # the sun hand-over: the in-frame light feeds the march
        if cfg["scene"]:
            lf.render_scene_term()
        if cull_share and deal == "rows" and gather_mode in ("torch", "host"):
            # the host owns the table's exchange: this rank's slab, one in-place all-gather, take-over
            lf.cull_prepare(spp)
            tptr, tn, _ = lf.cull_table_view()
            if tn:
                t_x = time.perf_counter()
                tab = torch.as_tensor(DevViewI64(tptr, tn), device=f"cuda:{local}")
                if gather_mode == "torch":
                    sharding.complete_cull_table(tab, rank, world, GroupDist)
                else:
                    htab = tab.cpu()
                    sharding.complete_cull_table(htab, rank, world, dist)
                    tab.copy_(htab)
                torch.cuda.synchronize()
                host_exchange[0] += time.perf_counter() - t_x
            lf.cull_commit()
        lf.trace_ghosts(spp, 0x1e45f1a4e)
        lf.render_flare_layer()
        # the exchange step: every rank ends up with the whole frame -- ONE all-gather per frame
        if gather_mode == "cabi":
            # on the context's second stream: the next frame's march overlaps this frame's exchange
            # (the barrier's lf.synchronize() joins the streams); LF_BENCH_GATHER_SYNC=1: in-stream
            if os.environ.get("LF_BENCH_GATHER_SYNC") == "1":
                lf.comm_gather(pkg.SAMPLE_BUFFER)
            else:
                lf.comm_gather_async(pkg.SAMPLE_BUFFER)
        elif gather_mode == "torch":
            lf.synchronize()
            t_x = time.perf_counter()
            (sharding.gather_blocks if deal == "blocks" else sharding.gather_frame)(frame_t, W, H, rank, world, GroupDist, scratch=scratch)
            torch.cuda.synchronize()   # the next frame rewrites these rows on the library's stream
            host_exchange[0] += time.perf_counter() - t_x
        elif gather_mode == "host":
            lf.synchronize()
            t_x = time.perf_counter()
            host = frame_t.cpu()
            (sharding.gather_blocks if deal == "blocks" else sharding.gather_frame)(host, W, H, rank, world, dist)
            frame_t.copy_(host)
            torch.cuda.synchronize()
            host_exchange[0] += time.perf_counter() - t_x

    def barrier():
        lf.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if cull_share == "rccl (C ABI)":
        # The first frame whose cull table is completed by the communicator (the in-place all-gather of table slabs
        # inside lf_trace_ghosts): under the same deadline and the same joint verdict as the first exchange -- a rank
        # whose peers never arrive gives up, everybody aborts, and frame and table fall back to torch.distributed together.
        all_ok, bad = sharding.first_exchange(dist, one_frame, lf.comm_test,
                                              timeout_s=float(os.environ.get("LF_BENCH_COMM_TIMEOUT", "120")), on_expire=lf.comm_poison)
        if not all_ok:
            try:
                lf.comm_abort()
            except Exception:  # noqa: BLE001
                pass
            gather_mode = "torch"
            gather_note = "C-ABI RCCL table exchange failed (" + "; ".join(bad) + "): torch.distributed nccl exchange of frame and table"
            nccl_group = dist.new_group(backend="nccl")
            if sharding.expired(bad):      # (as above: the frame that never came back still stands in the old context)
                lf, lights = build_context()
            lf.set_row_interleave(rank, world)
            lf.set_cull_share(rank, world)
            cull_share = "torch.distributed nccl"
            ptr, nbytes = lf.device_buffer(pkg.SAMPLE_BUFFER)
            frame_t = torch.as_tensor(DevView(ptr, nbytes // 8), device=f"cuda:{local}")
    for _ in range(args.warmup):
        one_frame()
    barrier()
    lf.reset_counters()
    lf.reset_scene_counters()
    lf.timing_reset()
    lf.timing_enable(True)
    host_exchange[0] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_frame()
    barrier()
    dt = time.perf_counter() - t0
    lf.timing_enable(False)

    cnt = lf.counters()
    stats = lf.march_stats()
    audit = lf.cull_audit()                      # what the timed frames' cull tables dropped, sampled (lf_set_cull_audit)
    n_audit, audit_ms = lf.timing_get("cull_audit")
    cull_reason = lf.cull_reason()
    # other sampling specifications, timed beside the default (N = 1 only, outside the timed region).  The
    # default since round 4: 64 x 64 pupil sub-cells per stratum shared by a wave whose pixel columns are 8
    # apart (tile correlation 7.4; profiles/r04_tile_stride.json).  Rounds 1-3: 4 x 4 sub-cells over 8 x 8
    # adjacent pixels (37.7).  1 x 1 sub-cells = every pixel draws on its own (1.0).
    sampling_variants = None
    if world == 1 and not cfg["scene"] and not args.no_cpu:   # (--no-cpu = the march alone: profiler passes, A/B runs)
        sampling_variants = {}

        def timed_variant():
            one_frame()
            lf.synchronize()
            lf.reset_counters()
            t_v = time.perf_counter()
            for _ in range(2):
                one_frame()
            lf.synchronize()
            ms = (time.perf_counter() - t_v) / 2 * 1e3
            return ms, lf.march_stats()["executed_events"] / 2.0

        for name, stride, bits, corr in (("rounds_1_to_3_stride1_subcells_4x4", 1, 2, 37.7),
                                         ("independent_pixels_subcells_1x1", pkg.DEFAULT_TILE_STRIDE, 0, 1.0)):
            lf.set_tile_stride(stride)
            lf.set_pupil_subcells(bits)
            ms_v, ev_v = timed_variant()
            sampling_variants[name] = {"ms_per_step": ms_v, "executed_events_per_s": ev_v / (ms_v * 1e-3),
                                       "tile_correlation": {"value": corr, "recorded_in": "profiles/r04_tile_stride.json"}}
        lf.set_tile_stride(pkg.DEFAULT_TILE_STRIDE)
        lf.set_pupil_subcells(pkg.DEFAULT_SUBCELL_BITS)
        sampling_variants["default_stride8_subcells_64x64"] = {"ms_per_step": dt / args.steps * 1e3,
                                                               "tile_correlation": {"value": 7.4, "recorded_in": "profiles/r04_tile_stride.json"}}
        # the same frame without the cull (identical pixels): every sample marches every path through the path tree
        # (the round-4 frame), and with the cull table kept between frames
        culled_frame = None
        if cull_mode != 0:          # the default frame once more (the variants above left theirs in the buffer), as it was timed
            one_frame()
            lf.synchronize()
            culled_frame = lf.read_buffer(pkg.GHOST_BUFFER)
        for name, mode in (("full_enumeration_path_tree", 0), ("culled_table_reused", 1)):
            if mode != cull_mode:
                lf.set_march_culling(mode)
                ms_v, ev_v = timed_variant()
                sampling_variants[name] = {"ms_per_step": ms_v, "executed_events_per_s": ev_v / (ms_v * 1e-3),
                                           "executed_events_per_frame": ev_v}
                if mode == 0 and culled_frame is not None:
                    # THIS run's comparison: the culled frame against the full enumeration of the same samples
                    full_frame = lf.read_buffer(pkg.GHOST_BUFFER)
                    sampling_variants[name]["culled_frame_values_differing"] = int((full_frame != culled_frame).sum())
                    sampling_variants[name]["lit_values"] = int((full_frame > 0).sum())
                    del full_frame
        del culled_frame
        lf.set_march_culling(cull_mode)
        if cull_mode != 0 and lf.cull_info()["culled"]:
            # SURVEY 8d's unit event (intersect + refract / reflect + Fresnel) on EVERY executed event: k_march_cull<K, true>,
            # the same pixels (tests/test_gpu_cull.py::test_weight_on_every_event_is_the_same_frame), timed by this run
            lf.test_knob("cull_weights_first", 1)
            try:
                ms_v, ev_v = timed_variant()
            finally:
                lf.test_knob("cull_weights_first", 0)
            sampling_variants["every_event_weighted"] = {"ms_per_step": ms_v, "executed_events_per_s": ev_v / (ms_v * 1e-3),
                                                         "executed_events_per_frame": ev_v}
        sampling_variants["note"] = ("tile_correlation = 64 Var(mean of 8 x 8 adjacent pixels) / mean pixel variance on this frame "
                                     "(1 = independent pixels, 64 = the block moves as one); every variant runs under the frame's "
                                     "culling mode except the two named after theirs")
    n_launch, march_ms = lf.timing_get("march")
    n_cull, cull_ms = lf.timing_get("cull_prepass")
    cull_info = lf.cull_info()
    cull_frac = lf.cull_started_fraction() if cull_info["culled"] else None
    n_xchg, xchg_ms = lf.timing_get("exchange") if gather_mode == "cabi" else (args.steps, host_exchange[0] * 1e3)
    n_scene, scene_ms = lf.timing_get("scene_term")
    scene_cnt = lf.scene_counters() if cfg["scene"] else None
    lens_cam = lf.lens_camera() if cfg["scene"] else None
    rccl_nranks, rccl_rank = lf.comm_info()
    fate_keys = ("rays_launched", "rays_clipped_stop", "rays_vignetted", "rays_tir", "rays_reached_scene", "rays_hit_light")
    ev = torch.tensor([float(cnt["surface_events"]), float(cnt["rays_launched"]), float(stats["executed_events"]),
                       float(stats["remarch_lane_events"]), float(stats["remarch_rows"])] +
                      [float(cnt[k]) for k in fate_keys] + [dt], dtype=torch.float64)
    # what rank 0 needs to tell a bad scaling curve's cause from the record: every rank's own numbers
    mine = {"rank": rank, "device": local, "tile_rows": my_trows if deal == "rows" else None, "blocks": my_blocks if deal == "blocks" else None,
            "prepass_ms": cull_ms / max(n_cull, 1) if n_cull else 0.0, "audit_ms": audit_ms / max(n_audit, 1) if n_audit else 0.0,
            "march_ms": march_ms / max(n_launch, 1),
            "exchange_ms": xchg_ms / max(n_xchg, 1), "scene_ms": scene_ms / max(n_scene, 1) if n_scene else 0.0,
            "wall_ms_per_step": dt / args.steps * 1e3, "rccl_nranks": rccl_nranks, "rccl_rank": rccl_rank}
    per_rank = [mine]
    if multi:
        tot = ev.clone()
        dist.all_reduce(tot[:-1], op=dist.ReduceOp.SUM)
        mx = ev[-1:].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx[0])
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    else:
        tot = ev
    logical, rays, executed, remarch_lane, remarch_rows = (float(v) for v in tot[:5])
    fate_tot = {k: float(v) for k, v in zip(fate_keys, tot[5:11])}

    if rank == 0:
        # ---- roofline of the dominant kernel (k_march) -------------------------------------------
        # algorithmic HBM bytes per launch = framebuffer rows it writes (f64 RGB) + the aperture
        # mask + the lens / program tables it reads
        px_per_launch = min(my_trows * TILE_ROWS, H) * W if deal == "rows" else min(my_blocks * 64 * 64, W * H)
        alg_bytes = px_per_launch * 24 + mask.size * 4 + 64 * 1024
        if cull_info["culled"]:   # + the cull table, read once
            alg_bytes += cull_info["blocks_x"] * cull_info["blocks_y"] * (cull_info["cells"] + 1) * 8
        avg_ms = march_ms / max(n_launch, 1)
        hbm_achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ev_per_launch = executed / max(1, n_launch * world)
        ev_rate_gpu = ev_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        # per-event instruction counts come from the rocprofv3 --pmc passes committed under
        # profiles/ (counters cannot be read inside this process); they are quoted as THIS binary's
        # only if the profile was taken from the same sources
        pmc, pmc_note = None, "no PMC summary for this config under profiles/"
        for pmc_file in PMC_FILES:   # the newest summary taken from the shipped sources, else the newest
            if not os.path.exists(pmc_file):
                continue
            try:
                allp = json.load(open(pmc_file))
                cand = allp.get(args.config) or allp.get("c3")
                if cand is not None and (pmc is None or cand.get("source_sha") == source_sha()):
                    pmc = cand
                    pmc_note = (f"profiles/{os.path.basename(pmc_file)}[{args.config if args.config in allp else 'c3'}], "
                                f"rocprofv3 --pmc, separate passes")
                    if cand.get("source_sha") == source_sha():
                        break
            except Exception as e:  # noqa: BLE001
                pmc_note = f"unreadable PMC summary: {e}"
        roof = {"bound": "valu", "unit": "wave-instr/s", "peak": VALU_PEAK, "achieved": None, "frac": None,
                "traffic": None, "kernel": "k_march_cull" if cull_info["culled"] else "k_march", "launches": n_launch, "avg_launch_ms": avg_ms,
                "executed_events_per_s_per_gpu": ev_rate_gpu, "practical_peak": VALU_PRACTICAL,
                "pmc_source": pmc_note,
                "note": "register-resident march: compulsory HBM traffic is O(frame), the binding resource "
                        "is vector issue (then the CU's single scalar unit); peak = 1024 SIMD-32 x 2.4 GHz / "
                        "2 clk per wave64 instruction, practical_peak = independent v_fma_f32 at 8 waves/SIMD "
                        "(profiles/microbench/valu_issue.hip)",
                "hbm": {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
                        "traffic_bytes_per_launch": None}}
        alg_flop_s = (executed * FLOP_PER_EVENT + remarch_lane * FLOP_PER_FRESNEL) / dt / world
        roof["flops"] = {"achieved": alg_flop_s / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": alg_flop_s / 1e12 / FP32_PEAK_TFLOPS,
                         "algorithmic_flop_per_executed_event": FLOP_PER_EVENT,
                         "algorithmic_flop_per_fresnel_weight": FLOP_PER_FRESNEL,
                         "note": "ALGORITHMIC flop (an FMA = 2, a square root = 0) per GPU against the dense FP32 vector peak: "
                                 "below the issue fraction because 12 of an event's 27 vector instructions are single "
                                 "multiplies, adds, compares and roots (an issue slot with at most one flop), and the walk "
                                 "around the events (liveness, forks, tallies) has none"}
        if pmc:
            same = pmc.get("source_sha") == source_sha()
            roof["pmc_matches_shipped_sources"] = same
            # (a record of a configuration whose launch took the other kernel holds no counters of this one: priced as absent)
            vi = pmc.get("valu_wave_instr_per_executed_event")
            si = pmc.get("scalar_instr_per_executed_event")
        if pmc and vi and si:
            roof["achieved"] = ev_rate_gpu * vi
            roof["frac"] = roof["achieved"] / VALU_PEAK
            roof["frac_of_practical"] = roof["achieved"] / VALU_PRACTICAL
            trans = (pmc.get("per_launch") or {}).get("SQ_INSTS_VALU_TRANS_F32")
            valu = pmc.get("valu_wave_instr_per_launch")
            if trans and valu:
                # the rate the vector pipe sustains for THIS instruction mix (a fraction f of v_sqrt_f32 / v_rcp_f32)
                f = trans / valu
                mix_peak = 1024.0 / (((1.0 - f) * T_VALU_NS + f * T_TRANS_NS) * 1e-9)
                roof["mix"] = {"transcendental_fraction": f, "peak": mix_peak, "frac": roof["achieved"] / mix_peak,
                               "unit": "wave-instr/s",
                               "note": "peak = 1024 SIMDs / ((1 - f) x 0.966 ns + f x 5.4 ns): what the pipe sustains for a "
                                       "dependent chain with this share of quarter-rate roots at 8 waves per SIMD, measured "
                                       "(profiles/microbench/valu_issue.hip, result_3); the march runs 6 waves per SIMD"}
            roof["valu_wave_instr_per_executed_event"] = vi
            roof["scalar"] = {"achieved": ev_rate_gpu * si, "peak": SCALAR_PEAK, "unit": "instr/s",
                              "frac": ev_rate_gpu * si / SCALAR_PEAK,
                              "note": "SALU + branch + scalar-memory instructions on the one scalar unit per CU"}
            if world == 1 and args.config in (pmc.get("config"), "c3"):
                roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                roof["hbm"]["traffic_bytes_per_launch"] = pmc.get("hbm_bytes_per_launch")
        out = {
            "metric": "Mray-surface-intersections/s + frame time, 1080p 256spp double-Gauss",
            "value": executed / dt / 1e6,
            "unit": "Mray-surface-intersections/s",
            # every path's intersections counted on their own (what a per-path tracer executes):
            # the march computes the legs shared by the paths of one sample once
            "value_logical": logical / dt / 1e6,
            "shared_leg_saving": logical / executed if executed else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['text']}" +
                                   (f" [overridden: {W}x{H}, {spp} spp]" if (args.width or args.height or args.spp) else ""),
                       "parallelism": f"{world} GPU(s), " + ("64 x 64-pixel blocks (the cull table's) dealt round-robin" if deal == "blocks"
                                                                  else "8-row sensor tile rows dealt round-robin")
                                      + ({"cabi": ", one ncclAllGather per frame inside the C ABI (lf_comm_gather_async: overlapped with the next frame's march)",
                                          "torch": ", one torch.distributed nccl all_gather per frame",
                                          "host": ", REHEARSAL: ranks share one GPU, exchange staged through host memory",
                                          "none": ""}[gather_mode]),
                       "deal": deal, "gather_mode": gather_mode, "gather_note": gather_note,
                       "exchange_dtype": ("f32" if os.environ.get("LF_BENCH_EXCHANGE") == "f32" else "f64") if world > 1 else None,
                       "rays_per_frame": rays / args.steps,
                       "events_executed_per_frame": executed / args.steps,
                       "events_logical_per_frame": logical / args.steps,
                       "focal_length_mm": efl},
            # ---- what the counted events are, and what became of the rays ---------------------------
            # A counted event = intersection + refraction / reflection, executed by the first pass.  Its
            # Fresnel / aperture WEIGHT is computed by marching a finished path a second time, only where a
            # lane reached the light's lobe: `remarch_events` (counted, NEVER added to `value`) is how many
            # events that second march evaluated for the lanes that needed them, and
            # fresnel_evaluated_fraction = that / executed.  profiles/r04_all_weights_ablation.json prices
            # the event of SURVEY 8d (weight on every event) against this: `every_event_weighted` quotes that
            # RECORDED measurement (c3 frame, a timing-only build) -- it is not measured by this run.
            "event_accounting": {
                "every_event_weighted": every_event_weighted(),
                # ... and since round 5 the run times it itself, on the shipped library (sampling_variants.every_event_weighted):
                "every_event_weighted_this_run": (sampling_variants or {}).get("every_event_weighted"),
                "fresnel_evaluated_fraction": remarch_lane / executed if executed else None,
                "remarch_events": remarch_lane / args.steps,
                "remarch_rows_x64": 64.0 * remarch_rows / args.steps,
                "remarch_rows_over_executed": 64.0 * remarch_rows / executed if executed else None,
                "note": "remarch_rows_x64 = what the SIMDs executed for the second marches (whole waves); per frame"},
            # ray = one (sample, wavelength, path); fractions of rays_launched, summed over ranks
            "fates": {k[5:]: (fate_tot[k] / fate_tot["rays_launched"] if fate_tot["rays_launched"] else None)
                      for k in fate_keys[1:]},
            "events_per_ray": {"executed": executed / rays if rays else None, "logical": logical / rays if rays else None},
            # ---- where the rays go: what the frame spends per unit of light ------------------------
            # culling: the pre-pass (k_cull_level, coarse to fine over the pupil square) and what its table starts;
            # with mode 2 its time is inside ms_per_step at every frame.  light: rays that leave the front element / end
            # inside the sun's lobe per second of the WHOLE frame.  equal_variance: the culled frame's pixels are the
            # round-4 default's bit for bit (same estimator, same samples), so its variance ratio is exactly 1 and
            # the frame time at equal variance is the frame time.
            "culling": {"mode": cull_mode, "culled": cull_info["culled"], "reason": cull_reason,
                        "prepass_shared_between_ranks": cull_share,
                        "prepass_ms_per_frame": cull_ms / args.steps, "prepass_builds": n_cull,
                        "started_fraction": cull_frac,
                        "table": {k: cull_info[k] for k in ("blocks_x", "blocks_y", "cells", "G", "P", "block_px")},
                        # the audit of every table the timed frames built: rays of the boxes the table does NOT start,
                        # marched with the march's events; one that reaches the light refutes the table (the launch then
                        # marches everything).  Inside ms_per_step.
                        "audit": {"rays_per_frame": audit["rays"] / args.steps, "lit": audit["lit"],
                                  "launches_refuted": audit["launches_refuted"], "ms_per_frame": audit_ms / args.steps,
                                  "launches": n_audit},
                        "note": "started_fraction = of all (block, pupil cell, path) combinations, the part the march starts; the "
                                "pre-pass bounds are second-order estimates (DESIGN.md section 5), compared with the full enumeration "
                                "in tests/test_gpu_cull.py and, for this run's frame, under equal_variance below"},
            "light": {"rays_reaching_scene_per_s": fate_tot["rays_reached_scene"] / dt,
                      "rays_hitting_light_per_s": fate_tot["rays_hit_light"] / dt,
                      "reached_scene_fraction_of_started": fate_tot["rays_reached_scene"] / fate_tot["rays_launched"] if fate_tot["rays_launched"] else None,
                      "hit_light_fraction_of_started": fate_tot["rays_hit_light"] / fate_tot["rays_launched"] if fate_tot["rays_launched"] else None},
            "equal_variance": None if args.config != "c3" or world != 1 else {
                # 1.0 only where THIS run compared the culled frame with the full enumeration of the same samples and found
                # no value differing (sampling_variants.full_enumeration_path_tree); None where it did not compare (--no-cpu)
                "variance_ratio_vs_r04_default_at_equal_spp": 1.0 if (sampling_variants or {}).get("full_enumeration_path_tree", {}).get("culled_frame_values_differing") == 0 else None,
                "culled_frame_values_differing_this_run": (sampling_variants or {}).get("full_enumeration_path_tree", {}).get("culled_frame_values_differing"),
                "ms_per_frame_equal_variance": dt / args.steps * 1e3,
                "r04_ms_per_frame": R04_C3_MS, "ratio_to_r04": dt / args.steps * 1e3 / R04_C3_MS,
                "basis": "the culled frame's pixels against the full enumeration's, compared by this run (same samples, same arithmetic)",
                # the events the SAME frame costs when every sample marches every path (this run's path-tree leg), per
                # second of the shipped frame: the rate at which the full enumeration's work is disposed of -- for
                # comparison with round 4's `value` only, it is NOT `value` (which counts executed events)
                "full_enumeration_events_per_frame": (sampling_variants or {}).get("full_enumeration_path_tree", {}).get("executed_events_per_frame"),
                "full_enumeration_events_disposed_per_s": ((sampling_variants or {}).get("full_enumeration_path_tree", {}).get("executed_events_per_frame") or 0.0)
                                                          / (dt / args.steps) or None},
            # `value` is measured with the default sampling specification (4x4 pupil sub-cells per tile and
            # sample: pixels, counters and goldens of rounds 1-2 hold).  More coherence is faster at the
            # same per-pixel variance but correlates the noise inside an 8x8 tile further, none is slower
            # with independent pixels: profiles/r03_sampling_efficiency.json has the quality side.
            "sampling_variants": sampling_variants,
            # ---- multi-GPU diagnostics (every rank's own numbers; ms per frame) -----------------------
            "rccl_nranks": max(r["rccl_nranks"] for r in per_rank),
            "march_ms": {"min": min(r["march_ms"] for r in per_rank), "max": max(r["march_ms"] for r in per_rank)},
            "exchange_ms": {"min": min(r["exchange_ms"] for r in per_rank), "max": max(r["exchange_ms"] for r in per_rank),
                            "timed": ("HIP events around pack -> ncclAllGather -> unpack on the stream the exchange runs on "
                                      "(overlapped with the next frame's march)" if gather_mode == "cabi" else
                                      "host clock around the exchange incl. its synchronisation" if world > 1 else "no exchange")},
            "per_rank": per_rank,
            # ---- the scene term (configs with a scene): the sample loop of raytrace_pixel on the device -----
            # lens_camera.mode != 0: every sample's primary path is marched through the prescription and the
            # scene's radiance comes back along the exit ray (lf_set_lens_camera); counters over the timed
            # frames: rays handed to the BVH (camera + shadow), primitive tests, lens samples / that left the lens
            "scene_term": None if not cfg["scene"] else {
                "ms_per_frame": max(r["scene_ms"] for r in per_rank),
                "march_ms_per_frame": max(r["march_ms"] for r in per_rank),
                "samples_per_pixel": int(os.environ.get("LF_BENCH_SCENE_SPP", spp)),
                "lens_camera": lens_cam,
                "counters_rank0": {k: v / args.steps for k, v in scene_cnt.items()},
                "bvh_rays_per_s_rank0": (scene_cnt["rays"] / args.steps) / (mine["scene_ms"] * 1e-3) if mine["scene_ms"] > 0 else None},
            "roofline": roof,
            "cpu_baseline": cpu,
            "reference_flare_path": ref_path,
        }
        print(json.dumps(out), flush=True)
    lf.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
