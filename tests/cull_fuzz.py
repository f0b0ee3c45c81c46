#!/usr/bin/env python3
"""Randomised search for a frame on which the culled march differs from the full enumeration.

The cull pre-pass (lens-flare_amd/csrc/lf_cull.hip) drops (sensor block, pupil cell, path) boxes on the evidence of 15
rays per box; its margins were set by comparing with the full enumeration on the frames of profiles/cull_check.py and
cull_block_size.py.  This script draws frames those scans did not: random suns over the field, lobe widths from 0.17 to
9 degrees, the shipped masks and synthetic ones (rings, slits, polygons, speckle), refocused / scaled / bent
prescriptions and the 8-wavelength one, cropped and enlarged sensors (cull blocks from 0.6 to 1.8 mm), pair subsets,
every sampling specification, sample counts that are no squares, bands and the multi-GPU row deal -- and compares
lf_trace_ghosts under lf_set_march_culling(2) with lf_set_march_culling(0) bit for bit, with the counter of rays that
reached the light.  The culled kernel is kept where the launch would fall back to the path tree (lf_test_knob
cull_force).  TEST INFRASTRUCTURE (tests/test_gpu_cull.py draws from it; profiles/ only holds what it recorded).

    python3 tests/cull_fuzz.py [cases] [seed] [families] [audit] > gpurun_out/r06_cull_fuzz_<seed>.json
    python3 tests/cull_fuzz.py reduce gpurun_out/r06_cull_fuzz_*.json > profiles/r06_cull_fuzz.json   (summaries + bad cases)

run(..., knobs=) installs pre-pass rules other than the shipped ones (lf_test_knob: the rules round 5 replaced),
audit= the rays per dropped box of the launch's audit (0: the raw table is compared -- a search for rule failures;
>= 1: what ships -- a refuted table sends the launch through the path tree, the frame must then be the enumeration's
and the record says the audit fired), n_keys= how many keys each drawn frame is rendered with.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
rng = None        # (set by run)


def synthetic_mask(kind):
    n = 256
    yy, xx = np.mgrid[0:n, 0:n]
    u, v = (xx - (n - 1) / 2) / (n / 2), (yy - (n - 1) / 2) / (n / 2)
    rr = np.hypot(u, v)
    if kind == "ring":
        a = rng.uniform(0.2, 0.6)
        return ((rr > a) & (rr < a + rng.uniform(0.1, 0.35))).astype(np.float32)
    if kind == "slit":
        th = rng.uniform(0, np.pi)
        d = np.abs(u * np.cos(th) + v * np.sin(th))
        return ((d < rng.uniform(0.02, 0.15)) & (rr < 0.9)).astype(np.float32)
    if kind == "polygon":
        k = int(rng.integers(3, 9))
        th0, r0 = rng.uniform(0, 2 * np.pi), rng.uniform(0.25, 0.9)
        m = np.ones((n, n), bool)
        for i in range(k):
            t = th0 + 2 * np.pi * i / k
            m &= (u * np.cos(t) + v * np.sin(t)) < r0 * np.cos(np.pi / k)
        cx, cy = rng.uniform(-0.2, 0.2, 2)            # off-centre: (shift by rolling)
        return np.roll(np.roll(m, int(cx * n / 2), 1), int(cy * n / 2), 0).astype(np.float32)
    if kind == "speckle":                             # a few small holes: the occupancy grid is mostly closed
        m = np.zeros((n, n), np.float32)
        for _ in range(int(rng.integers(1, 6))):
            cx, cy, r = rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(0.03, 0.12)
            m[np.hypot(u - cx, v - cy) < r] = rng.uniform(0.3, 1.0)
        return m
    raise ValueError(kind)


MASKS = ["pentbig500_14.png", "pentbiglines.png", "octagonbokeh.png", "ring", "slit", "polygon", "speckle"]
_png = {}


def draw_mask():
    k = MASKS[int(rng.integers(len(MASKS)))]
    if k.endswith(".png"):
        if k not in _png:
            _png[k] = pkg.load_aperture_png(k)
        return k, _png[k]
    return k, synthetic_mask(k)


def triplet():
    """A Cooke-type triplet, f ~ 50 mm, f/4.5 (after the classic tabulations; indices C / d / F synthesised from n_d and
    the Abbe number like data/dgauss11.lens): ANOTHER design family -- 7 interfaces, the stop behind the negative element,
    steeper relative curvatures.  The back focal distance is set by lf_focus_lens (paraxial focus at infinity)."""
    def glass(nd, V):
        return [nd - 0.3 * (nd - 1) / V, nd, nd + 0.7 * (nd - 1) / V]
    air, sk16, f2 = [1.0, 1.0, 1.0], glass(1.6204, 60.3), glass(1.6200, 36.4)
    rows = [(21.25, 4.0, sk16, 8.0), (-158.6, 5.9, air, 8.0), (-20.25, 1.5, f2, 6.0), (20.25, 2.0, air, 6.0),
            (0.0, 3.0, [0.0, 0.0, 0.0], 5.0), (141.0, 3.5, sk16, 7.0), (-17.47, 42.0, air, 7.0)]
    return {"n": len(rows), "stop": 4, "radius": np.array([r[0] for r in rows], np.float32),
            "thickness": np.array([r[1] for r in rows], np.float32),
            "ior": np.array([[r[2][l] for r in rows] for l in range(3)], np.float32),
            "semi_aperture": np.array([r[3] for r in rows], np.float32), "sensor_width_mm": 36.0}


def retrofocus():
    """A retrofocus arrangement (negative meniscus in front of a positive group, f ~ 65 mm, 9 interfaces): a THIRD family --
    strongly diverging front element, long paths between the groups.  Same index synthesis as the triplet."""
    def glass(nd, V):
        return [nd - 0.3 * (nd - 1) / V, nd, nd + 0.7 * (nd - 1) / V]
    air, bk7, sk16, f2 = [1.0, 1.0, 1.0], glass(1.5168, 64.2), glass(1.6204, 60.3), glass(1.6200, 36.4)
    rows = [(60.0, 2.0, bk7, 14.0), (18.0, 14.0, air, 11.0), (35.0, 5.0, sk16, 9.0), (-45.0, 3.0, air, 9.0),
            (0.0, 3.0, [0.0, 0.0, 0.0], 5.0), (-25.0, 1.5, f2, 6.0), (30.0, 1.0, air, 6.0), (80.0, 4.0, sk16, 7.0), (-20.0, 40.0, air, 7.0)]
    return {"n": len(rows), "stop": 4, "radius": np.array([r[0] for r in rows], np.float32),
            "thickness": np.array([r[1] for r in rows], np.float32),
            "ior": np.array([[r[2][l] for r in rows] for l in range(3)], np.float32),
            "semi_aperture": np.array([r[3] for r in rows], np.float32), "sensor_width_mm": 36.0}


HARSH = os.environ.get("FUZZ_HARSH") == "1"     # wider perturbations of the prescriptions, wider suns (the same number of draws)
FAMILIES = 4      # (run(..., families=5) adds the retrofocus; 4 keeps the stream of the recorded draws: tests ONCE_LOST)


def draw_lens():
    name = ["dgauss11.lens", "dgauss11.lens", "dgauss11_8lambda.lens", "triplet", "retrofocus"][int(rng.integers(FAMILIES))]
    lens = triplet() if name == "triplet" else retrofocus() if name == "retrofocus" else dict(pkg.load_lens_file(name))
    how = ["as_is", "as_is", "scaled", "bent", "stop_moved"][int(rng.integers(5))]
    if how == "scaled":                               # the same design at another focal length
        k = rng.uniform(0.5, 2.0) if HARSH else rng.uniform(0.7, 1.4)
        for key in ("radius", "thickness", "semi_aperture"):
            lens[key] = (np.asarray(lens[key], np.float32) * np.float32(k)).astype(np.float32)
    elif how == "bent":                               # every curvature off by up to 3 %: another aberration balance
        r = np.asarray(lens["radius"], np.float32).copy()
        b = 0.08 if HARSH else 0.03
        r *= (1.0 + rng.uniform(-b, b, r.shape)).astype(np.float32)
        lens["radius"] = r
    elif how == "stop_moved":                         # the diaphragm 1 mm towards the front or the rear group
        t = np.asarray(lens["thickness"], np.float32).copy()
        s, d = int(lens["stop"]), np.float32(rng.uniform(-2.5, 2.5) if HARSH else rng.uniform(-1.0, 1.0))
        t[s - 1] += d; t[s] -= d
        lens["thickness"] = t
    return name, how, lens


def run(N, SEED, log=sys.stderr, only=None, hook=None, families=4, knobs=None, audit=0, n_keys=1):
    """-> {"summary": ..., "cases": [...]}"""
    global rng, FAMILIES
    rng = np.random.default_rng(SEED)
    FAMILIES = families
    lf = pkg.LensFlare(0)
    lf.test_knob("cull_force", 1)
    for k, v in (knobs or {}).items():
        lf.test_knob(k, v)
    lf.set_cull_audit(audit)
    out, bad = [], 0
    t0 = time.time()
    for case in range(N):
        lname, how, lens = draw_lens()
        mname, mask = draw_mask()
        big = rng.random() < 0.08 or os.environ.get("FUZZ_BIG") == "1"      # (FUZZ_BIG=1: every frame 4K)
        W, H = [(1920, 1080), (1920, 1080), (1920, 1080), (1280, 720), (1900, 1000), (2560, 1440)][int(rng.integers(6))]
        if big:
            W, H = 3840, 2160
        # sensor width: cull blocks (64 px) between 0.6 mm and the 1.8 mm limit
        blk = rng.uniform(0.6, 1.8)
        lens["sensor_width_mm"] = np.float32(min(48.0, blk * W / 64.0))
        spp = int([4, 9, 16, 16, 36, 50, 64, 100, 200, 256][int(rng.integers(10))]) if not big else int([4, 16][int(rng.integers(2))])
        if os.environ.get("FUZZ_SPP"):                 # (every frame at one sample count: more rays per frame, the same stream otherwise)
            spp = int(os.environ["FUZZ_SPP"]) if not big else min(64, int(os.environ["FUZZ_SPP"]))
        half = 0.5 * float(lens["sensor_width_mm"]) / 50.0
        sun = [float(rng.uniform(-1.1, 1.1) * half), float(rng.uniform(-1.1, 1.1) * half * H / W), -1.0]
        alpha = float(np.exp(rng.uniform(np.log(0.003), np.log(0.3 if HARSH else 0.16))))
        stride, bits = int([1, 2, 4, 8, 8][int(rng.integers(5))]), int([0, 1, 2, 4, 6, 6, 8][int(rng.integers(7))])
        rec = {"case": case, "lens": lname, "how": how, "mask": mname, "W": W, "H": H, "spp": spp, "sun": sun, "alpha": alpha,
               "block_mm": 64.0 * float(lens["sensor_width_mm"]) / W, "stride": stride, "subcell_bits": bits}
        # every draw of the case BEFORE anything runs: a case can be replayed alone (only=[...]) from the same stream
        focus_mm = float(rng.uniform(300.0, 5000.0)) if rng.random() < 0.3 else None
        n_if, stop = int(lens["n"]), int(lens["stop"])
        sel, primary = None, True
        if rng.random() >= 0.6:
            allp = [(j, i) for i in range(n_if) for j in range(i) if i != stop and j != stop]
            k = int(rng.integers(1, len(allp)))
            sel = [allp[i] for i in rng.choice(len(allp), k, replace=False)]
            primary = bool(rng.random() < 0.7)
        band = None
        if rng.random() < 0.2:
            y0 = int(rng.integers(0, H // 2)) & ~7
            band = (y0, int(min(H, y0 + int(rng.integers(8, H)))))
        deal = None
        if rng.random() < 0.2:
            per = int(rng.integers(2, 9))
            deal = (int(rng.integers(per)), per)
        key = int(rng.integers(1, 2 ** 40))
        if focus_mm is not None:
            rec["focus_mm"] = focus_mm
        rec["pairs"] = "all" if sel is None else len(sel)
        if only is not None and case not in only:
            continue
        try:
            lf.set_frame(W, H)
            lf.set_aperture(pkg.APERTURE_STARBURST, mask)
            lf.set_lens(lens)
            if lname in ("triplet", "retrofocus"):
                lf.focus_lens(0.0)                    # the sensor at the paraxial focus of an object at infinity
            if lname.endswith("8lambda.lens"):
                lam, _ = pkg.spectral_weights(lens["lambda_nm"])
                lf.set_lambda_rgb(lam)
            if focus_mm is not None:
                lf.focus_lens(focus_mm)
            lf.set_sun(sun, [1.0, 0.9, 0.5], alpha)
            lf.set_ghost_pairs(sel, primary)
            lf.set_tile_stride(stride)
            lf.set_pupil_subcells(bits)
            lf.set_band(*(band or (0, H)))
            lf.set_row_interleave(*(deal or (0, 1)))
            if hook is not None:                       # (diagnosis: the case is set up, the caller takes over)
                hook(lf, rec, dict(spp=spp, key=key, sel=sel, primary=primary, lens=lens, mask=mask))
                out.append(rec)
                continue
            rec.update(values_differing=0, lit_values=0, lit_rays_full=0, lit_rays_culled=0, audit_refuted=0, audit_lit=0, audit_rays=0,
                       differing_unnoticed=0, differing_although_refuted=0)
            for kk in range(n_keys):
                res = {}
                for mode in (0, 2):
                    lf.set_march_culling(mode)
                    lf.reset_counters()
                    lf.trace_ghosts(spp, key + kk)
                    res[mode] = (lf.read_buffer(pkg.GHOST_BUFFER), lf.counters(), lf.cull_info(), lf.cull_audit())
                d = int((res[0][0] != res[2][0]).sum())
                info, aud = res[2][2], res[2][3]
                refuted = info["reason"] == "audit_refuted"
                lost = d != 0 or res[0][1]["rays_hit_light"] != res[2][1]["rays_hit_light"]
                rec["values_differing"] += d
                rec["lit_values"] += int((res[0][0] > 0).sum())
                rec["lit_rays_full"] += res[0][1]["rays_hit_light"]
                rec["lit_rays_culled"] += res[2][1]["rays_hit_light"]
                rec["audit_refuted"] += int(refuted)
                rec["audit_lit"] += aud["lit"]
                rec["audit_rays"] += aud["rays"]
                rec["differing_unnoticed"] += int(lost and not refuted)
                rec["differing_although_refuted"] += int(lost and refuted)
                if kk == 0:
                    rec.update(culled=info["culled"] or refuted, started=res[2][1]["rays_launched"] / max(1, res[0][1]["rays_launched"]))
            if rec["values_differing"] or rec["lit_rays_full"] != rec["lit_rays_culled"] or (audit and not knobs and rec["audit_refuted"]):
                bad += 1
                rec["BAD"] = True
                rec["key"] = key
        except pkg.LensFlareError as e:                # (a drawn prescription the library refuses: not a finding)
            rec["refused"] = str(e)[:200]
        out.append(rec)
        if log:
            print(json.dumps(rec), file=log, flush=True)
    lf.close()
    done = [r for r in out if "started" in r]
    summary = {"seed": SEED, "cases": N, "compared": len(done), "refused": len(out) - len(done),
               "culled_kernel_ran": sum(1 for r in done if r["culled"]), "frames_differing": bad,
               "audit_rays": int(sum(r["audit_rays"] for r in done)), "audit_lit": int(sum(r["audit_lit"] for r in done)),
               "launches_refuted_by_the_audit": int(sum(r["audit_refuted"] for r in done)),
               "launches_differing_unnoticed": int(sum(r["differing_unnoticed"] for r in done)),
               "launches_differing_although_refuted": int(sum(r["differing_although_refuted"] for r in done)),
               "lit_rays_compared": int(sum(r["lit_rays_full"] for r in done)),
               "frames_with_light": sum(1 for r in done if r["lit_rays_full"] > 0),
               "started_min_median_max": [float(np.min([r["started"] for r in done])), float(np.median([r["started"] for r in done])),
                                          float(np.max([r["started"] for r in done]))] if done else None,
               "seconds": time.time() - t0}
    return {"summary": summary, "cases": out}


def reduce(files):
    """the draws' summaries and every case that differed: what profiles/ keeps of a run"""
    draws, bad = [], []
    for f in files:
        d = json.load(open(f))
        draws.append(dict(d["summary"], file=os.path.basename(f), env=d.get("env", {}), argv=d.get("argv")))
        bad += [dict(c, file=os.path.basename(f)) for c in d["cases"] if c.get("BAD")]
    return {"draws": draws, "frames": sum(x["compared"] for x in draws), "frames_differing": sum(x["frames_differing"] for x in draws),
            "lit_rays_compared": sum(x["lit_rays_compared"] for x in draws), "audit_rays": sum(x.get("audit_rays", 0) for x in draws),
            "audit_lit": sum(x.get("audit_lit", 0) for x in draws), "cases_that_differed": bad}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "reduce":
        print(json.dumps(reduce(sys.argv[2:]), indent=1))
        sys.exit(0)
    r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 20261004,
            families=int(sys.argv[3]) if len(sys.argv) > 3 else 4, audit=int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    r["argv"] = sys.argv[1:]
    r["env"] = {k: v for k, v in os.environ.items() if k.startswith("FUZZ_")}
    print(json.dumps(r, indent=1))
    print("SUMMARY", json.dumps(r["summary"]), file=sys.stderr)
