"""Helpers shared by the tests: load golden fixtures (tests/golden/, generated from the real
reference by oracle/make_golden.py) and the aperture PNG assets."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FRAME_CASES = ["f64x48_pentbiglines", "f97x65_odd_rotcam", "f96x64_naive_rgba", "f80x50_two_suns",
               "f256_pentbiglines", "f1080p_pentbig500_14_scatter", "f4k_pentbiglines_scatter",
               # random flare-only frames (oracle/make_golden_fuzz.py): every branch of the starburst shaping
               "q47x31_fuzz0", "q38x52_fuzz1", "q61x33_fuzz2", "q33x33_fuzz3", "q52x40_fuzz4", "q45x29_fuzz5"]


def load_red(name):
    """Red channel of a PNG as lodepng::decode (RGBA8 conversion) yields it (camera.h:39-57)."""
    from PIL import Image
    im = Image.open(os.path.join(GOLD, "apertures", name)).convert("RGBA")
    return np.ascontiguousarray(np.asarray(im)[:, :, 0])


def load_texels(name):
    """float texels exactly as Color(const unsigned char*) makes them (CGL/src/color.cpp:16-21)."""
    return load_red(name).astype(np.float32) * np.float32(1.0 / 255.0)


def aperture_stats_golden():
    return json.load(open(os.path.join(GOLD, "apertures.json")))


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.W, self.H = self.meta["W"], self.meta["H"]
        g = np.zeros(self.W * self.H * 3)
        g[z["ghost_idx"]] = z["ghost_val"]
        self.ghost = g.reshape(self.H, self.W, 3)
        self.sample = z["sample"] if "sample" in z else None
        self.rgba = z["rgba"] if "rgba" in z else None
        self.order = z["order"] if "order" in z else None
        self.sample_at_order = z["sample_at_order"] if "sample_at_order" in z else None
        self.rgba_at_order = z["rgba_at_order"] if "rgba_at_order" in z else None

    @property
    def flares(self):
        return [tuple(f) for f in self.meta["flares"]]


def rel_err(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor if floor > 0 else np.finfo(np.float64).tiny)


def ray_budget(lf, counters, budget):
    """The march's rays_launched against the full enumeration's budget (pixels x samples x wavelengths x paths):
    equal when the launch marched every path of every sample (lf_set_march_culling(0), or a table that starts too
    much to pay); with the path cull on (the default, lens-flare_amd/csrc/lf_cull.hip) it counts the rays that were
    STARTED -- a part of the budget, never more, and never nothing on a frame with light in it."""
    n = counters["rays_launched"]
    return n == budget if not lf.cull_info()["culled"] else 0 < n <= budget
