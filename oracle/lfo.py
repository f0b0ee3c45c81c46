"""TEST INFRASTRUCTURE: ctypes binding of the CPU parity oracle (oracle/lf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (lens-flare_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liblf_oracle.so")
MAX_SURF = 16


class ApertureStats(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("min_x", C.c_int), ("min_y", C.c_int),
                ("max_x", C.c_int), ("max_y", C.c_int), ("total_value", C.c_double)]


class ParaxialLens(C.Structure):
    _fields_ = [("n", C.c_int), ("stop", C.c_int), ("thickness", C.c_float * MAX_SURF),
                ("curvature", C.c_float * MAX_SURF), ("ior", (C.c_float * MAX_SURF) * 3),
                ("clip", C.c_double), ("recast_pos", C.c_float), ("recast_neg", C.c_float),
                ("marginal", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("ns_aa", C.c_int), ("flare_radius", C.c_double),
                ("flare_intensity", C.c_double), ("n_flares", C.c_int),
                ("flare_origin", (C.c_double * 2) * 8), ("flare_radiance", (C.c_double * 3) * 8),
                ("axis_ray", C.c_double * 2), ("angle_to_sun", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.lfo_convert_coordinate.restype = C.c_double
        _lib.lfo_convert_coordinate.argtypes = [C.c_size_t, C.c_int, C.c_int]
        _lib.lfo_random_uniform_from_raw.restype = C.c_double
        _lib.lfo_random_uniform_from_raw.argtypes = [C.c_uint32]
        _lib.lfo_tile_order.restype = C.c_size_t
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_lens():
    L = ParaxialLens()
    lib().lfo_default_lens(C.byref(L))
    return L


def aperture_from_red(red):
    red = np.ascontiguousarray(red, dtype=np.uint8)
    h, w = red.shape
    tex = np.empty((h, w), np.float32)
    st = ApertureStats()
    lib().lfo_aperture_from_red(_p(red, C.c_uint8), w, h, _p(tex, C.c_float), C.byref(st))
    return tex, st


def aperture_stats(tex):
    tex = np.ascontiguousarray(tex, dtype=np.float32)
    st = ApertureStats()
    lib().lfo_aperture_stats_from_texels(_p(tex, C.c_float), tex.shape[1], tex.shape[0], C.byref(st))
    return st


def make_frame(W, H, ns_aa=1, flare_radius=25.0, flare_intensity=1.0, flares=(), axis_ray=(0, 0),
               angle_to_sun=0.0):
    f = Frame()
    f.W, f.H, f.ns_aa = W, H, ns_aa
    f.flare_radius, f.flare_intensity = flare_radius, flare_intensity
    f.n_flares = len(flares)
    for k, (ox, oy, r, g, b) in enumerate(flares):
        f.flare_origin[k][0], f.flare_origin[k][1] = ox, oy
        f.flare_radiance[k][0], f.flare_radiance[k][1], f.flare_radiance[k][2] = r, g, b
    f.axis_ray[0], f.axis_ray[1] = axis_ray
    f.angle_to_sun = angle_to_sun
    return f


def find_sun_pos(c2w, cam_pos, hfov, vfov, lights, frame):
    c2w = np.ascontiguousarray(c2w, np.float64).reshape(9)
    cam_pos = np.ascontiguousarray(cam_pos, np.float64)
    lights = np.ascontiguousarray(lights, np.float64).reshape(-1, 6)
    lib().lfo_find_sun_pos(_p(c2w, C.c_double), _p(cam_pos, C.c_double), C.c_double(hfov),
                           C.c_double(vfov), _p(lights, C.c_double), len(lights), C.byref(frame))
    return frame


def trace(L, kind, r, theta, i, j, colour):
    out = (C.c_double * 2)()
    fn = lib().lfo_trace_ray_auto_before if kind == "before" else lib().lfo_trace_ray_auto_after
    fn(C.byref(L), C.c_float(r), C.c_float(theta), i, j, colour, out)
    return out[0], out[1]


def convert_coordinate(p, length, y):
    return lib().lfo_convert_coordinate(p, length, int(y))


def ghost_buffer(L, frame, ghost_tex):
    ghost_tex = np.ascontiguousarray(ghost_tex, np.float32)
    out = np.zeros((frame.H, frame.W, 3), np.float64)
    lib().lfo_generate_ghost_buffer(C.byref(L), C.byref(frame), _p(ghost_tex, C.c_float),
                                    ghost_tex.shape[1], ghost_tex.shape[0], _p(out, C.c_double))
    return out


def mt19937_raw(seed, skip, n):
    out = np.empty(n, np.uint32)
    lib().lfo_mt19937_raw(C.c_uint32(seed), C.c_size_t(skip), C.c_size_t(n), _p(out, C.c_uint32))
    return out


def random_uniform_from_raw(raw):
    return lib().lfo_random_uniform_from_raw(C.c_uint32(int(raw)))


def tile_order(W, H, tile=32):
    out = np.empty(W * H, np.uint32)
    n = lib().lfo_tile_order(W, H, tile, _p(out, C.c_uint32))
    assert n == W * H
    return out


def starburst_pixel(frame, ap, st, x, y):
    ap = np.ascontiguousarray(ap, np.float32)
    rgb = (C.c_double * 3)()
    un = C.c_double()
    lib().lfo_starburst_pixel(C.byref(frame), _p(ap, C.c_float), C.byref(st), C.c_size_t(x),
                              C.c_size_t(y), rgb, C.byref(un))
    return np.array(rgb[:]), un.value


def falloff_pixel(frame, x, y, raw32, radius=5.0):
    raw32 = np.ascontiguousarray(raw32, np.uint32)
    rgb = (C.c_double * 3)()
    lib().lfo_irradiance_falloff_pixel(C.byref(frame), C.c_size_t(x), C.c_size_t(y),
                                       C.c_double(radius), _p(raw32, C.c_uint32), rgb)
    return np.array(rgb[:])


def render_pixels(frame, ap, st, ghost, order, seed=5489, n_threads=1):
    ap = np.ascontiguousarray(ap, np.float32)
    order = np.ascontiguousarray(order, np.uint32)
    out = np.zeros((frame.H, frame.W, 3), np.float64)
    gp = None
    if ghost is not None:
        ghost = np.ascontiguousarray(ghost, np.float64)
        gp = _p(ghost, C.c_double)
    lib().lfo_render_pixels(C.byref(frame), _p(ap, C.c_float), C.byref(st), gp,
                            _p(order, C.c_uint32), C.c_size_t(len(order)), C.c_uint32(seed),
                            n_threads, _p(out, C.c_double))
    return out


def to_color(sample):
    sample = np.ascontiguousarray(sample, np.float64)
    n = sample.size // 3
    out = np.empty(n, np.uint32)
    lib().lfo_to_color(_p(sample, C.c_double), C.c_size_t(n), _p(out, C.c_uint32))
    return out.reshape(sample.shape[:-1])
