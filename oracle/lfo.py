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


def starburst_pixel_spectral(frame, ap, st, x, y, scale, rgb_w):
    """Row f4 (parity unpinned): per-wavelength starburst, see lf_oracle.c."""
    ap = np.ascontiguousarray(ap, np.float32)
    scale = np.ascontiguousarray(scale, np.float64).ravel()
    rgb_w = np.ascontiguousarray(rgb_w, np.float64).reshape(len(scale), 3)
    rgb = (C.c_double * 3)()
    lib().lfo_starburst_pixel_spectral(C.byref(frame), _p(ap, C.c_float), C.byref(st), C.c_size_t(x),
                                       C.c_size_t(y), len(scale), _p(scale, C.c_double),
                                       _p(rgb_w, C.c_double), rgb)
    return np.array(rgb[:])


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


# ------------------------------------------------------------------------------------------------
# geometric march oracle (oracle/lf_geo_oracle.c) -- PARITY UNPINNED, see that file's header
# ------------------------------------------------------------------------------------------------
GEO_MAX_SURF, GEO_MAX_LAMBDA = 16, 8


class GeoLens(C.Structure):
    _fields_ = [("n_surf", C.c_int), ("stop", C.c_int), ("n_lambda", C.c_int),
                ("radius", C.c_float * GEO_MAX_SURF), ("thickness", C.c_float * GEO_MAX_SURF),
                ("semi_ap", C.c_float * GEO_MAX_SURF),
                ("ior", (C.c_float * GEO_MAX_SURF) * GEO_MAX_LAMBDA), ("sensor_w_mm", C.c_float),
                ("sun_dir", C.c_float * 3), ("sun_radiance", C.c_float * 3),
                ("sun_angular_radius", C.c_float),
                ("lambda_rgb", (C.c_float * 3) * GEO_MAX_LAMBDA)]


class GeoCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_launched", "surface_events", "rays_clipped_stop",
                                          "rays_vignetted", "rays_tir", "rays_reached_scene",
                                          "rays_hit_light")]


def geo_lens(lens, sun_dir=(0, 0, -1), sun_radiance=(1, 1, 1), sun_angular_radius=0.05,
             lambda_rgb=None):
    """lens: dict as lens_flare_amd.load_lens_file returns (raw prescription)."""
    import math
    L = GeoLens()
    L.n_surf, L.stop = int(lens["n"]), int(lens["stop"])
    ior = np.asarray(lens["ior"], np.float32)
    L.n_lambda = ior.shape[0]
    for k in range(L.n_surf):
        L.radius[k] = float(lens["radius"][k])
        L.thickness[k] = float(lens["thickness"][k])
        L.semi_ap[k] = float(lens["semi_aperture"][k])
        for l in range(L.n_lambda):
            L.ior[l][k] = float(ior[l, k])
    L.sensor_w_mm = float(lens["sensor_width_mm"])
    # same normalisation as lf_set_sun: double sqrt of the float components' squares, then narrow
    d = [float(np.float32(v)) for v in sun_dir]
    n = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    for c in range(3):
        L.sun_dir[c] = float(np.float32(d[c] / n))
        L.sun_radiance[c] = float(np.float32(sun_radiance[c]))
    L.sun_angular_radius = float(np.float32(sun_angular_radius))
    for l in range(L.n_lambda):
        for c in range(3):
            if lambda_rgb is not None:
                L.lambda_rgb[l][c] = float(np.float32(lambda_rgb[l][c]))
            else:
                L.lambda_rgb[l][c] = (1.0 if l == c else 0.0) if L.n_lambda == 3 else float(
                    np.float32(1.0) / np.float32(L.n_lambda))
    return L


def set_pupil_target(radius_mm=0.0, z_mm=0.0):
    """Both tracers: the disc sensor samples aim at, as the float32 values lf_get_pupil_target reports
    (radius <= 0: the rear element's clear aperture, the default)."""
    lib().geo_set_pupil_target(C.c_float(radius_mm), C.c_float(z_mm))
    lib().g64_set_pupil_target(C.c_double(float(np.float32(radius_mm))), C.c_double(float(np.float32(z_mm))))


def set_tile_stride(stride=8):
    """Both tracers: the pixel stride in x of a wave's tile (lf_set_tile_stride; 1, 2, 4 or 8)."""
    xs = {1: 0, 2: 1, 4: 2, 8: 3}[int(stride)]
    lib().geo_set_tile_stride_log2(xs)
    lib().g64_set_tile_stride_log2(xs)


def all_pairs(lens, include_primary=True):
    n, stop = int(lens["n"]), int(lens["stop"])
    out = [(-1, -1)] if include_primary else []
    out += [(i, j) for i in range(n) for j in range(i + 1, n) if i != stop and j != stop]
    return np.array(out, np.int32)


_followed_lf = None        # the device geo_follow_device() follows: its cull table is applied to the counters
last_culled_lit = 0        # rays the device's table skips that DID reach the light in the last geo_trace (must be 0)


def _device_cull(cull, W, H, spp):
    """cull: None (count every ray), "auto" (the table of the followed device's last trace_ghosts, if it culled)
    or a table from LensFlare.cull_table().  -> contiguous uint64 array (blocks_y, blocks_x, cells + 1) or None."""
    auto = isinstance(cull, str)
    if auto:
        assert cull == "auto"
        cull = _followed_lf.cull_table() if _followed_lf is not None and hasattr(_followed_lf, "cull_table") else None
    if cull is None:
        return None
    if isinstance(cull, tuple):      # (table, block_px)
        return np.ascontiguousarray(cull[0], np.uint64), int(cull[1])
    cull = np.ascontiguousarray(cull, np.uint64)
    G = int(np.floor(np.sqrt(spp)))
    while (G + 1) * (G + 1) <= spp:
        G += 1
    while G * G > spp:
        G -= 1
    want = {((H + bp - 1) // bp, (W + bp - 1) // bp, (G * m) ** 2 + 1): bp for bp in (128, 64, 32, 16) for m in (1, 2, 4)}
    if auto and cull.shape not in want:
        return None      # the followed device's last launch was another frame (it did not render this one): count every ray
    assert cull.shape in want, f"cull table {cull.shape} is not the one of a {W}x{H} frame at {spp} spp {list(want)}: " \
                               "the device's last trace_ghosts was another launch"
    block_px = want[cull.shape]
    if auto:
        block_px = _followed_lf.cull_info()["block_px"]
    return cull, block_px


def geo_trace(lens, W, H, y0, y1, spp, key, pairs, include_primary, mask, sun_dir, sun_radiance,
              sun_angular_radius, n_threads=8, lambda_rgb=None, cull="auto"):
    """-> (ghost H x W x 3, counters).  The PIXELS are always the full enumeration's (every path of every sample
    is marched); with a cull table (default: the followed device's, geo_follow_device) the COUNTERS count only
    the rays the device starts -- so `pixels equal and counters equal` says both that the device's arithmetic is
    the oracle's and that nothing it skipped could have contributed."""
    global last_culled_lit
    L = geo_lens(lens, sun_dir, sun_radiance, sun_angular_radius, lambda_rgb)
    table = _device_cull(cull, W, H, spp)
    if pairs is None:
        pairs = all_pairs(lens, include_primary)
    else:
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        if include_primary:
            pairs = np.concatenate([np.array([[-1, -1]], np.int32), pairs])
    pairs = np.ascontiguousarray(pairs, np.int32)
    mask = np.ascontiguousarray(mask, np.float32)
    ghost = np.zeros((H, W, 3), np.float64)
    cnt = GeoCounters()
    k = (C.c_uint32 * 2)(key & 0xffffffff, (key >> 32) & 0xffffffff)
    lib().geo_culled_lit.restype = C.c_uint64
    if table is not None:
        table, block_px = table
        lib().geo_set_cull(_p(table, C.c_uint64), table.shape[1], table.shape[0], table.shape[2] - 1, block_px)
    try:
        lib().geo_trace(C.byref(L), W, H, y0, y1, spp, k, _p(pairs, C.c_int), len(pairs),
                        _p(mask, C.c_float), mask.shape[1], mask.shape[0], _p(ghost, C.c_double),
                        C.byref(cnt), n_threads)
    finally:
        lib().geo_set_cull(None, 0, 0, 0, 64)
    last_culled_lit = int(lib().geo_culled_lit())
    if last_culled_lit:
        print(f"lfo.geo_trace: {last_culled_lit} rays that the device's cull table skips reached the light")
    return ghost, {n: int(getattr(cnt, n)) for n, _ in cnt._fields_}


def geo_glass_event(p, d, w, zv, c, h2, eta, reflect, forward):
    pp = (C.c_float * 3)(*p)
    dd = (C.c_float * 3)(*d)
    ww = C.c_float(w)
    lib().geo_glass_event.restype = C.c_int
    st = lib().geo_glass_event(pp, dd, C.byref(ww), C.c_float(zv), C.c_float(c), C.c_float(h2),
                               C.c_float(eta), int(reflect), int(forward))
    return st, np.array(pp[:], np.float32), np.array(dd[:], np.float32), ww.value


def geo_trace_ray(lens, lam, i, j, p, d, w=1.0, mask=None):
    L = geo_lens(lens)
    if mask is None:
        mask = np.ones((4, 4), np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    pp = (C.c_float * 3)(*p)
    dd = (C.c_float * 3)(*d)
    ww = C.c_float(w)
    ne = C.c_int()
    st = lib().geo_trace_ray(C.byref(L), int(lam), int(i), int(j), pp, dd, C.byref(ww),
                             _p(mask, C.c_float), mask.shape[1], mask.shape[0], C.byref(ne))
    return st, np.array(pp[:], np.float64), np.array(dd[:], np.float64), ww.value, ne.value


def geo_z_sensor(lens):
    L = geo_lens(lens)
    lib().geo_z_sensor.restype = C.c_float
    return lib().geo_z_sensor(C.byref(L))


_sqrt_table_keepalive = None


def geo_set_sqrt_table(dev):
    """Install (or with None remove) the device's v_sqrt_f32 deviation table: 2^24 int8 in
    {-1, 0, 1}, index = float32 bits & 0xffffff (exponent parity + significand); see
    lf_geo_oracle.c's header.  sqrt_deviation_table() builds it from a device sqrt function."""
    global _sqrt_table_keepalive
    if dev is None:
        lib().geo_set_sqrt_table(None)
        _sqrt_table_keepalive = None
        return
    dev = np.ascontiguousarray(dev, np.int8)
    assert dev.shape == (1 << 24,)
    _sqrt_table_keepalive = dev
    lib().geo_set_sqrt_table(_p(dev, C.c_int8))


def sqrt_table_inputs():
    """One float32 per (exponent parity, significand) pattern, in [0.5, 2)."""
    return (np.arange(1 << 24, dtype=np.uint32) | np.uint32(0x3F000000)).view(np.float32)


def sqrt_deviation_table(device_sqrt):
    """device_sqrt: float32 array -> float32 array as the device computes it (lf_native_sqrt)."""
    x = sqrt_table_inputs()
    hw = np.asarray(device_sqrt(x), np.float32).view(np.int32)
    exact = np.sqrt(x).view(np.int32)           # numpy's float32 sqrt is correctly rounded
    dev = hw - exact
    assert np.abs(dev).max() <= 1, "v_sqrt_f32 is documented to be accurate to 1 ulp"
    return dev.astype(np.int8)


_rcp_table_keepalive = None


def geo_set_rcp_table(dev):
    """int8[2^23]: deviation of the device's v_rcp_f32 from the correctly rounded reciprocal, index = float32
    bits & 0x7fffff (the significand); None = the correctly rounded reciprocal."""
    global _rcp_table_keepalive
    if dev is None:
        lib().geo_set_rcp_table(None)
        _rcp_table_keepalive = None
        return
    dev = np.ascontiguousarray(dev, np.int8)
    assert dev.shape == (1 << 23,)
    _rcp_table_keepalive = dev
    lib().geo_set_rcp_table(_p(dev, C.c_int8))


def rcp_table_inputs():
    """One float32 per significand, in [1, 2)."""
    return (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)


def rcp_deviation_table(device_rcp):
    """device_rcp: float32 array -> float32 array as the device computes it (lf_native_rcp)."""
    x = rcp_table_inputs()
    hw = np.asarray(device_rcp(x), np.float32).view(np.int32)
    exact = (np.float32(1.0) / x).view(np.int32)     # IEEE float32 division: correctly rounded
    dev = hw - exact
    assert np.abs(dev).max() <= 1, "v_rcp_f32 is documented to be accurate to 1 ulp"
    return dev.astype(np.int8)


def geo_follow_device(lf):
    """Make the float32 oracle reproduce the device's two non-IEEE instructions -- v_sqrt_f32 and v_rcp_f32 --
    through their measured deviation tables (lf.native_sqrt / lf.native_rcp); None: back to the correctly
    rounded operations.  lf may also be a (sqrt table, rcp table) pair measured earlier."""
    global _followed_lf
    if lf is None:
        geo_set_sqrt_table(None)
        geo_set_rcp_table(None)
        _followed_lf = None
        return None
    _followed_lf = None if isinstance(lf, tuple) else lf     # (its cull table is read when geo_trace runs)
    tabs = lf if isinstance(lf, tuple) else (sqrt_deviation_table(lf.native_sqrt), rcp_deviation_table(lf.native_rcp))
    geo_set_sqrt_table(tabs[0])
    geo_set_rcp_table(tabs[1])
    return tabs


def geo_rcp(x):
    lib().geo_rcp_f32.restype = C.c_float
    lib().geo_rcp_f32.argtypes = [C.c_float]
    return np.array([lib().geo_rcp_f32(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def geo_sqrt(x):
    lib().geo_sqrt_f32.restype = C.c_float
    lib().geo_sqrt_f32.argtypes = [C.c_float]
    return np.array([lib().geo_sqrt_f32(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def geo_philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().geo_philox(c, k, o)
    return list(o)


# ------------------------------------------------------------------------------------------------
# scene-radiance term (oracle/lf_scene_oracle.c) -- pinned by the s*.npz golden frames
# ------------------------------------------------------------------------------------------------
def scene_term(W, H, ns_aa, c2w, pos, hfov, vfov, spheres, tris, lights, order, seed=5489,
               samples_per_batch=32, max_tol=0.05, nclip=0.01, fclip=100.0):
    """spheres: (cx,cy,cz,r,kind,a,b,c); tris: 18 numbers + (kind,a,b,c); lights: (type,x,y,z,r,g,b)."""
    mats, sp, spm, tp, tn, tm = [], [], [], [], [], []
    for s in spheres:
        sp.append(list(s[:4])); spm.append(len(mats)); mats.append([1.0 if s[4] == "e" else 0.0] + list(s[5:8]))
    for t in tris:
        tp.append(list(t[:9])); tn.append(list(t[9:18])); tm.append(len(mats))
        mats.append([1.0 if t[18] == "e" else 0.0] + list(t[19:22]))
    f64 = lambda a: np.ascontiguousarray(np.array(a, np.float64).reshape(-1))  # noqa: E731
    i32 = lambda a: np.ascontiguousarray(np.array(a, np.int32).reshape(-1))    # noqa: E731
    spa, spma, tpa, tna, tma, ma, la = f64(sp), i32(spm), f64(tp), f64(tn), i32(tm), f64(mats), f64(lights)
    c2w = f64(c2w); pos = f64(pos)
    order = np.ascontiguousarray(order, np.uint32)
    out = np.zeros((H, W, 3), np.float64)
    lib().lfo_scene_term(W, H, ns_aa, samples_per_batch, C.c_double(max_tol), _p(c2w, C.c_double),
                         _p(pos, C.c_double), C.c_double(hfov), C.c_double(vfov), C.c_double(nclip),
                         C.c_double(fclip), len(sp), _p(spa, C.c_double), _p(spma, C.c_int), len(tp),
                         _p(tpa, C.c_double), _p(tna, C.c_double), _p(tma, C.c_int), _p(ma, C.c_double),
                         len(lights), _p(la, C.c_double), _p(order, C.c_uint32), C.c_size_t(len(order)),
                         C.c_uint32(seed), _p(out, C.c_double))
    return out


def set_scene_term(scene):
    """Scene term used by render_pixels (None = nothing hit).  Keep the array alive while in use."""
    if scene is None:
        lib().lfo_set_scene_term(None)
    else:
        lib().lfo_set_scene_term(_p(scene, C.c_double))


# ------------------------------------------------------------------------------------------------
# independent float64 tracer (oracle/lf_geo_f64.c): the second opinion for the geometric march.
# Shares no code / recipe / sqrt table with lf_geo_oracle.c; parity unpinned like it (no reference
# implementation exists), anchored by tests/test_geo_f64_kat.py
# ------------------------------------------------------------------------------------------------
class G64Lens(C.Structure):
    _fields_ = [("n_surf", C.c_int), ("stop", C.c_int), ("n_lambda", C.c_int),
                ("radius", C.c_double * GEO_MAX_SURF), ("thickness", C.c_double * GEO_MAX_SURF),
                ("semi_ap", C.c_double * GEO_MAX_SURF),
                ("ior", (C.c_double * GEO_MAX_SURF) * GEO_MAX_LAMBDA), ("sensor_w_mm", C.c_double),
                ("sun_dir", C.c_double * 3), ("sun_radiance", C.c_double * 3),
                ("sun_angular_radius", C.c_double),
                ("lambda_rgb", (C.c_double * 3) * GEO_MAX_LAMBDA),
                ("eps_mm", C.c_double), ("eps_texel", C.c_double), ("eps_cos", C.c_double)]


def g64_lens(lens, sun_dir=(0, 0, -1), sun_radiance=(1, 1, 1), sun_angular_radius=0.05,
             lambda_rgb=None, eps_mm=5e-4, eps_texel=0.02, eps_cos=1e-4):
    """The prescription and the light exactly as the device receives them (float32 values), held
    in doubles.  eps_*: how close to a decision boundary a ray must pass to be called fragile --
    about 10x the float32 march's accumulated position error."""
    import math
    L = G64Lens()
    L.n_surf, L.stop = int(lens["n"]), int(lens["stop"])
    ior = np.asarray(lens["ior"], np.float32)
    L.n_lambda = ior.shape[0]
    for k in range(L.n_surf):
        L.radius[k] = float(np.float32(lens["radius"][k]))
        L.thickness[k] = float(np.float32(lens["thickness"][k]))
        L.semi_ap[k] = float(np.float32(lens["semi_aperture"][k]))
        for l in range(L.n_lambda):
            L.ior[l][k] = float(ior[l, k])
    L.sensor_w_mm = float(np.float32(lens["sensor_width_mm"]))
    d = [float(np.float32(v)) for v in sun_dir]
    n = math.sqrt(sum(v * v for v in d))
    for c in range(3):
        L.sun_dir[c] = float(np.float32(d[c] / n))        # lf_set_sun stores the unit vector as float
        L.sun_radiance[c] = float(np.float32(sun_radiance[c]))
    L.sun_angular_radius = float(np.float32(sun_angular_radius))
    for l in range(L.n_lambda):
        for c in range(3):
            if lambda_rgb is not None:
                L.lambda_rgb[l][c] = float(np.float32(lambda_rgb[l][c]))
            else:
                L.lambda_rgb[l][c] = (1.0 if l == c else 0.0) if L.n_lambda == 3 else float(
                    np.float32(1.0) / np.float32(L.n_lambda))
    L.eps_mm, L.eps_texel, L.eps_cos = eps_mm, eps_texel, eps_cos
    return L


G64_COUNTERS = ("rays_launched", "surface_events", "rays_clipped_stop", "rays_vignetted", "rays_tir",
                "rays_reached_scene", "rays_hit_light", "rays_fragile")


g64_last_causes = None
G64_CAUSES = ("aperture_rim", "mask_texel_edge", "critical_angle", "grazing_miss")


def g64_trace(lens, W, H, y0, y1, spp, key, pairs, include_primary, mask, sun_dir, sun_radiance,
              sun_angular_radius, n_threads=8, lambda_rgb=None, sub_bits=6, cull=None, causes=False, **eps):
    """-> (image, frag, counters): image / frag are H x W x 3; a faithful float32 evaluation of the
    same estimator satisfies |pixel32 - image| <= tol * image + frag (see lf_geo_f64.c).  cull: a table from
    LensFlare.cull_table() -- the image stays the full enumeration's, the counters count the rays the device starts.
    causes=True: g64_last_causes = H x W x 4, the fragile weight per pixel (summed over the channels) by cause --
    aperture rim, mask texel edge, critical angle, grazing miss of a sphere."""
    global g64_last_causes
    L = g64_lens(lens, sun_dir, sun_radiance, sun_angular_radius, lambda_rgb, **eps)
    table = _device_cull(cull, W, H, spp) if cull is not None else None
    if pairs is None:
        pairs = all_pairs(lens, include_primary)
    else:
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        if include_primary:
            pairs = np.concatenate([np.array([[-1, -1]], np.int32), pairs])
    pairs = np.ascontiguousarray(pairs, np.int32)
    mask = np.ascontiguousarray(mask, np.float32)
    image = np.zeros((H, W, 3), np.float64)
    frag = np.zeros((H, W, 3), np.float64)
    cnt = (C.c_uint64 * 8)()
    k = (C.c_uint32 * 2)(key & 0xffffffff, (key >> 32) & 0xffffffff)
    if table is not None:
        table, block_px = table
        lib().g64_set_cull(_p(table, C.c_uint64), table.shape[1], table.shape[0], table.shape[2] - 1, block_px)
    g64_last_causes = np.zeros((H, W, 4), np.float64) if causes else None
    lib().g64_set_cause_buffer(_p(g64_last_causes, C.c_double) if causes else None)
    try:
        lib().g64_trace(C.byref(L), W, H, y0, y1, spp, k, int(sub_bits), _p(pairs, C.c_int), len(pairs),
                        _p(mask, C.c_float), mask.shape[1], mask.shape[0], _p(image, C.c_double),
                        _p(frag, C.c_double), cnt, n_threads)
    finally:
        lib().g64_set_cull(None, 0, 0, 0, 64)
        lib().g64_set_cause_buffer(None)
    return image, frag, dict(zip(G64_COUNTERS, (int(v) for v in cnt)))


def g64_glass_event(lens_struct, lam, k, mirror, p, d, w=1.0):
    pp = (C.c_double * 3)(*p)
    dd = (C.c_double * 3)(*d)
    ww = C.c_double(w)
    st = lib().g64_glass_event(C.byref(lens_struct), int(lam), int(k), int(mirror), pp, dd, C.byref(ww))
    return st, np.array(pp[:]), np.array(dd[:]), ww.value


def g64_trace_ray(lens, lam, i, j, p, d, w=1.0, mask=None):
    L = g64_lens(lens)
    if mask is None:
        mask = np.ones((4, 4), np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    pp = (C.c_double * 3)(*p)
    dd = (C.c_double * 3)(*d)
    ww = C.c_double(w)
    ne = C.c_int()
    st = lib().g64_trace_ray(C.byref(L), int(lam), int(i), int(j), pp, dd, C.byref(ww),
                             _p(mask, C.c_float), mask.shape[1], mask.shape[0], C.byref(ne))
    return st, np.array(pp[:]), np.array(dd[:]), ww.value, ne.value


def g64_trace_ray_ex(lens, lam, i, j, p, d, w=1.0, mask=None, **eps):
    """-> (dead, exit point, exit direction, weight, events, fragile, potential weight)"""
    L = g64_lens(lens, **eps)
    if mask is None:
        mask = np.ones((4, 4), np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    pp = (C.c_double * 3)(*[float(v) for v in p])
    dd = (C.c_double * 3)(*[float(v) for v in d])
    ww = C.c_double(w)
    ne = C.c_int()
    out = (C.c_double * 2)()
    lib().g64_trace_ray_ex.restype = C.c_int
    st = lib().g64_trace_ray_ex(C.byref(L), int(lam), int(i), int(j), pp, dd, C.byref(ww), _p(mask, C.c_float),
                                mask.shape[1], mask.shape[0], C.byref(ne), out)
    return st, np.array(pp[:]), np.array(dd[:]), ww.value, ne.value, int(out[0]), out[1]


def g64_set_x_window(x0=0, x1=1 << 30):
    """g64_trace only traces columns [x0, x1) (default: all)."""
    lib().g64_set_x_window(int(x0), int(x1))


def g64_sensor_z(lens):
    lib().g64_sensor_z.restype = C.c_double
    return lib().g64_sensor_z(C.byref(g64_lens(lens)))


def g64_philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().g64_philox(c, k, o)
    return list(o)


# ------------------------------------------------------------------------------------------------
# the lens camera of the scene term (lf_set_lens_camera): primary paths of sensor samples by both
# tracers, the scene radiance along explicit rays, and the composition into pixels
# ------------------------------------------------------------------------------------------------
def geo_lens_rays(lens, W, lam, xy, uv, mask):
    """lf_generate_lens_rays as the float32 oracle computes it: n x 8 {origin, direction, weight, alive}."""
    L = geo_lens(lens)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    mask = np.ascontiguousarray(mask, np.float32)
    out = np.zeros((len(xy), 8), np.float32)
    lib().geo_lens_rays(C.byref(L), int(W), int(lam), len(xy), _p(xy, C.c_float), _p(uv, C.c_float),
                        _p(mask, C.c_float), mask.shape[1], mask.shape[0], _p(out, C.c_float))
    return out


def geo_lens_samples(lens, W, H, ns, key, lam, pixels, mask):
    """float32 oracle: primary path of samples 0 .. ns-1 of the listed pixels -> (n_pix, ns, 8)."""
    L = geo_lens(lens)
    pixels = np.ascontiguousarray(pixels, np.int32)
    mask = np.ascontiguousarray(mask, np.float32)
    out = np.zeros((len(pixels), ns, 8), np.float32)
    k = (C.c_uint32 * 2)(key & 0xffffffff, (key >> 32) & 0xffffffff)
    lib().geo_lens_samples(C.byref(L), int(W), int(H), int(ns), k, int(lam), _p(pixels, C.c_int), len(pixels),
                           _p(mask, C.c_float), mask.shape[1], mask.shape[0], _p(out, C.c_float))
    return out


def g64_lens_samples(lens, W, H, ns, key, lam, pixels, mask, sub_bits=6, **eps):
    """float64 tracer: (n_pix, ns, 10) {origin, unit direction, weight, potential weight, fragile, dead}."""
    L = g64_lens(lens, **eps)
    pixels = np.ascontiguousarray(pixels, np.int32)
    mask = np.ascontiguousarray(mask, np.float32)
    out = np.zeros((len(pixels), ns, 10), np.float64)
    k = (C.c_uint32 * 2)(key & 0xffffffff, (key >> 32) & 0xffffffff)
    lib().g64_lens_samples(C.byref(L), int(W), int(H), int(ns), k, int(sub_bits), int(lam), _p(pixels, C.c_int),
                           len(pixels), _p(mask, C.c_float), mask.shape[1], mask.shape[0], _p(out, C.c_double))
    return out


def lens_exposure(lens, W, mask, lam=None):
    """The lens camera's calibration (lf_lens_camera.hip calibrate_exposure): 1 / mean transmitted weight of
    the on-axis sensor point over the 64 x 64 grid of pupil-square cell centres."""
    lam = int(np.asarray(lens["ior"]).shape[0]) // 2 if lam is None else lam
    g = (np.float32(2.0) * (np.arange(64, dtype=np.float32) + np.float32(0.5))) / np.float32(64.0) - np.float32(1.0)
    uv = np.stack(np.meshgrid(g, g), axis=-1).reshape(-1, 2)   # row j, column i: (g[i], g[j])
    out = geo_lens_rays(lens, W, lam, np.zeros_like(uv), uv, mask)
    return 4096.0 / float(np.sum(out[:, 6].astype(np.float64)))


def scene_radiance_rays(spheres, tris, lights, rays):
    """est_radiance_global_illumination along explicit rays (n x 8: o, d, min_t, max_t) -> n x 3.
    Scene description as scene_term()."""
    mats, sp, spm, tp, tn, tm = [], [], [], [], [], []
    for s in spheres:
        sp.append(list(s[:4])); spm.append(len(mats)); mats.append([1.0 if s[4] == "e" else 0.0] + list(s[5:8]))
    for t in tris:
        tp.append(list(t[:9])); tn.append(list(t[9:18])); tm.append(len(mats))
        mats.append([1.0 if t[18] == "e" else 0.0] + list(t[19:22]))
    f64 = lambda a: np.ascontiguousarray(np.array(a, np.float64).reshape(-1))  # noqa: E731
    i32 = lambda a: np.ascontiguousarray(np.array(a, np.int32).reshape(-1))    # noqa: E731
    spa, spma, tpa, tna, tma, ma, la = f64(sp), i32(spm), f64(tp), f64(tn), i32(tm), f64(mats), f64(lights)
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 8)
    out = np.zeros((len(rays), 3), np.float64)
    lib().lfo_scene_radiance_rays(len(sp), _p(spa, C.c_double), _p(spma, C.c_int), len(tp), _p(tpa, C.c_double),
                                  _p(tna, C.c_double), _p(tma, C.c_int), _p(ma, C.c_double), len(lights),
                                  _p(la, C.c_double), C.c_size_t(len(rays)), _p(rays, C.c_double),
                                  _p(out, C.c_double))
    return out


def lens_exit_to_world(origin_mm, direction, c2w, pos, world_per_mm, z_ref_mm):
    """The lens camera's hand-over (lf_scene.hip scene_pixel<.., LENS>): exit state in lens space (mm) ->
    world-space ray origin / direction, the device's order of operations in float64."""
    o = np.asarray(origin_mm, np.float64)
    d = np.asarray(direction, np.float64)
    c = np.asarray(c2w, np.float64).reshape(3, 3)
    oc = np.stack([o[..., 0] * world_per_mm, o[..., 1] * world_per_mm, (o[..., 2] - z_ref_mm) * world_per_mm], -1)
    rn = 1.0 / np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
    dc = d * rn[..., None]
    rot = lambda v: np.stack([(v[..., 0] * c[k, 0] + v[..., 1] * c[k, 1]) + v[..., 2] * c[k, 2] for k in range(3)], -1)  # noqa: E731
    return np.asarray(pos, np.float64) + rot(oc), rot(dc)
