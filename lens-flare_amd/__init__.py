"""lens_flare_amd: Python plumbing over the C ABI of liblensflare_hip.so (include/lensflare.h).

The product is the C-ABI library (hand-written gfx950 kernels); the C++ host shim that mirrors the
reference's PathTracer surface lives in host/.  This module only binds the same entry points with
ctypes so tests/ and bench.py can drive them; it contains no compute and no fallback: if the
library (or a GPU) is missing every call fails loudly.

The directory is named `lens-flare_amd` (not importable by name); load it with
`__graft_entry__.load_package()` which registers it as `lens_flare_amd`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LF_LIB: experiments only (profiles/ab_*.sh time alternative builds of the same library)
LIB_PATH = os.environ.get("LF_LIB") or os.path.join(HERE, "liblensflare_hip.so")
DATA = os.path.join(HERE, "data")

STATUS = {0: "LF_OK", 1: "LF_ERR_INVALID", 2: "LF_ERR_NO_DEVICE", 3: "LF_ERR_HIP",
          4: "LF_ERR_STATE", 5: "LF_ERR_OOM"}
APERTURE_STARBURST, APERTURE_GHOST = 0, 1
SAMPLE_BUFFER, GHOST_BUFFER, STARBURST_BUFFER, SCENE_BUFFER = 0, 1, 2, 3
# the sampling specification's defaults (lf_internal.h): 64 x 64 pupil sub-cells, wave tiles with columns 8 apart
DEFAULT_SUBCELL_BITS, DEFAULT_TILE_STRIDE = 6, 8

# every symbol include/lensflare.h declares
ABI_SYMBOLS = [
    "lf_create", "lf_destroy", "lf_last_error", "lf_abi_version", "lf_set_stream", "lf_synchronize",
    "lf_set_frame", "lf_set_band", "lf_set_row_interleave", "lf_set_block_deal", "lf_set_params", "lf_set_aperture", "lf_get_aperture_stats",
    "lf_set_paraxial_lens", "lf_set_camera", "lf_find_sun_pos", "lf_set_flares", "lf_get_flares",
    "lf_set_jitter_mt19937", "lf_set_jitter_counter", "lf_set_scene_term", "lf_set_scene",
    "lf_set_sampling", "lf_scene_bounds", "lf_set_scene_lights", "lf_set_light_samples", "lf_set_environment_map",
    "lf_set_direct_hemisphere_sample", "lf_collada_check", "lf_render_scene_term",
    "lf_generate_ghost_buffer", "lf_render_flare_layer", "lf_read_tile", "lf_read_pixel",
    "lf_write_to_framebuffer", "lf_save_image_rgba", "lf_device_buffer", "lf_set_lens", "lf_set_lambda_rgb", "lf_set_sun",
    "lf_set_sun_from_flares", "lf_paraxial_efl", "lf_paraxial_image_scale", "lf_set_ghost_pairs", "lf_set_pupil_subcells", "lf_set_tile_stride", "lf_trace_ghosts", "lf_set_march_culling", "lf_get_cull_info", "lf_get_cull_table", "lf_get_cull_started_fraction", "lf_get_cull_reason", "lf_set_cull_audit", "lf_get_cull_audit", "lf_test_knob", "lf_comm_share_cull", "lf_set_cull_share", "lf_cull_prepare", "lf_cull_table_view", "lf_cull_commit", "lf_get_march_fix_bits", "lf_generate_lens_rays", "lf_get_counters", "lf_reset_counters", "lf_get_executed_events", "lf_get_march_stats", "lf_native_sqrt", "lf_native_rcp", "lf_set_starburst_spectrum", "lf_load_collada", "lf_march_tables",
    "lf_timing_enable", "lf_timing_reset", "lf_timing_get",
    "lf_clear_ghost_buffer", "lf_draw_ghost", "lf_rasterize_textured_triangle", "lf_fill_textured_pixel",
    "lf_shift_vertex", "lf_compute_phase", "lf_irradiance_falloff", "lf_scene_trace_ray", "lf_scene_shade",
    "lf_load_lens_file", "lf_get_lens_info",
    "lf_set_pupil_target", "lf_get_pupil_target", "lf_aim_at_exit_pupil", "lf_paraxial_exit_pupil", "lf_set_ghost_accumulate",
    "lf_set_lens_camera", "lf_get_lens_camera", "lf_set_lens_camera_aim", "lf_paraxial_entrance_pupil", "lf_focus_lens", "lf_focus_lens_from_pupil",
    "lf_get_scene_counters", "lf_reset_scene_counters", "lf_set_flare_arithmetic",
    "lf_comm_get_unique_id", "lf_comm_init_rank", "lf_comm_gather", "lf_comm_gather_async", "lf_comm_wait",
    "lf_comm_destroy", "lf_comm_available", "lf_comm_info", "lf_comm_test", "lf_comm_abort",
    "lf_comm_set_exchange_precision", "lf_comm_poison", "lf_comm_is_poisoned", "lf_comm_exchange_plan",
    "lf_group_create", "lf_group_destroy", "lf_group_size", "lf_group_ctx", "lf_group_last_error",
    "lf_group_set_frame", "lf_group_for_each", "lf_group_gather", "lf_group_share_cull", "lf_group_set_block_deal",
]


def test_knob_default(name, value):
    """TEST HOOK: lf_test_knob(NULL, ...) -- the knob's value for every context created from now on (0: none)"""
    st = load_library().lf_test_knob(None, name.encode(), C.c_double(float(value)))
    if st != 0:
        raise LensFlareError(st, "lf_test_knob")


test_knob_default.__test__ = False      # (not a test, whatever pytest thinks of the name)


def paraxial_exit_pupil(lens, lam=None):
    """(z_mm, magnification) of the paraxial image of the stop through the rear group (host arithmetic)."""
    lib = load_library()
    ior = np.ascontiguousarray(lens["ior"], np.float32)
    lam = ior.shape[0] // 2 if lam is None else lam
    r = np.ascontiguousarray(lens["radius"], np.float32)
    t = np.ascontiguousarray(lens["thickness"], np.float32)
    row = np.ascontiguousarray(ior[lam], np.float32)
    z, m = C.c_double(), C.c_double()
    st = lib.lf_paraxial_exit_pupil(int(lens["n"]), int(lens["stop"]), _fp(r, C.c_float), _fp(t, C.c_float),
                                    _fp(row, C.c_float), C.byref(z), C.byref(m))
    if st != 0:
        raise LensFlareError(st, "lf_paraxial_exit_pupil")
    return z.value, m.value


def paraxial_entrance_pupil(lens, lam=None):
    """(z_mm, magnification) of the paraxial image of the stop through the front group (host arithmetic)."""
    lib = load_library()
    ior = np.ascontiguousarray(lens["ior"], np.float32)
    lam = ior.shape[0] // 2 if lam is None else lam
    r = np.ascontiguousarray(lens["radius"], np.float32)
    t = np.ascontiguousarray(lens["thickness"], np.float32)
    row = np.ascontiguousarray(ior[lam], np.float32)
    z, m = C.c_double(), C.c_double()
    st = lib.lf_paraxial_entrance_pupil(int(lens["n"]), int(lens["stop"]), _fp(r, C.c_float), _fp(t, C.c_float),
                                        _fp(row, C.c_float), C.byref(z), C.byref(m))
    if st != 0:
        raise LensFlareError(st, "lf_paraxial_entrance_pupil")
    return z.value, m.value


class LensFlareError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


class ApertureStats(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("min_x", C.c_int), ("min_y", C.c_int),
                ("max_x", C.c_int), ("max_y", C.c_int), ("total_value", C.c_double)]


class ColladaCamera(C.Structure):
    _fields_ = [("present", C.c_int), ("hfov", C.c_double), ("vfov", C.c_double), ("nclip", C.c_double),
                ("fclip", C.c_double), ("pos", C.c_double * 3), ("dir", C.c_double * 3), ("up", C.c_double * 3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_launched", "surface_events", "rays_clipped_stop",
                                          "rays_vignetted", "rays_tir", "rays_reached_scene",
                                          "rays_hit_light")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def load_library():
    """dlopen the in-tree library.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # never built lazily: a build started from a process that has touched the GPU (or from
            # N torchrun ranks at once, or under rocprofv3) is exactly what must not happen
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                                    "(make -C lens-flare_amd) before any GPU work")
        lib = C.CDLL(LIB_PATH)
        lib.lf_last_error.restype = C.c_char_p
        lib.lf_last_error.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _fp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def load_aperture_png(path):
    """An aperture PNG -> float texels exactly as CameraApertureTexture::init derives them
    (camera.h:39-63: lodepng RGBA8, red byte x float(1/255), CGL/src/color.cpp:16-21).  The decode is
    host plumbing (the ABI takes texels); names without a directory are looked up in data/."""
    from PIL import Image
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(DATA, path)
    red = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"))[:, :, 0])
    return red.astype(np.float32) * np.float32(1.0 / 255.0)


def load_lens_file(path):
    """Parse a .lens prescription (see data/dgauss11.lens) -> dict of float32 arrays."""
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(DATA, path)
    rows, sensor_w, lambda_nm = [], 36.0, None
    for line in open(path):
        line = line.split("#")[0].strip()
        if not line:
            continue
        t = line.split()
        if t[0] == "sensor_width_mm":
            sensor_w = float(t[1])
            continue
        if t[0] == "lambda_nm":      # the wavelengths of the index columns (spectral prescriptions)
            lambda_nm = [float(v) for v in t[1:]]
            continue
        rows.append([float(v) for v in t])
    rows = np.array(rows, np.float64)
    n = len(rows)
    stop = [k for k in range(n) if rows[k, 0] == 0 and rows[k, 3] == 0]
    ior = rows[:, 2:-1].T.copy()  # n_lambda x n
    if stop:
        ior[:, stop[0]] = 1.0
    lens = dict(n=n, stop=stop[0] if stop else -1, radius=rows[:, 0].astype(np.float32),
                thickness=rows[:, 1].astype(np.float32), ior=ior.astype(np.float32),
                semi_aperture=rows[:, -1].astype(np.float32), sensor_width_mm=float(sensor_w))
    if lambda_nm is not None:
        if len(lambda_nm) != ior.shape[0]:
            raise ValueError(f"{path}: lambda_nm lists {len(lambda_nm)} wavelengths for {ior.shape[0]} index columns")
        lens["lambda_nm"] = np.array(lambda_nm, np.float64)
    return lens


# Fraunhofer lines of the three index columns of the shipped prescriptions (C, d, F)
LINES_NM = (656.3, 587.6, 486.1)


def cauchy_indices(lens3, lambda_nm):
    """"indices by a 2-term Cauchy fit through each glass's three tabulated values"
    (the reference tabulates three per glass, pathtracer.cpp:553-555): n(lambda) = A + B / lambda^2, A and
    B by least squares through (C, d, F) of every interface; air and the stop stay exactly 1."""
    lam = np.asarray(lambda_nm, np.float64)
    basis = np.stack([np.ones(3), 1.0 / np.square(np.array(LINES_NM))], axis=1)   # 3 x 2
    ior3 = np.asarray(lens3["ior"], np.float64)                                 # 3 x n
    coef, *_ = np.linalg.lstsq(basis, ior3, rcond=None)                         # 2 x n
    out = coef[0][None, :] + coef[1][None, :] / np.square(lam)[:, None]
    out[:, np.all(ior3 == 1.0, axis=0)] = 1.0
    return out.astype(np.float32)


def spectral_weights(lambda_nm):
    """RGB weight of every wavelength (tents centred on the C, d, F lines, normalised so that the
    weights of a channel sum to 1) and the starburst's scale lambda_d / lambda (its pattern grows
    with the wavelength)."""
    lam = np.asarray(lambda_nm, np.float64)
    t = np.interp(-lam, [-LINES_NM[0], -LINES_NM[1], -LINES_NM[2]], [0.0, 1.0, 2.0])
    w = np.maximum(0.0, 1.0 - np.abs(t[:, None] - np.arange(3)[None, :]))
    tot = w.sum(axis=0, keepdims=True)
    w = np.where(tot > 0, w / np.where(tot > 0, tot, 1.0), 0.0)   # (a channel no wavelength reaches stays dark)
    return w.astype(np.float32), LINES_NM[1] / lam


def spectral_lens(lens3, n_lambda=8):
    """The prescription with n_lambda index columns by the Cauchy fit, wavelengths equally spaced from the
    C to the F line -> (lens, rgb weights, starburst scales).  data/dgauss11_8lambda.lens is
    spectral_lens(dgauss11.lens, 8) written out (data/make_spectral_lens.py; tests/test_spectral_lens_cpu.py)."""
    lam = np.linspace(LINES_NM[0], LINES_NM[2], n_lambda) if n_lambda > 1 else np.array([LINES_NM[1]])
    lens = dict(lens3, ior=cauchy_indices(lens3, lam), lambda_nm=lam)
    w, scale = spectral_weights(lam)
    return lens, w, scale


def paraxial_efl(lens, lam=None):
    """Paraxial focal length of a prescription dict (host arithmetic, needs no device)."""
    lib = load_library()
    ior = np.ascontiguousarray(lens["ior"], np.float32)
    lam = ior.shape[0] // 2 if lam is None else lam
    r = np.ascontiguousarray(lens["radius"], np.float32)
    t = np.ascontiguousarray(lens["thickness"], np.float32)
    row = np.ascontiguousarray(ior[lam], np.float32)
    out = C.c_double()
    st = lib.lf_paraxial_efl(int(lens["n"]), int(lens["stop"]), _fp(r, C.c_float), _fp(t, C.c_float),
                             _fp(row, C.c_float), C.byref(out))
    if st != 0:
        raise LensFlareError(st, "lf_paraxial_efl")
    return out.value


def paraxial_image_scale(lens, lam=None):
    """lf_paraxial_image_scale: chief-ray landing height per unit field angle on the sensor as the
    prescription dict places it (what lf_set_sun_from_flares(efl_mm <= 0) uses; host arithmetic)."""
    lib = load_library()
    ior = np.ascontiguousarray(lens["ior"], np.float32)
    lam = ior.shape[0] // 2 if lam is None else lam
    r = np.ascontiguousarray(lens["radius"], np.float32)
    t = np.ascontiguousarray(lens["thickness"], np.float32)
    row = np.ascontiguousarray(ior[lam], np.float32)
    out = C.c_double()
    st = lib.lf_paraxial_image_scale(int(lens["n"]), int(lens["stop"]), _fp(r, C.c_float), _fp(t, C.c_float),
                                     _fp(row, C.c_float), C.byref(out))
    if st != 0:
        raise LensFlareError(st, "lf_paraxial_image_scale")
    return out.value


def collada_check(path):
    """lf_collada_check: None if the device scene term can render the file, else the reason (no device needed)."""
    msg = C.create_string_buffer(512)
    st = load_library().lf_collada_check(os.fsencode(path), msg, C.c_size_t(512))
    return None if st == 0 else msg.value.decode()


def aim_camera(pos, world_point, ns, hfov_deg, vfov_deg):
    """Row-major c2w of a camera at `pos` under which `world_point` projects to the normalised screen
    position `ns` (Camera::analyze_world_coord, camera.cpp:245-273): any rotation that maps the
    camera-space direction of that screen point onto the direction towards the point."""
    import math
    ex, ey = math.tan(math.radians(hfov_deg) / 2), math.tan(math.radians(vfov_deg) / 2)
    d_cam = np.array([(2 * ns[0] - 1) * ex, (2 * ns[1] - 1) * ey, -1.0])
    d_cam /= np.linalg.norm(d_cam)
    fwd = np.asarray(world_point, float) - np.asarray(pos, float)
    fwd /= np.linalg.norm(fwd)

    def frame(z):
        h = np.array([0.0, 1.0, 0.0]) if abs(z[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        x = np.cross(h, z); x /= np.linalg.norm(x)
        return np.column_stack([x, np.cross(z, x), z])
    return frame(fwd) @ frame(d_cam).T


COMM_ID_BYTES = 128


def comm_available():
    """lf_comm_available: can RCCL be loaded in this process? (no blocking call, no device)"""
    return load_library().lf_comm_available() == 0


def comm_unique_id():
    """lf_comm_get_unique_id: the RCCL id rank 0 makes and the host shares (bytes)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    st = load_library().lf_comm_get_unique_id(buf)
    if st != 0:
        raise LensFlareError(st, "lf_comm_get_unique_id (is librccl.so.1 installed?)")
    return bytes(buf)


class LensFlare:
    """One context = one GPU.  Thin, checked wrappers; names follow include/lensflare.h."""

    def __init__(self, device=0, _borrowed=None):
        self.lib = load_library()
        self.lib.lf_group_ctx.restype = C.c_void_p
        self.W = self.H = 0
        self._owned = _borrowed is None
        if _borrowed is not None:
            self.ctx = C.c_void_p(_borrowed)
            return
        self.ctx = C.c_void_p()
        st = self.lib.lf_create(C.byref(self.ctx), int(device))
        if st != 0:
            self.ctx = C.c_void_p()
            raise LensFlareError(st, "lf_create failed (no gfx950 device visible?)")

    def close(self):
        if self.ctx and self._owned:
            # (LF_ERR_STATE: a communicator call this host gave up on -- comm_poison -- is still blocked in another
            # thread and stands on the context: the library leaks it on purpose rather than free it under that thread)
            self.leaked = self.lib.lf_destroy(self.ctx) != 0
        self.ctx = C.c_void_p()

    def comm_poison(self):
        """Give up on a communicator call that is blocked in another thread (sharding.call_with_deadline's on_expire):
        whatever that call still does, it publishes nothing into this context any more; every later comm_* call is
        refused; close() leaks the context while the call has not come back."""
        self.lib.lf_comm_poison(self.ctx)

    def comm_is_poisoned(self):
        return bool(self.lib.lf_comm_is_poisoned(self.ctx))

    # ---- multi-GPU, one process per GPU
    def comm_init_rank(self, nranks, rank, unique_id):
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self.lib.lf_comm_init_rank(self.ctx, int(nranks), int(rank), buf))

    def comm_gather(self, which):
        self._ck(self.lib.lf_comm_gather(self.ctx, int(which)))

    def comm_gather_async(self, which):
        """The exchange on the context's second stream: overlaps whatever is queued next."""
        self._ck(self.lib.lf_comm_gather_async(self.ctx, int(which)))

    def comm_wait(self):
        self._ck(self.lib.lf_comm_wait(self.ctx))

    def comm_info(self):
        """(nranks, rank) as RCCL reports them; (0, -1) without a communicator."""
        n, r = C.c_int(), C.c_int()
        self._ck(self.lib.lf_comm_info(self.ctx, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_test(self):
        d = C.c_int()
        self._ck(self.lib.lf_comm_test(self.ctx, C.byref(d)))
        return bool(d.value)

    def comm_abort(self):
        self._ck(self.lib.lf_comm_abort(self.ctx))

    def comm_set_exchange_precision(self, bits):
        self._ck(self.lib.lf_comm_set_exchange_precision(self.ctx, int(bits)))

    def comm_exchange_plan(self, world):
        v = (C.c_uint64 * 6)()
        self._ck(self.lib.lf_comm_exchange_plan(self.ctx, int(world), v))
        return dict(sendcount=int(v[0]), element_bytes=int(v[1]), recv_offset_bytes=int(v[2]),
                    staging_bytes=int(v[3]), groups=int(v[4]), tile_row_elements=int(v[5]))

    def comm_destroy(self):
        self._ck(self.lib.lf_comm_destroy(self.ctx))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise LensFlareError(st, self.lib.lf_last_error(self.ctx).decode())

    # ---- frame / params
    def set_stream(self, stream_ptr):
        self._ck(self.lib.lf_set_stream(self.ctx, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ck(self.lib.lf_synchronize(self.ctx))

    def set_frame(self, W, H):
        self._ck(self.lib.lf_set_frame(self.ctx, int(W), int(H)))
        self.W, self.H = int(W), int(H)

    def set_band(self, y0, y1):
        self._ck(self.lib.lf_set_band(self.ctx, int(y0), int(y1)))

    def set_row_interleave(self, phase, period):
        self._ck(self.lib.lf_set_row_interleave(self.ctx, int(phase), int(period)))

    def set_block_deal(self, rank, nranks):
        """the frame dealt by 64 x 64-pixel blocks: block b (row-major) belongs to rank b % nranks (lf_set_block_deal)"""
        self._ck(self.lib.lf_set_block_deal(self.ctx, int(rank), int(nranks)))

    def set_params(self, ns_aa=1, flare_radius=25.0, flare_intensity=1.0):
        self._ck(self.lib.lf_set_params(self.ctx, int(ns_aa), C.c_double(flare_radius),
                                        C.c_double(flare_intensity)))

    # ---- inputs
    def set_aperture(self, slot, texels):
        texels = np.ascontiguousarray(texels, np.float32)
        h, w = texels.shape
        self._ck(self.lib.lf_set_aperture(self.ctx, int(slot), _fp(texels, C.c_float), w, h))

    def aperture_stats(self, slot):
        st = ApertureStats()
        self._ck(self.lib.lf_get_aperture_stats(self.ctx, int(slot), C.byref(st)))
        return st

    def set_paraxial_lens(self, n=None, stop=None, thickness=None, curvature=None, ior_rgb=None):
        if thickness is None:
            self._ck(self.lib.lf_set_paraxial_lens(self.ctx, 0, 0, None, None, None))
            return
        th = np.ascontiguousarray(thickness, np.float32)
        cu = np.ascontiguousarray(curvature, np.float32)
        io = np.ascontiguousarray(ior_rgb, np.float32)
        self._ck(self.lib.lf_set_paraxial_lens(self.ctx, int(n), int(stop), _fp(th, C.c_float),
                                               _fp(cu, C.c_float), _fp(io, C.c_float)))

    def set_camera(self, c2w, pos, hfov_deg, vfov_deg):
        c2w = np.ascontiguousarray(c2w, np.float64).reshape(9)
        pos = np.ascontiguousarray(pos, np.float64).reshape(3)
        self._ck(self.lib.lf_set_camera(self.ctx, _fp(c2w, C.c_double), _fp(pos, C.c_double),
                                        C.c_double(hfov_deg), C.c_double(vfov_deg)))

    def find_sun_pos(self, lights):
        lights = np.ascontiguousarray(lights, np.float64).reshape(-1, 6)
        self._ck(self.lib.lf_find_sun_pos(self.ctx, _fp(lights, C.c_double), len(lights)))

    def set_flares(self, origins, radiance, axis_ray, angle_to_sun):
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 2)
        r = np.ascontiguousarray(radiance, np.float64).reshape(-1, 3)
        a = np.ascontiguousarray(axis_ray, np.float64).reshape(2)
        self._ck(self.lib.lf_set_flares(self.ctx, len(o), _fp(o, C.c_double), _fp(r, C.c_double),
                                        _fp(a, C.c_double), C.c_float(angle_to_sun)))

    def get_flares(self):
        n = C.c_int()
        o = np.zeros((8, 2), np.float64)
        r = np.zeros((8, 3), np.float64)
        a = np.zeros(2, np.float64)
        ang = C.c_float()
        self._ck(self.lib.lf_get_flares(self.ctx, C.byref(n), _fp(o, C.c_double), _fp(r, C.c_double),
                                        _fp(a, C.c_double), C.byref(ang)))
        return dict(n=n.value, origins=o[:n.value], radiance=r[:n.value], axis_ray=a,
                    angle_to_sun=ang.value)

    def set_jitter_mt19937(self, seed=5489, order=None):
        if order is None:
            self._ck(self.lib.lf_set_jitter_mt19937(self.ctx, C.c_uint32(seed), None, C.c_size_t(0)))
        else:
            order = np.ascontiguousarray(order, np.uint32)
            self._ck(self.lib.lf_set_jitter_mt19937(self.ctx, C.c_uint32(seed),
                                                    _fp(order, C.c_uint32), C.c_size_t(len(order))))

    def set_jitter_counter(self, key):
        self._ck(self.lib.lf_set_jitter_counter(self.ctx, C.c_uint64(key)))

    def set_scene_term(self, rgb):
        if rgb is None:
            self._ck(self.lib.lf_set_scene_term(self.ctx, None))
        else:
            rgb = np.ascontiguousarray(rgb, np.float64)
            assert rgb.size == self.W * self.H * 3
            self._ck(self.lib.lf_set_scene_term(self.ctx, _fp(rgb, C.c_double)))

    # ---- scene term
    def set_scene(self, spheres=(), tris=(), lights=()):
        """spheres: (cx,cy,cz,r,kind,a,b,c); tris: 18 numbers + (kind,a,b,c), kind 'd'|'e';
        lights: (type, x,y,z, r,g,b) with type 0 = directional (dirToLight), 1 = point."""
        mats, sp, spm, tp, tn, tm = [], [], [], [], [], []

        def mat(kind, a, b, c):
            # 'd' diffuse reflectance, 'e' emitted radiance, 'm' one of the reference's stub BSDFs
            # (mirror / glass / ...: f() = 0, a black occluder under its integrator)
            mats.append([1.0, a, b, c] if kind == "e" else [0.0, 0.0, 0.0, 0.0] if kind == "m" else [0.0, a, b, c])
            return len(mats) - 1

        for s in spheres:
            sp.append(list(s[:4])); spm.append(mat(*s[4:8]))
        for t in tris:
            tp.append(list(t[:9])); tn.append(list(t[9:18])); tm.append(mat(*t[18:22]))
        arr = lambda a, t: np.ascontiguousarray(np.array(a, t).reshape(-1))  # noqa: E731
        spa, spma = arr(sp, np.float64), arr(spm, np.int32)
        tpa, tna, tma = arr(tp, np.float64), arr(tn, np.float64), arr(tm, np.int32)
        ma, la = arr(mats, np.float64), arr(lights, np.float64)
        self._ck(self.lib.lf_set_scene(self.ctx, len(sp), _fp(spa, C.c_double), _fp(spma, C.c_int),
                                       len(tp), _fp(tpa, C.c_double), _fp(tna, C.c_double),
                                       _fp(tma, C.c_int), len(mats), _fp(ma, C.c_double),
                                       len(lights), _fp(la, C.c_double)))

    def set_scene_lights(self, rows):
        """rows: n x 16 {type, rgb, v0, v1, v2, v3} (include/lensflare.h, lf_set_scene_lights)."""
        r = np.ascontiguousarray(rows, np.float64).reshape(-1, 16)
        self._ck(self.lib.lf_set_scene_lights(self.ctx, len(r), _fp(r, C.c_double)))

    def set_light_samples(self, ns_area_light):
        self._ck(self.lib.lf_set_light_samples(self.ctx, int(ns_area_light)))

    def set_environment_map(self, rgb):
        """PathTracer::envLight: rgb = (h, w, 3) doubles (HDRImageBuffer::data), or None to remove it."""
        if rgb is None:
            self._ck(self.lib.lf_set_environment_map(self.ctx, 0, 0, None))
            return
        a = np.ascontiguousarray(rgb, np.float64)
        assert a.ndim == 3 and a.shape[2] == 3
        self._ck(self.lib.lf_set_environment_map(self.ctx, a.shape[1], a.shape[0], _fp(a, C.c_double)))

    def set_direct_hemisphere_sample(self, on):
        self._ck(self.lib.lf_set_direct_hemisphere_sample(self.ctx, 1 if on else 0))

    def load_collada(self, path, max_suns=8):
        """Row f3: parse a .dae, upload its static scene; returns (camera dict or None, sun lights
        as [[px, py, pz, r, g, b], ...] for find_sun_pos)."""
        cam = ColladaCamera()
        suns = np.zeros((max_suns, 6), np.float64)
        n = C.c_int(0)
        self._ck(self.lib.lf_load_collada(self.ctx, os.fsencode(path), C.byref(cam), _fp(suns, C.c_double),
                                          max_suns, C.byref(n)))
        camera = None
        if cam.present:
            camera = dict(hfov=cam.hfov, vfov=cam.vfov, nclip=cam.nclip, fclip=cam.fclip,
                          pos=list(cam.pos), dir=list(cam.dir), up=list(cam.up))
        return camera, suns[:min(n.value, max_suns)].tolist()

    def scene_bounds(self):
        lo, hi, n = (C.c_double * 3)(), (C.c_double * 3)(), C.c_int()
        self._ck(self.lib.lf_scene_bounds(self.ctx, lo, hi, C.byref(n)))
        return np.array(lo[:]), np.array(hi[:]), n.value

    def set_sampling(self, samples_per_batch=32, max_tolerance=0.05, n_clip=0.01, f_clip=100.0):
        self._ck(self.lib.lf_set_sampling(self.ctx, int(samples_per_batch), C.c_double(max_tolerance),
                                          C.c_double(n_clip), C.c_double(f_clip)))

    def render_scene_term(self):
        self._ck(self.lib.lf_render_scene_term(self.ctx))

    # ---- render
    def generate_ghost_buffer(self):
        self._ck(self.lib.lf_generate_ghost_buffer(self.ctx))

    def render_flare_layer(self):
        self._ck(self.lib.lf_render_flare_layer(self.ctx))

    # ---- read back
    def read_tile(self, which, x0, y0, x1, y1, pixel_stride=3):
        out = np.zeros((y1 - y0, x1 - x0, pixel_stride), np.float64)
        self._ck(self.lib.lf_read_tile(self.ctx, int(which), x0, y0, x1, y1, _fp(out, C.c_double),
                                       C.c_size_t(pixel_stride)))
        return out

    def read_buffer(self, which):
        return self.read_tile(which, 0, 0, self.W, self.H)

    def read_pixel(self, which, x, y):
        rgb = (C.c_double * 3)()
        self._ck(self.lib.lf_read_pixel(self.ctx, int(which), int(x), int(y), rgb))
        return np.array(rgb[:])

    def write_to_framebuffer(self, x0, y0, x1, y1):
        out = np.zeros((y1 - y0, x1 - x0), np.uint32)
        self._ck(self.lib.lf_write_to_framebuffer(self.ctx, x0, y0, x1, y1, _fp(out, C.c_uint32),
                                                  C.c_size_t(x1 - x0)))
        return out

    def save_image_rgba(self):
        out = np.zeros((self.H, self.W), np.uint32)
        self._ck(self.lib.lf_save_image_rgba(self.ctx, _fp(out, C.c_uint32)))
        return out

    def device_buffer(self, which):
        p = C.c_void_p()
        n = C.c_size_t()
        self._ck(self.lib.lf_device_buffer(self.ctx, int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- geometric lens
    def set_lens(self, lens):
        r = np.ascontiguousarray(lens["radius"], np.float32)
        t = np.ascontiguousarray(lens["thickness"], np.float32)
        i = np.ascontiguousarray(lens["ior"], np.float32)
        h = np.ascontiguousarray(lens["semi_aperture"], np.float32)
        self._ck(self.lib.lf_set_lens(self.ctx, int(lens["n"]), int(lens["stop"]), int(i.shape[0]),
                                      _fp(r, C.c_float), _fp(t, C.c_float), _fp(i, C.c_float),
                                      _fp(h, C.c_float), C.c_float(lens["sensor_width_mm"])))

    def set_lambda_rgb(self, weights):
        w = np.ascontiguousarray(weights, np.float32)
        self._ck(self.lib.lf_set_lambda_rgb(self.ctx, _fp(w, C.c_float)))

    def set_sun(self, direction, radiance, angular_radius):
        d = np.ascontiguousarray(direction, np.float32)
        r = np.ascontiguousarray(radiance, np.float32)
        self._ck(self.lib.lf_set_sun(self.ctx, _fp(d, C.c_float), _fp(r, C.c_float),
                                     C.c_float(angular_radius)))

    def set_sun_from_flares(self, flare=0, efl_mm=0.0, angular_radius=0.05):
        self._ck(self.lib.lf_set_sun_from_flares(self.ctx, int(flare), C.c_double(efl_mm),
                                                 C.c_float(angular_radius)))

    def set_ghost_pairs(self, pairs=None, include_primary=True):
        if pairs is None or len(pairs) == 0:
            self._ck(self.lib.lf_set_ghost_pairs(self.ctx, None, 0, int(include_primary)))
        else:
            p = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
            self._ck(self.lib.lf_set_ghost_pairs(self.ctx, _fp(p, C.c_int), len(p),
                                                 int(include_primary)))

    def set_pupil_subcells(self, bits):
        self._ck(self.lib.lf_set_pupil_subcells(self.ctx, int(bits)))

    def set_tile_stride(self, stride):
        self._ck(self.lib.lf_set_tile_stride(self.ctx, int(stride)))

    def trace_ghosts(self, spp, key=0x1e45f1a4e):
        self._ck(self.lib.lf_trace_ghosts(self.ctx, int(spp), C.c_uint64(key)))

    def set_march_culling(self, mode=1):
        """0 = every sample marches every path (rounds 1-4), 1 = start only the paths that can reach the light
        (default), 2 = the same with the cull table rebuilt at every launch."""
        self._ck(self.lib.lf_set_march_culling(self.ctx, int(mode)))

    def cull_info(self):
        v = (C.c_int * 8)()
        self._ck(self.lib.lf_get_cull_info(self.ctx, v))
        return dict(mode=v[0], culled=bool(v[1]), blocks_x=v[2], blocks_y=v[3], cells=v[4], G=v[5], P=v[6], block_px=v[7],
                    reason=self.cull_reason())

    CULL_REASONS = ("applied", "off", "no_stop", "too_many_paths", "too_many_samples", "block_too_large", "table_too_full",
                    "audit_refuted", "dispersion_not_monotonic")

    def cull_reason(self):
        """why the last trace_ghosts did (not) cull: one of CULL_REASONS (lf_cull_reason)"""
        r = C.c_int()
        self._ck(self.lib.lf_get_cull_reason(self.ctx, C.byref(r)))
        return self.CULL_REASONS[r.value]

    def set_cull_audit(self, rays_per_dropped_box=1):
        """rays marched per (block, cell, path) combination the cull table does not start (0 = no audit)"""
        self._ck(self.lib.lf_set_cull_audit(self.ctx, int(rays_per_dropped_box)))

    def cull_audit(self):
        """{rays, lit, launches_refuted} since reset_counters: the audit of what the cull tables dropped"""
        r, l, n = C.c_uint64(), C.c_uint64(), C.c_int()
        self._ck(self.lib.lf_get_cull_audit(self.ctx, C.byref(r), C.byref(l), C.byref(n)))
        return dict(rays=r.value, lit=l.value, launches_refuted=n.value)

    def test_knob(self, name, value):
        """TEST HOOK (lensflare.h lf_test_knob): the suite's and the bench's A/B switches, by name"""
        self._ck(self.lib.lf_test_knob(self.ctx, name.encode(), C.c_double(float(value))))

    def cull_started_fraction(self):
        f = C.c_double(0.0)
        self._ck(self.lib.lf_get_cull_started_fraction(self.ctx, C.byref(f)))
        return f.value

    # ---- the pre-pass of a multi-GPU frame, shared (include/lensflare.h)
    def comm_share_cull(self, on=True):
        self._ck(self.lib.lf_comm_share_cull(self.ctx, int(bool(on))))

    def set_cull_share(self, rank, nranks):
        self._ck(self.lib.lf_set_cull_share(self.ctx, int(rank), int(nranks)))

    def cull_prepare(self, spp):
        self._ck(self.lib.lf_cull_prepare(self.ctx, int(spp)))

    def cull_table_view(self):
        """(device pointer, entries, entries per rank) of the table the host completes; (0, 0, 0): this launch does not cull"""
        p, n, k = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._ck(self.lib.lf_cull_table_view(self.ctx, C.byref(p), C.byref(n), C.byref(k)))
        return p.value or 0, n.value, k.value

    def cull_commit(self):
        self._ck(self.lib.lf_cull_commit(self.ctx))

    def cull_table(self):
        """The path masks of the last trace_ghosts, (blocks_y, blocks_x, cells + 1) uint64; None if it did not cull."""
        i = self.cull_info()
        if not i["culled"]:
            return None
        t = np.zeros((i["blocks_y"], i["blocks_x"], i["cells"] + 1), np.uint64)
        self._ck(self.lib.lf_get_cull_table(self.ctx, _fp(t, C.c_uint64), C.c_size_t(t.size)))
        return t

    def cull_table_and_block(self):
        """(table, block side in pixels) for a checker that follows the device's table; None if not culled."""
        t = self.cull_table()
        return None if t is None else (t, self.cull_info()["block_px"])

    def march_fix_bits(self):
        b = C.c_int(0)
        self._ck(self.lib.lf_get_march_fix_bits(self.ctx, C.byref(b)))
        return b.value

    def generate_lens_rays(self, lam, sensor_xy_mm, pupil_uv):
        xy = np.ascontiguousarray(sensor_xy_mm, np.float32).reshape(-1, 2)
        uv = np.ascontiguousarray(pupil_uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(xy), 8), np.float32)
        self._ck(self.lib.lf_generate_lens_rays(self.ctx, int(lam), C.c_size_t(len(xy)),
                                                _fp(xy, C.c_float), _fp(uv, C.c_float),
                                                _fp(out, C.c_float)))
        return out

    def counters(self):
        c = Counters()
        self._ck(self.lib.lf_get_counters(self.ctx, C.byref(c)))
        return c.as_dict()

    def set_starburst_spectrum(self, scale=None, rgb_weights=None):
        """Row f4: per-wavelength starburst; None / empty = the reference's monochrome starburst."""
        if scale is None or len(scale) == 0:
            self._ck(self.lib.lf_set_starburst_spectrum(self.ctx, 0, None, None))
            return
        sc = np.ascontiguousarray(scale, np.float64).ravel()
        w = np.ascontiguousarray(rgb_weights, np.float64).reshape(len(sc), 3)
        self._ck(self.lib.lf_set_starburst_spectrum(self.ctx, len(sc), _fp(sc, C.c_double),
                                                    _fp(w, C.c_double)))

    # ---- the reference's public helper members, single-shot on the device
    def clear_ghost_buffer(self):
        self._ck(self.lib.lf_clear_ghost_buffer(self.ctx))

    def draw_ghost(self, channel, r1, r2):
        bbox = (C.c_int * 4)()
        self._ck(self.lib.lf_draw_ghost(self.ctx, int(channel), C.c_float(r1), C.c_float(r2), bbox))
        return tuple(bbox)

    def rasterize_textured_triangle(self, verts12, colour):
        v = np.ascontiguousarray(verts12, np.float32).reshape(12)
        c = np.ascontiguousarray(colour, np.float64).reshape(3)
        bbox = (C.c_int * 4)()
        self._ck(self.lib.lf_rasterize_textured_triangle(self.ctx, _fp(v, C.c_float), _fp(c, C.c_double), bbox))
        return tuple(bbox)

    def fill_textured_pixel(self, verts12, x, y, colour):
        v = np.ascontiguousarray(verts12, np.float32).reshape(12)
        c = np.ascontiguousarray(colour, np.float64).reshape(3)
        self._ck(self.lib.lf_fill_textured_pixel(self.ctx, _fp(v, C.c_float), int(x), int(y), _fp(c, C.c_double)))

    def shift_vertex(self, x, y, scale, shift_amount):
        out = (C.c_double * 2)()
        self._ck(self.lib.lf_shift_vertex(self.ctx, C.c_float(x), C.c_float(y), C.c_float(scale),
                                          C.c_float(shift_amount), out))
        return out[0], out[1]

    def compute_phase(self, flare, u, v):
        out, pos = (C.c_double * 2)(), (C.c_double * 2)()
        self._ck(self.lib.lf_compute_phase(self.ctx, int(flare), C.c_double(u), C.c_double(v), out, pos))
        return complex(out[0], out[1]), (pos[0], pos[1])

    def irradiance_falloff(self, x, y, radius):
        out = (C.c_double * 3)()
        self._ck(self.lib.lf_irradiance_falloff(self.ctx, int(x), int(y), C.c_double(radius), out))
        return np.array(out[:], np.float64)

    def scene_trace_ray(self, origin, direction, min_t=0.0, max_t=float("inf"), seq=0):
        ray = (C.c_double * 8)(*origin, *direction, min_t, max_t)
        out = (C.c_double * 8)()
        self._ck(self.lib.lf_scene_trace_ray(self.ctx, ray, C.c_uint64(seq), out))
        return dict(hit=bool(out[0]), t=out[1], n=np.array(out[2:5]), radiance=np.array(out[5:8]))

    def scene_shade(self, what, origin, direction, t, n, material, min_t=0.0, max_t=float("inf"), seq=0):
        ray = (C.c_double * 8)(*origin, *direction, min_t, max_t)
        nn, mm, out = (C.c_double * 3)(*n), (C.c_double * 4)(*material), (C.c_double * 3)()
        self._ck(self.lib.lf_scene_shade(self.ctx, int(what), ray, C.c_double(t), nn, mm, C.c_uint64(seq), out))
        return np.array(out[:], np.float64)

    def load_lens_file(self, path):
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(DATA, path)
        self._ck(self.lib.lf_load_lens_file(self.ctx, path.encode()))

    def set_pupil_target(self, radius_mm=0.0, z_mm=0.0):
        self._ck(self.lib.lf_set_pupil_target(self.ctx, C.c_float(radius_mm), C.c_float(z_mm)))

    def pupil_target(self):
        r, z, zs = C.c_float(), C.c_float(), C.c_float()
        self._ck(self.lib.lf_get_pupil_target(self.ctx, C.byref(r), C.byref(z), C.byref(zs)))
        return dict(radius_mm=r.value, z_mm=z.value, z_sensor_mm=zs.value)

    def aim_at_exit_pupil(self, margin=1.0):
        self._ck(self.lib.lf_aim_at_exit_pupil(self.ctx, C.c_float(margin)))
        return self.pupil_target()

    def set_flare_arithmetic(self, mode=0):
        """0 auto (exact pow in MT19937 parity mode, fast forms with the counter RNG), 1 exact, 2 fast."""
        self._ck(self.lib.lf_set_flare_arithmetic(self.ctx, int(mode)))

    def set_lens_camera(self, mode=1, world_per_mm=0.001, exposure=0.0):
        """The scene term's sample loop images the scene through the prescription (0 = pinhole)."""
        self._ck(self.lib.lf_set_lens_camera(self.ctx, int(mode), C.c_double(world_per_mm), C.c_double(exposure)))

    def set_lens_camera_aim(self, margin=0.0):
        """> 0: the lens camera's samples aim at the exit pupil's image x margin (0: at the march's disc)."""
        self._ck(self.lib.lf_set_lens_camera_aim(self.ctx, C.c_float(margin)))

    def lens_camera(self):
        m, w, e, z = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._ck(self.lib.lf_get_lens_camera(self.ctx, C.byref(m), C.byref(w), C.byref(e), C.byref(z)))
        return dict(mode=m.value, world_per_mm=w.value, exposure=e.value, entrance_pupil_z_mm=z.value)

    def focus_lens(self, object_distance_mm):
        d = C.c_float()
        self._ck(self.lib.lf_focus_lens(self.ctx, C.c_double(object_distance_mm), C.byref(d)))
        return d.value

    def focus_lens_from_pupil(self, distance_mm):
        """focus_lens with the distance counted from the lens camera's position (the entrance pupil's centre):
        what Camera::focalDistance means"""
        d = C.c_float()
        self._ck(self.lib.lf_focus_lens_from_pupil(self.ctx, C.c_double(distance_mm), C.byref(d)))
        return d.value

    def scene_counters(self):
        v = (C.c_uint64 * 4)()
        self._ck(self.lib.lf_get_scene_counters(self.ctx, v))
        return dict(rays=int(v[0]), isects=int(v[1]), lens_samples=int(v[2]), lens_left=int(v[3]))

    def reset_scene_counters(self):
        self._ck(self.lib.lf_reset_scene_counters(self.ctx))

    def set_ghost_accumulate(self, on):
        self._ck(self.lib.lf_set_ghost_accumulate(self.ctx, int(bool(on))))

    def lens_info(self):
        n, stop, nl, sw, efl = C.c_int(), C.c_int(), C.c_int(), C.c_float(), C.c_double()
        self._ck(self.lib.lf_get_lens_info(self.ctx, C.byref(n), C.byref(stop), C.byref(nl), C.byref(sw), C.byref(efl)))
        return dict(n=n.value, stop=stop.value, n_lambda=nl.value, sensor_width_mm=sw.value, efl_mm=efl.value)

    def native_sqrt(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._ck(self.lib.lf_native_sqrt(self.ctx, x.ctypes.data_as(C.POINTER(C.c_float)),
                                         y.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(x.size)))
        return y

    def native_rcp(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._ck(self.lib.lf_native_rcp(self.ctx, x.ctypes.data_as(C.POINTER(C.c_float)),
                                        y.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(x.size)))
        return y

    def executed_events(self):
        v = C.c_uint64(0)
        self._ck(self.lib.lf_get_executed_events(self.ctx, C.byref(v)))
        return int(v.value)

    def march_stats(self):
        v = (C.c_uint64 * 4)()
        self._ck(self.lib.lf_get_march_stats(self.ctx, v))
        return dict(executed_events=int(v[0]), remarch_lane_events=int(v[1]), remarch_rows=int(v[2]))

    def reset_counters(self):
        self._ck(self.lib.lf_reset_counters(self.ctx))

    # ---- measurement
    def timing_enable(self, on=True):
        self._ck(self.lib.lf_timing_enable(self.ctx, int(on)))

    def timing_reset(self):
        self._ck(self.lib.lf_timing_reset(self.ctx))

    def timing_get(self, kernel):
        n = C.c_int()
        ms = C.c_double()
        self._ck(self.lib.lf_timing_get(self.ctx, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value


GROUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)


class LensFlareGroup:
    """lf_group_*: one process, n devices (one context + stream per device, one RCCL communicator)."""

    def __init__(self, devices):
        self.lib = load_library()
        self.lib.lf_group_ctx.restype = C.c_void_p
        self.lib.lf_group_last_error.restype = C.c_char_p
        self.g = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices)
        st = self.lib.lf_group_create(C.byref(self.g), len(devices), dv)
        if st != 0:
            self.g = C.c_void_p()
            raise LensFlareError(st, "lf_group_create failed")
        self.ranks = [LensFlare(_borrowed=self.lib.lf_group_ctx(self.g, r)) for r in range(len(devices))]

    def _ck(self, st):
        if st != 0:
            raise LensFlareError(st, self.lib.lf_group_last_error(self.g).decode())

    def set_frame(self, W, H):
        self._ck(self.lib.lf_group_set_frame(self.g, int(W), int(H)))
        for r in self.ranks:
            r.W, r.H = int(W), int(H)

    def for_each(self, fn):
        """fn(LensFlare, rank) on one host thread per device, concurrently."""
        errors = []

        def tramp(ctx, rank, _user):
            try:
                fn(self.ranks[rank], rank)
                return 0
            except LensFlareError as e:
                errors.append(e)
                return e.status
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                return 1
        cb = GROUP_FN(tramp)
        st = self.lib.lf_group_for_each(self.g, cb, None)
        if errors:
            raise errors[0]
        self._ck(st)

    def gather(self, which):
        self._ck(self.lib.lf_group_gather(self.g, int(which)))

    def set_block_deal(self, on=True):
        """the group's frame dealt by 64 x 64-pixel blocks (True) or by tile rows (False, the default)"""
        self._ck(self.lib.lf_group_set_block_deal(self.g, int(bool(on))))

    def share_cull(self, spp):
        """the pre-pass of the next trace_ghosts(spp) shared between the group's devices"""
        self._ck(self.lib.lf_group_share_cull(self.g, int(spp)))

    def close(self):
        if self.g:
            for r in self.ranks:
                r.ctx = C.c_void_p()
            self.lib.lf_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
