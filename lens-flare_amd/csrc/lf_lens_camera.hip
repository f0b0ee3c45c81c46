// lf_lens_camera.hip -- the lens camera of the scene term (round 4): host side.
//
// The north star asks that "for each sensor sample" a ray be marched through the lens prescription and
// radiance be accumulated into the sensor buffer, behind LensCamera::generate_ray.  The reference has
// the call site -- the sample loop of PathTracer::raytrace_pixel calls camera->generate_ray(x, y) and
// traces what comes back (pathtracer.cpp:841-850) -- but only a pinhole behind it (camera.cpp:278-305;
// the thin-lens variant is a stub that returns a constant ray, camera_lens.cpp:22-30, and
// Camera::generate_ray is not virtual, camera.h:166).  With a lens camera selected
// (lf_set_lens_camera) the device sample loop (lf_scene.hip, k_scene_term<.., LENS = true>) replaces
// that one call: sample s of pixel (x, y) is the MARCH's sample -- the same Philox block, sensor point,
// pupil stratum and sub-cell (lf_march_events.h sample_start) -- its primary path N-1 .. 0 is marched
// with the Fresnel / aperture weight (primary_path: the arithmetic of k_lens_rays and of the ghost
// march), the exit ray is carried into the scene by the camera's pose, traced through the BVH and
// shaded as before, and the radiance is weighted by the transmitted fraction.  A sample that the lens
// blocks contributes nothing and still counts in the mean.
//
// This file: the interface table of the primary path (all wavelengths), the paraxial entrance pupil
// (where the camera position sits in the lens), focusing, and the exposure calibration.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "lf_internal.h"

// host: one row per interface in the order the primary path meets them, constants as pack_program
// derives them for a ray travelling -z (n_in = the medium behind the interface, n_out = in front)
lf_status lf_upload_primary_table(lf_ctx* ctx) {
  const LfLensDev& L = ctx->lens;
  LfPrimaryDev P;
  std::memset(&P, 0, sizeof(P));
  P.n = L.n_surf; P.n_lambda = L.n_lambda;
  P.inv_stop_h = 1.0f / L.stop_h;
  P.front_zv = L.surf[0].zv;
  for (int l = 0; l < L.n_lambda; l++) P.n_start[l] = L.n_start[l];
  for (int e = 0; e < L.n_surf; e++) {
    const int k = L.n_surf - 1 - e;
    const LfSurfaceDev& s = L.surf[k];
    LfPrimaryRow& w = P.row[e];
    w.dzv = (k == L.n_surf - 1 ? L.z_sensor : L.surf[k + 1].zv) - s.zv;
    w.curv = s.curv; w.ch = 0.5f * s.curv; w.c2 = 2.0f * s.curv; w.sc = -s.curv; w.h2 = s.h2;
    w.kind = (s.is_stop != 0.0f ? LF_EV_STOP : 0) | (s.curv == 0.0f ? LF_EV_FLAT : 0);
    for (int l = 0; l < L.n_lambda; l++) {
      const float n_in = s.n_after[l], n_out = s.n_before[l];
      const float n_in2 = n_in * n_in, n_out2 = n_out * n_out;
      w.cn22[l] = w.c2 * n_in2;
      w.rn2[l] = s.curv == 0.0f ? 0.0f : s.radius / n_in2;
      w.delta[l] = n_out2 - n_in2;
      const float q = std::fmaf(n_out2, n_in, n_in2 * n_out);
      w.fs[l] = 1.0f / (n_in + n_out);
      w.fo[l] = n_out2 / q;
      w.fi[l] = n_in2 / q;
    }
  }
  if (!ctx->primary_dev) LF_HIP(ctx, hipMalloc((void**)&ctx->primary_dev, sizeof(LfPrimaryDev)));
  // (the context's stream is non-blocking: a kernel that still reads the previous table must finish first)
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(ctx->primary_dev, &P, sizeof(P), hipMemcpyHostToDevice));
  return LF_OK;
}

void lf_fill_lenscam_args(const lf_ctx* ctx, LfLensCamArgs* a) {
  const LfLensDev& L = ctx->lens;
  std::memset(a, 0, sizeof(*a));
  a->mode = ctx->lenscam_mode;
  a->lambda_ref = L.n_lambda / 2;
  a->n_lambda = L.n_lambda;
  a->W = ctx->W;
  // the strata of the march's sampling specification, for ns_aa samples per pixel
  const int spp = std::max(1, ctx->ns_aa);
  int G = (int)std::floor(std::sqrt((double)spp));
  while ((G + 1) * (G + 1) <= spp) G++;
  while (G * G > spp) G--;
  a->G = G; a->inv_G = 1.0f / (float)G;
  // The march's sample s aims at pupil cell s (row-major over the G x G strata): fine for a loop that
  // always runs to its end, but the sample loop of raytrace_pixel may stop early (the adaptive test every
  // samplesPerBatch samples, pathtracer.cpp:862-868), and the first rows of cells are the pupil's rim.
  // So the loop visits the SAME samples in a scattered order -- i -> (i * step) mod ns_aa, step the
  // integer nearest to ns_aa / golden ratio that is coprime to ns_aa (a bijection) -- whose every prefix
  // is spread over the pupil.  Run to its end the loop has marched exactly the march's samples.
  {
    auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
    int step = std::max(1, (int)std::lround(0.6180339887498949 * (double)spp));
    while (gcd(step, spp) != 1) step++;
    a->order_step = step % spp == 0 ? 1 : step;
  }
  a->xs = ctx->march_xstride_log2;
  a->sub_bits = ctx->march_sub_bits;
  a->inv_sub = 1.0f / (float)(1 << ctx->march_sub_bits);
  a->pitch = ctx->sensor_w_mm / (float)std::max(1, ctx->W);
  a->half_w = 0.5f * (float)ctx->W; a->half_h = 0.5f * (float)ctx->H;
  if (ctx->lenscam_aim_margin > 0.0f) {   // the lens camera's own disc: the exit pupil's image (lf_set_lens_camera_aim)
    a->pupil_h = ctx->lenscam_pupil_h; a->vz = ctx->lenscam_pupil_z - L.z_sensor; a->geom_norm = ctx->lenscam_geom_norm;
  } else {                                // the march's disc: the scene ray IS the march's primary path
    a->pupil_h = L.pupil_h; a->vz = L.pupil_z - L.z_sensor; a->geom_norm = L.geom_norm;
  }
  a->mw = ctx->ap[LF_APERTURE_STARBURST].w; a->mh = ctx->ap[LF_APERTURE_STARBURST].h;
  a->exposure = ctx->lenscam_exposure;
  a->world_per_mm = ctx->lenscam_world_per_mm;
  a->z_ref_mm = ctx->lenscam_z_ref;
  for (int l = 0; l < L.n_lambda; l++)
    for (int c = 0; c < 3; c++) a->lambda_rgb[l][c] = L.lambda_rgb[l][c];
}

// The calibration: the mean transmitted weight of the on-axis sensor point over a fixed 64 x 64 grid of
// pupil-square points (cell centres), reference wavelength.  1 / that mean is the exposure under which
// a scene of uniform radiance L renders as L at the centre of the frame -- what the pinhole camera of
// the reference returns everywhere (camera.cpp:278-305 has no cos^4, no vignetting, no glass).
static lf_status calibrate_exposure(lf_ctx* ctx, double* exposure) {
  constexpr int kGrid = 64, kN = kGrid * kGrid;
  // (the grid covers the disc the lens camera's samples aim at: its own, if it has one)
  struct Swap {
    lf_ctx* c; float h, z, g; bool on;
    explicit Swap(lf_ctx* ctx) : c(ctx), h(ctx->lens.pupil_h), z(ctx->lens.pupil_z), g(ctx->lens.geom_norm),
                                 on(ctx->lenscam_aim_margin > 0.0f) {
      if (on) { c->lens.pupil_h = c->lenscam_pupil_h; c->lens.pupil_z = c->lenscam_pupil_z; c->lens.geom_norm = c->lenscam_geom_norm; }
    }
    ~Swap() { if (on) { c->lens.pupil_h = h; c->lens.pupil_z = z; c->lens.geom_norm = g; } }
  } swap(ctx);
  std::vector<float> xy(2 * kN, 0.0f), uv(2 * kN), out(8 * kN);
  for (int j = 0; j < kGrid; j++)
    for (int i = 0; i < kGrid; i++) {
      uv[2 * (j * kGrid + i)] = (2.0f * ((float)i + 0.5f)) / (float)kGrid - 1.0f;
      uv[2 * (j * kGrid + i) + 1] = (2.0f * ((float)j + 0.5f)) / (float)kGrid - 1.0f;
    }
  const lf_status st = lf_generate_lens_rays(ctx, ctx->lens.n_lambda / 2, kN, xy.data(), uv.data(), out.data());
  if (st != LF_OK) return st;
  double sum = 0.0;
  for (int i = 0; i < kN; i++) sum += (double)out[8 * i + 6];
  if (!(sum > 0.0))
    return lf_fail(ctx, LF_ERR_INVALID, "lens camera: no on-axis sample passes the lens (closed stop / wrong pupil target)");
  *exposure = (double)kN / sum;
  return LF_OK;
}

lf_status lf_lenscam_prepare(lf_ctx* ctx) {
  if (!ctx->lenscam_dirty) return LF_OK;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lens camera: no prescription (lf_set_lens / lf_load_lens_file)");
  if (!ctx->ap[LF_APERTURE_STARBURST].valid)
    return lf_fail(ctx, LF_ERR_STATE, "lens camera: aperture mask (LF_APERTURE_STARBURST slot) not set");
  // the camera position is the centre of the entrance pupil: with the stop closed to a point every
  // chief ray passes through it, which is the pinhole the reference's camera is
  double z = 0.0, m = 1.0;
  if (ctx->raw_stop >= 0) {
    const lf_status st = lf_paraxial_entrance_pupil(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                                                    ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, &z, &m);
    if (st != LF_OK)
      return lf_fail(ctx, LF_ERR_INVALID, "lens camera: the stop has no finite paraxial image through the front group");
  }
  ctx->lenscam_z_ref = z;
  if (ctx->lenscam_aim_margin > 0.0f) {
    // the paraxial image of the stop's open part through the rear group, times the margin (room for the pupil's
    // aberration off the axis): unbiased for the primary path, the only path the lens camera marches
    if (ctx->raw_stop < 0) return lf_fail(ctx, LF_ERR_INVALID, "lf_set_lens_camera_aim: the prescription has no stop");
    double zx = 0.0, mx = 0.0;
    if (lf_paraxial_exit_pupil(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                               ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, &zx, &mx) != LF_OK ||
        !(zx < (double)ctx->lens.z_sensor))
      return lf_fail(ctx, LF_ERR_INVALID, "lf_set_lens_camera_aim: the stop has no usable paraxial image behind it");
    const double open = std::min(1.0, ctx->ap[LF_APERTURE_STARBURST].open_radius);
    ctx->lenscam_pupil_h = (float)((double)ctx->lens.stop_h * open * std::fabs(mx) * (double)ctx->lenscam_aim_margin);
    ctx->lenscam_pupil_z = (float)zx;
    const double D = (double)ctx->lens.z_sensor - (double)ctx->lenscam_pupil_z;
    ctx->lenscam_geom_norm = (float)((3.14159265358979323846 * (double)ctx->lenscam_pupil_h * (double)ctx->lenscam_pupil_h) / (D * D));
  }
  lf_status st = lf_upload_primary_table(ctx);
  if (st != LF_OK) return st;
  if (ctx->lenscam_exposure_req > 0.0) {
    ctx->lenscam_exposure = ctx->lenscam_exposure_req;
  } else {
    st = calibrate_exposure(ctx, &ctx->lenscam_exposure);
    if (st != LF_OK) return st;
  }
  ctx->lenscam_dirty = false;
  return LF_OK;
}

extern "C" {

lf_status lf_paraxial_entrance_pupil(int n, int stop, const float* radius, const float* thickness,
                                     const float* ior_row, double* z_mm, double* magnification) {
  if (n < 1 || n > LF_MAX_SURFACES || stop < 0 || stop >= n || !radius || !thickness || !ior_row || !z_mm || !magnification)
    return LF_ERR_INVALID;
  // The stop's centre imaged by the interfaces IN FRONT of it, the ray travelling towards the scene.
  // In the mirrored coordinate s = -z the ray travels +s and interface k has curvature -c_k, so the
  // reference's operators apply as they are (pathtracer.cpp:527-533): T(d) = [[1, d], [0, 1]],
  // R(c, n1, n2) = [[1, 0], [c (n1 - n2) / n2, n1 / n2]] with n1 = the medium behind the interface.
  double A = 1, B = 0, Cc = 0, D = 1;   // system matrix stop plane -> front vertex (height, angle)
  double z_stop = 0.0;
  for (int k = 0; k < stop; k++) z_stop += thickness[k];
  (void)z_stop;
  // the medium the stop sits in = behind interface stop - 1
  double nm = stop > 0 ? (double)ior_row[stop - 1] : 1.0;
  for (int k = stop - 1; k >= 0; k--) {
    const double d = thickness[k];                   // vertex k -> vertex k + 1
    A += d * Cc; B += d * D;                         // translate to interface k
    const double c = radius[k] == 0.0f ? 0.0 : -1.0 / (double)radius[k];
    const double n2 = k > 0 ? (double)ior_row[k - 1] : 1.0;   // the medium in front of interface k
    if (!(nm >= 1.0) || !(n2 >= 1.0)) return LF_ERR_INVALID;
    const double r10 = c * (nm - n2) / n2, r11 = nm / n2;
    const double c2 = r10 * A + r11 * Cc, d2 = r10 * B + r11 * D;
    Cc = c2; D = d2;
    nm = n2;
  }
  if (D == 0.0) return LF_ERR_INVALID;   // the stop is imaged at infinity (object-space telecentric)
  const double l = -B / D;               // image distance beyond the front vertex along +s (towards the scene)
  *z_mm = 0.0 - l;                       // the front vertex is at z = 0
  *magnification = A + l * Cc;
  return LF_OK;
}

lf_status lf_set_lens_camera(lf_ctx* ctx, int mode, double world_per_mm, double exposure) {
  if (!ctx || mode < 0 || mode > 2) return LF_ERR_INVALID;
  if (mode != 0 && (!(world_per_mm > 0.0) || !std::isfinite(world_per_mm) || !std::isfinite(exposure)))
    return lf_fail(ctx, LF_ERR_INVALID, "lens camera: world_per_mm must be > 0 and finite");
  ctx->lenscam_mode = mode;
  if (mode != 0) {
    // (a host that repeats the call every frame does not trigger a new calibration)
    if (exposure != ctx->lenscam_exposure_req) ctx->lenscam_dirty = true;
    ctx->lenscam_world_per_mm = world_per_mm;
    ctx->lenscam_exposure_req = exposure;
  }
  return LF_OK;
}

lf_status lf_set_lens_camera_aim(lf_ctx* ctx, float margin) {
  if (!ctx || !std::isfinite(margin)) return LF_ERR_INVALID;
  const float m = margin > 0.0f ? margin : 0.0f;
  if (m != ctx->lenscam_aim_margin) ctx->lenscam_dirty = true;
  ctx->lenscam_aim_margin = m;
  return LF_OK;
}

lf_status lf_get_lens_camera(lf_ctx* ctx, int* mode, double* world_per_mm, double* exposure,
                             double* entrance_pupil_z_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->lenscam_mode != 0) {
    LF_HIP(ctx, hipSetDevice(ctx->device));
    const lf_status st = lf_lenscam_prepare(ctx);
    if (st != LF_OK) return st;
  }
  if (mode) *mode = ctx->lenscam_mode;
  if (world_per_mm) *world_per_mm = ctx->lenscam_world_per_mm;
  if (exposure) *exposure = ctx->lenscam_exposure;
  if (entrance_pupil_z_mm) *entrance_pupil_z_mm = ctx->lenscam_z_ref;
  return LF_OK;
}

// Camera::focalDistance is measured from the CAMERA POSITION, and the lens camera puts that at the centre of the
// paraxial entrance pupil (z_ep behind the first vertex: +19.95 mm for the double Gauss) -- not at the first vertex,
// from which lf_focus_lens measures.  At 4 m the 2 cm do not matter; at 0.3 - 0.5 m they are several pixels of blur.
lf_status lf_focus_lens_from_pupil(lf_ctx* ctx, double distance_from_entrance_pupil_mm, float* sensor_distance_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_focus_lens_from_pupil before lf_set_lens");
  if (!(distance_from_entrance_pupil_mm > 0.0) || std::isinf(distance_from_entrance_pupil_mm))
    return lf_focus_lens(ctx, distance_from_entrance_pupil_mm, sensor_distance_mm);   // infinity
  double z_ep = 0.0, mag = 1.0;
  if (ctx->raw_stop >= 0 &&
      lf_paraxial_entrance_pupil(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                                 ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, &z_ep, &mag) != LF_OK)
    z_ep = 0.0;
  const double from_vertex = distance_from_entrance_pupil_mm - z_ep;
  if (!(from_vertex > 0.0)) return lf_fail(ctx, LF_ERR_INVALID, "lf_focus_lens_from_pupil: the object lies inside the lens");
  return lf_focus_lens(ctx, from_vertex, sensor_distance_mm);
}

lf_status lf_focus_lens(lf_ctx* ctx, double object_distance_mm, float* sensor_distance_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_focus_lens before lf_set_lens");
  if (std::isnan(object_distance_mm)) return LF_ERR_INVALID;
  const int n = ctx->raw_n;
  const float* ior = ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * n;
  // a paraxial ray from the axial object point (or parallel to the axis for an object at infinity)
  // through every interface: T / R as in lf_paraxial_efl; where it crosses the axis behind the last
  // vertex is where the sensor goes
  const bool at_infinity = !(object_distance_mm > 0.0) || std::isinf(object_distance_mm);
  double y = at_infinity ? 1.0 : object_distance_mm * 1e-3, u = at_infinity ? 0.0 : 1e-3, n1 = 1.0;
  for (int k = 0; k < n; k++) {
    if (k != ctx->raw_stop) {
      const double c = ctx->raw_radius[k] == 0.0f ? 0.0 : 1.0 / (double)ctx->raw_radius[k], n2 = ior[k];
      u = c * (n1 - n2) / n2 * y + n1 / n2 * u;
      n1 = n2;
    }
    if (k + 1 < n) y += (double)ctx->raw_thickness[k] * u;
  }
  if (!(u < 0.0)) return lf_fail(ctx, LF_ERR_INVALID, "lf_focus_lens: the object has no real image behind the lens");
  const double back = -y / u;
  if (!(back > 0.0) || !std::isfinite(back))
    return lf_fail(ctx, LF_ERR_INVALID, "lf_focus_lens: the image lies inside the lens");
  float radius[LF_MAX_SURFACES], thick[LF_MAX_SURFACES], semi[LF_MAX_SURFACES];
  float iorc[LF_MAX_LAMBDA * LF_MAX_SURFACES];
  std::memcpy(radius, ctx->raw_radius, sizeof(radius));
  std::memcpy(thick, ctx->raw_thickness, sizeof(thick));
  std::memcpy(semi, ctx->raw_semi_ap, sizeof(semi));
  std::memcpy(iorc, ctx->raw_ior, sizeof(iorc));
  thick[n - 1] = (float)back;
  // (the pair selection, the wavelength weights and a pupil target that is still in front of the
  // sensor survive: only the last thickness changes)
  float keep_rgb[LF_MAX_LAMBDA][3];
  std::memcpy(keep_rgb, ctx->lens.lambda_rgb, sizeof(keep_rgb));
  const float keep_h = ctx->pupil_target_h, keep_z = ctx->pupil_target_z;
  lf_derive_lens(ctx, n, ctx->raw_stop, ctx->lens.n_lambda, radius, thick, iorc, semi, ctx->sensor_w_mm);
  std::memcpy(ctx->lens.lambda_rgb, keep_rgb, sizeof(keep_rgb));
  ctx->raw_thickness[n - 1] = thick[n - 1];
  if (keep_h > 0.0f && keep_z < ctx->lens.z_sensor) { ctx->pupil_target_h = keep_h; ctx->pupil_target_z = keep_z; }
  else { ctx->pupil_target_h = 0.0f; ctx->pupil_target_z = 0.0f; }
  lf_apply_pupil_target(ctx);
  ctx->events_dirty = true;
  ctx->lenscam_dirty = true;
  if (sensor_distance_mm) *sensor_distance_mm = thick[n - 1];
  return LF_OK;
}

lf_status lf_get_scene_counters(lf_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return LF_ERR_INVALID;
  for (int i = 0; i < 4; i++) out[i] = 0;
  if (!ctx->scene_counters_dev) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long h[kSceneCounters];
  LF_HIP(ctx, hipMemcpy(h, ctx->scene_counters_dev, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < kSceneCounters; i++) out[i] = h[i];
  return LF_OK;
}

lf_status lf_reset_scene_counters(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->scene_counters_dev) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipMemsetAsync(ctx->scene_counters_dev, 0, sizeof(unsigned long long) * kSceneCounters, ctx->stream));
  return LF_OK;
}

}  // extern "C"
