// lf_flare_kernels.hip -- gfx950 kernels for everything PathTracer::raytrace_pixel adds on top
// of the scene term (reference: src/pathtracer/pathtracer.cpp):
//
//   frame_setup   find_sun_pos (:32-64) + the 78 paraxial marginal-ray traces (:588-689) +
//                 draw_ghost's quad set-up (:412-508), one lane per ghost
//   ghost_raster  rasterize_textured_triangle / fill_textured_pixel (:305-410), gather form
//   spectrum      raytrace_starburst's per-pixel direct DFT (:947-974) collapsed to ONE separable
//                 DFT of the aperture (phases are integer multiples of 2*pi/Aw)
//   flare_layer   starburst shaping (:976-1000) + calculate_irradiance_falloff (:1043-1063) +
//                 the sampleBuffer composition of raytrace_pixel (:875-891)
//   tonemap       HDRImageBuffer::toColor (util/image.h:208-223, :53-62)
//
// Built with -ffp-contract=off: the float barycentrics and the float determinant of invert2x2
// must round exactly like the reference's (which is built without FMA) for the ghost buffer to be
// bit-identical.  All double transcendental calls go to ROCm's device libm (<= 1 ulp).
#include <algorithm>
#include <cstring>

#include "lf_internal.h"

namespace {

// =============================================================================================
// aperture statistics (CameraApertureTexture::init, camera.h:54-72)
// =============================================================================================
__global__ void k_aperture_stats(const float* __restrict__ tex, int w, int h,
                                 lf_aperture_stats* __restrict__ st) {
  // total_value: every partial sum of these float texels is exactly representable in a double
  // (< 2^18 texels of <= 24 significant bits down to 2^-31), so the reduction order is free.
  double sum = 0.0;
  int mnx = w, mny = w, mxx = -1, mxy = -1;
  const int n = w * h;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v = tex[i];
    sum += (double)v;
    if (v > 0) {
      int x = i % w, y = i / w;
      mnx = min(mnx, x); mny = min(mny, y);
      mxx = max(mxx, x); mxy = max(mxy, y);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off);
    mnx = min(mnx, __shfl_down(mnx, off)); mny = min(mny, __shfl_down(mny, off));
    mxx = max(mxx, __shfl_down(mxx, off)); mxy = max(mxy, __shfl_down(mxy, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&st->total_value, sum);
    atomicMin(&st->min_x, mnx); atomicMin(&st->min_y, mny);
    atomicMax(&st->max_x, mxx); atomicMax(&st->max_y, mxy);
  }
}

// =============================================================================================
// starburst spectrum
// =============================================================================================
// tw[m] = exp(+j 2 pi m / Aw)
__global__ void k_twiddle(double2* __restrict__ tw, int aw) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= aw) return;
  double s, c;
  sincospi(2.0 * (double)m / (double)aw, &s, &c);
  tw[m] = make_double2(c, s);
}

// Row pass: G[yc - min_y][ka] = sum_{xc in bbox} A[yc][xc] * tw[(xc*ka) mod Aw].
// One workgroup per aperture row; the twiddle table and the row live in LDS.
__global__ void k_dft_rows(const float* __restrict__ tex, const lf_aperture_stats* __restrict__ st,
                           const double2* __restrict__ tw, double2* __restrict__ G) {
  extern __shared__ double lds[];
  const int aw = st->width;
  const int min_x = st->min_x, max_x = st->max_x, min_y = st->min_y, max_y = st->max_y;
  const int yc = min_y + blockIdx.x;
  if (yc > max_y) return;
  double2* s_tw = reinterpret_cast<double2*>(lds);
  double* s_row = lds + 2 * aw;
  for (int i = threadIdx.x; i < aw; i += blockDim.x) {
    s_tw[i] = tw[i];
    s_row[i] = (double)tex[(size_t)yc * aw + i];
  }
  __syncthreads();
  for (int ka = threadIdx.x; ka < aw; ka += blockDim.x) {
    double re = 0.0, im = 0.0;
    int idx = (int)(((long long)min_x * ka) % aw);
    for (int xc = min_x; xc <= max_x; xc++) {
      double a = s_row[xc];
      double2 t = s_tw[idx];
      re += a * t.x;
      im += a * t.y;
      idx += ka;
      if (idx >= aw) idx -= aw;
    }
    G[(size_t)blockIdx.x * aw + ka] = make_double2(re, im);
  }
}

// Column pass: S[kb][ka] = | sum_{yc in bbox} G[yc][ka] * tw[(yc*kb) mod Aw] | / total_value
__global__ void k_dft_cols(const lf_aperture_stats* __restrict__ st, const double2* __restrict__ tw,
                           const double2* __restrict__ G, double* __restrict__ S) {
  const int aw = st->width;
  const int min_y = st->min_y, max_y = st->max_y;
  const int kb = blockIdx.x;
  const double total = st->total_value;
  for (int ka = threadIdx.x; ka < aw; ka += blockDim.x) {
    double re = 0.0, im = 0.0;
    int idx = (int)(((long long)min_y * kb) % aw);
    for (int yc = min_y; yc <= max_y; yc++) {
      double2 g = G[(size_t)(yc - min_y) * aw + ka];
      double2 t = tw[idx];  // wave-uniform address
      re += g.x * t.x - g.y * t.y;
      im += g.x * t.y + g.y * t.x;
      idx += kb;
      if (idx >= aw) idx -= aw;
    }
    S[(size_t)kb * aw + ka] = hypot(re, im) / total;
  }
}

// =============================================================================================
// frame set-up: sun projection, paraxial traces, ghost quads
// =============================================================================================
struct M2 { double a, b, c, d; };  // [[a b][c d]]; the reference's zero-padded Matrix3x3

__device__ inline M2 m2_mul(const M2& A, const M2& B) {
  // Matrix3x3::operator* (CGL/src/matrix3x3.cpp:99-114): column k of the product is
  // B(0,k)*A[0] + B(1,k)*A[1] (+ 0), evaluated left to right in double
  M2 P;
  P.a = B.a * A.a + B.c * A.b;
  P.c = B.a * A.c + B.c * A.d;
  P.b = B.b * A.a + B.d * A.b;
  P.d = B.b * A.c + B.d * A.d;
  return P;
}
__device__ inline M2 m2_T(float d) { return M2{1.0, (double)d, 0.0, 1.0}; }       // :527-529
__device__ inline M2 m2_R(float c, float n1, float n2) {                           // :531-533
  float e = __fdiv_rn(c * (n1 - n2), n2);
  float f = __fdiv_rn(n1, n2);
  return M2{1.0, 0.0, (double)e, (double)f};
}
__device__ inline M2 m2_L(float c) { return M2{1.0, 0.0, (double)(2 * c), 1.0}; }  // :535-537
__device__ inline M2 m2_inv(const M2& m) {                                         // :519-525
  float a = (float)m.a, b = (float)m.b, c = (float)m.c, d = (float)m.d;
  float det = a * d - b * c;  // float, unfused
  double s = 1.0 / (double)det;
  return M2{(double)d * s, (double)(-b) * s, (double)(-c) * s, (double)a * s};
}

struct ParaxialTables { M2 T[LF_MAX_SURFACES], R[LF_MAX_SURFACES], L[LF_MAX_SURFACES]; };

__device__ void build_tables(const LfParaxialLens& pl, int colour, ParaxialTables& tb) {
  float prev_n = 1.00f;  // create_Rs_for_color :559-568
  for (int k = 0; k < pl.n; k++) {
    tb.T[k] = m2_T(pl.thickness[k]);
    tb.R[k] = m2_R(pl.curvature[k], prev_n, pl.ior[colour][k]);
    prev_n = pl.ior[colour][k];
    tb.L[k] = m2_L(pl.curvature[k]);
  }
}

__device__ inline void clip_at_stop(const LfParaxialLens& pl, const M2& M, float r, float theta,
                                    double& ray_r, double& ray_t) {
  // :619-629 / :654-664
  double after = ray_r * M.a + ray_t * M.b;
  if (after > pl.clip || after < -pl.clip) {
    float r_a = pl.recast_pos;
    if (r < 0) r_a = pl.recast_neg;
    float r_e = (float)(((double)r_a - M.b * (double)theta) / M.a);
    ray_r = (double)r_e;
    ray_t = (double)theta;
  }
}

// height of the marginal ray at the sensor after reflecting at j then i (i < j)
__device__ double trace_pair(const LfParaxialLens& pl, const ParaxialTables& tb, float r,
                             float theta, int i, int j, bool after_stop) {
  double ray_r = (double)r, ray_t = (double)theta;
  M2 M{1.0, 0.0, 0.0, 1.0};
  for (int k = 0; k < j; k++) {
    if (after_stop && k == pl.stop) {  // trace_ray_auto_after clips in the first leg (:653-670)
      clip_at_stop(pl, M, r, theta, ray_r, ray_t);
      M = m2_mul(tb.T[k], M);
      continue;
    }
    M = m2_mul(m2_mul(tb.T[k], tb.R[k]), M);
  }
  M = m2_mul(tb.L[j], M);
  for (int k = j - 1; k > i; k--) M = m2_mul(m2_mul(m2_inv(tb.R[k]), tb.T[k]), M);
  M = m2_mul(m2_mul(m2_mul(tb.T[i], m2_inv(tb.L[i])), tb.T[i]), M);
  for (int k = i + 1; k < pl.n; k++) {
    if (!after_stop && k == pl.stop) {  // trace_ray_auto_before clips in the last leg (:618-633)
      clip_at_stop(pl, M, r, theta, ray_r, ray_t);
      M = m2_mul(tb.T[k], M);
      continue;
    }
    M = m2_mul(m2_mul(tb.T[k], tb.R[k]), M);
  }
  return ray_r * M.a + ray_t * M.b;
}

__device__ inline void shift_vertex(double sc, double cs, double nsn, double sn, double shx,
                                    double shy, double x, double y, double& ox, double& oy) {
  // shift*rotation*scaling*(x,y,1) of :412-430 with the 3x3 products written out; every
  // product and sum below is one the reference performs, in the same order
  ox = (x * (sc * cs) + y * (sc * nsn)) + shx;
  oy = (x * (sc * sn) + y * (sc * cs)) + shy;
}

__device__ void make_tri(LfGhostTri& t, float x0, float y0, float u0, float v0, float x1, float y1,
                         float u1, float v1, float x2, float y2, float u2, float v2, int W, int H,
                         const double colour[3]) {
  // rasterize_textured_triangle :350-394
#define LF_SWAP(a, b) { float tmp_ = a; a = b; b = tmp_; }
  if (y1 < y0) { LF_SWAP(x0, x1) LF_SWAP(y0, y1) LF_SWAP(u0, u1) LF_SWAP(v0, v1) }
  if (y2 < y0) { LF_SWAP(x0, x2) LF_SWAP(y0, y2) LF_SWAP(u0, u2) LF_SWAP(v0, v2) }
  if (y2 < y1) { LF_SWAP(x1, x2) LF_SWAP(y1, y2) LF_SWAP(u1, u2) LF_SWAP(v1, v2) }
#undef LF_SWAP
  x0 -= 0.5f; y0 -= 0.5f; x1 -= 0.5f; y1 -= 0.5f; x2 -= 0.5f; y2 -= 0.5f;
  t.x0 = x0; t.y0 = y0; t.u0 = u0; t.v0 = v0;
  t.x1 = x1; t.y1 = y1; t.u1 = u1; t.v1 = v1;
  t.x2 = x2; t.y2 = y2; t.u2 = u2; t.v2 = v2;
  t.bx0 = max(0, (int)floorf(fminf(fminf(x0, x1), x2)));
  t.bx1 = min(W - 1, (int)ceilf(fmaxf(fmaxf(x0, x1), x2)));
  t.by0 = max(0, (int)floorf(y0));
  t.by1 = min(H - 1, (int)ceilf(y2));
  t.colour[0] = colour[0]; t.colour[1] = colour[1]; t.colour[2] = colour[2];
}

// draw_ghost :433-508: the textured quad (two triangles) of one ghost from the sensor heights r1, r2
// of its two marginal rays; channel 0, 1, 2 = "red", "green", anything else ("blue", :482-488)
__device__ void make_ghost_quad(LfGhostTri* two, float r1, float r2, int channel, double ax, double ay,
                                int W, int H, int tex_w, int tex_h) {
  float shift_amt = (float)((double)(-(r1 + r2) / 2) * 0.4);
  float scale_amt = (float)((double)fabsf(r2 - r1) * 0.2);
  double mid_w = ceil(ax * (double)W), mid_h = ceil(ay * (double)H);
  float ang = (float)atan((ay - 0.5) / (ax - 0.5));
  float cs = (float)cos((double)ang), sn = (float)sin((double)ang);  // cosf / sinf
  float shx = shift_amt * cs, shy = shift_amt * sn;
  double ulx, uly, llx, lly, urx, ury, lrx, lry;
  shift_vertex(scale_amt, cs, -sn, sn, shx, shy, -1, 1, ulx, uly);
  shift_vertex(scale_amt, cs, -sn, sn, shx, shy, -1, -1, llx, lly);
  shift_vertex(scale_amt, cs, -sn, sn, shx, shy, 1, 1, urx, ury);
  shift_vertex(scale_amt, cs, -sn, sn, shx, shy, 1, -1, lrx, lry);
  float intensity_scalar = 10;
  float size_scalar = __fdiv_rn(1.0f, scale_amt * scale_amt);
  double colour[3] = {0.0, 0.0, 0.0};   // Vector3D(1,0,0) *= float: the zeros stay zeros
  colour[channel] = (double)(intensity_scalar * size_scalar);
  float th = (float)tex_h, tw = (float)tex_w;
  make_tri(two[0], (float)(mid_w + ulx), (float)(mid_h + uly), 0, 0,
           (float)(mid_w + llx), (float)(mid_h + lly), 0, th, (float)(mid_w + urx),
           (float)(mid_h + ury), tw, 0, W, H, colour);
  make_tri(two[1], (float)(mid_w + lrx), (float)(mid_h + lry), 0, 0,
           (float)(mid_w + llx), (float)(mid_h + lly), 0, th, (float)(mid_w + urx),
           (float)(mid_h + ury), tw, 0, W, H, colour);
}

__global__ void k_frame_setup(const LfParaxialLens* __restrict__ plp, LfCamera cam,
                              LfSunLightArgs light_args, const double* __restrict__ lights_mem,
                              int n_lights, int project, int W,
                              int H, int tex_w, int tex_h, LfFlares* __restrict__ fl,
                              LfGhostList* __restrict__ gl) {
  const LfParaxialLens& pl = *plp;
  if (project) {
    if (threadIdx.x == 0) {
      // find_sun_pos :32-64 + Camera::analyze_world_coord camera.cpp:245-273
      const double PI_ = 3.14159265358979323;
      double edge_x = tan(0.5 * (cam.hfov_deg * (PI_ / 180.0)));
      double edge_y = tan(0.5 * (cam.vfov_deg * (PI_ / 180.0)));
      int n = 0;
      const double* lights = lights_mem ? lights_mem : light_args.v;
      for (int l = 0; l < n_lights; l++) {
        double dx = lights[6 * l] - cam.pos[0], dy = lights[6 * l + 1] - cam.pos[1],
               dz = lights[6 * l + 2] - cam.pos[2];
        // c2w.T() * d, each component (dx*col0 + dy*col1) + dz*col2 of the transposed matrix
        double px = (dx * cam.c2w[0] + dy * cam.c2w[3]) + dz * cam.c2w[6];
        double py = (dx * cam.c2w[1] + dy * cam.c2w[4]) + dz * cam.c2w[7];
        double pz = (dx * cam.c2w[2] + dy * cam.c2w[5]) + dz * cam.c2w[8];
        double rc = 1.0 / fabs(pz);
        double nx = (((rc * px) / edge_x) + 1) / 2.0;
        double ny = (((rc * py) / edge_y) + 1) / 2.0;
        if ((nx >= 0 && nx <= 1) && (ny >= 0 && ny <= 1) && n < LF_MAX_FLARES) {
          fl->origin[n][0] = nx; fl->origin[n][1] = ny;
          fl->radiance[n][0] = lights[6 * l + 3];
          fl->radiance[n][1] = lights[6 * l + 4];
          fl->radiance[n][2] = lights[6 * l + 5];
          fl->angle_to_sun = (float)atan(ny / nx);
          fl->axis_ray[0] = nx; fl->axis_ray[1] = ny;
          n++;
        }
      }
      fl->n_flares = n;
    }
    __threadfence_block();
    __syncthreads();
  }
  const double ax = fl->axis_ray[0], ay = fl->axis_ray[1];
  const float theta = fl->angle_to_sun;
  const int n_before = pl.stop * (pl.stop - 1) / 2;
  const int na = pl.n - pl.stop - 1;
  const int n_after = na * (na - 1) / 2;
  const int n_ghosts = 3 * (n_before + n_after);
  if (ax == 0 && ay == 0) {  // generate_ghost_buffer :724-726
    if (threadIdx.x == 0) gl->n_tris = 0;
    return;
  }
  if (threadIdx.x == 0) gl->n_tris = 2 * n_ghosts;
  for (int g = threadIdx.x; g < n_ghosts; g += blockDim.x) {
    int pair = g / 3, colour = g % 3;
    bool after = pair >= n_before;
    int lo = after ? pl.stop + 1 : 0, hi = after ? pl.n : pl.stop;
    int q = after ? pair - n_before : pair, pi = lo, pj = lo + 1;
    for (int i = lo; i < hi; i++) {  // (i, j) in the reference's loop order :735-736, :750-751
      int cnt = hi - 1 - i;
      if (q < cnt) { pi = i; pj = i + 1 + q; break; }
      q -= cnt;
    }
    ParaxialTables tb;
    build_tables(pl, colour, tb);
    float r1 = (float)trace_pair(pl, tb, pl.marginal, theta, pi, pj, after);
    float r2 = (float)trace_pair(pl, tb, -pl.marginal, theta, pi, pj, after);
    make_ghost_quad(&gl->tri[2 * g], r1, r2, colour, ax, ay, W, H, tex_w, tex_h);
  }
}

// =============================================================================================
// ghost rasteriser, gather form: one lane per sensor pixel, triangles visited in the
// reference's draw order so every += lands in the same order as on the CPU
// =============================================================================================
constexpr int kTileW = 64, kTileH = 4;

// fill_textured_pixel :305-343, all float, unfused: acc += texel * ghost_color when (x, y) is inside
__device__ inline void fill_textured_pixel(const LfGhostTri& tr, int x, int y,
                                           const float* __restrict__ tex, int tex_w, int tex_h,
                                           double acc[3]) {
  const float x0 = tr.x0, y0f = tr.y0, x1 = tr.x1, y1f = tr.y1, x2 = tr.x2, y2f = tr.y2;
  float xy_to_01 = -(y1f - y0f) * (x - x0) + (x1 - x0) * (y - y0f);
  float two_to_01 = -(y1f - y0f) * (x2 - x0) + (x1 - x0) * (y2f - y0f);
  float alpha = __fdiv_rn(xy_to_01, two_to_01);
  float xy_to_12 = -(y2f - y1f) * (x - x1) + (x2 - x1) * (y - y1f);
  float zero_to_12 = -(y2f - y1f) * (x0 - x1) + (x2 - x1) * (y0f - y1f);
  float beta = __fdiv_rn(xy_to_12, zero_to_12);
  float gamma = 1 - alpha - beta;
  if (gamma >= 0 && alpha >= 0 && beta >= 0) {
    float u = tr.u2 * alpha + tr.u0 * beta + tr.u1 * gamma;
    float v = tr.v2 * alpha + tr.v0 * beta + tr.v1 * gamma;
    int idx = (int)(floor((double)v) * (double)tex_w + (double)u);  // :338
    // the reference reads its vector unchecked; an index past the end is defined as 0 here
    float s = (idx >= 0 && idx < tex_w * tex_h) ? tex[idx] : 0.0f;
    // sample * ghost_color, then update_pixel_additive (image.h:145): a zero component adds +0.0
    acc[0] += (double)s * tr.colour[0];
    acc[1] += (double)s * tr.colour[1];
    acc[2] += (double)s * tr.colour[2];
  }
}

// accumulate = 0: the pixel is REPLACED by the sum over the list (generate_ghost_buffer: the buffer
// was cleared, :719); 1: the list is added on top of what the pixel holds, triangle by triangle like
// update_pixel_additive (the single-ghost / single-triangle entry points)
__global__ __launch_bounds__(256) void k_ghost_raster(const LfGhostList* __restrict__ gl,
                                                      const float* __restrict__ tex, int tex_w,
                                                      int tex_h, int W, int y0, int y1, int accumulate,
                                                      double* __restrict__ ghost) {
  __shared__ unsigned char s_hit[kMaxGhostTris];
  const int n_tris = gl->n_tris;
  const int tx0 = blockIdx.x * kTileW, ty0 = y0 + blockIdx.y * kTileH;
  const int tx1 = min(tx0 + kTileW, W), ty1 = min(ty0 + kTileH, y1);
  for (int t = threadIdx.x; t < n_tris; t += blockDim.x) {
    const LfGhostTri& tr = gl->tri[t];
    s_hit[t] = (tr.bx0 < tx1 && tr.bx1 > tx0 && tr.by0 < ty1 && tr.by1 > ty0) ? 1 : 0;
  }
  __syncthreads();
  const int x = tx0 + (threadIdx.x & (kTileW - 1)), y = ty0 + (threadIdx.x / kTileW);
  if (x >= W || y >= y1) return;
  double* px = ghost + 3 * ((size_t)x + (size_t)y * W);
  double acc[3] = {0.0, 0.0, 0.0};
  if (accumulate) { acc[0] = px[0]; acc[1] = px[1]; acc[2] = px[2]; }
  for (int t = 0; t < n_tris; t++) {
    if (!s_hit[t]) continue;           // workgroup-uniform
    const LfGhostTri& tr = gl->tri[t]; // uniform address -> scalar loads
    if (x < tr.bx0 || x >= tr.bx1 || y < tr.by0 || y >= tr.by1) continue;
    fill_textured_pixel(tr, x, y, tex, tex_w, tex_h, acc);
  }
  px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2];
}

// ---- the single-shot forms of the reference's public helper members (one wave each) ----------
// PathTracer::draw_ghost(color, r1, r2): the quad of one ghost into the list
__global__ void k_one_ghost(const LfFlares* __restrict__ fl, float r1, float r2, int channel, int W,
                            int H, int tex_w, int tex_h, LfGhostList* __restrict__ gl) {
  if (threadIdx.x != 0) return;
  gl->n_tris = 2;
  make_ghost_quad(gl->tri, r1, r2, channel, fl->axis_ray[0], fl->axis_ray[1], W, H, tex_w, tex_h);
}

struct LfTriArgs { float v[12]; double colour[3]; };
// PathTracer::rasterize_textured_triangle(x0, y0, u0, v0, ..., ghost_color): one triangle into the list
__global__ void k_one_triangle(LfTriArgs a, int W, int H, LfGhostList* __restrict__ gl) {
  if (threadIdx.x != 0) return;
  gl->n_tris = 1;
  make_tri(gl->tri[0], a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], a.v[9],
           a.v[10], a.v[11], W, H, a.colour);
}
// PathTracer::fill_textured_pixel(x0, ..., v2, x, y, ghost_color): vertices as given (the caller has
// sorted and shifted them), one pixel
__global__ void k_one_pixel(LfTriArgs a, int x, int y, const float* __restrict__ tex, int tex_w,
                            int tex_h, int W, double* __restrict__ ghost) {
  if (threadIdx.x != 0) return;
  LfGhostTri tr;
  tr.x0 = a.v[0]; tr.y0 = a.v[1]; tr.u0 = a.v[2]; tr.v0 = a.v[3];
  tr.x1 = a.v[4]; tr.y1 = a.v[5]; tr.u1 = a.v[6]; tr.v1 = a.v[7];
  tr.x2 = a.v[8]; tr.y2 = a.v[9]; tr.u2 = a.v[10]; tr.v2 = a.v[11];
  tr.colour[0] = a.colour[0]; tr.colour[1] = a.colour[1]; tr.colour[2] = a.colour[2];
  double* px = ghost + 3 * ((size_t)x + (size_t)y * W);
  double acc[3] = {px[0], px[1], px[2]};
  fill_textured_pixel(tr, x, y, tex, tex_w, tex_h, acc);
  px[0] = acc[0]; px[1] = acc[1]; px[2] = acc[2];
}
// PathTracer::shift_vertex(x, y, scale, shift_amount) :412-430 with the members' axis_ray
__global__ void k_shift_vertex(const LfFlares* __restrict__ fl, float x, float y, float scale,
                               float shift_amount, double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const double ax = fl->axis_ray[0], ay = fl->axis_ray[1];
  float ang = (float)atan((ay - 0.5) / (ax - 0.5));
  float cs = (float)cos((double)ang), sn = (float)sin((double)ang);
  shift_vertex(scale, cs, -sn, sn, shift_amount * cs, shift_amount * sn, x, y, out[0], out[1]);
}
// PathTracer::compute_phase(flare, u, v, screen_pos) :917-931 with complex_exp(., false) :901-915
__global__ void k_compute_phase(const LfFlares* __restrict__ fl, int flare, double u, double v, int W,
                                int H, double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double lr = ceil(fl->origin[flare][0] * (double)W), ud = ceil(fl->origin[flare][1] * (double)H);
  out[2] = lr; out[3] = ud;
  lr -= (double)W / 2.0;
  ud = -ud + (double)H / 2.0;
  const double e = u * lr + v * ud;
  out[0] = cos(2.0 * 3.14159265358979323846 * e);
  out[1] = sin(2.0 * 3.14159265358979323846 * e);
}

// =============================================================================================
// counter RNG: Philox4x32-10 (Salmon et al. 2011), the order-free replacement of the shared
// MT19937 for throughput runs
// =============================================================================================
__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

__device__ inline double random_uniform_from_raw(unsigned raw) {
  // random_uniform(): clamp(double(raw) * (1/(2^32-1)), 1e-7, 0.99999999) util/random_util.h:15-22
  double rmax = 1.0 / (4294967295.0 - 0.0);
  double v = (double)raw * rmax;
  v = v < 0.0000001 ? 0.0000001 : v;
  v = 0.99999999 < v ? 0.99999999 : v;
  return v;
}

// =============================================================================================
// flare layer: starburst gather + shaping, irradiance falloff, composition
// =============================================================================================
__device__ inline double convert_coordinate(int p, int length, bool y) {
  // convertCoordinate :933-945 (the float casts are exact for any frame size below 2^24)
  double c = y ? (-((double)(float)p) + ((double)(float)length / 2.0))
               : (((double)(float)p) - ((double)(float)length / 2.0));
  return c >= 0 ? c : (double)length + c;
}

// calculate_irradiance_falloff(x, y, radius) :1043-1063; the 32 draws of pixel p come from the MT19937
// table in the reference's visit order (jitter_mode 0) or from the counter RNG
// EXACT: the reference's own calls -- 1 / pow(r, 1.5) and (raytrace_starburst) pow(factor, 8.0) through the
// double-precision pow -- instead of the cheaper forms below: what MT19937 parity mode runs, so that
// agreement with the reference's frames holds by construction and not by rounding luck (lf_set_flare_arithmetic).
template <bool EXACT>
__device__ inline void irradiance_falloff(const LfFlares* __restrict__ fl, int n_flares, int x, int y,
                                          size_t p, double radius, double dW, double dH,
                                          const uint32_t* __restrict__ jitter_raw, int jitter_mode,
                                          uint64_t key, double out[3]) {
  double t[3] = {0.0, 0.0, 0.0};
  uint4 raw4 = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < 16; s++) {
    // four draws = two samples at a time: one 16-byte load of the pixel's table row, or one Philox block
    if ((s & 1) == 0) {
      if (jitter_mode == 0)
        raw4 = reinterpret_cast<const uint4*>(jitter_raw + p * 32)[s >> 1];
      else
        raw4 = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), (unsigned)(s >> 1),
                                        0x0fa110ffu),
                             make_uint2((unsigned)key, (unsigned)(key >> 32)));
    }
    const unsigned ra = (s & 1) ? raw4.z : raw4.x;
    const unsigned rb = (s & 1) ? raw4.w : raw4.y;
    // Vector2D(random_uniform(), random_uniform()) (sampler.cpp:8-12): g++ evaluates the
    // second argument first, so the first draw is the y jitter
    double sy = (double)y + random_uniform_from_raw(ra);
    double sx = (double)x + random_uniform_from_raw(rb);
    for (int l = 0; l < n_flares; l++) {
      double fx = fl->origin[l][0] * dW, fy = fl->origin[l][1] * dH;
      double ex = fx - sx, ey = fy - sy;
      double nrm = sqrt(ex * ex + ey * ey) - radius;
      double r = 1 + (0.0 < nrm ? nrm : 0.0);
      // 1 / pow(r, 1.5) as rsqrt(r^3): two roundings in r^3 and a reciprocal root within 2 ulp, so
      // ~3 ulp from the exact value (glibc's pow, which the reference calls, is within 1) at a tenth of
      // the instructions of a general double-precision pow followed by a division -- 16 x n_flares of
      // them per pixel made this kernel compute-bound
      double rc = EXACT ? 1.0 / pow(r, 1.5) : rsqrt((r * r) * r);
      t[0] += rc * fl->radiance[l][0];
      t[1] += rc * fl->radiance[l][1];
      t[2] += rc * fl->radiance[l][2];
    }
  }
  const double rc16 = 1.0 / 16.0;   // total / (double)num_samples: Vector3D::operator/ multiplies
  out[0] = rc16 * t[0]; out[1] = rc16 * t[1]; out[2] = rc16 * t[2];
}

// PathTracer::calculate_irradiance_falloff(x, y, radius) on its own
template <bool EXACT>
__global__ void k_one_falloff(const LfFlares* __restrict__ fl, int x, int y, double radius, int W, int H,
                              const uint32_t* __restrict__ jitter_raw, int jitter_mode, uint64_t key,
                              double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  irradiance_falloff<EXACT>(fl, fl->n_flares, x, y, (size_t)x + (size_t)y * W, radius, (double)W, (double)H,
                            jitter_raw, jitter_mode, key, out);
}

template <bool EXACT>
__global__ __launch_bounds__(256) void k_flare_layer(
    const LfFlares* __restrict__ fl, const lf_aperture_stats* __restrict__ st,
    const double* __restrict__ S, const double* __restrict__ ghost,
    const double* __restrict__ scene, const uint32_t* __restrict__ jitter_raw, int jitter_mode,
    uint64_t key, int W, int H, int y0, int y1, LfDeal deal, int ns_aa,
    double flare_radius, double flare_intensity, LfStarSpectrum spec, double* __restrict__ sample,
    double* __restrict__ star_out) {
  const size_t p = (size_t)y0 * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)y1 * W) return;
  const int x = (int)(p % W), y = (int)(p / W);
  // multi-GPU: only the 8-row tile rows (lf_set_row_interleave) or the 64 x 64 blocks (lf_set_block_deal) this context
  // owns; the others are marched, composed and delivered by their owners
  if (!lf_deal_mine(deal, x, y)) return;
  const int n_flares = fl->n_flares;
  const double dW = (double)W, dH = (double)H;

  double star[3] = {0.0, 0.0, 0.0};
  if (n_flares > 0) {  // the reference dereferences flare_origins[0] unconditionally (UB if empty)
    // ---- raytrace_starburst :947-1000 ---------------------------------------------------
    const int aw = st->width;
    const double daw = (double)aw;
    double lr0 = ceil(fl->origin[0][0] * dW), ud0 = ceil(fl->origin[0][1] * dH);  // :921-922
    double lr = lr0 - dW / 2.0, ud = -ud0 + dH / 2.0;                              // :927-928
    double xprime = convert_coordinate(x, W, false), yprime = convert_coordinate(y, H, true);
    // every term's phase is 2*pi*((xc*a + yc*b)/Aw - (a+b)/2) with a, b integers, so the sum's
    // magnitude is |DFT2(A)[b mod Aw][a mod Aw]|
    long long a = (long long)(lr - xprime), b = (long long)(ud - yprime);
    double dx = lr0 - (double)x, dy = ud0 - (double)y;
    double d = sqrt(dx * dx + dy * dy);
    double intensity = -flare_intensity + 3.0;
    if (intensity <= 0) intensity = 2.0;
    if (spec.n == 0) {
      int ka = (int)(((a % aw) + aw) % aw), kb = (int)(((b % aw) + aw) % aw);
      double I = S[(size_t)kb * aw + ka];
      if (d > daw / 2.0) {          // flare suppression :979-985
        double factor = (daw / 2.0) / d;
        // pow(factor, 8.0) by three squarings: within 4 ulp of the exact power (the general pow is the
        // most expensive thing this pixel would otherwise do, and nearly every pixel lies out here)
        const double f2 = factor * factor, f4 = f2 * f2;
        I = (EXACT ? pow(factor, 8.0) : f4 * f4) * I;
      } else if (d <= flare_radius) {  // flare amplification :986-992
        I = pow(I, d / flare_radius);
      }
      double pw = pow(I, intensity);
      for (int l = 0; l < n_flares; l++) {
        star[0] += pw * fl->radiance[l][0];
        star[1] += pw * fl->radiance[l][1];
        star[2] += pw * fl->radiance[l][2];
      }
    } else {
      // row f4 (no reference counterpart): wavelength l reads the pattern magnified by
      // 1 / scale[l] -- S at (a, b) * scale[l], bilinear on the periodic table -- shaped like the
      // reference's value and weighted into R, G, B
      for (int w = 0; w < spec.n; w++) {
        double fa = (double)a * spec.scale[w], fb = (double)b * spec.scale[w];
        double ia = floor(fa), ib = floor(fb);
        double ta = fa - ia, tb = fb - ib;
        long long ka0 = ((((long long)ia) % aw) + aw) % aw, kb0 = ((((long long)ib) % aw) + aw) % aw;
        long long ka1 = (ka0 + 1) % aw, kb1 = (kb0 + 1) % aw;
        double s00 = S[(size_t)kb0 * aw + ka0], s01 = S[(size_t)kb0 * aw + ka1];
        double s10 = S[(size_t)kb1 * aw + ka0], s11 = S[(size_t)kb1 * aw + ka1];
        double I = (1.0 - tb) * ((1.0 - ta) * s00 + ta * s01) + tb * ((1.0 - ta) * s10 + ta * s11);
        if (d > daw / 2.0) {
          const double factor = (daw / 2.0) / d, f2 = factor * factor, f4 = f2 * f2;
          I = (EXACT ? pow(factor, 8.0) : f4 * f4) * I;
        } else if (d <= flare_radius) {
          I = pow(I, d / flare_radius);
        }
        double pw = pow(I, intensity);
        for (int l = 0; l < n_flares; l++)
          for (int c = 0; c < 3; c++) star[c] += (pw * fl->radiance[l][c]) * spec.rgb[w][c];
      }
    }
    // ---- calculate_irradiance_falloff(x, y, 5.0) :1002 -------------------------------------
    double t[3];
    irradiance_falloff<EXACT>(fl, n_flares, x, y, p, 5.0, dW, dH, jitter_raw, jitter_mode, key, t);
    star[0] += t[0]; star[1] += t[1]; star[2] += t[2];  // :1004
  }
  // ---- raytrace_pixel :875-891 ---------------------------------------------------------------
  // scene holds the already averaged radiance (sum / (ns_aa+1), :875); absent = nothing was hit
  for (int c = 0; c < 3; c++) {
    double sc = scene ? scene[3 * p + c] : 0.0 * (1. / (double)(ns_aa + 1));
    sample[3 * p + c] = (sc + ghost[3 * p + c]) + star[c];
    star_out[3 * p + c] = star[c];
  }
}

// =============================================================================================
// tonemap: HDRImageBuffer::toColor + ImageBuffer::update_pixel
// =============================================================================================
__global__ void k_tonemap(const double* __restrict__ sample, int W, int y0, int y1,
                          uint32_t* __restrict__ rgba) {
  const size_t p = (size_t)y0 * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)y1 * W) return;
  const float one_over_gamma = __fdiv_rn(1.0f, 2.2f);
  const float exposure = (float)sqrt(pow(2.0, (double)1.0f));
  uint32_t px = 0xFF000000u;
  for (int k = 0; k < 3; k++) {
    double v = pow(sample[3 * p + k] * (double)exposure, (double)one_over_gamma);
    double mn = (1.0 < v) ? 1.0 : v;    // std::min(v, 1.0)
    double mx = (0.0 < mn) ? mn : 0.0;  // std::max(0.0, mn)
    float c = (float)mx;
    float cl = (c < 1.f) ? c : 1.f;     // clamp(0.f, 1.f, c) == min(max(0,1), c), misc.h:70-72
    px |= ((uint32_t)(cl * 255)) << (8 * k);
  }
  rgba[p] = px;
}

// RaytracedRenderer::save_image's pixel preparation (raytraced_renderer.cpp:739-746): rows flipped
// (PNG is top-down, the sensor buffer bottom-up) and alpha forced to 0xFF
__global__ void k_flip_rows(const uint32_t* __restrict__ in, int W, int H, uint32_t* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)W * H) return;
  const int x = (int)(p % W), y = (int)(p / W);
  out[(size_t)(H - 1 - y) * W + x] = in[p] | 0xFF000000u;
}

}  // namespace

// =============================================================================================
// launchers
// =============================================================================================
lf_status lfk_flip_rows(lf_ctx* ctx, uint32_t* out_dev) {
  size_t n = (size_t)ctx->W * ctx->H;
  hipLaunchKernelGGL(k_flip_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->rgba, ctx->W, ctx->H, out_dev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status lfk_aperture_stats(lf_ctx* ctx, int slot) {
  LfApertureDev& a = ctx->ap[slot];
  lf_aperture_stats init{a.w, a.h, a.w, a.w, -1, -1, 0.0};  // camera.h:54-56
  LF_HIP(ctx, hipMemcpyAsync(a.stats, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  int n = a.w * a.h;
  int blocks = std::min(1024, (n + 255) / 256);
  hipLaunchKernelGGL(k_aperture_stats, dim3(blocks), dim3(256), 0, ctx->stream, a.texels, a.w, a.h,
                     a.stats);
  LF_HIP(ctx, hipGetLastError());
  LF_HIP(ctx, hipMemcpyAsync(&a.host_stats, a.stats, sizeof(init), hipMemcpyDeviceToHost,
                             ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LF_OK;
}

lf_status lfk_build_spectrum(lf_ctx* ctx) {
  LfApertureDev& a = ctx->ap[LF_APERTURE_STARBURST];
  const int aw = a.w;
  const lf_aperture_stats& hs = a.host_stats;
  hipEvent_t ev = lf_timing_begin(ctx, LFK_DFT);
  hipLaunchKernelGGL(k_twiddle, dim3((aw + 255) / 256), dim3(256), 0, ctx->stream, ctx->twiddle, aw);
  if (hs.max_y >= hs.min_y) {
    int rows = hs.max_y - hs.min_y + 1;
    size_t lds = sizeof(double) * 3 * (size_t)aw;
    hipLaunchKernelGGL(k_dft_rows, dim3(rows), dim3(256), lds, ctx->stream, a.texels, a.stats,
                       ctx->twiddle, ctx->dft_rows);
  }
  hipLaunchKernelGGL(k_dft_cols, dim3(aw), dim3(256), 0, ctx->stream, a.stats, ctx->twiddle,
                     ctx->dft_rows, ctx->spectrum);
  lf_timing_end(ctx, LFK_DFT, ev);
  LF_HIP(ctx, hipGetLastError());
  ctx->spectrum_valid = true;
  return LF_OK;
}

lf_status lfk_frame_setup(lf_ctx* ctx, const LfSunLightArgs* lights_arg, const double* lights_dev,
                          int n_lights, bool project) {
  const LfApertureDev& g = ctx->ap[LF_APERTURE_GHOST];
  LfSunLightArgs none;
  if (!lights_arg) { std::memset(&none, 0, sizeof(none)); lights_arg = &none; }
  hipEvent_t ev = lf_timing_begin(ctx, LFK_FRAME_SETUP);
  hipLaunchKernelGGL(k_frame_setup, dim3(1), dim3(64), 0, ctx->stream, ctx->pl_dev, ctx->cam,
                     *lights_arg, lights_dev, n_lights, project ? 1 : 0, ctx->W, ctx->H, g.w, g.h, ctx->flares,
                     ctx->ghosts);
  lf_timing_end(ctx, LFK_FRAME_SETUP, ev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status lfk_ghost_raster(lf_ctx* ctx) {
  if (ctx->y1 <= ctx->y0) return LF_OK;  // empty band
  const LfApertureDev& g = ctx->ap[LF_APERTURE_GHOST];
  dim3 grid((ctx->W + kTileW - 1) / kTileW, (ctx->y1 - ctx->y0 + kTileH - 1) / kTileH);
  hipEvent_t ev = lf_timing_begin(ctx, LFK_GHOST_RASTER);
  hipLaunchKernelGGL(k_ghost_raster, grid, dim3(256), 0, ctx->stream, ctx->ghosts, g.texels, g.w,
                     g.h, ctx->W, ctx->y0, ctx->y1, 0, ctx->ghost);
  lf_timing_end(ctx, LFK_GHOST_RASTER, ev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

// the one or two triangles a single-shot set-up kernel left in ctx->ghosts, added on top of the
// ghost buffer; bbox (may be null) receives the pixel rectangle [x0, x1) x [y0, y1) they can touch
static lf_status raster_list_additive(lf_ctx* ctx, int bbox[4]) {
  struct { int n_tris, pad; LfGhostTri tri[2]; } head;
  LF_HIP(ctx, hipMemcpyAsync(&head, ctx->ghosts, sizeof(head), hipMemcpyDeviceToHost, ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int x0 = ctx->W, y0 = ctx->H, x1 = 0, y1 = 0;
  for (int t = 0; t < head.n_tris && t < 2; t++) {
    x0 = std::min(x0, head.tri[t].bx0); x1 = std::max(x1, head.tri[t].bx1);
    y0 = std::min(y0, head.tri[t].by0); y1 = std::max(y1, head.tri[t].by1);
  }
  if (x1 <= x0 || y1 <= y0) { x0 = x1 = y0 = y1 = 0; }
  if (bbox) { bbox[0] = x0; bbox[1] = y0; bbox[2] = x1; bbox[3] = y1; }
  if (y1 > y0) {
    const LfApertureDev& g = ctx->ap[LF_APERTURE_GHOST];
    dim3 grid((ctx->W + kTileW - 1) / kTileW, (y1 - y0 + kTileH - 1) / kTileH);
    hipLaunchKernelGGL(k_ghost_raster, grid, dim3(256), 0, ctx->stream, ctx->ghosts, g.texels, g.w, g.h,
                       ctx->W, y0, y1, 1, ctx->ghost);
    LF_HIP(ctx, hipGetLastError());
  }
  return LF_OK;
}

lf_status lfk_draw_ghost(lf_ctx* ctx, int channel, float r1, float r2, int bbox[4]) {
  const LfApertureDev& g = ctx->ap[LF_APERTURE_GHOST];
  hipLaunchKernelGGL(k_one_ghost, dim3(1), dim3(64), 0, ctx->stream, ctx->flares, r1, r2, channel, ctx->W,
                     ctx->H, g.w, g.h, ctx->ghosts);
  LF_HIP(ctx, hipGetLastError());
  return raster_list_additive(ctx, bbox);
}

lf_status lfk_raster_triangle(lf_ctx* ctx, const float v[12], const double colour[3], int bbox[4]) {
  LfTriArgs a;
  std::memcpy(a.v, v, sizeof(a.v));
  std::memcpy(a.colour, colour, sizeof(a.colour));
  hipLaunchKernelGGL(k_one_triangle, dim3(1), dim3(64), 0, ctx->stream, a, ctx->W, ctx->H, ctx->ghosts);
  LF_HIP(ctx, hipGetLastError());
  return raster_list_additive(ctx, bbox);
}

lf_status lfk_fill_pixel(lf_ctx* ctx, const float v[12], int x, int y, const double colour[3]) {
  LfTriArgs a;
  std::memcpy(a.v, v, sizeof(a.v));
  std::memcpy(a.colour, colour, sizeof(a.colour));
  const LfApertureDev& g = ctx->ap[LF_APERTURE_GHOST];
  hipLaunchKernelGGL(k_one_pixel, dim3(1), dim3(64), 0, ctx->stream, a, x, y, g.texels, g.w, g.h, ctx->W,
                     ctx->ghost);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

// the scalar-valued helper members: one wave writes `n_out` (<= 8) doubles, which come straight back
template <typename Launch>
static lf_status probe_doubles(lf_ctx* ctx, int n_out, double* out, Launch launch) {
  if (!ctx->probe_dev) LF_HIP(ctx, hipMalloc((void**)&ctx->probe_dev, sizeof(double) * 8));
  launch(ctx->probe_dev);
  LF_HIP(ctx, hipGetLastError());
  LF_HIP(ctx, hipMemcpyAsync(out, ctx->probe_dev, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LF_OK;
}

lf_status lfk_shift_vertex(lf_ctx* ctx, float x, float y, float scale, float shift_amount, double out[2]) {
  return probe_doubles(ctx, 2, out, [&](double* d) {
    hipLaunchKernelGGL(k_shift_vertex, dim3(1), dim3(64), 0, ctx->stream, ctx->flares, x, y, scale,
                       shift_amount, d);
  });
}

lf_status lfk_compute_phase(lf_ctx* ctx, int flare, double u, double v, double out[4]) {
  return probe_doubles(ctx, 4, out, [&](double* d) {
    hipLaunchKernelGGL(k_compute_phase, dim3(1), dim3(64), 0, ctx->stream, ctx->flares, flare, u, v, ctx->W,
                       ctx->H, d);
  });
}

lf_status lfk_irradiance_falloff(lf_ctx* ctx, int x, int y, double radius, double out[3]) {
  return probe_doubles(ctx, 3, out, [&](double* d) {
    if (lf_flare_exact(ctx))
      hipLaunchKernelGGL(k_one_falloff<true>, dim3(1), dim3(64), 0, ctx->stream, ctx->flares, x, y, radius, ctx->W,
                         ctx->H, ctx->jitter_raw, ctx->jitter_mode, ctx->jitter_key, d);
    else
      hipLaunchKernelGGL(k_one_falloff<false>, dim3(1), dim3(64), 0, ctx->stream, ctx->flares, x, y, radius, ctx->W,
                         ctx->H, ctx->jitter_raw, ctx->jitter_mode, ctx->jitter_key, d);
  });
}

lf_status lfk_flare_layer(lf_ctx* ctx) {
  size_t n = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
  if (n == 0) return LF_OK;  // empty band: nothing to render (a 0-block launch is an error)
  hipEvent_t ev = lf_timing_begin(ctx, LFK_FLARE_LAYER);
#define LF_LAUNCH_FLARE(EXACT)                                                                      \
  hipLaunchKernelGGL(k_flare_layer<EXACT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, \
                     ctx->flares, ctx->ap[LF_APERTURE_STARBURST].stats, ctx->spectrum, ctx->ghost,     \
                     ctx->scene, ctx->jitter_raw, ctx->jitter_mode, ctx->jitter_key, ctx->W, ctx->H,   \
                     ctx->y0, ctx->y1, lf_deal_of(ctx), ctx->ns_aa, ctx->flare_radius,                   \
                     ctx->flare_intensity, ctx->star_spec, ctx->sample, ctx->star)
  if (lf_flare_exact(ctx)) LF_LAUNCH_FLARE(true); else LF_LAUNCH_FLARE(false);
#undef LF_LAUNCH_FLARE
  lf_timing_end(ctx, LFK_FLARE_LAYER, ev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status lfk_tonemap(lf_ctx* ctx, int ya, int yb) {
  if (yb <= ya) return LF_OK;
  size_t n = (size_t)(yb - ya) * ctx->W;
  hipEvent_t ev = lf_timing_begin(ctx, LFK_TONEMAP);
  hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->sample, ctx->W, ya, yb, ctx->rgba);
  lf_timing_end(ctx, LFK_TONEMAP, ev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}
