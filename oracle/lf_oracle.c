/* TEST INFRASTRUCTURE -- the parity oracle (see lf_oracle.h).  Plain C11, built with
 * `gcc -O2 -mavx2 -ffp-contract=off` so that no multiply-add is fused: the reference is built
 * -O3 -mavx2 WITHOUT -mfma (CGL/CMakeLists.txt:45-47, CGL/find_avx.cmake:85-89).
 *
 * All `file:line` citations are relative to the reference checkout (/root/reference).
 * The float/double mixing below is deliberate and follows the reference expression by
 * expression; do not "clean it up".
 */
#define _GNU_SOURCE
#include "lf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * CGL::Matrix3x3 / Vector3D semantics (CGL/src/matrix3x3.cpp:10-16, :99-114;
 * CGL/include/CGL/vector3D.h:99-150): column-major storage, mat*vec evaluated as
 * x.x*col0 + x.y*col1 + x.z*col2, left to right, in double.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double x, y, z; } v3;
typedef struct { v3 c[3]; } m3; /* c[j] = column j; entry(i,j) = component i of c[j] */

static v3 v3_scale(double s, v3 v) { v3 r = {s * v.x, s * v.y, s * v.z}; return r; }
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }

static m3 m3_rows(double m00, double m01, double m02, double m10, double m11, double m12,
                  double m20, double m21, double m22) {
  m3 m;
  m.c[0].x = m00; m.c[1].x = m01; m.c[2].x = m02;
  m.c[0].y = m10; m.c[1].y = m11; m.c[2].y = m12;
  m.c[0].z = m20; m.c[1].z = m21; m.c[2].z = m22;
  return m;
}
static v3 m3_mulv(const m3* A, v3 x) {
  return v3_add(v3_add(v3_scale(x.x, A->c[0]), v3_scale(x.y, A->c[1])), v3_scale(x.z, A->c[2]));
}
static m3 m3_mul(const m3* A, const m3* B) {
  m3 C;
  C.c[0] = m3_mulv(A, B->c[0]);
  C.c[1] = m3_mulv(A, B->c[1]);
  C.c[2] = m3_mulv(A, B->c[2]);
  return C;
}
static m3 m3_scale(double s, const m3* A) { /* matrix3x3.cpp:88-97 */
  m3 C;
  C.c[0] = v3_scale(s, A->c[0]);
  C.c[1] = v3_scale(s, A->c[1]);
  C.c[2] = v3_scale(s, A->c[2]);
  return C;
}

/* ------------------------------------------------------------------------------------------
 * lens table + paraxial operators (pathtracer.cpp:511-586)
 * ---------------------------------------------------------------------------------------- */
void lfo_default_lens(lfo_paraxial_lens* L) {
  /* pathtracer.cpp:541-556.  Every literal is narrowed to float exactly where the reference
   * narrows it: T(7.700) takes a float parameter; the curvature initialisers are double
   * divisions stored into a float array. */
  static const double th[9] = {7.700, 1.850, 3.520, 1.850, 4.180, 3.000, 1.850, 7.270, 83.91};
  static const double red[9] = {1.652, 1.5991, 1, 1.6396, 1, 1, 1.5776, 1.68990, 1};
  static const double green[9] = {1.652, 1.6113, 1, 1.65, 1, 1, 1.5885, 1.6999, 1};
  static const double blue[9] = {1.652, 1.6164, 1, 1.6542, 1, 1, 1.5930, 1.7040, 1};
  memset(L, 0, sizeof(*L));
  L->n = 9;
  L->stop = 5;
  for (int k = 0; k < 9; k++) {
    L->thickness[k] = (float)th[k];
    L->ior[0][k] = (float)red[k];
    L->ior[1][k] = (float)green[k];
    L->ior[2][k] = (float)blue[k];
  }
  L->curvature[0] = (float)(1 / 30.810);
  L->curvature[1] = (float)(1 / -89.350);
  L->curvature[2] = (float)(1 / 580.380);
  L->curvature[3] = (float)(1 / -80.630);
  L->curvature[4] = (float)(1 / 28.340);
  L->curvature[5] = 0;
  L->curvature[6] = 0;
  L->curvature[7] = (float)(1 / 32.190);
  L->curvature[8] = (float)(1 / -52.990);
  L->clip = 11.6;
  L->recast_pos = 11.6f;
  L->recast_neg = -11.5f;
  L->marginal = 14.5f;
}

static m3 make_2_matrix(float a, float b, float c, float d) { /* :511-516 */
  return m3_rows(a, b, 0, c, d, 0, 0, 0, 0);
}
static m3 invert2x2(const m3* mat) { /* :519-525 -- entries read back as float, det in float */
  float a = (float)mat->c[0].x;
  float b = (float)mat->c[1].x;
  float c = (float)mat->c[0].y;
  float d = (float)mat->c[1].y;
  float det = a * d - b * c;
  m3 adj = make_2_matrix(d, -b, -c, a);
  return m3_scale(1.0 / det, &adj);
}
static m3 mat_T(float d) { return make_2_matrix(1, d, 0, 1); }               /* :527-529 */
static m3 mat_R(float c, float n1, float n2) {                               /* :531-533 */
  return make_2_matrix(1, 0, c * (n1 - n2) / n2, n1 / n2);
}
static m3 mat_L(float c) { return make_2_matrix(1, 0, 2 * c, 1); }          /* :535-537 */

typedef struct { m3 Ts[LFO_MAX_SURF], Rs[LFO_MAX_SURF], Ls[LFO_MAX_SURF]; } lens_mats;

static void build_mats(const lfo_paraxial_lens* L, int colour, lens_mats* M) {
  float prev_n = 1.00f; /* create_Rs_for_color :559-568 */
  for (int k = 0; k < L->n; k++) {
    M->Ts[k] = mat_T(L->thickness[k]);
    M->Rs[k] = mat_R(L->curvature[k], prev_n, L->ior[colour][k]);
    prev_n = L->ior[colour][k];
    M->Ls[k] = mat_L(L->curvature[k]); /* create_Ls :570-576 */
  }
}

/* the aperture clip shared by both tracers (:619-629, :654-664) */
static void clip_at_stop(const lfo_paraxial_lens* L, const m3* M, float r, float theta, v3* ray) {
  v3 after_ap = m3_mulv(M, *ray);
  if (after_ap.x > L->clip || after_ap.x < -L->clip) {
    float r_a = L->recast_pos;
    if (r < 0) r_a = L->recast_neg;
    float r_e = (float)((r_a - M->c[1].x * theta) / M->c[0].x);
    ray->x = r_e;
    ray->y = theta;
    ray->z = 0;
  }
}

void lfo_trace_ray_auto_before(const lfo_paraxial_lens* L, float r, float theta, int i, int j,
                               int colour, double out[2]) {
  /* pathtracer.cpp:588-641 */
  lens_mats S;
  build_mats(L, colour, &S);
  v3 ray = {r, theta, 0};
  int mini = i < j ? i : j, maxj = i < j ? j : i;
  i = mini; j = maxj;
  m3 M = make_2_matrix(1, 0, 0, 1), t;
  for (int k = 0; k < j; k++) { t = m3_mul(&S.Ts[k], &S.Rs[k]); M = m3_mul(&t, &M); }
  M = m3_mul(&S.Ls[j], &M);
  for (int k = j - 1; k > i; k--) {
    m3 inv = invert2x2(&S.Rs[k]);
    t = m3_mul(&inv, &S.Ts[k]);
    M = m3_mul(&t, &M);
  }
  {
    m3 inv = invert2x2(&S.Ls[i]);
    t = m3_mul(&S.Ts[i], &inv);
    t = m3_mul(&t, &S.Ts[i]);
    M = m3_mul(&t, &M);
  }
  for (int k = i + 1; k < L->n; k++) {
    if (k == L->stop) {
      clip_at_stop(L, &M, r, theta, &ray);
      M = m3_mul(&S.Ts[k], &M);
      continue;
    }
    t = m3_mul(&S.Ts[k], &S.Rs[k]);
    M = m3_mul(&t, &M);
  }
  v3 res = m3_mulv(&M, ray);
  out[0] = res.x;
  out[1] = res.y;
}

void lfo_trace_ray_auto_after(const lfo_paraxial_lens* L, float r, float theta, int i, int j,
                              int colour, double out[2]) {
  /* pathtracer.cpp:643-689 */
  lens_mats S;
  build_mats(L, colour, &S);
  v3 ray = {r, theta, 0};
  int mini = i < j ? i : j, maxj = i < j ? j : i;
  i = mini; j = maxj;
  m3 M = make_2_matrix(1, 0, 0, 1), t;
  for (int k = 0; k < j; k++) {
    if (k == L->stop) {
      clip_at_stop(L, &M, r, theta, &ray);
      M = m3_mul(&S.Ts[k], &M);
      continue;
    }
    t = m3_mul(&S.Ts[k], &S.Rs[k]);
    M = m3_mul(&t, &M);
  }
  M = m3_mul(&S.Ls[j], &M);
  for (int k = j - 1; k > i; k--) {
    m3 inv = invert2x2(&S.Rs[k]);
    t = m3_mul(&inv, &S.Ts[k]);
    M = m3_mul(&t, &M);
  }
  {
    m3 inv = invert2x2(&S.Ls[i]);
    t = m3_mul(&S.Ts[i], &inv);
    t = m3_mul(&t, &S.Ts[i]);
    M = m3_mul(&t, &M);
  }
  for (int k = i + 1; k < L->n; k++) { t = m3_mul(&S.Ts[k], &S.Rs[k]); M = m3_mul(&t, &M); }
  v3 res = m3_mulv(&M, ray);
  out[0] = res.x;
  out[1] = res.y;
}

/* ------------------------------------------------------------------------------------------
 * aperture texture (camera.h:26-83, CGL/src/color.cpp:16-21)
 * ---------------------------------------------------------------------------------------- */
void lfo_aperture_stats_from_texels(const float* texels, int w, int h, lfo_aperture_stats* st) {
  st->width = w; st->height = h;
  st->min_x = st->min_y = w; /* camera.h:54 (both initialised to width) */
  st->max_x = st->max_y = -1;
  st->total_value = 0.0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float r = texels[(size_t)y * w + x];
      st->total_value += r;
      if (r > 0) {
        if (x < st->min_x) st->min_x = x;
        if (y < st->min_y) st->min_y = y;
        if (x > st->max_x) st->max_x = x;
        if (y > st->max_y) st->max_y = y;
      }
    }
}
void lfo_aperture_from_red(const uint8_t* red, int w, int h, float* texels,
                           lfo_aperture_stats* st) {
  float inv = (float)(1.0 / 255.0); /* color.cpp:17 */
  for (size_t i = 0; i < (size_t)w * h; i++) texels[i] = red[i] * inv;
  lfo_aperture_stats_from_texels(texels, w, h, st);
}

/* ------------------------------------------------------------------------------------------
 * sun detection (camera.cpp:245-273, pathtracer.cpp:32-64)
 * ---------------------------------------------------------------------------------------- */
void lfo_analyze_world_coord(const double c2w[9], const double cam_pos[3], double hFov,
                             double vFov, const double pw[3], double* ns_x, double* ns_y) {
  const double PI_ = 3.14159265358979323; /* CGL/include/CGL/misc.h:11 */
  double hFOV_rads = hFov * (PI_ / 180.0);
  double vFOV_rads = vFov * (PI_ / 180.0);
  double edge_x = tan(0.5 * hFOV_rads);
  double edge_y = tan(0.5 * vFOV_rads);
  /* c2w.T() * (pos_world - pos): c2w(i,j) row-major in the argument */
  m3 c2wT = m3_rows(c2w[0], c2w[3], c2w[6], c2w[1], c2w[4], c2w[7], c2w[2], c2w[5], c2w[8]);
  v3 d = {pw[0] - cam_pos[0], pw[1] - cam_pos[1], pw[2] - cam_pos[2]};
  v3 pc = m3_mulv(&c2wT, d);
  double rc = 1.0 / fabs(pc.z); /* Vector3D::operator/(double) multiplies by 1/c (vector3D.h:138-141) */
  double ix = rc * pc.x, iy = rc * pc.y;
  *ns_x = ((ix / edge_x) + 1) / 2.0;
  *ns_y = ((iy / edge_y) + 1) / 2.0;
}

void lfo_find_sun_pos(const double c2w[9], const double cam_pos[3], double hFov, double vFov,
                      const double* lights, int n_lights, lfo_frame* f) {
  f->n_flares = 0;
  for (int l = 0; l < n_lights; l++) {
    double nx, ny;
    lfo_analyze_world_coord(c2w, cam_pos, hFov, vFov, lights + 6 * l, &nx, &ny);
    if ((nx >= 0 && nx <= 1) && (ny >= 0 && ny <= 1) && f->n_flares < 8) {
      int k = f->n_flares++;
      f->flare_origin[k][0] = nx;
      f->flare_origin[k][1] = ny;
      memcpy(f->flare_radiance[k], lights + 6 * l + 3, 3 * sizeof(double));
      f->angle_to_sun = (float)atan(ny / nx); /* :50, float member pathtracer.h:135 */
      f->axis_ray[0] = nx;
      f->axis_ray[1] = ny;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * ghost quads (pathtracer.cpp:305-508)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float* tex; int tex_w, tex_h;
  double* ghost; int W, H;
} ghost_ctx;

static void swapf(float* a, float* b) { float t = *a; *a = *b; *b = t; }

static void fill_textured_pixel(const ghost_ctx* g, float x0, float y0, float u0, float v0,
                                float x1, float y1, float u1, float v1, float x2, float y2,
                                float u2, float v2, int x, int y, const double color[3]) {
  /* :305-343, all float */
  float xy_to_01 = -(y1 - y0) * (x - x0) + (x1 - x0) * (y - y0);
  float two_to_01 = -(y1 - y0) * (x2 - x0) + (x1 - x0) * (y2 - y0);
  float alpha = xy_to_01 / two_to_01;
  float xy_to_12 = -(y2 - y1) * (x - x1) + (x2 - x1) * (y - y1);
  float zero_to_12 = -(y2 - y1) * (x0 - x1) + (x2 - x1) * (y0 - y1);
  float beta = xy_to_12 / zero_to_12;
  float gamma = 1 - alpha - beta;
  if (gamma >= 0 && alpha >= 0 && beta >= 0) {
    float u = u2 * alpha + u0 * beta + u1 * gamma;
    float v = v2 * alpha + v0 * beta + v1 * gamma;
    double uvx = u, uvy = v;
    /* :338 -- index computed in double, truncated; u is NOT floored on its own */
    int idx = (int)(floor(uvy) * (double)(size_t)g->tex_w + uvx);
    /* the reference indexes the vector unchecked; v == tex_h reads past the end (UB).
     * We define that read as 0. */
    float sample = (idx >= 0 && idx < g->tex_w * g->tex_h) ? g->tex[idx] : 0.0f;
    double s = sample;
    double* px = g->ghost + 3 * ((size_t)x + (size_t)y * g->W);
    px[0] += s * color[0];
    px[1] += s * color[1];
    px[2] += s * color[2];
  }
}

static void rasterize_textured_triangle(const ghost_ctx* g, float x0, float y0, float u0,
                                        float v0, float x1, float y1, float u1, float v1,
                                        float x2, float y2, float u2, float v2,
                                        const double color[3]) {
  /* :346-410 */
  if (y1 < y0) { swapf(&x0, &x1); swapf(&y0, &y1); swapf(&u0, &u1); swapf(&v0, &v1); }
  if (y2 < y0) { swapf(&x0, &x2); swapf(&y0, &y2); swapf(&u0, &u2); swapf(&v0, &v2); }
  if (y2 < y1) { swapf(&x1, &x2); swapf(&y1, &y2); swapf(&u1, &u2); swapf(&v1, &v2); }
  x0 -= 0.5; y0 -= 0.5; x1 -= 0.5; y1 -= 0.5; x2 -= 0.5; y2 -= 0.5;
  float mn = fminf(fminf(x0, x1), x2), mx = fmaxf(fmaxf(x0, x1), x2);
  int bx0 = (int)floorf(mn); if (bx0 < 0) bx0 = 0;
  int bx1 = (int)ceilf(mx); if (bx1 > g->W - 1) bx1 = g->W - 1;
  int by0 = (int)floorf(y0); if (by0 < 0) by0 = 0;
  int by1 = (int)ceilf(y2); if (by1 > g->H - 1) by1 = g->H - 1;
  float min_x = bx0, max_x = bx1, min_y = by0, max_y = by1; /* stored as float in the reference */
  for (int y = (int)min_y; y < max_y; y++)
    for (int x = (int)min_x; x < max_x; x++)
      fill_textured_pixel(g, x0, y0, u0, v0, x1, y1, u1, v1, x2, y2, u2, v2, x, y, color);
}

static void shift_vertex(const lfo_frame* f, float x, float y, float scale, float shift_amount,
                         double out[2]) {
  /* :412-430; cos/sin are the float overloads (camera.h:14 `using namespace std`) */
  v3 v = {x, y, 1};
  float ang = (float)atan((f->axis_ray[1] - 0.5) / (f->axis_ray[0] - 0.5));
  m3 scaling = m3_rows(scale, 0, 0, 0, scale, 0, 0, 0, 1);
  m3 rotation = m3_rows(cosf(ang), -sinf(ang), 0, sinf(ang), cosf(ang), 0, 0, 0, 1);
  m3 shift = m3_rows(1, 0, shift_amount * cosf(ang), 0, 1, shift_amount * sinf(ang), 0, 0, 1);
  m3 t = m3_mul(&shift, &rotation);
  t = m3_mul(&t, &scaling);
  v3 r = m3_mulv(&t, v);
  out[0] = r.x;
  out[1] = r.y;
}

static void draw_ghost(const ghost_ctx* g, const lfo_frame* f, int colour, float r1, float r2) {
  /* :433-508 */
  float shift_amt = (float)(-(r1 + r2) / 2 * 0.4);
  float scale_amt = (float)(fabsf(r2 - r1) * 0.2);
  double gb_mid_w = ceil(f->axis_ray[0] * (double)g->W);
  double gb_mid_h = ceil(f->axis_ray[1] * (double)g->H);
  double ul[2], ll[2], ur[2], lr[2];
  shift_vertex(f, -1, 1, scale_amt, shift_amt, ul);
  shift_vertex(f, -1, -1, scale_amt, shift_amt, ll);
  shift_vertex(f, 1, 1, scale_amt, shift_amt, ur);
  shift_vertex(f, 1, -1, scale_amt, shift_amt, lr);
  double color[3] = {0, 0, 0};
  float intensity_scalar = 10;
  float size_scalar = 1 / (scale_amt * scale_amt);
  color[colour] = 1.0;
  double k = intensity_scalar * size_scalar; /* float product widened (:494) */
  color[0] *= k; color[1] *= k; color[2] *= k;
  float th = (float)(size_t)g->tex_h, tw = (float)(size_t)g->tex_w;
  rasterize_textured_triangle(g, (float)(gb_mid_w + ul[0]), (float)(gb_mid_h + ul[1]), 0, 0,
                              (float)(gb_mid_w + ll[0]), (float)(gb_mid_h + ll[1]), 0, th,
                              (float)(gb_mid_w + ur[0]), (float)(gb_mid_h + ur[1]), tw, 0, color);
  /* :498 -- the second triangle reuses the first one's UVs */
  rasterize_textured_triangle(g, (float)(gb_mid_w + lr[0]), (float)(gb_mid_h + lr[1]), 0, 0,
                              (float)(gb_mid_w + ll[0]), (float)(gb_mid_h + ll[1]), 0, th,
                              (float)(gb_mid_w + ur[0]), (float)(gb_mid_h + ur[1]), tw, 0, color);
}

void lfo_generate_ghost_buffer(const lfo_paraxial_lens* L, const lfo_frame* f,
                               const float* ghost_tex, int tex_w, int tex_h, double* ghost) {
  /* :714-762 */
  ghost_ctx g = {ghost_tex, tex_w, tex_h, ghost, f->W, f->H};
  memset(ghost, 0, sizeof(double) * 3 * (size_t)f->W * f->H);
  if (f->axis_ray[0] == 0 && f->axis_ray[1] == 0) return;
  double s1[2], s2[2];
  for (int i = 0; i < L->stop; i++)
    for (int j = i + 1; j < L->stop; j++)
      for (int c = 0; c < 3; c++) {
        lfo_trace_ray_auto_before(L, L->marginal, f->angle_to_sun, i, j, c, s1);
        lfo_trace_ray_auto_before(L, -L->marginal, f->angle_to_sun, i, j, c, s2);
        draw_ghost(&g, f, c, (float)s1[0], (float)s2[0]);
      }
  for (int i = L->stop + 1; i < L->n; i++)
    for (int j = i + 1; j < L->n; j++)
      for (int c = 0; c < 3; c++) {
        lfo_trace_ray_auto_after(L, L->marginal, f->angle_to_sun, i, j, c, s1);
        lfo_trace_ray_auto_after(L, -L->marginal, f->angle_to_sun, i, j, c, s2);
        draw_ghost(&g, f, c, (float)s1[0], (float)s2[0]);
      }
}

/* ------------------------------------------------------------------------------------------
 * RNG (util/random_util.h:10-22): std::mt19937 with the default seed, 32-bit outputs
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t s[624]; int idx; } mt_state;
static void mt_seed(mt_state* m, uint32_t seed) {
  m->s[0] = seed;
  for (int i = 1; i < 624; i++) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
  m->idx = 624;
}
static uint32_t mt_next(mt_state* m) {
  if (m->idx >= 624) {
    for (int k = 0; k < 624; k++) {
      uint32_t y = (m->s[k] & 0x80000000u) | (m->s[(k + 1) % 624] & 0x7fffffffu);
      m->s[k] = m->s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    m->idx = 0;
  }
  uint32_t y = m->s[m->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
void lfo_mt19937_raw(uint32_t seed, size_t skip, size_t n, uint32_t* out) {
  mt_state m;
  mt_seed(&m, seed);
  for (size_t i = 0; i < skip; i++) (void)mt_next(&m);
  for (size_t i = 0; i < n; i++) out[i] = mt_next(&m);
}
double lfo_random_uniform_from_raw(uint32_t raw) {
  double rmax = 1.0 / (4294967295.0 - 0.0);
  double v = (double)raw * rmax;
  /* clamp(x, lo, hi) = min(max(x, lo), hi) (CGL/include/CGL/misc.h:70-72) */
  v = v > 0.0000001 ? v : 0.0000001;
  v = v < 0.99999999 ? v : 0.99999999;
  return v;
}

/* ------------------------------------------------------------------------------------------
 * starburst (pathtracer.cpp:901-1004)
 * ---------------------------------------------------------------------------------------- */
double lfo_convert_coordinate(size_t pixel_coord, int length, int y) {
  double coord_center; /* :933-945 */
  if (y) coord_center = -((float)pixel_coord) + ((float)length / 2.0);
  else   coord_center = ((float)pixel_coord) - ((float)length / 2.0);
  if (coord_center >= 0) return coord_center;
  return length + coord_center;
}

static void complex_exp(double exponent, int negative, double* re, double* im) { /* :901-915 */
  double c = cos(2.0 * M_PI * exponent);
  double s = sin(2.0 * M_PI * exponent);
  if (negative) s *= -1.0;
  *re = c; *im = s;
}

void lfo_starburst_pixel(const lfo_frame* f, const float* ap, const lfo_aperture_stats* st,
                         size_t x, size_t y, double rgb[3], double* abs_avg_unshaped) {
  double W = (double)(size_t)f->W, H = (double)(size_t)f->H;
  double xprime = lfo_convert_coordinate(x, f->W, 0);
  double yprime = lfo_convert_coordinate(y, f->H, 1);
  double ci_re = 0, ci_im = 0;
  /* compute_phase(0, ...) :917-931 -- flare index 0 only */
  double lr0 = ceil(f->flare_origin[0][0] * W);
  double ud0 = ceil(f->flare_origin[0][1] * H);
  double lr = lr0 - W / 2.0;
  double ud = -ud0 + H / 2.0;
  double aw = (double)(size_t)st->width;
  for (int yc = st->min_y; yc <= st->max_y; yc++)
    for (int xc = st->min_x; xc <= st->max_x; xc++) {
      double sampled_value = (double)ap[(size_t)yc * st->width + xc];
      double u = ((double)xc / aw) - 0.5;
      double v = ((double)yc / aw) - 0.5; /* :963 divides by width, not height */
      double exponent = u * xprime + v * yprime;
      double e_re, e_im, p_re, p_im;
      complex_exp(exponent, 1, &e_re, &e_im);
      complex_exp(u * lr + v * ud, 0, &p_re, &p_im);
      /* (sampled_value * additional_phase) * complex_exponential, std::complex<double> (:970) */
      double a = p_re * sampled_value, b = p_im * sampled_value;
      double t_re = a * e_re - b * e_im;
      double t_im = a * e_im + b * e_re;
      ci_re += t_re;
      ci_im += t_im;
    }
  double I = hypot(ci_re, ci_im) / st->total_value; /* std::abs(complex) = cabs = hypot */
  if (abs_avg_unshaped) *abs_avg_unshaped = I;
  double radius = f->flare_radius;
  double dx = lr0 - (double)x, dy = ud0 - (double)y;
  double d = sqrt(dx * dx + dy * dy);
  if (d > aw / 2.0) {
    double factor = (aw / 2.0) / d;
    I = pow(factor, 8.0) * I;
  } else if (d <= radius) {
    double factor = d / radius;
    I = pow(I, factor);
  }
  double intensity = -f->flare_intensity + 3.0;
  if (intensity <= 0) intensity = 2.0;
  rgb[0] = rgb[1] = rgb[2] = 0;
  for (int l = 0; l < f->n_flares; l++) {
    double p = pow(I, intensity);
    rgb[0] += p * f->flare_radiance[l][0];
    rgb[1] += p * f->flare_radiance[l][1];
    rgb[2] += p * f->flare_radiance[l][2];
  }
}

/* ---- spectral starburst -- PARITY UNPINNED -------------------------
 * The reference's starburst is monochrome.  The Fraunhofer pattern of the aperture scales with the
 * wavelength, so wavelength l sees the pattern of the reference magnified by 1/scale[l]
 * (scale = lambda_ref / lambda_l): with a = lr - x', b = ud - y' (integers)
 * the reference reads S[b mod Aw][a mod Aw], S = |DFT2(aperture)| / total_value; wavelength l reads
 * S at (a scale, b scale), bilinearly interpolated on the periodic table, is shaped exactly like
 * the reference's value (:979-1000) and is added with its RGB weight.  With one wavelength,
 * scale 1 and weight (1,1,1) this IS the reference formula, which pins the spectral form at that
 * point (tests/test_oracle_vs_reference.py); everything else about it is this repository's
 * specification. */
static double spectrum_at(const float* ap, const lfo_aperture_stats* st, long long ka, long long kb) {
  const long long aw = st->width;
  double re = 0, im = 0;
  for (int yc = st->min_y; yc <= st->max_y; yc++)
    for (int xc = st->min_x; xc <= st->max_x; xc++) {
      double v = (double)ap[(size_t)yc * st->width + xc];
      long long k = ((long long)xc * ka + (long long)yc * kb) % aw; /* exact phase index */
      double ph = 2.0 * M_PI * (double)k / (double)aw;
      re += v * cos(ph);
      im += v * sin(ph);
    }
  return hypot(re, im) / st->total_value;
}

void lfo_starburst_pixel_spectral(const lfo_frame* f, const float* ap, const lfo_aperture_stats* st,
                                  size_t x, size_t y, int n_lambda, const double* scale,
                                  const double* rgb_w, double rgb[3]) {
  double W = (double)(size_t)f->W, H = (double)(size_t)f->H;
  double xprime = lfo_convert_coordinate(x, f->W, 0);
  double yprime = lfo_convert_coordinate(y, f->H, 1);
  double lr0 = ceil(f->flare_origin[0][0] * W), ud0 = ceil(f->flare_origin[0][1] * H);
  double lr = lr0 - W / 2.0, ud = -ud0 + H / 2.0;
  long long a = (long long)(lr - xprime), b = (long long)(ud - yprime);
  const long long aw = st->width;
  double daw = (double)aw;
  double dx = lr0 - (double)x, dy = ud0 - (double)y;
  double d = sqrt(dx * dx + dy * dy);
  double intensity = -f->flare_intensity + 3.0;
  if (intensity <= 0) intensity = 2.0;
  rgb[0] = rgb[1] = rgb[2] = 0;
  for (int l = 0; l < n_lambda; l++) {
    double fa = (double)a * scale[l], fb = (double)b * scale[l];
    double ia = floor(fa), ib = floor(fb);
    double ta = fa - ia, tb = fb - ib;
    long long ka0 = (((long long)ia % aw) + aw) % aw, kb0 = (((long long)ib % aw) + aw) % aw;
    long long ka1 = (ka0 + 1) % aw, kb1 = (kb0 + 1) % aw;
    double s00 = spectrum_at(ap, st, ka0, kb0), s01 = spectrum_at(ap, st, ka1, kb0);
    double s10 = spectrum_at(ap, st, ka0, kb1), s11 = spectrum_at(ap, st, ka1, kb1);
    double I = (1.0 - tb) * ((1.0 - ta) * s00 + ta * s01) + tb * ((1.0 - ta) * s10 + ta * s11);
    if (d > daw / 2.0) {
      I = pow((daw / 2.0) / d, 8.0) * I;
    } else if (d <= f->flare_radius) {
      I = pow(I, d / f->flare_radius);
    }
    double pw = pow(I, intensity);
    for (int k = 0; k < f->n_flares; k++)
      for (int c = 0; c < 3; c++) rgb[c] += (pw * f->flare_radiance[k][c]) * rgb_w[3 * l + c];
  }
}

void lfo_irradiance_falloff_pixel(const lfo_frame* f, size_t x, size_t y, double radius,
                                  const uint32_t* raw32, double rgb[3]) {
  /* :1043-1063.  UniformGridSampler2D::get_sample (sampler.cpp:8-12) builds
   * Vector2D(random_uniform(), random_uniform()); g++ evaluates the two constructor arguments
   * right to left, so the FIRST draw lands in .y -- pinned by the golden frames. */
  double W = (double)(size_t)f->W, H = (double)(size_t)f->H;
  double t[3] = {0, 0, 0};
  for (int s = 0; s < 16; s++) {
    double sy = (double)y + lfo_random_uniform_from_raw(raw32[2 * s]);
    double sx = (double)x + lfo_random_uniform_from_raw(raw32[2 * s + 1]);
    for (int l = 0; l < f->n_flares; l++) {
      double fx = f->flare_origin[l][0] * W, fy = f->flare_origin[l][1] * H;
      double dx = fx - sx, dy = fy - sy;
      double nrm = sqrt(dx * dx + dy * dy) - radius;
      double r = 1 + (0.0 < nrm ? nrm : 0.0); /* std::max(0.0, x) */
      double r2 = pow(r, 1.5);
      double rc = 1.0 / r2; /* Vector3D / double */
      t[0] += rc * f->flare_radiance[l][0];
      t[1] += rc * f->flare_radiance[l][1];
      t[2] += rc * f->flare_radiance[l][2];
    }
  }
  double rc = 1.0 / (double)16;
  rgb[0] = rc * t[0]; rgb[1] = rc * t[1]; rgb[2] = rc * t[2];
}

size_t lfo_tile_order(int W, int H, int tile, uint32_t* order) {
  size_t n = 0;
  for (int ty = 0; ty < H; ty += tile)
    for (int tx = 0; tx < W; tx += tile) {
      int x1 = tx + tile < W ? tx + tile : W, y1 = ty + tile < H ? ty + tile : H;
      for (int y = ty; y < y1; y++)
        for (int x = tx; x < x1; x++) order[n++] = (uint32_t)(x + y * W);
    }
  return n;
}

/* optional scene term (W*H*3, already averaged as in pathtracer.cpp:875), e.g. from
 * lfo_scene_term in lf_scene_oracle.c; NULL = nothing is hit */
static const double* g_scene_term = NULL;
void lfo_set_scene_term(const double* scene) { g_scene_term = scene; }

void lfo_render_pixels(const lfo_frame* f, const float* ap, const lfo_aperture_stats* st,
                       const double* ghost, const uint32_t* order, size_t n_order,
                       uint32_t mt_seed_, int n_threads, double* sample) {
  /* raytrace_pixel :819-899 with a zero scene term.  Draws per visited pixel: 2*ns_aa (pixel
   * jitter, consumed but irrelevant when nothing is hit) then 32 (falloff). */
  size_t per = 2 * (size_t)f->ns_aa + 32;
  uint32_t* raw = (uint32_t*)malloc(sizeof(uint32_t) * per * n_order);
  lfo_mt19937_raw(mt_seed_, 0, per * n_order, raw);
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
  for (long long v = 0; v < (long long)n_order; v++) {
    size_t p = order[v];
    size_t x = p % (size_t)f->W, y = p / (size_t)f->W;
    double sb[3], fo[3];
    lfo_starburst_pixel(f, ap, st, x, y, sb, NULL);
    lfo_irradiance_falloff_pixel(f, x, y, 5.0, raw + per * (size_t)v + 2 * (size_t)f->ns_aa, fo);
    double total[3] = {0, 0, 0};
    double rc = 1. / (double)(f->ns_aa + 1); /* :875 divides by the loop variable = ns_aa+1 */
    for (int c = 0; c < 3; c++) {
      double scene = g_scene_term ? g_scene_term[3 * p + c] : total[c] * rc;
      double g = ghost ? ghost[3 * p + c] : 0.0;
      double star = sb[c] + fo[c];              /* :1004 */
      sample[3 * p + c] = (scene + g) + star;   /* :891 */
    }
  }
  free(raw);
}

void lfo_to_color(const double* sample, size_t n_pixels, uint32_t* rgba) {
  /* HDRImageBuffer::toColor util/image.h:208-223 + ImageBuffer::update_pixel :53-62 */
  float gamma = 2.2f;
  float level = 1.0f;
  float one_over_gamma = 1.0f / gamma;
  float exposure = (float)sqrt(pow(2, level));
  for (size_t i = 0; i < n_pixels; i++) {
    float c[3];
    for (int k = 0; k < 3; k++) {
      double p = pow(sample[3 * i + k] * exposure, one_over_gamma);
      double mn = (1.0 < p) ? 1.0 : p;     /* std::min(p, 1.0) */
      double mx = (0.0 < mn) ? mn : 0.0;   /* std::max(0.0, mn) */
      c[k] = (float)mx;
    }
    uint32_t px = 0;
    /* clamp(0.f, 1.f, c) evaluates min(max(0,1), c) = min(1, c) (misc.h:70-72) */
    px |= ((uint32_t)((c[2] < 1.f ? c[2] : 1.f) * 255)) << 16;
    px |= ((uint32_t)((c[1] < 1.f ? c[1] : 1.f) * 255)) << 8;
    px |= ((uint32_t)((c[0] < 1.f ? c[0] : 1.f) * 255));
    px |= 0xFF000000u;
    rgba[i] = px;
  }
}
