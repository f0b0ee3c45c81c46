/* TEST INFRASTRUCTURE -- CPU oracle of the GEOMETRIC lens march.  Never linked into, imported by
 * or shipped with the product library.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.
 *
 * PARITY STATUS: **parity unpinned**.  The reference has no geometric lens (its ghosts are
 * paraxial ABCD matrices, src/pathtracer/pathtracer.cpp:511-689; its thin-lens camera is a stub,
 * src/pathtracer/camera_lens.cpp:22-30; BSDF::refract is empty, advanced_bsdf.cpp:156-169), so no
 * reference output exists for this path.  This file restates the *specification* in DESIGN.md
 * ("march arithmetic") and is itself pinned by analytic known-answer tests
 * (tests/test_geo_oracle_kat.py): Snell's law, Fresnel at normal incidence / Brewster / TIR, the
 * lensmaker's focal length, and agreement with the reference's own paraxial T/R/L formalism
 * (pathtracer.cpp:527-537) in the small-angle limit.
 *
 * Arithmetic contract: float32; fused multiply-adds are explicit fmaf(); division is the IEEE
 * correctly rounded one; built with -ffp-contract=off -mfma so nothing else fuses.
 * Square roots: the device uses the CDNA4 hardware instruction v_sqrt_f32, which is deterministic
 * but only accurate to 1 ulp and whose deviation from the correctly rounded root depends on the
 * significand and the exponent's parity alone.  geo_set_sqrt_table() installs that deviation
 * (-1, 0, +1 ulp for each of the 2^24 (parity, significand) patterns, measured on the device by the
 * test with lf_native_sqrt) and geo_sqrt() then reproduces v_sqrt_f32 bit for bit; without a table
 * (CPU-only use: known-answer tests, bench.py's cpu_baseline) it is the correctly rounded sqrtf().
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GEO_MAX_SURF 16
#define GEO_MAX_LAMBDA 8

typedef struct {
  int n_surf, stop, n_lambda;
  float radius[GEO_MAX_SURF], thickness[GEO_MAX_SURF], semi_ap[GEO_MAX_SURF];
  float ior[GEO_MAX_LAMBDA][GEO_MAX_SURF];
  float sensor_w_mm;
  float sun_dir[3], sun_radiance[3], sun_angular_radius;
  float lambda_rgb[GEO_MAX_LAMBDA][3];
} geo_lens;

typedef struct {
  uint64_t rays_launched, surface_events, rays_clipped_stop, rays_vignetted, rays_tir,
      rays_reached_scene, rays_hit_light;
} geo_counters;

/* ---- Philox4x32-10, Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11) ---- */
static void philox(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void geo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox(ctr, key, out); }

/* ---- sqrt as the device computes it (see the header) --------------------------------------- */
static const int8_t* g_sqrt_dev = NULL; /* 2^24 entries, index = float bits & 0xffffff */
void geo_set_sqrt_table(const int8_t* dev) { g_sqrt_dev = dev; }
static float geo_sqrt(float x) {
  float r = sqrtf(x);
  if (g_sqrt_dev && x > 0.0f && x < INFINITY) {
    uint32_t xb, rb;
    memcpy(&xb, &x, 4);
    if ((xb >> 23) != 0) { /* normal input */
      memcpy(&rb, &r, 4);
      rb += (uint32_t)(int32_t)g_sqrt_dev[xb & 0xffffffu];
      memcpy(&r, &rb, 4);
    }
  }
  return r;
}
float geo_sqrt_f32(float x) { return geo_sqrt(x); }

/* ---- the reciprocal as the device computes it: v_rcp_f32, followed like the root through a table of its
 * deviation (-1 / 0 / +1 ulp) from the correctly rounded 1 / x, which depends on the significand alone
 * (2^23 entries, measured through lf_native_rcp).  Without a table: the correctly rounded reciprocal.
 * The march's divisions on the per-event and per-sample path are multiplications by it. */
static const int8_t* g_rcp_dev = NULL; /* 2^23 entries, index = float bits & 0x7fffff */
void geo_set_rcp_table(const int8_t* dev) { g_rcp_dev = dev; }
static float geo_rcp(float x) {
  float r = 1.0f / x;
  if (g_rcp_dev && x == x && fabsf(x) < INFINITY && x != 0.0f) {
    uint32_t xb, rb;
    memcpy(&xb, &x, 4);
    memcpy(&rb, &r, 4);
    const uint32_t ex = (xb >> 23) & 0xffu, er = (rb >> 23) & 0xffu;
    if (ex != 0 && er != 0 && er != 0xffu) { /* normal in, normal out */
      rb += (uint32_t)(int32_t)g_rcp_dev[xb & 0x7fffffu];
      memcpy(&r, &rb, 4);
    }
  }
  return r;
}
float geo_rcp_f32(float x) { return geo_rcp(x); }

static float unit24(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

/* ---- derived per-interface constants ------------------------------------------------------ */
typedef struct {
  float zv[GEO_MAX_SURF], curv[GEO_MAX_SURF], h2[GEO_MAX_SURF];
  /* index of the medium on the scene side / the sensor side of interface k (the stop sits inside
   * one medium), and of the medium between the last interface and the sensor */
  float n_before[GEO_MAX_LAMBDA][GEO_MAX_SURF], n_after[GEO_MAX_LAMBDA][GEO_MAX_SURF], n_start[GEO_MAX_LAMBDA];
  float z_sensor, pitch, pupil_h, pupil_z, geom_norm, inv_stop_h, inv_1mc, sun_ss, lobe_thr;
} geo_derived;

/* the disc the sensor samples aim at (lf_set_pupil_target): h <= 0 = the rear element's clear aperture */
static float g_pupil_h = 0.0f, g_pupil_z = 0.0f;
void geo_set_pupil_target(float h, float z) { g_pupil_h = h; g_pupil_z = z; }

static void derive(const geo_lens* L, int W, geo_derived* D) {
  float z = 0.0f;
  for (int k = 0; k < L->n_surf; k++) {
    D->zv[k] = z;
    z = z + L->thickness[k];
    D->curv[k] = L->radius[k] == 0.0f ? 0.0f : 1.0f / L->radius[k];
    D->h2[k] = L->semi_ap[k] * L->semi_ap[k];
  }
  D->z_sensor = z;
  for (int l = 0; l < L->n_lambda; l++) {
    float nb = 1.0f;
    for (int k = 0; k < L->n_surf; k++) {
      float na = (k == L->stop) ? nb : L->ior[l][k];
      D->n_before[l][k] = nb;
      D->n_after[l][k] = na;
      nb = na;
    }
    D->n_start[l] = nb;
  }
  D->pitch = L->sensor_w_mm / (float)W;
  D->pupil_h = L->semi_ap[L->n_surf - 1];
  D->pupil_z = D->zv[L->n_surf - 1];
  if (g_pupil_h > 0.0f) { D->pupil_h = g_pupil_h; D->pupil_z = g_pupil_z; }
  double dist = (double)D->z_sensor - (double)D->pupil_z;
  D->geom_norm = (float)((3.14159265358979323846 * (double)D->pupil_h * (double)D->pupil_h) / (dist * dist));
  D->inv_stop_h = 1.0f / (L->stop >= 0 ? L->semi_ap[L->stop] : 1.0f);
  D->inv_1mc = (float)(1.0 / (1.0 - cos((double)L->sun_angular_radius)));
  {
    /* candidate selection: d.s above this MAY lie inside the lobe (the contract's threshold: a 1/16
       margin on 1 - cos plus 4e-7 absolute, which covers the rounding of the float dot product for
       any lobe size -- a float test on (1 - d.s) * inv loses sub-milliradian suns); the lobe factor
       itself (cancellation-free, below) decides */
    const double thr = 1.0 - (1.0625 / (double)D->inv_1mc) * (1.0 + 1e-6) - 4e-7;
    float t = (float)thr;
    if ((double)t > thr) t = nextafterf(t, -2.0f);
    D->lobe_thr = t;
  }
  /* |s|^2 of the float unit vector as stored (exact in double, then narrowed) */
  D->sun_ss = (float)((double)L->sun_dir[0] * L->sun_dir[0] + (double)L->sun_dir[1] * L->sun_dir[1] +
                      (double)L->sun_dir[2] * L->sun_dir[2]);
}

/* ---- one ray ---------------------------------------------------------------------------- */
/* the weight is carried as the fraction wn / wd (one division at the very end) */
/* p[0], p[1]: x, y; p[2]: z RELATIVE to the vertex of the interface the ray sits on (0 on the sensor
 * and on the stop plane); r2 = x^2 + y^2 of that point as the previous event computed it */
typedef struct { float p[3], d[3], wn, wd, r2; } geo_ray;
enum { OK_ = 0, CLIPPED = 1, VIGNETTED = 2, TIR = 3 };

/* DESIGN.md "march arithmetic", glass interface.  reflect: 0 = Snell refraction, 1 = mirror.
 * forward: the ray travels +z (scene -> sensor).  r->d is the OPTICAL direction K = n d, |K| = n_in, the
 * index of the medium the ray arrives in; n_out: the index on the far side of the interface (a
 * refracted ray leaves with |K| = n_out, a reflected one keeps n_in; its Fresnel factor needs n_out
 * all the same). */
static int glass_event(geo_ray* r, float dzv, float c, float rad, float h2, float n_in, float n_out,
                       int reflect, int forward) {
  /* the row's constants, as lf_march.hip's pack_program derives them (float arithmetic) */
  const float sgn = forward ? 1.0f : -1.0f;
  const float n_in2 = n_in * n_in, n_out2 = n_out * n_out;
  const float ch = 0.5f * c, c2 = 2.0f * c, sc = sgn * c;
  const float cn22 = c2 * n_in2;                          /* 2 c n^2 */
  const float rn2 = c == 0.0f ? 0.0f : rad / n_in2;       /* R / n^2 */
  const float delta = n_out2 - n_in2;                     /* n'^2 - n^2 */
  /* dzv: vertex z of the interface the ray comes from (or of the sensor) minus this one's */
  float oz = r->p[2] + dzv;
  float od = fmaf(r->p[0], r->d[0], fmaf(r->p[1], r->d[1], oz * r->d[2]));
  float oo = fmaf(oz, oz, r->r2);
  float Fh = fmaf(ch, oo, -oz);                           /* F / 2, F = c |o|^2 - 2 o_z */
  float G = fmaf(-c, od, r->d[2]);
  float disc = fmaf(G, G, -(cn22 * Fh));                  /* G^2 - c n^2 F = (n cos(incidence))^2 */
  if (disc < 0.0f) return VIGNETTED;
  float root = geo_sqrt(disc);
  /* the root next to the vertex of c n^2 s^2 - 2 s G + F = 0 (ray o + s K): (G -+ root) R / n^2 for a
   * curved interface, F times the reciprocal of (G +- root) for flat glass */
  float t = (c == 0.0f) ? (Fh + Fh) * geo_rcp(fmaf(sgn, root, G)) : fmaf(-sgn, root, G) * rn2;
  float hx = fmaf(t, r->d[0], r->p[0]), hy = fmaf(t, r->d[1], r->p[1]), hz = fmaf(t, r->d[2], oz);
  float r2 = fmaf(hx, hx, hy * hy);
  if (!(r2 <= h2)) return VIGNETTED;
  /* K . N against the unit normal N = (-c hx, -c hy, 1 - c hz) is sgn * root exactly; Snell:
   * (n' cos t')^2 = disc + (n'^2 - n^2) */
  float k2 = disc + delta;
  if (k2 < 0.0f && !reflect) return TIR;
  float ct = geo_sqrt(k2 >= 0.0f ? k2 : 0.0f);
  /* unpolarised Fresnel straight from the optical cosines root = n cos t, ct = n' cos t':
   * rs = (root - ct) / (root + ct), rp = (n'^2 root - n^2 ct) / (n'^2 root + n^2 ct), R = (rs^2 + rp^2) / 2
   * as ONE fraction Rn / D, Rn = ((a B)^2 + (A b)^2) / 2, D = (b B)^2, scaled by the row's constants
   * fs = 1 / (n + n'), fo = n'^2 / q, fi = n^2 / q, q = n'^2 n + n^2 n' (b = B = 1 at normal incidence) */
  const float q = fmaf(n_out2, n_in, n_in2 * n_out);
  const float fs = 1.0f / (n_in + n_out), fo = n_out2 / q, fi = n_in2 / q;
  float a = (root - ct) * fs, b = (root + ct) * fs;
  float pc = fi * ct;
  float A = fmaf(fo, root, -pc), B = fmaf(fo, root, pc);
  float u = a * B, v = A * b;
  float Rn = 0.5f * fmaf(u, u, v * v);
  float bB = b * B;
  float D = bB * bB;
  if (!reflect) {   /* K' = K + sgn (ct - root) N */
    r->wn *= D - Rn;
    r->wd *= D;
    float gs = ct - root, gcs = gs * sc;
    float nd[3] = {fmaf(-gcs, hx, r->d[0]), fmaf(-gcs, hy, r->d[1]), fmaf(-gcs, hz, fmaf(sgn, gs, r->d[2]))};
    memcpy(r->d, nd, sizeof(nd));
  } else {          /* K' = K - 2 (K . N) N */
    if (k2 >= 0.0f) { r->wn *= Rn; r->wd *= D; } /* else total reflection: R = 1 */
    float m = root * (c2 * sgn);
    float nd[3] = {fmaf(m, hx, r->d[0]), fmaf(m, hy, r->d[1]), fmaf(m, hz, fmaf(-2.0f * sgn, root, r->d[2]))};
    memcpy(r->d, nd, sizeof(nd));
  }
  r->p[0] = hx; r->p[1] = hy; r->p[2] = hz; r->r2 = r2;
  return OK_;
}

static int stop_event(geo_ray* r, float dzv, float h2, float inv_h, const float* mask, int mw, int mh) {
  float t = -(r->p[2] + dzv) * geo_rcp(r->d[2]);
  float hx = fmaf(t, r->d[0], r->p[0]), hy = fmaf(t, r->d[1], r->p[1]);
  float r2 = fmaf(hx, hx, hy * hy);
  if (!(r2 <= h2)) return CLIPPED;
  float fu = fmaf(hx, inv_h, 1.0f) * (0.5f * (float)mw);
  float fv = fmaf(hy, inv_h, 1.0f) * (0.5f * (float)mh);
  int ix = (int)fu, iy = (int)fv;
  if (ix < 0) ix = 0; if (ix > mw - 1) ix = mw - 1;
  if (iy < 0) iy = 0; if (iy > mh - 1) iy = mh - 1;
  float a = mask[iy * mw + ix];
  if (!(a > 0.0f)) return CLIPPED;
  r->wn *= a;
  r->p[0] = hx; r->p[1] = hy; r->p[2] = 0.0f; r->r2 = r2;
  return OK_;
}

/* exported for the known-answer tests: one event on a caller-supplied ray */
int geo_glass_event(float p[3], float d[3], float* w, float zv, float c, float h2, float eta,
                    int reflect, int forward) {
  /* absolute coordinates in and out: the ray "comes from" a vertex at z = 0 */
  /* the caller speaks unit directions and an index ratio: arrive in a medium of index eta, leave into 1 */
  geo_ray r = {{p[0], p[1], p[2]}, {eta * d[0], eta * d[1], eta * d[2]}, *w, 1.0f, fmaf(p[0], p[0], p[1] * p[1])};
  int st = glass_event(&r, 0.0f - zv, c, c == 0.0f ? 0.0f : 1.0f / c, h2, eta, 1.0f, reflect, forward);
  p[0] = r.p[0]; p[1] = r.p[1]; p[2] = zv + r.p[2];
  const float back = reflect ? 1.0f / eta : 1.0f;   /* a reflected ray stays in the first medium */
  d[0] = r.d[0] * back; d[1] = r.d[1] * back; d[2] = r.d[2] * back;
  *w = r.wn / r.wd;
  return st;
}

/* the interface sequence of one ghost pair, in the order the backward ray meets it */
typedef struct { int k, reflect, forward; } geo_step;
static int build_sequence(int n_surf, int i, int j, geo_step* seq) {
  int n = 0;
  if (i < 0) {
    for (int k = n_surf - 1; k >= 0; k--) seq[n++] = (geo_step){k, 0, 0};
    return n;
  }
  for (int k = n_surf - 1; k > i; k--) seq[n++] = (geo_step){k, 0, 0};
  seq[n++] = (geo_step){i, 1, 0};
  for (int k = i + 1; k < j; k++) seq[n++] = (geo_step){k, 0, 1};
  seq[n++] = (geo_step){j, 1, 1};
  for (int k = j - 1; k >= 0; k--) seq[n++] = (geo_step){k, 0, 0};
  return n;
}

/* sensor sample -> ray on the sensor aimed at the rear pupil.  Returns the start weight. */
/* pupil strata: G x G cells, G = floor(sqrt(spp)); sample s < G*G aims at cell (s % G, s / G) */
static int strata(int spp) {
  int g = (int)floor(sqrt((double)spp));
  while ((g + 1) * (g + 1) <= spp) g++;
  while (g * g > spp) g--;
  return g;
}

/* each stratum is split into 2^SUB_BITS x 2^SUB_BITS sub-cells; the sub-cell of sample s is drawn
 * once per 8x8 sensor tile: Philox(ctr = (tile id, s, 0x51bce110, 0), key) */
static int g_sub_bits = 6;   /* the library's default (lf_internal.h) */
void geo_set_sub_bits(int b) { g_sub_bits = b; }
/* log2 of the pixel stride in x of a wave's tile (lf_set_tile_stride): which pixels share a sub-cell draw */
static int g_xs = 3;         /* the library's default: columns 8 apart */
void geo_set_tile_stride_log2(int xs) { g_xs = xs; }

/* sensor point (X, Y) mm + pupil-square point (pa, pb) in [-1, 1]^2 -> start ray (unit direction, weight =
 * the disc's solid-angle factor x cos^4): what k_march, k_lens_rays and the lens camera all start from */
static void aim_at_pupil(const geo_derived* D, float X, float Y, float pa, float pb, geo_ray* r) {
  float qx = 0.0f, qy = 0.0f;
  if (pa != 0.0f || pb != 0.0f) {
    int wide = fabsf(pa) > fabsf(pb);
    float rr = wide ? pa : pb;
    float th = 0.78539816339744831f * ((wide ? pb : pa) * geo_rcp(rr));
    float t2 = th * th;
    float sn = th * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f),
                                  -1.6666667e-1f), 1.0f);
    float cs = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f),
                             -0.5f), 1.0f);
    qx = wide ? rr * cs : rr * sn;
    qy = wide ? rr * sn : rr * cs;
  }
  float vx = fmaf(D->pupil_h, qx, -X), vy = fmaf(D->pupil_h, qy, -Y), vz = D->pupil_z - D->z_sensor;
  float len = geo_sqrt(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
  float rl = geo_rcp(len);
  r->p[0] = X; r->p[1] = Y; r->p[2] = 0.0f;   /* on the sensor plane, relative to it */
  r->r2 = fmaf(X, X, Y * Y);
  r->d[0] = vx * rl; r->d[1] = vy * rl; r->d[2] = vz * rl;
  float c2 = r->d[2] * r->d[2];
  r->wn = D->geom_norm * (c2 * c2);
  r->wd = 1.0f;
}

/* the pupil-square coordinates of the last start_ray of this thread (the device's cull table is looked up by them) */
static _Thread_local float t_ua = 0.5f, t_ub = 0.5f;
static float start_ray(const geo_derived* D, int W, int H, int x, int y, int s, int G,
                       const uint32_t key[2], const uint32_t rnd[4], geo_ray* r) {
  float jx = unit24(rnd[0]), jy = unit24(rnd[1]);
  float ua = unit24(rnd[2]), ub = unit24(rnd[3]);
  if (s < G * G) {
    float inv_g = 1.0f / (float)G, inv_sub = 1.0f / (float)(1 << g_sub_bits);
    int cy = s / G, cx = s - cy * G;
    int tiles_x = ((W + (8 << g_xs) - 1) >> (3 + g_xs)) << g_xs;
    int tx = ((x >> (3 + g_xs)) << g_xs) + (x & ((1 << g_xs) - 1));
    uint32_t tile = (uint32_t)((y >> 3) * tiles_x + tx);
    uint32_t ctr[4] = {tile, (uint32_t)s, 0x51bce110u, 0u}, r2[4];
    philox(ctr, key, r2);
    uint32_t sxi = g_sub_bits ? (r2[0] >> (32 - g_sub_bits)) : 0u;
    uint32_t syi = g_sub_bits ? (r2[1] >> (32 - g_sub_bits)) : 0u;
    ua = ((float)cx + ((float)sxi + ua) * inv_sub) * inv_g;
    ub = ((float)cy + ((float)syi + ub) * inv_sub) * inv_g;
  }
  t_ua = ua; t_ub = ub;
  float pa = fmaf(2.0f, ua, -1.0f), pb = fmaf(2.0f, ub, -1.0f);
  float X = -(((float)x + jx) - 0.5f * (float)W) * D->pitch;
  float Y = -(((float)y + jy) - 0.5f * (float)H) * D->pitch;
  aim_at_pupil(D, X, Y, pa, pb, r);
  return r->wn;
}

/* the primary path N-1 .. 0 of a start ray at wavelength `lambda`: the exit state on the front element
 * (p[2] relative to interface 0's vertex, d = the unit direction in air, weight wn / wd) */
static int primary_path(const geo_lens* L, const geo_derived* D, int lambda, geo_ray* r, const float* mask,
                        int mw, int mh) {
  { const float ns = D->n_start[lambda]; r->d[0] *= ns; r->d[1] *= ns; r->d[2] *= ns; }   /* K = n d */
  float z_from = D->z_sensor;
  for (int k = L->n_surf - 1; k >= 0; k--) {
    float dzv = z_from - D->zv[k];
    int st = (k == L->stop) ? stop_event(r, dzv, D->h2[k], D->inv_stop_h, mask, mw, mh)
                            : glass_event(r, dzv, D->curv[k], L->radius[k], D->h2[k], D->n_after[lambda][k],
                                          D->n_before[lambda][k], 0, 0);
    if (st != OK_) return st;
    z_from = D->zv[k];
  }
  return OK_;
}

static void exit_state(const geo_derived* D, const geo_ray* r, int st, float* o) {
  o[0] = r->p[0]; o[1] = r->p[1]; o[2] = D->zv[0] + r->p[2];
  o[3] = r->d[0]; o[4] = r->d[1]; o[5] = r->d[2];
  o[6] = st == OK_ ? r->wn / r->wd : 0.0f;
  o[7] = st == OK_ ? 1.0f : 0.0f;
}

/* LensCamera::generate_ray, batched, as lf_generate_lens_rays / k_lens_rays computes it: n sensor points
 * (mm) + pupil-square points -> out n x 8 {origin xyz on the front element, direction xyz, weight, alive}.
 * (the first six values of a blocked ray are whatever the march left: compare the alive ones) */
void geo_lens_rays(const geo_lens* L, int W, int lambda, int n, const float* xy, const float* uv,
                   const float* mask, int mw, int mh, float* out) {
  geo_derived D;
  derive(L, W, &D);
  for (int i = 0; i < n; i++) {
    geo_ray r;
    aim_at_pupil(&D, xy[2 * i], xy[2 * i + 1], uv[2 * i], uv[2 * i + 1], &r);
    exit_state(&D, &r, primary_path(L, &D, lambda, &r, mask, mw, mh), out + 8 * (size_t)i);
  }
}

/* The lens camera of the scene term (lf_set_lens_camera; lf_scene.hip k_scene_term<.., LENS>): sample s =
 * 0 .. ns-1 of pixel p is the march's sample -- Philox(ctr = (p, s, 0x6e5f1a2e, 0), key), strata for ns
 * samples, the sub-cell of (tile, s) -- and its primary path at wavelength `lambda`.
 * out: n_pix x ns x 8 floats as geo_lens_rays. */
void geo_lens_samples(const geo_lens* L, int W, int H, int ns, const uint32_t key[2], int lambda,
                      const int* pixels, int n_pix, const float* mask, int mw, int mh, float* out) {
  geo_derived D;
  derive(L, W, &D);
  const int G = strata(ns);
  for (int i = 0; i < n_pix; i++) {
    const int p = pixels[i], x = p % W, y = p / W;
    for (int s = 0; s < ns; s++) {
      uint32_t ctr[4] = {(uint32_t)p, (uint32_t)s, 0x6e5f1a2eu, 0u}, rnd[4];
      philox(ctr, key, rnd);
      geo_ray r;
      start_ray(&D, W, H, x, y, s, G, key, rnd, &r);
      exit_state(&D, &r, primary_path(L, &D, lambda, &r, mask, mw, mh), out + 8 * ((size_t)i * ns + s));
    }
  }
}

/* The device's path culling (lens-flare_amd/csrc/lf_cull.hip, lf_get_cull_table): table[block][cell] = mask of the
 * paths the device STARTS for sample cell `cell` (its pupil stratum s < G*G, else entry `cells`) of the pixels of
 * sensor block (x / block_px, y / block_px), block_px = 16, 32, 64 or 128.  With a table installed geo_trace still marches EVERY path -- its pixels are the
 * full enumeration's, so that equality with the device's pixels proves that nothing the device skipped could
 * have contributed -- but its counters count only the rays the device starts; culled_lit tallies skipped rays
 * that did reach the light (must stay 0). */
static const uint64_t* g_cull = NULL;
static int g_cull_bx = 0, g_cull_by = 0, g_cull_cells = 0, g_cull_shift = 6;
static uint64_t g_culled_lit = 0;
void geo_set_cull(const uint64_t* table, int blocks_x, int blocks_y, int cells, int block_px) {
  g_cull = table; g_cull_bx = blocks_x; g_cull_by = blocks_y; g_cull_cells = cells;
  g_cull_shift = block_px == 128 ? 7 : block_px == 32 ? 5 : block_px == 16 ? 4 : 6;
}
uint64_t geo_culled_lit(void) { return g_culled_lit; }

/* the fixed-point grid of a launch (lens-flare_amd/csrc/lf_march.hip lf_march_fix_bits, the same double
 * arithmetic): 2^36 unless spp x paths x geom_norm x max_c sum_l (radiance[c] * lambda_rgb[l][c]) x 2^bits would
 * reach 2^62 -- then the largest exponent that keeps the largest possible pixel sum below it */
int geo_fix_bits(const geo_lens* L, float geom_norm, int n_paths, int spp) {
  double worst = 0.0;
  for (int c = 0; c < 3; c++) {
    double s = 0.0;
    for (int l = 0; l < L->n_lambda; l++) s += (double)(L->sun_radiance[c] * L->lambda_rgb[l][c]);
    if (s > worst) worst = s;
  }
  worst *= (double)spp * (double)n_paths * (double)geom_norm;
  int bits = 36;
  while (bits > -100 && ldexp(worst, bits) >= 4611686018427387904.0) bits--;
  return bits;
}

/* March `spp` samples of every pixel in rows [y0, y1); pairs = n x (i, j), (-1,-1) = primary.
 * ghost: W*H*3 doubles (only the band is written).  Returns counters. */
void geo_trace(const geo_lens* L, int W, int H, int y0, int y1, int spp, const uint32_t key[2],
               const int* pairs, int n_pairs, const float* mask, int mw, int mh, double* ghost,
               geo_counters* cnt, int n_threads) {
  geo_derived D;
  derive(L, W, &D);
  geo_counters total;
  memset(&total, 0, sizeof(total));
  if (n_threads < 1) n_threads = 1;
  const int fix_bits = geo_fix_bits(L, D.geom_norm, n_pairs, spp);
  const float fix_scale = ldexpf(1.0f, fix_bits);
  const double inv_fix = ldexp(1.0, -fix_bits);
  const int GG = strata(spp) * strata(spp);
  uint64_t culled_lit_total = 0;
#pragma omp parallel num_threads(n_threads)
  {
    geo_counters c, skipped;
    memset(&c, 0, sizeof(c));
    memset(&skipped, 0, sizeof(skipped));
    uint64_t culled_lit = 0;
    geo_step seq[3 * GEO_MAX_SURF];
#pragma omp for schedule(dynamic, 64)
    for (long long p = (long long)y0 * W; p < (long long)y1 * W; p++) {
      int x = (int)(p % W), y = (int)(p / W);
      uint64_t acc[3] = {0, 0, 0};
      for (int s = 0; s < spp; s++) {
        uint32_t ctr[4] = {(uint32_t)p, (uint32_t)s, 0x6e5f1a2eu, 0u}, rnd[4];
        philox(ctr, key, rnd);
        geo_ray r0;
        start_ray(&D, W, H, x, y, s, strata(spp), key, rnd, &r0);
        uint64_t started = ~(uint64_t)0;
        if (g_cull) {
          /* The table cell of the pupil point the sample aims at; P = G m cells per axis.  A stratified sample of a
           * specification with at least m sub-cells per stratum axis: the cell of the sub-cell (sxi, syi) the pixel's
           * wave tile drew (one scalar lookup per wave on the device).  Independent pixels: the cell that holds the
           * pixel's own point, floor(ua P), floor(ub P) in float.  An unstratified sample (s >= G * G): the union entry. */
          int entry = g_cull_cells;
          const int G = strata(spp), P = (int)(sqrt((double)g_cull_cells) + 0.5), m = P / G;
          if (s >= GG) {
            entry = g_cull_cells;              /* an unstratified sample: the block's union entry */
          } else if ((1 << g_sub_bits) < m) {
            int fx = (int)(t_ua * (float)P), fy = (int)(t_ub * (float)P);
            if (fx > P - 1) fx = P - 1;
            if (fy > P - 1) fy = P - 1;
            entry = fy * P + fx;
          } else {
            const int cy = s / G, cx = s - cy * G;
            int tiles_x = ((W + (8 << g_xs) - 1) >> (3 + g_xs)) << g_xs;
            int tx = ((x >> (3 + g_xs)) << g_xs) + (x & ((1 << g_xs) - 1));
            uint32_t c2[4] = {(uint32_t)((y >> 3) * tiles_x + tx), (uint32_t)s, 0x51bce110u, 0u}, r2[4];
            philox(c2, key, r2);
            const uint32_t sxi = g_sub_bits ? (r2[0] >> (32 - g_sub_bits)) : 0u, syi = g_sub_bits ? (r2[1] >> (32 - g_sub_bits)) : 0u;
            entry = (cy * m + (int)((syi * (uint32_t)m) >> g_sub_bits)) * P + cx * m + (int)((sxi * (uint32_t)m) >> g_sub_bits);
          }
          started = g_cull[((size_t)(y >> g_cull_shift) * g_cull_bx + (x >> g_cull_shift)) * (size_t)(g_cull_cells + 1) + (size_t)entry];
        }
        for (int l = 0; l < L->n_lambda; l++)
          for (int q = 0; q < n_pairs; q++) {
            /* a path the device does not start: marched all the same (the pixels are the full enumeration's),
             * tallied aside */
            const int on = q >= 64 || ((started >> q) & 1u);
            geo_counters* const cc = on ? &c : &skipped;
            int n = build_sequence(L->n_surf, pairs[2 * q], pairs[2 * q + 1], seq);
            geo_ray r = r0;
            { const float ns = D.n_start[l]; r.d[0] *= ns; r.d[1] *= ns; r.d[2] *= ns; }   /* K = n d */
            int st = OK_;
            float z_from = D.z_sensor;   /* vertex z of where the ray sits: the sensor, then each interface */
            cc->rays_launched++;
            for (int e = 0; e < n; e++) {
              int k = seq[e].k;
              float dzv = z_from - D.zv[k];
              if (k == L->stop)
                st = stop_event(&r, dzv, D.h2[k], D.inv_stop_h, mask, mw, mh);
              else
                st = glass_event(&r, dzv, D.curv[k], L->radius[k], D.h2[k],
                                 seq[e].forward ? D.n_before[l][k] : D.n_after[l][k],
                                 seq[e].forward ? D.n_after[l][k] : D.n_before[l][k], seq[e].reflect,
                                 seq[e].forward);
              if (st != OK_) break;
              z_from = D.zv[k];
              cc->surface_events++;
            }
            if (st == CLIPPED) { cc->rays_clipped_stop++; continue; }
            if (st == VIGNETTED) { cc->rays_vignetted++; continue; }
            if (st == TIR) { cc->rays_tir++; continue; }
            cc->rays_reached_scene++;
            /* the sun's lobe, DESIGN.md "march arithmetic": candidates are selected on d.s alone
             * against the conservative threshold of geo_derive (1 - d.s cancels: it cannot decide),
             * then 1 - cos(theta) = |d x s|^2 / (|d|^2 |s|^2 + sqrt(|d|^2 |s|^2) d.s), which has no
             * cancellation and does not assume |d| = |s| = 1 */
            float sx = L->sun_dir[0], sy = L->sun_dir[1], sz = L->sun_dir[2];
            float cg = fmaf(r.d[0], sx, fmaf(r.d[1], sy, r.d[2] * sz));
            float qq = 2.0f;
            if (cg > D.lobe_thr) {
              float cx = fmaf(r.d[1], sz, -(r.d[2] * sy)), cy = fmaf(r.d[2], sx, -(r.d[0] * sz));
              float cz = fmaf(r.d[0], sy, -(r.d[1] * sx));
              float c2 = fmaf(cx, cx, fmaf(cy, cy, cz * cz));
              float dd = fmaf(r.d[0], r.d[0], fmaf(r.d[1], r.d[1], r.d[2] * r.d[2]));
              float ds = dd * D.sun_ss;
              float den = fmaf(geo_sqrt(ds), cg, ds);
              qq = (c2 / den) * D.inv_1mc;
            }
            if (qq < 1.0f) {
              float om = 1.0f - qq;
              float contrib = (r.wn / r.wd) * (om * om);
              if (contrib > 0.0f) {
                cc->rays_hit_light++;
                if (!on) culled_lit++;
                for (int ch = 0; ch < 3; ch++) {
                  float v = contrib * (L->sun_radiance[ch] * L->lambda_rgb[l][ch]);
                  acc[ch] += (uint64_t)(v * fix_scale);
                }
              }
            }
          }
      }
      for (int ch = 0; ch < 3; ch++)
        ghost[3 * p + ch] = ((double)acc[ch] * inv_fix) / (double)spp;
    }
#pragma omp critical
    {
      total.rays_launched += c.rays_launched; total.surface_events += c.surface_events;
      total.rays_clipped_stop += c.rays_clipped_stop; total.rays_vignetted += c.rays_vignetted;
      total.rays_tir += c.rays_tir; total.rays_reached_scene += c.rays_reached_scene;
      total.rays_hit_light += c.rays_hit_light;
      culled_lit_total += culled_lit;
    }
  }
  g_culled_lit = culled_lit_total;
  if (cnt) *cnt = total;
}

/* Trace ONE explicit ray (position/direction given) through an explicit sequence; for the KATs.
 * Returns status; p/d/w updated; events executed in *n_events. */
int geo_trace_ray(const geo_lens* L, int lambda, int i, int j, float p[3], float d[3], float* w,
                  const float* mask, int mw, int mh, int* n_events) {
  geo_derived D;
  derive(L, 64, &D);
  geo_step seq[3 * GEO_MAX_SURF];
  int n = build_sequence(L->n_surf, i, j, seq);
  /* absolute coordinates in and out; inside, z is relative to the sensor plane, then to each vertex */
  const float ns = D.n_start[lambda];   /* K = n d in the medium in front of the sensor */
  geo_ray r = {{p[0], p[1], p[2] - D.z_sensor}, {d[0] * ns, d[1] * ns, d[2] * ns}, *w, 1.0f, fmaf(p[0], p[0], p[1] * p[1])};
  int st = OK_, ev = 0;
  float z_from = D.z_sensor;
  for (int e = 0; e < n; e++) {
    int k = seq[e].k;
    float dzv = z_from - D.zv[k];
    if (k == L->stop) st = stop_event(&r, dzv, D.h2[k], D.inv_stop_h, mask, mw, mh);
    else st = glass_event(&r, dzv, D.curv[k], L->radius[k], D.h2[k],
                          seq[e].forward ? D.n_before[lambda][k] : D.n_after[lambda][k],
                          seq[e].forward ? D.n_after[lambda][k] : D.n_before[lambda][k],
                          seq[e].reflect, seq[e].forward);
    if (st != OK_) break;
    z_from = D.zv[k];
    ev++;
  }
  p[0] = r.p[0]; p[1] = r.p[1]; p[2] = z_from + r.p[2];
  memcpy(d, r.d, sizeof(r.d)); *w = r.wn / r.wd;
  if (n_events) *n_events = ev;
  return st;
}

/* Diagnostics for kernel design: alive[q][e] = rays of pair q still alive after e events
 * (e = 0 .. 32), over `spp` samples of the pixels in rows [y0,y1), wavelength 1. */
void geo_survival(const geo_lens* L, int W, int H, int y0, int y1, int spp, const uint32_t key[2],
                  const int* pairs, int n_pairs, const float* mask, int mw, int mh,
                  uint64_t* alive /* n_pairs x 33 */) {
  geo_derived D;
  derive(L, W, &D);
  geo_step seq[3 * GEO_MAX_SURF];
  memset(alive, 0, sizeof(uint64_t) * 33 * (size_t)n_pairs);
  int l = L->n_lambda > 1 ? 1 : 0;
  for (long long p = (long long)y0 * W; p < (long long)y1 * W; p++)
    for (int s = 0; s < spp; s++) {
      uint32_t ctr[4] = {(uint32_t)p, (uint32_t)s, 0x6e5f1a2eu, 0u}, rnd[4];
      philox(ctr, key, rnd);
      geo_ray r0;
      start_ray(&D, W, H, (int)(p % W), (int)(p / W), s, strata(spp), key, rnd, &r0);
      for (int q = 0; q < n_pairs; q++) {
        int n = build_sequence(L->n_surf, pairs[2 * q], pairs[2 * q + 1], seq);
        geo_ray r = r0;
        { const float ns = D.n_start[l]; r.d[0] *= ns; r.d[1] *= ns; r.d[2] *= ns; }
        float z_from = D.z_sensor;
        alive[q * 33]++;
        for (int e = 0; e < n; e++) {
          int k = seq[e].k, st;
          float dzv = z_from - D.zv[k];
          if (k == L->stop) st = stop_event(&r, dzv, D.h2[k], D.inv_stop_h, mask, mw, mh);
          else st = glass_event(&r, dzv, D.curv[k], L->radius[k], D.h2[k],
                                seq[e].forward ? D.n_before[l][k] : D.n_after[l][k],
                                seq[e].forward ? D.n_after[l][k] : D.n_before[l][k], seq[e].reflect,
                                seq[e].forward);
          if (st != OK_) break;
          z_from = D.zv[k];
          alive[q * 33 + e + 1]++;
        }
      }
    }
}

float geo_z_sensor(const geo_lens* L) {
  geo_derived D;
  derive(L, 64, &D);
  return D.z_sensor;
}
