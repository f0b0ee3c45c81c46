/* TEST INFRASTRUCTURE -- an INDEPENDENT double-precision tracer for the geometric lens march.
 * Never linked into, imported by or shipped with the product library; only tests/ load it.
 *
 * Why it exists: the device march (lens-flare_amd/csrc/lf_march.hip) is compared bit for bit with
 * oracle/lf_geo_oracle.c, which restates the same float32 recipe and follows the device's
 * v_sqrt_f32 through a measured table.  That proves the two agree, not that either is right.  This
 * file is the second opinion at the north star's actual bar (<= 1e-4 relative per pixel): it
 * shares NO code, NO arithmetic recipe and NO square-root table with lf_geo_oracle.c --
 *   * float64 throughout, libm sqrt / sin / cos;
 *   * textbook formulations: sphere given by centre + radius and the quadratic in the ray
 *     parameter (both roots, the hit nearer the vertex plane is taken), the unit normal
 *     (hit - centre) / R, vector Snell d' = eta d + (eta cos_i - cos_t) n, mirror d' = d - 2 (d.n) n,
 *     Fresnel from the two amplitude coefficients r_s, r_p with the media's actual indices;
 *   * its own Philox4x32-10 (Salmon et al., SC'11) and its own pair sequencing.
 * The ONLY thing it takes from the specification is what defines the
 * estimator itself: which sensor point and which rear-pupil point sample (pixel, s) uses, and how a
 * ray that leaves the front element is weighted by the sun's lobe.
 *
 * PARITY STATUS: the reference has no geometric lens (src/pathtracer/camera_lens.cpp:22-30 is a
 * stub, advanced_bsdf.cpp:156-169 an empty refract), so like lf_geo_oracle.c this tracer is
 * "parity unpinned" against the reference; it is anchored by the same analytic known-answer tests
 * and by the small-angle agreement with the reference's own T / R / L matrices
 * (src/pathtracer/pathtracer.cpp:527-537, :588-689) -- tests/test_geo_f64_kat.py.
 *
 * Fragile rays.  Float32 and float64 disagree about the FATE of a ray that passes within rounding
 * distance of a decision boundary (edge of a clear aperture, edge of an aperture-mask texel whose
 * neighbour differs, critical angle, grazing miss).  Such decisions are taken as float64 takes
 * them, but the ray is marked fragile, followed as if every fragile decision had passed, and the
 * contribution it then could make is summed per pixel in `frag`.  A faithful float32 evaluation
 * then satisfies  |pixel32 - pixel64| <= tol * pixel64 + frag  with tol far below 1e-4.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define G64_MAX_SURF 16
#define G64_MAX_LAMBDA 8

typedef struct {
  int n_surf, stop, n_lambda;
  double radius[G64_MAX_SURF], thickness[G64_MAX_SURF], semi_ap[G64_MAX_SURF];
  double ior[G64_MAX_LAMBDA][G64_MAX_SURF]; /* medium BEHIND interface k (sensor side) */
  double sensor_w_mm;
  double sun_dir[3], sun_radiance[3], sun_angular_radius;
  double lambda_rgb[G64_MAX_LAMBDA][3];
  double eps_mm;    /* a geometric decision closer than this to its boundary is fragile */
  double eps_texel; /* ... in mask-texel units */
  double eps_cos;   /* ... for cos^2 of the refraction angle near the critical angle */
} g64_lens;

typedef struct { double x, y, z; } vec;
static vec V(double x, double y, double z) { vec v = {x, y, z}; return v; }
static vec add(vec a, vec b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static vec sub(vec a, vec b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static vec scale(vec a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static double dotp(vec a, vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static vec normalise(vec a) { return scale(a, 1.0 / sqrt(dotp(a, a))); }

/* Philox4x32-10 with 128-bit products */
static void philox64(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
  uint32_t c[4] = {c_in[0], c_in[1], c_in[2], c_in[3]}, k[2] = {k_in[0], k_in[1]};
  for (int round = 0; round < 10; round++) {
    unsigned __int128 both = ((unsigned __int128)((uint64_t)0xCD9E8D57u * c[2]) << 64) |
                             (uint64_t)((uint64_t)0xD2511F53u * c[0]);
    uint64_t lo_prod = (uint64_t)both, hi_prod = (uint64_t)(both >> 64);
    uint32_t next[4] = {(uint32_t)(hi_prod >> 32) ^ c[1] ^ k[0], (uint32_t)hi_prod,
                        (uint32_t)(lo_prod >> 32) ^ c[3] ^ k[1], (uint32_t)lo_prod};
    memcpy(c, next, sizeof(c));
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  memcpy(out, c, sizeof(c));
}
void g64_philox(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]) { philox64(c, k, out); }

/* ---- the optical system in world terms ---------------------------------------------------- */
typedef struct {
  double vertex_z[G64_MAX_SURF];
  double n_scene[G64_MAX_LAMBDA][G64_MAX_SURF];  /* index on the scene side (lower z) of interface k */
  double n_sensor[G64_MAX_LAMBDA][G64_MAX_SURF]; /* index on the sensor side */
  double sensor_z;
} g64_system;

static void lay_out(const g64_lens* L, g64_system* S) {
  double z = 0.0;
  for (int k = 0; k < L->n_surf; k++) { S->vertex_z[k] = z; z += L->thickness[k]; }
  S->sensor_z = z;
  for (int l = 0; l < L->n_lambda; l++) {
    double medium = 1.0; /* air in front of the lens */
    for (int k = 0; k < L->n_surf; k++) {
      S->n_scene[l][k] = medium;
      if (k != L->stop) medium = L->ior[l][k]; /* the stop sits inside one medium */
      S->n_sensor[l][k] = medium;
    }
  }
}

typedef struct {
  vec o, d;
  double w;      /* weight as float64 decides */
  double w_pot;  /* weight if every fragile decision passes (upper bound for texel choices) */
  int fragile;
  int dead;      /* 0 alive, else cause: 1 mask/stop, 2 aperture/miss, 3 total reflection */
  int cause;     /* why fragile, bits: 1 the rim of a clear aperture / of the stop's housing, 2 a mask texel's edge,
                    4 the critical angle, 8 a grazing miss of a sphere */
} g64_ray;

/* one spherical (or flat) glass interface: refraction or mirror reflection */
static void glass(const g64_lens* L, const g64_system* S, int lam, int k, int mirror, g64_ray* r) {
  const double R = L->radius[k], h = L->semi_ap[k];
  double t;
  if (R == 0.0) {
    t = (S->vertex_z[k] - r->o.z) / r->d.z;
  } else {
    const vec centre = V(0.0, 0.0, S->vertex_z[k] + R);
    const vec oc = sub(r->o, centre);
    const double b = dotp(oc, r->d), c = dotp(oc, oc) - R * R;
    const double disc = b * b - c;
    if (disc < 0.0) {
      if (!r->dead) r->dead = 2;
      /* a grazing miss within rounding distance cannot be followed further: charge the whole
       * weight it carried as the fragile bound (rare: the clear apertures cut in long before) */
      if (disc > -L->eps_mm * fabs(R)) { r->fragile = 2; r->cause |= 8; }
      r->w = 0.0;
      return;
    }
    const double sq = sqrt(disc);
    const double t1 = -b - sq, t2 = -b + sq;
    const double z1 = r->o.z + t1 * r->d.z - S->vertex_z[k], z2 = r->o.z + t2 * r->d.z - S->vertex_z[k];
    t = fabs(z1) <= fabs(z2) ? t1 : t2; /* the intersection nearer the vertex plane */
  }
  const vec hit = add(r->o, scale(r->d, t));
  const double rho = sqrt(hit.x * hit.x + hit.y * hit.y);
  if (fabs(rho - h) < L->eps_mm) { r->fragile = 1; r->cause |= 1; }
  if (rho > h) {
    if (!r->dead) r->dead = 2;
    if (!r->fragile) { r->w = 0.0; return; }
    r->w = 0.0; /* float64 says vignetted; follow it for the potential weight only */
  }
  vec n = R == 0.0 ? V(0, 0, 1) : scale(sub(hit, V(0.0, 0.0, S->vertex_z[k] + R)), 1.0 / R);
  double cos_i = -dotp(r->d, n);
  if (cos_i < 0.0) { n = scale(n, -1.0); cos_i = -cos_i; } /* normal against the ray */
  /* media: a ray travelling +z goes from the scene side to the sensor side */
  const int towards_sensor = r->d.z > 0.0;
  const double n1 = towards_sensor ? S->n_scene[lam][k] : S->n_sensor[lam][k];
  const double n2 = towards_sensor ? S->n_sensor[lam][k] : S->n_scene[lam][k];
  const double eta = n1 / n2;
  const double sin2_t = eta * eta * (1.0 - cos_i * cos_i);
  const int tir = sin2_t > 1.0;
  if (fabs(1.0 - sin2_t) < L->eps_cos) { r->fragile = 1; r->cause |= 4; }
  double reflectance = 1.0, cos_t = 0.0;
  if (!tir) {
    cos_t = sqrt(1.0 - sin2_t);
    const double rs = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t);
    const double rp = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t);
    reflectance = 0.5 * (rs * rs + rp * rp);
  }
  if (mirror) {
    r->w *= reflectance;
    r->w_pot *= reflectance;
    r->d = add(r->d, scale(n, 2.0 * cos_i));
  } else {
    if (tir) {
      if (!r->dead) r->dead = 3;
      r->w = 0.0;
      if (!r->fragile) return;
      cos_t = 0.0; /* follow the critical ray for the potential weight */
    }
    r->w *= 1.0 - reflectance;
    r->w_pot *= tir ? 1.0 : 1.0 - reflectance;
    r->d = normalise(add(scale(r->d, eta), scale(n, eta * cos_i - cos_t)));
  }
  r->o = hit;
}

/* the stop: a plane with a round housing and the aperture mask (nearest texel) */
static void stop_plane(const g64_lens* L, const g64_system* S, int k, const float* mask, int mw, int mh,
                       g64_ray* r) {
  const double t = (S->vertex_z[k] - r->o.z) / r->d.z;
  const vec hit = add(r->o, scale(r->d, t));
  const double h = L->semi_ap[k];
  const double rho = sqrt(hit.x * hit.x + hit.y * hit.y);
  if (fabs(rho - h) < L->eps_mm) { r->fragile = 1; r->cause |= 1; }
  int outside = rho > h;
  const double fu = (hit.x / h + 1.0) * (0.5 * mw), fv = (hit.y / h + 1.0) * (0.5 * mh);
  int ix = (int)fu, iy = (int)fv; /* truncation, like the specification's (int) cast */
  if (ix < 0) ix = 0; if (ix > mw - 1) ix = mw - 1;
  if (iy < 0) iy = 0; if (iy > mh - 1) iy = mh - 1;
  const double a = mask[iy * mw + ix];
  /* texel neighbours a float32 evaluation could land in instead */
  double a_max = a, a_min = a;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      if (!dx && !dy) continue;
      if (dx < 0 && fu - floor(fu) > L->eps_texel) continue;
      if (dx > 0 && ceil(fu) - fu > L->eps_texel && fu != floor(fu)) continue;
      if (dx > 0 && fu == floor(fu)) continue;
      if (dy < 0 && fv - floor(fv) > L->eps_texel) continue;
      if (dy > 0 && ceil(fv) - fv > L->eps_texel && fv != floor(fv)) continue;
      if (dy > 0 && fv == floor(fv)) continue;
      int jx = ix + dx, jy = iy + dy;
      if (jx < 0 || jy < 0 || jx >= mw || jy >= mh) continue;
      const double b = mask[jy * mw + jx];
      if (b > a_max) a_max = b;
      if (b < a_min) a_min = b;
    }
  if (a_max != a_min) { r->fragile = 1; r->cause |= 2; }
  if (outside || !(a > 0.0)) {
    if (!r->dead) r->dead = 1;
    r->w = 0.0;
    if (!r->fragile) return;
  }
  r->w *= a;
  r->w_pot *= a_max;
  r->o = hit;
}

/* the interfaces a ray meets, in order, for ghost pair (i, j); i < 0 = no reflection */
static int itinerary(int n, int i, int j, int* surf, int* is_mirror) {
  int m = 0;
  if (i < 0) {
    for (int k = n - 1; k >= 0; k--) { surf[m] = k; is_mirror[m++] = 0; }
    return m;
  }
  for (int k = n - 1; k > i; k--) { surf[m] = k; is_mirror[m++] = 0; }
  surf[m] = i; is_mirror[m++] = 1;
  for (int k = i + 1; k < j; k++) { surf[m] = k; is_mirror[m++] = 0; }
  surf[m] = j; is_mirror[m++] = 1;
  for (int k = j - 1; k >= 0; k--) { surf[m] = k; is_mirror[m++] = 0; }
  return m;
}

/* follow one ray along one itinerary; returns the events completed while (really) alive */
static int follow(const g64_lens* L, const g64_system* S, int lam, int i, int j, const float* mask,
                  int mw, int mh, g64_ray* r) {
  int surf[3 * G64_MAX_SURF], mir[3 * G64_MAX_SURF];
  const int m = itinerary(L->n_surf, i, j, surf, mir);
  int events = 0;
  for (int e = 0; e < m; e++) {
    const int was_dead = r->dead;
    if (surf[e] == L->stop) stop_plane(L, S, surf[e], mask, mw, mh, r);
    else glass(L, S, lam, surf[e], mir[e], r);
    if (!r->dead) events++;
    if (r->dead && !was_dead && !r->fragile) return events;  /* settled: nothing more to learn */
    if (r->fragile == 2) return events;                        /* cannot be followed */
    if (r->dead && !r->fragile) return events;
  }
  return events;
}

/* exported for the known-answer tests: one glass event / one whole path on a caller-supplied ray */
int g64_glass_event(const g64_lens* L, int lam, int k, int mirror, double p[3], double d[3], double* w) {
  g64_system S;
  lay_out(L, &S);
  g64_ray r = {V(p[0], p[1], p[2]), V(d[0], d[1], d[2]), *w, *w, 0, 0};
  glass(L, &S, lam, k, mirror, &r);
  p[0] = r.o.x; p[1] = r.o.y; p[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z; *w = r.w;
  return r.dead;
}

int g64_trace_ray(const g64_lens* L, int lam, int i, int j, double p[3], double d[3], double* w,
                  const float* mask, int mw, int mh, int* n_events) {
  g64_system S;
  lay_out(L, &S);
  g64_ray r = {V(p[0], p[1], p[2]), V(d[0], d[1], d[2]), *w, *w, 0, 0};
  const int ev = follow(L, &S, lam, i, j, mask, mw, mh, &r);
  p[0] = r.o.x; p[1] = r.o.y; p[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z; *w = r.w;
  if (n_events) *n_events = ev;
  return r.dead;
}

/* ... the same with the tracer's own verdict on how close the ray came to a decision boundary:
 * out = {fragile (0 / 1 / 2), potential weight} */
int g64_trace_ray_ex(const g64_lens* L, int lam, int i, int j, double p[3], double d[3], double* w,
                     const float* mask, int mw, int mh, int* n_events, double out[2]) {
  g64_system S;
  lay_out(L, &S);
  g64_ray r = {V(p[0], p[1], p[2]), V(d[0], d[1], d[2]), *w, *w, 0, 0};
  const int ev = follow(L, &S, lam, i, j, mask, mw, mh, &r);
  p[0] = r.o.x; p[1] = r.o.y; p[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z; *w = r.w;
  if (n_events) *n_events = ev;
  out[0] = (double)r.fragile; out[1] = r.w_pot;
  return r.dead;
}

double g64_sensor_z(const g64_lens* L) {
  g64_system S;
  lay_out(L, &S);
  return S.sensor_z;
}

static double unit_interval(uint32_t bits) { return (double)(bits >> 8) / 16777216.0; }

/* the disc the sensor samples aim at (lf_set_pupil_target, float values held in doubles):
 * h <= 0 = the rear element's clear aperture at its vertex plane */
static double g64_pupil_h = 0.0, g64_pupil_z = 0.0;
void g64_set_pupil_target(double h, double z) { g64_pupil_h = h; g64_pupil_z = z; }
/* g64_trace only traces the pixels with x in [x0, x1) (the others stay 0 and launch nothing): a window of
 * a wide band keeps a full-sample-count check inside a test's time budget.  Default: every column. */
/* log2 of the pixel stride in x of a wave's tile (lf_set_tile_stride): which pixels share a sub-cell draw */
static int g64_xs = 3;   /* the library's default: columns 8 apart */
void g64_set_tile_stride_log2(int xs) { g64_xs = xs; }
static int g64_x0 = 0, g64_x1 = 1 << 30;
void g64_set_x_window(int x0, int x1) { g64_x0 = x0; g64_x1 = x1; }

/* The estimator's sample: sensor point and rear-pupil point of sample s of
 * pixel (x, y).  Returns the start direction and the start weight. */
static _Thread_local double t64_ua = 0.5, t64_ub = 0.5;   /* the pupil-square point of this thread's last sample_ray */
static double sample_ray(const g64_lens* L, const g64_system* S, int W, int H, int x, int y, int s, int spp,
                         int sub_bits, const uint32_t key[2], vec* origin, vec* dir) {
  const uint32_t ctr[4] = {(uint32_t)(y * W + x), (uint32_t)s, 0x6e5f1a2eu, 0u};
  uint32_t rnd[4];
  philox64(ctr, key, rnd);
  double ua = unit_interval(rnd[2]), ub = unit_interval(rnd[3]);
  int G = (int)floor(sqrt((double)spp));
  while ((G + 1) * (G + 1) <= spp) G++;
  while (G * G > spp) G--;
  if (s < G * G) {
    const int cy = s / G, cx = s % G;
    const int per_block = 1 << g64_xs, block_w = 8 * per_block;
    const int tiles_x = ((W + block_w - 1) / block_w) * per_block;
    const uint32_t tile = (uint32_t)((y / 8) * tiles_x + (x / block_w) * per_block + x % per_block);
    const uint32_t c2[4] = {tile, (uint32_t)s, 0x51bce110u, 0u};
    uint32_t r2[4];
    philox64(c2, key, r2);
    const double sub = (double)(1 << sub_bits);
    const double sx = sub_bits ? (double)(r2[0] >> (32 - sub_bits)) : 0.0;
    const double sy = sub_bits ? (double)(r2[1] >> (32 - sub_bits)) : 0.0;
    ua = (cx + (sx + ua) / sub) / G;
    ub = (cy + (sy + ub) / sub) / G;
  }
  t64_ua = ua; t64_ub = ub;
  const double pitch = L->sensor_w_mm / W;
  const double X = -((x + unit_interval(rnd[0])) - 0.5 * W) * pitch;
  const double Y = -((y + unit_interval(rnd[1])) - 0.5 * H) * pitch;
  /* concentric square -> disc (Shirley & Chiu) */
  const double a = 2.0 * ua - 1.0, b = 2.0 * ub - 1.0;
  double qx = 0.0, qy = 0.0;
  if (a != 0.0 || b != 0.0) {
    if (fabs(a) > fabs(b)) { const double th = (M_PI / 4.0) * (b / a); qx = a * cos(th); qy = a * sin(th); }
    else { const double th = (M_PI / 4.0) * (a / b); qx = b * sin(th); qy = b * cos(th); }
  }
  const int last = L->n_surf - 1;
  const double pupil_h = g64_pupil_h > 0.0 ? g64_pupil_h : L->semi_ap[last];
  const double pupil_z = g64_pupil_h > 0.0 ? g64_pupil_z : S->vertex_z[last];
  *origin = V(X, Y, S->sensor_z);
  *dir = normalise(sub(V(pupil_h * qx, pupil_h * qy, pupil_z), *origin));
  const double dist = S->sensor_z - pupil_z;
  const double cos2 = dir->z * dir->z;
  return (M_PI * pupil_h * pupil_h / (dist * dist)) * cos2 * cos2;
}

/* The device's path culling (lf_get_cull_table), as in lf_geo_oracle.c: with a table installed the IMAGE is still
 * the full enumeration's -- the independent evidence that what the device skips adds nothing -- while the
 * counters count the rays the device starts (block = 64 x 64 pixels, entry = the sample's pupil stratum). */
static const uint64_t* g64_cull = NULL;
static int g64_cull_bx = 0, g64_cull_cells = 0, g64_cull_shift = 6;
/* W * H * 4 doubles (or NULL): per pixel, the fragile rays' potential weight (summed over the channels, / spp) by cause --
 * aperture rim, mask texel edge, critical angle, grazing miss: what tests/test_gpu_march_f64.py tabulates */
static double* g64_cause = NULL;
void g64_set_cause_buffer(double* buf) { g64_cause = buf; }
void g64_set_cull(const uint64_t* table, int blocks_x, int blocks_y, int cells, int block_px) {
  (void)blocks_y;
  g64_cull = table; g64_cull_bx = blocks_x; g64_cull_cells = cells;
  g64_cull_shift = block_px == 128 ? 7 : block_px == 32 ? 5 : block_px == 16 ? 4 : 6;
}

/* image, frag: W*H*3 doubles (rows [y0, y1) are written); counters: launched, events, clipped at
 * the stop, vignetted, totally reflected, reached the scene, hit the light, fragile rays */
void g64_trace(const g64_lens* L, int W, int H, int y0, int y1, int spp, const uint32_t key[2],
               int sub_bits, const int* pairs, int n_pairs, const float* mask, int mw, int mh,
               double* image, double* frag, uint64_t counters[8], int n_threads) {
  g64_system S;
  lay_out(L, &S);
  const double lobe = 1.0 / (1.0 - cos(L->sun_angular_radius));
  const vec sun = normalise(V(L->sun_dir[0], L->sun_dir[1], L->sun_dir[2])); /* the angle is between directions */
  uint64_t total[8] = {0};
  if (n_threads < 1) n_threads = 1;
  int G = (int)floor(sqrt((double)spp));
  while ((G + 1) * (G + 1) <= spp) G++;
  while (G * G > spp) G--;
  const int GG = G * G;
#pragma omp parallel num_threads(n_threads)
  {
    uint64_t counted[8] = {0}, skipped[8] = {0};
#pragma omp for schedule(dynamic, 16)
    for (long long p = (long long)y0 * W; p < (long long)y1 * W; p++) {
      const int x = (int)(p % W), y = (int)(p / W);
      if (x < g64_x0 || x >= g64_x1) continue;
      double sum[3] = {0, 0, 0}, fsum[3] = {0, 0, 0};
      for (int s = 0; s < spp; s++) {
        vec o, d;
        const double w0 = sample_ray(L, &S, W, H, x, y, s, spp, sub_bits, key, &o, &d);
        uint64_t started = ~(uint64_t)0;
        if (g64_cull) {
          /* P = G m table cells per axis; the cell of the sub-cell the pixel's wave tile aims sample s at */
          int entry = g64_cull_cells;
          const int P = (int)(sqrt((double)g64_cull_cells) + 0.5), m = P / G;
          if (s >= GG) {
            entry = g64_cull_cells;                 /* an unstratified sample: the block's union entry */
          } else if ((1 << sub_bits) < m) {         /* the cell of the pixel's own pupil point (the device: in float) */
            int fx = (int)(t64_ua * P), fy = (int)(t64_ub * P);
            if (fx > P - 1) fx = P - 1;
            if (fy > P - 1) fy = P - 1;
            entry = fy * P + fx;
          } else {
            const int cy = s / G, cx = s % G;
            const int per_block = 1 << g64_xs, block_w = 8 * per_block;
            const int tiles_x = ((W + block_w - 1) / block_w) * per_block;
            const uint32_t c2[4] = {(uint32_t)((y / 8) * tiles_x + (x / block_w) * per_block + x % per_block), (uint32_t)s, 0x51bce110u, 0u};
            uint32_t r2[4];
            philox64(c2, key, r2);
            const uint32_t sxi = sub_bits ? (r2[0] >> (32 - sub_bits)) : 0u, syi = sub_bits ? (r2[1] >> (32 - sub_bits)) : 0u;
            entry = (cy * m + (int)((syi * (uint32_t)m) >> sub_bits)) * P + cx * m + (int)((sxi * (uint32_t)m) >> sub_bits);
          }
          started = g64_cull[((size_t)(y >> g64_cull_shift) * g64_cull_bx + (x >> g64_cull_shift)) * (size_t)(g64_cull_cells + 1) + (size_t)entry];
        }
        for (int lam = 0; lam < L->n_lambda; lam++)
          for (int q = 0; q < n_pairs; q++) {
            uint64_t* const c = (q >= 64 || ((started >> q) & 1u)) ? counted : skipped;
            g64_ray r = {o, d, w0, w0, 0, 0, 0};
            c[0]++;
            c[1] += (uint64_t)follow(L, &S, lam, pairs[2 * q], pairs[2 * q + 1], mask, mw, mh, &r);
            if (r.fragile) c[7]++;
            if (r.dead == 1) c[2]++;
            else if (r.dead == 2) c[3]++;
            else if (r.dead == 3) c[4]++;
            else c[5]++;
            if (r.dead && !r.fragile) continue;
            double shade = 1.0; /* a fragile ray that could not be followed: bound by its weight */
            if (r.fragile != 2) {
              const double qq = (1.0 - dotp(normalise(r.d), sun)) * lobe;
              if (!(qq < 1.0)) continue;
              shade = (1.0 - qq) * (1.0 - qq);
            }
            if (!r.dead && r.w * shade > 0.0) c[6]++;
            for (int ch = 0; ch < 3; ch++) {
              const double colour = L->sun_radiance[ch] * L->lambda_rgb[lam][ch];
              if (!r.dead) sum[ch] += r.w * shade * colour;
              if (r.fragile) fsum[ch] += r.w_pot * shade * colour;
              /* ... and by cause (a ray with several: each of them), if the caller asked (g64_set_cause_buffer) */
              if (r.fragile && g64_cause)
                for (int k = 0; k < 4; k++)
                  if (r.cause & (1 << k)) g64_cause[4 * (size_t)p + k] += r.w_pot * shade * colour / spp;
            }
          }
      }
      for (int ch = 0; ch < 3; ch++) {
        image[3 * p + ch] = sum[ch] / spp;
        frag[3 * p + ch] = fsum[ch] / spp;
      }
    }
#pragma omp critical
    for (int k = 0; k < 8; k++) total[k] += counted[k];
  }
  if (counters) memcpy(counters, total, sizeof(total));
}

/* The lens camera of the scene term (lf_set_lens_camera): the primary path of sample s = 0 .. ns-1 of the
 * listed pixels at wavelength `lam`, in float64 with the textbook formulations above.
 * out: n_pix x ns x 10 doubles {origin xyz on the front element (mm, lens space), unit direction xyz,
 * weight as float64 decides (0 when blocked), weight if every fragile decision passes, fragile, dead} */
void g64_lens_samples(const g64_lens* L, int W, int H, int ns, const uint32_t key[2], int sub_bits, int lam,
                      const int* pixels, int n_pix, const float* mask, int mw, int mh, double* out) {
  g64_system S;
  lay_out(L, &S);
  for (int i = 0; i < n_pix; i++) {
    const int p = pixels[i], x = p % W, y = p / W;
    for (int s = 0; s < ns; s++) {
      vec o, d;
      const double w0 = sample_ray(L, &S, W, H, x, y, s, ns, sub_bits, key, &o, &d);
      g64_ray r = {o, d, w0, w0, 0, 0};
      follow(L, &S, lam, -1, -1, mask, mw, mh, &r);
      double* q = out + 10 * ((size_t)i * ns + s);
      const vec dn = normalise(r.d);
      q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = dn.x; q[4] = dn.y; q[5] = dn.z;
      q[6] = r.dead ? 0.0 : r.w; q[7] = r.w_pot; q[8] = (double)r.fragile; q[9] = (double)r.dead;
    }
  }
}
