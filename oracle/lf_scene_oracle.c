/* TEST INFRASTRUCTURE -- CPU restatement of the scene-radiance term of PathTracer::raytrace_pixel
 * (SURVEY.md section 8 row f2).  Never linked into, imported by or shipped with the product.
 *
 * Follows, expression by expression (CGL::Vector3D operator order included):
 *   the sample loop                         src/pathtracer/pathtracer.cpp:831-875
 *   Camera::generate_ray                    src/pathtracer/camera.cpp:278-305
 *   Sphere::test / intersect                src/scene/sphere.cpp:11-111
 *   moller_trumbore / Triangle::intersect   src/scene/triangle.cpp:25-112
 *   est_radiance_global_illumination        src/pathtracer/pathtracer.cpp:282-302
 *   estimate_direct_lighting_importance     src/pathtracer/pathtracer.cpp:142-213
 *   make_coord_space, DiffuseBSDF::f        src/pathtracer/bsdf.cpp:21-60
 *   DirectionalLight / PointLight::sample_L src/scene/light.cpp:18-24, :52-60
 * The closest hit is found by brute force over the primitives (the reference's BVH only prunes).
 * PARITY STATUS: pinned -- tests/test_oracle_vs_reference.py checks it against frames rendered by
 * the real reference from programmatic scenes (tests/golden/s*.npz).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lf_oracle.h"

typedef struct { double x, y, z; } v3;
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 scl(v3 a, double c) { return V(a.x * c, a.y * c, a.z * c); }     /* Vector3D * double */
static v3 lscl(double c, v3 a) { return V(c * a.x, c * a.y, c * a.z); }    /* double * Vector3D */
static v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static double dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } /* AVX: dp(x,y) + z */
static v3 cross(v3 u, v3 v) { return V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
static double norm(v3 a) { return sqrt(dot(a, a)); }
static v3 unit(v3 a) { double rn = 1. / norm(a); return scl(a, rn); }
static v3 divs(v3 a, double c) { double rc = 1.0 / c; return V(rc * a.x, rc * a.y, rc * a.z); }

typedef struct { v3 o, d; double min_t, max_t; } ray;
typedef struct { double t; v3 n; int mat; } isect;

typedef struct {
  int n_spheres; const double* spheres; const int* sph_mat;
  int n_tris; const double* tri_pos; const double* tri_nrm; const int* tri_mat;
  const double* materials; int n_lights; const double* lights;
} scene_t;

static int hit_sphere(const double* s, int mat, ray* r, isect* h) {
  v3 c = V(s[0], s[1], s[2]);
  double r2 = s[3] * s[3];
  v3 oc = sub(r->o, c);
  double a = dot(r->d, r->d), b = 2 * dot(oc, r->d), cc = dot(oc, oc) - r2, t1;
  if (b * b < 4.0 * a * cc) return 0;
  if (b * b == 4.0 * a * cc) {
    double root = (-b) / (2.0 * a);
    if (root < r->min_t || root > r->max_t) return 0;
    t1 = root;
  } else {
    double q = sqrt(b * b - 4.0 * a * cc);
    double r1 = (-b - q) / (2.0 * a), r2_ = (-b + q) / (2.0 * a);
    double p1 = r2_ < r1 ? r2_ : r1, p2 = r1 < r2_ ? r2_ : r1;
    if (p1 > r->max_t || p2 < r->min_t) return 0;
    if (p1 < r->min_t) { if (p2 > r->max_t) return 0; t1 = p2; } else t1 = p1;
  }
  r->max_t = t1;
  if (h) { h->t = t1; h->n = unit(sub(add(r->o, lscl(t1, r->d)), c)); h->mat = mat; }
  return 1;
}

static int hit_tri(const double* P, const double* N, int mat, ray* r, isect* h) {
  v3 p0 = V(P[0], P[1], P[2]), p1 = V(P[3], P[4], P[5]), p2 = V(P[6], P[7], P[8]);
  v3 e1 = sub(p1, p0), e2 = sub(p2, p0), s = sub(r->o, p0);
  v3 s1 = cross(r->d, e2), s2 = cross(s, e1);
  double rc = 1. / dot(s1, e1);
  double t = dot(s2, e2) * rc, b1 = dot(s1, s) * rc, b2 = dot(s2, r->d) * rc;
  if (t < r->min_t || t > r->max_t) return 0;
  if (b1 < 0 || b1 > 1) return 0;
  if (b2 < 0 || b2 > 1) return 0;
  if (b1 + b2 > 1) return 0;
  r->max_t = t;
  if (h) {
    double b0 = 1 - b1 - b2;
    v3 n1 = V(N[0], N[1], N[2]), n2 = V(N[3], N[4], N[5]), n3 = V(N[6], N[7], N[8]);
    h->t = t; h->n = unit(add(add(lscl(b0, n1), lscl(b1, n2)), lscl(b2, n3))); h->mat = mat;
  }
  return 1;
}

static int closest(const scene_t* S, ray* r, isect* h) {
  int any = 0;
  for (int i = 0; i < S->n_spheres; i++) any |= hit_sphere(S->spheres + 4 * i, S->sph_mat[i], r, h);
  for (int i = 0; i < S->n_tris; i++) any |= hit_tri(S->tri_pos + 9 * i, S->tri_nrm + 9 * i, S->tri_mat[i], r, h);
  return any;
}

static v3 radiance(const scene_t* S, ray r) {
  isect is;
  if (!closest(S, &r, &is)) return V(0, 0, 0);
  const double* m = S->materials + 4 * is.mat;
  v3 emission = m[0] == 1.0 ? V(m[1], m[2], m[3]) : V(0, 0, 0);
  /* make_coord_space */
  v3 z = is.n, hh = is.n;
  if (fabs(hh.x) <= fabs(hh.y) && fabs(hh.x) <= fabs(hh.z)) hh.x = 1.0;
  else if (fabs(hh.y) <= fabs(hh.x) && fabs(hh.y) <= fabs(hh.z)) hh.y = 1.0;
  else hh.z = 1.0;
  z = scl(z, 1. / norm(z));
  v3 y = cross(hh, z); y = scl(y, 1. / norm(y));
  v3 x = cross(z, y); x = scl(x, 1. / norm(x));
  v3 hit_p = add(r.o, scl(r.d, is.t));
  v3 L = V(0, 0, 0);
  const double eps = (double)0.00001f;
  for (int l = 0; l < S->n_lights; l++) {
    const double* lt = S->lights + 7 * l;
    v3 wi; double dist;
    if (lt[0] == 0.0) { wi = V(lt[1], lt[2], lt[3]); dist = INFINITY; }
    else { v3 d = sub(V(lt[1], lt[2], lt[3]), hit_p); wi = unit(d); dist = norm(d); }
    v3 wo = V((wi.x * x.x + wi.y * x.y) + wi.z * x.z, (wi.x * y.x + wi.y * y.y) + wi.z * y.z,
              (wi.x * z.x + wi.y * z.y) + wi.z * z.z);
    if (wo.z < 0) continue;
    ray sh = {hit_p, wi, eps, dist - eps};
    if (!closest(S, &sh, NULL)) {
      double cos_theta = unit(wo).z;
      double ipi = 1.0 / 3.14159265358979323;
      v3 f = m[0] == 0.0 ? mulv(V(ipi, ipi, ipi), V(m[1], m[2], m[3])) : V(0, 0, 0);
      L = add(L, divs(scl(mulv(f, V(lt[4], lt[5], lt[6])), cos_theta), 1.0));
    }
  }
  if (S->n_lights > 0) L = divs(L, (double)S->n_lights);
  return add(emission, L);
}

/* scene: W*H*3 doubles; visited pixels get the averaged radiance (sum / loop variable).  Draws per
 * visited pixel: 2*ns_aa (consumed here) + 32 (falloff), one std::mt19937 in visit order. */
void lfo_scene_term(int W, int H, int ns_aa, int samples_per_batch, double max_tol,
                    const double c2w[9], const double pos[3], double hfov, double vfov, double nclip,
                    double fclip, int n_spheres, const double* spheres, const int* sph_mat, int n_tris,
                    const double* tri_pos, const double* tri_nrm, const int* tri_mat,
                    const double* materials, int n_lights, const double* lights,
                    const uint32_t* order, size_t n_order, uint32_t seed, double* scene) {
  scene_t S = {n_spheres, spheres, sph_mat, n_tris, tri_pos, tri_nrm, tri_mat, materials, n_lights, lights};
  size_t per = 2 * (size_t)ns_aa + 32;
  uint32_t* raw = (uint32_t*)malloc(sizeof(uint32_t) * per * n_order);
  lfo_mt19937_raw(seed, 0, per * n_order, raw);
  const double PI_ = 3.14159265358979323;
  double edge_x = tan(0.5 * (hfov * (PI_ / 180.0))), edge_y = tan(0.5 * (vfov * (PI_ / 180.0)));
  for (size_t v = 0; v < n_order; v++) {
    size_t p = order[v];
    int px = (int)(p % (size_t)W), py = (int)(p / (size_t)W);
    const uint32_t* rw = raw + per * v;
    v3 total = V(0, 0, 0);
    float s1 = 0.0f, s2 = 0.0f;
    int sample;
    for (sample = 1; sample <= ns_aa; sample++) {
      /* first draw -> y (g++ evaluates Vector2D(random_uniform(), random_uniform()) right to left) */
      double sy = (double)py + lfo_random_uniform_from_raw(rw[2 * (sample - 1)]);
      double sx = (double)px + lfo_random_uniform_from_raw(rw[2 * (sample - 1) + 1]);
      double nx = sx / (double)W, ny = sy / (double)H;
      v3 dir = unit(V(edge_x * (2 * nx - 1), edge_y * (2 * ny - 1), -1));
      ray r;
      r.o = V(pos[0], pos[1], pos[2]);
      r.d = V((dir.x * c2w[0] + dir.y * c2w[1]) + dir.z * c2w[2], (dir.x * c2w[3] + dir.y * c2w[4]) + dir.z * c2w[5],
              (dir.x * c2w[6] + dir.y * c2w[7]) + dir.z * c2w[8]);
      r.min_t = nclip; r.max_t = fclip;
      v3 L = radiance(&S, r);
      float illum = (float)((0.2126f * L.x + 0.7152f * L.y) + 0.0722f * L.z);
      s1 += illum; s2 += illum * illum;
      total = add(total, L);
      if (sample > 1 && sample % samples_per_batch == 0) {
        float sd = (float)sqrt(1.0 / (sample - 1) * (double)(s2 - s1 * s1 / (float)sample));
        float ci = (float)(1.96 * (double)sd / sqrt((double)sample));
        if ((double)ci <= max_tol * (double)s1 / (double)sample) break;
      }
    }
    double rc = 1. / (double)sample;
    scene[3 * p] = total.x * rc; scene[3 * p + 1] = total.y * rc; scene[3 * p + 2] = total.z * rc;
  }
  free(raw);
}

/* est_radiance_global_illumination (pathtracer.cpp:282-302) for explicit rays: what the sample loop adds
 * for a camera ray, whoever generated it (the lens camera's exit rays, tests/test_gpu_lens_camera.py).
 * rays: n x 8 doubles {o xyz, d xyz, min_t, max_t}; rgb: n x 3 */
void lfo_scene_radiance_rays(int n_spheres, const double* spheres, const int* sph_mat, int n_tris,
                             const double* tri_pos, const double* tri_nrm, const int* tri_mat,
                             const double* materials, int n_lights, const double* lights, size_t n_rays,
                             const double* rays, double* rgb) {
  scene_t S = {n_spheres, spheres, sph_mat, n_tris, tri_pos, tri_nrm, tri_mat, materials, n_lights, lights};
  for (size_t i = 0; i < n_rays; i++) {
    const double* q = rays + 8 * i;
    ray r = {V(q[0], q[1], q[2]), V(q[3], q[4], q[5]), q[6], q[7]};
    v3 L = radiance(&S, r);
    rgb[3 * i] = L.x; rgb[3 * i + 1] = L.y; rgb[3 * i + 2] = L.z;
  }
}
