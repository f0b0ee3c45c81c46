// lf_scene.hip -- the scene-radiance term of PathTracer::raytrace_pixel on the device
// (SURVEY.md section 8 row f2): the per-pixel sample loop (pathtracer.cpp:841-875), pinhole ray
// generation (camera.cpp:278-305), closest-hit search over spheres and triangles
// (scene/sphere.cpp:11-111, scene/triangle.cpp:25-112, scene/bvh.cpp:201-222), emission +
// direct lighting of diffuse surfaces by delta lights with shadow rays
// (pathtracer.cpp:142-232, bsdf.cpp:21-60, light.cpp:11-24, :47-60).
//
// Everything is double precision and follows the reference expression by expression (operator
// order of CGL::Vector3D / Matrix3x3 included) so that pixels agree with the CPU renderer far
// inside the 1e-4 bar.  The BVH is our own (median split on the host, stack traversal on the
// device): the closest hit does not depend on the tree, only the amount of work does.
//
// Sampled lights -- AreaLight and InfiniteHemisphereLight (scene/light.cpp:35-48, :82-101) with
// ns_area_light samples each, exactly the estimator of estimate_direct_lighting_importance
// (pathtracer.cpp:143-213) -- draw from the order-free Philox counter RNG: the reference draws them
// from its shared MT19937 only when a camera ray hits something, which makes every later pixel's
// jitter depend on every earlier pixel's hits, so no device schedule can reproduce its stream.
// They are therefore validated statistically against frames the reference rendered
// (tests/test_gpu_area_lights.py) and refused in MT19937 parity mode.
// Not covered (the call fails loudly rather than approximating): environment maps, uniform
// hemisphere sampling of emitters (-H), spot lights (a stub in the reference, light.cpp:64-72) and
// the Mirror/Glass/Microfacet BSDFs, which are unfilled stubs in the reference (advanced_bsdf.cpp).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "lf_internal.h"

namespace {

struct V3 { double x, y, z; };
__host__ __device__ inline V3 v3(double x, double y, double z) { V3 r{x, y, z}; return r; }
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline V3 operator*(V3 a, double c) { return v3(a.x * c, a.y * c, a.z * c); }
__host__ __device__ inline V3 operator*(double c, V3 a) { return v3(c * a.x, c * a.y, c * a.z); }
__host__ __device__ inline V3 mulv(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
// dot(): (x*x' + y*y') + z*z' (the AVX build's _mm_dp_pd pairs x,y first; vector3D.h:256-262)
__host__ __device__ inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ inline V3 cross(V3 u, V3 v) {
  return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ inline double norm(V3 a) { return sqrt(dot(a, a)); }
// Vector3D::unit(): multiply by 1/norm (vector3D.h:214-217); operator/(double) does the same
__device__ inline V3 unit(V3 a) { double rn = 1. / norm(a); return a * rn; }
__device__ inline V3 divs(V3 a, double c) { const double rc = 1.0 / c; return v3(rc * a.x, rc * a.y, rc * a.z); }

struct DRay { V3 o, d; double min_t, max_t; };
struct Hit { double t; V3 n; int material; };

// ---- primitives ------------------------------------------------------------------------------
// Sphere::test + intersect (scene/sphere.cpp:11-111)
__device__ inline bool hit_sphere(const LfPrim& s, DRay& r, Hit* h) {
  const V3 c = v3(s.d[0], s.d[1], s.d[2]);
  const V3 oc = r.o - c;
  const double a = dot(r.d, r.d);
  const double b = 2 * dot(oc, r.d);
  const double cc = dot(oc, oc) - s.d[4];
  double t1;
  if (b * b < 4.0 * a * cc) return false;
  if (b * b == 4.0 * a * cc) {
    const double root = (-b) / (2.0 * a);
    if (root < r.min_t || root > r.max_t) return false;
    t1 = root;
  } else {
    const double q = sqrt(b * b - 4.0 * a * cc);
    const double r1 = (-b - q) / (2.0 * a), r2 = (-b + q) / (2.0 * a);
    const double p1 = r2 < r1 ? r2 : r1, p2 = r1 < r2 ? r2 : r1;  // std::min / std::max
    if (p1 > r.max_t || p2 < r.min_t) return false;
    if (p1 < r.min_t) {
      if (p2 > r.max_t) return false;
      t1 = p2;
    } else {
      t1 = p1;
    }
  }
  r.max_t = t1;
  if (h) {
    h->t = t1;
    h->n = unit((r.o + t1 * r.d) - c);  // Sphere::normal (sphere.h:73-75)
    h->material = s.material;
  }
  return true;
}

// moller_trumbore + is_valid_intersection + Triangle::intersect (scene/triangle.cpp:25-112)
__device__ inline bool hit_triangle(const LfPrim& t, DRay& r, Hit* h) {
  const V3 p0 = v3(t.d[0], t.d[1], t.d[2]), p1 = v3(t.d[3], t.d[4], t.d[5]), p2 = v3(t.d[6], t.d[7], t.d[8]);
  const V3 e1 = p1 - p0, e2 = p2 - p0, s = r.o - p0;
  const V3 s1 = cross(r.d, e2), s2 = cross(s, e1);
  const double rc = 1. / dot(s1, e1);  // operator/= multiplies by the reciprocal
  const double tt = dot(s2, e2) * rc, b1 = dot(s1, s) * rc, b2 = dot(s2, r.d) * rc;
  if (tt < r.min_t || tt > r.max_t) return false;
  if (b1 < 0 || b1 > 1) return false;
  if (b2 < 0 || b2 > 1) return false;
  if (b1 + b2 > 1) return false;
  r.max_t = tt;
  if (h) {
    const double b0 = 1 - b1 - b2;
    const V3 n1 = v3(t.d[9], t.d[10], t.d[11]), n2 = v3(t.d[12], t.d[13], t.d[14]),
             n3 = v3(t.d[15], t.d[16], t.d[17]);
    h->t = tt;
    h->n = unit((b0 * n1 + b1 * n2) + b2 * n3);
    h->material = t.material;
  }
  return true;
}

// closest hit (BVHAccel::intersect, scene/bvh.cpp:201-222: the recursion shrinks r.max_t as it goes,
// so whatever the visiting order the last accepted primitive is the closest one).  h == nullptr is
// the shadow-ray query (has_intersection): the first accepted primitive settles it.
// The traversal stack lives in LDS ([depth][thread]: conflict-free), not in scratch memory; the slab
// test multiplies by the reciprocal direction (3 f64 divisions per ray instead of 6 per node) and is
// widened by 2 ulp so that it can only be MORE permissive than the exact quotients -- the box test
// is pure culling, every accepted leaf still runs the reference's exact primitive tests.
constexpr int kStackDepth = 32;
__device__ bool closest_hit(const LfBvhNode* __restrict__ nodes, const LfPrim* __restrict__ prims,
                            DRay& r, Hit* h, int* __restrict__ stack /* [kStackDepth][256] + tid */) {
  int sp = 0;
  stack[256 * sp++] = 0;
  bool any = false;
  const double ix = 1.0 / r.d.x, iy = 1.0 / r.d.y, iz = 1.0 / r.d.z;
  while (sp > 0) {
    const LfBvhNode& nd = nodes[stack[256 * --sp]];
    // BBox::intersect (scene/bbox.cpp:12-49); fmin/fmax drop NaNs (0 * inf on a slab plane), which
    // only makes the test more permissive
    const double tx1 = (nd.bmin[0] - r.o.x) * ix, tx2 = (nd.bmax[0] - r.o.x) * ix;
    const double ty1 = (nd.bmin[1] - r.o.y) * iy, ty2 = (nd.bmax[1] - r.o.y) * iy;
    const double tz1 = (nd.bmin[2] - r.o.z) * iz, tz2 = (nd.bmax[2] - r.o.z) * iz;
    const double tmin = fmax(fmax(fmin(tx1, tx2), fmin(ty1, ty2)), fmin(tz1, tz2));
    const double tmax = fmin(fmin(fmax(tx1, tx2), fmax(ty1, ty2)), fmax(tz1, tz2));
    // (an infinite bound -- a ray parallel to a slab and outside it has tmin = +inf -- must not turn
    // the slack into inf: inf - inf compares false with everything and the box would never be culled;
    // such rays, e.g. the texel-grid directions of an environment light, then walk the whole tree)
    const double tm = fmax(fabs(tmin), fabs(tmax));
    const double slack = tm < 1e300 ? 4.5e-16 * tm : 0.0;
    if (tmin - slack > tmax + slack || tmax + slack < r.min_t || tmin - slack > r.max_t) continue;
    if (nd.count > 0) {
      for (int i = 0; i < nd.count; i++) {
        const LfPrim& p = prims[nd.first + i];
        const bool hit = p.type == 0 ? hit_sphere(p, r, h) : hit_triangle(p, r, h);
        any = any || hit;
      }
      if (any && !h) return true;
    } else {
      // (left < 0: the childless node of an empty scene, should a ray ever pass its degenerate box)
      if (nd.left >= 0 && sp < kStackDepth - 1) { stack[256 * sp++] = nd.right; stack[256 * sp++] = nd.left; }
    }
  }
  return any;
}

// make_coord_space (pathtracer/bsdf.cpp:21-41): returns the rows of w2o = o2w.T()
__device__ inline void make_coord_space(V3 n, V3& X, V3& Y, V3& Z) {
  V3 z = n, hh = n;
  if (fabs(hh.x) <= fabs(hh.y) && fabs(hh.x) <= fabs(hh.z)) hh.x = 1.0;
  else if (fabs(hh.y) <= fabs(hh.x) && fabs(hh.y) <= fabs(hh.z)) hh.y = 1.0;
  else hh.z = 1.0;
  z = z * (1. / norm(z));            // normalize(): *= 1/norm
  V3 y = cross(hh, z);
  y = y * (1. / norm(y));
  V3 x = cross(z, y);
  x = x * (1. / norm(x));
  X = x; Y = y; Z = z;
}

// est_radiance_global_illumination (pathtracer.cpp:282-302) = zero_bounce + one_bounce with
// estimate_direct_lighting_importance (:142-213)
__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key);
__device__ inline double random_uniform_from_raw(unsigned raw);

// EnvironmentLight::sample_dir (environment_light.cpp:173-182) = bilerp(theta_phi_to_xy(
// dir_to_theta_phi(r.d))) (:86-91, :102-107, :121-138)
__device__ V3 env_sample_dir(const LfEnvDev& ev, V3 d) {
  const double PI_ = 3.14159265358979323;
  const V3 u = unit(d);
  const double theta = acos(u.y), phi = atan2(-u.z, u.x) + PI_;
  const double x = phi / 2. / PI_ * (double)ev.w, y = theta / PI_ * (double)ev.h;
  long right = lround(x), left, v = lround(y);
  const double u1 = (double)right - x + .5;
  double v1;
  if (right == 0 || right == ev.w) { left = ev.w - 1; right = 0; } else left = right - 1;
  if (v == 0) { v = 1; v1 = 1.0; } else if (v == ev.h) { v = ev.h - 1; v1 = 0.0; } else v1 = (double)v - y + .5;
  const long bottom = (long)ev.w * v, top = bottom - ev.w;
  const double u0 = 1 - u1;
  auto px = [&](long i) { return v3(ev.data[3 * i], ev.data[3 * i + 1], ev.data[3 * i + 2]); };
  return (px(top + left) * u1 + px(top + right) * u0) * v1 +
         (px(bottom + left) * u1 + px(bottom + right) * u0) * (1 - v1);
}

// std::upper_bound on a non-decreasing table: the first index whose entry is > x (n if none)
__device__ inline int upper_bound_d(const double* __restrict__ a, int n, double x) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (x < a[mid]) hi = mid; else lo = mid + 1; }
  return lo;
}

// SOFT = false is the kernel of scenes with delta lights only (no sampled light, no environment, no
// hemisphere sampling): the sampled-light code costs ~100 vector registers, i.e. one of the three
// waves a SIMD otherwise holds (25 -> 35 ms on the 1080p timing frame), so it is compiled out there.
// what: bit 0 = zero_bounce_radiance (:215-220, the hit surface's emission), bit 1 =
// one_bounce_radiance (:222-232: the hemisphere or the importance estimator)
enum { kShadeZero = 1, kShadeOne = 2 };
template <bool SOFT>
__device__ V3 shade_hit(const LfSceneDev& sc, const LfEnvDev& ev, bool hemisphere, const DRay& r,
                        double isect_t, V3 isect_n, const LfMaterial& m, int what,
                        int* __restrict__ stack, int ns_area_light, uint4 rng_ctr, uint2 rng_key) {
  const V3 emission = (m.kind == 1 && (what & kShadeZero)) ? v3(m.rgb[0], m.rgb[1], m.rgb[2]) : v3(0, 0, 0);
  if (!(what & kShadeOne)) return emission;
  V3 X, Y, Z;
  make_coord_space(isect_n, X, Y, Z);
  const V3 hit_p = r.o + r.d * isect_t;
  V3 L = v3(0, 0, 0);
  const double kEpsF = (double)0.00001f;  // EPS_F (misc.h:13)
  int total_samples = 0;
  const double ipi = 1.0 / 3.14159265358979323;
  // DiffuseBSDF::f = Vector3D(1/PI) * reflectance (bsdf.cpp:52-60); EmissionBSDF::f = 0
  // (kind 2: the value of f itself, as a host that can only call BSDF::f hands it over)
  const V3 f = m.kind == 0 ? mulv(v3(ipi, ipi, ipi), v3(m.rgb[0], m.rgb[1], m.rgb[2]))
             : m.kind == 2 ? v3(m.rgb[0], m.rgb[1], m.rgb[2]) : v3(0, 0, 0);
  if (SOFT && hemisphere) {
    // estimate_direct_lighting_hemisphere (pathtracer.cpp:86-138): uniform directions over the
    // hemisphere of the hit point, lights.size() * ns_area_light of them; what they find is the
    // EMISSION of whatever surface they hit (lights as such are not sampled, the environment not seen)
    const int num_samples = sc.n_lights * ns_area_light;
    const double p_w = 1.0 / (2.0 * 3.14159265358979323);
    for (int k = 0; k < num_samples; k++) {
      const uint4 rr = philox4x32_10(make_uint4(rng_ctr.x, rng_ctr.y, 0x11650000u, (unsigned)k), rng_key);
      const double xi1 = random_uniform_from_raw(rr.x), xi2 = random_uniform_from_raw(rr.y);
      // UniformHemisphereSampler3D::get_sample (sampler.cpp:30-44): sinf / cosf of the double angles
      const double theta = acos(xi1), phi = 2.0 * 3.14159265358979323 * xi2;
      const V3 wi = v3((double)(sinf((float)theta) * cosf((float)phi)),
                       (double)(sinf((float)theta) * sinf((float)phi)), (double)cosf((float)theta));
      // o2w * wi: the columns of o2w are X, Y, Z
      const V3 ww = v3((wi.x * X.x + wi.y * Y.x) + wi.z * Z.x, (wi.x * X.y + wi.y * Y.y) + wi.z * Z.y,
                       (wi.x * X.z + wi.y * Y.z) + wi.z * Z.z);
      DRay out{hit_p, ww, kEpsF, INFINITY};
      Hit h2;
      if (closest_hit(sc.nodes, sc.prims, out, &h2, stack)) {
        const LfMaterial& m2 = sc.materials[h2.material];
        const V3 em2 = m2.kind == 1 ? v3(m2.rgb[0], m2.rgb[1], m2.rgb[2]) : v3(0, 0, 0);
        const double cos_theta = unit(wi).z;
        L = L + divs(mulv(f, em2) * cos_theta, p_w);
      }
    }
    return emission + divs(L, (double)num_samples);   // (0 / 0 = NaN without lights, as in the reference)
  }
  for (int l = 0; l < sc.n_lights; l++) {
    const LfLight& lt = sc.lights[l];
    if (!SOFT && lt.type >= 2) continue;   // (never listed when this instantiation is launched)
    const int num_samples = lt.type >= 2 ? ns_area_light : 1;   // is_delta_light() ? 1 : ns_area_light
    total_samples += num_samples;
    for (int k = 0; k < num_samples; k++) {
      V3 wi;
      double dist, pdf = 1.0;
      V3 emit = v3(lt.rgb[0], lt.rgb[1], lt.rgb[2]);
      if (lt.type == 0) {  // DirectionalLight::sample_L (light.cpp:18-24)
        wi = v3(lt.v[0], lt.v[1], lt.v[2]);
        dist = INFINITY;
      } else if (lt.type == 1) {  // PointLight::sample_L (light.cpp:52-60)
        const V3 d = v3(lt.v[0], lt.v[1], lt.v[2]) - hit_p;
        wi = unit(d);
        dist = norm(d);
      } else if (SOFT) {
        const uint4 rr = philox4x32_10(make_uint4(rng_ctr.x, rng_ctr.y, 0x11640000u + (unsigned)l, (unsigned)k),
                                       rng_key);
        const double xi1 = random_uniform_from_raw(rr.x), xi2 = random_uniform_from_raw(rr.y);
        if (lt.type == 2) {  // InfiniteHemisphereLight::sample_L (light.cpp:35-48)
          const double theta = acos(xi1), phi = 2.0 * 3.14159265358979323 * xi2;
          const double xs = sin(theta) * cos(phi), ys = sin(theta) * sin(phi), zs = cos(theta);
          wi = v3(xs, zs, -ys);   // sampleToWorld: columns (1,0,0), (0,0,-1), (0,1,0)
          dist = INFINITY;
          pdf = 1.0 / (2.0 * 3.14159265358979323);
        } else if (lt.type == 4) {  // EnvironmentLight::sample_L, importance sampled (:159-171)
          // (upper_bound can return one past the end when the draw exceeds the table's last entry,
          // ~1 - 1e-16 against draws clamped to 0.99999999: out of bounds in the reference, clamped here)
          int yy = upper_bound_d(ev.marginal, ev.h, xi2);
          yy = yy < ev.h ? yy : ev.h - 1;
          int xx = upper_bound_d(ev.conds + (size_t)ev.w * yy, ev.w, xi1);
          xx = xx < ev.w ? xx : ev.w - 1;
          const double PI_ = 3.14159265358979323;
          const double phi = (double)xx / (double)ev.w * 2.0 * PI_, theta = (double)yy / (double)ev.h * PI_;
          wi = v3(cos(phi - PI_) * sin(theta), cos(theta), -sin(phi - PI_) * sin(theta));
          dist = INFINITY;
          const size_t t = (size_t)ev.w * yy + xx;
          pdf = ev.pdf[t] * (double)ev.w * (double)ev.h / 2. / PI_ / PI_ / sin(theta);
          emit = v3(ev.data[3 * t], ev.data[3 * t + 1], ev.data[3 * t + 2]);
        } else {             // AreaLight::sample_L (light.cpp:82-101)
          const double sx = xi1 - (double)0.5f, sy = xi2 - (double)0.5f;
          const V3 d = ((v3(lt.v[0], lt.v[1], lt.v[2]) + sx * v3(lt.dim_x[0], lt.dim_x[1], lt.dim_x[2])) +
                        sy * v3(lt.dim_y[0], lt.dim_y[1], lt.dim_y[2])) - hit_p;
          const double cos_l = dot(d, v3(lt.dir[0], lt.dir[1], lt.dir[2]));
          const double sq = dot(d, d);
          dist = sqrt(sq);
          wi = divs(d, dist);
          pdf = sq / (lt.area * fabs(cos_l));
          if (!(cos_l < 0)) emit = v3(0, 0, 0);   // the light shines to one side only
        }
      }
      // w2o * wi: rows of w2o are the columns of o2w; Matrix3x3 * Vector3D sums column-wise:
      // wi.x*col0 + wi.y*col1 + wi.z*col2 of w2o, i.e. component k = (wi.x*R0[k] + wi.y*R1[k]) + wi.z*R2[k]
      const V3 wo = v3((wi.x * X.x + wi.y * X.y) + wi.z * X.z, (wi.x * Y.x + wi.y * Y.y) + wi.z * Y.z,
                       (wi.x * Z.x + wi.y * Z.y) + wi.z * Z.z);
      if (wo.z < 0) continue;
      DRay sh{hit_p, wi, kEpsF, dist - kEpsF};
      if (!closest_hit(sc.nodes, sc.prims, sh, nullptr, stack)) {
        const double cos_theta = unit(wo).z;
        L = L + divs(mulv(f, emit) * cos_theta, pdf);  // / pdf (1 for delta lights)
      }
    }
  }
  if (total_samples > 0) L = divs(L, (double)total_samples);  // L_out / total_samples (:211)
  return emission + L;
}

template <bool SOFT>
__device__ V3 radiance(const LfSceneDev& sc, const LfEnvDev& ev, bool hemisphere, DRay r,
                       int* __restrict__ stack, int ns_area_light, uint4 rng_ctr, uint2 rng_key) {
  Hit isect;
  if (!closest_hit(sc.nodes, sc.prims, r, &isect, stack))   // pathtracer.cpp:291-292
    return (SOFT && ev.w) ? env_sample_dir(ev, r.d) : v3(0, 0, 0);
  return shade_hit<SOFT>(sc, ev, hemisphere, r, isect.t, isect.n, sc.materials[isect.material],
                         kShadeZero | kShadeOne, stack, ns_area_light, rng_ctr, rng_key);
}

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

__device__ inline double random_uniform_from_raw(unsigned raw) {  // util/random_util.h:15-22
  double v = (double)raw * (1.0 / (4294967295.0 - 0.0));
  v = v < 0.0000001 ? 0.0000001 : v;
  v = 0.99999999 < v ? 0.99999999 : v;
  return v;
}

// the sample loop of raytrace_pixel (pathtracer.cpp:831-875)
// (second bound: 4 waves per SIMD = 128 registers.  Left alone the sampled-light instantiation takes
// 246 and runs 2 waves; bounded it spills ~120 registers to scratch in its cold paths and is still
// 30 % faster -- timing frame: 155 -> 110 ms with an area light and the environment, 25.4 -> 22.5 ms
// with delta lights only; profiles/r02_scene_term_timing.json)
template <bool SOFT>
__global__ __launch_bounds__(256, 4) void k_scene_term(LfSceneDev sc, LfEnvDev ev, int hemisphere,
                                                    LfCamera cam, int W, int H, int y0,
                                                    int y1, int row_phase, int row_period,
                                                    int ns_aa, int ns_area_light,
                                                    int samples_per_batch, double max_tolerance,
                                                    const uint32_t* __restrict__ aa_raw, int jitter_mode,
                                                    uint64_t key, double* __restrict__ scene) {
  // traversal stacks of the 256 threads: the median-split tree of n primitives is ceil(log2(n / 4))
  // deep (<= 29), the stack holds at most depth + 1 entries
  __shared__ int s_stack[kStackDepth * 256];
  int* const stack = s_stack + threadIdx.x;
  const size_t p = (size_t)y0 * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)y1 * W) return;
  const int x = (int)(p % W), y = (int)(p / W);
  // multi-GPU: only the 8-row tile rows this context owns, like k_flare_layer, which is the only
  // reader of this buffer (whole waves leave: 64 consecutive pixels share a tile row unless W < 64)
  if (row_period > 1 && (y >> 3) % row_period != row_phase) return;
  const double PI_ = 3.14159265358979323;
  const double edge_x = tan(0.5 * (cam.hfov_deg * (PI_ / 180.0)));
  const double edge_y = tan(0.5 * (cam.vfov_deg * (PI_ / 180.0)));
  V3 total = v3(0, 0, 0);
  float s1 = 0.0f, s2 = 0.0f;
  int sample;
  for (sample = 1; sample <= ns_aa; sample++) {
    unsigned ra, rb;
    if (jitter_mode == 0) {
      ra = aa_raw[p * (size_t)(2 * ns_aa) + 2 * (sample - 1)];
      rb = aa_raw[p * (size_t)(2 * ns_aa) + 2 * (sample - 1) + 1];
    } else {
      const uint4 r4 = philox4x32_10(make_uint4((unsigned)p, (unsigned)sample, 0x5ce4e000u, 0u),
                                     make_uint2((unsigned)key, (unsigned)(key >> 32)));
      ra = r4.x; rb = r4.y;
    }
    // Vector2D(random_uniform(), random_uniform()): g++ evaluates right to left, the first draw is y
    const double sy = (double)y + random_uniform_from_raw(ra);
    const double sx = (double)x + random_uniform_from_raw(rb);
    const double nx = sx / (double)W, ny = sy / (double)H;
    // Camera::generate_ray (camera.cpp:278-305)
    V3 dir = unit(v3(edge_x * (2 * nx - 1), edge_y * (2 * ny - 1), -1));
    DRay r;
    r.o = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    r.d = v3((dir.x * cam.c2w[0] + dir.y * cam.c2w[1]) + dir.z * cam.c2w[2],
             (dir.x * cam.c2w[3] + dir.y * cam.c2w[4]) + dir.z * cam.c2w[5],
             (dir.x * cam.c2w[6] + dir.y * cam.c2w[7]) + dir.z * cam.c2w[8]);
    r.min_t = cam.n_clip; r.max_t = cam.f_clip;
    const V3 L = radiance<SOFT>(sc, ev, hemisphere != 0, r, stack, ns_area_light, make_uint4((unsigned)p, (unsigned)sample, 0u, 0u),
                          make_uint2((unsigned)key, (unsigned)(key >> 32)));
    // Vector3D::illum (vector3D.h:231-233): float coefficients, double arithmetic, float result
    const float illum = (float)((0.2126f * L.x + 0.7152f * L.y) + 0.0722f * L.z);
    s1 += illum;
    s2 += illum * illum;
    total = total + L;
    if (sample > 1 && sample % samples_per_batch == 0) {  // :862-868
      const float sd = (float)sqrt(1.0 / (sample - 1) * (double)(s2 - s1 * s1 / (float)sample));
      const float ci = (float)(1.96 * (double)sd / sqrt((double)sample));
      if ((double)ci <= max_tolerance * (double)s1 / (double)sample) break;
    }
  }
  const double rc = 1. / (double)sample;  // :875 -- ns_aa + 1 when the loop ran to its end
  scene[3 * p] = total.x * rc;
  scene[3 * p + 1] = total.y * rc;
  scene[3 * p + 2] = total.z * rc;
}

// ---- single-ray forms of the integrator's public members (pathtracer.h:66-77) ----------------
struct LfProbeRay { double o[3], d[3], min_t, max_t; };
// est_radiance_global_illumination(r) (:282-302) and the closest hit behind autofocus (:1065-1072):
// out = {hit, t, n xyz, radiance rgb}
__global__ void k_scene_trace_ray(LfSceneDev sc, LfEnvDev ev, int hemisphere, LfProbeRay pr,
                                  int ns_area_light, uint64_t seq, uint64_t key, double* __restrict__ out) {
  __shared__ int s_stack[kStackDepth * 256];
  if (threadIdx.x != 0) return;
  DRay r{v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.min_t, pr.max_t};
  const uint4 ctr = make_uint4((unsigned)seq, (unsigned)(seq >> 32) | 0x80000000u, 0u, 0u);
  const uint2 k2 = make_uint2((unsigned)key, (unsigned)(key >> 32));
  Hit h;
  V3 L;
  if (closest_hit(sc.nodes, sc.prims, r, &h, s_stack)) {
    out[0] = 1.0; out[1] = h.t; out[2] = h.n.x; out[3] = h.n.y; out[4] = h.n.z;
    L = shade_hit<true>(sc, ev, hemisphere != 0, r, h.t, h.n, sc.materials[h.material], kShadeZero | kShadeOne,
                        s_stack, ns_area_light, ctr, k2);
  } else {
    out[0] = 0.0; out[1] = out[2] = out[3] = out[4] = 0.0;
    L = ev.w ? env_sample_dir(ev, r.d) : v3(0, 0, 0);
  }
  out[5] = L.x; out[6] = L.y; out[7] = L.z;
}
// zero_bounce_radiance / one_bounce_radiance / estimate_direct_lighting_{hemisphere, importance}
// (r, isect) for an intersection the host found itself
__global__ void k_scene_shade(LfSceneDev sc, LfEnvDev ev, int hemisphere, LfProbeRay pr, double t,
                              double nx, double ny, double nz, LfMaterial m, int what,
                              int ns_area_light, uint64_t seq, uint64_t key, double* __restrict__ out) {
  __shared__ int s_stack[kStackDepth * 256];
  if (threadIdx.x != 0) return;
  const DRay r{v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.min_t, pr.max_t};
  const V3 L = shade_hit<true>(sc, ev, hemisphere != 0, r, t, v3(nx, ny, nz), m, what, s_stack, ns_area_light,
                               make_uint4((unsigned)seq, (unsigned)(seq >> 32) | 0x80000000u, 0u, 0u),
                               make_uint2((unsigned)key, (unsigned)(key >> 32)));
  out[0] = L.x; out[1] = L.y; out[2] = L.z;
}

// ---- host: BVH over the primitives (median split of the centroids along the widest axis) -----
struct Box { double mn[3], mx[3]; };

Box prim_box(const LfPrim& p) {
  Box b;
  if (p.type == 0) {
    for (int a = 0; a < 3; a++) { b.mn[a] = p.d[a] - p.d[3]; b.mx[a] = p.d[a] + p.d[3]; }
  } else {
    for (int a = 0; a < 3; a++) {
      b.mn[a] = std::min({p.d[a], p.d[3 + a], p.d[6 + a]});
      b.mx[a] = std::max({p.d[a], p.d[3 + a], p.d[6 + a]});
    }
  }
  return b;
}

// returns the node id; *depth = levels below (and including) this node
int build_node(std::vector<LfBvhNode>& nodes, std::vector<LfPrim>& prims, int first, int count,
               int* depth) {
  LfBvhNode nd;
  for (int a = 0; a < 3; a++) { nd.bmin[a] = INFINITY; nd.bmax[a] = -INFINITY; }
  double cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = first; i < first + count; i++) {
    Box b = prim_box(prims[i]);
    for (int a = 0; a < 3; a++) {
      nd.bmin[a] = std::min(nd.bmin[a], b.mn[a]); nd.bmax[a] = std::max(nd.bmax[a], b.mx[a]);
      double c = 0.5 * (b.mn[a] + b.mx[a]);
      cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c);
    }
  }
  nd.left = nd.right = -1; nd.first = first; nd.count = count;
  const int id = (int)nodes.size();
  nodes.push_back(nd);
  if (depth) *depth = 1;
  if (count <= 4) return id;
  int axis = 0;
  for (int a = 1; a < 3; a++) if (cmx[a] - cmn[a] > cmx[axis] - cmn[axis]) axis = a;
  const int mid = first + count / 2;
  std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                   [axis](const LfPrim& a, const LfPrim& b) {
                     Box ba = prim_box(a), bb = prim_box(b);
                     return ba.mn[axis] + ba.mx[axis] < bb.mn[axis] + bb.mx[axis];
                   });
  int dl = 0, dr = 0;
  const int l = build_node(nodes, prims, first, mid - first, &dl);
  const int r = build_node(nodes, prims, mid, first + count - mid, &dr);
  nodes[id].left = l; nodes[id].right = r; nodes[id].count = 0;
  if (depth) *depth = 1 + std::max(dl, dr);
  return id;
}

// one wave writes `n_out` doubles, which come straight back
template <typename Launch>
static lf_status scene_probe(lf_ctx* ctx, int n_out, double* out, Launch launch) {
  double* d = nullptr;
  LF_HIP(ctx, hipMalloc((void**)&d, sizeof(double) * (size_t)n_out));
  launch(d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  LF_HIP(ctx, e);
  return LF_OK;
}

static lf_status scene_probe_ready(lf_ctx* ctx, const char* who) {
  if (!ctx->scene_valid) return lf_fail(ctx, LF_ERR_STATE, std::string(who) + " before lf_set_scene");
  if (ctx->scene_dev.n_env_lights > 0 && ctx->env_dev.w == 0)
    return lf_fail(ctx, LF_ERR_STATE, "an environment light (type 4) is listed but no map is set (lf_set_environment_map)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  return LF_OK;
}

}  // namespace

extern "C" {

lf_status lf_set_scene(lf_ctx* ctx, int n_spheres, const double* spheres, const int* sphere_material,
                       int n_triangles, const double* tri_positions, const double* tri_normals,
                       const int* tri_material, int n_materials, const double* materials,
                       int n_lights, const double* lights) {
  if (!ctx || n_spheres < 0 || n_triangles < 0 || n_materials < 0 || n_lights < 0) return LF_ERR_INVALID;
  if ((n_spheres && (!spheres || !sphere_material)) ||
      (n_triangles && (!tri_positions || !tri_normals || !tri_material)) ||
      (n_materials && !materials) || (n_lights && !lights))
    return LF_ERR_INVALID;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<LfPrim> prims;
  for (int i = 0; i < n_spheres; i++) {
    LfPrim p; std::memset(&p, 0, sizeof(p));
    p.type = 0; p.material = sphere_material[i];
    for (int k = 0; k < 4; k++) p.d[k] = spheres[4 * i + k];
    p.d[4] = p.d[3] * p.d[3];  // Sphere::r2
    prims.push_back(p);
  }
  for (int i = 0; i < n_triangles; i++) {
    LfPrim p; std::memset(&p, 0, sizeof(p));
    p.type = 1; p.material = tri_material[i];
    for (int k = 0; k < 9; k++) { p.d[k] = tri_positions[9 * i + k]; p.d[9 + k] = tri_normals[9 * i + k]; }
    prims.push_back(p);
  }
  for (auto& p : prims)
    if (p.material < 0 || p.material >= n_materials) return lf_fail(ctx, LF_ERR_INVALID, "scene: material index out of range");
  std::vector<LfMaterial> mats(n_materials);
  for (int i = 0; i < n_materials; i++) {
    mats[i].kind = (int)materials[4 * i];
    if (mats[i].kind < 0 || mats[i].kind > 2)
      return lf_fail(ctx, LF_ERR_INVALID, "scene: only diffuse (0 / 2) and emission (1) materials are supported");
    for (int c = 0; c < 3; c++) mats[i].rgb[c] = materials[4 * i + 1 + c];
  }
  std::vector<LfLight> lts(n_lights);
  for (int i = 0; i < n_lights; i++) {
    std::memset(&lts[i], 0, sizeof(LfLight));
    lts[i].type = (int)lights[7 * i];
    if (lts[i].type != 0 && lts[i].type != 1)
      return lf_fail(ctx, LF_ERR_INVALID, "scene: lf_set_scene takes directional (0) and point (1) lights; "
                                          "hemisphere and area lights go through lf_set_scene_lights");
    for (int c = 0; c < 3; c++) { lts[i].v[c] = lights[7 * i + 1 + c]; lts[i].rgb[c] = lights[7 * i + 4 + c]; }
  }
  std::vector<LfBvhNode> nodes;
  int depth = 0;
  if (!prims.empty()) {
    build_node(nodes, prims, 0, (int)prims.size(), &depth);
  } else {   // an empty scene: one node whose inverted box no ray enters
    LfBvhNode e; std::memset(&e, 0, sizeof(e)); e.bmin[0] = 1; e.bmax[0] = -1; e.count = 0; e.left = e.right = -1; nodes.push_back(e);
  }
  // the device traversal keeps at most depth + 1 node ids on its LDS stack and would otherwise
  // have to drop children (= lose geometry silently); the median split is ceil(log2(n / 4)) + 1
  // deep, so this only triggers beyond ~2^32 primitives -- but it must fail, not drop
  if (depth + 1 > kStackDepth - 1)
    return lf_fail(ctx, LF_ERR_INVALID, "scene: BVH deeper than the device traversal stack");
  LfSceneDev& S = ctx->scene_dev;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  void* old[] = {S.nodes, S.prims, S.materials, S.lights};
  for (void* o : old) if (o) (void)hipFree(o);
  std::memset(&S, 0, sizeof(S));
  auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
    if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  LF_HIP(ctx, up((void**)&S.nodes, nodes.data(), nodes.size() * sizeof(LfBvhNode)));
  LF_HIP(ctx, up((void**)&S.prims, prims.data(), prims.size() * sizeof(LfPrim)));
  LF_HIP(ctx, up((void**)&S.materials, mats.data(), mats.size() * sizeof(LfMaterial)));
  LF_HIP(ctx, up((void**)&S.lights, lts.data(), lts.size() * sizeof(LfLight)));
  for (int a = 0; a < 3; a++) { ctx->scene_bmin[a] = nodes[0].bmin[a]; ctx->scene_bmax[a] = nodes[0].bmax[a]; }
  S.n_nodes = (int)nodes.size(); S.n_prims = (int)prims.size();
  S.n_materials = n_materials; S.n_lights = n_lights; S.n_soft_lights = 0;
  ctx->scene_valid = true;
  return LF_OK;
}

lf_status lf_scene_bounds(lf_ctx* ctx, double bmin[3], double bmax[3], int* n_primitives) {
  if (!ctx || !bmin || !bmax) return LF_ERR_INVALID;
  if (!ctx->scene_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_scene_bounds before lf_set_scene");
  for (int a = 0; a < 3; a++) { bmin[a] = ctx->scene_bmin[a]; bmax[a] = ctx->scene_bmax[a]; }
  if (n_primitives) *n_primitives = ctx->scene_dev.n_prims;
  return LF_OK;
}

lf_status lf_set_scene_lights(lf_ctx* ctx, int n_lights, const double* rows) {
  if (!ctx || n_lights < 0 || (n_lights && !rows)) return LF_ERR_INVALID;
  if (!ctx->scene_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_set_scene_lights before lf_set_scene");
  std::vector<LfLight> lts(n_lights);
  int soft = 0, envs = 0;
  for (int i = 0; i < n_lights; i++) {
    const double* r = rows + 16 * (size_t)i;
    LfLight& l = lts[i];
    std::memset(&l, 0, sizeof(l));
    l.type = (int)r[0];
    if (l.type < 0 || l.type > 4) return lf_fail(ctx, LF_ERR_INVALID, "scene light type must be 0 .. 4");
    if (l.type == 4) envs++;
    for (int c = 0; c < 3; c++) {
      l.rgb[c] = r[1 + c]; l.v[c] = r[4 + c]; l.dir[c] = r[7 + c]; l.dim_x[c] = r[10 + c]; l.dim_y[c] = r[13 + c];
    }
    if (l.type >= 2) soft++;
    if (l.type == 3) {
      // AreaLight's constructor: area = dim_x.norm() * dim_y.norm() (light.cpp:76-80)
      const double nx = std::sqrt((l.dim_x[0] * l.dim_x[0] + l.dim_x[1] * l.dim_x[1]) + l.dim_x[2] * l.dim_x[2]);
      const double ny = std::sqrt((l.dim_y[0] * l.dim_y[0] + l.dim_y[1] * l.dim_y[1]) + l.dim_y[2] * l.dim_y[2]);
      l.area = nx * ny;
      if (!(l.area > 0)) return lf_fail(ctx, LF_ERR_INVALID, "area light with zero area");
    }
  }
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LfSceneDev& S = ctx->scene_dev;
  if (S.lights) (void)hipFree(S.lights);
  S.lights = nullptr;
  LF_HIP(ctx, hipMalloc((void**)&S.lights, std::max<size_t>(lts.size() * sizeof(LfLight), 16)));
  if (n_lights) LF_HIP(ctx, hipMemcpy(S.lights, lts.data(), lts.size() * sizeof(LfLight), hipMemcpyHostToDevice));
  S.n_lights = n_lights;
  S.n_soft_lights = soft;
  S.n_env_lights = envs;
  return LF_OK;
}

// EnvironmentLight::EnvironmentLight -> init() (environment_light.cpp:7-59): the sampling tables in
// the reference's order of operations (doubles; illum() is a float, vector3D.h:231-233)
lf_status lf_set_environment_map(lf_ctx* ctx, int w, int h, const double* rgb) {
  if (!ctx || w < 0 || h < 0 || ((w == 0) != (h == 0)) || (w && !rgb)) return LF_ERR_INVALID;
  if (w && (w < 2 || h < 2)) return lf_fail(ctx, LF_ERR_INVALID, "environment map smaller than 2 x 2 (bilerp reads two rows and columns)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->env_block) { (void)hipFree(ctx->env_block); ctx->env_block = nullptr; }
  ctx->env_dev = LfEnvDev{};
  if (w == 0) return LF_OK;
  const size_t n = (size_t)w * h;
  std::vector<double> blk(3 * n + n + n + (size_t)h);
  double* data = blk.data(); double* pdf = data + 3 * n; double* conds = pdf + n; double* marg = conds + n;
  std::memcpy(data, rgb, 3 * n * sizeof(double));
  const double PI_ = 3.14159265358979323846;   // PI (CGL/misc.h)
  double sum = 0;
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) {
      const double* t = rgb + 3 * ((size_t)w * j + i);
      const float illum = (float)((0.2126f * t[0] + 0.7152f * t[1]) + 0.0722f * t[2]);
      pdf[(size_t)w * j + i] = illum * std::sin(PI_ * (j + .5) / h);
      sum += pdf[(size_t)w * j + i];
    }
  if (!(sum > 0) || !std::isfinite(sum)) return lf_fail(ctx, LF_ERR_INVALID, "environment map without light (the reference divides by its total)");
  for (size_t k = 0; k < n; k++) pdf[k] /= sum;
  for (int j = 0; j < h; ++j) {
    marg[j] = (j == 0 ? 0 : marg[j - 1]);
    for (int i = 0; i < w; ++i) marg[j] += pdf[(size_t)w * j + i];
  }
  for (int j = 0; j < h; ++j) {
    const double marginal_density = marg[j] - (j == 0 ? 0 : marg[j - 1]);
    for (int i = 0; i < w; ++i)
      conds[(size_t)w * j + i] = (i == 0 ? 0 : conds[(size_t)w * j + i - 1]) + pdf[(size_t)w * j + i] / marginal_density;
  }
  LF_HIP(ctx, hipMalloc((void**)&ctx->env_block, blk.size() * sizeof(double)));
  LF_HIP(ctx, hipMemcpy(ctx->env_block, blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice));
  ctx->env_dev.data = ctx->env_block; ctx->env_dev.pdf = ctx->env_block + 3 * n;
  ctx->env_dev.conds = ctx->env_block + 4 * n; ctx->env_dev.marginal = ctx->env_block + 5 * n;
  ctx->env_dev.w = w; ctx->env_dev.h = h;
  return LF_OK;
}

lf_status lf_set_direct_hemisphere_sample(lf_ctx* ctx, int on) {
  if (!ctx) return LF_ERR_INVALID;
  ctx->hemisphere_sample = on != 0;
  return LF_OK;
}

lf_status lf_set_light_samples(lf_ctx* ctx, int ns_area_light) {
  if (!ctx || ns_area_light < 1) return LF_ERR_INVALID;
  ctx->ns_area_light = ns_area_light;
  return LF_OK;
}

lf_status lf_set_sampling(lf_ctx* ctx, int samples_per_batch, double max_tolerance, double n_clip,
                          double f_clip) {
  if (!ctx || samples_per_batch < 1 || !(f_clip > n_clip)) return LF_ERR_INVALID;
  ctx->samples_per_batch = samples_per_batch;
  ctx->max_tolerance = max_tolerance;
  ctx->cam.n_clip = n_clip;
  ctx->cam.f_clip = f_clip;
  return LF_OK;
}

lf_status lf_scene_trace_ray(lf_ctx* ctx, const double ray[8], uint64_t seq, double out[8]) {
  if (!ctx || !ray || !out) return LF_ERR_INVALID;
  lf_status st = scene_probe_ready(ctx, "lf_scene_trace_ray");
  if (st != LF_OK) return st;
  LfProbeRay pr;
  std::memcpy(&pr, ray, sizeof(pr));
  return scene_probe(ctx, 8, out, [&](double* d) {
    hipLaunchKernelGGL(k_scene_trace_ray, dim3(1), dim3(64), 0, ctx->stream, ctx->scene_dev, ctx->env_dev,
                       ctx->hemisphere_sample ? 1 : 0, pr, ctx->ns_area_light, seq, ctx->jitter_key, d);
  });
}

lf_status lf_scene_shade(lf_ctx* ctx, int what, const double ray[8], double t, const double n[3],
                         const double material[4], uint64_t seq, double rgb[3]) {
  if (!ctx || !ray || !n || !material || !rgb || what < 0 || what > 3) return LF_ERR_INVALID;
  lf_status st = scene_probe_ready(ctx, "lf_scene_shade");
  if (st != LF_OK) return st;
  LfMaterial m;
  std::memset(&m, 0, sizeof(m));
  m.kind = (int)material[0];
  if (m.kind < 0 || m.kind > 2) return lf_fail(ctx, LF_ERR_INVALID, "lf_scene_shade: material kind must be 0, 1 or 2");
  for (int c = 0; c < 3; c++) m.rgb[c] = material[1 + c];
  LfProbeRay pr;
  std::memcpy(&pr, ray, sizeof(pr));
  // 0 zero_bounce, 1 one_bounce (by the direct_hemisphere_sample flag), 2 / 3 the two estimators by name
  const int bits = what == 0 ? kShadeZero : kShadeOne;
  const int hemi = what == 2 ? 1 : what == 3 ? 0 : (ctx->hemisphere_sample ? 1 : 0);
  return scene_probe(ctx, 3, rgb, [&](double* d) {
    hipLaunchKernelGGL(k_scene_shade, dim3(1), dim3(64), 0, ctx->stream, ctx->scene_dev, ctx->env_dev, hemi, pr,
                       t, n[0], n[1], n[2], m, bits, ctx->ns_area_light, seq, ctx->jitter_key, d);
  });
}

lf_status lf_render_scene_term(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_render_scene_term before lf_set_frame");
  if (!ctx->scene_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_render_scene_term before lf_set_scene");
  if (!ctx->cam_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_render_scene_term before lf_set_camera");
  if (ctx->jitter_mode == 0 && ctx->scene_dev.n_soft_lights > 0)
    return lf_fail(ctx, LF_ERR_INVALID,
                   "MT19937 parity mode cannot serve area / hemisphere lights: the reference samples them from "
                   "its shared generator in hit order (use lf_set_jitter_counter)");
  if (ctx->scene_dev.n_env_lights > 0 && ctx->env_dev.w == 0)
    return lf_fail(ctx, LF_ERR_STATE, "an environment light (type 4) is listed but no map is set (lf_set_environment_map)");
  if (ctx->jitter_mode == 0 && ctx->hemisphere_sample && ctx->scene_dev.n_lights > 0)
    return lf_fail(ctx, LF_ERR_INVALID,
                   "MT19937 parity mode cannot serve hemisphere sampling: the reference draws its directions from "
                   "the shared generator in hit order (use lf_set_jitter_counter)");
  if (ctx->jitter_mode == 0) {
    if (!ctx->jitter_table_valid || (ctx->ns_aa > 0 && !ctx->jitter_aa_raw) || ctx->jitter_aa_ns != ctx->ns_aa)
      return lf_fail(ctx, LF_ERR_STATE, "MT19937 jitter: call lf_set_jitter_mt19937 after lf_set_params");
    if (ctx->ns_aa >= ctx->samples_per_batch)
      return lf_fail(ctx, LF_ERR_INVALID,
                     "MT19937 parity mode needs ns_aa < samplesPerBatch: the adaptive early-out makes "
                     "every later pixel's draws depend on earlier pixels (use lf_set_jitter_counter)");
  }
  LF_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->W * ctx->H * 3;
  if (!ctx->scene) {
    LF_HIP(ctx, hipMalloc((void**)&ctx->scene, n * sizeof(double)));
    LF_HIP(ctx, hipMemsetAsync(ctx->scene, 0, n * sizeof(double), ctx->stream));
  }
  const size_t px = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
  if (px == 0) return LF_OK;
  const bool soft = ctx->scene_dev.n_soft_lights > 0 || ctx->env_dev.w > 0 || ctx->hemisphere_sample;
#define LF_LAUNCH_SCENE(SOFT)                                                                            \
  hipLaunchKernelGGL(k_scene_term<SOFT>, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream,  \
                     ctx->scene_dev, ctx->env_dev, ctx->hemisphere_sample ? 1 : 0, ctx->cam, ctx->W,    \
                     ctx->H, ctx->y0, ctx->y1, ctx->row_phase, ctx->row_period, ctx->ns_aa,              \
                     ctx->ns_area_light, ctx->samples_per_batch,                                         \
                     ctx->max_tolerance, ctx->jitter_aa_raw, ctx->jitter_mode, ctx->jitter_key, ctx->scene)
  hipEvent_t ev = lf_timing_begin(ctx, LFK_SCENE);
  if (soft) LF_LAUNCH_SCENE(true); else LF_LAUNCH_SCENE(false);
  lf_timing_end(ctx, LFK_SCENE, ev);
#undef LF_LAUNCH_SCENE
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

}  // extern "C"
