// lf_scene.hip -- the scene-radiance term of PathTracer::raytrace_pixel on the device
// (SURVEY.md section 8 row f2): the per-pixel sample loop (pathtracer.cpp:841-875), pinhole ray
// generation (camera.cpp:278-305), closest-hit search over spheres and triangles
// (scene/sphere.cpp:11-111, scene/triangle.cpp:25-112, scene/bvh.cpp:201-222), emission +
// direct lighting of diffuse surfaces by delta lights with shadow rays
// (pathtracer.cpp:142-232, bsdf.cpp:21-60, light.cpp:11-24, :47-60).
//
// Everything is double precision and follows the reference expression by expression (operator
// order of CGL::Vector3D / Matrix3x3 included) so that pixels agree with the CPU renderer far
// inside the 1e-4 bar.  The BVH is our own (binned SAH on the host; on the device a nearest-first walk
// over two-child nodes with conservative FLOAT boxes, lf_scene_types.h): the closest hit does not
// depend on the tree, only the amount of work does -- the primitive tests stay the reference's doubles.
//
// Sampled lights -- AreaLight, InfiniteHemisphereLight and EnvironmentLight (scene/light.cpp:35-48,
// :82-101; scene/environment_light.cpp) with ns_area_light samples each, exactly the estimator of
// estimate_direct_lighting_importance (pathtracer.cpp:143-213), and the uniform hemisphere sampling of
// emitters (-H, pathtracer.cpp:86-138) -- draw from the order-free Philox counter RNG: the reference
// draws them from its shared MT19937 only when a camera ray hits something, which makes every later
// pixel's jitter depend on every earlier pixel's hits, so no device schedule can reproduce its stream.
// They are therefore validated statistically against frames the reference rendered
// (tests/test_gpu_area_lights.py, tests/test_gpu_env_light.py) and refused in MT19937 parity mode.
// Not covered (the call fails loudly rather than approximating): spot lights (a stub in the reference,
// light.cpp:64-72).  The Mirror / Glass / Microfacet BSDFs are unfilled stubs in the reference
// (advanced_bsdf.cpp: f() = 0): a host hands them over as black diffuse occluders.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "lf_internal.h"
#include "lf_march_events.h"

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

// A ray as the traversal keeps it: the reference's Ray (o, d, min_t, max_t in double: every primitive
// test uses exactly these) plus what the box tests use -- floats that BRACKET the doubles.
struct DRay {
  V3 o, d; double min_t, max_t;
  float olx, oly, olz, ohx, ohy, ohz;  // floats strictly below / above o
  float ix, iy, iz;                    // 1 / d; NaN when |1 / d| is no finite float: the axis then never culls
  float tmin_c, tmax_c;                // min_t / max_t, widened by the box test's slack
};
// the closest hit so far: primitive + what its test produced; the normal is formed ONCE, at the end
struct Hit { double t, b1, b2; int prim; };

__device__ inline float f32_down(double x) { const float f = (float)x; return (double)f > x ? nextafterf(f, -INFINITY) : f; }
__device__ inline float f32_up(double x) { const float f = (float)x; return (double)f < x ? nextafterf(f, INFINITY) : f; }
// STRICTLY below / above: the origin's bracket.  With a direction component of exactly 0 the slab
// products are (lo - oh) * inf and (hi - ol) * inf; a difference of exactly 0 gives NaN, which the
// min / max of box_miss resolve towards "outside".  That is only right if a zero difference MEANS
// outside: lo == oh > o does, lo == oh == o (an origin exactly on the slab's plane, e.g. a shadow ray
// leaving an axis-aligned face along an axis) would not.
__device__ inline float f32_below(double x) { const float f = (float)x; return (double)f >= x ? nextafterf(f, -INFINITY) : f; }
__device__ inline float f32_above(double x) { const float f = (float)x; return (double)f <= x ? nextafterf(f, INFINITY) : f; }
__device__ inline float inv_dir(double d) {
  const double q = 1.0 / d;           // d = 0: +-inf, which culls correctly (a ray parallel to a slab)
  return (d != 0.0 && fabs(q) > 3.0e38) ? __int_as_float(0x7fc00000) : (float)q;
}
// max_t shrinks with every accepted hit: the float bound follows (2^-20 relative + 1e-30 absolute:
// four times what the box test's own rounding can be off, see box_miss)
__device__ inline void widen_max_t(DRay& r) {
  const float m = f32_up(r.max_t);
  r.tmax_c = fmaf(9.6e-7f, fabsf(m), m) + 1e-30f;
}
__device__ inline DRay make_ray(V3 o, V3 d, double min_t, double max_t) {
  DRay r;
  r.o = o; r.d = d; r.min_t = min_t; r.max_t = max_t;
  r.olx = f32_below(o.x); r.oly = f32_below(o.y); r.olz = f32_below(o.z);
  r.ohx = f32_above(o.x); r.ohy = f32_above(o.y); r.ohz = f32_above(o.z);
  r.ix = inv_dir(d.x); r.iy = inv_dir(d.y); r.iz = inv_dir(d.z);
  const float m = f32_down(min_t);
  r.tmin_c = fmaf(-9.6e-7f, fabsf(m), m) - 1e-30f;
  widen_max_t(r);
  return r;
}

// ---- primitives ------------------------------------------------------------------------------
// Sphere::test + intersect (scene/sphere.cpp:11-111)
__device__ inline bool hit_sphere(const LfPrim& s, int idx, DRay& r, Hit* h) {
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
  if (h) { h->t = t1; h->prim = idx; }
  return true;
}

// moller_trumbore + is_valid_intersection + Triangle::intersect (scene/triangle.cpp:25-112)
__device__ inline bool hit_triangle(const LfPrim& t, int idx, DRay& r, Hit* h) {
  // (e1 = p1 - p0 and e2 = p2 - p0 come with the primitive: the host's IEEE subtraction gives the bits
  // the device's would, and the test holds 6 doubles less)
  const V3 p0 = v3(t.d[0], t.d[1], t.d[2]), e1 = v3(t.d[3], t.d[4], t.d[5]), e2 = v3(t.d[6], t.d[7], t.d[8]);
  const V3 s = r.o - p0;
  const V3 s1 = cross(r.d, e2), s2 = cross(s, e1);
  const double rc = 1. / dot(s1, e1);  // operator/= multiplies by the reciprocal
  const double tt = dot(s2, e2) * rc, b1 = dot(s1, s) * rc, b2 = dot(s2, r.d) * rc;
  if (tt < r.min_t || tt > r.max_t) return false;
  if (b1 < 0 || b1 > 1) return false;
  if (b2 < 0 || b2 > 1) return false;
  if (b1 + b2 > 1) return false;
  r.max_t = tt;
  if (h) { h->t = tt; h->b1 = b1; h->b2 = b2; h->prim = idx; }
  return true;
}

// the closest hit's surface normal and material: Sphere::normal (sphere.h:73-75) at o + t d, or the
// triangle's barycentric blend (triangle.cpp:100-105)
__device__ inline V3 hit_normal(const LfSceneDev& sc, const DRay& r, const Hit& h, int* material) {
  const LfPrim& p = sc.prims[h.prim];
  *material = p.material;
  if (p.type == 0) return unit((r.o + h.t * r.d) - v3(p.d[0], p.d[1], p.d[2]));
  const LfPrimNormals& q = sc.normals[h.prim];
  const double b0 = 1 - h.b1 - h.b2;
  const V3 n1 = v3(q.n[0], q.n[1], q.n[2]), n2 = v3(q.n[3], q.n[4], q.n[5]), n3 = v3(q.n[6], q.n[7], q.n[8]);
  return unit((b0 * n1 + h.b1 * n2) + h.b2 * n3);
}

// BBox::intersect (scene/bbox.cpp:12-49) as a conservative float test: true only when the DOUBLE ray
// provably misses the DOUBLE box or leaves it outside [min_t, max_t].
//   (lo - oh) <= lo - o and (hi - ol) >= hi - o exactly (box rounded outward, origin bracketed), so
//   after the multiplication by 1 / d the smaller product bounds the entry from below and the larger
//   one the exit from above, each off by at most 3 roundings (2^-24 each, relative) + underflow.
// tn > tf is decided with 2^-21 (|tn| + |tf|) + 1e-30 of slack, the two ends against bounds that were
// widened once per ray (make_ray / widen_max_t).  fminf / fmaxf drop NaNs: an axis without a usable
// reciprocal (both products NaN) simply does not cull; 0 * inf only arises for an origin strictly
// outside the slab of an axis the ray is parallel to (f32_below / f32_above), where the surviving
// infinite product culls, correctly.  The sum of the
// magnitudes is clamped so that an infinite entry (a ray parallel to a slab and outside it) still
// compares as "misses" instead of inf > inf.
__device__ inline bool box_miss(const DRay& r, float lx, float ly, float lz, float hx, float hy, float hz,
                                float* entry) {
  const float ax = (lx - r.ohx) * r.ix, bx = (hx - r.olx) * r.ix;
  const float ay = (ly - r.ohy) * r.iy, by = (hy - r.oly) * r.iy;
  const float az = (lz - r.ohz) * r.iz, bz = (hz - r.olz) * r.iz;
  const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  const float mag = fminf(fabsf(tn) + fabsf(tf), 3.0e38f);
  *entry = tn;
  return (tn - tf > fmaf(4.8e-7f, mag, 1e-30f)) || tf < r.tmin_c || tn > r.tmax_c;
}

// closest hit (BVHAccel::intersect, scene/bvh.cpp:201-222: the recursion shrinks r.max_t as it goes,
// so whatever the visiting order the last accepted primitive is the closest one).  h == nullptr is
// the shadow-ray query (has_intersection): the first accepted primitive settles it.
// A node visit tests the boxes of both children (LfBvhNode) and descends into the nearer one first:
// the nearer subtree usually holds the closest hit, which then culls the farther one when it is
// popped.  The stack of deferred children lives in LDS ([depth][thread]: conflict-free), not in
// scratch memory; it holds one entry per level at most (the host refuses deeper trees).
constexpr int kStackDepth = 24;
// what the reference's BVHAccel counts for its end-of-frame log (bvh.h:85,105; bvh.cpp:211;
// raytraced_renderer.cpp:706-709): rays handed to intersect() / has_intersection(), and primitive tests of
// the closest-hit queries (has_intersection does not count its tests) -- here per lane, summed at the
// kernel's end (lf_get_scene_counters)
struct SceneTally { unsigned rays, isects; };
__device__ bool closest_hit(const LfSceneDev& sc, DRay& r, Hit* h, int* __restrict__ stack /* [kStackDepth][256] + tid */,
                            SceneTally& tally) {
  int sp = 0, cur = 0;
  bool any = false;
  tally.rays++;
  for (;;) {
    if (cur >= 0) {
      const float4* __restrict__ q = reinterpret_cast<const float4*>(sc.nodes + cur);
      const float4 q0 = q[0], q1 = q[1], q2 = q[2];
      const int2 ch = *reinterpret_cast<const int2*>(q + 3);
      float t0, t1;
      const bool m0 = box_miss(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, &t0) || ch.x == kLfNoChild;
      const bool m1 = box_miss(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, &t1) || ch.y == kLfNoChild;
      if (!m0 && !m1) {
        const bool second_first = t1 < t0;
        stack[256 * sp++] = second_first ? ch.x : ch.y;
        cur = second_first ? ch.y : ch.x;
        continue;
      }
      if (!m0) { cur = ch.x; continue; }
      if (!m1) { cur = ch.y; continue; }
    } else {
      const int code = ~cur, first = code >> 2, count = (code & 3) + 1;
      bool hit_here = false;
      if (h) tally.isects += (unsigned)count;
      for (int i = 0; i < count; i++) {
        const LfPrim& p = sc.prims[first + i];
        const bool hit = p.type == 0 ? hit_sphere(p, first + i, r, h) : hit_triangle(p, first + i, r, h);
        hit_here = hit_here || hit;
      }
      if (hit_here) {
        if (!h) return true;
        any = true;
        widen_max_t(r);
      }
    }
    if (sp == 0) break;
    cur = stack[256 * --sp];
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
                        int* __restrict__ stack, int ns_area_light, uint4 rng_ctr, uint2 rng_key, SceneTally& tally) {
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
      DRay out = make_ray(hit_p, ww, kEpsF, INFINITY);
      Hit h2;
      if (closest_hit(sc, out, &h2, stack, tally)) {
        const LfMaterial& m2 = sc.materials[sc.prims[h2.prim].material];
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
      DRay sh = make_ray(hit_p, wi, kEpsF, dist - kEpsF);
      if (!closest_hit(sc, sh, nullptr, stack, tally)) {
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
                       int* __restrict__ stack, int ns_area_light, uint4 rng_ctr, uint2 rng_key, SceneTally& tally) {
  Hit isect;
  if (!closest_hit(sc, r, &isect, stack, tally))   // pathtracer.cpp:291-292
    return (SOFT && ev.w) ? env_sample_dir(ev, r.d) : v3(0, 0, 0);
  int material;
  const V3 n = hit_normal(sc, r, isect, &material);
  return shade_hit<SOFT>(sc, ev, hemisphere, r, isect.t, n, sc.materials[material],
                         kShadeZero | kShadeOne, stack, ns_area_light, rng_ctr, rng_key, tally);
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
// 256 and the delta-light one 163; bounded they spill their cold paths to scratch and are still
// faster -- timing frames at 2 / 3 / 4 / 5 waves: 32.6 / 26.0 / 24.8 / 25.0 ms with an area light and the
// environment, 4.1 / 4.1 / 3.9 / 4.9 ms with delta lights only)
//
// LENS = true (round 4; lf_set_lens_camera): the one call `camera->generate_ray(x, y)` of the loop
// (pathtracer.cpp:848) is replaced by the primary path of the MARCH's sensor sample through the
// prescription -- sample_start + primary_path of lf_march_events.h, float32, the arithmetic of the
// ghost march -- and the radiance that comes back along the exit ray is weighted by the transmitted
// fraction (Fresnel losses at every interface, the aperture mask, cos^4 and the pupil's solid angle,
// times the exposure).  A sample the lens blocks contributes 0 and still counts: the mean is taken
// over the loop variable exactly as before.  Lens space = camera space (x right, y up, the scene at
// -z) scaled by world_per_mm, with the entrance pupil's centre at the camera position.
#ifndef LF_SCENE_WAVES
#define LF_SCENE_WAVES 4
#endif
struct ScenePixelArgs {
  int W, H, ns_aa, ns_area_light, samples_per_batch, jitter_mode, hemisphere;
  double max_tolerance;
  uint64_t key;
};
template <bool SOFT, bool LENS>
__device__ __forceinline__ void scene_pixel(const LfSceneDev& sc, const LfEnvDev& ev, const LfCamera& cam,
                                            const ScenePixelArgs& a, const uint32_t* __restrict__ aa_raw,
                                            const LfLensCamArgs& lc, const LfPrimaryDev* __restrict__ prim,
                                            const float* __restrict__ mask, int x, int y, int lane,
                                            int* __restrict__ stack, SceneTally& tally, unsigned& lens_started,
                                            unsigned& lens_left, double* __restrict__ scene) {
  const int W = a.W, H = a.H, ns_aa = a.ns_aa;
  const uint2 key2 = make_uint2((unsigned)a.key, (unsigned)(a.key >> 32));
  const size_t p = (size_t)y * W + x;
  const double PI_ = 3.14159265358979323;
  const double edge_x = tan(0.5 * (cam.hfov_deg * (PI_ / 180.0)));
  const double edge_y = tan(0.5 * (cam.vfov_deg * (PI_ / 180.0)));
  V3 total = v3(0, 0, 0);
  float s1 = 0.0f, s2 = 0.0f;
  int sample;
  for (sample = 1; sample <= ns_aa; sample++) {
    V3 L;
    if (LENS) {
      lfm::SampleSpec spec;
      spec.W = W; spec.xs = lc.xs; spec.G = lc.G; spec.inv_G = lc.inv_G; spec.sub_bits = lc.sub_bits; spec.inv_sub = lc.inv_sub;
      spec.key = key2; spec.pitch = lc.pitch; spec.half_w = lc.half_w; spec.half_h = lc.half_h;
      spec.pupil_h = lc.pupil_h; spec.vz = lc.vz; spec.geom_norm = lc.geom_norm;
      // (the march's samples in an order whose prefixes cover the pupil: lf_fill_lenscam_args)
      const int s_idx = (int)(((long long)(sample - 1) * (long long)lc.order_step) % (long long)ns_aa);
      const lfm::StartRay st = lfm::sample_start(spec, x, y, s_idx);
      L = v3(0, 0, 0);
      const int n_rays = lc.mode == 2 ? lc.n_lambda : 1;
      for (int li = 0; li < n_rays; li++) {
        const int l = lc.mode == 2 ? li : lc.lambda_ref;
        lfm::Ray r{st.X, st.Y, 0.0f, fmaf(st.X, st.X, st.Y * st.Y), st.dx, st.dy, st.dz, st.w0, 1.0f};
        const bool left = lfm::primary_path(prim, l, r, mask, lc.mw, lc.mh, lane);
        lens_started++;
        if (!left) continue;
        lens_left++;
        const double wt = (double)__fdiv_rn(r.wn, r.wd) * lc.exposure;
        // exit state (float, lens space, mm) -> camera space -> world: doubles from here on
        const double wpm = lc.world_per_mm;
        const V3 oc = v3((double)r.px * wpm, (double)r.py * wpm,
                         ((double)(prim->front_zv + r.hz) - lc.z_ref_mm) * wpm);
        const V3 dc = unit(v3((double)r.dx, (double)r.dy, (double)r.dz));
        const V3 ow = v3(cam.pos[0] + ((oc.x * cam.c2w[0] + oc.y * cam.c2w[1]) + oc.z * cam.c2w[2]),
                         cam.pos[1] + ((oc.x * cam.c2w[3] + oc.y * cam.c2w[4]) + oc.z * cam.c2w[5]),
                         cam.pos[2] + ((oc.x * cam.c2w[6] + oc.y * cam.c2w[7]) + oc.z * cam.c2w[8]));
        const DRay ray = make_ray(ow,
                                  v3((dc.x * cam.c2w[0] + dc.y * cam.c2w[1]) + dc.z * cam.c2w[2],
                                     (dc.x * cam.c2w[3] + dc.y * cam.c2w[4]) + dc.z * cam.c2w[5],
                                     (dc.x * cam.c2w[6] + dc.y * cam.c2w[7]) + dc.z * cam.c2w[8]),
                                  cam.n_clip, cam.f_clip);
        const V3 Ll = radiance<SOFT>(sc, ev, a.hemisphere != 0, ray, stack, a.ns_area_light,
                                     make_uint4((unsigned)p, (unsigned)sample, 0u, 0u), key2, tally);
        if (lc.mode == 2)
          L = L + v3(Ll.x * (wt * (double)lc.lambda_rgb[l][0]), Ll.y * (wt * (double)lc.lambda_rgb[l][1]),
                     Ll.z * (wt * (double)lc.lambda_rgb[l][2]));
        else
          L = L + Ll * wt;
      }
    } else {
      unsigned ra, rb;
      if (a.jitter_mode == 0) {
        ra = aa_raw[p * (size_t)(2 * ns_aa) + 2 * (sample - 1)];
        rb = aa_raw[p * (size_t)(2 * ns_aa) + 2 * (sample - 1) + 1];
      } else {
        const uint4 r4 = philox4x32_10(make_uint4((unsigned)p, (unsigned)sample, 0x5ce4e000u, 0u), key2);
        ra = r4.x; rb = r4.y;
      }
      // Vector2D(random_uniform(), random_uniform()): g++ evaluates right to left, the first draw is y
      const double sy = (double)y + random_uniform_from_raw(ra);
      const double sx = (double)x + random_uniform_from_raw(rb);
      const double nx = sx / (double)W, ny = sy / (double)H;
      // Camera::generate_ray (camera.cpp:278-305)
      V3 dir = unit(v3(edge_x * (2 * nx - 1), edge_y * (2 * ny - 1), -1));
      const DRay r = make_ray(v3(cam.pos[0], cam.pos[1], cam.pos[2]),
                              v3((dir.x * cam.c2w[0] + dir.y * cam.c2w[1]) + dir.z * cam.c2w[2],
                                 (dir.x * cam.c2w[3] + dir.y * cam.c2w[4]) + dir.z * cam.c2w[5],
                                 (dir.x * cam.c2w[6] + dir.y * cam.c2w[7]) + dir.z * cam.c2w[8]),
                              cam.n_clip, cam.f_clip);
      L = radiance<SOFT>(sc, ev, a.hemisphere != 0, r, stack, a.ns_area_light,
                         make_uint4((unsigned)p, (unsigned)sample, 0u, 0u), key2, tally);
    }
    // Vector3D::illum (vector3D.h:231-233): float coefficients, double arithmetic, float result
    const float illum = (float)((0.2126f * L.x + 0.7152f * L.y) + 0.0722f * L.z);
    s1 += illum;
    s2 += illum * illum;
    total = total + L;
    if (sample > 1 && sample % a.samples_per_batch == 0) {  // :862-868
      const float sd = (float)sqrt(1.0 / (sample - 1) * (double)(s2 - s1 * s1 / (float)sample));
      const float ci = (float)(1.96 * (double)sd / sqrt((double)sample));
      if ((double)ci <= a.max_tolerance * (double)s1 / (double)sample) break;
    }
  }
  const double rc = 1. / (double)sample;  // :875 -- ns_aa + 1 when the loop ran to its end
  scene[3 * p] = total.x * rc;
  scene[3 * p + 1] = total.y * rc;
  scene[3 * p + 2] = total.z * rc;
}

template <bool SOFT, bool LENS>
__global__ __launch_bounds__(256, LF_SCENE_WAVES) void k_scene_term(LfSceneDev sc, LfEnvDev ev, LfCamera cam,
                                                    ScenePixelArgs a, int y0, int y1, LfDeal deal,
                                                    const uint32_t* __restrict__ aa_raw,
                                                    LfLensCamArgs lc, const LfPrimaryDev* __restrict__ prim,
                                                    const float* __restrict__ mask,
                                                    unsigned long long* __restrict__ counters,
                                                    double* __restrict__ scene) {
  // traversal stacks of the 256 threads: the median-split tree of n primitives is ceil(log2(n / 4))
  // deep (<= 29), the stack holds at most depth + 1 entries
  __shared__ int s_stack[kStackDepth * 256];
  __shared__ unsigned long long s_cnt[kSceneCounters];
  if (threadIdx.x < kSceneCounters) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  int* const stack = s_stack + threadIdx.x;
  // a wave = one 8 x 8 pixel tile (its 64 camera rays walk the same part of the tree: fewer divergent
  // visits and better hit rates in the vector cache than a 64 x 1 strip), a workgroup = 4 tiles side by
  // side; blockIdx.y counts 8-row tile rows from the one that holds y0
  const int lane = threadIdx.x & 63;
  const int x = ((int)blockIdx.x * 4 + ((int)threadIdx.x >> 6)) * 8 + (lane & 7);
  const int y = ((y0 >> 3) + (int)blockIdx.y) * 8 + (lane >> 3);
  // multi-GPU: only the tile rows / blocks this context owns, like k_flare_layer, which is the only
  // reader of this buffer
  const bool mine = x < a.W && y >= y0 && y < y1 && lf_deal_mine(deal, x, y);
  SceneTally tally{0u, 0u};
  unsigned lens_started = 0u, lens_left = 0u;
  if (mine)
    scene_pixel<SOFT, LENS>(sc, ev, cam, a, aa_raw, lc, prim, mask, x, y, lane, stack, tally, lens_started,
                            lens_left, scene);
  // the frame's counters (every lane arrives here): LDS adds, then one global add per counter
  if (tally.rays) atomicAdd(&s_cnt[0], (unsigned long long)tally.rays);
  if (tally.isects) atomicAdd(&s_cnt[1], (unsigned long long)tally.isects);
  if (LENS && lens_started) atomicAdd(&s_cnt[2], (unsigned long long)lens_started);
  if (LENS && lens_left) atomicAdd(&s_cnt[3], (unsigned long long)lens_left);
  __syncthreads();
  if (threadIdx.x < kSceneCounters && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
}

// ---- the lens camera with its scene rays compacted (round 5) ------------------------------------------
// k_scene_term<.., true> walks the tree with the lanes whose sample LEFT the lens: a quarter of them on the
// bench frames (the rest end at the stop or a clear aperture), so three quarters of every traversal
// instruction are idle lanes.  Here a wave still owns one 8 x 8 tile and marches ITS lanes' primary paths in
// lockstep (the loop variable `sample` is wave-uniform: lanes only drop out at a batch's convergence check),
// but a sample that leaves the lens is appended to a queue in LDS (exit state in float32, owner lane,
// wavelength, round) and the tree is walked 64 queue entries at a time, every lane busy, by whichever lane
// the entry lands on.  The radiance of an entry depends on (pixel, sample, ray) only -- the Philox counter is
// the OWNER's -- and goes back to the owner through LDS in queue order = sample order, so every pixel sees
// the additions of scene_pixel<.., true> in the same order: bit-identical frames
// (tests/test_gpu_lens_camera.py compares the two kernels).  A sample the lens blocks adds +0.0 to every sum
// of scene_pixel and is simply never queued.
constexpr int kLensQueue = 128;      // < 64 left over + one round of <= 64
constexpr int kLensStridedMaxPrims = 1024;
// (the traversal stack is dynamic LDS sized by the tree at hand -- [levels][256] ints -- so that a shallow tree leaves
// the CU's LDS to more workgroups: 25.6 KB of queues + 1 KB per level)
#ifndef LF_SCENE_LENS_WAVES
#define LF_SCENE_LENS_WAVES 3     // (2 / 3 / 4 / 5 waves per SIMD: 10.0 / 8.2 / 10.3 / 14.2 ms on the c4 bench frame -- spills beyond 3)
#endif
template <bool SOFT>
__global__ __launch_bounds__(256, LF_SCENE_LENS_WAVES) void k_scene_lens(LfSceneDev sc, LfEnvDev ev, LfCamera cam, ScenePixelArgs a,
                                                       int y0, int y1, LfDeal deal, int mxs, LfLensCamArgs lc,
                                                       const LfPrimaryDev* __restrict__ prim,
                                                       const float* __restrict__ mask,
                                                       unsigned long long* __restrict__ counters,
                                                       double* __restrict__ scene) {
  extern __shared__ int s_stack[];
  __shared__ unsigned long long s_cnt[kSceneCounters];
  __shared__ float s_qf[4][7][kLensQueue];       // px py hz dx dy dz, transmitted weight
  __shared__ unsigned s_qt[4][kLensQueue];       // owner lane | wavelength slot << 6 | sample << 10: tag >> 6 = the ROUND, rising along the queue
  __shared__ double s_res[4][3][64];             // a round's results on their way back, by owner lane
  __shared__ unsigned s_stamp[4][64];            // round of the result that lies in s_res (0: none)
  if (threadIdx.x < kSceneCounters) s_cnt[threadIdx.x] = 0ull;
  const int wv = (int)threadIdx.x >> 6, lane = threadIdx.x & 63;
  s_stamp[wv][lane] = 0u;
  __syncthreads();
  int* const stack = s_stack + threadIdx.x;
  // which 64 pixels a wave owns (any choice gives the same frame).  mxs = xs: the MARCH's wave tile, 8 columns 2^xs
  // apart (lf_set_tile_stride) -- its lanes share the sub-cell of the pupil a sample draws (lfm::sample_start), so where
  // the pupil is closed it is closed for the whole wave and primary_path stops early.  mxs = 0: 8 x 8 adjacent pixels,
  // whose scene rays walk the same part of a large tree.  Measured at 4K (4 / 50 801 triangles): 7.9 / 18.5 ms against
  // 8.2 / 16.4 ms -- the host picks by the size of the tree (lf_render_scene_term).
  const int tile = (int)blockIdx.x * 4 + wv;
  const int tile_x = ((tile >> mxs) << (3 + mxs)) + (tile & ((1 << mxs) - 1)), tile_y = ((y0 >> 3) + (int)blockIdx.y) * 8;
  const int x = tile_x + ((lane & 7) << mxs), y = tile_y + (lane >> 3);
  const bool mine = x < a.W && y >= y0 && y < y1 && lf_deal_mine(deal, x, y);
  const int ns_aa = a.ns_aa;
  const uint2 key2 = make_uint2((unsigned)a.key, (unsigned)(a.key >> 32));
  SceneTally tally{0u, 0u};
  unsigned lens_started = 0u, lens_left = 0u;
  lfm::SampleSpec spec;
  spec.W = a.W; spec.xs = lc.xs; spec.G = lc.G; spec.inv_G = lc.inv_G; spec.sub_bits = lc.sub_bits; spec.inv_sub = lc.inv_sub;
  spec.key = key2; spec.pitch = lc.pitch; spec.half_w = lc.half_w; spec.half_h = lc.half_h;
  spec.pupil_h = lc.pupil_h; spec.vz = lc.vz; spec.geom_norm = lc.geom_norm;
  const int n_rays = lc.mode == 2 ? lc.n_lambda : 1;

  // the owner's sums (scene_pixel's total / s1 / s2) and the sample whose rays are still arriving
  V3 total = v3(0, 0, 0), Lcur = v3(0, 0, 0);
  float s1 = 0.0f, s2 = 0.0f;
  int cur_s = 0, final_sample = 0;
  bool active = mine;
  int q_count = 0;
  auto finish_sample = [&]() {
    if (cur_s) {
      const float illum = (float)((0.2126f * Lcur.x + 0.7152f * Lcur.y) + 0.0722f * Lcur.z);
      s1 += illum;
      s2 += illum * illum;
      total = total + Lcur;
      cur_s = 0;
    }
  };
  // the first n entries of the queue: radiance along each, back to the owners round by round, queue shifted
  auto flush = [&](int n) {
    const bool work = lane < n;
    unsigned tag = 0u;
    V3 C = v3(0, 0, 0);
    if (work) {
      tag = s_qt[wv][lane];
      const float px = s_qf[wv][0][lane], py = s_qf[wv][1][lane], hz = s_qf[wv][2][lane];
      const float dx = s_qf[wv][3][lane], dy = s_qf[wv][4][lane], dz = s_qf[wv][5][lane], wq = s_qf[wv][6][lane];
      const int owner = (int)(tag & 63u), l = lc.mode == 2 ? (int)((tag >> 6) & 15u) : lc.lambda_ref, sample = (int)(tag >> 10);
      const size_t p = (size_t)(tile_y + (owner >> 3)) * a.W + (tile_x + ((owner & 7) << mxs));
      const double wt = (double)wq * lc.exposure;
      const double wpm = lc.world_per_mm;
      const V3 oc = v3((double)px * wpm, (double)py * wpm, ((double)(prim->front_zv + hz) - lc.z_ref_mm) * wpm);
      const V3 dc = unit(v3((double)dx, (double)dy, (double)dz));
      const V3 ow = v3(cam.pos[0] + ((oc.x * cam.c2w[0] + oc.y * cam.c2w[1]) + oc.z * cam.c2w[2]),
                       cam.pos[1] + ((oc.x * cam.c2w[3] + oc.y * cam.c2w[4]) + oc.z * cam.c2w[5]),
                       cam.pos[2] + ((oc.x * cam.c2w[6] + oc.y * cam.c2w[7]) + oc.z * cam.c2w[8]));
      const DRay ray = make_ray(ow,
                                v3((dc.x * cam.c2w[0] + dc.y * cam.c2w[1]) + dc.z * cam.c2w[2],
                                   (dc.x * cam.c2w[3] + dc.y * cam.c2w[4]) + dc.z * cam.c2w[5],
                                   (dc.x * cam.c2w[6] + dc.y * cam.c2w[7]) + dc.z * cam.c2w[8]),
                                cam.n_clip, cam.f_clip);
      const V3 Ll = radiance<SOFT>(sc, ev, a.hemisphere != 0, ray, stack, a.ns_area_light,
                                   make_uint4((unsigned)p, (unsigned)sample, 0u, 0u), key2, tally);
      if (lc.mode == 2)
        C = v3(Ll.x * (wt * (double)lc.lambda_rgb[l][0]), Ll.y * (wt * (double)lc.lambda_rgb[l][1]),
               Ll.z * (wt * (double)lc.lambda_rgb[l][2]));
      else
        C = Ll * wt;
    }
    // back to the owners: one round at a time (within a round every owner has at most one entry), in queue order
    lfm::lanemask pend = __ballot(work);
    while (pend) {
      const int src = __ffsll((long long)pend) - 1;
      const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)(tag >> 6), src);
      const bool snd = work && (tag >> 6) == r0;
      if (snd) {
        const int owner = (int)(tag & 63u);
        s_res[wv][0][owner] = C.x; s_res[wv][1][owner] = C.y; s_res[wv][2][owner] = C.z;
        s_stamp[wv][owner] = r0;
      }
      __builtin_amdgcn_wave_barrier();
      if (s_stamp[wv][lane] == r0) {
        const int sample = (int)(r0 >> 4);
        if (sample != cur_s) { finish_sample(); cur_s = sample; Lcur = v3(0, 0, 0); }
        Lcur = Lcur + v3(s_res[wv][0][lane], s_res[wv][1][lane], s_res[wv][2][lane]);
        s_stamp[wv][lane] = 0u;                // (a round can straddle two flushes: taken once)
      }
      __builtin_amdgcn_wave_barrier();
      pend &= ~__ballot(snd);
    }
    // what is left moves to the front
    const int rem = q_count - n;
    unsigned mt = 0u; float mf[7];
    const bool mv = lane < rem;
    if (mv) { mt = s_qt[wv][n + lane]; for (int k = 0; k < 7; k++) mf[k] = s_qf[wv][k][n + lane]; }
    __builtin_amdgcn_wave_barrier();
    if (mv) { s_qt[wv][lane] = mt; for (int k = 0; k < 7; k++) s_qf[wv][k][lane] = mf[k]; }
    __builtin_amdgcn_wave_barrier();
    q_count = rem;
  };

  int sample, s_idx = 0;
  const int s_step = (int)((long long)lc.order_step % (long long)ns_aa);
  const unsigned g_magic = spec.G > 1 ? (unsigned)(0x100000000ull / (unsigned long long)spec.G) : 0xffffffffu;
  int next_check = a.samples_per_batch > 1 ? a.samples_per_batch : 2;      // the first multiple of samples_per_batch above 1
  for (sample = 1; sample <= ns_aa; sample++) {     // wave-uniform
    if (__ballot(active) == 0ull) break;
    lfm::StartRay st{0.0f, 0.0f, 0.0f, 0.0f, -1.0f, 0.0f};
    // (the march's samples in an order whose prefixes cover the pupil, scene_pixel: ((sample - 1) order_step) mod ns_aa)
    if (active) {
      // s_idx / G without the division: the quotient by floor(2^32 / G) is the true one or one less
      unsigned cy = __umulhi((unsigned)s_idx, g_magic);
      if ((unsigned)s_idx - cy * (unsigned)spec.G >= (unsigned)spec.G) cy++;
      const int cx = s_idx - (int)cy * spec.G;
      unsigned sxi = 0u, syi = 0u;
      if (s_idx < spec.G * spec.G) lfm::sample_subcell(spec, lfm::sample_tile_id(spec, x, y), s_idx, sxi, syi);
      st = lfm::sample_start_in(spec, x, y, s_idx, cx, (int)cy, sxi, syi);
    }
    s_idx += s_step;
    if (s_idx >= ns_aa) s_idx -= ns_aa;
    for (int li = 0; li < n_rays; li++) {
      const int l = lc.mode == 2 ? li : lc.lambda_ref;
      bool left = false;
      lfm::Ray r{st.X, st.Y, 0.0f, fmaf(st.X, st.X, st.Y * st.Y), st.dx, st.dy, st.dz, st.w0, 1.0f};
      if (active) {
        left = lfm::primary_path(prim, l, r, mask, lc.mw, lc.mh, lane);
        lens_started++;
        if (left) lens_left++;
      }
      const lfm::lanemask lm = __ballot(left);
      if (lm) {
        if (left) {
          const int slot = q_count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
          s_qf[wv][0][slot] = r.px; s_qf[wv][1][slot] = r.py; s_qf[wv][2][slot] = r.hz;
          s_qf[wv][3][slot] = r.dx; s_qf[wv][4][slot] = r.dy; s_qf[wv][5][slot] = r.dz;
          s_qf[wv][6][slot] = __fdiv_rn(r.wn, r.wd);
          s_qt[wv][slot] = (unsigned)lane | ((unsigned)li << 6) | ((unsigned)sample << 10);
        }
        q_count += __popcll(lm);
        __builtin_amdgcn_wave_barrier();
        if (q_count >= 64) flush(64);
      }
    }
    if (sample == next_check) {  // sample > 1 && sample % samples_per_batch == 0 (pathtracer.cpp:862-868), every ray of the batch in
      next_check += a.samples_per_batch;
      while (q_count > 0) flush(q_count < 64 ? q_count : 64);
      if (active) {
        finish_sample();
        const float sd = (float)sqrt(1.0 / (sample - 1) * (double)(s2 - s1 * s1 / (float)sample));
        const float ci = (float)(1.96 * (double)sd / sqrt((double)sample));
        if ((double)ci <= a.max_tolerance * (double)s1 / (double)sample) { active = false; final_sample = sample; }
      }
    }
  }
  while (q_count > 0) flush(q_count < 64 ? q_count : 64);
  finish_sample();
  if (mine) {
    if (active) final_sample = ns_aa + 1;              // the loop ran to its end (:875)
    const size_t p = (size_t)y * a.W + x;
    const double rc = 1. / (double)final_sample;
    scene[3 * p] = total.x * rc;
    scene[3 * p + 1] = total.y * rc;
    scene[3 * p + 2] = total.z * rc;
  }
  if (tally.rays) atomicAdd(&s_cnt[0], (unsigned long long)tally.rays);
  if (tally.isects) atomicAdd(&s_cnt[1], (unsigned long long)tally.isects);
  if (lens_started) atomicAdd(&s_cnt[2], (unsigned long long)lens_started);
  if (lens_left) atomicAdd(&s_cnt[3], (unsigned long long)lens_left);
  __syncthreads();
  if (threadIdx.x < kSceneCounters && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
}

// ---- single-ray forms of the integrator's public members (pathtracer.h:66-77) ----------------
struct LfProbeRay { double o[3], d[3], min_t, max_t; };
// est_radiance_global_illumination(r) (:282-302) and the closest hit behind autofocus (:1065-1072):
// out = {hit, t, n xyz, radiance rgb}
__global__ void k_scene_trace_ray(LfSceneDev sc, LfEnvDev ev, int hemisphere, LfProbeRay pr,
                                  int ns_area_light, uint64_t seq, uint64_t key, double* __restrict__ out) {
  __shared__ int s_stack[kStackDepth * 256];
  if (threadIdx.x != 0) return;
  DRay r = make_ray(v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.min_t, pr.max_t);
  const uint4 ctr = make_uint4((unsigned)seq, (unsigned)(seq >> 32) | 0x80000000u, 0u, 0u);
  const uint2 k2 = make_uint2((unsigned)key, (unsigned)(key >> 32));
  Hit h;
  V3 L;
  SceneTally tally{0u, 0u};
  if (closest_hit(sc, r, &h, s_stack, tally)) {
    int material;
    const V3 n = hit_normal(sc, r, h, &material);
    out[0] = 1.0; out[1] = h.t; out[2] = n.x; out[3] = n.y; out[4] = n.z;
    L = shade_hit<true>(sc, ev, hemisphere != 0, r, h.t, n, sc.materials[material], kShadeZero | kShadeOne,
                        s_stack, ns_area_light, ctr, k2, tally);
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
  const DRay r = make_ray(v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.min_t, pr.max_t);
  SceneTally tally{0u, 0u};
  const V3 L = shade_hit<true>(sc, ev, hemisphere != 0, r, t, v3(nx, ny, nz), m, what, s_stack, ns_area_light,
                               make_uint4((unsigned)seq, (unsigned)(seq >> 32) | 0x80000000u, 0u, 0u),
                               make_uint2((unsigned)key, (unsigned)(key >> 32)), tally);
  out[0] = L.x; out[1] = L.y; out[2] = L.z;
}

// ---- host: BVH over the primitives ------------------------------------------------------------
// The closest hit does not depend on the tree, only the amount of work does (the
// reference's own tree, scene/bvh.cpp:60-170, is not reproduced).  Binned surface-area heuristic over
// the centroids (16 bins per axis), leaves of <= 4 primitives; a split that would leave one side empty
// (coincident centroids) falls back to the median.
struct Box { double mn[3], mx[3]; };
struct HostPrim { int type, material; double d[18]; Box box; double cen[3]; };
struct HostNode { Box box; int left, right, first, count; };   // leaf: left < 0

void grow(Box& b, const Box& o) {
  for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], o.mn[a]); b.mx[a] = std::max(b.mx[a], o.mx[a]); }
}
Box empty_box() { Box b; for (int a = 0; a < 3; a++) { b.mn[a] = INFINITY; b.mx[a] = -INFINITY; } return b; }
double half_area(const Box& b) {
  const double x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
  return x * y + y * z + z * x;
}

void set_prim_box(HostPrim& p) {
  if (p.type == 0) {
    const double rr = std::fabs(p.d[3]);   // (the test only reads r^2: a negative radius is still a sphere)
    for (int a = 0; a < 3; a++) { p.box.mn[a] = p.d[a] - rr; p.box.mx[a] = p.d[a] + rr; }
  } else {
    for (int a = 0; a < 3; a++) {
      p.box.mn[a] = std::min({p.d[a], p.d[3 + a], p.d[6 + a]});
      p.box.mx[a] = std::max({p.d[a], p.d[3 + a], p.d[6 + a]});
    }
  }
  for (int a = 0; a < 3; a++) p.cen[a] = 0.5 * (p.box.mn[a] + p.box.mx[a]);
}

// returns the node id; *depth = levels below (and including) this node
// (levels_left: how many more levels the SAH may still use below this node; at 0 the subtree is built by
// median splits, whose depth is logarithmic -- a degenerate scene, e.g. geometrically spaced centroids that
// leave all but one primitive in one bin at every level, cannot drive the recursion O(n) deep)
int build_node(std::vector<HostNode>& nodes, std::vector<HostPrim>& prims, int first, int count, bool sah,
               int leaf_max, int* depth, int levels_left = 1 << 30) {
  if (levels_left <= 0) sah = false;
  HostNode nd;
  nd.box = empty_box();
  double cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = first; i < first + count; i++) {
    grow(nd.box, prims[i].box);
    for (int a = 0; a < 3; a++) { cmn[a] = std::min(cmn[a], prims[i].cen[a]); cmx[a] = std::max(cmx[a], prims[i].cen[a]); }
  }
  nd.left = nd.right = -1; nd.first = first; nd.count = count;
  const int id = (int)nodes.size();
  nodes.push_back(nd);
  *depth = 1;
  if (count <= leaf_max) return id;
  int mid = -1;
  if (sah) {
    constexpr int kBins = 16;
    double best = INFINITY; int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; a++) {
      const double ext = cmx[a] - cmn[a];
      if (!(ext > 0) || !std::isfinite(ext)) continue;
      Box bb[kBins]; int bn[kBins];
      for (int k = 0; k < kBins; k++) { bb[k] = empty_box(); bn[k] = 0; }
      const double scale = kBins / ext;
      for (int i = first; i < first + count; i++) {
        const int k = std::min(kBins - 1, std::max(0, (int)((prims[i].cen[a] - cmn[a]) * scale)));
        grow(bb[k], prims[i].box); bn[k]++;
      }
      double right_area[kBins]; int right_n[kBins];
      Box acc = empty_box(); int n = 0;
      for (int k = kBins - 1; k > 0; k--) { grow(acc, bb[k]); n += bn[k]; right_area[k] = n ? half_area(acc) : 0.0; right_n[k] = n; }
      acc = empty_box(); n = 0;
      for (int k = 0; k + 1 < kBins; k++) {   // split between bin k and k + 1
        grow(acc, bb[k]); n += bn[k];
        if (n == 0 || right_n[k + 1] == 0) continue;
        const double cost = half_area(acc) * n + right_area[k + 1] * right_n[k + 1];
        if (cost < best) { best = cost; best_axis = a; best_bin = k; }
      }
    }
    if (best_axis >= 0) {
      const int a = best_axis;
      const double scale = kBins / (cmx[a] - cmn[a]), c0 = cmn[a];
      const int kb = best_bin;
      auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [=](const HostPrim& p) {
        return std::min(kBins - 1, std::max(0, (int)((p.cen[a] - c0) * scale))) <= kb;
      });
      mid = (int)(it - prims.begin());
      if (mid == first || mid == first + count) mid = -1;
    }
  }
  if (mid < 0) {   // median of the centroids along the widest axis
    int axis = 0;
    for (int a = 1; a < 3; a++) if (cmx[a] - cmn[a] > cmx[axis] - cmn[axis]) axis = a;
    mid = first + count / 2;
    std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                     [axis](const HostPrim& a, const HostPrim& b) { return a.cen[axis] < b.cen[axis]; });
  }
  int dl = 0, dr = 0;
  const int l = build_node(nodes, prims, first, mid - first, sah, leaf_max, &dl, levels_left - 1);
  const int r = build_node(nodes, prims, mid, first + count - mid, sah, leaf_max, &dr, levels_left - 1);
  nodes[id].left = l; nodes[id].right = r;
  *depth = 1 + std::max(dl, dr);
  return id;
}

float round_down(double x) { const float f = (float)x; return (double)f > x ? std::nextafter(f, -INFINITY) : f; }
float round_up(double x) { const float f = (float)x; return (double)f < x ? std::nextafter(f, INFINITY) : f; }

// the device form: one LfBvhNode per inner node, holding its children's boxes (rounded outward) and
// either their node index or their leaf range; depth-first, so a subtree is contiguous
int flatten(const std::vector<HostNode>& h, int id, std::vector<LfBvhNode>& out) {
  const int me = (int)out.size();
  out.emplace_back();
  std::memset(&out[me], 0, sizeof(LfBvhNode));
  const int kids[2] = {h[id].left, h[id].right};
  for (int k = 0; k < 2; k++) {
    const HostNode& c = h[kids[k]];
    float* lo = k == 0 ? out[me].lo0 : out[me].lo1;
    float* hi = k == 0 ? out[me].hi0 : out[me].hi1;
    for (int a = 0; a < 3; a++) { lo[a] = round_down(c.box.mn[a]); hi[a] = round_up(c.box.mx[a]); }
    const int ref = c.left < 0 ? ~(c.first * 4 + (c.count - 1)) : flatten(h, kids[k], out);
    out[me].child[k] = ref;   // (out may have been reallocated by the recursion: index, not pointer)
  }
  return me;
}

// one wave writes `n_out` (<= 8) doubles, which come straight back
template <typename Launch>
static lf_status scene_probe(lf_ctx* ctx, int n_out, double* out, Launch launch) {
  if (!ctx->probe_dev) LF_HIP(ctx, hipMalloc((void**)&ctx->probe_dev, sizeof(double) * 8));
  launch(ctx->probe_dev);
  LF_HIP(ctx, hipGetLastError());
  LF_HIP(ctx, hipMemcpyAsync(out, ctx->probe_dev, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
  if ((size_t)n_spheres + (size_t)n_triangles >= (1u << 29))
    return lf_fail(ctx, LF_ERR_INVALID, "scene: more than 2^29 primitives");
  std::vector<HostPrim> prims;
  prims.reserve((size_t)n_spheres + (size_t)n_triangles);
  for (int i = 0; i < n_spheres; i++) {
    HostPrim p; std::memset(&p, 0, sizeof(p));
    p.type = 0; p.material = sphere_material[i];
    for (int k = 0; k < 4; k++) p.d[k] = spheres[4 * i + k];
    p.d[4] = p.d[3] * p.d[3];  // Sphere::r2
    prims.push_back(p);
  }
  for (int i = 0; i < n_triangles; i++) {
    HostPrim p; std::memset(&p, 0, sizeof(p));
    p.type = 1; p.material = tri_material[i];
    for (int k = 0; k < 9; k++) { p.d[k] = tri_positions[9 * i + k]; p.d[9 + k] = tri_normals[9 * i + k]; }
    prims.push_back(p);
  }
  for (auto& p : prims) {
    if (p.material < 0 || p.material >= n_materials) return lf_fail(ctx, LF_ERR_INVALID, "scene: material index out of range");
    for (int k = 0; k < (p.type == 0 ? 4 : 18); k++)
      if (!std::isfinite(p.d[k])) return lf_fail(ctx, LF_ERR_INVALID, "scene: a primitive has a non-finite coordinate");
    set_prim_box(p);
  }
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
  // The device traversal defers at most one child per level on its LDS stack; a deeper tree would
  // have to drop children (= lose geometry silently).  The SAH tree of a degenerate scene can be deep:
  // then the median split (ceil(log2(n / 4)) + 1 levels) is built instead; if even that is too deep
  // (> 2^25 primitives) the call fails -- it must fail, not drop.
  std::vector<HostNode> hnodes;
  std::vector<LfBvhNode> nodes;
  Box root = empty_box();
  if (!prims.empty()) {
    int depth = 0;
    bool sah = !ctx->bvh_median;                                       // (lf_test_knob: the round-2 tree, A/B measurements)
    const int leaf_max = std::min(4, std::max(1, ctx->bvh_leaf_max));
    // the SAH may use 64 levels (a well-behaved scene needs ~log2(n) + a few); below them a subtree is
    // built by median splits: the host recursion is bounded by 64 + log2(n) frames whatever the scene, and a
    // tree that ends up deeper than the device's stack is rebuilt by median splits alone (below)
    const int sah_levels = 64;
    build_node(hnodes, prims, 0, (int)prims.size(), sah, leaf_max, &depth, sah_levels);
    if (sah && depth - 1 > kStackDepth) {
      hnodes.clear();
      build_node(hnodes, prims, 0, (int)prims.size(), false, leaf_max, &depth);
    }
    if (depth - 1 > kStackDepth)
      return lf_fail(ctx, LF_ERR_INVALID, "scene: BVH deeper than the device traversal stack");
    ctx->scene_tree_depth = depth;
    root = hnodes[0].box;
    if (hnodes[0].left >= 0) {
      flatten(hnodes, 0, nodes);
    } else {   // a single leaf: a root whose first child is that leaf
      LfBvhNode e; std::memset(&e, 0, sizeof(e));
      for (int a = 0; a < 3; a++) { e.lo0[a] = round_down(root.mn[a]); e.hi0[a] = round_up(root.mx[a]); }
      e.child[0] = ~(0 * 4 + (hnodes[0].count - 1)); e.child[1] = kLfNoChild;
      nodes.push_back(e);
    }
  } else {   // an empty scene: a root without children
    LfBvhNode e; std::memset(&e, 0, sizeof(e));
    e.child[0] = e.child[1] = kLfNoChild;
    nodes.push_back(e);
    ctx->scene_tree_depth = 0;
    for (int a = 0; a < 3; a++) { root.mn[a] = 0; root.mx[a] = 0; }
    root.mn[0] = 1; root.mx[0] = -1;   // (what lf_scene_bounds reported for it before)
  }
  std::vector<LfPrim> dprims(prims.size());
  std::vector<LfPrimNormals> dnormals(prims.size());
  for (size_t i = 0; i < prims.size(); i++) {
    std::memset(&dprims[i], 0, sizeof(LfPrim));
    for (int k = 0; k < 9; k++) { dprims[i].d[k] = prims[i].d[k]; dnormals[i].n[k] = prims[i].d[9 + k]; }
    if (prims[i].type == 1)   // a triangle travels as p0, e1 = p1 - p0, e2 = p2 - p0 (triangle.cpp:29-30)
      for (int k = 0; k < 3; k++) {
        volatile double e1 = prims[i].d[3 + k] - prims[i].d[k], e2 = prims[i].d[6 + k] - prims[i].d[k];
        dprims[i].d[3 + k] = e1; dprims[i].d[6 + k] = e2;
      }
    dprims[i].type = prims[i].type; dprims[i].material = prims[i].material;
  }
  // upload into a fresh set of buffers and swap only when every one of them arrived: a failure half way
  // leaves the previous scene in place and valid, never a descriptor with null tables
  LfSceneDev N;
  std::memset(&N, 0, sizeof(N));
  auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 64));
    if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = up((void**)&N.nodes, nodes.data(), nodes.size() * sizeof(LfBvhNode));
  if (e == hipSuccess) e = up((void**)&N.prims, dprims.data(), dprims.size() * sizeof(LfPrim));
  if (e == hipSuccess) e = up((void**)&N.normals, dnormals.data(), dnormals.size() * sizeof(LfPrimNormals));
  if (e == hipSuccess) e = up((void**)&N.materials, mats.data(), mats.size() * sizeof(LfMaterial));
  if (e == hipSuccess) e = up((void**)&N.lights, lts.data(), lts.size() * sizeof(LfLight));
  if (e != hipSuccess) {
    void* fresh[] = {N.nodes, N.prims, N.normals, N.materials, N.lights};
    for (void* o : fresh) if (o) (void)hipFree(o);
    return lf_fail(ctx, LF_ERR_HIP, std::string("lf_set_scene: upload failed (the previous scene stays): ") + hipGetErrorString(e));
  }
  LfSceneDev& S = ctx->scene_dev;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));   // kernels in flight may still walk the old tables
  void* old[] = {S.nodes, S.prims, S.normals, S.materials, S.lights};
  for (void* o : old) if (o) (void)hipFree(o);
  S = N;
  for (int a = 0; a < 3; a++) { ctx->scene_bmin[a] = root.mn[a]; ctx->scene_bmax[a] = root.mx[a]; }
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
  const bool lens = ctx->lenscam_mode != 0;
  if (lens) {
    // the lens camera's samples are the march's (a counter RNG): the MT19937 table of the parity mode
    // holds two draws per sample, the pupil point needs two more
    if (ctx->jitter_mode == 0)
      return lf_fail(ctx, LF_ERR_INVALID, "the lens camera samples with the counter RNG of the march "
                                          "(lf_set_jitter_counter); MT19937 parity mode is the pinhole's");
    const lf_status st = lf_lenscam_prepare(ctx);
    if (st != LF_OK) return st;
  }
  const size_t n = (size_t)ctx->W * ctx->H * 3;
  if (!ctx->scene) {
    LF_HIP(ctx, hipMalloc((void**)&ctx->scene, n * sizeof(double)));
    LF_HIP(ctx, hipMemsetAsync(ctx->scene, 0, n * sizeof(double), ctx->stream));
  }
  const size_t px = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
  if (px == 0) return LF_OK;
  const bool soft = ctx->scene_dev.n_soft_lights > 0 || ctx->env_dev.w > 0 || ctx->hemisphere_sample;
#define LF_LAUNCH_SCENE(SOFT, LENS)                                                                      \
  hipLaunchKernelGGL((k_scene_term<SOFT, LENS>), dim3((unsigned)((ctx->W + 31) / 32), (unsigned)(((ctx->y1 + 7) >> 3) - (ctx->y0 >> 3))), \
                     dim3(256), 0, ctx->stream, ctx->scene_dev, ctx->env_dev, ctx->cam, pa, ctx->y0, ctx->y1, \
                     lf_deal_of(ctx), ctx->jitter_aa_raw, lc, ctx->primary_dev,                           \
                     ctx->ap[LF_APERTURE_STARBURST].texels, ctx->scene_counters_dev, ctx->scene)
  ScenePixelArgs pa;
  pa.W = ctx->W; pa.H = ctx->H; pa.ns_aa = ctx->ns_aa; pa.ns_area_light = ctx->ns_area_light;
  pa.samples_per_batch = ctx->samples_per_batch; pa.jitter_mode = ctx->jitter_mode;
  pa.hemisphere = ctx->hemisphere_sample ? 1 : 0; pa.max_tolerance = ctx->max_tolerance; pa.key = ctx->jitter_key;
  LfLensCamArgs lc;
  lf_fill_lenscam_args(ctx, &lc);
  hipEvent_t ev = lf_timing_begin(ctx, LFK_SCENE);
#define LF_LAUNCH_SCENE_LENS(SOFT)                                                                         \
  hipLaunchKernelGGL((k_scene_lens<SOFT>), dim3((unsigned)((lens_tiles_x + 3) / 4), (unsigned)(((ctx->y1 + 7) >> 3) - (ctx->y0 >> 3))), \
                     dim3(256), (size_t)std::min(kStackDepth, std::max(1, ctx->scene_tree_depth + 1)) * 256 * sizeof(int), \
                     ctx->stream, ctx->scene_dev, ctx->env_dev, ctx->cam, pa, ctx->y0, ctx->y1,           \
                     lf_deal_of(ctx), lens_mxs, lc, ctx->primary_dev,                                      \
                     ctx->ap[LF_APERTURE_STARBURST].texels, ctx->scene_counters_dev, ctx->scene)
  // a small tree is resident in the caches whichever lanes walk it: the wave takes the march's strided tile (see the
  // kernel)
  int lens_mxs = ctx->scene_dev.n_prims <= kLensStridedMaxPrims ? lc.xs : 0;
  if (ctx->scene_lens_strided >= 0) lens_mxs = ctx->scene_lens_strided ? lc.xs : 0;     // (lf_test_knob)
  const int lens_tiles_x = ((ctx->W + (8 << lens_mxs) - 1) >> (3 + lens_mxs)) << lens_mxs;   // wave tiles along x
  // (lf_test_knob("scene_compact", 0): the round-4 kernel, one traversal per lane's own sample -- kept as the A/B of the tests)
  const bool compact = ctx->scene_compact != 0;
  if (lens && compact) { if (soft) LF_LAUNCH_SCENE_LENS(true); else LF_LAUNCH_SCENE_LENS(false); }
  else if (lens) { if (soft) LF_LAUNCH_SCENE(true, true); else LF_LAUNCH_SCENE(false, true); }
  else { if (soft) LF_LAUNCH_SCENE(true, false); else LF_LAUNCH_SCENE(false, false); }
  lf_timing_end(ctx, LFK_SCENE, ev);
#undef LF_LAUNCH_SCENE
#undef LF_LAUNCH_SCENE_LENS
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

}  // extern "C"
