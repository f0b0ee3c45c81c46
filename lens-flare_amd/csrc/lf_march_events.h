// lf_march_events.h -- the surface events of the geometric lens march, shared by
// the ghost march (lf_march.hip: k_march, k_lens_rays) and by the lens-imaged scene term
// (lf_scene.hip: k_scene_term<.., LENS = true>, round 4): ONE definition of the float32 event
// arithmetic, so that the primary path of a sensor sample is the same bits wherever it is marched.
// Device code only; nothing here is part of the ABI.
#pragma once

#include "lf_internal.h"

namespace lfm {

constexpr unsigned kDomainMarch = 0x6e5f1a2eu;
constexpr unsigned kDomainSubcell = 0x51bce110u;

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
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

// Square roots are the hardware's v_sqrt_f32: one transcendental-rate instruction, accurate to 1 ulp.
// A correctly rounded root costs 8 more VALU instructions (the +-1 ulp residual test), and the two
// roots of a surface event would then be 18 of its 48 instructions.  v_sqrt_f32 is deterministic
// and its deviation from the correctly rounded root depends only on the significand and the parity
// of the exponent, so the CPU oracle reproduces it exactly from a table measured once through
// lf_native_sqrt (oracle/lf_geo_oracle.c, geo_set_sqrt_table): the march stays bit-for-bit
// comparable with the oracle.
__device__ __forceinline__ float lf_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Divisions on the per-event / per-sample path are multiplications by the hardware reciprocal v_rcp_f32
// (1 ulp; one quarter-rate instruction + a multiply where the IEEE division is ~10 instructions around the
// same v_rcp: round 4, the stop event's division alone was 2.2 % of the bench frame).  Like v_sqrt_f32 the
// instruction is deterministic and its deviation from the correctly rounded reciprocal depends only on the
// significand, so the CPU oracle follows it exactly through a table measured once (lf_native_rcp,
// geo_set_rcp_table).  The rare divisions whose accuracy matters most -- the weight's wn / wd and the lobe
// factor -- stay IEEE (__fdiv_rn).
__device__ __forceinline__ float lf_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

// The transmitted weight is carried as a fraction wn / wd: every Fresnel factor is a ratio of
// two cheap products, so the march multiplies numerators and denominators separately and divides
// ONCE, and only for the ~0.4 % of rays that end inside the sun's lobe.
// Position: (px, py) and hz = z RELATIVE to the vertex of the interface the ray sits on (0 on the
// sensor / the stop plane), plus r2 = px^2 + py^2 of that point, which the previous event's aperture
// test has already computed and the next event's |o|^2 reuses.  A row carries dzv = (vertex z of the
// interface the ray comes from) - (vertex z of this one): one add gives the origin's z in this
// interface's frame, absolute z never exists (3 vector instructions per event less than tracking it,
// and less cancellation).  (dx, dy, dz) is the OPTICAL direction K = n d (see surface_event).
struct Ray {
  float px, py, hz, r2, dx, dy, dz, wn, wd;
};

// One glass-surface event, straight-line (no divergent branches): a lane that misses the surface,
// leaves the clear aperture or is totally reflected just gets ok = false -- its ray state turns
// into garbage/NaN that nobody reads again.  Liveness is kept as explicit 64-bit wave masks (one
// SGPR pair, plain s_and/s_or), not as per-lane bools: the compiler lowers loop-carried bools to
// exec-merge triples that tripled the scalar-unit load of the loop.  geom_ok tells a vignetted ray
// from a TIR one.
// sgn = +1 for a ray travelling +z, -1 for -z (wave-uniform, lives in an SGPR).
typedef unsigned long long lanemask;

// W = false is the geometry-only march (positions, directions, liveness); W = true additionally
// carries the Fresnel / aperture weight.  The frame runs W = false for every ray and repeats the
// sequence with W = true only for the waves in which some lane ended inside the sun's lobe (~1 % of
// the wave-sequences): the weight is ~16 vector instructions on top of an event's 27 and is read by
// 0.4 % of the rays.  Both instantiations do the same arithmetic on the ray itself, so the
// repeated march reproduces the first one bit for bit.
// ch = c / 2 and c2 = 2 c travel with the row (exact scalings): F = c |o|^2 - 2 o_z is formed as its
// half Fh = fma(ch, |o|^2, -o_z) -- the same bits, shifted by one exponent.
//
// The direction is the OPTICAL direction K = n d (|K| = n, the index of the medium the ray is in: a
// property of the row).  With the ray o + s K the vertex-form quadratic is
//   c n^2 s^2 - 2 s G + F = 0,   G = K_z - c (o . K)
// so disc = G^2 - (c n^2) F, the root next to the vertex is s = (G - sgn sqrt(disc)) R / n^2 for a
// curved interface and s = F / (G + sgn sqrt(disc)) for flat glass, and with N = (-c hx, -c hy,
// 1 - c hz) the unit normal at the hit, K . N = G - c n^2 s = sgn sqrt(disc) EXACTLY: sqrt(disc) is
// n |cos(incidence)|.  Snell: (n' cos t')^2 = disc + (n'^2 - n^2) -- one add, negative = total
// reflection -- and K' = K + sgn (n' cos t' - n cos t) N: no multiplication of K by an index ratio.
// Per event that is 27 vector instructions where the unit-direction form of rounds 1-2 had 31.
// cn22 = 2 c n^2, rn2 = R / n^2, delta = n'^2 - n^2, sc = sgn c come with the row.
// W = true additionally takes the Fresnel scale factors fs, fo, fi of the row (LfWeightRow).
template <bool W>
__device__ __forceinline__ lanemask surface_event(Ray& r, float dzv, float c, float ch, float c2, float sc,
                                                  float cn22, float rn2, float delta, float h2, bool reflect,
                                                  bool flat, float sgn, lanemask& geom_ok, float fs = 1.0f,
                                                  float fo = 1.0f, float fi = 1.0f) {
  const float oz = r.hz + dzv;
  const float od = fmaf(r.px, r.dx, fmaf(r.py, r.dy, oz * r.dz));
  const float oo = fmaf(oz, oz, r.r2);
  const float Fh = fmaf(ch, oo, -oz);            // F / 2, F = c |o|^2 - 2 o_z
  const float G = fmaf(-c, od, r.dz);
  const float cF = cn22 * Fh;                    // = c n^2 F
  const float disc = fmaf(G, G, -cF);
  const float sq = lf_sqrt(disc);                // n |cos(incidence)|
  float t;
  if (flat) t = (Fh + Fh) * lf_rcp(fmaf(sgn, sq, G));   // wave-uniform branch
  else t = fmaf(-sgn, sq, G) * rn2;
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py), hz = fmaf(t, r.dz, oz);
  const float r2 = fmaf(hx, hx, hy * hy);
  // a ray that misses the sphere (disc < 0) has sq = t = r2 = NaN, and NaN <= h2 is false
  geom_ok = __ballot(r2 <= h2);
  lanemask ok = geom_ok;
  float Rn = 0.0f, D = 1.0f, ct = 0.0f;
  bool no_tir = true;
  if (W || !reflect) {
    const float k2 = disc + delta;                 // (n' cos(refraction))^2
    no_tir = k2 >= 0.0f;
    // (a totally reflected ray only survives a mirror event, and only W = true reads ct there)
    ct = lf_sqrt(reflect ? fmaxf(k2, 0.0f) : k2);
    if (W) {
      // unpolarised Fresnel straight from the optical cosines sq = n cos t, ct = n' cos t':
      //   rs = (sq - ct) / (sq + ct),   rp = (n'^2 sq - n^2 ct) / (n'^2 sq + n^2 ct),   R = (rs^2 + rp^2) / 2
      // as ONE fraction R = Rn / D, Rn = ((a B)^2 + (A b)^2) / 2, D = (b B)^2, with the numerators and
      // denominators scaled by row constants so that b = B = 1 at normal incidence (the running
      // denominator of a path stays near 1): fs = 1 / (n + n'), fo = n'^2 / q, fi = n^2 / q,
      // q = n'^2 n + n^2 n'.  No true cosine, no index ratio: nothing is divided by n on the way.
      const float a = (sq - ct) * fs, b = (sq + ct) * fs;
      const float pc = fi * ct;
      const float A = fmaf(fo, sq, -pc), B = fmaf(fo, sq, pc);
      const float u = a * B, v = A * b;
      Rn = 0.5f * fmaf(u, u, v * v);
      const float bB = b * B;
      D = bB * bB;
    }
  }
  if (reflect) {  // wave-uniform: K' = K - 2 (K . N) N, K . N = sgn sqrt(disc)
    if (W) {
      r.wn *= no_tir ? Rn : 1.0f;  // total reflection: R = 1
      r.wd *= no_tir ? D : 1.0f;
    }
    const float m = sq * (c2 * sgn);
    r.dx = fmaf(m, hx, r.dx);
    r.dy = fmaf(m, hy, r.dy);
    r.dz = fmaf(m, hz, fmaf(-2.0f * sgn, sq, r.dz));
  } else {        // K' = K + sgn (ct - sq) N
    ok &= __ballot(no_tir);
    if (W) {
      r.wn *= D - Rn;
      r.wd *= D;
    }
    const float gs = ct - sq;
    const float gcs = gs * sc;                     // sgn (ct - sq) c
    r.dx = fmaf(-gcs, hx, r.dx);
    r.dy = fmaf(-gcs, hy, r.dy);
    r.dz = fmaf(-gcs, hz, fmaf(sgn, gs, r.dz));
  }
  r.px = hx; r.py = hy; r.hz = hz; r.r2 = r2;
  return ok;
}

// the stop: flat pass-through, clipped by its housing and by the aperture mask
template <bool W>
__device__ __forceinline__ lanemask stop_event(Ray& r, float dzv, float h2, float inv_h,
                                               const float* __restrict__ mask, int mw, int mh) {
  const float t = -(r.hz + dzv) * lf_rcp(r.dz);
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py);
  const float r2 = fmaf(hx, hx, hy * hy);
  const float fu = fmaf(hx, inv_h, 1.0f) * (0.5f * (float)mw);
  const float fv = fmaf(hy, inv_h, 1.0f) * (0.5f * (float)mh);
  int ix = (int)fu, iy = (int)fv;  // NaN / out-of-range of a dead lane is clamped, never faults
  ix = min(max(ix, 0), mw - 1);
  iy = min(max(iy, 0), mh - 1);
  const float a = mask[iy * mw + ix];
  if (W) r.wn *= a;
  r.px = hx; r.py = hy; r.hz = 0.0f; r.r2 = r2;
  return __ballot(r2 <= h2) & __ballot(a > 0.0f);
}

// ---- one sensor sample's start ray ("sample") --------------------------------
// What k_march's sample loop computes inline, as a function: the lens camera of the scene term
// (lf_scene.hip) starts its primary path from the SAME sample -- the same Philox block, the same
// stratum and sub-cell of the pupil, the same float expressions -- so that the ray which images the
// scene is bit for bit the ray whose ghosts the march accumulates.
struct SampleSpec {
  int W;               // frame width (the pixel index of the counter is x + y W)
  int G;               // G x G pupil strata; samples s >= G * G are unstratified
  float inv_G;
  int xs;              // log2 of the pixel stride in x of a wave's tile (lf_set_tile_stride)
  int sub_bits;        // 2^sub_bits x 2^sub_bits sub-cells per stratum, drawn per (wave tile, s)
  float inv_sub;
  uint2 key;
  float pitch, half_w, half_h;
  float pupil_h, vz;   // the disc the samples aim at: radius, z - z_sensor
  float geom_norm;     // its solid-angle factor pi h^2 / vz^2
};
struct StartRay { float X, Y, dx, dy, dz, w0; };

// the concentric square -> disc map with the fixed polynomials of the contract
__device__ __forceinline__ void pupil_disc(float pa, float pb, float& qx, float& qy) {
  qx = 0.0f; qy = 0.0f;
  if (pa != 0.0f || pb != 0.0f) {
    const bool wide = fabsf(pa) > fabsf(pb);
    const float rr = wide ? pa : pb;
    const float th = 0.78539816339744831f * ((wide ? pb : pa) * lf_rcp(rr));
    const float t2 = th * th;
    const float sn = th * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.7557319e-6f, -1.9841270e-4f),
                                                 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    const float cs = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.4801587e-5f, -1.3888889e-3f),
                                            4.1666667e-2f), -0.5f), 1.0f);
    qx = wide ? rr * cs : rr * sn;
    qy = wide ? rr * sn : rr * cs;
  }
}

// sensor point X, Y (mm) + pupil square coordinates in [-1, 1]^2 -> unit start direction and weight
__device__ __forceinline__ StartRay aim_at_pupil(float X, float Y, float pa, float pb, float pupil_h,
                                                 float vz, float geom_norm) {
  float qx, qy;
  pupil_disc(pa, pb, qx, qy);
  const float vx = fmaf(pupil_h, qx, -X), vy = fmaf(pupil_h, qy, -Y);
  const float len = lf_sqrt(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
  const float rl = lf_rcp(len);
  StartRay s;
  s.X = X; s.Y = Y; s.dx = vx * rl; s.dy = vy * rl; s.dz = vz * rl;
  const float c2 = s.dz * s.dz;
  s.w0 = geom_norm * (c2 * c2);
  return s;
}

// the wave tile of pixel (x, y) as k_march numbers it: tile row y / 8, and along x block x / (8 * 2^xs) with phase
// x mod 2^xs
__device__ __forceinline__ unsigned sample_tile_id(const SampleSpec& a, int x, int y) {
  const int tiles_x = ((a.W + (8 << a.xs) - 1) >> (3 + a.xs)) << a.xs;
  const int tx = ((x >> (3 + a.xs)) << a.xs) + (x & ((1 << a.xs) - 1));
  return (unsigned)((y >> 3) * tiles_x + tx);
}
// the sub-cell of its stratum that wave tile `tile_id` draws for sample s (< G * G)
__device__ __forceinline__ void sample_subcell(const SampleSpec& a, unsigned tile_id, int s, unsigned& sxi, unsigned& syi) {
  const uint4 r2 = philox4x32_10(make_uint4(tile_id, (unsigned)s, kDomainSubcell, 0u), a.key);
  sxi = a.sub_bits ? (r2.x >> (32 - a.sub_bits)) : 0u;
  syi = a.sub_bits ? (r2.y >> (32 - a.sub_bits)) : 0u;
}
// ... and the sample itself, given its stratum (cx, cy) = (s mod G, s / G) and that sub-cell (all four unused for the
// unstratified samples s >= G * G)
__device__ __forceinline__ StartRay sample_start_in(const SampleSpec& a, int x, int y, int s, int cx, int cy, unsigned sxi,
                                                    unsigned syi) {
  const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
  const uint4 rnd = philox4x32_10(make_uint4(p, (unsigned)s, kDomainMarch, 0u), a.key);
  const float jx = u01(rnd.x), jy = u01(rnd.y);
  float ua = u01(rnd.z), ub = u01(rnd.w);
  if (s < a.G * a.G) {
    ua = ((float)cx + ((float)sxi + ua) * a.inv_sub) * a.inv_G;
    ub = ((float)cy + ((float)syi + ub) * a.inv_sub) * a.inv_G;
  }
  const float pa = fmaf(2.0f, ua, -1.0f), pb = fmaf(2.0f, ub, -1.0f);
  const float X = -(((float)x + jx) - a.half_w) * a.pitch;
  const float Y = -(((float)y + jy) - a.half_h) * a.pitch;
  return aim_at_pupil(X, Y, pa, pb, a.pupil_h, a.vz, a.geom_norm);
}
__device__ __forceinline__ StartRay sample_start(const SampleSpec& a, int x, int y, int s) {
  unsigned sxi = 0u, syi = 0u;
  const int cy = s / a.G, cx = s - cy * a.G;
  if (s < a.G * a.G) sample_subcell(a, sample_tile_id(a, x, y), s, sxi, syi);
  return sample_start_in(a, x, y, s, cx, cy, sxi, syi);
}

// ---- the primary path N-1 .. 0 with its weight (LensCamera::generate_ray) ---------------------------
// One lane = one ray; the interface table (LfPrimaryDev, built by the host with the float arithmetic
// of pack_program) is wave-uniform and arrives through the scalar cache.  Returns whether this
// lane's ray left the front element; r then holds the exit state (hz relative to interface 0's vertex,
// K = the unit direction in air) and wn / wd the transmitted weight.
__device__ __forceinline__ bool primary_path(const LfPrimaryDev* __restrict__ P, int lambda, Ray& r,
                                             const float* __restrict__ mask, int mw, int mh, int lane) {
  { const float ns = P->n_start[lambda]; r.dx *= ns; r.dy *= ns; r.dz *= ns; }   // K = n d
  bool alive = true;
  const int n = P->n;
  for (int e = 0; e < n; e++) {   // wave-uniform
    const LfPrimaryRow& w = P->row[e];
    lanemask ok, geom_ok;
    if (w.kind & LF_EV_STOP) {
      ok = stop_event<true>(r, w.dzv, w.h2, P->inv_stop_h, mask, mw, mh);
    } else {
      ok = surface_event<true>(r, w.dzv, w.curv, w.ch, w.c2, w.sc, w.cn22[lambda], w.rn2[lambda],
                               w.delta[lambda], w.h2, false, (w.kind & LF_EV_FLAT) != 0, -1.0f, geom_ok,
                               w.fs[lambda], w.fo[lambda], w.fi[lambda]);
    }
    alive = alive && ((ok >> lane) & 1ull) != 0ull;
    if (__ballot(alive) == 0ull) break;   // every lane of the wave is blocked (a closed part of the pupil): nothing left to march
  }
  return alive;
}

}  // namespace lfm
