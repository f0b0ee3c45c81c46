// lf_march.hip -- the geometric lens march (north-star path; no reference counterpart:
// the reference's lens is paraxial, pathtracer.cpp:511-689, and its thin-lens camera is a stub,
// camera_lens.cpp:22-30).
//
// For every sensor sample (pixel, sample index) a ray is started on the sensor towards a point
// of the rear pupil and marched BACKWARDS (against the light) through the stack of spherical
// interfaces, once per wavelength and per ghost pair (i, j):
//     refract N-1 .. i+1, reflect at i, refract i+1 .. j-1, reflect at j, refract j-1 .. 0
// (N + 2(j-i) surface events; the primary path is N refractions).  Each event = intersection +
// semi-aperture test + Snell refraction or mirror reflection + unpolarised Fresnel weight; the
// stop is a flat pass-through interface whose event is the aperture-mask lookup.  A ray that
// leaves the front element collects the sun's radiance through a smooth angular lobe.
//
// Arithmetic contract (DESIGN.md "march arithmetic"): float32, every multiply-add written as an
// explicit fmaf, IEEE-correct division and square root (__fdiv_rn / lf_sqrt), no other libm.
// Contributions are accumulated as 2^-36 fixed point in 64-bit integers, so the result does not
// depend on the order in which lanes finish.  The CPU oracle (oracle/lf_geo_oracle.c) follows the
// same contract, which makes pixels and event counters comparable bit for bit.
#include <cstring>

#include "lf_internal.h"

namespace {

constexpr float kFixScale = 68719476736.0f;  // 2^36
constexpr unsigned kDomainMarch = 0x6e5f1a2eu;

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

// IEEE correctly rounded sqrt: plain sqrt under -fhip-fp32-correctly-rounded-divide-sqrt.
// (ROCm's __fsqrt_rn maps to the 1-ulp hardware approximation unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined, which broke bit parity with the CPU in 1 ray of 10^4.)
__device__ __forceinline__ float lf_sqrt(float x) { return __builtin_sqrtf(x); }

__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

struct Ray {
  float px, py, pz, dx, dy, dz, w;
};

enum { EV_REFRACT = 0, EV_REFLECT = 1 };
// death causes (per-lane bookkeeping for the counters)
enum { ALIVE = 0, DEAD_STOP = 1, DEAD_VIGNETTE = 2, DEAD_TIR = 3 };

// one glass-surface event; returns the new liveness
__device__ __forceinline__ int surface_event(Ray& r, float zv, float c, float h2, float eta,
                                             int mode, bool fwd) {
  const float oz = r.pz - zv;
  const float od = fmaf(r.px, r.dx, fmaf(r.py, r.dy, oz * r.dz));
  const float oo = fmaf(r.px, r.px, fmaf(r.py, r.py, oz * oz));
  const float F = fmaf(c, oo, -2.0f * oz);
  const float G = fmaf(-c, od, r.dz);
  const float cF = c * F;
  const float disc = fmaf(G, G, -cF);
  if (disc < 0.0f) return DEAD_VIGNETTE;
  const float sq = lf_sqrt(disc);
  const float den = fwd ? G + sq : G - sq;
  const float t = __fdiv_rn(F, den);
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py), hz = fmaf(t, r.dz, oz);
  const float r2 = fmaf(hx, hx, hy * hy);
  if (!(r2 <= h2)) return DEAD_VIGNETTE;
  const float nx = -c * hx, ny = -c * hy, nz = fmaf(-c, hz, 1.0f);
  const float mu = fmaf(r.dx, nx, fmaf(r.dy, ny, r.dz * nz));
  const float ci = fabsf(mu);
  const float s2 = fmaf(-mu, mu, 1.0f);
  const float k2 = fmaf(-(eta * eta), s2, 1.0f);
  float R = 1.0f, ct = 0.0f;
  if (k2 >= 0.0f) {
    ct = lf_sqrt(k2);
    const float a = fmaf(eta, ci, -ct), b = fmaf(eta, ci, ct);
    const float e = fmaf(-eta, ct, ci), f = fmaf(eta, ct, ci);
    const float af = a * f, eb = e * b, bf = b * f;
    R = __fdiv_rn(0.5f * fmaf(af, af, eb * eb), bf * bf);
  } else if (mode == EV_REFRACT) {
    return DEAD_TIR;
  }
  if (mode == EV_REFRACT) {
    r.w *= (1.0f - R);
    const float g = fmaf(-eta, mu, copysignf(ct, mu));
    r.dx = fmaf(eta, r.dx, g * nx);
    r.dy = fmaf(eta, r.dy, g * ny);
    r.dz = fmaf(eta, r.dz, g * nz);
  } else {
    r.w *= R;
    const float m2 = -2.0f * mu;
    r.dx = fmaf(m2, nx, r.dx);
    r.dy = fmaf(m2, ny, r.dy);
    r.dz = fmaf(m2, nz, r.dz);
  }
  r.px = hx; r.py = hy; r.pz = zv + hz;
  return ALIVE;
}

// the stop: flat pass-through, clipped by its housing and by the aperture mask
__device__ __forceinline__ int stop_event(Ray& r, float zv, float h2, float inv_h,
                                          const float* __restrict__ mask, int mw, int mh) {
  const float t = __fdiv_rn(zv - r.pz, r.dz);
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py);
  const float r2 = fmaf(hx, hx, hy * hy);
  if (!(r2 <= h2)) return DEAD_STOP;
  const float fu = fmaf(hx, inv_h, 1.0f) * (0.5f * (float)mw);
  const float fv = fmaf(hy, inv_h, 1.0f) * (0.5f * (float)mh);
  int ix = (int)fu, iy = (int)fv;
  ix = min(max(ix, 0), mw - 1);
  iy = min(max(iy, 0), mh - 1);
  const float a = mask[iy * mw + ix];
  if (!(a > 0.0f)) return DEAD_STOP;
  r.w *= a;
  r.px = hx; r.py = hy; r.pz = zv;
  return ALIVE;
}

struct LaneStats {
  unsigned events = 0, launched = 0, clip = 0, vign = 0, tir = 0, scene = 0, light = 0;
};

__global__ __launch_bounds__(256) void k_march(const LfLensDev* __restrict__ lens,
                                               const LfPairsDev* __restrict__ pairs,
                                               const float* __restrict__ mask, int mw, int mh,
                                               int W, int H, int y0, int y1, int spp, int ppb,
                                               uint2 key, double* __restrict__ ghost,
                                               unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long s_acc[256 * 3];
  __shared__ unsigned long long s_cnt[8];
  const int tid = threadIdx.x;
  for (int i = tid; i < ppb * 3; i += 256) s_acc[i] = 0ull;
  if (tid < 8) s_cnt[tid] = 0ull;
  __syncthreads();

  const size_t band_px = (size_t)(y1 - y0) * W;
  const size_t first_px = (size_t)blockIdx.x * ppb;  // band-relative
  const int n_surf = lens->n_surf, n_lambda = lens->n_lambda, n_pairs = pairs->n;
  const float z_sensor = lens->z_sensor, pitch = lens->pitch, pupil_h = lens->pupil_h;
  const float pupil_z = lens->pupil_z, geom_norm = lens->geom_norm;
  const float inv_stop_h = __fdiv_rn(1.0f, lens->stop_h);
  const float sx = lens->sun_dir[0], sy = lens->sun_dir[1], sz = lens->sun_dir[2];
  const float inv_1mc = lens->sun_inv_one_minus_cos;
  LaneStats st;

  const int block_samples = ppb * spp;
  for (int id = tid; id < block_samples; id += 256) {
    const int lp = id / spp, s = id - lp * spp;
    const size_t bp = first_px + lp;
    if (bp >= band_px) break;
    const size_t p = (size_t)y0 * W + bp;
    const int x = (int)(p % W), y = (int)(p / W);

    // ---- sensor sample -> initial ray ----------------------------------------------------
    const uint4 rnd = philox4x32_10(make_uint4((unsigned)p, (unsigned)s, kDomainMarch, 0u), key);
    const float jx = u01(rnd.x), jy = u01(rnd.y);
    const float pa = fmaf(2.0f, u01(rnd.z), -1.0f), pb = fmaf(2.0f, u01(rnd.w), -1.0f);
    const float X = -(((float)x + jx) - 0.5f * (float)W) * pitch;
    const float Y = -(((float)y + jy) - 0.5f * (float)H) * pitch;
    // concentric square -> disc map; sin/cos of (pi/4)*t by fixed polynomials (fmaf only)
    float qx = 0.0f, qy = 0.0f;
    if (pa != 0.0f || pb != 0.0f) {
      const bool wide = fabsf(pa) > fabsf(pb);
      const float rr = wide ? pa : pb;
      const float th = 0.78539816339744831f * __fdiv_rn(wide ? pb : pa, rr);
      const float t2 = th * th;
      const float sn = th * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.7557319e-6f, -1.9841270e-4f),
                                                   8.3333333e-3f), -1.6666667e-1f), 1.0f);
      const float cs = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 2.4801587e-5f, -1.3888889e-3f),
                                              4.1666667e-2f), -0.5f), 1.0f);
      qx = wide ? rr * cs : rr * sn;
      qy = wide ? rr * sn : rr * cs;
    }
    const float vx = fmaf(pupil_h, qx, -X), vy = fmaf(pupil_h, qy, -Y), vz = pupil_z - z_sensor;
    const float len = lf_sqrt(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
    const float rl = __fdiv_rn(1.0f, len);
    const float d0x = vx * rl, d0y = vy * rl, d0z = vz * rl;
    const float c2 = d0z * d0z;
    const float w0 = geom_norm * (c2 * c2);

    unsigned long long acc[3] = {0ull, 0ull, 0ull};
    for (int l = 0; l < n_lambda; l++) {
      float lobe_sum = 0.0f;  // per-wavelength radiance factor is applied once per pair below
      (void)lobe_sum;
      for (int q = 0; q < n_pairs; q++) {
        const int pi = pairs->ij[q][0], pj = pairs->ij[q][1];  // wave-uniform
        Ray r{X, Y, z_sensor, d0x, d0y, d0z, w0};
        int dead = ALIVE;
        st.launched++;
        // leg boundaries: backwards N-1..lo1, [reflect i], forwards i+1..j-1, [reflect j],
        // backwards j-1..0.  The primary path is one backward leg N-1..0.
        const int n_legs = pi < 0 ? 1 : 3;
        for (int leg = 0; leg < n_legs && dead == ALIVE; leg++) {
          const bool fwd = (leg == 1);
          int k, k_end, refl;
          if (leg == 0) { k = n_surf - 1; k_end = pi < 0 ? 0 : pi; refl = pi; }
          else if (leg == 1) { k = pi + 1; k_end = pj; refl = pj; }
          else { k = pj - 1; k_end = 0; refl = -1; }
          const int step = fwd ? 1 : -1;
          for (;; k += step) {
            if (fwd ? k > k_end : k < k_end) break;
            const LfSurfaceDev& sf = lens->surf[k];
            if (sf.is_stop != 0.0f) {
              dead = stop_event(r, sf.zv, sf.h2, inv_stop_h, mask, mw, mh);
            } else {
              const float eta = fwd ? sf.eta_fwd[l] : sf.eta_bwd[l];
              dead = surface_event(r, sf.zv, sf.curv, sf.h2, eta,
                                   k == refl ? EV_REFLECT : EV_REFRACT, fwd);
            }
            if (dead != ALIVE) break;
            st.events++;
          }
        }
        if (dead == ALIVE) {
          st.scene++;
          const float cg = fmaf(r.dx, sx, fmaf(r.dy, sy, r.dz * sz));
          const float qq = (1.0f - cg) * inv_1mc;
          if (qq < 1.0f) {
            const float om = 1.0f - qq;
            const float contrib = r.w * (om * om);
            if (contrib > 0.0f) {
              st.light++;
#pragma unroll
              for (int c = 0; c < 3; c++) {
                const float v = contrib * (lens->sun_radiance[c] * lens->lambda_rgb[l][c]);
                acc[c] += (unsigned long long)(v * kFixScale);
              }
            }
          }
        } else if (dead == DEAD_STOP) st.clip++;
        else if (dead == DEAD_VIGNETTE) st.vign++;
        else st.tir++;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
      if (acc[c]) atomicAdd(&s_acc[lp * 3 + c], acc[c]);
  }

  // ---- counters: wave reduce, one LDS add per wave, one global add per workgroup ------------
  unsigned vals[7] = {st.launched, st.events, st.clip, st.vign, st.tir, st.scene, st.light};
#pragma unroll
  for (int i = 0; i < 7; i++) {
    unsigned long long v = vals[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((tid & 63) == 0 && v) atomicAdd(&s_cnt[i], v);
  }
  __syncthreads();
  if (tid < 7 && s_cnt[tid]) atomicAdd(&counters[tid], s_cnt[tid]);

  // ---- one coalesced write of the tile's pixels ---------------------------------------------
  for (int i = tid; i < ppb * 3; i += 256) {
    const size_t bp = first_px + i / 3;
    if (bp < band_px) {
      const size_t p = (size_t)y0 * W + bp;
      ghost[3 * p + (i % 3)] =
          ((double)s_acc[i] * (1.0 / 68719476736.0)) / (double)spp;
    }
  }
}

}  // namespace

// host: derive the per-interface march constants from the raw prescription (float arithmetic,
// mirrored by the oracle)
void lf_derive_lens(lf_ctx* ctx, int n, int stop, int n_lambda, const float* radius,
                    const float* thickness, const float* ior, const float* semi_ap,
                    float sensor_w_mm) {
  LfLensDev& L = ctx->lens;
  // keep the sun / lambda weights across a lens change
  float keep_sun_dir[3], keep_sun_rad[3], keep_inv = L.sun_inv_one_minus_cos;
  for (int c = 0; c < 3; c++) { keep_sun_dir[c] = L.sun_dir[c]; keep_sun_rad[c] = L.sun_radiance[c]; }
  std::memset(&L, 0, sizeof(L));
  for (int c = 0; c < 3; c++) { L.sun_dir[c] = keep_sun_dir[c]; L.sun_radiance[c] = keep_sun_rad[c]; }
  L.sun_inv_one_minus_cos = keep_inv;
  L.n_surf = n; L.stop = stop; L.n_lambda = n_lambda;
  float z = 0.0f;
  for (int k = 0; k < n; k++) {
    LfSurfaceDev& s = L.surf[k];
    s.zv = z;
    z = z + thickness[k];
    s.curv = radius[k] == 0.0f ? 0.0f : 1.0f / radius[k];
    s.h2 = semi_ap[k] * semi_ap[k];
    s.is_stop = (k == stop) ? 1.0f : 0.0f;
  }
  L.z_sensor = z;
  for (int l = 0; l < n_lambda; l++) {
    float n_before = 1.0f;
    for (int k = 0; k < n; k++) {
      float n_after = (k == stop) ? n_before : ior[l * n + k];
      L.surf[k].eta_fwd[l] = n_before / n_after;
      L.surf[k].eta_bwd[l] = n_after / n_before;
      n_before = n_after;
    }
  }
  L.pitch = sensor_w_mm / (float)ctx->W;
  L.pupil_h = semi_ap[n - 1];
  L.pupil_z = L.surf[n - 1].zv;
  double D = (double)L.z_sensor - (double)L.pupil_z;
  L.geom_norm = (float)((3.14159265358979323846 * (double)L.pupil_h * (double)L.pupil_h) / (D * D));
  L.stop_h = stop >= 0 ? semi_ap[stop] : 1.0f;
  for (int l = 0; l < n_lambda; l++)
    for (int c = 0; c < 3; c++) L.lambda_rgb[l][c] = (n_lambda == 3) ? (l == c ? 1.0f : 0.0f)
                                                                      : 1.0f / (float)n_lambda;
}

lf_status lfk_march(lf_ctx* ctx, int spp, uint64_t key) {
  const LfApertureDev& m = ctx->ap[LF_APERTURE_STARBURST];
  LF_HIP(ctx, hipMemcpyAsync(ctx->lens_dev, &ctx->lens, sizeof(LfLensDev), hipMemcpyHostToDevice,
                             ctx->stream));
  LF_HIP(ctx, hipMemcpyAsync(ctx->pairs_dev, &ctx->pairs, sizeof(LfPairsDev), hipMemcpyHostToDevice,
                             ctx->stream));
  const int ppb = spp >= 256 ? 1 : 256 / spp;
  const size_t band_px = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
  if (band_px == 0) return LF_OK;
  const size_t blocks = (band_px + ppb - 1) / ppb;
  if (blocks > 0x7fffffffull) return lf_fail(ctx, LF_ERR_INVALID, "band too large for one launch");
  hipEvent_t ev = lf_timing_begin(ctx, LFK_MARCH);
  hipLaunchKernelGGL(k_march, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->lens_dev,
                     ctx->pairs_dev, m.texels, m.w, m.h, ctx->W, ctx->H, ctx->y0, ctx->y1, spp, ppb,
                     make_uint2((unsigned)key, (unsigned)(key >> 32)), ctx->ghost,
                     ctx->counters_dev);
  lf_timing_end(ctx, LFK_MARCH, ev);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}
