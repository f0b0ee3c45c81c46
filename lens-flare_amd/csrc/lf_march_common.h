// lf_march_common.h -- what the two march kernels share (lf_march.hip: k_march, the path-tree walk of every
// sample; lf_cull.hip: k_march_cull, the march of the paths a pre-pass found able to reach the light): the
// launch arguments, the scalar-cache row loads and the sun's lobe.  Device code; nothing here is part of the ABI.
#pragma once

#include "lf_internal.h"
#include "lf_march_events.h"

namespace lfm {

// The sun's lobe: q = (1 - cos theta) / (1 - cos alpha), theta between the ray and the sun.
// 1 - d.s would cancel (d.s ~ 0.999: an absolute error of 1e-7 is 1e-4 of a 0.05 rad lobe, and the
// float unit vectors are only unit to 1e-7 as well), so
//   1 - cos theta = sin^2 theta / (1 + cos theta) = |d x s|^2 / (|d|^2 |s|^2 + sqrt(|d|^2 |s|^2) d.s)
// which has no cancellation and does not assume |d| = |s| = 1 (ss = |s|^2 from the host).  Valid for
// d.s > 0, which the pre-test guarantees.  (tests/test_gpu_march_f64.py checks the march against an
// independent float64 tracer; with the plain 1 - d.s the pixels were off by up to 1e-2.)
__device__ __forceinline__ float lobe_q(float dx, float dy, float dz, float sx, float sy, float sz,
                                        float ss, float inv_1mc) {
  const float cg = fmaf(dx, sx, fmaf(dy, sy, dz * sz));
  const float cx = fmaf(dy, sz, -(dz * sy)), cy = fmaf(dz, sx, -(dx * sz)), cz = fmaf(dx, sy, -(dy * sx));
  const float c2 = fmaf(cx, cx, fmaf(cy, cy, cz * cz));
  const float dd = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
  const float ds = dd * ss;
  const float den = fmaf(lf_sqrt(ds), cg, ds);
  return __fdiv_rn(c2, den) * inv_1mc;
}

struct MarchArgs {
  int mw, mh, W, H, y0, y1;
  int spp, G;          // G x G pupil strata, G = floor(sqrt(spp))
  float inv_G;
  int sub_bits;        // each stratum is split into 2^sub_bits x 2^sub_bits sub-cells
  float inv_sub;
  int trow0, tperiod;  // tile rows handled: trow0 + j * tperiod, j = 0 ..   (the frame dealt by tile rows)
  LfDeal deal;         // ... or by 64 x 64-pixel blocks (deal.bx > 0, lf_set_block_deal): own block k = rank + k n, 64 wave tiles each
  int sgroups;         // the samples of a tile are split over this many workgroups (power of two)
  int tail_from, tail_groups;   // k_march_cull: the launch's LAST tiles (slots >= tail_from, a multiple of 64) split over tail_groups
                       // workgroups each, so that the launch ends on short workgroups (0 / 1: no tail); the rest as `sgroups` says
  uint2 key;
  float inv_stop_h;    // 1 / stop_h (correctly rounded)
  float half_w, half_h;  // 0.5 * W, 0.5 * H
  float vz;            // pupil_z - z_sensor
  float lobe_thr;      // d.s above this may lie inside the sun's lobe (conservative, see lfk_march)
  int accumulate;      // add the launch's pixels to the ghost buffer instead of replacing them
  int n_tiles;         // wave tiles of the launch (the grid holds them padded to a multiple of 64)
  int xs;              // log2 of the lanes' pixel stride in x (lf_set_tile_stride): 0 = an 8 x 8 block of
                       // adjacent pixels per wave, 3 = columns 8 apart (a 64 x 8 block shared by 8 waves)
};

// wave tile `tile_lin` of the launch -> its position along x (tx: see k_march) and its tile row.  Dealt by rows: the launch's
// tile rows one after the other; by blocks: the 8 x 8 wave tiles of own block tile_lin / 64 (8 tile rows x 8 tiles across its
// 64 columns, whatever the tile stride).  All wave-uniform.
__device__ __forceinline__ void march_tile_of(const MarchArgs& a, int tile_lin, int tiles_x, int& tx, int& trow) {
  if (a.deal.bx > 0) {
    const int b = a.deal.rank + (tile_lin >> 6) * a.deal.n, w = tile_lin & 63;
    const int by = b / a.deal.bx;
    trow = by * 8 + (w >> 3);
    tx = (b - by * a.deal.bx) * 8 + (w & 7);
  } else {
    const int tj = tile_lin / tiles_x;
    tx = tile_lin - tj * tiles_x;
    trow = a.trow0 + tj * a.tperiod;
  }
}
// does this launch render pixel row y / pixel (x, y)?  (k_march_finish, k_scale_rows)
__device__ __forceinline__ bool march_owns(const MarchArgs& a, int x, int y) {
  if (a.deal.bx > 0) return lf_deal_mine(a.deal, x, y);
  const int t = y >> 3;
  return t >= a.trow0 && (t - a.trow0) % a.tperiod == 0;
}

// The program of a GROUP of up to 3 wavelengths, in two levels (LfProgHdr / LfProgRow in
// lf_internal.h): per row a 16-byte header (ONE s_load_dwordx4), per distinct (interface, direction)
// a 64-byte record (ONE s_load_dwordx16: the geometry once, the index ratios of each wavelength of
// the group).  A header names its own record and the NEXT row's, so stepping to the next row issues
// both loads at once; only a jump (a wave that died as a whole) loads header, then record.
typedef int lf_i16 __attribute__((ext_vector_type(16)));
typedef int lf_i4 __attribute__((ext_vector_type(4)));
typedef const lf_i16 __attribute__((address_space(4))) * lf_const_prow_ptr;
typedef const lf_i4 __attribute__((address_space(4))) * lf_const_phdr_ptr;
__device__ __forceinline__ LfProgRow load_prec(const LfProgRow* __restrict__ base, unsigned off) {
  // (base + 32-bit byte offset: the load takes it as its SGPR offset)
  typedef const char __attribute__((address_space(4))) * cptr;
  const lf_i16 v = *(lf_const_prow_ptr)((cptr)(base) + off);
  LfProgRow r;
  r.dzv = __int_as_float(v[0]); r.curv = __int_as_float(v[1]); r.h2 = __int_as_float(v[2]);
  r.sc = __int_as_float(v[3]); r.sgn = __int_as_float(v[4]);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    r.delta[j] = __int_as_float(v[5 + j]); r.cn22[j] = __int_as_float(v[8 + j]); r.rn2[j] = __int_as_float(v[12 + j]);
  }
  r.ch = __int_as_float(v[11]); r.c2 = __int_as_float(v[15]);
  return r;
}
__device__ __forceinline__ LfWeightRow load_wrec(const LfWeightRow* __restrict__ base, unsigned off) {
  typedef const char __attribute__((address_space(4))) * cptr;
  const lf_i16 v = *(lf_const_prow_ptr)((cptr)(base) + off);
  LfWeightRow r;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    r.fs[j] = __int_as_float(v[j]); r.fo[j] = __int_as_float(v[4 + j]); r.fi[j] = __int_as_float(v[8 + j]);
  }
  r.pad0 = r.pad1 = r.pad2 = 0.0f;
  r.pad3[0] = r.pad3[1] = r.pad3[2] = r.pad3[3] = 0.0f;
  return r;
}
__device__ __forceinline__ LfProgHdr load_phdr(const LfProgHdr* __restrict__ base, unsigned off) {
  typedef const char __attribute__((address_space(4))) * cptr;
  const lf_i4 v = *(lf_const_phdr_ptr)((cptr)(base) + off);
  LfProgHdr h;
  h.flags = v[0]; h.skip = v[1]; h.rec = v[2]; h.rec_next = v[3];
  return h;
}


}  // namespace lfm
