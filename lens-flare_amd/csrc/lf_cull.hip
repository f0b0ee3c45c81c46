// lf_cull.hip -- spend the rays where the light is (rounds 5 and 6).
//
// The north star's loop -- for each sensor sample, enumerate the ghost pairs, march each path, accumulate --
// leaves open which of those marches are worth starting.  On the bench frame 99.4 % of the (wave tile, sample,
// path, wavelength) combinations end with no lane inside the sun's lobe: their paths run into a diaphragm or
// leave the front element pointing elsewhere, and add exactly 0 to the sensor (profiles/r05_pair_table.json).
// Which ones can be known in advance: for a path q the map (sensor point x, pupil point u) -> exit direction
// is smooth, so over a small 4-D box (a block of 64 x 64 sensor pixels times one cell of the pupil square)
// a handful of marched rays bound where the whole box can go -- on every diaphragm of the path and in
// direction space when it leaves the lens.
//
//   k_cull_level     the pre-pass, coarse to fine over the pupil square (levels P = 8 -> 16 -> 32 -> P_final, four
//                    children per kept box, work lists per path): one LANE = one box (sensor block, pupil cell,
//                    path) with its 15 rays in registers -- 13 at the middle wavelength (a 3 x 3 grid over the cell
//                    whose corners sit on the block's corners, + the cell's centre at the block's +-x, +-y edges) and
//                    the centre at both ends of the spectrum -- marched WITHOUT dying on a diaphragm.  The footprint of
//                    a box on an interface is a zonotope (centre + central-difference generators along the two pupil and
//                    two sensor axes, inflated, + a second-order slack); the box is dropped when a separating axis puts
//                    it wholly outside the clear aperture (the stop: outside its housing or on closed cells of the mask's
//                    occupancy grid), at the end the same in direction space against the sun's lobe.  What the samples
//                    CANNOT bound is kept: a box that lost a sample (to total reflection or a missed sphere: the map is not
//                    Lipschitz at that edge), a box whose samples all end unless the bound of the pass scalar stays below
//                    zero.  Result: per (block, cell) a 64-bit mask of the paths that may contribute.
//                    Round 6: the rules are round 5's, the kernel is not -- it builds a footprint only where a test can
//                    fire (the centre sample outside the clear aperture, the stop, the exit) instead of after every event
//                    of every box, needs no scratch, and builds the SAME table bit for bit in half the time (14.1 -> 6.9 ms
//                    on the bench frame); k_cull_level_general keeps round 5's kernel with the rules as arguments, for the
//                    regression test and for the rules that were replaced (lf_test_knob).
//   k_cull_audit     every table is CHECKED where it is used: a ray of every (block, cell, path) box it does not start,
//                    marched with the march's own events; one that reaches the light refutes the table and the launch
//                    marches everything (lf_set_cull_audit).  1.7 ms on the bench frame.
//   k_march_cull<K>  the march of exactly the started paths: per wave tile and sample one scalar load tells which; the
//                    paths of a sample share their common leg from the sensor (march_started_set, round 6: the bench
//                    frame's 5.9e10 events of started paths take 3.3e10 executed ones), geometry first and the Fresnel /
//                    aperture weight by a second march of the lanes that reach the lobe, the event arithmetic being
//                    lf_march_events.h's -- so a started ray is bit for bit the ray k_march and the oracle march, and
//                    since an unstarted one contributes 0 the PIXELS are those of the full enumeration.  Counters count
//                    what was started, every path as if marched alone (the oracle follows the same table:
//                    oracle/lf_geo_oracle.c geo_set_cull).  The row loop is built around the CU's ONE scalar unit, which
//                    it waits for (the row's kind decided once for its K wavelengths, the sequence dword prefetched
//                    unconditionally, one test at the row's end: 36.5 -> 33.1 ms), and a launch ends on short workgroups
//                    (its last tiles split over 4 workgroups whose integer sums meet in a small buffer: MarchArgs::tail_from).
//   k_march_items<K> the same for sampling specifications without pupil sub-cells: (pixel, sample) items compacted
//                    per path by ballot + an LDS prefix sum.
// WHAT THE BOUNDS ARE.  A second-order Taylor estimate of the bundle's map over the box from finite differences of 15
// rays -- the 4 first derivatives and the 4 pure second derivatives measured, of the 6 mixed ones two sums -- with
// factors for what is not measured (x 1.25 on the generators, x 1.2 on the lobe test) that were FOUND: lowered, each
// loses its first lit ray between x 0.9 and x 1.0.  They are not proofs.  Round 6 tried to replace them by proofs
// and by a complete model (profiles/r06_cull_bounds.txt): affine arithmetic on the box itself (every operation of the
// march as a form with a rigorous remainder, private terms folded back into the bundle's frame after every event) is
// sound by construction but its remainders compound over the 11 - 27 events of a path -- at the table's resolution it
// starts 50 % of everything against 7.7 %; a 17-ray stencil that measures all ten second derivatives with a geometric
// estimate of the third order starts 7.6 % and lost light on one of 160 random frames.  So the rules stand as round 5
// left them, on the evidence of the search (tests/cull_fuzz.py: 39 000 + this round's frames, none differing) -- and
// since a search covers what it drew, the AUDIT ships with them.
//
// No reference counterpart: the reference enumerates 13 fixed pairs per channel and draws each as one textured
// quad (src/pathtracer/pathtracer.cpp:735-762, :452-508) -- its "cull" is that a quad covers few pixels.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lf_internal.h"
#include "lf_march_events.h"
#include "lf_march_common.h"

namespace {

using namespace lfm;

// ---- the pre-pass ------------------------------------------------------------------------------------
constexpr int kCullSamples = 15;   // per box, at the middle wavelength: a 3 x 3 grid over the pupil cell (0 .. 8, 4 = the
                                   // centre; the mid-edge ones at the block's centre, the corners on the block's
                                   // corners) + the cell's centre at the block's +x, -x, +y, -y edges (9 .. 12);
                                   // and the centre sample at the first and the last wavelength (13, 14)
struct CullLevelArgs {
  int W, H;
  float pitch, half_w, half_h;
  int blocks_x, blocks_y, blk_log2;
  int share_rank, share_n, share_nb;   // a shared table (lf_cull_row_of_block): the first level runs this rank's blocks only
  int P;                   // pupil cells per axis at THIS level
  int P_final;             // ... of the table (the last level)
  int last;                // the last level writes the table, the others the next level's work list
  int n_paths;
  int lam[3];              // the wavelengths marched: the middle one (13 samples), the first and the last (the centre sample)
  int march_k, prog_recs;  // how the record table is grouped (lf_march.hip pack_program)
  float pupil_h, vz, geom_norm;
  float stop_h, inv_stop_h;
  float sx, sy, rho;       // the sun's direction (x, y) and the lobe's radius in direction space
  float margin;            // footprint inflation at this level
  float geo_margin;        // ... of the zonotope's generators alone (experiments: LF_CULL_GEO_MARGIN)
  int strict;              // footprint tests: 0 every box, 1 (shipped) only boxes with EVERY sample alive, 2 not for boxes that lost samples to
                           // total reflection (a missed sphere has the same square-root edge: two frames of a harsher random draw, wide
                           // suns on a perturbed 8-wavelength prescription, lost 19 and 213 lit rays under 2)
  int strict_lost;         // "all samples end here" by a scalar zonotope bound instead of the range rule
  float lobe_k;            // the footprint in direction space once more inflated for the lobe test (third order: a small lobe sees it)
  int slack_mode;          // experiments: LF_CULL_SLACK (1: second order summed over the four axes + twice the corners' cross terms)
  int keep_partial;        // a box that lost samples (total reflection, a missed sphere) is never dropped by the lobe test
  float lost_rel, lost_abs;  // "every sample ends here" drops a box only beyond this margin (see firmly_lost)
  int disable;             // experiments: bit 0 no aperture test, 1 no mask test, 2 no lobe test, 3 no all-samples-lost test
  unsigned list_stride;    // entries per path in the work lists
  unsigned occ[kCullOcc];  // occupancy rows of the stop mask
};

// A glass event of the pre-pass: the arithmetic of surface_event<false> (lf_march_events.h) WITHOUT a clear aperture
// -- the sample goes on wherever the sphere is -- that also hands out HOW FAR the sample is from being lost:
// disc (< 0: no intersection) and, for a refraction, tir = (n' cos t')^2 / (|disc| + |n'^2 - n^2|) (< 0: total
// reflection), relative and of order 1 away from the boundary.  Not bit-critical: nothing here reaches a pixel.
// (The raw (n' cos t')^2 with one scale per box saves the reciprocal and 1.5 ms of the bench frame's pre-pass but
// loses its first lit ray on 3.6 mm blocks instead of 4.8 mm: profiles/r05_march_variants.txt.)
__device__ __forceinline__ void virtual_event(Ray& r, const LfProgRow& w, float cn22, float rn2, float delta, bool reflect,
                                              bool flat, float& disc_out, float& tir) {
  const float oz = r.hz + w.dzv;
  const float od = fmaf(r.px, r.dx, fmaf(r.py, r.dy, oz * r.dz));
  const float oo = fmaf(oz, oz, fmaf(r.px, r.px, r.py * r.py));
  const float Fh = fmaf(w.ch, oo, -oz);
  const float G = fmaf(-w.curv, od, r.dz);
  const float disc = fmaf(G, G, -(cn22 * Fh));
  disc_out = disc;
  const float sq = lf_sqrt(disc);
  const float t = flat ? (Fh + Fh) * lf_rcp(fmaf(w.sgn, sq, G)) : fmaf(-w.sgn, sq, G) * rn2;
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py), hz = fmaf(t, r.dz, oz);
  if (reflect) {
    tir = 1.0f;
    const float m = sq * (w.c2 * w.sgn);
    r.dx = fmaf(m, hx, r.dx); r.dy = fmaf(m, hy, r.dy); r.dz = fmaf(m, hz, fmaf(-2.0f * w.sgn, sq, r.dz));
  } else {
    const float k2 = disc + delta;
    tir = k2 * lf_rcp(fabsf(disc) + fabsf(delta) + 1e-30f);
    const float gs = lf_sqrt(k2) - sq, gcs = gs * w.sc;
    r.dx = fmaf(-gcs, hx, r.dx); r.dy = fmaf(-gcs, hy, r.dy); r.dz = fmaf(-gcs, hz, fmaf(w.sgn, gs, r.dz));
  }
  r.px = hx; r.py = hy; r.hz = hz;
}

// One LANE = one box (sensor block x pupil cell) of path blockIdx.y; its 13 rays live in registers, so a wave
// marches 64 boxes of ONE path in lockstep -- wave-uniform event sequence, rows through the scalar cache, no
// cross-lane traffic.  (The first version gave a box to a 16-lane row and reduced with ds_bpermute: 19 ms
// for the bench frame's 6e6 boxes at P = 16; profiles/r05_march_variants.txt.)
// Work: `items` = this level's list for the path (cell index = block * P * P + cell; null = every box of the
// level); a box that cannot be ruled out appends its four children to `next` (cells of 2P) or, on the last
// level, sets the path's bit in the table.
#ifndef LF_CULL_WAVES
#define LF_CULL_WAVES 3      // waves per SIMD: 2 / 3 / 4 -> 8.6 / 7.15 / 18.8 ms on the bench frame (193 / 168 / 128 VGPR)
#endif
#ifndef LF_CULL_WG
#define LF_CULL_WG 64        // lanes per workgroup: 256 / 128 / 64 -> 7.18 / 7.11 / 6.95 ms (a wave of decided boxes frees its slot at once)
#endif
__global__ __launch_bounds__(LF_CULL_WG, LF_CULL_WAVES) void k_cull_level_general(const LfLensDev* __restrict__ lens,
                                                    const LfPairsDev* __restrict__ pairs,
                                                    const int* __restrict__ seq_table,
                                                    const LfProgRow* __restrict__ rec_table, CullLevelArgs a,
                                                    const unsigned* __restrict__ items,
                                                    const unsigned* __restrict__ counts, unsigned items_stride,
                                                    unsigned* __restrict__ next, unsigned* __restrict__ next_counts,
                                                    unsigned long long* __restrict__ table,
                                                    unsigned long long* __restrict__ stats) {
  const int q = blockIdx.y;
  const unsigned PP = (unsigned)(a.P * a.P);
  // (first level: every box of the blocks this rank builds -- all of them unless the table is shared)
  const unsigned n_blk = (unsigned)(a.blocks_x * a.blocks_y);
  const unsigned n_mine = (n_blk + (unsigned)a.share_n - 1u - (unsigned)a.share_rank) / (unsigned)a.share_n;
  const unsigned n_items = items ? min(counts[q], items_stride) : n_mine * PP;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  // (whole waves past the end leave; a partial last wave keeps its idle lanes: the loop below is wave-uniform)
  if ((i & ~63u) >= n_items) return;
  const bool valid = i < n_items;
  const unsigned item = valid ? (items ? items[(size_t)q * items_stride + i]
                                       : ((unsigned)a.share_rank + (unsigned)a.share_n * (i / PP)) * PP + i % PP) : 0u;
  const int blk = (int)(item / PP), cell = (int)(item % PP);
  const int ci = cell % a.P, cj = cell / a.P;
  const int bx = blk % a.blocks_x, by = blk / a.blocks_x;

  const float px0 = (float)(bx << a.blk_log2), px1 = fminf((float)a.W, (float)((bx + 1) << a.blk_log2));
  const float py0 = (float)(by << a.blk_log2), py1 = fminf((float)a.H, (float)((by + 1) << a.blk_log2));
  const float Xc = -((0.5f * (px0 + px1)) - a.half_w) * a.pitch, Yc = -((0.5f * (py0 + py1)) - a.half_h) * a.pitch;
  const float hX = 0.5f * (px1 - px0) * a.pitch, hY = 0.5f * (py1 - py0) * a.pitch;
  const float invP = 1.0f / (float)a.P;

  const int n_ev = pairs->ev_cnt[q];
  const int* const seq = seq_table + pairs->ev_off[q];
  const int lane = (int)(threadIdx.x & 63u);
  // ---- the box's 15 rays: 13 at the middle of the spectrum + the centre sample at its two ends ----------------
  // Dispersion moves the whole footprint, monotonically in the wavelength, between where the two ends of the spectrum
  // put it: the box must be ruled out for everything in between, so the centre sample is marched at both ends
  // as well and its largest deviation from the middle one widens every footprint.  (Testing the ends one after the
  // other and dropping the box when EACH misses is wrong -- the lobe may lie between them: 7042 of 1.3e10 lit rays
  // of the 8-wavelength 4K frame were lost that way, found by the full-enumeration comparison.)
  constexpr unsigned kAll = (1u << kCullSamples) - 1u;
  const LfProgRow* recs_of[3];
  int j_of[3];
  float ns_of[3];
#pragma unroll
  for (int w = 0; w < 3; w++) {
    const int l = a.lam[w], g = l / a.march_k;
    j_of[w] = l - g * a.march_k;
    recs_of[w] = rec_table + (size_t)g * (size_t)a.prog_recs;
    ns_of[w] = lens->n_start[l];
  }
  Ray r[kCullSamples];
#pragma unroll
  for (int t = 0; t < kCullSamples; t++) {
    float X = Xc, Y = Yc, fu = 0.5f, fv = 0.5f;
    if (t < 9) {
      fu = 0.5f * (float)(t % 3); fv = 0.5f * (float)(t / 3);
      // the four corners of the pupil cell sit on the four corners of the BLOCK as well (diagonals of the 4-D
      // box): what they deviate from the linear model by holds the cross terms between sensor and pupil
      if ((t % 3) != 1 && (t / 3) != 1) { X = Xc + (float)(t % 3 - 1) * hX; Y = Yc + (float)(t / 3 - 1) * hY; }
    }
    else if (t == 9) X = Xc + hX;
    else if (t == 10) X = Xc - hX;
    else if (t == 11) Y = Yc + hY;
    else if (t == 12) Y = Yc - hY;
    const float ns = t == 13 ? ns_of[1] : t == 14 ? ns_of[2] : ns_of[0];
    const float ua = ((float)ci + fu) * invP, ub = ((float)cj + fv) * invP;
    const StartRay s0 = aim_at_pupil(X, Y, fmaf(2.0f, ua, -1.0f), fmaf(2.0f, ub, -1.0f), a.pupil_h, a.vz, a.geom_norm);
    r[t] = Ray{X, Y, 0.0f, 0.0f, s0.dx * ns, s0.dy * ns, s0.dz * ns, 0.0f, 0.0f};
  }
  // `live`: bit t = sample t is still on the path (not lost to a missed sphere or to total reflection).  A box
  // that has lost samples is PARTIAL: what is left of it lies next to a region where the path ends -- total reflection or
  // the rim of a sphere, either way a square-root edge where the map is not Lipschitz -- and is not bounded by the samples
  // left: no footprint test drops it (strict = 1; the ball footprints below serve strict = 0 / 2, experiments).
  unsigned live = kAll;
  bool culled = !valid, keep = false, partial = false;
  bool tir_partial = false;     // some sample of the box ended by TOTAL REFLECTION: next to that boundary the refracted ray is grazing and
                                // the map unbounded -- what is left of the box cannot be bounded by the samples left (strict = 2)
  int why = 0;
  // Footprint of the box in a plane (an interface's, or direction space).  With every sample in use the
  // image of the box is modelled as a ZONOTOPE: centre c + the four generators g1, g2 (half the cell along the
  // two pupil axes: central differences of the mid-edge samples), gx, gy (half the block along x and y) + an
  // isotropic slack: what the linear model misses (largest deviation of the nine pupil samples and of the
  // edge mid-points from it) and how far the ends of the spectrum move the centre.  Its extent along a unit vector
  // n is sum |g . n|: a separating-axis test against a disc needs only that -- far tighter than a ball around c for
  // the elongated footprints of defocused ghosts.  A box that has lost samples falls back to a ball around a
  // sample still in use, inflated twice as much.
  struct Foot { float cx, cy, g1x, g1y, g2x, g2y, gxx, gxy, gyx, gyy, slack, ball; bool zono; };
  auto footprint = [&](bool dirs, unsigned use, float eps) {
    float vx[kCullSamples], vy[kCullSamples];
#pragma unroll
    for (int t = 0; t < kCullSamples; t++) { vx[t] = dirs ? r[t].dx : r[t].px; vy[t] = dirs ? r[t].dy : r[t].py; }
    Foot f;
    f.zono = use == kAll;
    const int ref = (use & 0x10u) ? 4 : (use ? __ffs((int)use) - 1 : 4);
    f.cx = vx[4]; f.cy = vy[4];
#pragma unroll
    for (int t = 0; t < kCullSamples; t++) if (t != 4 && ref == t) { f.cx = vx[t]; f.cy = vy[t]; }
    f.g1x = 0.5f * (vx[5] - vx[3]); f.g1y = 0.5f * (vy[5] - vy[3]);
    f.g2x = 0.5f * (vx[7] - vx[1]); f.g2y = 0.5f * (vy[7] - vy[1]);
    f.gxx = 0.5f * (vx[9] - vx[10]); f.gxy = 0.5f * (vy[9] - vy[10]);
    f.gyx = 0.5f * (vx[11] - vx[12]); f.gyy = 0.5f * (vy[11] - vy[12]);
    float dev2 = 0.0f, ru2 = 0.0f, rx2 = 0.0f, ry2 = 0.0f, rl2 = 0.0f;
#pragma unroll
    for (int t = 0; t < kCullSamples; t++) {
      const float ex = vx[t] - f.cx, ey = vy[t] - f.cy;
      const float d2 = ((use >> t) & 1u) ? fmaf(ex, ex, ey * ey) : 0.0f;
      if (t < 9) {
        ru2 = fmaxf(ru2, d2);
        const float at = (float)(t % 3 - 1), bt = (float)(t / 3 - 1);
        const bool corner = (t % 3) != 1 && (t / 3) != 1;     // (also displaced to the block's corner)
        const float mx = ex - fmaf(at, f.g1x, bt * f.g2x) - (corner ? fmaf(at, f.gxx, bt * f.gyx) : 0.0f);
        const float my = ey - fmaf(at, f.g1y, bt * f.g2y) - (corner ? fmaf(at, f.gxy, bt * f.gyy) : 0.0f);
        dev2 = fmaxf(dev2, fmaf(mx, mx, my * my));
      } else if (t < 11) rx2 = fmaxf(rx2, d2);
      else if (t < 13) ry2 = fmaxf(ry2, d2);
      else rl2 = fmaxf(rl2, d2);                               // the ends of the spectrum
    }
    {   // the edge mid-points against the centre: second order along x and y
      const float mx = 0.5f * (vx[9] + vx[10]) - vx[4], my = 0.5f * (vy[9] + vy[10]) - vy[4];
      const float nx = 0.5f * (vx[11] + vx[12]) - vx[4], ny = 0.5f * (vy[11] + vy[12]) - vy[4];
      dev2 = fmaxf(dev2, fmaxf(fmaf(mx, mx, my * my), fmaf(nx, nx, ny * ny)));
    }
    const float ru = lf_sqrt(ru2), rl = lf_sqrt(rl2);
    const float rx = (use & 0x600u) ? lf_sqrt(rx2) : ru, ry = (use & 0x1800u) ? lf_sqrt(ry2) : ru;
    f.ball = fmaf(partial ? 2.0f * a.margin : a.margin, ((ru + rx) + ry) + rl, eps);
    f.slack = fmaf(2.0f, lf_sqrt(dev2), fmaf(a.margin, rl, eps));
    if (a.slack_mode == 1) {
      // second order per axis (mid-points of opposite samples against the centre), SUMMED: at the extreme vertex of the
      // box all four add; and what the corners deviate by beyond that sum (cross terms), twice
      const float dax = 0.5f * (vx[5] + vx[3]) - vx[4], day = 0.5f * (vy[5] + vy[3]) - vy[4];
      const float dbx = 0.5f * (vx[7] + vx[1]) - vx[4], dby = 0.5f * (vy[7] + vy[1]) - vy[4];
      const float dxx = 0.5f * (vx[9] + vx[10]) - vx[4], dxy = 0.5f * (vy[9] + vy[10]) - vy[4];
      const float dyx = 0.5f * (vx[11] + vx[12]) - vx[4], dyy = 0.5f * (vy[11] + vy[12]) - vy[4];
      const float sx = (dax + dbx) + (dxx + dyx), sy = (day + dby) + (dxy + dyy);
      float c2 = 0.0f;
#pragma unroll
      for (int t = 0; t < 9; t += 2) {
        if (t == 4) continue;
        const float at = (float)(t % 3 - 1), bt = (float)(t / 3 - 1);
        const float mx = (vx[t] - vx[4]) - fmaf(at, f.g1x + f.gxx, bt * (f.g2x + f.gyx)) - sx;
        const float my = (vy[t] - vy[4]) - fmaf(at, f.g1y + f.gxy, bt * (f.g2y + f.gyy)) - sy;
        c2 = fmaxf(c2, fmaf(mx, mx, my * my));
      }
      const float sum2 = lf_sqrt(fmaf(dax, dax, day * day)) + lf_sqrt(fmaf(dbx, dbx, dby * dby)) +
                         lf_sqrt(fmaf(dxx, dxx, dxy * dxy)) + lf_sqrt(fmaf(dyx, dyx, dyy * dyy));
      f.slack = sum2 + fmaf(2.0f, lf_sqrt(c2), fmaf(a.margin, rl, eps));
    }
    return f;
  };
  // extent of the footprint along the unit vector (nx, ny), and along the axes (for the mask's grid)
  auto extent = [&](const Foot& f, float nx, float ny) {
    if (!f.zono) return f.ball;
    const float e = fabsf(fmaf(f.g1x, nx, f.g1y * ny)) + fabsf(fmaf(f.g2x, nx, f.g2y * ny)) +
                    fabsf(fmaf(f.gxx, nx, f.gxy * ny)) + fabsf(fmaf(f.gyx, nx, f.gyy * ny));
    return fminf(f.ball, fmaf(a.geo_margin, e, f.slack));
  };
  for (int e = 0; e < n_ev; e++) {
    if (__ballot(!culled && !keep) == 0ull) break;      // every box of the wave is decided
    const unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + e);
    const unsigned kind = se >> 16;
    // the interface once (its geometry is the same in every wavelength group), the index terms of the three wavelengths
    const LfProgRow wr = load_prec(recs_of[0], se & 0xffffu);
    float cn22_of[3], rn2_of[3], delta_of[3];
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const LfProgRow x = w == 0 ? wr : load_prec(recs_of[w], se & 0xffffu);
      const int j = j_of[w];
      cn22_of[w] = j == 0 ? x.cn22[0] : j == 1 ? x.cn22[1] : x.cn22[2];
      rn2_of[w] = j == 0 ? x.rn2[0] : j == 1 ? x.rn2[1] : x.rn2[2];
      delta_of[w] = j == 0 ? x.delta[0] : j == 1 ? x.delta[1] : x.delta[2];
    }
    unsigned hit = 0u, okm = 0u;
    const unsigned live_before = live;
    // A ray goes on iff it meets the sphere (disc >= 0) and is not totally reflected ((n' cos t')^2 = disc + delta >= 0):
    // iff p = disc + min(0, delta) >= 0 -- ONE smooth scalar for both ways of ending (a mirror: p = disc)
    float pv[kCullSamples];
    // the total-reflection margins (virtual_event) of the samples that reach the interface: their range over the
    // box, and the one nearest to going on among those that end here by total reflection
    float t_max = -2.0f, t_min = 2.0f, t_lost = -2.0f;
#pragma unroll
    for (int t = 0; t < kCullSamples; t++) {
      const int w = t == 13 ? 1 : t == 14 ? 2 : 0;
      bool ok;
      if (kind & LF_EV_STOP) {
        const float tt = -(r[t].hz + wr.dzv) * lf_rcp(r[t].dz);
        const float hx = fmaf(tt, r[t].dx, r[t].px), hy = fmaf(tt, r[t].dy, r[t].py);
        r[t].px = hx; r[t].py = hy; r[t].hz = 0.0f;
        ok = hx == hx && hy == hy;
        if (ok) hit |= 1u << t;
        pv[t] = 1.0f;
      } else {
        float disc, tir;
        virtual_event(r[t], wr, cn22_of[w], rn2_of[w], delta_of[w], (kind & LF_EV_REFLECT) != 0, (kind & LF_EV_FLAT) != 0, disc, tir);
        pv[t] = (kind & LF_EV_REFLECT) ? disc : disc + fminf(0.0f, delta_of[w]);
        const bool reaches = disc >= 0.0f;
        if (reaches) hit |= 1u << t;                // (a totally reflected ray did reach the interface)
        ok = reaches && tir >= 0.0f;
        if (reaches && ((live >> t) & 1u)) {
          t_max = fmaxf(t_max, tir); t_min = fminf(t_min, tir);
          if (tir < 0.0f) { t_lost = fmaxf(t_lost, tir); tir_partial = true; }
        }
      }
      if (ok) okm |= 1u << t;
    }
    hit &= live;
    // "Every ray of the box ends here".  Round 5 first dropped such a box when even the sample nearest to going on was
    // further from it than half the range of the margins over the box (a rule fitted to the double Gauss: pair (3, 7),
    // profiles/r05_march_variants.txt) -- a draw of 6000 random frames with a second design family (a Cooke triplet,
    // steeper surfaces) found 14 frames where a sliver between the samples went on.  Now (strict_lost, the default):
    // the LARGEST value p can take over the box, bounded from its 15 samples like a footprint -- centre + the four
    // central-difference generators (x the footprints' inflation) + the second order of the four axes summed + twice
    // what the corners deviate by beyond that + how far the ends of the spectrum move the centre -- must stay below
    // zero; and only a box that had lost no sample before is bounded by its samples at all.
    bool firmly_lost = t_lost < -1.5f || t_lost < -(fmaf(a.lost_rel, t_max - t_min, a.lost_abs));
    if (a.strict_lost) {
      auto upper = [&](const float* v) {
        const float ga = 0.5f * (v[5] - v[3]), gb = 0.5f * (v[7] - v[1]), gx = 0.5f * (v[9] - v[10]), gy = 0.5f * (v[11] - v[12]);
        const float da = fabsf(0.5f * (v[5] + v[3]) - v[4]), db = fabsf(0.5f * (v[7] + v[1]) - v[4]);
        const float dx = fabsf(0.5f * (v[9] + v[10]) - v[4]), dy = fabsf(0.5f * (v[11] + v[12]) - v[4]);
        const float second = (da + db) + (dx + dy);
        float cross = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; t += 2) {
          if (t == 4) continue;
          const float at = (float)(t % 3 - 1), bt = (float)(t / 3 - 1);
          cross = fmaxf(cross, fabsf((v[t] - v[4]) - fmaf(at, ga + gx, bt * (gb + gy))) - second);
        }
        const float disp = fmaxf(fabsf(v[13] - v[4]), fabsf(v[14] - v[4]));
        return v[4] + fmaf(a.geo_margin, (fabsf(ga) + fabsf(gb)) + (fabsf(gx) + fabsf(gy)), second + fmaf(2.0f, fmaxf(cross, 0.0f), a.margin * disp));
      };
      // (only a box whose samples all end at THIS event asks)
      firmly_lost = live_before == kAll && (okm & kAll) == 0u && !(kind & LF_EV_STOP) && upper(pv) < -1.0e-4f;
    }
    const Foot f = footprint(false, hit, 1e-3f);
    live &= okm;
    if (!culled && !keep) {
      if (hit == 0u) {                                       // no sample reaches the interface
        if (firmly_lost && !(a.disable & 8)) { culled = true; why = 7; } else { keep = true; why = 2; }
      }
      else {
        const float cx = f.cx, cy = f.cy;
        const float cr = lf_sqrt(fmaf(cx, cx, cy * cy));
        const float icr = cr > 0.0f ? lf_rcp(cr) : 0.0f;
        const bool bounded = a.strict == 0 || f.zono || (a.strict == 2 && !tir_partial);   // (STRICT: a box that lost samples is not bounded by the ones left)
        if (bounded && cr - extent(f, cx * icr, cy * icr) > lf_sqrt(wr.h2) && !(a.disable & 1)) { culled = true; why = 4; }   // wholly outside the clear aperture
        else if (bounded && (kind & LF_EV_STOP)) {
          // ... or on closed cells of the mask: texel coordinate = (h / stop_h + 1) / 2 of the mask's width
          const float s = 0.5f * (float)kCullOcc;
          const float radx = extent(f, 1.0f, 0.0f), rady = extent(f, 0.0f, 1.0f);
          const int ix0 = max(0, (int)floorf(fmaf(cx - radx, a.inv_stop_h, 1.0f) * s));
          const int ix1 = min(kCullOcc - 1, (int)floorf(fmaf(cx + radx, a.inv_stop_h, 1.0f) * s));
          const int iy0 = max(0, (int)floorf(fmaf(cy - rady, a.inv_stop_h, 1.0f) * s));
          const int iy1 = min(kCullOcc - 1, (int)floorf(fmaf(cy + rady, a.inv_stop_h, 1.0f) * s));
          bool open = false;
          if (ix0 <= ix1) {
            const unsigned span = (ix1 - ix0 >= 31 ? 0xffffffffu : ((2u << (ix1 - ix0)) - 1u)) << ix0;
            for (int iy = iy0; iy <= iy1; iy++) open = open || (a.occ[iy] & span) != 0u;
          }
          if (!open && !(a.disable & 2)) { culled = true; why = 5; }
        }
        if (!culled) {
          if (live == 0u && firmly_lost && !(a.disable & 8)) { culled = true; why = 7; }   // every sample ends here, by a margin
          else if (__popc(live & 0x1ffu) < 3) { keep = true; why = 2; }     // too little left to bound anything
          else if (live != kAll) partial = true;
        }
      }
    }
  }
  if (!culled && !keep) {
    // the path is complete: where can the box point?  (K is the unit direction in air again)
    const Foot f = footprint(true, live, 2e-5f);
    const float ex = a.sx - f.cx, ey = a.sy - f.cy;
    const float dist = lf_sqrt(fmaf(ex, ex, ey * ey));
    const float id = dist > 0.0f ? lf_rcp(dist) : 0.0f;
    if (partial && (a.keep_partial || a.strict == 1 || (a.strict == 2 && tir_partial))) { keep = true; why = 1; }
    else if (dist - a.lobe_k * extent(f, ex * id, ey * id) > a.rho && !(a.disable & 4)) { culled = true; why = 6; }
    else { keep = true; why = partial ? 1 : 3; }
  }
  if (stats && valid && why) atomicAdd(&stats[why], 1ull);
  // (measurements only, UNSAFE -- what the boxes nothing bounds cost the march: disable bit 4 drops the boxes kept with too few
  // samples left on the last level, bit 5 those kept because they had lost samples)
  if (a.last && (((a.disable & 16) && why == 2) || ((a.disable & 32) && why == 1))) keep = false;
  const bool enabled = valid && keep;
  if (a.last) {
    if (valid && enabled) {
      unsigned long long* row = table + lf_cull_row_of_block(blk, a.share_n, a.share_nb) * (size_t)(a.P * a.P + 1);
      const unsigned long long bit = 1ull << q;
      atomicOr(&row[cell], bit);
      atomicOr(&row[a.P * a.P], bit);
    }
    // how many (block, cell, path) combinations the march will start: one add per wave
    const lanemask em = __ballot(valid && enabled);
    if (em != 0ull && lane == (int)__builtin_ctzll(em)) atomicAdd(&next_counts[q], (unsigned)__popcll(em));
  } else {
    // the four children (cells of 2P) of every box kept, appended to the path's next list: one atomic per wave
    const lanemask em = __ballot(valid && enabled);
    if (em != 0ull) {
      unsigned base = 0u;
      if (lane == (int)__builtin_ctzll(em)) base = atomicAdd(&next_counts[q], 4u * (unsigned)__popcll(em));
      base = __shfl(base, (int)__builtin_ctzll(em));
      if (valid && enabled) {
        const unsigned at = base + 4u * (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        const unsigned P2 = 2u * (unsigned)a.P;
        unsigned* out = next + (size_t)q * a.list_stride;
        if (at + 3u < a.list_stride) {
#pragma unroll
          for (int c = 0; c < 4; c++)
            out[at + c] = (unsigned)blk * (P2 * P2) + (unsigned)(2 * cj + (c >> 1)) * P2 + (unsigned)(2 * ci + (c & 1));
        }
      }
    }
  }
}

// ---- the pre-pass as it ships ---------------------------------------------------------------------------
// k_cull_level_general above evaluates whatever rules its arguments carry (a test's: lf_test_knob); the rules that SHIP
// are one set, and most of what the general kernel computes they never look at.  Under them
//   * a footprint is only ever taken over ALL 15 samples: a box that lost a sample is bounded by nothing (it is kept, or
//     dropped by the pass-scalar bound of the event at which all its samples end), so there is no ball around the samples
//     left, no reference sample other than the centre, no use-masks;
//   * on a glass interface the footprint can drop a box only if the centre sample already lies outside the clear
//     aperture (the test is  |c| - extent > h,  extent >= 0) -- a footprint is built only for the waves in which some
//     lane's centre does (round 5 built one after every event of every box: 233 lane-instructions per marched
//     ray-event against the march's 40); at the stop the mask's grid needs it for every box that got there whole;
//   * the total-reflection margin is only a sign: no reciprocal.
// Same expressions in the same order (the build is -ffp-contract=off): the table is the general kernel's BIT FOR BIT
// (tests/test_gpu_cull.py test_the_shipped_kernel_is_the_general_one).
__device__ __forceinline__ void cull_event(Ray& r, const LfProgRow& w, float cn22, float rn2, float delta, bool reflect, bool flat,
                                           float& disc_out, bool& goes_on) {
  const float oz = r.hz + w.dzv;
  const float od = fmaf(r.px, r.dx, fmaf(r.py, r.dy, oz * r.dz));
  const float oo = fmaf(oz, oz, fmaf(r.px, r.px, r.py * r.py));
  const float Fh = fmaf(w.ch, oo, -oz);
  const float G = fmaf(-w.curv, od, r.dz);
  const float disc = fmaf(G, G, -(cn22 * Fh));
  disc_out = disc;
  const float sq = lf_sqrt(disc);
  const float t = flat ? (Fh + Fh) * lf_rcp(fmaf(w.sgn, sq, G)) : fmaf(-w.sgn, sq, G) * rn2;
  const float hx = fmaf(t, r.dx, r.px), hy = fmaf(t, r.dy, r.py), hz = fmaf(t, r.dz, oz);
  if (reflect) {
    goes_on = true;
    const float m = sq * (w.c2 * w.sgn);
    r.dx = fmaf(m, hx, r.dx); r.dy = fmaf(m, hy, r.dy); r.dz = fmaf(m, hz, fmaf(-2.0f * w.sgn, sq, r.dz));
  } else {
    const float k2 = disc + delta;
    goes_on = k2 >= 0.0f;
    const float gs = lf_sqrt(k2) - sq, gcs = gs * w.sc;
    r.dx = fmaf(-gcs, hx, r.dx); r.dy = fmaf(-gcs, hy, r.dy); r.dz = fmaf(-gcs, hz, fmaf(w.sgn, gs, r.dz));
  }
  r.px = hx; r.py = hy; r.hz = hz;
}

// the shipped rules' constants (lf_ctx::CullRules' defaults: what k_cull_level_general is given when no test interferes)
constexpr float kShipLobeK = 1.2f;

// the footprint of a box whose 15 samples are all in use, in a plane (positions on an interface / directions at the exit):
// centre, generators, the slack that sums the second order of the four axes + twice the corners' cross terms, and the ball
struct ShipFoot { float cx, cy, g1x, g1y, g2x, g2y, gxx, gxy, gyx, gyy, slack, ball; };
template <bool DIRS>
__device__ __forceinline__ ShipFoot ship_footprint(const Ray (&r)[kCullSamples], float margin, float eps) {
  auto vx = [&](int t) { return DIRS ? r[t].dx : r[t].px; };
  auto vy = [&](int t) { return DIRS ? r[t].dy : r[t].py; };
  ShipFoot f;
  f.cx = vx(4); f.cy = vy(4);
  f.g1x = 0.5f * (vx(5) - vx(3)); f.g1y = 0.5f * (vy(5) - vy(3));
  f.g2x = 0.5f * (vx(7) - vx(1)); f.g2y = 0.5f * (vy(7) - vy(1));
  f.gxx = 0.5f * (vx(9) - vx(10)); f.gxy = 0.5f * (vy(9) - vy(10));
  f.gyx = 0.5f * (vx(11) - vx(12)); f.gyy = 0.5f * (vy(11) - vy(12));
  float ru2 = 0.0f, rx2 = 0.0f, ry2 = 0.0f, rl2 = 0.0f;
#pragma unroll
  for (int t = 0; t < kCullSamples; t++) {
    const float ex = vx(t) - f.cx, ey = vy(t) - f.cy;
    const float d2 = fmaf(ex, ex, ey * ey);
    if (t < 9) ru2 = fmaxf(ru2, d2);
    else if (t < 11) rx2 = fmaxf(rx2, d2);
    else if (t < 13) ry2 = fmaxf(ry2, d2);
    else rl2 = fmaxf(rl2, d2);
  }
  const float ru = lf_sqrt(ru2), rl = lf_sqrt(rl2);
  const float rx = lf_sqrt(rx2), ry = lf_sqrt(ry2);
  f.ball = fmaf(margin, ((ru + rx) + ry) + rl, eps);
  const float dax = 0.5f * (vx(5) + vx(3)) - vx(4), day = 0.5f * (vy(5) + vy(3)) - vy(4);
  const float dbx = 0.5f * (vx(7) + vx(1)) - vx(4), dby = 0.5f * (vy(7) + vy(1)) - vy(4);
  const float dxx = 0.5f * (vx(9) + vx(10)) - vx(4), dxy = 0.5f * (vy(9) + vy(10)) - vy(4);
  const float dyx = 0.5f * (vx(11) + vx(12)) - vx(4), dyy = 0.5f * (vy(11) + vy(12)) - vy(4);
  const float sx = (dax + dbx) + (dxx + dyx), sy = (day + dby) + (dxy + dyy);
  float c2 = 0.0f;
#pragma unroll
  for (int t = 0; t < 9; t += 2) {
    if (t == 4) continue;
    const float at = (float)(t % 3 - 1), bt = (float)(t / 3 - 1);
    const float mx = (vx(t) - vx(4)) - fmaf(at, f.g1x + f.gxx, bt * (f.g2x + f.gyx)) - sx;
    const float my = (vy(t) - vy(4)) - fmaf(at, f.g1y + f.gxy, bt * (f.g2y + f.gyy)) - sy;
    c2 = fmaxf(c2, fmaf(mx, mx, my * my));
  }
  const float sum2 = lf_sqrt(fmaf(dax, dax, day * day)) + lf_sqrt(fmaf(dbx, dbx, dby * dby)) +
                     lf_sqrt(fmaf(dxx, dxx, dxy * dxy)) + lf_sqrt(fmaf(dyx, dyx, dyy * dyy));
  f.slack = sum2 + fmaf(2.0f, lf_sqrt(c2), fmaf(margin, rl, eps));
  return f;
}
__device__ __forceinline__ float ship_extent(const ShipFoot& f, float geo_margin, float nx, float ny) {
  const float e = fabsf(fmaf(f.g1x, nx, f.g1y * ny)) + fabsf(fmaf(f.g2x, nx, f.g2y * ny)) +
                  fabsf(fmaf(f.gxx, nx, f.gxy * ny)) + fabsf(fmaf(f.gyx, nx, f.gyy * ny));
  return fminf(f.ball, fmaf(geo_margin, e, f.slack));
}

__global__ __launch_bounds__(LF_CULL_WG, LF_CULL_WAVES) void k_cull_level(const LfLensDev* __restrict__ lens,
                                                    const LfPairsDev* __restrict__ pairs,
                                                    const int* __restrict__ seq_table,
                                                    const LfProgRow* __restrict__ rec_table, CullLevelArgs a,
                                                    const unsigned* __restrict__ items,
                                                    const unsigned* __restrict__ counts, unsigned items_stride,
                                                    unsigned* __restrict__ next, unsigned* __restrict__ next_counts,
                                                    unsigned long long* __restrict__ table,
                                                    unsigned long long* __restrict__ stats) {
  const int q = blockIdx.y;
  const unsigned PP = (unsigned)(a.P * a.P);
  const unsigned n_blk = (unsigned)(a.blocks_x * a.blocks_y);
  const unsigned n_mine = (n_blk + (unsigned)a.share_n - 1u - (unsigned)a.share_rank) / (unsigned)a.share_n;
  const unsigned n_items = items ? min(counts[q], items_stride) : n_mine * PP;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i & ~63u) >= n_items) return;
  const bool valid = i < n_items;
  const unsigned item = valid ? (items ? items[(size_t)q * items_stride + i]
                                       : ((unsigned)a.share_rank + (unsigned)a.share_n * (i / PP)) * PP + i % PP) : 0u;
  const int blk = (int)(item / PP), cell = (int)(item % PP);
  const int ci = cell % a.P, cj = cell / a.P;
  const int bx = blk % a.blocks_x, by = blk / a.blocks_x;

  const float px0 = (float)(bx << a.blk_log2), px1 = fminf((float)a.W, (float)((bx + 1) << a.blk_log2));
  const float py0 = (float)(by << a.blk_log2), py1 = fminf((float)a.H, (float)((by + 1) << a.blk_log2));
  const float Xc = -((0.5f * (px0 + px1)) - a.half_w) * a.pitch, Yc = -((0.5f * (py0 + py1)) - a.half_h) * a.pitch;
  const float hX = 0.5f * (px1 - px0) * a.pitch, hY = 0.5f * (py1 - py0) * a.pitch;
  const float invP = 1.0f / (float)a.P;

  const int n_ev = pairs->ev_cnt[q];
  const int* const seq = seq_table + pairs->ev_off[q];
  const int lane = (int)(threadIdx.x & 63u);
  constexpr unsigned kAll = (1u << kCullSamples) - 1u;
  const LfProgRow* recs_of[3];
  int j_of[3];
  float ns_of[3];
#pragma unroll
  for (int w = 0; w < 3; w++) {
    const int l = a.lam[w], g = l / a.march_k;
    j_of[w] = l - g * a.march_k;
    recs_of[w] = rec_table + (size_t)g * (size_t)a.prog_recs;
    ns_of[w] = lens->n_start[l];
  }
  // the box's 15 rays (the sample layout of k_cull_level_general)
  Ray r[kCullSamples];
#pragma unroll
  for (int t = 0; t < kCullSamples; t++) {
    float X = Xc, Y = Yc, fu = 0.5f, fv = 0.5f;
    if (t < 9) {
      fu = 0.5f * (float)(t % 3); fv = 0.5f * (float)(t / 3);
      if ((t % 3) != 1 && (t / 3) != 1) { X = Xc + (float)(t % 3 - 1) * hX; Y = Yc + (float)(t / 3 - 1) * hY; }
    }
    else if (t == 9) X = Xc + hX;
    else if (t == 10) X = Xc - hX;
    else if (t == 11) Y = Yc + hY;
    else if (t == 12) Y = Yc - hY;
    const float ns = t == 13 ? ns_of[1] : t == 14 ? ns_of[2] : ns_of[0];
    const float ua = ((float)ci + fu) * invP, ub = ((float)cj + fv) * invP;
    const StartRay s0 = aim_at_pupil(X, Y, fmaf(2.0f, ua, -1.0f), fmaf(2.0f, ub, -1.0f), a.pupil_h, a.vz, a.geom_norm);
    r[t] = Ray{X, Y, 0.0f, 0.0f, s0.dx * ns, s0.dy * ns, s0.dz * ns, 0.0f, 0.0f};
  }
  unsigned live = kAll;
  bool culled = !valid, keep = false;
  int why = 0;
  for (int e = 0; e < n_ev; e++) {
    if (__ballot(!culled && !keep) == 0ull) break;      // every box of the wave is decided
    const unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + e);
    const unsigned kind = se >> 16;
    const LfProgRow wr = load_prec(recs_of[0], se & 0xffffu);
    float cn22_of[3], rn2_of[3], delta_of[3];
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const LfProgRow x = w == 0 ? wr : load_prec(recs_of[w], se & 0xffffu);
      const int j = j_of[w];
      cn22_of[w] = j == 0 ? x.cn22[0] : j == 1 ? x.cn22[1] : x.cn22[2];
      rn2_of[w] = j == 0 ? x.rn2[0] : j == 1 ? x.rn2[1] : x.rn2[2];
      delta_of[w] = j == 0 ? x.delta[0] : j == 1 ? x.delta[1] : x.delta[2];
    }
    const bool stop = (kind & LF_EV_STOP) != 0u;
    unsigned hit = 0u, okm = 0u;
    const unsigned live_before = live;
    float pv[kCullSamples];      // the pass scalar of every sample: p = disc + min(0, n'^2 - n^2) (see k_cull_level_general)
#pragma unroll
    for (int t = 0; t < kCullSamples; t++) {
      const int w = t == 13 ? 1 : t == 14 ? 2 : 0;
      bool ok;
      if (stop) {
        const float tt = -(r[t].hz + wr.dzv) * lf_rcp(r[t].dz);
        const float hx = fmaf(tt, r[t].dx, r[t].px), hy = fmaf(tt, r[t].dy, r[t].py);
        r[t].px = hx; r[t].py = hy; r[t].hz = 0.0f;
        ok = hx == hx && hy == hy;
        if (ok) hit |= 1u << t;
        pv[t] = 1.0f;
      } else {
        float disc;
        bool goes_on;
        cull_event(r[t], wr, cn22_of[w], rn2_of[w], delta_of[w], (kind & LF_EV_REFLECT) != 0, (kind & LF_EV_FLAT) != 0, disc, goes_on);
        pv[t] = (kind & LF_EV_REFLECT) ? disc : disc + fminf(0.0f, delta_of[w]);
        const bool reaches = disc >= 0.0f;
        if (reaches) hit |= 1u << t;
        ok = reaches && goes_on;
      }
      if (ok) okm |= 1u << t;
    }
    hit &= live;
    const bool undecided = !culled && !keep;
    // "every ray of the box ends here": the zonotope bound of the pass scalar must stay below zero, on a box whole until now
    const bool all_end = undecided && live_before == kAll && (okm & kAll) == 0u && !stop;
    bool firmly_lost = false;
    if (__ballot(all_end) != 0ull) {
      const float* v = pv;
      const float ga = 0.5f * (v[5] - v[3]), gb = 0.5f * (v[7] - v[1]), gx = 0.5f * (v[9] - v[10]), gy = 0.5f * (v[11] - v[12]);
      const float da = fabsf(0.5f * (v[5] + v[3]) - v[4]), db = fabsf(0.5f * (v[7] + v[1]) - v[4]);
      const float dx = fabsf(0.5f * (v[9] + v[10]) - v[4]), dy = fabsf(0.5f * (v[11] + v[12]) - v[4]);
      const float second = (da + db) + (dx + dy);
      float cross = 0.0f;
#pragma unroll
      for (int t = 0; t < 9; t += 2) {
        if (t == 4) continue;
        const float at = (float)(t % 3 - 1), bt = (float)(t / 3 - 1);
        cross = fmaxf(cross, fabsf((v[t] - v[4]) - fmaf(at, ga + gx, bt * (gb + gy))) - second);
      }
      const float disp = fmaxf(fabsf(v[13] - v[4]), fabsf(v[14] - v[4]));
      const float upper = v[4] + fmaf(a.geo_margin, (fabsf(ga) + fabsf(gb)) + (fabsf(gx) + fabsf(gy)), second + fmaf(2.0f, fmaxf(cross, 0.0f), a.margin * disp));
      firmly_lost = all_end && upper < -1.0e-4f;
    }
    live &= okm;
    // a footprint can drop a box only where every sample reached the interface AND (the stop's mask, or the centre sample
    // outside the clear aperture)
    const float cx = r[4].px, cy = r[4].py;
    const float cr = lf_sqrt(fmaf(cx, cx, cy * cy));
    const float h = lf_sqrt(wr.h2);
    const bool whole = undecided && hit == kAll;
    if (__ballot(whole && (stop || cr > h)) != 0ull) {
      const ShipFoot f = ship_footprint<false>(r, a.margin, 1e-3f);
      const float icr = cr > 0.0f ? lf_rcp(cr) : 0.0f;
      if (whole && cr - ship_extent(f, a.geo_margin, cx * icr, cy * icr) > h) { culled = true; why = 4; }
      else if (whole && stop) {
        const float s = 0.5f * (float)kCullOcc;
        const float radx = ship_extent(f, a.geo_margin, 1.0f, 0.0f), rady = ship_extent(f, a.geo_margin, 0.0f, 1.0f);
        const int ix0 = max(0, (int)floorf(fmaf(cx - radx, a.inv_stop_h, 1.0f) * s));
        const int ix1 = min(kCullOcc - 1, (int)floorf(fmaf(cx + radx, a.inv_stop_h, 1.0f) * s));
        const int iy0 = max(0, (int)floorf(fmaf(cy - rady, a.inv_stop_h, 1.0f) * s));
        const int iy1 = min(kCullOcc - 1, (int)floorf(fmaf(cy + rady, a.inv_stop_h, 1.0f) * s));
        bool open = false;
        if (ix0 <= ix1) {
          const unsigned span = (ix1 - ix0 >= 31 ? 0xffffffffu : ((2u << (ix1 - ix0)) - 1u)) << ix0;
          for (int iy = iy0; iy <= iy1; iy++) open = open || (a.occ[iy] & span) != 0u;
        }
        if (!open) { culled = true; why = 5; }
      }
    }
    if (undecided && !culled) {
      if (hit == 0u) {                                       // no sample reaches the interface
        if (firmly_lost) { culled = true; why = 7; } else { keep = true; why = 2; }
      }
      else if (live == 0u && firmly_lost) { culled = true; why = 7; }   // every sample ends here, by a margin
      else if (__popc(live & 0x1ffu) < 3) { keep = true; why = 2; }     // too little left to bound anything
    }
  }
  if (!culled && !keep) {
    // the path is complete.  A box that lost samples is bounded by nothing: kept.  A whole one: where can it point?
    if (live != kAll) { keep = true; why = 1; }
  }
  if (__ballot(!culled && !keep) != 0ull) {
    const ShipFoot f = ship_footprint<true>(r, a.margin, 2e-5f);
    const float ex = a.sx - f.cx, ey = a.sy - f.cy;
    const float dist = lf_sqrt(fmaf(ex, ex, ey * ey));
    const float id = dist > 0.0f ? lf_rcp(dist) : 0.0f;
    if (!culled && !keep) {
      if (dist - kShipLobeK * ship_extent(f, a.geo_margin, ex * id, ey * id) > a.rho) { culled = true; why = 6; }
      else { keep = true; why = 3; }
    }
  }
  if (stats && valid && why) atomicAdd(&stats[why], 1ull);
  const bool enabled = valid && keep;
  if (a.last) {
    if (valid && enabled) {
      unsigned long long* row = table + lf_cull_row_of_block(blk, a.share_n, a.share_nb) * (size_t)(a.P * a.P + 1);
      const unsigned long long bit = 1ull << q;
      atomicOr(&row[cell], bit);
    }
    const lanemask em = __ballot(valid && enabled);
    // the row's summary word (its last): one atomic per wave and block, not per box (the lanes of a wave are boxes of ONE path)
    for (lanemask todo = em; todo != 0ull;) {
      const int first = (int)__builtin_ctzll(todo);
      const int b0 = __shfl(blk, first);
      if (lane == first) atomicOr(&table[lf_cull_row_of_block(b0, a.share_n, a.share_nb) * (size_t)(a.P * a.P + 1) + (size_t)(a.P * a.P)], 1ull << q);
      todo &= ~__ballot(blk == b0);
    }
    if (em != 0ull && lane == (int)__builtin_ctzll(em)) atomicAdd(&next_counts[q], (unsigned)__popcll(em));
  } else {
    const lanemask em = __ballot(valid && enabled);
    if (em != 0ull) {
      unsigned base = 0u;
      if (lane == (int)__builtin_ctzll(em)) base = atomicAdd(&next_counts[q], 4u * (unsigned)__popcll(em));
      base = __shfl(base, (int)__builtin_ctzll(em));
      if (valid && enabled) {
        const unsigned at = base + 4u * (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        const unsigned P2 = 2u * (unsigned)a.P;
        unsigned* out = next + (size_t)q * a.list_stride;
        if (at + 3u < a.list_stride) {
#pragma unroll
          for (int c = 0; c < 4; c++)
            out[at + c] = (unsigned)blk * (P2 * P2) + (unsigned)(2 * cj + (c >> 1)) * P2 + (unsigned)(2 * ci + (c & 1));
        }
      }
    }
  }
}

// ---- the march of the enabled paths ---------------------------------------------------------------------
constexpr int kWgWaves = 8;
constexpr int kListMax = 4096;   // samples of one tile's workgroup (spp / sgroups) listed at a time

// what a wave tallies while it marches (slots of lf_counters / lf_get_march_stats)
struct PathTally {
  unsigned long long n_rays = 0, events = 0, n_clip = 0, n_vign = 0, n_tir = 0, n_scene = 0, n_rm_lane = 0, n_rm_rows = 0;
  unsigned long long executed = 0;   // rows really executed (a row of the common leg once): march_started_set; march_started_path: = events
  unsigned n_light = 0;   // per lane
};

// One STARTED path q of one sensor sample per lane (start ray X, Y, s0; lanes in start_mask), every wavelength group:
// the events of the path's own sequence, K wavelengths together, tallies, the lobe test, the weighted second march
// (W1 = false) and the fixed-point add into the tile's LDS sums at pixel slot acc_slot.  Shared by the two culled
// kernels below: lane = pixel (k_march_cull) and lane = one compacted (pixel, sample) item (k_march_items).
template <int K, bool W1>
__device__ __forceinline__ void march_started_path(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
                                                   const int* __restrict__ seq_table, const LfProgRow* __restrict__ rec_table,
                                                   const LfWeightRow* __restrict__ wrec_table, const float* __restrict__ mask,
                                                   const MarchArgs& a, int q, lanemask start_mask, float X, float Y,
                                                   const StartRay& s0, int lane, int acc_slot,
                                                   unsigned long long* __restrict__ s_acc, PathTally& T) {
  const int n_lambda = lens->n_lambda, prog_recs = pairs->prog_recs;
  const int n_groups = (n_lambda + K - 1) / K;
  const float inv_stop_h = a.inv_stop_h, lobe_thr = a.lobe_thr;
  const float sx = lens->sun_dir[0], sy = lens->sun_dir[1], sz = lens->sun_dir[2];
  const float inv_1mc = lens->sun_inv_one_minus_cos, sun_ss = lens->sun_ss;
  const int n_ev = pairs->ev_cnt[q];
  const int* const seq = seq_table + pairs->ev_off[q];
  for (int g = 0; g < n_groups; g++) {
    const LfProgRow* const recs = rec_table + (size_t)g * (size_t)prog_recs;
    const LfWeightRow* const wrecs = wrec_table + (size_t)g * (size_t)prog_recs;
    Ray r[K];
    lanemask alive[K];
    unsigned nlive = 0u;
#pragma unroll
    for (int j = 0; j < K; j++) {
      r[j] = Ray{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx, s0.dy, s0.dz, s0.w0, 1.0f};
      const float ns = lens->n_start[min(g * K + j, n_lambda - 1)];
      r[j].dx *= ns; r[j].dy *= ns; r[j].dz *= ns;
      alive[j] = (g * K + j < n_lambda) ? start_mask : 0ull;
      nlive += (unsigned)__popcll(alive[j]);
    }
    T.n_rays += nlive;
    unsigned ev32 = 0u;
    unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq);
    int e = 0;
    if (n_ev > 0 && nlive != 0u) for (;;) {     // (the row's end tests ONE thing: a wave without rays leaves through the row counter)
      const unsigned cur = se;
      se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + e + 1);     // (the table ends on a spare dword: pack_program)
      const LfProgRow wr = load_prec(recs, cur & 0xffffu);
      LfWeightRow ww;
      if (W1) ww = load_wrec(wrecs, cur & 0xffffu);
      else { for (int j = 0; j < 3; j++) { ww.fs[j] = 1.0f; ww.fo[j] = 1.0f; ww.fi[j] = 1.0f; } }
      const unsigned kind = cur >> 16;
      lanemask okv[K], gv[K], died = 0ull;
      if (kind & LF_EV_STOP) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
          okv[j] = stop_event<W1>(r[j], wr.dzv, wr.h2, inv_stop_h, mask, a.mw, a.mh);
          gv[j] = okv[j];
          died |= alive[j] & ~okv[j];
        }
        if (__builtin_expect(died != 0ull, 0)) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            const unsigned nd = (unsigned)__popcll(alive[j] & ~okv[j]);
            T.n_clip += nd; nlive -= nd; alive[j] &= okv[j];
          }
          if (nlive == 0u) e = n_ev;
        }
      } else {
        if (kind == 0u) {          // refraction at a curved interface: the common row, straight-line
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
            okv[j] = surface_event<W1>(r[j], wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, wr.cn22[j], wr.rn2[j],
                                       wr.delta[j], wr.h2, false, false, wr.sgn, gv[j], ww.fs[j], ww.fo[j], ww.fi[j]);
            died |= alive[j] & ~okv[j];
          }
        } else if (kind == (unsigned)LF_EV_REFLECT) {      // a curved mirror: two rows of every pair
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
            okv[j] = surface_event<W1>(r[j], wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, wr.cn22[j], wr.rn2[j],
                                       wr.delta[j], wr.h2, true, false, wr.sgn, gv[j], ww.fs[j], ww.fo[j], ww.fi[j]);
            died |= alive[j] & ~okv[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
            okv[j] = surface_event<W1>(r[j], wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, wr.cn22[j], wr.rn2[j],
                                       wr.delta[j], wr.h2, (kind & LF_EV_REFLECT) != 0, (kind & LF_EV_FLAT) != 0,
                                       wr.sgn, gv[j], ww.fs[j], ww.fo[j], ww.fi[j]);
            died |= alive[j] & ~okv[j];
          }
        }
        if (__builtin_expect(died != 0ull, 0)) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            T.n_vign += (unsigned)__popcll(alive[j] & ~gv[j]);
            T.n_tir += (unsigned)__popcll(alive[j] & gv[j] & ~okv[j]);
            nlive -= (unsigned)__popcll(alive[j] & ~okv[j]);
            alive[j] &= okv[j];
          }
          if (nlive == 0u) e = n_ev;
        }
      }
      ev32 += nlive;       // events completed: one per ray still alive after the row
      if (++e >= n_ev) break;
    }
    T.events += ev32;
    T.executed += ev32;
    if (nlive == 0u) continue;
    // ---- the path is complete for nlive rays --------------------------------------------------
    T.n_scene += nlive;
    lanemask lit[K], lit_any = 0ull;
#pragma unroll
    for (int j = 0; j < K; j++) {
      const float cg = fmaf(r[j].dx, sx, fmaf(r[j].dy, sy, r[j].dz * sz));
      lit[j] = alive[j] & __ballot(cg > lobe_thr);
      lit_any |= lit[j];
    }
    if (lit_any == 0ull) continue;
    for (int j = 0; j < K; j++) {        // not unrolled (W1 = false): one copy of the weighted march
      lanemask lj = lit[0];
#pragma unroll
      for (int jj = 1; jj < K; jj++) lj = (j == jj) ? lit[jj] : lj;
      if (lj == 0ull) continue;
      const int l = g * K + j;
      Ray rw = j == 0 ? r[0] : j == 1 ? r[K > 1 ? 1 : 0] : r[K > 2 ? 2 : 0];
      if (!W1) {
        // the path again, alone and with its weight: the same arithmetic on the ray, so the same ray bit for bit
        rw = Ray{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx, s0.dy, s0.dz, s0.w0, 1.0f};
        { const float ns = lens->n_start[l]; rw.dx *= ns; rw.dy *= ns; rw.dz *= ns; }
        T.n_rm_lane += (unsigned long long)((unsigned)n_ev * (unsigned)__popcll(lj));
        T.n_rm_rows += (unsigned)n_ev;
        const int* w = seq;
        for (int left = n_ev; left > 0; --left, ++w) {
          const unsigned se2 = (unsigned)*(const int __attribute__((address_space(4)))*)(w);
          const LfProgRow wr = load_prec(recs, se2 & 0xffffu);
          const LfWeightRow ww = load_wrec(wrecs, se2 & 0xffffu);
          const unsigned wfl = se2 >> 16;
          const float w_cn22 = j == 0 ? wr.cn22[0] : j == 1 ? wr.cn22[1] : wr.cn22[2];
          const float w_rn2 = j == 0 ? wr.rn2[0] : j == 1 ? wr.rn2[1] : wr.rn2[2];
          const float w_delta = j == 0 ? wr.delta[0] : j == 1 ? wr.delta[1] : wr.delta[2];
          const float w_fs = j == 0 ? ww.fs[0] : j == 1 ? ww.fs[1] : ww.fs[2];
          const float w_fo = j == 0 ? ww.fo[0] : j == 1 ? ww.fo[1] : ww.fo[2];
          const float w_fi = j == 0 ? ww.fi[0] : j == 1 ? ww.fi[1] : ww.fi[2];
          if (wfl & LF_EV_STOP) {
            (void)stop_event<true>(rw, wr.dzv, wr.h2, inv_stop_h, mask, a.mw, a.mh);
          } else {
            lanemask geom_ok;
            (void)surface_event<true>(rw, wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, w_cn22, w_rn2, w_delta, wr.h2,
                                      (wfl & LF_EV_REFLECT) != 0, (wfl & LF_EV_FLAT) != 0, wr.sgn, geom_ok, w_fs, w_fo, w_fi);
          }
        }
      }
      const float qq = lobe_q(rw.dx, rw.dy, rw.dz, sx, sy, sz, sun_ss, inv_1mc);
      const float om = 1.0f - qq;
      float contrib = __fdiv_rn(rw.wn, rw.wd) * (om * om);
      contrib = (((lj >> lane) & 1ull) != 0ull && qq < 1.0f && contrib > 0.0f) ? contrib : 0.0f;
      T.n_light += contrib > 0.0f ? 1u : 0u;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v = contrib * (lens->sun_radiance[c] * lens->lambda_rgb[l][c]);
        const unsigned long long fx = (unsigned long long)(v * 68719476736.0f);
        if (fx) atomicAdd(&s_acc[acc_slot * 3 + c], fx);
      }
    }
  }
}

// ---- the started paths of ONE sample, their common leg marched once (round 6) ----------------------------------------
// Every path starts with the same leg from the sensor towards the scene -- the primary path's events -- and leaves it at
// its first mirror i after N - 1 - i of them (the primary path never does).  march_started_path marches that leg again for
// every started path; the paths a sample starts, though, are few and MOST of what they execute is this leg (a pair (i, j)
// of the 11-interface lens: 10 - i of its 11 + 2 (j - i) events, and the rays that end early end in it).  Here the wave
// keeps ONE running state of the common leg: the started paths are taken in the order of their first mirrors, rear ones
// first (= descending path index: the selection lists pairs by (i, j) ascending with the primary path in front; the host
// checks it, LfCullArgs::prefix_ok), the leg is extended to where the next path leaves it, the path's own events run on a
// copy.  The arithmetic on a ray is the same events in the same order: pixels and counters are those of every path
// marched alone -- a row of the common leg counts once for every started path that shares it (the paths not yet done) --
// while the rows EXECUTED fall by the shared part (the device's executed-events counter, slot 7).
// Geometry first (W1 = false): a lane that ends inside the lobe marches its path again, alone, with the weight, as before.
// one surface row for the K wavelengths of a group, its kind a compile-time constant -> the lanes that ended on it
template <int K, bool REFLECT, bool FLAT>
__device__ __forceinline__ lanemask surface_rows(Ray (&r)[K], const lanemask (&alive)[K], const LfProgRow& wr, lanemask (&okv)[K],
                                                 lanemask (&gv)[K]) {
  lanemask died = 0ull;
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
    okv[j] = surface_event<false>(r[j], wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, wr.cn22[j], wr.rn2[j], wr.delta[j], wr.h2,
                                  REFLECT, FLAT, wr.sgn, gv[j]);
    died |= alive[j] & ~okv[j];
  }
  return died;
}

template <int K>
__device__ __forceinline__ void march_started_set(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
                                                  const int* __restrict__ seq_table, const LfProgRow* __restrict__ rec_table,
                                                  const LfWeightRow* __restrict__ wrec_table, const float* __restrict__ mask,
                                                  const MarchArgs& a, unsigned long long todo, lanemask active_mask, float X, float Y,
                                                  const StartRay& s0, int lane, unsigned long long* __restrict__ s_acc,
                                                  const int2* __restrict__ s_meta, PathTally& T) {
  const int n_lambda = lens->n_lambda, prog_recs = pairs->prog_recs;
  const int n_groups = (n_lambda + K - 1) / K;
  const float inv_stop_h = a.inv_stop_h, lobe_thr = a.lobe_thr;
  const float sx = lens->sun_dir[0], sy = lens->sun_dir[1], sz = lens->sun_dir[2];
  const float inv_1mc = lens->sun_inv_one_minus_cos, sun_ss = lens->sun_ss;
  const unsigned n_started = (unsigned)__popcll(todo);
  for (int g = 0; g < n_groups; g++) {
    const LfProgRow* const recs = rec_table + (size_t)g * (size_t)prog_recs;
    const LfWeightRow* const wrecs = wrec_table + (size_t)g * (size_t)prog_recs;
    // ONE running state: the common leg while it is common, then the path that left it -- whose start (the leg's state at
    // the fork) is parked in p and taken back when the path is done
    Ray r[K], p[K];
    lanemask alive[K], palive[K];
    unsigned nlive = 0u, pn = 0u;
#pragma unroll
    for (int j = 0; j < K; j++) {
      r[j] = Ray{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx, s0.dy, s0.dz, s0.w0, 1.0f};
      const float ns = lens->n_start[min(g * K + j, n_lambda - 1)];
      r[j].dx *= ns; r[j].dy *= ns; r[j].dz *= ns;
      alive[j] = (g * K + j < n_lambda) ? active_mask : 0ull;
      nlive += (unsigned)__popcll(alive[j]);
      p[j] = r[j]; palive[j] = alive[j];
    }
    T.n_rays += (unsigned long long)nlive * n_started;
    int depth = 0;            // events of the common leg done
    unsigned long long left = todo;
    while (left != 0ull && nlive != 0u) {
      const int q = 63 - __builtin_clzll(left);
      left &= ~(1ull << q);
      // (the path's events and where it leaves the common leg: from the workgroup's LDS copy, not three dependent scalar loads)
      const int2 meta = s_meta[q];
      const int mx = __builtin_amdgcn_readfirstlane(meta.x);       // (q is wave-uniform: so are the path's lengths and rows)
      const int n_ev = mx >> 16, L = mx & 0xffff;
      const int* const seq = seq_table + __builtin_amdgcn_readfirstlane(meta.y);
      unsigned mult = (unsigned)__popcll(left) + 1u;               // this path and those still to come share the leg so far
      bool forked = false;
      unsigned seg = 0u;                 // sum of the rays alive after each row of the current segment (leg / path)
      unsigned long long ev_logical = 0ull;
      unsigned ev_exec = 0u;
      // (the sequence dword one row ahead, unconditionally: the table ends on a spare dword -- pack_program)
      unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + depth);
      int e = depth;
      if (e < n_ev && nlive != 0u) for (;;) {
        if (e == L) {      // the path leaves the common leg: park the leg
#pragma unroll
          for (int j = 0; j < K; j++) { p[j] = r[j]; palive[j] = alive[j]; }
          pn = nlive; forked = true;
          ev_logical = (unsigned long long)seg * mult; ev_exec = seg; seg = 0u; mult = 1u;
        }
        const unsigned cur = se;
        se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + e + 1);
        const LfProgRow wr = load_prec(recs, cur & 0xffffu);
        const unsigned kind = cur >> 16;
        lanemask okv[K], gv[K], died = 0ull;
        // (the scalar unit is what this loop waits for: the row's kind is decided ONCE for its K wavelengths -- curved glass
        // crossed, by far the most frequent, first -- and the event inlined with its flags constant; a wave that lost its
        // last ray leaves through the row counter, so that the row's end tests one thing)
        if (__builtin_expect(kind == 0u, 1)) {
          died = surface_rows<K, false, false>(r, alive, wr, okv, gv);
        } else if (kind & LF_EV_STOP) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
            okv[j] = stop_event<false>(r[j], wr.dzv, wr.h2, inv_stop_h, mask, a.mw, a.mh);
            gv[j] = okv[j];
            died |= alive[j] & ~okv[j];
          }
        } else if (kind & LF_EV_REFLECT) {
          if (kind & LF_EV_FLAT) died = surface_rows<K, true, true>(r, alive, wr, okv, gv);
          else died = surface_rows<K, true, false>(r, alive, wr, okv, gv);
        } else {
          died = surface_rows<K, false, true>(r, alive, wr, okv, gv);
        }
        if (__builtin_expect(died != 0ull, 0)) {
          // (a stop row: gv = okv, every ray it ends is clipped; a surface row: missed / outside the aperture, or totally reflected)
          const bool stop_row = (kind & LF_EV_STOP) != 0;
#pragma unroll
          for (int j = 0; j < K; j++) {
            const unsigned nd = (unsigned)__popcll(alive[j] & ~okv[j]);
            const unsigned nv = (unsigned)__popcll(alive[j] & ~gv[j]);
            T.n_clip += stop_row ? (unsigned long long)nd * mult : 0ull;
            T.n_vign += stop_row ? 0ull : (unsigned long long)nv * mult;
            T.n_tir += stop_row ? 0ull : (unsigned long long)(nd - nv) * mult;
            nlive -= nd;
            alive[j] &= okv[j];
          }
          if (nlive == 0u) e = n_ev;
        }
        seg += nlive;       // events completed: one per ray still alive after the row
        if (++e >= n_ev) break;
      }
      // ... counted once for every logical path that shares the row (the leg's rows: `mult` paths), and once as executed
      T.events += ev_logical + (unsigned long long)seg * mult;
      T.executed += ev_exec + seg;
      if (!forked && L < n_ev) break;       // the common leg ended for every lane before path q left it: so did every path still to come
      if (nlive != 0u) {
        // ---- the path is complete for nlive rays (as march_started_path) ---------------------------------
        T.n_scene += nlive;
        lanemask lit[K], lit_any = 0ull;
#pragma unroll
        for (int j = 0; j < K; j++) {
          const float cg = fmaf(r[j].dx, sx, fmaf(r[j].dy, sy, r[j].dz * sz));
          lit[j] = alive[j] & __ballot(cg > lobe_thr);
          lit_any |= lit[j];
        }
        if (lit_any != 0ull) {
          for (int j = 0; j < K; j++) {        // not unrolled: one copy of the weighted march
            lanemask lj = lit[0];
#pragma unroll
            for (int jj = 1; jj < K; jj++) lj = (j == jj) ? lit[jj] : lj;
            if (lj == 0ull) continue;
            const int l = g * K + j;
            // the path again, alone and with its weight: the same arithmetic on the ray, so the same ray bit for bit
            Ray rw = Ray{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx, s0.dy, s0.dz, s0.w0, 1.0f};
            { const float ns = lens->n_start[l]; rw.dx *= ns; rw.dy *= ns; rw.dz *= ns; }
            T.n_rm_lane += (unsigned long long)((unsigned)n_ev * (unsigned)__popcll(lj));
            T.n_rm_rows += (unsigned)n_ev;
            const int* w = seq;
            for (int rem = n_ev; rem > 0; --rem, ++w) {
              const unsigned se2 = (unsigned)*(const int __attribute__((address_space(4)))*)(w);
              const LfProgRow wr = load_prec(recs, se2 & 0xffffu);
              const LfWeightRow ww = load_wrec(wrecs, se2 & 0xffffu);
              const unsigned wfl = se2 >> 16;
              const float w_cn22 = j == 0 ? wr.cn22[0] : j == 1 ? wr.cn22[1] : wr.cn22[2];
              const float w_rn2 = j == 0 ? wr.rn2[0] : j == 1 ? wr.rn2[1] : wr.rn2[2];
              const float w_delta = j == 0 ? wr.delta[0] : j == 1 ? wr.delta[1] : wr.delta[2];
              const float w_fs = j == 0 ? ww.fs[0] : j == 1 ? ww.fs[1] : ww.fs[2];
              const float w_fo = j == 0 ? ww.fo[0] : j == 1 ? ww.fo[1] : ww.fo[2];
              const float w_fi = j == 0 ? ww.fi[0] : j == 1 ? ww.fi[1] : ww.fi[2];
              if (wfl & LF_EV_STOP) {
                (void)stop_event<true>(rw, wr.dzv, wr.h2, inv_stop_h, mask, a.mw, a.mh);
              } else {
                lanemask geom_ok;
                (void)surface_event<true>(rw, wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, w_cn22, w_rn2, w_delta, wr.h2,
                                          (wfl & LF_EV_REFLECT) != 0, (wfl & LF_EV_FLAT) != 0, wr.sgn, geom_ok, w_fs, w_fo, w_fi);
              }
            }
            const float qq = lobe_q(rw.dx, rw.dy, rw.dz, sx, sy, sz, sun_ss, inv_1mc);
            const float om = 1.0f - qq;
            float contrib = __fdiv_rn(rw.wn, rw.wd) * (om * om);
            contrib = (((lj >> lane) & 1ull) != 0ull && qq < 1.0f && contrib > 0.0f) ? contrib : 0.0f;
            T.n_light += contrib > 0.0f ? 1u : 0u;
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const float v = contrib * (lens->sun_radiance[c] * lens->lambda_rgb[l][c]);
              const unsigned long long fx = (unsigned long long)(v * 68719476736.0f);
              if (fx) atomicAdd(&s_acc[lane * 3 + c], fx);
            }
          }
        }
      }
      if (!forked) break;                     // (the primary path: the common leg to its end, nothing after it)
      // back to the common leg where path q left it
#pragma unroll
      for (int j = 0; j < K; j++) { r[j] = p[j]; alive[j] = palive[j]; }
      nlive = pn;
      depth = L;
    }
  }
}

// W1: the first (and then only) march of a started path carries its Fresnel / aperture weight.  false = geometry first,
// and the path is marched again with the weight, one wavelength at a time, only where a lane ended inside the lobe
// pre-test (k_march's scheme: 6 % of the STARTED rays are lit on the bench frame, so the weight's 17 of 44 vector
// instructions per event are mostly wasted in the first march: measured in profiles/r05_march_variants.txt).
// SHARED: the wave looks ONE table entry up per sample (its lanes share the sample's pupil sub-cell, a block holds the whole
// tile) and the started paths' common leg is marched once (march_started_set); otherwise (independent pixels, blocks smaller
// than a wave tile, the weight on every event, a selection out of order) every started path is marched alone.
#ifndef LF_SHARED_WAVES
#define LF_SHARED_WAVES 6     // waves per SIMD of the shared-leg kernel with K > 1 (two ray states per lane: 80 VGPR, a few spilled;
                              // a workgroup holds 2 waves per SIMD, so 5 runs as 4: 47 ms against 38 on the bench frame)
#endif
// MODE 0: every started path alone; 1 = SHARED (one table entry per wave and sample, the leg once: march_started_set)
template <int K, bool W1, int MODE>
__global__ __launch_bounds__(64 * kWgWaves, (K == 1 ? 8 : MODE == 1 ? LF_SHARED_WAVES : 6))
void k_march_cull(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
                  const int* __restrict__ seq_table, const LfProgRow* __restrict__ rec_table,
                  const LfWeightRow* __restrict__ wrec_table, const float* __restrict__ mask, MarchArgs a,
                  LfCullArgs cull, double* __restrict__ ghost, unsigned long long* __restrict__ accum,
                  unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long s_acc[64 * 3];
  __shared__ unsigned long long s_cnt[kMarchCounters];
  __shared__ int s_next, s_nlist;
  __shared__ unsigned short s_list[kListMax];
  __shared__ int2 s_meta[MODE == 1 ? kCullMaxPaths : 1];      // per path: events << 16 | events of the common leg; first row of its sequence
  const int tid = threadIdx.x;
  if (tid < 64 * 3) s_acc[tid] = 0ull;
  if (tid < kMarchCounters) s_cnt[tid] = 0ull;
  if (tid == 0) { s_next = 0; s_nlist = 0; }
  if (MODE == 1 && tid < pairs->n && tid < kCullMaxPaths) {
    const int n_ev = pairs->ev_cnt[tid], i1 = pairs->ij[tid][0];
    s_meta[tid] = make_int2((n_ev << 16) | (i1 < 0 ? n_ev : lens->n_surf - 1 - i1), pairs->ev_off[tid]);
  }
  __syncthreads();

  // tile of the workgroup: as k_march (XCD-aware slot swizzle, wave tile = 8 rows x 8 columns 2^xs apart)
  const int tiles_x = ((a.W + (8 << a.xs) - 1) >> (3 + a.xs)) << a.xs;
#ifdef LF_EXPERIMENTS
  if (cull.wg_clock && tid == 0) cull.wg_clock[2 * (size_t)blockIdx.x] = wall_clock64();
  struct ClockAtExit { unsigned long long* p; __device__ ~ClockAtExit() { if (p && threadIdx.x == 0) *p = wall_clock64(); } }
      clock_at_exit{cull.wg_clock ? cull.wg_clock + 2 * (size_t)blockIdx.x + 1 : nullptr};
#endif
  // the launch's last tiles are split finer than the rest (a.tail_from): the grid ends on short workgroups
  const unsigned n_head = (unsigned)a.tail_from * (unsigned)a.sgroups;
  const bool tail = a.tail_groups > 1 && blockIdx.x >= n_head;
  const int sgroups = tail ? a.tail_groups : a.sgroups;
  const int sg = tail ? (int)((blockIdx.x - n_head) % (unsigned)sgroups) : (int)(blockIdx.x % (unsigned)sgroups);
  const unsigned slot = tail ? (unsigned)a.tail_from + (blockIdx.x - n_head) / (unsigned)sgroups : blockIdx.x / (unsigned)sgroups;
  const int tile_lin = (int)((slot & ~63u) | ((slot & 7u) << 3) | ((slot >> 3) & 7u));
  if (tile_lin >= a.n_tiles) return;
  int tx, trow;
  march_tile_of(a, tile_lin, tiles_x, tx, trow);
  const unsigned tile_id = (unsigned)(trow * tiles_x + tx);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int x = ((tx >> a.xs) << (3 + a.xs)) + ((lane & 7) << a.xs) + (tx & ((1 << a.xs) - 1)), y = trow * 8 + (lane >> 3);
  const bool active = x < a.W && y >= a.y0 && y < a.y1;
  const lanemask active_mask = __ballot(active);
  const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
  // the tile's cull row: its columns and 8 rows lie inside one block -- unless the blocks are smaller than the tile
  // (cull.multi: 16 / 32-pixel blocks of a small frame), where every lane has the row of its own pixel's block
  const int blk = ((trow * 8) >> cull.blk_log2) * cull.blocks_x + ((((tx >> a.xs) << (3 + a.xs))) >> cull.blk_log2);
  const unsigned long long* const crow = cull.table + lf_cull_row_of_block(blk, cull.share_n, cull.share_nb) * (size_t)(cull.cells + 1);
  const int blk_lane = (min(y, a.H - 1) >> cull.blk_log2) * cull.blocks_x + (min(x, a.W - 1) >> cull.blk_log2);
  const unsigned long long* const crow_lane = cull.multi ? cull.table + lf_cull_row_of_block(blk_lane, cull.share_n, cull.share_nb) * (size_t)(cull.cells + 1) : crow;

  const float pitch = lens->pitch, pupil_h = lens->pupil_h, geom_norm = lens->geom_norm;
  const float half_w = a.half_w, half_h = a.half_h, vz_u = a.vz;
  const int GG = a.G * a.G;
  // do the lanes of a wave aim a stratified sample at ONE cell of the table?  (they share a sub-cell of the stratum;
  // the table has m cells per stratum axis)
  const bool per_lane = (1 << a.sub_bits) < cull.m;

  PathTally T;

  const int n_mine = (a.spp - sg + sgroups - 1) / sgroups;     // samples sg, sg + sgroups, ...
  for (int chunk0 = 0; chunk0 < n_mine; chunk0 += kListMax) {
    // ---- which of the tile's samples start any path at all (most do not) -----------------------------
    const int chunk_n = min(kListMax, n_mine - chunk0);
    for (int k = tid; k < chunk_n; k += 64 * kWgWaves) {
      const int s = sg + (chunk0 + k) * sgroups;
      bool work = true;     // (a sample whose lanes look their cells up one by one is listed: the sample loop finds out)
      if (s >= GG && !cull.multi) work = crow[cull.cells] != 0ull;
      if (s < GG && !per_lane && !cull.multi) {
        // the table cell of the sub-cell the wave's lanes all aim sample s at (the draw of the sample loop below)
        const uint4 r2 = philox4x32_10(make_uint4(tile_id, (unsigned)s, kDomainSubcell, 0u), a.key);
        const unsigned sxi = a.sub_bits ? (r2.x >> (32 - a.sub_bits)) : 0u, syi = a.sub_bits ? (r2.y >> (32 - a.sub_bits)) : 0u;
        const int cy = s / a.G, cx = s - cy * a.G;
        work = crow[(cy * cull.m + (int)((syi << cull.m_shift) >> a.sub_bits)) * cull.P + cx * cull.m + (int)((sxi << cull.m_shift) >> a.sub_bits)] != 0ull;
      }
      if (work) s_list[atomicAdd(&s_nlist, 1)] = (unsigned short)k;
    }
    __syncthreads();
    const int n_list = s_nlist;
    for (;;) {
      int k = 0;
      if (lane == 0) k = atomicAdd(&s_next, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= n_list) break;
      const int s = sg + (chunk0 + (int)s_list[k]) * sgroups;       // wave-uniform
      // ---- sensor sample -> initial ray (the expressions of k_march / sample_start) -----------------
      const uint4 rnd = philox4x32_10(make_uint4(p, (unsigned)s, kDomainMarch, 0u), a.key);
      const float jx = u01(rnd.x), jy = u01(rnd.y);
      float ua = u01(rnd.z), ub = u01(rnd.w);
      int entry = -1;
      if (s < GG) {
        const int cy = s / a.G, cx = s - cy * a.G;
        const uint4 r2 = philox4x32_10(make_uint4(tile_id, (unsigned)s, kDomainSubcell, 0u), a.key);
        const unsigned sxi = a.sub_bits ? (r2.x >> (32 - a.sub_bits)) : 0u;
        const unsigned syi = a.sub_bits ? (r2.y >> (32 - a.sub_bits)) : 0u;
        ua = ((float)cx + ((float)sxi + ua) * a.inv_sub) * a.inv_G;
        ub = ((float)cy + ((float)syi + ub) * a.inv_sub) * a.inv_G;
        if (!per_lane) entry = (cy * cull.m + (int)((syi << cull.m_shift) >> a.sub_bits)) * cull.P + cx * cull.m + (int)((sxi << cull.m_shift) >> a.sub_bits);
      }
      // the paths to start: one scalar load for the wave -- or, where the lanes aim at different cells of the table
      // (independent pixels), each lane's own mask and their union.  An unstratified sample (s >= G * G: a sample count
      // that is no square) takes the block's union entry.
      if (s >= GG) entry = cull.cells;
      const float pa0 = fmaf(2.0f, ua, -1.0f), pb0 = fmaf(2.0f, ub, -1.0f);
      const float X = -(((float)x + jx) - half_w) * pitch;
      const float Y = -(((float)y + jy) - half_h) * pitch;
      if (MODE == 1) {
        const unsigned long long todo = crow[entry];
        if (todo == 0ull) continue;
        const StartRay s0 = aim_at_pupil(X, Y, pa0, pb0, pupil_h, vz_u, geom_norm);
        march_started_set<K>(lens, pairs, seq_table, rec_table, wrec_table, mask, a, todo, active_mask, X, Y, s0, lane, s_acc, s_meta, T);
        continue;
      }
      unsigned long long mine, todo;
      if (entry >= 0 && !cull.multi) { todo = crow[entry]; mine = todo; }
      else {
        const int fx = min(cull.P - 1, (int)(ua * (float)cull.P)), fy = min(cull.P - 1, (int)(ub * (float)cull.P));
        mine = active ? crow_lane[entry >= 0 ? entry : fy * cull.P + fx] : 0ull;
        unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
        for (int off = 32; off > 0; off >>= 1) { lo |= __shfl_xor(lo, off); hi |= __shfl_xor(hi, off); }
        todo = ((unsigned long long)__builtin_amdgcn_readfirstlane(hi) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane(lo);
      }
      if (todo == 0ull) continue;
      const StartRay s0 = aim_at_pupil(X, Y, pa0, pb0, pupil_h, vz_u, geom_norm);
      unsigned long long left_q = todo;
      while (left_q != 0ull) {
        const int q = __builtin_ctzll(left_q);
        left_q &= left_q - 1ull;
        const lanemask start_mask = active_mask & __ballot(((mine >> q) & 1ull) != 0ull);   // (all active lanes when the wave shares a cell)
        march_started_path<K, W1>(lens, pairs, seq_table, rec_table, wrec_table, mask, a, q, start_mask, X, Y, s0, lane, lane, s_acc, T);
      }
    }
    __syncthreads();
    if (tid == 0) { s_next = 0; s_nlist = 0; }
    __syncthreads();
  }

  // ---- counters: one LDS add per wave, one global add per workgroup (slots as k_march: executed events =
  // events, every path marched on its own; no second march) ---------------------------------------------
  {
    unsigned long long v6 = T.n_light;
    for (int off = 32; off > 0; off >>= 1) v6 += __shfl_down(v6, off);
    const unsigned long long vals[kMarchCounters] = {T.n_rays, T.events, T.n_clip, T.n_vign, T.n_tir, T.n_scene, v6, T.executed, T.n_rm_lane, T.n_rm_rows};
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < kMarchCounters; i++)
        if (vals[i]) atomicAdd(&s_cnt[i], vals[i]);
    }
  }
  __syncthreads();
  if (wave == 0 && lane < kMarchCounters && s_cnt[lane]) atomicAdd(&counters[lane], s_cnt[lane]);

  if (tail) {
    // a tile of the split tail: the integer sums of its workgroups meet in tail_acc; the LAST to arrive converts them (the same
    // conversion as a whole tile's: the sums do not depend on who added what) and leaves sums and arrivals zero for the next launch
    if (wave != 0) return;
    const unsigned tt = slot - (unsigned)a.tail_from;
    unsigned long long* const acc = cull.tail_acc + (size_t)tt * 192u;
#pragma unroll
    for (int c = 0; c < 3; c++)
      if (s_acc[lane * 3 + c]) atomicAdd(&acc[lane * 3 + c], s_acc[lane * 3 + c]);
    __threadfence();
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(&cull.tail_done[tt], 1);
    arrived = __shfl(arrived, 0);
    if (arrived != sgroups - 1) return;
    __threadfence();
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const unsigned long long sum = atomicExch(&acc[lane * 3 + c], 0ull);
      if (active) {
        const double v = ((double)sum * (1.0 / 68719476736.0)) / (double)a.spp;
        ghost[3 * (size_t)p + c] = a.accumulate ? ghost[3 * (size_t)p + c] + v : v;
      }
    }
    if (lane == 0) atomicExch(&cull.tail_done[tt], 0);
    return;
  }
  if (wave == 0 && active) {
    if (a.sgroups == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double v = ((double)s_acc[lane * 3 + c] * (1.0 / 68719476736.0)) / (double)a.spp;
        ghost[3 * (size_t)p + c] = a.accumulate ? ghost[3 * (size_t)p + c] + v : v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (s_acc[lane * 3 + c]) atomicAdd(&accum[3 * (size_t)p + c], s_acc[lane * 3 + c]);
    }
  }
}

// ---- the same march, COMPACTED: one lane = one (pixel, sample) that starts the path --------------------------
// Where every pixel draws its own pupil point (lf_set_pupil_subcells(0): the independent-pixel estimator) the lanes of
// a wave tile aim at different cells of the table, want different paths, and a wave that marches path q for its tile
// and sample does so with the few lanes that want it: the same events as the coherent specification in 2.4 x the
// time (profiles/r05_march_variants.txt).  Here the tile's workgroup first LISTS what is to be started -- for a chunk
// of samples every (pixel, sample) looks its mask up, the paths are counted by ballot, and the (pixel, sample) items
// are written into one LDS array sorted by path -- and then marches the list: a wave takes 64 items of ONE path,
// rebuilds each lane's start ray from its (pixel, sample) and marches with every lane started.  Sums are integers
// in LDS, so pixels and counters do not depend on the order: bit for bit the frame of k_march_cull and of the oracle.
constexpr int kItemCap = 16384;      // items of a chunk (2 bytes each)
constexpr int kItemChunk = 256;      // samples per chunk at most (8 bits of an item; the pixel takes 6)
constexpr int kItemCellCache = 64;   // chunks of up to this many samples keep each (sample, pixel)'s table cell between the two listing passes

template <int K>
__global__ __launch_bounds__(64 * kWgWaves, 6)      // (42 KB of LDS lists: three workgroups per CU)
void k_march_items(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
                   const int* __restrict__ seq_table, const LfProgRow* __restrict__ rec_table,
                   const LfWeightRow* __restrict__ wrec_table, const float* __restrict__ mask, MarchArgs a,
                   LfCullArgs cull, double* __restrict__ ghost, unsigned long long* __restrict__ accum,
                   unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long s_acc[64 * 3];
  __shared__ unsigned long long s_cnt[kMarchCounters];
  __shared__ int s_pcount[kCullMaxPaths], s_poff[kCullMaxPaths + 1], s_goff[kCullMaxPaths + 1], s_pfill[kCullMaxPaths];
  __shared__ int s_next;
  __shared__ unsigned short s_items[kItemCap];
  __shared__ unsigned short s_cell[kItemCellCache * 64];      // the table cell of every (sample of the chunk, pixel): count -> fill
  const int tid = threadIdx.x;
  if (tid < 64 * 3) s_acc[tid] = 0ull;
  if (tid < kMarchCounters) s_cnt[tid] = 0ull;
  __syncthreads();

  const int tiles_x = ((a.W + (8 << a.xs) - 1) >> (3 + a.xs)) << a.xs;
  const int sg = blockIdx.x % a.sgroups;
  const unsigned slot = blockIdx.x / a.sgroups;
  const int tile_lin = (int)((slot & ~63u) | ((slot & 7u) << 3) | ((slot >> 3) & 7u));
  if (tile_lin >= a.n_tiles) return;
  int tx, trow;
  march_tile_of(a, tile_lin, tiles_x, tx, trow);
  const unsigned tile_id = (unsigned)(trow * tiles_x + tx);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int n_paths = pairs->n;
  const float pitch = lens->pitch, pupil_h = lens->pupil_h, geom_norm = lens->geom_norm;
  const float half_w = a.half_w, half_h = a.half_h, vz_u = a.vz;
  const int GG = a.G * a.G;
  // the cull row of THIS lane's pixel (one block holds the whole tile unless the blocks are smaller than it: cull.multi)
  const unsigned long long* crow;
  {
    const int lx = ((tx >> a.xs) << (3 + a.xs)) + ((lane & 7) << a.xs) + (tx & ((1 << a.xs) - 1)), ly = trow * 8 + (lane >> 3);
    const int blk = cull.multi ? (min(ly, a.H - 1) >> cull.blk_log2) * cull.blocks_x + (min(lx, a.W - 1) >> cull.blk_log2)
                               : ((trow * 8) >> cull.blk_log2) * cull.blocks_x + ((((tx >> a.xs) << (3 + a.xs))) >> cull.blk_log2);
    crow = cull.table + lf_cull_row_of_block(blk, cull.share_n, cull.share_nb) * (size_t)(cull.cells + 1);
  }
  // pixel `px` of the tile (the lane order of k_march / k_march_cull)
  auto pixel_of = [&](int px, int& x, int& y) {
    x = ((tx >> a.xs) << (3 + a.xs)) + ((px & 7) << a.xs) + (tx & ((1 << a.xs) - 1));
    y = trow * 8 + (px >> 3);
    return x < a.W && y >= a.y0 && y < a.y1;
  };
  // sample s of pixel (x, y): the point of the pupil square it aims at (the expressions of sample_start), with the
  // pixel jitter beside it
  auto pupil_point = [&](int x, int y, int s, float& ua, float& ub, float& jx, float& jy) {
    const uint4 rnd = philox4x32_10(make_uint4((unsigned)y * (unsigned)a.W + (unsigned)x, (unsigned)s, kDomainMarch, 0u), a.key);
    jx = u01(rnd.x); jy = u01(rnd.y);
    ua = u01(rnd.z); ub = u01(rnd.w);
    if (s < GG) {
      const int cy = s / a.G, cx = s - cy * a.G;
      unsigned sxi = 0u, syi = 0u;
      if (a.sub_bits) {     // (no sub-cells, no draw: s differs from lane to lane here, the draw would be a vector one)
        const uint4 r2 = philox4x32_10(make_uint4(tile_id, (unsigned)s, kDomainSubcell, 0u), a.key);
        sxi = r2.x >> (32 - a.sub_bits); syi = r2.y >> (32 - a.sub_bits);
      }
      ua = ((float)cx + ((float)sxi + ua) * a.inv_sub) * a.inv_G;
      ub = ((float)cy + ((float)syi + ub) * a.inv_sub) * a.inv_G;
    }
  };
  // the paths pixel (lane) starts for the chunk's k-th sample: the mask of the table cell that holds its pupil point
  // ... as the index of its table entry (cells = the block's union entry: an unstratified sample; 0xffff: no pixel there)
  auto cell_of = [&](int k_global) -> unsigned {
    int x, y;
    if (!pixel_of(lane, x, y)) return 0xffffu;
    const int s = sg + k_global * a.sgroups;
    if (s >= GG) return (unsigned)cull.cells;
    float ua, ub, jx, jy;
    pupil_point(x, y, s, ua, ub, jx, jy);
    const int fx = min(cull.P - 1, (int)(ua * (float)cull.P)), fy = min(cull.P - 1, (int)(ub * (float)cull.P));
    return (unsigned)(fy * cull.P + fx);
  };
  auto mask_at = [&](unsigned cell) -> unsigned long long { return cell == 0xffffu ? 0ull : crow[cell]; };
  auto wave_or = [](unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    for (int off = 32; off > 0; off >>= 1) { lo |= __shfl_xor(lo, off); hi |= __shfl_xor(hi, off); }
    return ((unsigned long long)__builtin_amdgcn_readfirstlane(hi) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane(lo);
  };

  PathTally T;
  const int n_mine = (a.spp - sg + a.sgroups - 1) / a.sgroups;     // samples sg, sg + sgroups, ...
  // (the chunk to begin with: what the table's started fraction says fits the item list -- a chunk that does not fit is
  // counted for nothing and halved)
  int c0 = 0, ch = min(max(1, min(kItemChunk, cull.items_chunk0)), n_mine);
  while (c0 < n_mine) {
    const int chn = min(ch, n_mine - c0);
    // ---- count: how many (pixel, sample) items of this chunk start each path (a wave = one sample's 64 pixels) ---
    if (tid < kCullMaxPaths) { s_pcount[tid] = 0; s_pfill[tid] = 0; }
    if (tid == 0) s_next = 0;
    __syncthreads();
    const bool cached = chn <= kItemCellCache;
    for (int k = wave; k < chn; k += kWgWaves) {
      const unsigned cell = cell_of(c0 + k);
      if (cached) s_cell[k * 64 + lane] = (unsigned short)cell;
      const unsigned long long mine = mask_at(cell);
      unsigned long long u = wave_or(mine);
      while (u != 0ull) {
        const int q = __builtin_ctzll(u);
        u &= u - 1ull;
        const int c = __popcll(__ballot(((mine >> q) & 1ull) != 0ull));
        if (lane == 0) atomicAdd(&s_pcount[q], c);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int off = 0, goff = 0;
      for (int q = 0; q < n_paths; q++) { s_poff[q] = off; s_goff[q] = goff; off += s_pcount[q]; goff += (s_pcount[q] + 63) >> 6; }
      s_poff[n_paths] = off; s_goff[n_paths] = goff;
    }
    __syncthreads();
    const int total = s_poff[n_paths];
    if (total > kItemCap && chn > 1) {      // too many for the list: a shorter chunk (one sample's 64 x 46 always fit)
      ch = max(1, chn >> 1);
      __syncthreads();
      continue;
    }
    // ---- fill: the items, sorted by path ------------------------------------------------------------------------
    for (int k = wave; k < chn; k += kWgWaves) {
      const unsigned long long mine = mask_at(cached ? (unsigned)s_cell[k * 64 + lane] : cell_of(c0 + k));
      unsigned long long u = wave_or(mine);
      while (u != 0ull) {
        const int q = __builtin_ctzll(u);
        u &= u - 1ull;
        const lanemask b = __ballot(((mine >> q) & 1ull) != 0ull);
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_pfill[q], (int)__popcll(b));
        base = __builtin_amdgcn_readfirstlane(base);
        if ((b >> lane) & 1ull)
          s_items[s_poff[q] + base + (int)__popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)((k << 6) | lane);
      }
    }
    __syncthreads();
    // ---- march the list: 64 items of one path per wave ------------------------------------------------------------
    const int n_groups_total = s_goff[n_paths];
    for (;;) {
      int gi = 0;
      if (lane == 0) gi = atomicAdd(&s_next, 1);
      gi = __builtin_amdgcn_readfirstlane(gi);
      if (gi >= n_groups_total) break;
      int q = 0;
      while (s_goff[q + 1] <= gi) q++;      // (wave-uniform: the path whose groups hold gi)
      const int first = s_poff[q] + ((gi - s_goff[q]) << 6), end = s_poff[q] + s_pcount[q];
      const bool have = first + lane < end;
      const unsigned item = have ? (unsigned)s_items[first + lane] : 0u;
      const int px = (int)(item & 63u), s = sg + (c0 + (int)(item >> 6)) * a.sgroups;
      int x, y;
      (void)pixel_of(px, x, y);
      float ua, ub, jx, jy;
      pupil_point(x, y, s, ua, ub, jx, jy);
      const float X = -(((float)x + jx) - half_w) * pitch;
      const float Y = -(((float)y + jy) - half_h) * pitch;
      const StartRay s0 = aim_at_pupil(X, Y, fmaf(2.0f, ua, -1.0f), fmaf(2.0f, ub, -1.0f), pupil_h, vz_u, geom_norm);
      march_started_path<K, false>(lens, pairs, seq_table, rec_table, wrec_table, mask, a, q, __ballot(have), X, Y, s0, lane, px, s_acc, T);
    }
    __syncthreads();
    c0 += chn;
  }

  {
    unsigned long long v6 = T.n_light;
    for (int off = 32; off > 0; off >>= 1) v6 += __shfl_down(v6, off);
    const unsigned long long vals[kMarchCounters] = {T.n_rays, T.events, T.n_clip, T.n_vign, T.n_tir, T.n_scene, v6, T.executed, T.n_rm_lane, T.n_rm_rows};
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < kMarchCounters; i++)
        if (vals[i]) atomicAdd(&s_cnt[i], vals[i]);
    }
  }
  __syncthreads();
  if (wave == 0 && lane < kMarchCounters && s_cnt[lane]) atomicAdd(&counters[lane], s_cnt[lane]);
  int x, y;
  const bool active = pixel_of(lane, x, y);
  if (wave == 0 && active) {
    const size_t p = (size_t)y * (size_t)a.W + (size_t)x;
    if (a.sgroups == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double v = ((double)s_acc[lane * 3 + c] * (1.0 / 68719476736.0)) / (double)a.spp;
        ghost[3 * p + c] = a.accumulate ? ghost[3 * p + c] + v : v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (s_acc[lane * 3 + c]) atomicAdd(&accum[3 * p + c], s_acc[lane * 3 + c]);
    }
  }
}

uint64_t fnv(uint64_t h, const void* data, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

}  // namespace

// table cells per axis inside one stratum: 4, in a table of at most 128 cells per axis.  Where a wave's lanes share
// one sub-cell of the stratum (2^sub_bits >= m sub-cells per axis) the wave looks its cell up ONCE; otherwise
// (independent pixels, unstratified samples) every lane looks up the cell of its own pupil point.
static int cull_m(const lf_ctx* ctx, int G) {
  (void)ctx;
  int m = 4;
#ifdef LF_EXPERIMENTS
  if (const char* e = std::getenv("LF_CULL_M")) m = std::max(1, std::atoi(e));
#endif
  while (m > 1 && G * m > 128) m >>= 1;
  return m;
}

// Does the cull apply to this launch, and if not, why (lf_get_cull_reason)
int lf_cull_reason_of(const lf_ctx* ctx, int G) {
  if (ctx->march_cull == 0) return LF_CULL_OFF;
#ifdef LF_EXPERIMENTS
  if (const char* e = std::getenv("LF_MARCH_CULL")) if (std::atoi(e) == 0) return LF_CULL_OFF;
#endif
  if (ctx->lens.stop < 0) return LF_CULL_NO_STOP;
  if (!ctx->lens_lambda_monotonic) return LF_CULL_DISPERSION;
  // (a mask has 64 bits; up to 128 paths go in two launches over the halves of the selection: lfk_march -- with a table
  // of this context's own)
  if (ctx->pairs.n > 2 * kCullMaxPaths || (ctx->pairs.n > kCullMaxPaths && ctx->cull_share_how != 0)) return LF_CULL_TOO_MANY_PATHS;
  if (G < 1 || G > 64) return LF_CULL_TOO_MANY_SAMPLES;
  // a block must be SMALL on the sensor for 15 rays to bound it: <= 1.8 mm (the full-enumeration comparison finds no
  // skipped lit ray up to 7.2 mm blocks, profiles/r05_cull_block_size.json) -- frames narrower than 1280 pixels on a
  // 36 mm sensor take blocks of 32 or 16 pixels (lf_cull_block_log2)
  if (lf_cull_block_log2(ctx, 1, 1) < 0) return LF_CULL_BLOCK_TOO_LARGE;
  return LF_CULL_APPLIED;
}
bool lf_cull_applies(const lf_ctx* ctx, int G) { return lf_cull_reason_of(ctx, G) == LF_CULL_APPLIED; }

// log2 of the side of a cull block in pixels for this frame: 6 (64 pixels) where that is <= kCullMaxBlockMm on the sensor;
// 7 where 128 are still <= kCullBigBlockMm AND the launch has few samples (measured at 4K: 256 spp x 3 wavelengths tie, 1024 x 8
// lose: profiles/r05_march_variants.txt); 5 or 4 (32 / 16 pixels) where 64 are too large (a frame narrower than 1280
// pixels on 36 mm): a wave tile, (8 << xs) pixels wide, then spans several blocks and its lanes look their rows up one by
// one (LfCullArgs::multi); -1: even 16 pixels are too large.
int lf_cull_block_log2(const lf_ctx* ctx, int spp, int n_lambda) {
  const double mm_per_px = (double)ctx->sensor_w_mm / (double)std::max(1, ctx->W);
  int lg = kCullBlockLog2;
  while (lg > 4 && (double)(1 << lg) * mm_per_px > kCullMaxBlockMm) lg--;
  if ((double)(1 << lg) * mm_per_px > kCullMaxBlockMm) return -1;
  if (lg == kCullBlockLog2 && !ctx->deal_by_block && (double)(2 << lg) * mm_per_px <= kCullBigBlockMm && (long long)spp * n_lambda < 768) {
#ifdef LF_EXPERIMENTS
    if (std::getenv("LF_CULL_SMALL_BLOCKS")) return lg;
#endif
    return lg + 1;
  }
  return lg;
}

// set bits of the table's cells (not of the union entries): the (block, cell, path) combinations the march will start
__global__ void k_cull_popcount(const unsigned long long* __restrict__ table, size_t rows, int cells,
                                unsigned long long* __restrict__ out) {
  const size_t n = rows * (size_t)(cells + 1);
  unsigned long long sum = 0ull;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if ((int)(i % (size_t)(cells + 1)) != cells) sum += (unsigned long long)__popcll(table[i]);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63u) == 0u && sum) atomicAdd(out, sum);
}

// ---- the audit: what the table drops, sampled -----------------------------------------------------------------------
// The pre-pass's bounds are estimates (see the head of this file): what stands behind them is a search for counter-examples,
// and a search covers the prescriptions it drew.  So every table is also CHECKED where it is used: for every (block, cell,
// path) combination it does not start, `density` rays of that box -- a random pixel position in the block, a random point of
// the pupil cell, one of the launch's wavelengths; Philox keyed by the launch -- are marched as the march would (geometry
// only, real apertures), and a ray that ends inside the sun's lobe REFUTES the table: the launch marches everything
// (k_march, the path tree) and says so (lf_get_cull_audit, lf_get_cull_reason).  One wave = 64 cells of one block, the
// paths one after the other (wave-uniform event sequence, rows through the scalar cache, lanes whose cell starts the
// path idle): ~10 events per ray, 8.8e7 rays on the bench frame.
constexpr unsigned kDomainAudit = 0x0a0d17c5u;
struct CullAuditArgs {
  int W, H;
  float pitch, half_w, half_h;
  int blocks_x, blocks_y, blk_log2, share_n, share_nb;
  int P, n_paths, n_lambda, march_k, prog_recs;
  float pupil_h, vz, geom_norm, inv_stop_h, lobe_thr;
  int mw, mh;
  uint2 key;
  int density;
  int blk_first, blk_step;     // the blocks audited: blk_first + k blk_step (all of them; the frame dealt by blocks: this rank's)
};
__global__ __launch_bounds__(256) void k_cull_audit(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
                                                    const int* __restrict__ seq_table, const LfProgRow* __restrict__ rec_table,
                                                    const float* __restrict__ mask, CullAuditArgs a,
                                                    const unsigned long long* __restrict__ table,
                                                    unsigned long long* __restrict__ out) {
  const int blk = a.blk_first + (int)blockIdx.y * a.blk_step;
  const int cells = a.P * a.P;
  const int cell = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if ((cell & ~63) >= cells) return;
  const bool valid = cell < cells;
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned long long* const row = table + lf_cull_row_of_block(blk, a.share_n, a.share_nb) * (size_t)(cells + 1);
  const unsigned long long bits = valid ? row[cell] : ~0ull;
  const int ci = cell % a.P, cj = cell / a.P;
  const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
  const float px0 = (float)(bx << a.blk_log2), px1 = fminf((float)a.W, (float)((bx + 1) << a.blk_log2));
  const float py0 = (float)(by << a.blk_log2), py1 = fminf((float)a.H, (float)((by + 1) << a.blk_log2));
  const float invP = 1.0f / (float)a.P;
  const float sx = lens->sun_dir[0], sy = lens->sun_dir[1], sz = lens->sun_dir[2];
  const float inv_1mc = lens->sun_inv_one_minus_cos, sun_ss = lens->sun_ss;
  unsigned n_rays = 0u, n_lit = 0u;
  for (int q = 0; q < a.n_paths; q++) {
    const bool dropped = valid && ((bits >> q) & 1ull) == 0ull;
    const lanemask todo = __ballot(dropped);
    if (todo == 0ull) continue;
    const int n_ev = pairs->ev_cnt[q];
    const int* const seq = seq_table + pairs->ev_off[q];
    for (int d = 0; d < a.density; d++) {
      // the wavelength of this wave's rays: one per (block, 64 cells, path, repetition), so the rows stay scalar
      const uint4 ru = philox4x32_10(make_uint4((unsigned)blk, blockIdx.x * 4u + (threadIdx.x >> 6), kDomainAudit, (unsigned)(q | (d << 8))), a.key);
      const int l = (int)(__builtin_amdgcn_readfirstlane(ru.x) % (unsigned)a.n_lambda);
      const int g = l / a.march_k, j = l - g * a.march_k;
      const LfProgRow* const recs = rec_table + (size_t)g * (size_t)a.prog_recs;
      const uint4 rnd = philox4x32_10(make_uint4((unsigned)(blk * cells + cell), (unsigned)(q | (d << 8)), kDomainAudit, 1u), a.key);
      const float X = -((px0 + u01(rnd.x) * (px1 - px0)) - a.half_w) * a.pitch;
      const float Y = -((py0 + u01(rnd.y) * (py1 - py0)) - a.half_h) * a.pitch;
      const float ua = ((float)ci + u01(rnd.z)) * invP, ub = ((float)cj + u01(rnd.w)) * invP;
      const StartRay s0 = aim_at_pupil(X, Y, fmaf(2.0f, ua, -1.0f), fmaf(2.0f, ub, -1.0f), a.pupil_h, a.vz, a.geom_norm);
      const float ns = lens->n_start[l];
      Ray r{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx * ns, s0.dy * ns, s0.dz * ns, 1.0f, 1.0f};
      lanemask alive = todo;
      n_rays += dropped ? 1u : 0u;
      for (int e = 0; e < n_ev && alive != 0ull; e++) {
        const unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(seq + e);
        const LfProgRow wr = load_prec(recs, se & 0xffffu);
        const unsigned kind = se >> 16;
        if (kind & LF_EV_STOP) alive &= stop_event<false>(r, wr.dzv, wr.h2, a.inv_stop_h, mask, a.mw, a.mh);
        else {
          lanemask geom_ok;
          const float cn22 = j == 0 ? wr.cn22[0] : j == 1 ? wr.cn22[1] : wr.cn22[2];
          const float rn2 = j == 0 ? wr.rn2[0] : j == 1 ? wr.rn2[1] : wr.rn2[2];
          const float delta = j == 0 ? wr.delta[0] : j == 1 ? wr.delta[1] : wr.delta[2];
          alive &= surface_event<false>(r, wr.dzv, wr.curv, wr.ch, wr.c2, wr.sc, cn22, rn2, delta, wr.h2, (kind & LF_EV_REFLECT) != 0,
                                        (kind & LF_EV_FLAT) != 0, wr.sgn, geom_ok);
        }
      }
      if (alive == 0ull) continue;
      const float cg = fmaf(r.dx, sx, fmaf(r.dy, sy, r.dz * sz));
      const bool lit = ((alive >> lane) & 1ull) != 0ull && cg > a.lobe_thr && lobe_q(r.dx, r.dy, r.dz, sx, sy, sz, sun_ss, inv_1mc) < 1.0f;
      n_lit += lit ? 1u : 0u;
    }
  }
  unsigned long long v0 = n_rays, v1 = n_lit;
  for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
  if (lane == 0) {
    if (v0) atomicAdd(&out[0], v0);
    if (v1) atomicAdd(&out[1], v1);
  }
}

// The table is complete (built here, or completed by an all-gather): what fraction of all (block, cell, path)
// combinations it starts -- counted from the table itself, so that every rank of a shared table finds the same number
// and takes the same kernel -- and what its audit says.  `hash`: of the inputs it was built from; published only here.
lf_status lfk_cull_finish(lf_ctx* ctx, uint64_t hash) {
  if (!ctx->cull_popc_dev) LF_HIP(ctx, hipMalloc((void**)&ctx->cull_popc_dev, 4 * sizeof(unsigned long long)));
  LF_HIP(ctx, hipMemsetAsync(ctx->cull_popc_dev, 0, 4 * sizeof(unsigned long long), ctx->stream));
  const size_t rows = ctx->cull_share_nb > 0 ? (size_t)ctx->cull_share_nb * (size_t)std::max(1, ctx->cull_share_n_resident)
                                             : (size_t)ctx->cull_bx * ctx->cull_by;
  // the blocks this context marches: all of them -- or, the frame dealt by blocks, its own (whose rows lie together: its slab)
  const int n_blk = ctx->cull_bx * ctx->cull_by;
  const bool own = ctx->cull_own_rows_only;
  const int own_n = own ? ctx->cull_share_n_resident : 1, own_rank = own ? ctx->cull_share_rank : 0;
  const int n_own_blk = (n_blk - own_rank + own_n - 1) / own_n;
  const size_t row_entries = (size_t)ctx->cull_cells + 1;
  if (own) hipLaunchKernelGGL(k_cull_popcount, dim3(1024), dim3(256), 0, ctx->stream, ctx->cull_dev + (size_t)own_rank * ctx->cull_share_nb * row_entries,
                              (size_t)n_own_blk, ctx->cull_cells, ctx->cull_popc_dev);
  else hipLaunchKernelGGL(k_cull_popcount, dim3(1024), dim3(256), 0, ctx->stream, ctx->cull_dev, rows, ctx->cull_cells, ctx->cull_popc_dev);
  LF_HIP(ctx, hipGetLastError());
  if (ctx->cull_audit_density > 0) {
    const LfLensDev& L = ctx->lens;
    const LfApertureDev& m = ctx->ap[LF_APERTURE_STARBURST];
    CullAuditArgs a;
    std::memset(&a, 0, sizeof(a));
    a.W = ctx->W; a.H = ctx->H; a.pitch = L.pitch; a.half_w = 0.5f * (float)ctx->W; a.half_h = 0.5f * (float)ctx->H;
    a.blocks_x = ctx->cull_bx; a.blocks_y = ctx->cull_by; a.blk_log2 = ctx->cull_blk_log2;
    a.share_n = ctx->cull_share_nb > 0 ? ctx->cull_share_n_resident : 1; a.share_nb = ctx->cull_share_nb;
    a.P = ctx->cull_P; a.n_paths = ctx->pairs.n; a.n_lambda = L.n_lambda; a.march_k = ctx->march_k; a.prog_recs = ctx->pairs.prog_recs;
    a.pupil_h = L.pupil_h; a.vz = L.pupil_z - L.z_sensor; a.geom_norm = L.geom_norm; a.inv_stop_h = 1.0f / L.stop_h;
    a.lobe_thr = lf_march_lobe_thr(L);
    a.mw = m.w; a.mh = m.h;
    const uint64_t k = (ctx->cull_audit_seq++) * 0x9e3779b97f4a7c15ull ^ hash;
    a.key = make_uint2((unsigned)k, (unsigned)(k >> 32));
    a.density = ctx->cull_audit_density;
    a.blk_first = own_rank; a.blk_step = own_n;
    const dim3 grid((unsigned)((ctx->cull_cells + 255) / 256), (unsigned)n_own_blk);
    hipEvent_t ev = lf_timing_begin(ctx, LFK_CULL_AUDIT);
    hipLaunchKernelGGL(k_cull_audit, grid, dim3(256), 0, ctx->stream, ctx->lens_dev, ctx->pairs_dev,
                       (const int*)(ctx->prog_dev + ctx->prog_seq_off), (const LfProgRow*)(ctx->prog_dev + ctx->prog_rec_off),
                       m.texels, a, ctx->cull_dev, ctx->cull_popc_dev + 1);
    lf_timing_end(ctx, LFK_CULL_AUDIT, ev);
    LF_HIP(ctx, hipGetLastError());
  }
  unsigned long long got[4] = {0ull, 0ull, 0ull, 0ull};
  LF_HIP(ctx, hipMemcpyAsync(got, ctx->cull_popc_dev, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // a bring-up call the host gave up on (lf_comm_poison) publishes nothing: the table it waited for may never have arrived
  if (ctx->comm_poisoned.load()) return lf_fail(ctx, LF_ERR_STATE, "cull table: the communicator was abandoned while the table was being completed");
  ctx->cull_started_fraction = (double)got[0] / ((double)n_own_blk * (double)ctx->cull_cells * (double)std::max(1, ctx->pairs.n));
  ctx->cull_audit_rays += got[1];
  ctx->cull_audit_lit += got[2];
  ctx->cull_hash = hash;
  ctx->cull_bad_hash = 0;
  if (got[2] != 0ull) {          // an audit ray of a dropped box reached the light: this table is not used
    ctx->cull_bad_hash = hash;
    ctx->cull_audit_tripped++;
  }
  return LF_OK;
}

lf_status lfk_cull_prepass(lf_ctx* ctx, int G, int spp) {
  const LfLensDev& L = ctx->lens;
  const lf_ctx::CullRules& R = ctx->cull_rules;
  CullLevelArgs a;
  std::memset(&a, 0, sizeof(a));
  a.W = ctx->W; a.H = ctx->H;
  a.pitch = L.pitch; a.half_w = 0.5f * (float)ctx->W; a.half_h = 0.5f * (float)ctx->H;
  a.blk_log2 = lf_cull_block_log2(ctx, spp, L.n_lambda);
  if (a.blk_log2 < 0) return lf_fail(ctx, LF_ERR_STATE, "cull pre-pass: no block size applies (lf_cull_applies comes first)");
  a.blocks_x = (ctx->W + (1 << a.blk_log2) - 1) >> a.blk_log2;
  a.blocks_y = (ctx->H + (1 << a.blk_log2) - 1) >> a.blk_log2;
  // a table shared between ranks: this one builds the rows of the blocks b with b % n == rank (lf_cull_row_of_block) -- or
  // (cull_share_how 3: the frame dealt by blocks, lf_set_block_deal) ONLY those, nobody needs the others.  The deal's block
  // is 64 pixels: a frame whose cull blocks are smaller builds the whole table on every rank (small frames: cheap)
  const bool shared = ctx->cull_share_how != 0 && ctx->cull_share_n > 1 && (ctx->cull_share_how != 3 || a.blk_log2 == kDealBlockLog2);
  a.share_n = shared ? ctx->cull_share_n : 1;
  a.share_rank = shared ? ctx->cull_share_rank : 0;
  a.share_nb = (a.blocks_x * a.blocks_y + a.share_n - 1) / a.share_n;
  const int m = cull_m(ctx, G);
  a.P_final = G * m;
  a.n_paths = ctx->pairs.n;
  // dispersion is monotonic in the wavelength's column (lf_derive_lens checks it; otherwise the launch does not cull:
  // LF_CULL_DISPERSION): the two ends of the spectrum bracket what lies between
  a.lam[0] = (L.n_lambda - 1) / 2; a.lam[1] = 0; a.lam[2] = L.n_lambda - 1;
  a.march_k = ctx->march_k; a.prog_recs = ctx->pairs.prog_recs;
  a.pupil_h = L.pupil_h; a.vz = L.pupil_z - L.z_sensor; a.geom_norm = L.geom_norm;
  a.stop_h = L.stop_h; a.inv_stop_h = 1.0f / L.stop_h;
  a.sx = L.sun_dir[0]; a.sy = L.sun_dir[1];
  {
    // a ray contributes only if d.s > lobe_thr (lfk_march): |d - s|^2 = 2 - 2 d.s < 2 (1 - thr) for unit vectors,
    // and the (x, y) projection is no longer than the vector; the float march's directions are unit to ~1e-6
    const double thr = 1.0 - (1.0625 / (double)L.sun_inv_one_minus_cos) * (1.0 + 1e-6) - 4e-7;
    a.rho = (float)(std::sqrt(2.0 * (1.0 - thr)) * 1.001 + 1e-5);
  }
  std::memcpy(a.occ, ctx->cull_occ, sizeof(a.occ));
  a.keep_partial = R.keep_partial; a.lost_rel = R.lost_rel; a.lost_abs = R.lost_abs; a.strict = R.strict; a.lobe_k = R.lobe_k;
  a.strict_lost = R.strict_lost; a.slack_mode = R.slack_mode; a.disable = R.disable;
  // the levels: P_final, halved while it stays even and >= 8 (a coarser box is too curved for 15 rays to bound)
  int levels[8], n_levels = 0;
  {
    int P = a.P_final;
    levels[n_levels++] = P;
    int coarsest = 8;
#ifdef LF_EXPERIMENTS
    if (const char* e = std::getenv("LF_CULL_P0")) coarsest = std::max(2, std::atoi(e));
#endif
    while (n_levels < 8 && P % 2 == 0 && P / 2 >= coarsest) { P /= 2; levels[n_levels++] = P; }
    std::reverse(levels, levels + n_levels);
  }

  // is the resident table the one these inputs give?
  uint64_t h = 0xcbf29ce484222325ull;
  a.margin = R.margin;
  h = fnv(h, &a, sizeof(a));
  h = fnv(h, levels, sizeof(int) * (size_t)n_levels);
  h = fnv(h, &L, sizeof(L));
  h = fnv(h, ctx->pairs.ij, sizeof(int) * 2 * (size_t)ctx->pairs.n);
  h = fnv(h, &ctx->mask_generation, sizeof(ctx->mask_generation));
  h = fnv(h, &ctx->cull_rules_custom, sizeof(ctx->cull_rules_custom));
  if (h == 0) h = 1;
  const size_t nblk = (size_t)a.blocks_x * a.blocks_y;
  const size_t rows = (size_t)a.share_nb * (size_t)a.share_n;         // (= nblk unless shared: equal slabs, the last ones padded)
  const size_t row_entries = (size_t)a.P_final * a.P_final + 1;
  const size_t entries = rows * row_entries;
  // (mode 2 rebuilds at every launch -- except the table lf_cull_commit has just completed for this very launch)
  bool reuse = (ctx->march_cull == 1 || ctx->cull_fresh) && ctx->cull_dev && ctx->cull_hash == h;
#ifdef LF_EXPERIMENTS
  if (std::getenv("LF_CULL_NO_REUSE")) reuse = false;
#endif
  ctx->cull_fresh = false;
  ctx->cull_bx = a.blocks_x; ctx->cull_by = a.blocks_y; ctx->cull_cells = a.P_final * a.P_final; ctx->cull_G = G;
  ctx->cull_P = a.P_final; ctx->cull_m = m; ctx->cull_blk_log2 = a.blk_log2;
  if (reuse) return LF_OK;
  if (shared && ctx->cull_share_how == 2 && !ctx->cull_prepare_only)
    return lf_fail(ctx, LF_ERR_STATE, "the cull table is shared through the host (lf_set_cull_share): lf_cull_prepare, the host's "
                                      "all-gather and lf_cull_commit come before lf_trace_ghosts, with the same inputs");
  ctx->cull_share_nb = shared ? a.share_nb : 0;
  ctx->cull_share_n_resident = a.share_n;
  if (entries > ctx->cull_cap) {
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cull_dev) (void)hipFree(ctx->cull_dev);
    ctx->cull_dev = nullptr; ctx->cull_cap = 0; ctx->cull_hash = 0;
    LF_HIP(ctx, hipMalloc((void**)&ctx->cull_dev, entries * sizeof(unsigned long long)));
    ctx->cull_cap = entries;
  }
  if (!ctx->cull_counts) LF_HIP(ctx, hipMalloc((void**)&ctx->cull_counts, 8 * kCullMaxPaths * sizeof(unsigned)));
  ctx->cull_hash = 0;
  unsigned long long* stats_dev = nullptr;
#ifdef LF_EXPERIMENTS
  if (std::getenv("LF_CULL_STATS")) LF_HIP(ctx, hipMalloc((void**)&stats_dev, 32 * sizeof(unsigned long long)));
#endif
  hipEvent_t ev = lf_timing_begin(ctx, LFK_CULL);
  LF_HIP(ctx, hipMemsetAsync(ctx->cull_dev, 0, entries * sizeof(unsigned long long), ctx->stream));
  LF_HIP(ctx, hipMemsetAsync(ctx->cull_counts, 0, 8 * kCullMaxPaths * sizeof(unsigned), ctx->stream));
  unsigned max_items = 0;     // of the level about to run (per path); level 0 runs every box
  for (int lv = 0; lv < n_levels; lv++) {
    a.P = levels[lv];
    a.last = lv + 1 == n_levels ? 1 : 0;
    // The ball and the dispersion slack grow with the level (a coarse box is more curved than 15 rays show); the zonotope's
    // generators are inflated by the same factor at every level: lowered one level at a time, each loses its first lit ray
    // between x 0.9 and x 1.0 (a zonotope is EXACT for the linear part of the map, the measured slack covers the rest:
    // profiles/r05_march_variants.txt)
    a.margin = R.margin * (a.P >= 64 ? 1.0f : a.P >= 32 ? 1.15f : a.P >= 16 ? 1.4f : 2.0f);
    a.geo_margin = R.margin;
    const size_t n_mine = (nblk + (size_t)a.share_n - 1 - (size_t)a.share_rank) / (size_t)a.share_n;   // blocks this rank builds
    const size_t n_items = lv == 0 ? n_mine * (size_t)a.P * a.P : (size_t)max_items;
    if (n_items == 0) break;
    const unsigned* items = lv == 0 ? nullptr : ctx->cull_list[(lv - 1) & 1];
    const unsigned in_stride = a.list_stride;    // (of the list being read: set when it was written)
    unsigned* next = nullptr;
    unsigned out_stride = 0;
    if (!a.last) {
      // every box may keep its four children
      const size_t need = n_items * 4;
      if (need > 0x7fffffffull) return lf_fail(ctx, LF_ERR_INVALID, "cull pre-pass: frame too large");
      out_stride = (unsigned)need;
      const size_t total = need * (size_t)a.n_paths;
      const int slot = lv & 1;
      if (total > ctx->cull_list_cap[slot]) {
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->cull_list[slot]) (void)hipFree(ctx->cull_list[slot]);
        ctx->cull_list[slot] = nullptr; ctx->cull_list_cap[slot] = 0;
        LF_HIP(ctx, hipMalloc((void**)&ctx->cull_list[slot], total * sizeof(unsigned)));
        ctx->cull_list_cap[slot] = total;
      }
      next = ctx->cull_list[slot];
    }
    if (stats_dev) LF_HIP(ctx, hipMemsetAsync(stats_dev, 0, 32 * sizeof(unsigned long long), ctx->stream));
    std::chrono::steady_clock::time_point t_lv;
    if (stats_dev) { LF_HIP(ctx, hipStreamSynchronize(ctx->stream)); t_lv = std::chrono::steady_clock::now(); }
    // the kernel reads its input with the stride it was written with and writes with the new one
    CullLevelArgs k = a;
    k.list_stride = a.last ? in_stride : out_stride;

    // (two strides are needed when reading AND writing: the input's travels in `items_stride`)
    const dim3 grid((unsigned)((n_items + LF_CULL_WG - 1) / LF_CULL_WG), (unsigned)a.n_paths);
#define LF_LAUNCH_LEVEL(KERNEL)                                                                                                     \
    hipLaunchKernelGGL(KERNEL, grid, dim3(LF_CULL_WG), 0, ctx->stream, ctx->lens_dev, ctx->pairs_dev,                                 \
                       (const int*)(ctx->prog_dev + ctx->prog_seq_off), (const LfProgRow*)(ctx->prog_dev + ctx->prog_rec_off),        \
                       k, items, lv == 0 ? nullptr : ctx->cull_counts + (size_t)(lv - 1) * kCullMaxPaths, in_stride, next,           \
                       ctx->cull_counts + (size_t)lv * kCullMaxPaths, ctx->cull_dev, stats_dev)
    if (ctx->cull_rules_custom) LF_LAUNCH_LEVEL(k_cull_level_general); else LF_LAUNCH_LEVEL(k_cull_level);
#undef LF_LAUNCH_LEVEL
    LF_HIP(ctx, hipGetLastError());
    a.list_stride = out_stride;
    if (stats_dev) {   // experiments only: why the boxes of this level ended as they did
      unsigned long long hs[32];
      LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
      std::fprintf(stderr, "CULL_LEVEL P %d items_per_path_max %zu ms %.3f\n", a.P, n_items,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lv).count());
      LF_HIP(ctx, hipMemcpy(hs, stats_dev, sizeof(hs), hipMemcpyDeviceToHost));
      static const char* names[8] = {"", "kept_inside_lobe_partial_box", "kept_too_few_samples_left", "kept_inside_lobe",
                                     "culled_aperture", "culled_mask", "culled_lobe", "culled_all_samples_lost"};
      for (int w = 1; w < 8; w++) std::fprintf(stderr, "CULL_STATS P %d %s %llu\n", a.P, names[w], hs[w]);
    }
    if (!a.last) {
      // how long the next level's lists are: the grid needs the longest
      unsigned cnt[kCullMaxPaths];
      LF_HIP(ctx, hipMemcpyAsync(cnt, ctx->cull_counts + (size_t)lv * kCullMaxPaths, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
      LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
      max_items = 0;
      for (int q = 0; q < a.n_paths; q++) max_items = std::max(max_items, std::min(cnt[q], out_stride));
    }
  }
  // (comm_force_exchange: tests only -- the collective also with a single rank, as lf_comm_gather does)
  ctx->cull_own_rows_only = shared && ctx->cull_share_how == 3;
  if (ctx->cull_share_how == 1 && (shared || ctx->comm_force_exchange)) {
    // every rank has built its slab: one in-place all-gather completes the table everywhere
    const lf_status st = lf_comm_allgather_u64_inplace(ctx, ctx->cull_dev, (size_t)a.share_nb * row_entries);
    if (st != LF_OK) return st;
  }
  lf_timing_end(ctx, LFK_CULL, ev);
  if (stats_dev) (void)hipFree(stats_dev);
  if (ctx->cull_prepare_only) {       // the host's exchange is outstanding: lf_cull_commit finishes
    ctx->cull_hash_pending = h;
    return LF_OK;
  }
  return lfk_cull_finish(ctx, h);
}

constexpr int kMarchTailTilesMax = 4096;
lf_status lfk_march_culled(lf_ctx* ctx, const MarchArgs& a_in, size_t blocks, size_t dyn_lds) {
  const LfApertureDev& m = ctx->ap[LF_APERTURE_STARBURST];
  MarchArgs a = a_in;
  LfCullArgs c;
  c.table = ctx->cull_dev; c.blocks_x = ctx->cull_bx; c.blocks_y = ctx->cull_by; c.cells = ctx->cull_cells;
  c.share_n = ctx->cull_share_nb > 0 ? ctx->cull_share_n_resident : 1; c.share_nb = ctx->cull_share_nb;
  c.blk_log2 = ctx->cull_blk_log2;
  c.multi = ctx->cull_blk_log2 < 3 + a.xs ? 1 : 0;
  // the started paths' common leg is marched once (march_started_set) if a higher path index never leaves it later:
  // the primary path in front, the pairs by ascending first mirror -- the order lf_set_ghost_pairs(NULL) and any sorted list give
  c.prefix_ok = 1;
  for (int q = 1; q < ctx->pairs.n; q++) {
    const int ia = ctx->pairs.ij[q - 1][0], ib = ctx->pairs.ij[q][0];
    if (ib < 0 || (ia >= 0 && ib < ia)) c.prefix_ok = 0;
  }
  if (ctx->cull_no_prefix) c.prefix_ok = 0;
  c.P = ctx->cull_P; c.m = ctx->cull_m; c.m_shift = ctx->cull_m == 4 ? 2 : ctx->cull_m == 2 ? 1 : 0;
  hipEvent_t ev = lf_timing_begin(ctx, LFK_MARCH);
#define LF_LAUNCH_CULL1(KK, WW, SS)                                                                           \
  hipLaunchKernelGGL((k_march_cull<KK, WW, SS>), dim3((unsigned)blocks), dim3(64 * kWgWaves), dyn_lds, ctx->stream, ctx->lens_dev, \
                     ctx->pairs_dev, (const int*)(ctx->prog_dev + ctx->prog_seq_off),                         \
                     (const LfProgRow*)(ctx->prog_dev + ctx->prog_rec_off),                                  \
                     (const LfWeightRow*)(ctx->prog_dev + ctx->prog_wrec_off), m.texels, a, c, ctx->ghost,   \
                     ctx->accum, ctx->counters_dev)
#define LF_LAUNCH_ITEMS(KK)                                                                                  \
  hipLaunchKernelGGL(k_march_items<KK>, dim3((unsigned)blocks), dim3(64 * kWgWaves), dyn_lds, ctx->stream, ctx->lens_dev, \
                     ctx->pairs_dev, (const int*)(ctx->prog_dev + ctx->prog_seq_off),                         \
                     (const LfProgRow*)(ctx->prog_dev + ctx->prog_rec_off),                                  \
                     (const LfWeightRow*)(ctx->prog_dev + ctx->prog_wrec_off), m.texels, a, c, ctx->ghost,   \
                     ctx->accum, ctx->counters_dev)
#define LF_LAUNCH_CULL(KK) do { if (items) LF_LAUNCH_ITEMS(KK); else if (weights_first) LF_LAUNCH_CULL1(KK, true, 0); \
                                else if (shared_leg) LF_LAUNCH_CULL1(KK, false, 1); else LF_LAUNCH_CULL1(KK, false, 0); } while (0)
  const bool weights_first = ctx->cull_weights_first;   // (lf_test_knob: the weight on every executed event)
  // every pixel its own pupil point (no sub-cells at all): the compacted march.  (2 x 2 sub-cells, where the lanes of a
  // wave still look their cells up one by one, stay with k_march_cull: 48 against 59 ms on the bench frame)
  bool items = ctx->march_sub_bits == 0;
#ifdef LF_EXPERIMENTS
  if (const char* e = std::getenv("LF_CULL_ITEMS")) items = std::atoi(e) != 0;
#endif
  // one table entry per (wave tile, sample) and a selection in order: the started paths' common leg once (march_started_set)
  const bool shared_leg = c.prefix_ok && !c.multi && (1 << a.sub_bits) >= c.m;
  // ---- the tail: a launch drains for as long as its last workgroups run -- 0.65 to 1.7 ms for a tile of the bench frame,
  // during which the resident slots empty one by one (measured per workgroup, wall clock at start and end: half of the 768
  // slots idle over the last 0.7 ms of a 5 ms launch, 1/8 of the frame).  The LAST tiles (half a round of resident
  // workgroups) are therefore split over `tail_groups` workgroups each (samples sg, sg + groups, ...): the launch ends on
  // workgroups a quarter as long.  Whole tiles write their pixels themselves, split ones meet in tail_acc (k_march_cull).
  // 1/8, 1/4, 1/2 of the bench frame: 5.00 -> 4.83, 9.46 -> 9.32, 18.40 -> 18.34 ms; all of it: 36.5 -> 36.4; more groups or a
  // longer tail cost more than they save (a split tile repeats the workgroup's set-up and ends on 8 waves waiting for one)
  // (profiles/r06_cull_bounds.txt, 7.)
  a.tail_from = 0; a.tail_groups = 1; c.tail_acc = nullptr; c.tail_done = nullptr;
  {
    // k_march_items: a (sample, 64 pixels) wave lists about 64 x paths x the table's started fraction items; the chunk to begin
    // with is the power of two of samples that fits the list with a third to spare (blocks differ)
    const double per_sample = 64.0 * (double)std::max(1, ctx->pairs.n) * std::max(1e-4, ctx->cull_started_fraction) * 1.35;
    int ch0 = 256;
    while (ch0 > 1 && (double)ch0 * per_sample > (double)kItemCap) ch0 >>= 1;
    c.items_chunk0 = ch0;
  }
  if (!items && a.sgroups == 1 && !weights_first) {
    hipDeviceProp_t prop;
    LF_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int resident = prop.multiProcessorCount * (ctx->march_k == 1 ? 4 : 3);     // workgroups of 8 waves at 8 / 6 waves per SIMD
    int tail_tiles = ctx->march_tail_tiles >= 0 ? ctx->march_tail_tiles : resident / 2;
    int groups = ctx->march_tail_groups >= 0 ? ctx->march_tail_groups : 4;
    while (groups > 1 && groups * 32 > a.spp) groups /= 2;
    const int tiles_pad = (a.n_tiles + 63) / 64 * 64;
    tail_tiles = std::min(std::min(tail_tiles, kMarchTailTilesMax), tiles_pad) / 64 * 64;
    if (groups > 1 && tail_tiles > 0) {
      if (!ctx->tail_acc) {
        LF_HIP(ctx, hipMalloc((void**)&ctx->tail_acc, (size_t)kMarchTailTilesMax * 192 * sizeof(unsigned long long)));
        LF_HIP(ctx, hipMalloc((void**)&ctx->tail_done, (size_t)kMarchTailTilesMax * sizeof(int)));
        LF_HIP(ctx, hipMemsetAsync(ctx->tail_acc, 0, (size_t)kMarchTailTilesMax * 192 * sizeof(unsigned long long), ctx->stream));
        LF_HIP(ctx, hipMemsetAsync(ctx->tail_done, 0, (size_t)kMarchTailTilesMax * sizeof(int), ctx->stream));
      }
      a.tail_from = tiles_pad - tail_tiles; a.tail_groups = groups;
      c.tail_acc = ctx->tail_acc; c.tail_done = ctx->tail_done;
      blocks = (size_t)a.tail_from + (size_t)tail_tiles * groups;
    }
  }
#ifdef LF_EXPERIMENTS
  c.wg_clock = nullptr;
  const char* clock_file = std::getenv("LF_MARCH_WG_CLOCK");
  if (clock_file) {
    LF_HIP(ctx, hipMalloc((void**)&c.wg_clock, blocks * 16));
    LF_HIP(ctx, hipMemset(c.wg_clock, 0, blocks * 16));
    LF_HIP(ctx, hipDeviceSynchronize());
  }
#endif
  switch (ctx->march_k) {
    case 1: LF_LAUNCH_CULL(1); break;
    case 2: LF_LAUNCH_CULL(2); break;
    default: LF_LAUNCH_CULL(3); break;
  }
#undef LF_LAUNCH_CULL
#undef LF_LAUNCH_CULL1
#undef LF_LAUNCH_ITEMS
  lf_timing_end(ctx, LFK_MARCH, ev);
  LF_HIP(ctx, hipGetLastError());
#ifdef LF_EXPERIMENTS
  if (clock_file) {
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h(blocks * 2);
    LF_HIP(ctx, hipMemcpy(h.data(), c.wg_clock, blocks * 16, hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(clock_file, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    (void)hipFree(c.wg_clock);
  }
#endif
  return LF_OK;
}
