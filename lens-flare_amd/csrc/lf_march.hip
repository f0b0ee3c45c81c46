// lf_march.hip -- the geometric lens march (north-star path; no reference counterpart:
// the reference's lens is paraxial, pathtracer.cpp:511-689, and its thin-lens camera is a stub,
// camera_lens.cpp:22-30).
//
// For every sensor sample (pixel, sample index) a ray is started on the sensor towards a point
// of the rear pupil and marched BACKWARDS (against the light) through the stack of spherical
// interfaces, once per wavelength and per ghost pair (i, j):
//     refract N-1 .. i+1, reflect at i, refract i+1 .. j-1, reflect at j, refract j-1 .. 0
// (N + 2(j-i) surface events; the primary path is N refractions).  Each event = intersection +
// semi-aperture test + Snell refraction or mirror reflection (+ unpolarised Fresnel weight); the
// stop is a flat pass-through interface whose event is the aperture-mask lookup.  A ray that
// leaves the front element collects the sun's radiance through a smooth angular lobe.
//
// Mapping to CDNA4 (DESIGN.md section 3 has the measurements behind each point):
//   * one wave = one 8x8 sensor tile, one lane = one pixel; the wave walks sample index s,
//     wavelength and path together, so the interface sequence is wave-uniform and every
//     per-interface constant comes from the scalar cache into SGPRs (no LDS/VGPR traffic for the
//     lens table at all);
//   * the pupil is stratified (G x G cells, G = floor(sqrt(spp)), 4 x 4 sub-cells per tile and
//     sample): the 64 lanes of a wave cross the stop in the same small patch of the aperture mask
//     and share their fate; a wave whose lanes are all dead skips what only they would visit;
//   * the paths of one (sample, wavelength) are walked as ONE tree (build_program): the backward
//     leg from the sensor and the forward leg after the reflection at i are computed once for all
//     the pairs that share them; forks park the ray state in LDS;
//   * the first pass carries geometry only; the Fresnel / aperture weight is computed by
//     re-marching the one path, and only for waves in which a lane reaches the sun's lobe;
//   * the 8 waves of a workgroup share the tile and pull sample indices from an LDS counter;
//     per-pixel sums are 64-bit fixed point, added straight into LDS by the rare lit lanes, written once.
//
// Arithmetic contract (DESIGN.md "march arithmetic"): the ray's direction is carried as optical
// direction cosines K = n d (round 3); float32, every multiply-add written as an
// explicit fmaf, IEEE-correct division (__fdiv_rn), the hardware's v_sqrt_f32 (lf_sqrt; the oracle
// follows it through a measured deviation table), no libm.  Contributions are accumulated as 2^-36
// fixed point in 64-bit integers, so the result does not depend on the order in which lanes
// finish.  The CPU oracle (oracle/lf_geo_oracle.c) follows the same contract and marches every
// path on its own, which makes pixels and event counters comparable bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <utility>

#include "lf_internal.h"
#include "lf_march_events.h"
#include "lf_march_common.h"

namespace {

using namespace lfm;


// Table rows are read with ONE wide scalar load each (left to itself the compiler sinks the field
// loads into the branches that use them: 4-5 dependent scalar-cache round trips per event), through
// the CONSTANT address space: such a load can never be clobbered by the kernel's own stores, so it
// always qualifies for the scalar unit and may be scheduled freely (load_phdr / load_prec below).

// parked ray state (a fork of the path tree) in LDS: the two slots would cost 12 VGPRs and the 8th
// wave of every SIMD
__device__ __forceinline__ void park(float2* __restrict__ slot, int lane, const Ray& r) {
  slot[lane] = make_float2(r.px, r.py);          // [pair][lane]: 8-byte lane stride, conflict-free
  slot[64 + lane] = make_float2(r.hz, r.dx);
  slot[128 + lane] = make_float2(r.dy, r.dz);
}
__device__ __forceinline__ void unpark(const float2* __restrict__ slot, int lane, Ray& r) {
  const float2 a = slot[lane], b = slot[64 + lane], c = slot[128 + lane];
  r.px = a.x; r.py = a.y; r.hz = b.x; r.dx = b.y; r.dy = c.x; r.dz = c.y;
  r.r2 = fmaf(r.px, r.px, r.py * r.py);   // the same expression that produced it: the same bits
}

// K = rays per lane: the K wavelengths of a group walk the program TOGETHER.  They start as the same
// ray and differ only by dispersion, so they share their fate almost always -- and the whole scalar
// side of the walk (row load, dispatch on the row kind, loop control, fork/join, dead-wave jumps) is
// paid once for K events.  That side was the binding resource of the one-ray walk: 7.6e10 scalar
// instructions per bench frame on ONE scalar unit per CU (85 % busy) against 9.5e10 vector
// instructions on four SIMDs (53 %); profiles/r02_*.  Each ray keeps its own liveness mask, tallies
// are per ray, so pixels and counters are exactly those of K separate walks.
// (second launch bound = waves per SIMD the register allocation must leave room for; the LDS
// footprint allows at least as many waves per CU: 32 / 24 / 24 for K = 1 .. 3.  A 7th
// wave for K = 3 is not to be had: with the LDS made to fit (5 KB per wave) the allocator, capped at 72
// registers, adds 4 % of vector instructions and the frame takes 123 instead of 117 ms; occupancy sweep
// 3 .. 7 waves per SIMD: 155, 133, 122, 117, 123 ms, profiles/r03_march_variants.txt)
// A workgroup = one 8 x 8 tile marched by kWgWaves waves that pull its samples from one LDS counter.  8
// waves rather than 4: a tile is finished in half the time, so half as many tiles are in flight when the
// grid runs dry and the launch's tail -- the last workgroups running on a part-empty GPU -- is half as
// long, at the same 24 waves per CU (3 workgroups of 50.8 KB LDS); c3 frame 115.5 -> 114.9 ms, shares of
// a frame 1-2 % (profiles/r03_march_variants.txt; 6 and 2 waves are worse, 12 no better).
constexpr int kWgWaves = 8;
#ifdef LF_MARCH_LIT_MAP
// instrumented build (profiles/r05_cull_floor.json), never shipped: which (sensor block, pupil cell at resolution P,
// path) combinations EVER end with a lane inside the lobe pre-test -- the floor of any table-driven cull at that
// granularity.  Eight maps: blocks of 64 x 64 and of 64 x 8 pixels, P = 16, 32, 64, 128 cells per pupil axis.
__device__ unsigned long long* g_lit_map[8];
#endif
#ifndef LF_SKIP_DEAD_LAMBDA
#define LF_SKIP_DEAD_LAMBDA 1   // experiments (profiles/r04_march_variants.txt): 0 = no per-wavelength liveness branch
#endif
template <int K>
__global__ __launch_bounds__(64 * kWgWaves, (K == 1 ? 8 : 6))
void k_march(const LfLensDev* __restrict__ lens, const LfPairsDev* __restrict__ pairs,
             const int* __restrict__ seq_table, const LfProgHdr* __restrict__ hdr_table,
             const LfProgRow* __restrict__ rec_table, const LfWeightRow* __restrict__ wrec_table,
             const float* __restrict__ mask, MarchArgs a,
             double* __restrict__ ghost, unsigned long long* __restrict__ accum,
             unsigned long long* __restrict__ counters) {
#ifdef LF_MARCH_ALL_WEIGHTS
  // ablation build (profiles/r03_all_weights_ablation.json), never shipped, TIMING ONLY: every event of
  // the first pass also evaluates its Fresnel / aperture weight -- the event of SURVEY 8d -- and a path's
  // end uses the carried weight instead of marching the path again (fork states do not carry the
  // weight here, so pixels are wrong; events, fates and the instruction mix are the real thing)
  constexpr bool kW1 = true;
  // ... with the row's real Fresnel scale factors, fetched beside the row like the re-march fetches them
#define LF_WROW_INIT LfWeightRow curw = load_wrec(wrecs, (unsigned)hdr.rec);
#define LF_WROW_LOAD(rec) curw = load_wrec(wrecs, (rec))
#define LF_WROW_ARGS(j) , curw.fs[j], curw.fo[j], curw.fi[j]
#else
  constexpr bool kW1 = false;
#define LF_WROW_INIT
#define LF_WROW_LOAD(rec) do { } while (0)
#define LF_WROW_ARGS(j)
#endif
  __shared__ unsigned long long s_acc[64 * 3];
  __shared__ unsigned long long s_cnt[kMarchCounters];
  __shared__ int s_next;
#ifdef LF_MARCH_LIVE_HIST
  // instrumented build (profiles/r03_march_variants.txt), never shipped: executed wave-ray events by
  // the number of live lanes, bucket = lanes / 8 (8 = all 64)
  // ... per row kind (round 5): 0 = refraction at a curved interface (the rows of a run), 1 = mirror / flat glass, 2 = the stop
  __shared__ unsigned long long s_hist[3 * 9];
  if (threadIdx.x < 27) s_hist[threadIdx.x] = 0ull;
#define LF_HIST(kind, mask) do { if ((mask) != 0ull && lane_id() == 0) atomicAdd(&s_hist[(kind) * 9 + (__popcll(mask) >> 3)], 1ull); } while (0)
#else
#define LF_HIST(kind, mask) do { } while (0)
#endif
#ifdef LF_MARCH_PAIR_STATS
  // instrumented build (profiles/r05_pair_table.json), never shipped: per path q, wave-rays (wave x wavelength)
  // that complete it with a live lane, those with a lane inside the sun's lobe pre-test, and those lanes
  __shared__ unsigned long long s_pair[3 * 64];
  if (threadIdx.x < 192) s_pair[threadIdx.x] = 0ull;
#define LF_PAIR_STAT(which, q, n) do { if ((n) != 0u && lane_id() == 0) atomicAdd(&s_pair[(which) * 64 + ((q) & 63)], (unsigned long long)(n)); } while (0)
#else
#define LF_PAIR_STAT(which, q, n) do { } while (0)
#endif
  // parked ray states: [wave][slot][ray][px py|pz dx|dy dz][lane]
  // fork slot 1 (the reflection at j, parked and restored once per pair) is parked here; slot 0 (the
  // reflection at i, once per sub-tree: 3-4x rarer, so its register copies are cheap) lives in
  // registers: with both in LDS a K = 3 wave needs 9.5 KB and only 4 waves fit a SIMD
  // (measured 147 -> 137 ms per bench frame)
  __shared__ float2 s_state[kWgWaves][K][3 * 64];
  // the start of the current sample's rays per lane (sensor point, direction, start weight): only
  // the head of each wavelength group and the rare weight re-march (~2.4x per sample) read it back,
  // so it must not occupy six registers during the walk.  1.5 KB of LDS per wave.  Round 3 tried the two other
  // homes (profiles/r03_march_variants.txt): RECOMPUTING it where a ray starts (two Philox draws + the
  // pupil map, ~250 instructions, 2.4x per sample) costs 7 % of the frame; a per-wave slice of a GLOBAL
  // scratch buffer is 1.2 % faster than LDS, but a seventh to a quarter of its stores leave the L2 for
  // the fabric -- 1.8 to 3.2 GB of write-back per frame against 50.8 MB of algorithmic traffic.
  __shared__ float s_start[kWgWaves][6][64];
#define LF_START(k, l) s_start[wave][k][l]
  // (read back through a lane index the compiler cannot see through -- launder() -- or it forwards the
  // stores to the loads and keeps the registers)
  auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
  const int tid = threadIdx.x;
  if (tid < 64 * 3) s_acc[tid] = 0ull;
  if (tid < kMarchCounters) s_cnt[tid] = 0ull;
  if (tid == 0) s_next = 0;
  __syncthreads();

  // a wave's 64 pixels: 8 rows x 8 columns that are 2^xs apart; 2^xs such waves interleave in a block of
  // 8 * 2^xs columns.  tx counts waves along x: block (tx >> xs), phase (tx & (2^xs - 1)).
  const int tiles_x = ((a.W + (8 << a.xs) - 1) >> (3 + a.xs)) << a.xs;
  // Workgroup -> tile, XCD-aware: blocks b and b + 8 share an XCD (its L2), and the 8 wave tiles that
  // interleave in one 64-column block write 24-byte pixels that alternate inside the same cache lines.  Swapping
  // the two 3-bit fields of the slot index puts those 8 tiles on ONE XCD, a few dispatch slots apart, so that
  // their partial lines meet in that XCD's L2 and leave it whole (without it the launch wrote 100 MB for its
  // 50 MB of pixels: profiles/r04_march_variants.txt).  The slot range is padded to a multiple of 64.
  const int sg = blockIdx.x % a.sgroups;
  const unsigned slot = blockIdx.x / a.sgroups;
  const int tile_lin = (int)((slot & ~63u) | ((slot & 7u) << 3) | ((slot >> 3) & 7u));
  if (tile_lin >= a.n_tiles) return;   // (the whole workgroup, before any barrier)
  int tx, trow;
  march_tile_of(a, tile_lin, tiles_x, tx, trow);
  const unsigned tile_id = (unsigned)(trow * tiles_x + tx);  // frame-absolute
  // Everything that depends on the lane (pixel coordinates, LDS addresses) is re-derived from the
  // thread id where it is used: held across the walk it costs ~10 registers, which at 6 waves per
  // SIMD the allocator can only find in scratch memory (430 MB of spill traffic per bench frame).
  // The volatile asm keeps the compiler from hoisting the derivation back out of the loop.
  // (the lane id comes from mbcnt, the wave id sits in an SGPR: not even the thread id is held)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto lane_id = [&]() {
    int t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    return t;
  };
  lanemask active_mask;
  {
    const int lane = tid & 63;
    const int x = ((tx >> a.xs) << (3 + a.xs)) + ((lane & 7) << a.xs) + (tx & ((1 << a.xs) - 1)), y = trow * 8 + (lane >> 3);
    active_mask = __ballot(x < a.W && y >= a.y0 && y < a.y1);
  }

  const int n_lambda = lens->n_lambda, n_pairs = pairs->n;
  const int prog_rows = pairs->prog_rows, prog_recs = pairs->prog_recs;
  const int n_groups = (n_lambda + K - 1) / K;
  const float pitch = lens->pitch, pupil_h = lens->pupil_h;
  const float geom_norm = lens->geom_norm;
  // (wave-uniform floats the device would have to compute on the vector unit -- and then hold in, or
  // spill from, vector registers -- arrive as kernel arguments: IEEE operations, the same on the host)
  const float inv_stop_h = a.inv_stop_h, half_w = a.half_w, half_h = a.half_h, vz_u = a.vz;
  const float sx = lens->sun_dir[0], sy = lens->sun_dir[1], sz = lens->sun_dir[2];
  const float inv_1mc = lens->sun_inv_one_minus_cos, sun_ss = lens->sun_ss;
  const float lobe_thr = a.lobe_thr;
  const int GG = a.G * a.G;

  unsigned n_light = 0;     // per lane
  unsigned n_samples = 0;   // per wave
  // wave-uniform, counted once per wave with s_bcnt1 (SALU)
  unsigned long long events = 0, n_clip = 0, n_vign = 0, n_tir = 0, n_scene = 0, n_exec = 0;
  unsigned long long n_rm_lane = 0, n_rm_rows = 0;   // the weight re-march (diagnostics, lf_get_march_stats)

  {
    // The waves of the workgroup pull sample indices from one LDS counter instead of owning
    // every kWgWaves-th one: a wave whose pupil cells fall outside the aperture finishes its samples in a
    // fraction of the time, and a static split leaves it idle until the slowest wave is done.
    // (Which wave marches which sample does not matter: the sums are integers.)
    for (;;) {
      int k = 0;
      if (lane_id() == 0) k = atomicAdd(&s_next, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      const int s = sg + k * a.sgroups;  // wave-uniform
      if (s >= a.spp) break;
      const int lane = lane_id();
      const int x = ((tx >> a.xs) << (3 + a.xs)) + ((lane & 7) << a.xs) + (tx & ((1 << a.xs) - 1)), y = trow * 8 + (lane >> 3);
      const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
      // ---- sensor sample -> initial ray --------------------------------------------------
      const uint4 rnd = philox4x32_10(make_uint4(p, (unsigned)s, kDomainMarch, 0u), a.key);
      const float jx = u01(rnd.x), jy = u01(rnd.y);
      float ua = u01(rnd.z), ub = u01(rnd.w);
#ifdef LF_MARCH_LIT_MAP
      float lm_u = 0.5f, lm_v = 0.5f;   // the wave's sub-cell centre in the pupil square (wave-uniform)
#endif
      if (s < GG) {  // stratum (s % G, s / G) of the pupil square
        const int cy = s / a.G, cx = s - cy * a.G;
        // ... and inside it ONE of sub x sub sub-cells, drawn per (tile, s): wave-uniform, so the
        // compiler keeps this Philox on the scalar unit.  Each pixel still covers its stratum
        // uniformly (the sub-cell is uniformly random), but the 64 lanes of the wave now cross the
        // stop within 1/(G*sub) of its width and share their fate at the mask even more often.
        const uint4 r2 = philox4x32_10(make_uint4(tile_id, (unsigned)s, kDomainSubcell, 0u), a.key);
        const unsigned sxi = a.sub_bits ? (r2.x >> (32 - a.sub_bits)) : 0u;
        const unsigned syi = a.sub_bits ? (r2.y >> (32 - a.sub_bits)) : 0u;
        ua = ((float)cx + ((float)sxi + ua) * a.inv_sub) * a.inv_G;
        ub = ((float)cy + ((float)syi + ub) * a.inv_sub) * a.inv_G;
#ifdef LF_MARCH_LIT_MAP
        lm_u = ((float)cx + ((float)sxi + 0.5f) * a.inv_sub) * a.inv_G;
        lm_v = ((float)cy + ((float)syi + 0.5f) * a.inv_sub) * a.inv_G;
#endif
      }
      const float pa = fmaf(2.0f, ua, -1.0f), pb = fmaf(2.0f, ub, -1.0f);
      const float X = -(((float)x + jx) - half_w) * pitch;
      const float Y = -(((float)y + jy) - half_h) * pitch;
      // concentric square -> disc map; sin/cos of (pi/4)*t by fixed polynomials (fmaf only)
      float qx = 0.0f, qy = 0.0f;
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
      const float vx = fmaf(pupil_h, qx, -X), vy = fmaf(pupil_h, qy, -Y), vz = vz_u;
      const float len = lf_sqrt(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
      const float rl = lf_rcp(len);
      const float d0x = vx * rl, d0y = vy * rl, d0z = vz * rl;
      const float c2 = d0z * d0z;
      const float w0 = geom_norm * (c2 * c2);
      LF_START(0, lane) = X; LF_START(1, lane) = Y; LF_START(2, lane) = d0x;
      LF_START(3, lane) = d0y; LF_START(4, lane) = d0z; LF_START(5, lane) = w0;
      n_samples++;
      // wave-uniform 32-bit tallies of this sample (64 lanes x pairs x wavelengths x rows < 2^32)
      unsigned ev32 = 0, clip32 = 0, vign32 = 0, tir32 = 0, scene32 = 0, exec32 = 0;

      for (int g = 0; g < n_groups; g++) {
        // ---- walk the group's program (the tree of all paths, depth first) ---------------------
        const LfProgHdr* const prog = hdr_table + (size_t)g * (size_t)(prog_rows + 1);   // (+ a spare header)
        const LfProgRow* const recs = rec_table + (size_t)g * (size_t)prog_recs;
        const LfWeightRow* const wrecs = wrec_table + (size_t)g * (size_t)prog_recs;
        constexpr unsigned kHdr = (unsigned)sizeof(LfProgHdr);
        const unsigned prog_end = (unsigned)prog_rows * kHdr;
        unsigned e = 0u;   // byte offset of the current row's header
        Ray r[K];
        lanemask alive[K], alive0[K], alive1[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
          // (read back from LDS rather than kept in registers across the groups of the sample)
          const int lo = launder(lane);
          r[j] = Ray{LF_START(0, lo), LF_START(1, lo), 0.0f, 0.0f, LF_START(2, lo),
                     LF_START(3, lo), LF_START(4, lo), kW1 ? LF_START(5, lo) : 0.0f, kW1 ? 1.0f : 0.0f};
          r[j].r2 = fmaf(r[j].px, r[j].px, r[j].py * r[j].py);
          // K = n d: the index of the medium between the last interface and the sensor (air: exact)
          const float ns = lens->n_start[min(g * K + j, n_lambda - 1)];
          r[j].dx *= ns; r[j].dy *= ns; r[j].dz *= ns;
          alive[j] = (g * K + j < n_lambda) ? active_mask : 0ull;  // a short last group: dead rays
          alive0[j] = 0ull; alive1[j] = 0ull;
        }
        // live rays of the group (sum of the K masks' populations), kept up to date where rays end and
        // where fork states come back, so that no row has to count its masks
        unsigned nlive = 0u, nlive0 = 0u, nlive1 = 0u;
#pragma unroll
        for (int j = 0; j < K; j++) nlive += (unsigned)__popcll(alive[j]);
        float r0[K][6];   // fork slot 0: px py hz dx dy dz (r2 is re-derived, like after an LDS restore)
        auto park_all = [&](int slot, lanemask* keep) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (slot == 0) {
              r0[j][0] = r[j].px; r0[j][1] = r[j].py; r0[j][2] = r[j].hz;
              r0[j][3] = r[j].dx; r0[j][4] = r[j].dy; r0[j][5] = r[j].dz;
            } else {
              park(s_state[wave][j], lane, r[j]);
            }
            keep[j] = alive[j];
          }
          if (slot == 0) nlive0 = nlive; else nlive1 = nlive;
        };
        auto unpark_all = [&](int slot, const lanemask* keep) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            if (slot == 0) {
              r[j].px = r0[j][0]; r[j].py = r0[j][1]; r[j].hz = r0[j][2];
              r[j].dx = r0[j][3]; r[j].dy = r0[j][4]; r[j].dz = r0[j][5];
              r[j].r2 = fmaf(r[j].px, r[j].px, r[j].py * r[j].py);
            } else {
              unpark(s_state[wave][j], lane, r[j]);
            }
            alive[j] = keep[j];
          }
          nlive = slot == 0 ? nlive0 : nlive1;
        };
        // `cur` always holds the row at e: whoever moves e loads the row it lands on, so a run's
        // last iteration has already fetched the row the dispatch below looks at next
        LfProgHdr hdr = load_phdr(prog, e);
        LfProgRow cur = load_prec(recs, (unsigned)hdr.rec);
        LF_WROW_INIT
        // the next row: its header and -- named by the current header -- its record, issued together
        auto step = [&]() {
          const unsigned rn = (unsigned)hdr.rec_next;
          e += kHdr;
          hdr = load_phdr(prog, e);
          cur = load_prec(recs, rn);
          LF_WROW_LOAD(rn);
        };
        while (e != prog_end) {
          const unsigned fl = (unsigned)hdr.flags;
          const unsigned run = (fl >> 8) & 0xffu, mult = (fl >> 16) & 0xffu;
          unsigned endfl = 0u;  // flags of the row just executed if it completes a path
          bool dead = false;    // no ray of the group is alive any more
          int sk = 0;           // ... then: the jump-table entry of the row they died at
          if (run) {
            // A run: plain events (refraction at a curved interface), possibly led by a curved
            // mirror (the fork of a sub-tree or of one pair); straight-line bodies.  The last row of
            // a run may complete a path.  Events are tallied per RUN, not per row: a run of n rows
            // entered by m live rays completes n * m events minus, for every ray that ends at a row
            // with k rows of the run left (that row included), k -- so the common row costs no
            // scalar tally work at all; the rare death branch does the arithmetic.
            // (the __builtin_expect hints keep the rare blocks -- a ray ends, a wavelength is gone -- out
            // of the straight-line path of the walk; measured together with the 32-bit row offset:
            // 119.4 -> 117.75 ms per bench frame, profiles/r02_march_variants.txt)
            unsigned k = run, lost = 0u;
            const unsigned live0 = nlive;
            if (fl & LF_EV_REFLECT) {
              if (fl & LF_EV_SAVE0) park_all(0, alive0);
              if (fl & LF_EV_SAVE1) park_all(1, alive1);
              lanemask okv[K], died = 0ull;
#pragma unroll
              for (int j = 0; j < K; j++) {
                if (LF_SKIP_DEAD_LAMBDA && K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; continue; }
                lanemask geom_ok;
                LF_HIST(1, alive[j]);
                okv[j] = surface_event<kW1>(r[j], cur.dzv, cur.curv, cur.ch, cur.c2, cur.sc, cur.cn22[j], cur.rn2[j],
                                            cur.delta[j], cur.h2, true, false, cur.sgn, geom_ok LF_WROW_ARGS(j));
                died |= alive[j] & ~okv[j];
              }
              if (__builtin_expect(died != 0ull, 0)) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                  const unsigned nd = (unsigned)__popcll(alive[j] & ~okv[j]);
                  vign32 += mult * nd;
                  lost += nd * k;
                  nlive -= nd;
                  alive[j] &= okv[j];
                }
                if (nlive == 0u) { dead = true; k = 1u; sk = hdr.skip; }
              }
              endfl = fl;
              --k;
              step();
            }
            while (k != 0u) {
              lanemask okv[K], gv[K], died = 0ull;
              // (a variant that runs the K events without the per-wavelength checks while every
              // wavelength is live, and this one otherwise, was SLOWER: 128 vs 121 ms -- the second copy
              // of the loop body costs more in instruction fetch than the three scalar branches)
              {
#pragma unroll
                for (int j = 0; j < K; j++) {
                  // a wavelength whose rays are all gone is not computed (one scalar branch; without
                  // it its lanes would keep marching garbage through every row the others still visit)
                  if (LF_SKIP_DEAD_LAMBDA && K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
                  LF_HIST(0, alive[j]);
                  okv[j] = surface_event<kW1>(r[j], cur.dzv, cur.curv, cur.ch, cur.c2, cur.sc, cur.cn22[j], cur.rn2[j],
                                              cur.delta[j], cur.h2, false, false, cur.sgn, gv[j] LF_WROW_ARGS(j));
                  died |= alive[j] & ~okv[j];
                }
              }
              endfl = (unsigned)hdr.flags;
              if (__builtin_expect(died != 0ull, 0)) {  // some ray ends here, in `mult` logical paths
#pragma unroll
                for (int j = 0; j < K; j++) {
                  vign32 += mult * (unsigned)__popcll(alive[j] & ~gv[j]);
                  tir32 += mult * (unsigned)__popcll(alive[j] & gv[j] & ~okv[j]);
                  const unsigned nd = (unsigned)__popcll(alive[j] & ~okv[j]);
                  lost += nd * k;
                  nlive -= nd;
                  alive[j] &= okv[j];
                }
                if (nlive == 0u) { dead = true; k = 1u; sk = hdr.skip; }
              }
              --k;
              step();   // (the table ends with a spare row)
            }
            const unsigned live_sum = run * live0 - lost;
            ev32 += mult * live_sum;   // logical events: one per path that shares these rows
            exec32 += live_sum;        // computed events
          } else if (fl & LF_EV_STOP) {
            lanemask okv[K], died = 0ull;
#pragma unroll
            for (int j = 0; j < K; j++) {
              if (LF_SKIP_DEAD_LAMBDA && K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; continue; }
              LF_HIST(2, alive[j]);
              okv[j] = stop_event<kW1>(r[j], cur.dzv, cur.h2, inv_stop_h, mask, a.mw, a.mh);
              died |= alive[j] & ~okv[j];
            }
            if (__builtin_expect(died != 0ull, 0)) {
#pragma unroll
              for (int j = 0; j < K; j++) {
                const unsigned nd = (unsigned)__popcll(alive[j] & ~okv[j]);
                clip32 += mult * nd;
                nlive -= nd;
                alive[j] &= okv[j];
              }
              dead = nlive == 0u;
            }
            const unsigned live = nlive;
            ev32 += mult * live;
            exec32 += live;
            endfl = fl;
            sk = hdr.skip;
            step();
          } else {
            // a single mirror event (a fork that ends its leg at once) or flat glass
            if (fl & LF_EV_SAVE0) park_all(0, alive0);
            if (fl & LF_EV_SAVE1) park_all(1, alive1);
            lanemask okv[K], gv[K], died = 0ull;
#pragma unroll
            for (int j = 0; j < K; j++) {
              if (LF_SKIP_DEAD_LAMBDA && K > 1 && __builtin_expect(alive[j] == 0ull, 0)) { okv[j] = 0ull; gv[j] = 0ull; continue; }
              LF_HIST(1, alive[j]);
              okv[j] = surface_event<kW1>(r[j], cur.dzv, cur.curv, cur.ch, cur.c2, cur.sc, cur.cn22[j], cur.rn2[j],
                                          cur.delta[j], cur.h2, (fl & LF_EV_REFLECT) != 0,
                                          (fl & LF_EV_FLAT) != 0, cur.sgn, gv[j] LF_WROW_ARGS(j));
              died |= alive[j] & ~okv[j];
            }
            if (__builtin_expect(died != 0ull, 0)) {
#pragma unroll
              for (int j = 0; j < K; j++) {
                vign32 += mult * (unsigned)__popcll(alive[j] & ~gv[j]);
                tir32 += mult * (unsigned)__popcll(alive[j] & gv[j] & ~okv[j]);
                nlive -= (unsigned)__popcll(alive[j] & ~okv[j]);
                alive[j] &= okv[j];
              }
              dead = nlive == 0u;
            }
            const unsigned live = nlive;
            ev32 += mult * live;
            exec32 += live;
            endfl = fl;
            sk = hdr.skip;
            step();
          }
          if (dead) {
            // every ray of the wave is dead: jump over everything only these rays would still visit
            // (the jump-table entry travels in the row itself: no dependent load in front of the next row)
            e = e - kHdr + (((unsigned)sk & ~3u) << 2);  // (sk >> 2) rows of 16 bytes
            hdr = load_phdr(prog, e);
            cur = load_prec(recs, (unsigned)hdr.rec);
            LF_WROW_LOAD((unsigned)hdr.rec);
            if ((sk & 3) == 1) unpark_all(1, alive1);
            else if ((sk & 3) == 2) unpark_all(0, alive0);
          } else if (endfl & LF_EV_END) {
            // ---- a path is complete ---------------------------------------------------------
            scene32 += nlive;
            // inside the sun's lobe?  (cheap pre-test on d.s alone: 1 - d.s cancels -- its absolute
            // error of ~1e-7 is ~1e-4 of a 0.05 rad lobe -- so it only SELECTS, against a threshold
            // with a 1/16 margin computed once on the host; the lobe factor itself is evaluated
            // without cancellation after the weight re-march, see lobe_q, and decides)
            lanemask lit[K], lit_any = 0ull;
#pragma unroll
            for (int j = 0; j < K; j++) {
              const float cg = fmaf(r[j].dx, sx, fmaf(r[j].dy, sy, r[j].dz * sz));
              lit[j] = alive[j] & __ballot(cg > lobe_thr);
              lit_any |= lit[j];
              LF_PAIR_STAT(0, endfl >> 24, alive[j] != 0ull ? 1u : 0u);
              LF_PAIR_STAT(1, endfl >> 24, lit[j] != 0ull ? 1u : 0u);
              LF_PAIR_STAT(2, endfl >> 24, (unsigned)__popcll(lit[j]));
            }
#ifdef LF_MARCH_LIT_MAP
            if (lit_any != 0ull && g_lit_map[0] != nullptr && lane_id() == 0) {
              const int bx64 = (tx >> a.xs) << (3 + a.xs) >> 6, trow_ = trow;
              const int nbx = (a.W + 63) >> 6;
              const int blkA = ((trow_ * 8) >> 6) * nbx + bx64, blkB = trow_ * nbx + bx64;
              for (int m = 0; m < 4; m++) {
                const int P = 16 << m;
                const int fx = min(P - 1, (int)(lm_u * (float)P)), fy = min(P - 1, (int)(lm_v * (float)P));
                atomicOr(&g_lit_map[m][((size_t)blkA * P + fy) * P + fx], 1ull << (endfl >> 24));
                atomicOr(&g_lit_map[4 + m][((size_t)blkB * P + fy) * P + fx], 1ull << (endfl >> 24));
              }
            }
#endif
            if (lit_any != 0ull) {
              // rare (about 1 % of the wave-paths): march this path again, alone and with the
              // weight, along its own row sequence -- once per wavelength that has a lit lane
              const int q = (int)(endfl >> 24);
              for (int j = 0; j < K; j++) {   // not unrolled: one copy of the weighted march
                lanemask lj = lit[0];
#pragma unroll
                for (int jj = 1; jj < K; jj++) lj = (j == jj) ? lit[jj] : lj;
                if (lj == 0ull) continue;
                const int l = g * K + j;
                // the path's own sequence: one dword per event (record | kind << 16), the constants come
                // from the same resident records the walk uses
                const int* __restrict__ w = seq_table + pairs->ev_off[q];
                const int lo = launder(lane);
                Ray rw{LF_START(0, lo), LF_START(1, lo), 0.0f, 0.0f, LF_START(2, lo),
                       LF_START(3, lo), LF_START(4, lo), LF_START(5, lo), 1.0f};
                rw.r2 = fmaf(rw.px, rw.px, rw.py * rw.py);
                { const float ns = lens->n_start[l]; rw.dx *= ns; rw.dy *= ns; rw.dz *= ns; }
                if (kW1) {   // (ablation: the weight travelled with the ray)
                  rw = j == 0 ? r[0] : j == 1 ? r[K > 1 ? 1 : 0] : r[K > 2 ? 2 : 0];
                } else {
                // events computed a second time, with the weight: for the lit lanes / as wave-wide rows
                n_rm_lane += (unsigned long long)((unsigned)pairs->ev_cnt[q] * (unsigned)__popcll(lj));
                n_rm_rows += (unsigned)pairs->ev_cnt[q];
                for (int left = pairs->ev_cnt[q]; left > 0; --left, ++w) {
                  const unsigned se = (unsigned)*(const int __attribute__((address_space(4)))*)(w);
                  const LfProgRow wr = load_prec(recs, se & 0xffffu);
                  const LfWeightRow ww = load_wrec(wrecs, se & 0xffffu);
                  const unsigned wfl = se >> 16;
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
                                              (wfl & LF_EV_REFLECT) != 0, (wfl & LF_EV_FLAT) != 0, wr.sgn, geom_ok,
                                              w_fs, w_fo, w_fi);
                  }
                }
                }
                // (selects, not branches: with no divergent branch anywhere in the walk the compiler
                // keeps its control flow as plain scalar branches)
                // (the re-march reproduces the first pass bit for bit: rw's direction is r[j]'s)
                const float qq = lobe_q(rw.dx, rw.dy, rw.dz, sx, sy, sz, sun_ss, inv_1mc);
                const float om = 1.0f - qq;
                float contrib = __fdiv_rn(rw.wn, rw.wd) * (om * om);
                contrib = (((lj >> lane) & 1ull) != 0ull && qq < 1.0f && contrib > 0.0f) ? contrib : 0.0f;
                n_light += contrib > 0.0f ? 1u : 0u;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                  // straight into the tile's LDS sums (integers: any order gives the same bits);
                  // lit lanes are ~0.4 % of the rays, six registers of per-lane sums are not worth it
                  const float v = contrib * (lens->sun_radiance[c] * lens->lambda_rgb[l][c]);
                                    const unsigned long long fx = (unsigned long long)(v * 68719476736.0f);
                  if (fx) atomicAdd(&s_acc[lane * 3 + c], fx);
                }
              }
            }
            // back to the fork this path left from (neither flag: that was the primary path)
            if (endfl & LF_EV_REST1) unpark_all(1, alive1);
            else if (endfl & LF_EV_REST0) unpark_all(0, alive0);
          }
        }
      }
      n_exec += exec32; events += ev32; n_clip += clip32; n_vign += vign32; n_tir += tir32; n_scene += scene32;
    }
  }

  // ---- counters: wave reduce, one LDS add per wave, one global add per workgroup ------------
  const int lane = lane_id();
  const int x = ((tx >> a.xs) << (3 + a.xs)) + ((lane & 7) << a.xs) + (tx & ((1 << a.xs) - 1)), y = trow * 8 + (lane >> 3);
  const bool active = x < a.W && y >= a.y0 && y < a.y1;
  const unsigned p = (unsigned)y * (unsigned)a.W + (unsigned)x;
  {
    unsigned long long v0 = active ? (unsigned long long)n_samples * (unsigned)(n_lambda * n_pairs) : 0ull;
    unsigned long long v6 = n_light;
    for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off); v6 += __shfl_down(v6, off); }
    const unsigned long long vals[kMarchCounters] = {v0, events, n_clip, n_vign, n_tir, n_scene, v6, n_exec,
                                                     n_rm_lane, n_rm_rows};
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < kMarchCounters; i++)
        if (vals[i]) atomicAdd(&s_cnt[i], vals[i]);
    }
  }
  __syncthreads();
  if (wave == 0 && lane < kMarchCounters && s_cnt[lane]) atomicAdd(&counters[lane], s_cnt[lane]);
#ifdef LF_MARCH_LIVE_HIST
  if (wave == 0 && lane < 27 && s_hist[lane]) atomicAdd(&counters[kMarchHistSlot + lane], s_hist[lane]);
#endif
#ifdef LF_MARCH_PAIR_STATS
  for (int i = tid; i < 192; i += 64 * kWgWaves)
    if (s_pair[i]) atomicAdd(&counters[kMarchPairSlot + i], s_pair[i]);
#endif

  // ---- the tile's pixels: 8 rows of 8 x 24 contiguous bytes -----------------------------------
  if (wave == 0 && active) {
    if (a.sgroups == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double v = ((double)s_acc[lane * 3 + c] * (1.0 / 68719476736.0)) / (double)a.spp;
        ghost[3 * (size_t)p + c] = a.accumulate ? ghost[3 * (size_t)p + c] + v : v;
      }
    } else {
      // several workgroups share the tile (short launches, e.g. 1/8 of a frame per GPU, would
      // otherwise leave the last wave of long-running workgroups running alone): integer partial
      // sums meet in HBM, k_march_finish converts them -- the same bits as the single-group path
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (s_acc[lane * 3 + c]) atomicAdd(&accum[3 * (size_t)p + c], s_acc[lane * 3 + c]);
    }
  }
}

// LensCamera::generate_ray, batched: one lane per sensor sample, the primary path N-1 .. 0 through
// the prescription (the same event arithmetic as the ghost march); out = {origin xyz on the front
// element, unit direction xyz towards the scene, transmitted weight, alive flag} in lens space
// (z along the axis, light travels +z, the scene is at z < 0).
__global__ __launch_bounds__(256) void k_lens_rays(const LfLensDev* __restrict__ lens,
                                                   const LfPrimaryDev* __restrict__ prim,
                                                   const float* __restrict__ mask, int mw, int mh,
                                                   int lambda, int n, const float* __restrict__ sensor_xy,
                                                   const float* __restrict__ pupil_uv,
                                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < n;
  const float X = active ? sensor_xy[2 * i] : 0.0f, Y = active ? sensor_xy[2 * i + 1] : 0.0f;
  const float pa = active ? pupil_uv[2 * i] : 0.0f, pb = active ? pupil_uv[2 * i + 1] : 0.0f;
  const StartRay s0 = aim_at_pupil(X, Y, pa, pb, lens->pupil_h, lens->pupil_z - lens->z_sensor, lens->geom_norm);
  Ray r{X, Y, 0.0f, fmaf(X, X, Y * Y), s0.dx, s0.dy, s0.dz, s0.w0, 1.0f};
  const bool alive = primary_path(prim, lambda, r, mask, mw, mh, lane);
  if (active) {
    float* o = out + 8 * (size_t)i;
    o[0] = r.px; o[1] = r.py; o[2] = prim->front_zv + r.hz; o[3] = r.dx; o[4] = r.dy; o[5] = r.dz;
    o[6] = alive ? __fdiv_rn(r.wn, r.wd) : 0.0f;
    o[7] = alive ? 1.0f : 0.0f;
  }
}

__global__ void k_native_sqrt(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = lf_sqrt(x[i]);
}

__global__ void k_native_rcp(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = lf_rcp(x[i]);
}

// the launch's rows of ghost_buffer times a power of two (exact): the kernels accumulate on the fixed 2^-36 grid of
// rounds 1-4; a launch whose sums need another exponent (lf_march_fix_bits: an HDR sun) is brought there by scaling the
// device's copy of the radiance by 2^(bits - 36) before and the rows it wrote by 2^(36 - bits) after -- so that the
// path tree's scalar-register-tight walk carries no value it did not carry in round 4 (one more costs it 3-6 %)
__global__ void k_scale_rows(double* __restrict__ ghost, MarchArgs a, double factor) {
  const size_t p = (size_t)a.y0 * a.W + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)a.y1 * a.W) return;
  if (!march_owns(a, (int)(p % a.W), (int)(p / a.W))) return;
#pragma unroll
  for (int c = 0; c < 3; c++) ghost[3 * p + c] *= factor;
}

__global__ void k_march_finish(const unsigned long long* __restrict__ accum, MarchArgs a,
                               double* __restrict__ ghost) {
  const size_t p = (size_t)a.y0 * a.W + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)a.y1 * a.W) return;
  if (!march_owns(a, (int)(p % a.W), (int)(p / a.W))) return;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double v = ((double)accum[3 * p + c] * (1.0 / 68719476736.0)) / (double)a.spp;
    ghost[3 * p + c] = a.accumulate ? ghost[3 * p + c] + v : v;
  }
}

}  // namespace

// host: the fixed-point grid of a launch.  Every contribution is added as (u64)(v * 2^bits), v = weight x lobe x
// (sun_radiance[c] * lambda_rgb[l][c]) <= geom_norm x that product, into 64-bit sums per pixel and channel: the
// largest sum a launch can produce is spp x paths x geom_norm x max_c sum_l (radiance[c] * lambda_rgb[l][c]).
// bits = 36 (the grid of rounds 1-4: every golden and oracle comparison holds) unless that bound times 2^bits
// reaches 2^62; then the largest exponent that keeps it below (a sun of radiance 1e9 at 1024 spp x 8 wavelengths: 2^17).
// The result is scale-covariant: a frame at radiance 2^k x L is 2^k x the frame at L, bit for bit, once both
// leave the default grid.  Mirrored by oracle/lf_geo_oracle.c (geo_fix_bits).
int lf_march_fix_bits(const LfLensDev& L, int n_paths, int spp) {
  double worst = 0.0;
  for (int c = 0; c < 3; c++) {
    double s = 0.0;
    for (int l = 0; l < L.n_lambda; l++) s += (double)(L.sun_radiance[c] * L.lambda_rgb[l][c]);
    worst = std::max(worst, s);
  }
  worst *= (double)spp * (double)n_paths * (double)L.geom_norm;
  int bits = 36;
  while (bits > -100 && std::ldexp(worst, bits) >= 4611686018427387904.0) bits--;   // 2^62
  return bits;
}

// host: the disc every sensor sample aims at -- by default the rear element's clear aperture at its
// vertex plane, or what lf_set_pupil_target names (part of the sampling specification) -- and the
// solid-angle factor of the start weight, pupil area / distance^2 (double, then float)
void lf_apply_pupil_target(lf_ctx* ctx) {
  LfLensDev& L = ctx->lens;
  const int n = L.n_surf;
  if (ctx->pupil_target_h > 0.0f) { L.pupil_h = ctx->pupil_target_h; L.pupil_z = ctx->pupil_target_z; }
  else { L.pupil_h = ctx->raw_semi_ap[n - 1]; L.pupil_z = L.surf[n - 1].zv; }
  const double D = (double)L.z_sensor - (double)L.pupil_z;
  L.geom_norm = (float)((3.14159265358979323846 * (double)L.pupil_h * (double)L.pupil_h) / (D * D));
  ctx->lenscam_dirty = true;   // the lens camera's table / calibration follow the lens and its pupil disc
}

// host: derive the per-interface march constants from the raw prescription (float arithmetic,
// mirrored by the oracle)
void lf_derive_lens(lf_ctx* ctx, int n, int stop, int n_lambda, const float* radius,
                    const float* thickness, const float* ior, const float* semi_ap,
                    float sensor_w_mm) {
  LfLensDev& L = ctx->lens;
  // keep the sun across a lens change
  float keep_sun_dir[3], keep_sun_rad[3], keep_inv = L.sun_inv_one_minus_cos, keep_ss = L.sun_ss;
  for (int c = 0; c < 3; c++) { keep_sun_dir[c] = L.sun_dir[c]; keep_sun_rad[c] = L.sun_radiance[c]; }
  std::memset(&L, 0, sizeof(L));
  for (int c = 0; c < 3; c++) { L.sun_dir[c] = keep_sun_dir[c]; L.sun_radiance[c] = keep_sun_rad[c]; }
  L.sun_inv_one_minus_cos = keep_inv;
  L.sun_ss = keep_ss;
  L.n_surf = n; L.stop = stop; L.n_lambda = n_lambda;
  float z = 0.0f;
  for (int k = 0; k < n; k++) {
    LfSurfaceDev& s = L.surf[k];
    s.zv = z;
    z = z + thickness[k];
    s.curv = radius[k] == 0.0f ? 0.0f : 1.0f / radius[k];
    s.radius = radius[k];
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
      L.surf[k].n_before[l] = n_before;
      L.surf[k].n_after[l] = n_after;
      n_before = n_after;
    }
    L.n_start[l] = n_before;   // the medium between the last interface and the sensor
  }
  // do all glasses disperse the same way along the wavelength columns?  (the cull's pre-pass brackets the spectrum by
  // its first and last column: lf_cull.hip; a table with a column out of order marches everything instead)
  ctx->lens_lambda_monotonic = true;
  {
    int way = 0;
    for (int k = 0; k < n; k++)
      for (int l = 0; l + 1 < n_lambda; l++) {
        const float d = ior[(l + 1) * n + k] - ior[l * n + k];
        if (k == stop || d == 0.0f) continue;
        const int w = d > 0.0f ? 1 : -1;
        if (way == 0) way = w;
        else if (w != way) ctx->lens_lambda_monotonic = false;
      }
  }
  ctx->sensor_w_mm = sensor_w_mm;
  L.pitch = sensor_w_mm / (float)ctx->W;  // refreshed at every launch: the frame may be resized
  for (int k = 0; k < n; k++) ctx->raw_semi_ap[k] = semi_ap[k];
  lf_apply_pupil_target(ctx);
  L.stop_h = stop >= 0 ? semi_ap[stop] : 1.0f;
  for (int l = 0; l < n_lambda; l++)
    for (int c = 0; c < 3; c++) L.lambda_rgb[l][c] = (n_lambda == 3) ? (l == c ? 1.0f : 0.0f)
                                                                      : 1.0f / (float)n_lambda;
}

// host: the march program of one wavelength -- the tree of all selected paths in depth-first order.
//   for k = N-1 .. 0 (the backward leg from the sensor, shared by the primary path and by every pair
//   that reflects below k):
//     if pairs (k, .) exist:  [SAVE0] reflect at k, then the forward leg k+1 .. max j; at every j of
//        a pair: [SAVE1] reflect at j, backward j-1 .. 0 [END, REST1 or REST0 after the last pair]
//     refract backward through k
// skip[r] = (rows to jump << 2) | restore, used when the whole wave is dead after row r: nothing of
// what only these rays would still visit is executed (restore: 0 = program end, 1 = slot 1, 2 = slot 0)
// vertex z of the interface the ray comes from (N = the sensor plane) minus that of interface k:
// what one add turns the ray's z (relative to where it sits) into z relative to k's vertex
static float lf_vertex_step(const LfLensDev& L, int from, int k) {
  return (from >= L.n_surf ? L.z_sensor : L.surf[from].zv) - L.surf[k].zv;
}

static void build_program(const LfLensDev& L, const LfPairsDev& P, int l, std::vector<LfEventRow>& prog,
                          std::vector<int>& skip) {
  prog.clear();
  std::vector<int> target, restore;  // per row: absolute row to jump to when dead, restore kind
  // `from`: the interface the ray comes from (N = the sensor) -- every row of the tree has exactly
  // one predecessor along its paths, so the vertex distance it needs is a property of the row
  auto put = [&](int k, int from, bool reflect, bool fwd, int extra_flags, int mult) {
    const LfSurfaceDev& s = L.surf[k];
    LfEventRow r;
    r.dzv = lf_vertex_step(L, from, k); r.curv = s.curv; r.h2 = s.h2;
    r.eta = fwd ? s.eta_fwd[l] : s.eta_bwd[l];
    r.sgn = fwd ? 1.0f : -1.0f;
    r.flags = (reflect ? LF_EV_REFLECT : 0) | (s.is_stop != 0.0f ? LF_EV_STOP : 0) |
              (s.curv == 0.0f ? LF_EV_FLAT : 0) | extra_flags | (mult << 16);
    r.radius = s.radius;
    r.eta2 = r.eta * r.eta;
    r.n_in = fwd ? s.n_before[l] : s.n_after[l];
    r.n_out = fwd ? s.n_after[l] : s.n_before[l];   // (a mirror's Fresnel factor needs the far side too)
    r.surf_dir = k | ((fwd ? 1 : 0) << 8);
    std::memset(r.pad, 0, sizeof(r.pad));
    prog.push_back(r);
    target.push_back(-1);
    restore.push_back(0);
  };
  const int N = L.n_surf;
  int primary = -1;
  std::vector<std::vector<std::pair<int, int>>> by_i(N);  // i -> (j, q), sorted by j
  for (int q = 0; q < P.n; q++) {
    if (P.ij[q][0] < 0) primary = q;
    else by_i[P.ij[q][0]].push_back({P.ij[q][1], q});
  }
  int kmin = N;
  for (int k = 0; k < N; k++) {
    std::stable_sort(by_i[k].begin(), by_i[k].end());
    if (!by_i[k].empty() && k < kmin) kmin = k;
  }
  if (primary >= 0) kmin = 0;
  std::vector<int> below(N + 1, 0);  // pairs that reflect below k
  for (int k = 0; k < N; k++) below[k + 1] = below[k] + (int)by_i[k].size();
  std::vector<int> prefix_rows;
  for (int k = N - 1; k >= kmin; k--) {
    const auto& js = by_i[k];
    if (!js.empty()) {
      const int sub_first = (int)prog.size();
      std::vector<int> fwd_rows;
      put(k, k + 1, true, false, LF_EV_SAVE0, (int)js.size());
      fwd_rows.push_back(sub_first);
      const int maxj = js.back().first;
      size_t p = 0;
      for (int m = k + 1; m <= maxj; m++) {
        while (p < js.size() && js[p].first == m) {
          const bool last = p + 1 == js.size();
          const int leg_first = (int)prog.size();
          put(m, m == k + 1 ? k : m - 1, true, true, last ? 0 : LF_EV_SAVE1, 1);
          for (int t = m - 1; t >= 0; t--)
            put(t, t + 1, false, false,
                t == 0 ? (LF_EV_END | (last ? LF_EV_REST0 : LF_EV_REST1) | (int)((unsigned)js[p].second << 24)) : 0, 1);
          for (int r = leg_first; r < (int)prog.size(); r++) {
            target[r] = (int)prog.size();
            restore[r] = last ? 2 : 1;
          }
          p++;
        }
        if (m < maxj) {
          fwd_rows.push_back((int)prog.size());
          put(m, m == k + 1 ? k : m - 1, false, true, 0, (int)(js.size() - p));
        }
      }
      for (int r : fwd_rows) { target[r] = (int)prog.size(); restore[r] = 2; }
    }
    const int mult = (primary >= 0 ? 1 : 0) + below[k];
    if (mult > 0) {
      prefix_rows.push_back((int)prog.size());
      put(k, k + 1, false, false, (k == 0 && primary >= 0) ? (LF_EV_END | (int)((unsigned)primary << 24)) : 0, mult);
    }
  }
  for (int r : prefix_rows) { target[r] = (int)prog.size(); restore[r] = 0; }
  // runs of plain rows (no flag in the low byte, the same multiplicity)
  // (a row that only completes a path is plain as well, but nothing can follow it in its run; a
  // curved mirror -- the fork rows -- may START a run: the plain rows of its leg follow in the
  // same pass of the walk)
  const int special = LF_EV_REFLECT | LF_EV_STOP | LF_EV_FLAT | LF_EV_SAVE0 | LF_EV_SAVE1;
  for (int r = (int)prog.size() - 1, run = 0; r >= 0; r--) {
    const int f = prog[r].flags;
    const bool plain = (f & special) == 0;
    const bool mirror = (f & LF_EV_REFLECT) && !(f & (LF_EV_STOP | LF_EV_FLAT | LF_EV_END));
    const bool same_mult = r + 1 < (int)prog.size() &&
                           ((prog[r + 1].flags >> 16) & 0xff) == ((f >> 16) & 0xff);
    const bool next_plain = r + 1 < (int)prog.size() && (prog[r + 1].flags & special) == 0;
    const bool chain = !(f & LF_EV_END) && run > 0 && run < 255 && same_mult && next_plain;
    if (plain) run = chain ? run + 1 : 1;
    else if (mirror) { prog[r].flags |= (chain ? run + 1 : 1) << 8; run = 0; continue; }
    else run = 0;
    prog[r].flags |= run << 8;
  }
  for (size_t r = 0; r < prog.size(); r++)  // a dead wave must always move forward, inside the program
    if (target[r] <= (int)r || target[r] > (int)prog.size()) { prog.clear(); skip.clear(); return; }
  if (l == 0) {
    skip.resize(prog.size());
    for (size_t r = 0; r < prog.size(); r++) skip[r] = ((target[r] - (int)r) << 2) | restore[r];
  }
}

// host: expand every selected pair into its event rows (per wavelength), see LfEventRow
// host only (no device call): the flat sequences followed by the per-wavelength programs, and the
// programs' jump table; fills the offsets of ctx->pairs
lf_status lf_build_march_tables(lf_ctx* ctx, std::vector<LfEventRow>& rows, std::vector<int>& skip) {
  const LfLensDev& L = ctx->lens;
  LfPairsDev& P = ctx->pairs;
  int total = 0;
  for (int q = 0; q < P.n; q++) {
    const int i = P.ij[q][0], j = P.ij[q][1];
    P.ev_off[q] = total;
    P.ev_cnt[q] = i < 0 ? L.n_surf : L.n_surf + 2 * (j - i);
    total += P.ev_cnt[q];
  }
  P.total_events = total;
  rows.assign((size_t)total * L.n_lambda, LfEventRow{});
  for (int l = 0; l < L.n_lambda; l++)
    for (int q = 0; q < P.n; q++) {
      LfEventRow* out = rows.data() + (size_t)l * total + P.ev_off[q];
      const int i = P.ij[q][0], j = P.ij[q][1];
      int n = 0;
      auto put = [&](int k, int from, bool reflect, bool fwd) {
        const LfSurfaceDev& s = L.surf[k];
        LfEventRow r;
        r.dzv = lf_vertex_step(L, from, k); r.curv = s.curv; r.h2 = s.h2;
        r.eta = fwd ? s.eta_fwd[l] : s.eta_bwd[l];
        r.sgn = fwd ? 1.0f : -1.0f;
        r.flags = (reflect ? LF_EV_REFLECT : 0) | (s.is_stop != 0.0f ? LF_EV_STOP : 0) |
                  (s.curv == 0.0f ? LF_EV_FLAT : 0);
        r.radius = s.radius;
        r.eta2 = r.eta * r.eta;
        r.n_in = fwd ? s.n_before[l] : s.n_after[l];
        r.n_out = fwd ? s.n_after[l] : s.n_before[l];
        r.surf_dir = k | ((fwd ? 1 : 0) << 8);
        std::memset(r.pad, 0, sizeof(r.pad));
        out[n++] = r;
      };
      if (i < 0) {
        for (int k = L.n_surf - 1; k >= 0; k--) put(k, k + 1, false, false);
      } else {
        for (int k = L.n_surf - 1; k > i; k--) put(k, k + 1, false, false);
        put(i, i + 1, true, false);
        for (int k = i + 1; k < j; k++) put(k, k == i + 1 ? i : k - 1, false, true);
        put(j, j == i + 1 ? i : j - 1, true, true);
        for (int k = j - 1; k >= 0; k--) put(k, k + 1, false, false);
      }
      if (n != P.ev_cnt[q]) return lf_fail(ctx, LF_ERR_STATE, "event table: sequence length mismatch");
      // bits 8.. : length of the run of plain rows (no flag set) that starts at this row
      for (int k = n - 1, run = 0; k >= 0; k--) {
        run = out[k].flags == 0 ? run + 1 : 0;
        out[k].flags |= run << 8;
      }
    }
  // the flat table is followed by the shared-prefix program (see LF_EV_SAVE0 in lf_internal.h)
  skip.clear();
  P.prog_off = (int)rows.size();
  {
    std::vector<LfEventRow> prog;
    for (int l = 0; l < L.n_lambda; l++) {
      build_program(L, P, l, prog, skip);
      if (prog.empty()) return lf_fail(ctx, LF_ERR_STATE, "march program: inconsistent jump table");
      if (l == 0) P.prog_rows = (int)prog.size();
      rows.insert(rows.end(), prog.begin(), prog.end());
    }
  }
  rows.push_back(LfEventRow{});  // spare
  return LF_OK;
}

// rays per lane for n wavelengths.  Measured on the 8-wavelength frame (4K, 64 spp): K = 1: 456 ms,
// 2: 361, 3 (groups 3 + 3 + 2): 355, 4 (4 + 4, only 5 waves per SIMD): 384 -- three is the sweet
// spot between sharing the scalar walk and keeping 6 waves per SIMD; four wavelengths go as 2 + 2.
static int rays_per_lane(int n_lambda) {
  int k = n_lambda == 4 ? 2 : std::min(n_lambda, 3);
#ifdef LF_EXPERIMENTS
  if (const char* kv = std::getenv("LF_MARCH_K")) {
    int v = std::atoi(kv);
    if (v >= 1 && v <= 3) k = v;
  }
#endif
  return k;
}

// host: merge the per-wavelength programs (identical but for the index ratios) into one program that
// serves K wavelengths at once, in two levels: a header per row, a record per distinct (interface,
// direction of travel) -- found by content: rows whose constants agree share a record.
// out = [n_groups x (prog_rows + 1) headers][padding to 64 bytes][n_groups x n_recs records]
//       [total_events sequence dwords]
static void pack_program(lf_ctx* ctx, const std::vector<LfEventRow>& rows, const std::vector<int>& skip, int K,
                         std::vector<unsigned char>& out, size_t* rec_off, size_t* wrec_off, size_t* seq_off,
                         bool* ok) {
  LfPairsDev& P = ctx->pairs;
  const int n_lambda = ctx->lens.n_lambda, n_groups = (n_lambda + K - 1) / K;
  auto row_at = [&](int l, int i) -> const LfEventRow& {
    return rows[(size_t)P.prog_off + (size_t)l * P.prog_rows + i];
  };
  // a record = one (interface, direction of travel, interface the ray comes from): keyed by what it IS,
  // not by the content of one wavelength's row (two interfaces of equal geometry whose glasses agree at
  // one wavelength and disperse differently must not share a record)
  auto same_record = [](const LfEventRow& q, const LfEventRow& r) {
    return q.surf_dir == r.surf_dir && std::memcmp(&q.dzv, &r.dzv, sizeof(float)) == 0;
  };
  std::vector<int> rec_of(P.prog_rows);
  std::vector<int> first_row;   // a row that uses record r
  for (int i = 0; i < P.prog_rows; i++) {
    const LfEventRow& r = row_at(0, i);
    int id = -1;
    for (size_t k = 0; k < first_row.size() && id < 0; k++)
      if (same_record(row_at(0, first_row[k]), r)) id = (int)k;
    if (id < 0) { id = (int)first_row.size(); first_row.push_back(i); }
    rec_of[i] = id;
  }
  const int n_recs = (int)first_row.size();
  P.prog_recs = n_recs;
  const size_t hdr_bytes = (size_t)n_groups * (P.prog_rows + 1) * sizeof(LfProgHdr);
  *rec_off = (hdr_bytes + 63) & ~(size_t)63;
  *wrec_off = *rec_off + (size_t)n_groups * n_recs * sizeof(LfProgRow);
  static_assert(sizeof(LfWeightRow) == sizeof(LfProgRow), "a weight record sits at its record's offset");
  *seq_off = *wrec_off + (size_t)n_groups * n_recs * sizeof(LfWeightRow);
  out.assign(*seq_off + ((size_t)P.total_events + 1) * sizeof(int), 0);
  // the per-pair sequences (read by the weight re-march): one dword per event, record | kind << 16;
  // every (interface, direction) a pair crosses is in the program, so its record exists
  {
    int* seq = reinterpret_cast<int*>(out.data() + *seq_off);
    for (int e = 0; e < P.total_events; e++) {
      const LfEventRow& r = rows[(size_t)e];   // wavelength 0
      int id = -1;
      for (int k = 0; k < n_recs && id < 0; k++)
        if (same_record(row_at(0, first_row[k]), r)) id = k;
      if (id < 0) { *ok = false; return; }
      seq[e] = id * (int)sizeof(LfProgRow) | ((r.flags & (LF_EV_REFLECT | LF_EV_STOP | LF_EV_FLAT)) << 16);
    }
  }
  LfProgHdr* hdrs = reinterpret_cast<LfProgHdr*>(out.data());
  LfProgRow* recs = reinterpret_cast<LfProgRow*>(out.data() + *rec_off);
  LfWeightRow* wrecs = reinterpret_cast<LfWeightRow*>(out.data() + *wrec_off);
  for (int g = 0; g < n_groups; g++) {
    for (int i = 0; i < P.prog_rows; i++) {
      LfProgHdr& h = hdrs[(size_t)g * (P.prog_rows + 1) + i];
      h.flags = row_at(0, i).flags;
      h.skip = skip[(size_t)i];
      h.rec = rec_of[i] * (int)sizeof(LfProgRow);
      h.rec_next = (i + 1 < P.prog_rows ? rec_of[i + 1] : 0) * (int)sizeof(LfProgRow);
    }
    for (int k = 0; k < n_recs; k++) {
      LfProgRow& o = recs[(size_t)g * n_recs + k];
      LfWeightRow& w = wrecs[(size_t)g * n_recs + k];
      for (int j = 0; j < 3; j++) {
        const int l = std::min(g * K + (j < K ? j : K - 1), n_lambda - 1);
        const LfEventRow& r = row_at(l, first_row[k]);
        if (j == 0) {
          o.dzv = r.dzv; o.curv = r.curv; o.h2 = r.h2; o.sgn = r.sgn;
          o.sc = r.sgn * r.curv;
          o.ch = 0.5f * r.curv; o.c2 = 2.0f * r.curv;   // exact
        }
        // the optical-direction constants (float arithmetic, mirrored by the oracle and k_lens_rays)
        const float n_in2 = r.n_in * r.n_in, n_out2 = r.n_out * r.n_out;
        o.cn22[j] = o.c2 * n_in2;
        o.rn2[j] = r.curv == 0.0f ? 0.0f : r.radius / n_in2;
        o.delta[j] = n_out2 - n_in2;
        const float q = std::fmaf(n_out2, r.n_in, n_in2 * r.n_out);
        w.fs[j] = 1.0f / (r.n_in + r.n_out);
        w.fo[j] = n_out2 / q;
        w.fi[j] = n_in2 / q;
      }
    }
  }
}

static lf_status build_event_table(lf_ctx* ctx) {
  std::vector<LfEventRow> rows;
  std::vector<int> skip;
  lf_status st = lf_build_march_tables(ctx, rows, skip);
  if (st != LF_OK) return st;
  ctx->march_k = rays_per_lane(ctx->lens.n_lambda);
  std::vector<unsigned char> prog;
  size_t rec_off = 0, wrec_off = 0, seq_off = 0;
  bool packed = true;
  pack_program(ctx, rows, skip, ctx->march_k, prog, &rec_off, &wrec_off, &seq_off, &packed);
  if (!packed || (size_t)ctx->pairs.prog_recs * sizeof(LfProgRow) > 0xffffu)
    return lf_fail(ctx, LF_ERR_STATE, "march program: a pair crosses an interface the program has no record for");
  // (the flat per-pair rows and the jump table stay on the host: the device walks headers, records and
  // sequence dwords only)
  // The context's stream is non-blocking, so the null-stream copies below are NOT ordered behind a
  // k_march that is still walking the previous program: a program / jump-table pair that changes
  // under a live kernel can send a wave past the program's end.  Drain the stream first.
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  auto upload = [&](void** dev, size_t* cap, const void* src, size_t bytes) -> hipError_t {
    if (bytes > *cap) {
      if (*dev) (void)hipFree(*dev);
      *dev = nullptr; *cap = 0;
      hipError_t e = hipMalloc(dev, bytes);
      if (e != hipSuccess) return e;
      *cap = bytes;
    }
    return bytes ? hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  LF_HIP(ctx, upload((void**)&ctx->prog_dev, &ctx->prog_cap, prog.data(), prog.size()));
  ctx->prog_rec_off = rec_off;
  ctx->prog_wrec_off = wrec_off;
  ctx->prog_seq_off = seq_off;
  ctx->events_dirty = false;
  return LF_OK;
}

static int march_strata(int spp) {
  int G = (int)std::floor(std::sqrt((double)spp));
  while ((G + 1) * (G + 1) <= spp) G++;
  while (G * G > spp) G--;
  return G;
}

// what a launch uploads before anything runs: the event program, the lens (with the launch's fixed-point grid), the paths
static lf_status march_upload(lf_ctx* ctx, int spp, int total_paths = 0) {
  ctx->lens.pitch = ctx->sensor_w_mm / (float)ctx->W;
  if (ctx->events_dirty) {
    lf_status st = build_event_table(ctx);
    if (st != LF_OK) return st;
  }
  {
    // the launch's fixed-point grid: the kernels scale by the literal 2^36 and back by 2^-36, the device's sun_radiance
    // carries 2^(bits - 36) (exact: a power of two) and k_scale_rows takes it out of the rows the launch wrote --
    // together (u64)(v 2^bits) and back, as the contract says
    const int bits = lf_march_fix_bits(ctx->lens, total_paths > 0 ? total_paths : ctx->pairs.n, spp);
    ctx->march_fix_bits = bits;
    LfLensDev up = ctx->lens;
    for (int c = 0; c < 3; c++) up.sun_radiance[c] = std::ldexp(ctx->lens.sun_radiance[c], bits - 36);
    // (pageable source of an asynchronous copy: the runtime stages it before the call returns)
    LF_HIP(ctx, hipMemcpyAsync(ctx->lens_dev, &up, sizeof(LfLensDev), hipMemcpyHostToDevice, ctx->stream));
  }
  LF_HIP(ctx, hipMemcpyAsync(ctx->pairs_dev, &ctx->pairs, sizeof(LfPairsDev), hipMemcpyHostToDevice,
                             ctx->stream));
  return LF_OK;
}

// lf_cull_prepare: this rank's share of the cull table for the launch lf_trace_ghosts(spp) is about to make, and nothing
// else (the host completes the table with its own all-gather, lf_cull_commit takes it over)
lf_status lfk_cull_prepare(lf_ctx* ctx, int spp) {
  lf_status st = march_upload(ctx, spp);
  if (st != LF_OK) return st;
  const int G = march_strata(spp);
  ctx->cull_hash_pending = 0;
  if (!lf_cull_applies(ctx, G)) return LF_OK;      // (the launch will march everything: nothing to share)
  ctx->cull_prepare_only = true;
  st = lfk_cull_prepass(ctx, G, spp);
  ctx->cull_prepare_only = false;
  return st;
}

// one launch of the march over the selection ctx->pairs holds.  A selection of more than 64 paths (a mask's bits) is
// marched culled in TWO such launches over its halves (lfk_march below): chunk c of n_chunks, the fixed-point grid sized
// for all total_paths, the integer sums of both halves in `accum` (exact, so the frame is the one launch's), converted
// once after the last.
static lf_status march_launch(lf_ctx* ctx, int spp, uint64_t key, int chunk, int n_chunks, int total_paths) {
  const LfApertureDev& m = ctx->ap[LF_APERTURE_STARBURST];
  {
    const lf_status st = march_upload(ctx, spp, total_paths);
    if (st != LF_OK) return st;
  }
  if (ctx->y1 <= ctx->y0) {
    // no rows of this context's own -- but a table shared through the communicator is completed by a COLLECTIVE:
    // every rank takes part, whatever it renders
    if (ctx->cull_share_how == 1 && ctx->cull_share_n > 1 && lf_cull_applies(ctx, march_strata(spp)))
      return lfk_cull_prepass(ctx, march_strata(spp), spp);
    return LF_OK;
  }
  MarchArgs a;
  a.mw = m.w; a.mh = m.h; a.W = ctx->W; a.H = ctx->H; a.y0 = ctx->y0; a.y1 = ctx->y1;
  a.spp = spp;
  a.G = march_strata(spp);
  a.inv_G = 1.0f / (float)a.G;
  a.sub_bits = ctx->march_sub_bits;
  a.inv_sub = 1.0f / (float)(1 << a.sub_bits);
  a.key = make_uint2((unsigned)key, (unsigned)(key >> 32));
  a.inv_stop_h = 1.0f / ctx->lens.stop_h;
  a.half_w = 0.5f * (float)ctx->W; a.half_h = 0.5f * (float)ctx->H;
  a.vz = ctx->lens.pupil_z - ctx->lens.z_sensor;
  a.accumulate = ctx->ghost_accumulate ? 1 : 0;
  a.xs = ctx->march_xstride_log2;
  a.lobe_thr = lf_march_lobe_thr(ctx->lens);      // candidate selection (the contract: lf_internal.h)
  // tile rows (8 sensor rows each) of the band that belong to this context's interleave phase -- or, the frame dealt by
  // blocks, the 64 wave tiles of every block of this context's
  a.deal = lf_deal_of(ctx);
  size_t tiles;
  if (a.deal.bx > 0 && a.deal.n > 1) {
    const int nblk = a.deal.bx * ((ctx->H + (1 << kDealBlockLog2) - 1) >> kDealBlockLog2);
    const int n_own = (nblk - a.deal.rank + a.deal.n - 1) / a.deal.n;
    if (n_own <= 0) return LF_OK;
    a.trow0 = 0; a.tperiod = 1;
    tiles = (size_t)n_own * 64;
  } else {
    a.deal.bx = 0;
    const int t_lo = ctx->y0 / 8, t_hi = (ctx->y1 + 7) / 8;  // [t_lo, t_hi)
    const int period = ctx->row_period, phase = ctx->row_phase;
    int first = t_lo + ((phase - t_lo) % period + period) % period;
    if (first >= t_hi) return LF_OK;
    const int n_trows = (t_hi - 1 - first) / period + 1;
    a.trow0 = first; a.tperiod = period;
    tiles = (size_t)n_trows * ((((size_t)ctx->W + (8u << a.xs) - 1) >> (3 + a.xs)) << a.xs);
  }
  // A launch that covers only part of the frame (one GPU's share) splits each tile's samples over
  // `sgroups` workgroups (power of two): more, shorter workgroups keep its tail short -- but a workgroup
  // needs >= 64 samples to amortise its set-up and its 192 global atomics.  The whole frame on one GPU
  // stays unsplit: the split is worth another 0.5 % there and costs 5x the HBM write traffic (atomics).
  // Measured per share of the 1080p bench frame (profiles/r03_share_timing.json): 1 / 2 / 2 / 2 groups
  // for 1, 1/2, 1/4, 1/8 of the frame.
  // The culled march (round 6: the started paths' common leg once) prefers its tiles whole: per rank of the block deal,
  // 1 / 2 / 4 / 8 groups: 4.86 / 4.91 / 5.35 / 6.52 ms for 1/8 of the bench frame, 9.09 / 9.38 / 10.4 / 12.7 for 1/4
  // (profiles/r06_cull_bounds.txt) -- split only launches of a few hundred tiles.
  a.sgroups = 1; a.tail_from = 0; a.tail_groups = 1;
  if (lf_cull_reason_of(ctx, a.G) == LF_CULL_APPLIED) {
    while (tiles * a.sgroups < 2000 && a.sgroups * 2 * 64 <= spp) a.sgroups *= 2;
  } else {
    while (tiles * a.sgroups < 8000 && a.sgroups * 2 * 64 <= spp) a.sgroups *= 2;
    if (a.sgroups == 1 && tiles < 20000 && 2 * 64 <= spp) a.sgroups = 2;
  }
#ifdef LF_EXPERIMENTS
  if (const char* sgv = std::getenv("LF_MARCH_SGROUPS")) {
    int v = std::atoi(sgv);
    if (v >= 1 && v * 4 <= std::max(4, spp) && (v & (v - 1)) == 0) a.sgroups = v;
  }
#endif
  if (n_chunks > 1) a.sgroups = std::max(a.sgroups, 2);      // (the halves meet in the integer accumulator)
  a.n_tiles = (int)tiles;
  const size_t blocks = ((tiles + 63) / 64 * 64) * a.sgroups;
  if (blocks > 0x7fffffffull) return lf_fail(ctx, LF_ERR_INVALID, "band too large for one launch");
  const size_t n_acc = (size_t)ctx->W * ctx->H_alloc * 3;
  if (a.sgroups > 1) {
    if (!ctx->accum) LF_HIP(ctx, hipMalloc((void**)&ctx->accum, n_acc * sizeof(unsigned long long)));
    if (chunk == 0)
      LF_HIP(ctx, hipMemsetAsync(ctx->accum + (size_t)ctx->y0 * ctx->W * 3, 0,
                                 (size_t)(ctx->y1 - ctx->y0) * ctx->W * 3 * sizeof(unsigned long long),
                                 ctx->stream));
  }
  // experiments only: unused dynamic LDS caps the workgroups a CU holds (occupancy sweeps,
  // profiles/r03_march_variants.txt)
  size_t dyn_lds = 0;
#ifdef LF_EXPERIMENTS
  if (const char* dl = std::getenv("LF_MARCH_DYN_LDS")) dyn_lds = (size_t)std::max(0, std::atoi(dl));
#endif
#ifdef LF_MARCH_LIT_MAP
  static unsigned long long* lit_dev[8] = {};
  static size_t lit_n[8] = {};
  if (std::getenv("LF_LIT_MAP") && !lit_dev[0]) {
    const size_t nbx = (ctx->W + 63) >> 6, nA = nbx * ((ctx->H + 63) >> 6), nB = nbx * ((ctx->H + 7) >> 3);
    for (int m = 0; m < 8; m++) {
      const size_t P = 16u << (m & 3);
      lit_n[m] = (m < 4 ? nA : nB) * P * P;
      LF_HIP(ctx, hipMalloc((void**)&lit_dev[m], lit_n[m] * 8));
      LF_HIP(ctx, hipMemset(lit_dev[m], 0, lit_n[m] * 8));
    }
    LF_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_lit_map), lit_dev, sizeof(lit_dev)));
  }
  if (std::getenv("LF_LIT_MAP_DUMP") && lit_dev[0]) {
    LF_HIP(ctx, hipDeviceSynchronize());
    for (int m = 0; m < 8; m++) {
      std::vector<unsigned long long> h(lit_n[m]);
      LF_HIP(ctx, hipMemcpy(h.data(), lit_dev[m], lit_n[m] * 8, hipMemcpyDeviceToHost));
      unsigned long long bits = 0, nz = 0;
      for (unsigned long long v : h) { bits += (unsigned long long)__builtin_popcountll(v); nz += v != 0; }
      std::fprintf(stderr, "LIT_MAP block %s P %d entries %zu nonzero %llu bits %llu fraction_of_entry_paths %.5f\n",
                   m < 4 ? "64x64" : "64x8", 16 << (m & 3), lit_n[m], nz, bits, (double)bits / ((double)lit_n[m] * ctx->pairs.n));
    }
  }
#endif
  const int fix_shift = 36 - ctx->march_fix_bits;       // != 0: an HDR launch (see k_scale_rows)
  const size_t band_px = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
  if (fix_shift != 0 && a.accumulate && chunk == 0) {     // what the buffer holds joins the launch's grid, exactly
    hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((band_px + 255) / 256)), dim3(256), 0, ctx->stream, ctx->ghost, a,
                       std::ldexp(1.0, -fix_shift));
    LF_HIP(ctx, hipGetLastError());
  }
  // the paths a pre-pass found able to reach the light (lf_cull.hip) -- or every path of every sample
  ctx->cull_reason = lf_cull_reason_of(ctx, a.G);
  ctx->last_march_culled = ctx->cull_reason == LF_CULL_APPLIED;
  if (ctx->last_march_culled) {
    lf_status st = lfk_cull_prepass(ctx, a.G, spp);
    if (st != LF_OK) return st;
    // a table that starts most of everything (a very wide sun, a handful of samples): the path tree is faster
    // Where the two kernels meet: the path tree executes ~(22 + 1.4 n) events per sample and wavelength for n paths
    // (shared legs: 85 at n = 46, 53 at n = 21), the culled march ~10.5 per STARTED path plus a pre-pass worth 0.3 of
    // its time -- equal at a started fraction of 0.10 + 1.6 / n (0.135 at 46 paths, 0.18 at 21: measured on c3 / c2).
    const double meet = std::min(0.5, ctx->cull_max_fraction + 1.6 / (double)std::max(1, ctx->pairs.n));
    if (ctx->cull_started_fraction > meet && !ctx->cull_force) ctx->cull_reason = LF_CULL_TABLE_TOO_FULL;
    // a table its audit refuted is not used, whatever a test asks for
    if (ctx->cull_bad_hash != 0 && ctx->cull_bad_hash == ctx->cull_hash) ctx->cull_reason = LF_CULL_AUDIT_REFUTED;
    ctx->last_march_culled = ctx->cull_reason == LF_CULL_APPLIED;
  }
  if (ctx->last_march_culled) {
    lf_status st = lfk_march_culled(ctx, a, blocks, dyn_lds);
    if (st != LF_OK) return st;
  } else {
  hipEvent_t ev = lf_timing_begin(ctx, LFK_MARCH);
#define LF_LAUNCH_MARCH(KK)                                                                        \
  hipLaunchKernelGGL(k_march<KK>, dim3((unsigned)blocks), dim3(64 * kWgWaves), dyn_lds, ctx->stream, ctx->lens_dev, \
                     ctx->pairs_dev, (const int*)(ctx->prog_dev + ctx->prog_seq_off),                \
                     (const LfProgHdr*)ctx->prog_dev,                                               \
                     (const LfProgRow*)(ctx->prog_dev + ctx->prog_rec_off),                         \
                     (const LfWeightRow*)(ctx->prog_dev + ctx->prog_wrec_off), m.texels, a,         \
                     ctx->ghost, ctx->accum, ctx->counters_dev)
  switch (ctx->march_k) {
    case 1: LF_LAUNCH_MARCH(1); break;
    case 2: LF_LAUNCH_MARCH(2); break;
    default: LF_LAUNCH_MARCH(3); break;
  }
#undef LF_LAUNCH_MARCH
  lf_timing_end(ctx, LFK_MARCH, ev);
  LF_HIP(ctx, hipGetLastError());
  }
  if (chunk + 1 < n_chunks) return LF_OK;                 // (the other half of the selection follows)
  if (a.sgroups > 1) {
    const size_t px = (size_t)(ctx->y1 - ctx->y0) * ctx->W;
    hipLaunchKernelGGL(k_march_finish, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->accum, a, ctx->ghost);
    LF_HIP(ctx, hipGetLastError());
  }
  if (fix_shift != 0) {
    hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((band_px + 255) / 256)), dim3(256), 0, ctx->stream, ctx->ghost, a,
                       std::ldexp(1.0, fix_shift));
    LF_HIP(ctx, hipGetLastError());
  }
  return LF_OK;
}

lf_status lfk_march(lf_ctx* ctx, int spp, uint64_t key) {
  const int n = ctx->pairs.n;
  ctx->cull_chunks = 1;
  // more paths than a mask has bits: the culled march in two launches over the halves of the selection (each with its own
  // table), where the cull applies at all and the table is this context's own
  if (n <= kCullMaxPaths || n > 2 * kCullMaxPaths || ctx->cull_share_how != 0 || ctx->y1 <= ctx->y0 ||
      lf_cull_reason_of(ctx, march_strata(spp)) != LF_CULL_APPLIED)
    return march_launch(ctx, spp, key, 0, 1, n);
  const LfPairsDev full = ctx->pairs;
  lf_status st = LF_OK;
  bool culled = true;
  int reason = LF_CULL_APPLIED;
  const int half = (n + 1) / 2;
  for (int c = 0; c < 2 && st == LF_OK; c++) {
    LfPairsDev P;
    std::memset(&P, 0, sizeof(P));
    const int q0 = c == 0 ? 0 : half, q1 = c == 0 ? half : n;
    for (int q = q0; q < q1; q++) { P.ij[P.n][0] = full.ij[q][0]; P.ij[P.n][1] = full.ij[q][1]; P.n++; }
    ctx->pairs = P;
    ctx->events_dirty = true;
    st = march_launch(ctx, spp, key, c, 2, n);
    culled = culled && ctx->last_march_culled;
    if (ctx->cull_reason != LF_CULL_APPLIED) reason = ctx->cull_reason;
  }
  ctx->pairs = full;
  ctx->events_dirty = true;
  ctx->cull_chunks = 2;
  ctx->last_march_culled = culled;
  ctx->cull_reason = reason;
  return st;
}

lf_status lfk_native_sqrt(lf_ctx* ctx, const float* d_x, float* d_y, size_t n) {
  if (n == 0) return LF_OK;
  hipLaunchKernelGGL(k_native_sqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_x,
                     d_y, n);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status lfk_native_rcp(lf_ctx* ctx, const float* d_x, float* d_y, size_t n) {
  if (n == 0) return LF_OK;
  hipLaunchKernelGGL(k_native_rcp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_x, d_y, n);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status lfk_lens_rays(lf_ctx* ctx, int lambda, int n, const float* d_xy, const float* d_uv,
                        float* d_out) {
  const LfApertureDev& m = ctx->ap[LF_APERTURE_STARBURST];
  ctx->lens.pitch = ctx->sensor_w_mm / (float)std::max(1, ctx->W);
  LF_HIP(ctx, hipMemcpyAsync(ctx->lens_dev, &ctx->lens, sizeof(LfLensDev), hipMemcpyHostToDevice,
                             ctx->stream));
  lf_status st = lf_upload_primary_table(ctx);
  if (st != LF_OK) return st;
  hipLaunchKernelGGL(k_lens_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->lens_dev, ctx->primary_dev, m.texels, m.w, m.h, lambda, n, d_xy, d_uv, d_out);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}
