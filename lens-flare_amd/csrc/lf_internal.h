// lf_internal.h -- shared declarations of liblensflare_hip.so (host + gfx950 device code).
// Nothing here is part of the ABI; the ABI is include/lensflare.h.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "lensflare.h"

// ---- limits -------------------------------------------------------------------------------
// k_march's counters: the seven of lf_counters, executed events, re-march lane events, re-march rows
constexpr int kMarchCounters = 10;
constexpr int kMarchCounterSlots = 256;  // allocated (instrumented experiment builds append their tallies)
constexpr int kMarchHistSlot = 16;       // -DLF_MARCH_LIVE_HIST: 3 row kinds x 9 live-lane buckets
constexpr int kMarchPairSlot = 64;       // -DLF_MARCH_PAIR_STATS: 3 x 64 per-path tallies
constexpr int kMaxParaxialGhosts = 3 * 105;          // 3 colours x C(15,2) pairs
constexpr int kMaxGhostTris = 2 * kMaxParaxialGhosts; // two triangles per quad

// ---- device-resident frame state ----------------------------------------------------------
// Mirrors the public per-frame fields of PathTracer (pathtracer.h:93-94,128-135).  Written either
// by the frame-setup kernel (find_sun_pos on the device) or by lf_set_flares.
struct LfFlares {
  int n_flares;
  float angle_to_sun;
  double origin[LF_MAX_FLARES][2];
  double radiance[LF_MAX_FLARES][3];
  double axis_ray[2];
};

// One rasterisable ghost triangle, vertices already y-sorted and shifted by -0.5
// (rasterize_textured_triangle, pathtracer.cpp:346-394).
struct LfGhostTri {
  float x0, y0, u0, v0, x1, y1, u1, v1, x2, y2, u2, v2;
  int bx0, bx1, by0, by1;  // loops run x in [bx0,bx1), y in [by0,by1)  (:402-403)
  double colour[3];        // ghost_color: unit R, G or B times 10 / scale_amt^2 widened (:482-494)
};

struct LfGhostList {
  int n_tris;
  int pad;
  LfGhostTri tri[kMaxGhostTris];
};

// Paraxial prescription (globals of pathtracer.cpp:541-556 + constants of :619-633, :737)
struct LfParaxialLens {
  int n, stop;
  float thickness[LF_MAX_SURFACES];
  float curvature[LF_MAX_SURFACES];
  float ior[3][LF_MAX_SURFACES];
  double clip;
  float recast_pos, recast_neg, marginal;
};

struct LfCamera {
  double c2w[9];  // row-major
  double pos[3];
  double hfov_deg, vfov_deg;
  double n_clip = 0.01, f_clip = 100.0;  // Camera::nClip / fClip (camera.h:188)
};

// ---- scene term (row f2): LfSceneDev, LfEnvDev and what they point at ---------------------
#include "lf_scene_types.h"

// per-wavelength starburst (row f4): n = 0 is the reference's monochrome starburst
struct LfStarSpectrum {
  int n, pad;
  double scale[LF_MAX_LAMBDA];     // lambda_ref / lambda_l
  double rgb[LF_MAX_LAMBDA][3];    // weight of wavelength l in R, G, B
};

// the DirectionalLights of find_sun_pos as kernel arguments (6 doubles each: posLight, radiance)
constexpr int kMaxSunLightArgs = 32;
struct LfSunLightArgs { double v[6 * kMaxSunLightArgs]; };

struct LfApertureDev {
  float* texels = nullptr;  // w*h
  int w = 0, h = 0;
  lf_aperture_stats* stats = nullptr;  // device copy
  lf_aperture_stats host_stats{};
  double open_radius = 1.0;   // radius of the circle around the texels > 0, in units of the half width (host, lf_set_aperture)
  bool valid = false;
};

// ---- geometric lens (north-star march) -------------------------------------------------------
// One entry per interface and wavelength, everything the march needs, pre-derived on the host in
// float from the raw prescription (see lf_march.hip for the formulas).
struct LfSurfaceDev {
  float zv;        // vertex z (scene side first, light travels +z)
  float curv;      // 1/R, 0 for flats and the stop
  float radius;    // R as given (0 for flats)
  float h2;        // semi-aperture squared
  float is_stop;   // 1.0f for the stop
  float eta_fwd[LF_MAX_LAMBDA];  // n_before / n_after   (ray travelling +z refracts with this)
  float eta_bwd[LF_MAX_LAMBDA];  // n_after / n_before   (ray travelling -z)
  float n_before[LF_MAX_LAMBDA]; // index of the medium on the scene side of the interface
  float n_after[LF_MAX_LAMBDA];  // ... on the sensor side (the stop leaves the medium unchanged)
};

struct LfLensDev {
  int n_surf, stop, n_lambda, pad;
  float z_sensor;      // sensor plane z
  float pitch;         // mm per pixel
  float pupil_h;       // semi-aperture of the rear-most surface (pupil sampling disc)
  float pupil_z;       // z of the rear-most vertex
  float geom_norm;     // pupil area / (pupil_z - z_sensor)^2
  float stop_h;        // semi-aperture of the stop
  float sun_dir[3];    // unit, towards the sun (z < 0)
  float sun_radiance[3];
  float sun_inv_one_minus_cos;  // 1 / (1 - cos(angular radius))
  float sun_ss;                 // |sun_dir|^2 of the float vector as stored (exact in double, narrowed)
  float lambda_rgb[LF_MAX_LAMBDA][3];
  float n_start[LF_MAX_LAMBDA];  // index of the medium between the last interface and the sensor
  LfSurfaceDev surf[LF_MAX_SURFACES];
};

struct LfPairsDev {
  int n;
  int total_events;              // rows of the event table per wavelength
  int ij[LF_MAX_PAIRS + 1][2];   // (-1,-1) = primary path
  int ev_off[LF_MAX_PAIRS + 1];  // first row of pair q in the event table
  int ev_cnt[LF_MAX_PAIRS + 1];  // N + 2(j - i) rows
  int prog_off;                  // first row of the shared-prefix program (wavelength 0)
  int prog_rows;                 // rows of the program per wavelength
  int prog_recs;                 // distinct interface records of a wavelength group's program
};

// One pre-expanded surface event: everything the march needs for it in ONE 32-byte scalar load.
// Two tables use it: the flat per-(wavelength, pair) sequences (n_lambda x total_events rows, read
// only by the rare weight re-march) and the per-wavelength path-tree program (see LF_EV_SAVE0).
struct alignas(32) LfEventRow {
  float dzv;       // vertex z of the interface the ray comes from (or the sensor) minus this one's
  float curv, h2, eta;
  float sgn;       // +1: the ray travels +z (towards the sensor), -1: -z
  int flags;       // bit 0: mirror reflection, bit 1: the stop, bit 2: flat (curv == 0)
  float radius;    // 1 / curv as given in the prescription (0 for flats)
  float eta2;      // eta * eta (float product)
  // (host only; lf_march_tables exports the eight fields above)
  float n_in, n_out;   // index of the medium the ray arrives in / leaves in (equal for a mirror and the stop)
  int surf_dir;        // interface index | direction of travel << 8: what a record is keyed by
  int pad[5];
};
enum { LF_EV_REFLECT = 1, LF_EV_STOP = 2, LF_EV_FLAT = 4 };
// A program row as the device walks it: the interface once, the index ratios of the up to three
// wavelengths that march it together (lf_march.hip, k_march<K>); one 64-byte scalar load.
// (The program is stored in two levels: a 16-byte header per row -- what is particular to the row --
// and one 64-byte record per distinct (interface, direction of travel) -- the constants of the
// event, shared by all the rows that cross that interface that way.  As one table of 64-byte rows
// the double-Gauss program is 27 KB per wavelength group and misses the CU's 16 KB scalar cache on
// 35 % of its loads (SQC_DCACHE_MISSES, profiles/r02_frontend_counters.json); as 6.8 KB of headers
// + 1.4 KB of records it stays resident.)
struct alignas(16) LfProgHdr {
  int flags;        // as LfEventRow::flags of the per-wavelength program rows
  int skip;         // jump-table entry of this row: (rows to jump << 2) | state to restore
  int rec;          // byte offset of this row's record in the group's record table
  int rec_next;     // ... of the NEXT row's: both loads of the next row can then be issued together
};
// The march carries the ray's direction as OPTICAL direction cosines K = n d (|K| = the index of the
// medium the ray is in, a property of the row): Snell's law is then K' = K + (n' cos t' - n cos t) N with
// no scaling of K by the index ratio, and cos^2 of the refraction angle is one add (round 3; rounds 1-2
// carried the unit direction d and multiplied it by eta = n / n' at every refraction: 4 vector
// instructions per event more).  Per wavelength the record therefore holds
//   cn22 = 2 c n^2      (c F of the vertex-form intersection, with the ray parameter in units of 1 / n)
//   rn2  = R / n^2      (the root (G - sgn sqrt(disc)) R / n^2 of that quadratic)
//   delta = n'^2 - n^2  ((n' cos t')^2 = (n cos t)^2 + delta; negative: total reflection)
// n = index the ray arrives in, n' = leaves in; all float products / quotients of the prescription's
// indices (lf_march.hip pack_program, mirrored by the oracle).
struct alignas(64) LfProgRow {
  float dzv, curv, h2, sc;   // sc = sgn * c
  float sgn;
  float delta[3];   // wavelength g*K + j of group g (repeated past the group's / the lens' last one)
  float cn22[3];
  float ch;         // curv / 2
  float rn2[3];
  float c2;         // 2 curv (both exact: lf_march.hip, surface_event)
};
// what only the weight re-march needs of an (interface, direction): the scale factors of the Fresnel
// fraction (surface_event<true>): fs = 1 / (n + n'), fo = n'^2 / q, fi = n^2 / q, q = n'^2 n + n^2 n'
struct alignas(64) LfWeightRow {
  float fs[3], pad0;
  float fo[3], pad1;
  float fi[3], pad2;
  float pad3[4];
};
// The march does not walk the per-pair sequences one by one: every path of a (sample, wavelength)
// starts with the same backward leg from the sensor, and all pairs (i, .) share the forward leg that
// follows the reflection at i.  The per-wavelength *program* is that tree in depth-first order:
// event rows as above plus, in the flags,
//   SAVE0 / SAVE1  before the event, park the ray state in slot 0 (a prefix fork: the row is the
//                  reflection at i) / slot 1 (a forward-leg fork: the row is the reflection at j)
//   END            after the event the path is complete (lobe test, tallies); bits 24.. = its index
//                  in LfPairsDev; then REST1 / REST0 take the parked state back (neither: the end)
//   bits 8..15     length of the run of plain rows starting here (0 for any flagged row)
//   bits 16..23    multiplicity: how many logical paths share this row (events and fates are
//                  tallied per logical path, exactly as if each had been marched on its own)
enum { LF_EV_REST1 = 8, LF_EV_SAVE0 = 0x10, LF_EV_SAVE1 = 0x20, LF_EV_END = 0x40, LF_EV_REST0 = 0x80 };

// ---- path culling (round 5; lf_cull.hip) ------------------------------------------------------------
// Which paths can carry light from the sun to which part of the sensor through which part of the pupil: for
// every block of 64 x 64 sensor pixels and every cell of a P x P grid over the pupil square (P = G * m: m x m
// cells inside each of the march's G x G strata -- a wave aims all its lanes at ONE sub-cell of its stratum, so
// it knows which table cell it is in) a 64-bit mask of the selected paths, built by a coarse-to-fine pre-pass
// (k_cull_level) that marches 13 rays per (block, cell, path) and bounds where the whole 4-D box can go.  The
// march then starts ONLY the paths whose bit is set: the others end outside the sun's lobe or on a diaphragm,
// i.e. add exactly 0 -- the pixels are those of the full enumeration, bit for bit.  Entry `cells` of a block
// is the union over its cells (the unstratified samples s >= G * G of a non-square sample count).
// sensor blocks of 2^6 = 64 or 2^7 = 128 pixels a side (lf_cull_block_log2): a wave tile of any stride lies inside one;
// 128 where that is still <= 1.25 mm on the sensor (4K on 36 mm: the blocks of 1080p in millimetres, a quarter of the
// pre-pass's boxes) and the launch has few samples (lfk_cull_prepass)
constexpr int kCullBlockLog2 = 6;          // ... 5 or 4 (32 / 16 pixels) on frames whose 64 pixels are too large on the sensor
constexpr double kCullBigBlockMm = 1.25;
constexpr int kCullMaxPaths = 64;        // bits of a mask
constexpr double kCullMaxBlockMm = 1.8;  // a block may be this large on the sensor at most (lf_cull_applies)
constexpr int kCullOcc = 32;             // the stop mask's occupancy grid: kCullOcc x kCullOcc cells, any texel > 0
// A table SHARED between n ranks (lf_set_cull_share): rank b mod n builds block b's row, the rows a rank builds lie
// together (its slab, share_nb rows), the slabs follow each other in rank order -- one in-place all-gather of equal
// slabs completes the table everywhere.  Dealing the blocks round-robin balances the pre-pass: what a block costs
// depends on how many ghosts cross it.  n = 1: row b is block b.
__host__ __device__ inline size_t lf_cull_row_of_block(int b, int share_n, int share_nb) {
  return share_n > 1 ? (size_t)(b % share_n) * (size_t)share_nb + (size_t)(b / share_n) : (size_t)b;
}
struct LfCullArgs {
  const unsigned long long* table;   // [rows][cells + 1], row = lf_cull_row_of_block(block); null = every path everywhere
  int blocks_x, blocks_y;
  int prefix_ok;                      // the selection's order lets the started paths share their common leg (march_started_set)
  int multi;                          // a wave tile spans several blocks (blocks of 16 / 32 pixels under a 64-pixel tile): rows per lane
  int share_n, share_nb;              // see lf_cull_row_of_block
  int blk_log2;                       // log2 of a block's side in pixels
  int cells;                          // P * P, P = G * m cells per axis of the pupil square
  int P, m, m_shift;                  // m (1, 2 or 4) table cells per axis inside one stratum; m = 2^m_shift <= the
                                      // sub-cells per axis of the sampling specification (lf_set_pupil_subcells)
  unsigned long long* tail_acc;       // [tail tile][64 lanes][3]: where the workgroups of a split tail tile meet (MarchArgs::tail_from);
  int* tail_done;                     // [tail tile] arrivals: the last one converts and clears both (all zero between launches)
  int items_chunk0;                   // k_march_items: samples per chunk to begin with (the host's estimate from the table's started fraction)
#ifdef LF_EXPERIMENTS
  unsigned long long* wg_clock;       // [workgroup][2]: wall clock at its start / end (LF_MARCH_WG_CLOCK=<file>)
#endif
};

// ---- how a frame is dealt to the ranks of a multi-GPU job ------------------------------------------------------
// by tile rows (8 sensor rows, t % n == rank: lf_set_row_interleave, rounds 1-5) or by BLOCKS of 64 x 64 pixels (row-major
// index b % n == rank: lf_set_block_deal, round 6) -- the block is the cull table's: a rank that owns whole blocks reads only
// the table rows its own pre-pass wrote, so no table ever crosses a link.  bx = blocks per frame row, 0 = dealt by rows.
constexpr int kDealBlockLog2 = 6;
struct LfDeal { int n, rank, bx; };
__host__ __device__ inline bool lf_deal_mine(const LfDeal& d, int x, int y) {
  if (d.n <= 1) return true;
  return d.bx > 0 ? ((y >> kDealBlockLog2) * d.bx + (x >> kDealBlockLog2)) % d.n == d.rank : (y >> 3) % d.n == d.rank;
}

// ---- lens camera (round 4): the scene imaged through the prescription --------------------------
// The primary path N-1 .. 0 of a sensor sample (the ray travels -z, against the light), one row per
// interface in the order the ray meets them, the constants of surface_event for EVERY wavelength
// (lf_march.hip pack_program's float arithmetic; lf_lens_camera.hip build_primary_table).  Read by
// k_lens_rays and by k_scene_term's lens mode through the scalar cache.
struct LfPrimaryRow {
  float dzv, curv, ch, c2, sc, h2;
  int kind;        // 0 = curved glass, LF_EV_STOP, LF_EV_FLAT
  int pad;
  float cn22[LF_MAX_LAMBDA], rn2[LF_MAX_LAMBDA], delta[LF_MAX_LAMBDA];
  float fs[LF_MAX_LAMBDA], fo[LF_MAX_LAMBDA], fi[LF_MAX_LAMBDA];
};
struct LfPrimaryDev {
  int n, n_lambda;
  float inv_stop_h;      // 1 / stop_h (correctly rounded)
  float front_zv;        // vertex z of interface 0 (0 by construction; kept for clarity)
  float n_start[LF_MAX_LAMBDA];
  LfPrimaryRow row[LF_MAX_SURFACES];
};
// what k_scene_term's lens mode needs beside the table (kernel argument, by value)
struct LfLensCamArgs {
  int mode;              // 1 = one reference wavelength carries R, G and B; 2 = one ray per wavelength
  int lambda_ref, n_lambda;
  int order_step;        // iteration i of the sample loop takes the march's sample (i * order_step) mod ns_aa
  int xs;                // log2 of the wave tile's pixel stride in x (lf_set_tile_stride)
  int W, G;              // sample_start's SampleSpec (the march's sampling specification)
  float inv_G;
  int sub_bits;
  float inv_sub;
  float pitch, half_w, half_h, pupil_h, vz, geom_norm;
  int mw, mh;
  double exposure;       // scale of the transmitted weight (lf_set_lens_camera)
  double world_per_mm;   // lens millimetres -> scene units
  double z_ref_mm;       // lens-space z that sits at the camera position (the entrance pupil's centre)
  float lambda_rgb[LF_MAX_LAMBDA][3];
};
// k_scene_term's device counters: rays handed to the closest-hit search (camera + shadow + hemisphere
// rays: BVHAccel::total_rays, bvh.h:85,105), primitive tests of closest-hit queries
// (BVHAccel::total_isects, bvh.cpp:211), lens samples started / that left the front element
constexpr int kSceneCounters = 4;

// ---- timing ---------------------------------------------------------------------------------
enum LfKernelId { LFK_MARCH = 0, LFK_FLARE_LAYER, LFK_GHOST_RASTER, LFK_DFT, LFK_FRAME_SETUP,
                  LFK_TONEMAP, LFK_EXCHANGE, LFK_SCENE, LFK_CULL, LFK_CULL_AUDIT, LFK_COUNT };

struct LfTimedLaunch { int kernel; hipEvent_t start, stop; };

// ---- the context ----------------------------------------------------------------------------
// makes ctx->stream wait for an exchange still running on ctx->comm_stream (lf_group.hip)
lf_status lf_comm_join(struct lf_ctx* ctx);

struct lf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  mutable std::string err;

  int W = 0, H = 0, y0 = 0, y1 = 0;
  int H_alloc = 0;                     // rows allocated (H rounded up to 64) for in-place gathers
  int row_period = 1, row_phase = 0;   // tile rows t (8 sensor rows) with t % period == phase -- or, dealt by blocks:
  bool deal_by_block = false;          // lf_set_block_deal: 64 x 64-pixel blocks b (row-major) with b % period == phase
  int ns_aa = 1;
  double flare_radius = 25.0, flare_intensity = 1.0;
  int flare_arithmetic = 0;   // lf_set_flare_arithmetic: 0 auto (exact pow in MT19937 parity mode), 1 exact, 2 fast

  LfApertureDev ap[2];
  // starburst spectrum |DFT2(aperture)| / total_value, aw*aw doubles; twiddles; row pass scratch
  double* spectrum = nullptr;
  double2* twiddle = nullptr;
  double2* dft_rows = nullptr;
  bool spectrum_valid = false;

  LfStarSpectrum star_spec{};
  LfParaxialLens pl{};
  LfCamera cam{};
  bool cam_valid = false;
  LfFlares* flares = nullptr;      // device
  LfGhostList* ghosts = nullptr;   // device
  LfParaxialLens* pl_dev = nullptr;
  bool flares_valid = false;
  double* sun_lights_dev = nullptr;  // staging for more than kMaxSunLightArgs directional lights
  size_t sun_lights_cap = 0;

  double* sample = nullptr;  // W*H*3
  double* ghost = nullptr;   // W*H*3
  double* scene = nullptr;   // W*H*3 or null
  double* star = nullptr;    // W*H*3: raytrace_starburst(x,y) alone (starburst + falloff)
  uint32_t* rgba = nullptr;  // W*H
  uint32_t* rgba_flip = nullptr;  // W*H, allocated by the first lf_save_image_rgba (rows top-down)
  bool ghost_valid = false, sample_valid = false;
  int rgba_y0 = 0, rgba_y1 = 0;  // rows [rgba_y0, rgba_y1) of rgba hold the tonemapped sample buffer

  // jitter
  int jitter_mode = 1;          // 0 = MT19937 table, 1 = counter
  uint64_t jitter_key = 0x1e45f1a4eULL;
  uint32_t* jitter_raw = nullptr;  // W*H*32 raw draws, pixel-major (MT mode)
  bool jitter_table_valid = false;
  uint32_t* jitter_aa_raw = nullptr;  // W*H*2*ns_aa pixel-jitter draws (MT mode), pixel-major
  int jitter_aa_ns = 0;

  // scene term
  LfSceneDev scene_dev{};
  bool scene_valid = false;
  double scene_bmin[3] = {}, scene_bmax[3] = {};   // the BVH root's box (lf_scene_bounds)
  int ns_area_light = 1;          // PathTracer::ns_area_light (pathtracer.h:108; the -l flag)
  LfEnvDev env_dev{};             // PathTracer::envLight (pathtracer.h:119)
  double* env_block = nullptr;    // one allocation behind env_dev's four tables
  double* probe_dev = nullptr;    // 8 doubles: where the single-ray scene probes leave their answer
  bool hemisphere_sample = false; // PathTracer::direct_hemisphere_sample (pathtracer.h:114; the -H flag)
  int samples_per_batch = 32;     // PathTracer::samplesPerBatch default (raytraced_renderer.h:67-81)
  double max_tolerance = 0.05;    // PathTracer::maxTolerance
  int scene_tree_depth = 24;                          // levels of the resident BVH (lf_set_scene): sizes k_scene_lens's stack
  unsigned long long* scene_counters_dev = nullptr;   // kSceneCounters x u64 (lf_get_scene_counters)

  // lens camera (lf_lens_camera.hip): the scene term's sample loop marches each sensor sample's primary
  // path through the prescription instead of calling the pinhole Camera::generate_ray
  int lenscam_mode = 0;               // 0 = off (pinhole), 1 = reference wavelength, 2 = per wavelength
  double lenscam_world_per_mm = 0.001;
  double lenscam_exposure_req = 0.0;  // what lf_set_lens_camera asked for; <= 0: calibrate on the axis
  double lenscam_exposure = 1.0;      // what the kernel uses
  double lenscam_z_ref = 0.0;         // entrance pupil z (mm, lens space)
  float lenscam_aim_margin = 0.0f;    // lf_set_lens_camera_aim: > 0: the lens camera's samples aim at the exit pupil's image
  float lenscam_pupil_h = 0.0f, lenscam_pupil_z = 0.0f, lenscam_geom_norm = 0.0f;   // its own disc (margin > 0)
  bool lenscam_dirty = true;          // table / calibration older than the lens, the mask or the pupil target
  LfPrimaryDev* primary_dev = nullptr;

  // geometric
  LfLensDev lens{};
  bool lens_valid = false, sun_valid = false;
  // the sampling specification's two coherence parameters (round 4 defaults, measured on the bench frame:
  // profiles/r04_tile_stride.json): 64 x 64 pupil sub-cells per stratum, shared by the 64 pixels of a wave
  // whose columns are 8 apart -- 108 ms and a tile correlation of 7.4 where rounds 1-3 (4 x 4 sub-cells,
  // adjacent pixels) had 110 ms and 37.7
  int march_sub_bits = 6;      // lf_set_pupil_subcells
  int march_xstride_log2 = 3;  // lf_set_tile_stride: the lanes of a wave take pixels 2^this apart in x
  float sensor_w_mm = 36.0f;
  int raw_n = 0, raw_stop = -1;   // the prescription as handed to lf_set_lens (for lf_paraxial_efl)
  float raw_radius[LF_MAX_SURFACES] = {}, raw_thickness[LF_MAX_SURFACES] = {};
  float raw_ior[LF_MAX_LAMBDA * LF_MAX_SURFACES] = {};
  float raw_semi_ap[LF_MAX_SURFACES] = {};
  float pupil_target_h = 0.0f, pupil_target_z = 0.0f;   // lf_set_pupil_target; h <= 0: the rear element
  bool ghost_accumulate = false;                          // lf_set_ghost_accumulate
  LfLensDev* lens_dev = nullptr;
  LfPairsDev pairs{};
  LfPairsDev* pairs_dev = nullptr;
  unsigned long long* counters_dev = nullptr;  // kMarchCounterSlots x u64
  unsigned long long* accum = nullptr;         // W*H_alloc*3 fixed-point partial sums (split launches)
  unsigned long long* tail_acc = nullptr;      // kMarchTailTilesMax x 192 sums + arrivals of the culled march's split tail (lf_cull.hip)
  int* tail_done = nullptr;
  int march_tail_tiles = -1, march_tail_groups = -1;   // lf_test_knob: the tail's size (-1: one round of resident workgroups) / split
  unsigned char* prog_dev = nullptr;           // the packed program: headers, then records (lf_march.hip pack_program)
  size_t prog_cap = 0, prog_rec_off = 0, prog_wrec_off = 0, prog_seq_off = 0;   // bytes; offsets of the records / weight records / pair sequences
  // path culling (lf_cull.hip): 0 = off (k_march walks every path of every sample), 1 = on, the table is reused
  // while its inputs (lens, pairs, sun, frame, pupil disc, mask, strata) are unchanged, 2 = on, rebuilt at every launch
  int march_cull = 1;
  unsigned long long* cull_dev = nullptr;
  size_t cull_cap = 0;                         // entries allocated
  unsigned* cull_list[2] = {nullptr, nullptr}; // work lists of the pre-pass levels (per path: list_stride entries)
  size_t cull_list_cap[2] = {0, 0};
  unsigned* cull_counts = nullptr;             // [levels][kCullMaxPaths] list lengths
  int cull_m = 1;                              // table cells per axis inside one stratum
  // the pre-pass shared between the ranks of a multi-GPU frame (lf_set_cull_share / lf_comm_share_cull):
  int cull_share_rank = 0, cull_share_n = 1;   // this context builds the rows of the blocks b with b % n == rank
  int cull_share_how = 0;                      // 1: the library's RCCL communicator completes the table inside lf_trace_ghosts;
                                               // 2: the host does (lf_cull_prepare -> its own all-gather -> lf_cull_commit);
                                               // 3: nobody does -- the frame is dealt by blocks (lf_set_block_deal), a rank needs its own rows only
  int cull_share_nb = 0, cull_share_n_resident = 1;   // rows per slab and slabs of the RESIDENT table (0 / 1: not shared)
  bool cull_own_rows_only = false;             // ... of which only this rank's slab is built (the frame dealt by blocks: nobody reads the others)
  uint64_t cull_hash_pending = 0;              // of the slab lf_cull_prepare built (the host's exchange is outstanding)
  bool cull_prepare_only = false;              // (lf_cull_prepare is inside lfk_march)
  bool cull_fresh = false;                     // lf_cull_commit just completed the table: the next launch takes it even in mode 2
  unsigned long long* cull_popc_dev = nullptr; // one u64: set bits of the table (k_cull_popcount)
  double cull_started_fraction = 0.0;          // of all (block, cell, path) combinations, what the resident table starts
  // Above this the culled march loses to the path tree: it marches every started path on its own and with its
  // weight, the tree shares legs and lets rays die early (measured crossover on the 1080p frame: a sun of 0.2 rad
  // starts 17 % and ties, profiles/r05_march_variants.txt).  lf_set_march_culling's mode stays what it is; the
  // launch just takes the other kernel.
  double cull_max_fraction = 0.10;             // + 1.6 / paths: see lfk_march
  uint64_t cull_hash = 0;                      // of the inputs the resident table was built from (0 = none)
  int cull_bx = 0, cull_by = 0, cull_cells = 0, cull_G = 0, cull_P = 0, cull_blk_log2 = 6;
  // The pre-pass's rules.  What ships is ONE set (lf_cull.hip k_cull_level, constants in the kernel); a test may install
  // another through lf_test_knob ("cull_strict", ...: the rules round 5 replaced, kept to show what the audit is for) --
  // the table is then built by k_cull_level_general, which reads them from here.
  struct CullRules {
    float margin = 1.25f;        // footprint inflation
    int strict = 1, strict_lost = 1, slack_mode = 1, keep_partial = 0, disable = 0;
    float lobe_k = 1.2f, lost_rel = 0.5f, lost_abs = 0.002f;
  } cull_rules;
  bool cull_rules_custom = false;              // a test installed rules / asked for the general kernel (lf_test_knob)
  bool cull_force = false;                     // lf_test_knob("cull_force"): the culled kernel whatever the table starts
  bool cull_weights_first = false;             // lf_test_knob("cull_weights_first"): k_march_cull<K, true>
  bool cull_no_prefix = false;                 // lf_test_knob("cull_no_prefix"): every started path marched alone from the sensor (round 5)
  int scene_compact = -1;                      // lf_test_knob("scene_compact"): -1 = by the tree's size, 0 / 1 forced
  bool comm_force_exchange = false;            // lf_test_knob("comm_force_exchange"): the collectives also with one rank
  int scene_lens_strided = -1;                 // lf_test_knob("scene_lens_strided"): k_scene_lens's wave tile: -1 by the tree's size, 0 / 1
  bool bvh_median = false;                     // lf_test_knob("bvh_median"): the round-2 median-split tree (A/B of the SAH tree)
  int bvh_leaf_max = 2;                        // lf_test_knob("bvh_leaf"): primitives per leaf at most (1 .. 4)
  // the audit of what the table drops (lf_cull.hip k_cull_audit): rays per dropped (block, cell, path), 0 = off
  int cull_audit_density = 1;
  unsigned long long* cull_audit_dev = nullptr;   // {rays, lit} of the table being completed
  unsigned long long cull_audit_rays = 0, cull_audit_lit = 0;   // since lf_reset_counters
  int cull_audit_tripped = 0;                  // launches since lf_reset_counters whose table an audit ray refuted
  uint64_t cull_bad_hash = 0;                  // the resident table was refuted: its launches march everything
  int cull_reason = 0;                         // why the last launch did (not) cull: lf_cull_reason
  int cull_chunks = 1;                         // launches the last lf_trace_ghosts split its selection into (> 64 paths: 2)
  uint64_t cull_audit_seq = 0;                 // tables audited by this context: keys the audit's rays
  bool lens_lambda_monotonic = true;           // every glass disperses the same way along the wavelength columns (lf_derive_lens)
  unsigned cull_occ[kCullOcc] = {};            // occupancy of the stop mask (host, lf_set_aperture)
  uint64_t mask_generation = 0;                // bumped by lf_set_aperture(STARBURST)
  bool last_march_culled = false;              // what the last lf_trace_ghosts ran
  int march_k = 1;                             // wavelengths (rays per lane) that walk together
  int march_fix_bits = 36;                     // the last launch's fixed-point exponent (lf_get_march_fix_bits)
  bool events_dirty = true;

  // multi-GPU (lf_group.hip): the communicator this context belongs to, staging for the exchange
  void* comm = nullptr;          // ncclComm_t, or null (single GPU / rehearsal group)
  int comm_nranks = 1, comm_rank = 0;
  double* comm_stage = nullptr;  // send [groups][e] followed by recv [world][groups][e]
  size_t comm_stage_cap = 0;     // doubles
  // lf_comm_gather_async: the exchange of frame k runs on its own stream while frame k + 1 is marched
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev_main = nullptr, comm_ev_pack = nullptr, comm_ev_done = nullptr;
  hipEvent_t comm_ev_table = nullptr, comm_ev_table_done = nullptr;   // the shared cull table's all-gather (lf_comm_allgather_u64_inplace)
  bool comm_pending = false;     // an exchange on comm_stream the main stream has not yet waited for
  bool comm_f32 = false;         // lf_comm_set_exchange_precision(32): the tile rows travel as floats
  // A communicator call can block the calling HOST thread for good (ncclCommInitRank, the first collective's
  // enqueue: a peer that never joins).  A host that gives up on such a call from another thread says so with
  // lf_comm_poison: from then on the blocked call -- should it ever return -- publishes NOTHING into the context
  // (a communicator that arrives late is aborted where it stands), every lf_comm_* call is refused, and lf_destroy
  // leaks the context instead of freeing what the blocked thread still stands on.
  std::atomic<int> comm_busy{0};         // threads inside a call that may block
  std::atomic<bool> comm_poisoned{false};
  std::mutex comm_mu;                    // publication of comm / comm_nranks / comm_rank / comm_pending

  bool timing = false;
  std::vector<LfTimedLaunch> timed;       // launches not folded yet (bounded, see lf_api.hip)
  std::vector<hipEvent_t> event_pool;     // recycled events
  double timed_ms[LFK_COUNT] = {};        // folded totals since lf_timing_reset
  int timed_n[LFK_COUNT] = {};
};

// implemented in lf_api.hip
lf_status lf_fail(const lf_ctx* ctx, lf_status st, const std::string& msg);
// (stream: where the timed work runs; null = the context's main stream)
hipEvent_t lf_timing_begin(lf_ctx* ctx, int kernel, hipStream_t stream = nullptr);
void lf_timing_end(lf_ctx* ctx, int kernel, hipEvent_t start, hipStream_t stream = nullptr);

#define LF_HIP(ctx, expr)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return lf_fail(ctx, LF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// candidate selection of the march (the contract, the same expression in oracle/lf_geo_oracle.c): d.s above
// 1 - 1.0625 (1 - cos alpha) - 4e-7 MAY lie inside the lobe.  The 1/16 margin is relative, the 4e-7 absolute: the float
// dot product of two unit vectors is only good to ~2e-7, so a relative margin alone loses part of a sub-milliradian
// sun's lobe (1 - cos(0.8 mrad) = 3.2e-7).  Rounded down: float(thr) never exceeds thr.
inline float lf_march_lobe_thr(const LfLensDev& L) {
  const double thr = 1.0 - (1.0625 / (double)L.sun_inv_one_minus_cos) * (1.0 + 1e-6) - 4e-7;
  float t = (float)thr;
  if ((double)t > thr) t = std::nextafterf(t, -2.0f);
  return t;
}

// the flare layer calls the reference's own pow() (exact) or its cheaper equivalents (fast): DESIGN.md section 3
inline LfDeal lf_deal_of(const lf_ctx* ctx) {
  return LfDeal{ctx->row_period, ctx->row_phase, ctx->deal_by_block ? (ctx->W + (1 << kDealBlockLog2) - 1) >> kDealBlockLog2 : 0};
}
inline bool lf_flare_exact(const lf_ctx* ctx) {
  return ctx->flare_arithmetic == 1 || (ctx->flare_arithmetic == 0 && ctx->jitter_mode == 0);
}
// kernels launchers (lf_flare_kernels.hip)
lf_status lfk_aperture_stats(lf_ctx* ctx, int slot);
lf_status lfk_build_spectrum(lf_ctx* ctx);
lf_status lfk_frame_setup(lf_ctx* ctx, const LfSunLightArgs* lights_arg, const double* lights_dev,
                          int n_lights, bool project);
lf_status lfk_ghost_raster(lf_ctx* ctx);
// single-shot forms of the reference's public helper members (lf_draw_ghost ... lf_irradiance_falloff)
lf_status lfk_draw_ghost(lf_ctx* ctx, int channel, float r1, float r2, int bbox[4]);
lf_status lfk_raster_triangle(lf_ctx* ctx, const float v[12], const double colour[3], int bbox[4]);
lf_status lfk_fill_pixel(lf_ctx* ctx, const float v[12], int x, int y, const double colour[3]);
lf_status lfk_shift_vertex(lf_ctx* ctx, float x, float y, float scale, float shift_amount, double out[2]);
lf_status lfk_compute_phase(lf_ctx* ctx, int flare, double u, double v, double out[4]);
lf_status lfk_irradiance_falloff(lf_ctx* ctx, int x, int y, double radius, double out[3]);
lf_status lfk_flare_layer(lf_ctx* ctx);
lf_status lfk_tonemap(lf_ctx* ctx, int ya, int yb);
lf_status lfk_flip_rows(lf_ctx* ctx, uint32_t* out_dev);
// lf_march.hip
lf_status lfk_march(lf_ctx* ctx, int spp, uint64_t key);
lf_status lfk_lens_rays(lf_ctx* ctx, int lambda, int n, const float* d_xy, const float* d_uv,
                        float* d_out);
// lf_lens_camera.hip
lf_status lf_lenscam_prepare(lf_ctx* ctx);            // table + calibration up to date (no-op when clean)
lf_status lf_upload_primary_table(lf_ctx* ctx);       // LfPrimaryDev from ctx->lens (k_lens_rays needs it too)
void lf_fill_lenscam_args(const lf_ctx* ctx, LfLensCamArgs* a);
lf_status lf_build_march_tables(lf_ctx* ctx, std::vector<LfEventRow>& rows, std::vector<int>& skip);
lf_status lfk_native_sqrt(lf_ctx* ctx, const float* d_x, float* d_y, size_t n);
lf_status lfk_native_rcp(lf_ctx* ctx, const float* d_x, float* d_y, size_t n);
void lf_apply_pupil_target(lf_ctx* ctx);
int lf_march_fix_bits(const LfLensDev& L, int n_paths, int spp);
// lf_cull.hip (a = the launch's arguments as lfk_march set them up: lf_march_common.h)
namespace lfm { struct MarchArgs; }
bool lf_cull_applies(const lf_ctx* ctx, int G);
int lf_cull_reason_of(const lf_ctx* ctx, int G);                       // lf_cull_reason: LF_CULL_APPLIED or why not
int lf_cull_block_log2(const lf_ctx* ctx, int spp, int n_lambda);      // log2 of a cull block's side in pixels (-1: none applies)
lf_status lfk_cull_prepass(lf_ctx* ctx, int G, int spp);
lf_status lfk_cull_finish(lf_ctx* ctx, uint64_t hash);
lf_status lfk_cull_prepare(lf_ctx* ctx, int spp);   // lf_march.hip          // the table is complete: count what it starts
// lf_group.hip: in-place all-gather of equal slabs of u64 on the communicator's stream, ordered after what the main
// stream has queued and before what it queues next
lf_status lf_comm_allgather_u64_inplace(lf_ctx* ctx, unsigned long long* base, size_t count_per_rank);
lf_status lfk_march_culled(lf_ctx* ctx, const lfm::MarchArgs& a, size_t blocks, size_t dyn_lds);
void lf_derive_lens(lf_ctx* ctx, int n, int stop, int n_lambda, const float* radius,
                    const float* thickness, const float* ior, const float* semi_ap,
                    float sensor_w_mm);
