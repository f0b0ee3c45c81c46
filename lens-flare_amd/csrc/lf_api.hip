// lf_api.hip -- the extern "C" entry points of liblensflare_hip.so (include/lensflare.h).
// Host-side orchestration only: validation, device buffers, stream/event handling.  Every
// number that reaches a sensor pixel is computed by the gfx950 kernels in lf_flare_kernels.hip /
// lf_march.hip; there is no CPU fallback -- without a device lf_create fails.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lf_internal.h"
#include "../host/lf_collada.h"

lf_status lf_fail(const lf_ctx* ctx, lf_status st, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return st;
}

// HIP-event timing.  Events come from a pool and the list of pending launches is bounded: once it
// holds kTimedCap entries the finished ones are folded into per-kernel totals and their events go
// back to the pool, so a long timed run neither grows without bound nor creates events per launch.
static constexpr size_t kTimedCap = 512;

static hipEvent_t timing_event(lf_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// fold every launch whose stop event has completed (all of them if `wait`)
static void timing_fold(lf_ctx* ctx, bool wait) {
  size_t keep = 0;
  for (size_t i = 0; i < ctx->timed.size(); i++) {
    LfTimedLaunch& t = ctx->timed[i];
    const bool done = wait ? hipEventSynchronize(t.stop) == hipSuccess : hipEventQuery(t.stop) == hipSuccess;
    float ms = 0.f;
    if (done && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      ctx->timed_ms[t.kernel] += ms;
      ctx->timed_n[t.kernel]++;
      ctx->event_pool.push_back(t.start);
      ctx->event_pool.push_back(t.stop);
    } else {
      ctx->timed[keep++] = t;
    }
  }
  ctx->timed.resize(keep);
}

hipEvent_t lf_timing_begin(lf_ctx* ctx, int kernel, hipStream_t stream) {
  if (!ctx->timing) return nullptr;
  (void)kernel;
  if (ctx->timed.size() >= kTimedCap) timing_fold(ctx, false);
  if (ctx->timed.size() >= kTimedCap) timing_fold(ctx, true);
  hipEvent_t e = timing_event(ctx);
  if (e) (void)hipEventRecord(e, stream ? stream : ctx->stream);
  return e;
}

void lf_timing_end(lf_ctx* ctx, int kernel, hipEvent_t start, hipStream_t stream) {
  if (!ctx->timing || !start) return;
  hipEvent_t e = timing_event(ctx);
  if (!e) { ctx->event_pool.push_back(start); return; }
  (void)hipEventRecord(e, stream ? stream : ctx->stream);
  ctx->timed.push_back(LfTimedLaunch{kernel, start, e});
}

namespace {

const char* kKernelNames[LFK_COUNT] = {"march", "flare_layer", "ghost_raster", "dft",
                                       "frame_setup", "tonemap", "exchange", "scene_term", "cull_prepass", "cull_audit"};

// the reference's hard-coded prescription (pathtracer.cpp:541-556); literals narrowed to float
// where the reference narrows them
void default_paraxial_lens(LfParaxialLens& L) {
  static const double th[9] = {7.700, 1.850, 3.520, 1.850, 4.180, 3.000, 1.850, 7.270, 83.91};
  static const double rgb[3][9] = {{1.652, 1.5991, 1, 1.6396, 1, 1, 1.5776, 1.68990, 1},
                                   {1.652, 1.6113, 1, 1.65, 1, 1, 1.5885, 1.6999, 1},
                                   {1.652, 1.6164, 1, 1.6542, 1, 1, 1.5930, 1.7040, 1}};
  static const double radii[9] = {30.810, -89.350, 580.380, -80.630, 28.340, 0, 0, 32.190, -52.990};
  std::memset(&L, 0, sizeof(L));
  L.n = 9;
  L.stop = 5;
  for (int k = 0; k < 9; k++) {
    L.thickness[k] = (float)th[k];
    for (int c = 0; c < 3; c++) L.ior[c][k] = (float)rgb[c][k];
    L.curvature[k] = radii[k] == 0 ? 0.0f : (float)(1 / radii[k]);
  }
  L.clip = 11.6;
  L.recast_pos = 11.6f;
  L.recast_neg = -11.5f;
  L.marginal = 14.5f;
}

template <typename T>
lf_status dev_alloc(lf_ctx* ctx, T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) return LF_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) return lf_fail(ctx, LF_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return LF_OK;
}

// std::mt19937 (util/random_util.h:10-14), written from the algorithm's definition
struct Mt19937 {
  uint32_t s[624];
  int idx;
  explicit Mt19937(uint32_t seed) {
    s[0] = seed;
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; k++) {
        uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
        s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = s[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

void free_frame_buffers(lf_ctx* ctx) {
  if (ctx->sample) (void)hipFree(ctx->sample);
  if (ctx->ghost) (void)hipFree(ctx->ghost);
  if (ctx->star) (void)hipFree(ctx->star);
  if (ctx->scene) (void)hipFree(ctx->scene);
  if (ctx->rgba) (void)hipFree(ctx->rgba);
  if (ctx->rgba_flip) (void)hipFree(ctx->rgba_flip);
  ctx->rgba_flip = nullptr;
  if (ctx->jitter_raw) (void)hipFree(ctx->jitter_raw);
  if (ctx->jitter_aa_raw) (void)hipFree(ctx->jitter_aa_raw);
  ctx->jitter_aa_raw = nullptr;
  if (ctx->accum) (void)hipFree(ctx->accum);  // sized by the frame
  ctx->accum = nullptr;
  ctx->sample = ctx->ghost = ctx->scene = ctx->star = nullptr;
  ctx->rgba = nullptr;
  ctx->jitter_raw = nullptr;
  ctx->jitter_table_valid = ctx->ghost_valid = ctx->sample_valid = false;
  ctx->rgba_y0 = ctx->rgba_y1 = 0;
}

lf_status upload_paraxial(lf_ctx* ctx) {
  LF_HIP(ctx, hipMemcpyAsync(ctx->pl_dev, &ctx->pl, sizeof(LfParaxialLens), hipMemcpyHostToDevice,
                             ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LF_OK;
}

}  // namespace

extern "C" {

int lf_abi_version(void) { return LF_ABI_VERSION; }

const char* lf_last_error(const lf_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// lf_test_knob's process-wide defaults (see there)
static std::mutex g_knob_mu;
static std::vector<std::pair<std::string, double>> g_knob_defaults;

lf_status lf_create(lf_ctx** out, int device) {
  if (!out) return LF_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LF_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return LF_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LF_ERR_NO_DEVICE;
  {
    // the library carries gfx950 code objects only: any other architecture would fail at the first
    // launch with an opaque HIP error (LF_ALLOW_ANY_ARCH=1: for experiments with a fat binary)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LF_ERR_NO_DEVICE;
    bool any_arch = false;
#ifdef LF_EXPERIMENTS
    any_arch = std::getenv("LF_ALLOW_ANY_ARCH") != nullptr;
#endif
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !any_arch) return LF_ERR_NO_DEVICE;
  }
  lf_ctx* ctx = new lf_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return LF_ERR_HIP;
  }
  ctx->own_stream = true;
#ifdef LF_EXPERIMENTS
  // (changes the sampling pattern; the oracles / tests use the default, 6)
  if (const char* sb = std::getenv("LF_MARCH_SUB_BITS")) {
    int v = std::atoi(sb);
    if (v >= 0 && v <= 8) ctx->march_sub_bits = v;
  }
#endif
  default_paraxial_lens(ctx->pl);
  bool ok = hipMalloc((void**)&ctx->flares, sizeof(LfFlares)) == hipSuccess &&
            hipMalloc((void**)&ctx->ghosts, sizeof(LfGhostList)) == hipSuccess &&
            hipMalloc((void**)&ctx->pl_dev, sizeof(LfParaxialLens)) == hipSuccess &&
            hipMalloc((void**)&ctx->lens_dev, sizeof(LfLensDev)) == hipSuccess &&
            hipMalloc((void**)&ctx->pairs_dev, sizeof(LfPairsDev)) == hipSuccess &&
            hipMalloc((void**)&ctx->counters_dev, kMarchCounterSlots * sizeof(unsigned long long)) == hipSuccess &&
            hipMalloc((void**)&ctx->scene_counters_dev, kSceneCounters * sizeof(unsigned long long)) == hipSuccess &&
            hipMalloc((void**)&ctx->primary_dev, sizeof(LfPrimaryDev)) == hipSuccess;
  for (int s = 0; s < 2 && ok; s++)
    ok = hipMalloc((void**)&ctx->ap[s].stats, sizeof(lf_aperture_stats)) == hipSuccess;
  if (!ok) { lf_destroy(ctx); return LF_ERR_OOM; }
  (void)hipMemset(ctx->flares, 0, sizeof(LfFlares));
  (void)hipMemset(ctx->ghosts, 0, sizeof(LfGhostList));
  (void)hipMemset(ctx->counters_dev, 0, kMarchCounterSlots * sizeof(unsigned long long));
  (void)hipMemset(ctx->scene_counters_dev, 0, kSceneCounters * sizeof(unsigned long long));
  if (upload_paraxial(ctx) != LF_OK) { lf_destroy(ctx); return LF_ERR_HIP; }
  {
    std::vector<std::pair<std::string, double>> defaults;
    { std::lock_guard<std::mutex> lock(g_knob_mu); defaults = g_knob_defaults; }
    for (const auto& kv : defaults) (void)lf_test_knob(ctx, kv.first.c_str(), kv.second);
  }
  *out = ctx;
  return LF_OK;
}

lf_status lf_destroy(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  // a communicator call the host gave up on (lf_comm_poison) may still be blocked in another thread, standing on
  // this context: nothing is freed then -- the context is LEAKED on purpose (the process is on its way out anyway)
  if (ctx->comm_poisoned.load() && ctx->comm_busy.load() > 0) return LF_ERR_STATE;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  (void)lf_comm_destroy(ctx);
  if (ctx->comm_stage) (void)hipFree(ctx->comm_stage);
  free_frame_buffers(ctx);
  for (auto& t : ctx->timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (int s = 0; s < 2; s++) {
    if (ctx->ap[s].texels) (void)hipFree(ctx->ap[s].texels);
    if (ctx->ap[s].stats) (void)hipFree(ctx->ap[s].stats);
  }
  void* ptrs[] = {ctx->spectrum, ctx->twiddle, ctx->dft_rows, ctx->flares, ctx->ghosts, ctx->pl_dev,
                  ctx->lens_dev, ctx->pairs_dev, ctx->counters_dev, ctx->accum,
                  ctx->prog_dev, ctx->sun_lights_dev,
                  ctx->scene_dev.nodes, ctx->scene_dev.prims, ctx->scene_dev.normals, ctx->scene_dev.materials,
                  ctx->scene_dev.lights, ctx->env_block, ctx->probe_dev, ctx->scene_counters_dev,
                  ctx->primary_dev, ctx->cull_dev, ctx->cull_list[0], ctx->cull_list[1], ctx->cull_counts, ctx->cull_popc_dev,
                  ctx->tail_acc, ctx->tail_done};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return LF_OK;
}

lf_status lf_set_stream(lf_ctx* ctx, void* hip_stream) {
  if (!ctx) return LF_ERR_INVALID;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  ctx->own_stream = false;
  return LF_OK;
}

lf_status lf_synchronize(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  lf_status js = lf_comm_join(ctx);   // an exchange still running on the second stream
  if (js != LF_OK) return js;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LF_OK;
}

// (re)allocate the frame-sized buffers for `rows` rows; content is lost
static lf_status alloc_frame_buffers(lf_ctx* ctx, int rows) {
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) { LF_HIP(ctx, hipStreamSynchronize(ctx->comm_stream)); ctx->comm_pending = false; }
  free_frame_buffers(ctx);
  ctx->H_alloc = rows;
  size_t n = (size_t)ctx->W * ctx->H_alloc;
  lf_status st;
  if ((st = dev_alloc(ctx, &ctx->sample, 3 * n)) != LF_OK) return st;
  if ((st = dev_alloc(ctx, &ctx->ghost, 3 * n)) != LF_OK) return st;
  if ((st = dev_alloc(ctx, &ctx->star, 3 * n)) != LF_OK) return st;
  if ((st = dev_alloc(ctx, &ctx->rgba, n)) != LF_OK) return st;
  LF_HIP(ctx, hipMemsetAsync(ctx->sample, 0, 3 * n * sizeof(double), ctx->stream));
  LF_HIP(ctx, hipMemsetAsync(ctx->ghost, 0, 3 * n * sizeof(double), ctx->stream));
  LF_HIP(ctx, hipMemsetAsync(ctx->star, 0, 3 * n * sizeof(double), ctx->stream));
  LF_HIP(ctx, hipMemsetAsync(ctx->rgba, 0, n * sizeof(uint32_t), ctx->stream));
  return LF_OK;
}

// rows the frame buffers need so that every group of `period` 8-row tile rows is addressable
// (the tile-row exchange of lf_gather / sharding.py views the frame as [groups][period][tile row])
static int padded_rows(int H, int period) {
  const int ntrows = (H + 7) / 8;
  return (ntrows + period - 1) / period * period * 8;
}

lf_status lf_set_frame(lf_ctx* ctx, int width, int height) {
  if (!ctx) return LF_ERR_INVALID;
  if (width <= 0 || height <= 0 || width > (1 << 15) || height > (1 << 15))
    return lf_fail(ctx, LF_ERR_INVALID, "frame size out of range");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  ctx->W = width; ctx->H = height; ctx->y0 = 0; ctx->y1 = height;
  ctx->row_period = 1; ctx->row_phase = 0;
  // padded for every interleave period up to 8 (ceil(n/p) p <= n + p - 1); larger periods
  // re-allocate in lf_set_row_interleave
  return alloc_frame_buffers(ctx, ((height + 7) / 8 + 7) * 8);
}

lf_status lf_set_band(lf_ctx* ctx, int y0, int y1) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_band before lf_set_frame");
  if (y0 < 0 || y1 > ctx->H || y0 > y1) return lf_fail(ctx, LF_ERR_INVALID, "band out of range");
  ctx->y0 = y0; ctx->y1 = y1;
  return LF_OK;
}

lf_status lf_set_row_interleave(lf_ctx* ctx, int phase, int period) {
  if (!ctx) return LF_ERR_INVALID;
  if (period < 1 || phase < 0 || phase >= period)
    return lf_fail(ctx, LF_ERR_INVALID, "row interleave: need 0 <= phase < period");
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_row_interleave before lf_set_frame");
  if (padded_rows(ctx->H, period) > ctx->H_alloc) {
    // the tile-row exchange needs ceil(tile rows / period) * period tile rows: grow the buffers
    // (their content is lost -- set the interleave before rendering)
    LF_HIP(ctx, hipSetDevice(ctx->device));
    lf_status st = alloc_frame_buffers(ctx, padded_rows(ctx->H, period));
    if (st != LF_OK) return st;
  }
  ctx->row_phase = phase; ctx->row_period = period;
  if (ctx->deal_by_block) { ctx->deal_by_block = false; ctx->cull_hash = 0; if (ctx->cull_share_how == 3) { ctx->cull_share_how = 0; ctx->cull_share_n = 1; ctx->cull_share_rank = 0; } }
  return LF_OK;
}

// The frame dealt by BLOCKS of 64 x 64 pixels (round 6): block b (row-major) belongs to rank b % nranks.  The block is the
// cull table's, so a rank's march reads only the table rows its own pre-pass wrote: the pre-pass, its audit and the march
// all shrink with the number of ranks and no table crosses a link.
lf_status lf_set_block_deal(lf_ctx* ctx, int rank, int nranks) {
  if (!ctx) return LF_ERR_INVALID;
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return lf_fail(ctx, LF_ERR_INVALID, "block deal: need 0 <= rank < nranks <= 64");
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_block_deal before lf_set_frame");
  if (ctx->cull_share_how == 1 || ctx->cull_share_how == 2)
    return lf_fail(ctx, LF_ERR_STATE, "lf_set_block_deal: the cull table is shared between ranks (lf_comm_share_cull / lf_set_cull_share): "
                                      "a frame dealt by blocks shares nothing");
  ctx->row_phase = rank; ctx->row_period = nranks;
  ctx->deal_by_block = nranks > 1;
  ctx->cull_share_how = nranks > 1 ? 3 : 0;       // the pre-pass builds the rows of this rank's blocks and nobody else's
  ctx->cull_share_rank = nranks > 1 ? rank : 0;
  ctx->cull_share_n = nranks;
  ctx->cull_hash = 0; ctx->cull_hash_pending = 0; ctx->cull_fresh = false;
  return LF_OK;
}

lf_status lf_set_params(lf_ctx* ctx, int ns_aa, double flare_radius, double flare_intensity) {
  if (!ctx || ns_aa < 0) return LF_ERR_INVALID;
  ctx->ns_aa = ns_aa;
  ctx->flare_radius = flare_radius;
  ctx->flare_intensity = flare_intensity;
  return LF_OK;
}

lf_status lf_set_flare_arithmetic(lf_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return LF_ERR_INVALID;
  ctx->flare_arithmetic = mode;
  return LF_OK;
}

lf_status lf_set_aperture(lf_ctx* ctx, lf_aperture_slot slot, const float* texels, int width,
                          int height) {
  if (!ctx || !texels || (slot != LF_APERTURE_STARBURST && slot != LF_APERTURE_GHOST))
    return LF_ERR_INVALID;
  if (width <= 0 || height <= 0 || width > 4096 || height > 4096)
    return lf_fail(ctx, LF_ERR_INVALID, "aperture size out of range (1..4096)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LfApertureDev& a = ctx->ap[slot];
  a.valid = false;
  lf_status st;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // kernels in flight may read the old texels / spectrum
  if ((st = dev_alloc(ctx, &a.texels, (size_t)width * height)) != LF_OK) return st;
  a.w = width; a.h = height;
  LF_HIP(ctx, hipMemcpyAsync(a.texels, texels, sizeof(float) * (size_t)width * height,
                             hipMemcpyHostToDevice, ctx->stream));
  if ((st = lfk_aperture_stats(ctx, slot)) != LF_OK) return st;
  {
    // how far from the centre the mask is open (a sampling parameter for lf_aim_at_exit_pupil, not pixel
    // arithmetic): the far corner of the outermost texel > 0, as the march maps texels onto [-h, h]^2
    double r2max = 0.0;
    const double hw = 0.5 * width, hh = 0.5 * height;
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++)
        if (texels[(size_t)y * width + x] > 0.0f) {
          const double ex = std::max(std::fabs(x - hw), std::fabs(x + 1 - hw)) / hw;
          const double ey = std::max(std::fabs(y - hh), std::fabs(y + 1 - hh)) / hh;
          r2max = std::max(r2max, ex * ex + ey * ey);
        }
    a.open_radius = std::sqrt(r2max);
  }
  a.valid = true;
  if (slot == LF_APERTURE_STARBURST) {
    // occupancy of the stop mask for the march's cull pre-pass (lf_cull.hip): which of kCullOcc x kCullOcc cells
    // of the mask holds a texel > 0 (a texel belongs to every cell it touches)
    for (int r = 0; r < kCullOcc; r++) ctx->cull_occ[r] = 0u;
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++)
        if (texels[(size_t)y * width + x] > 0.0f) {
          const int cx0 = (int)((long long)x * kCullOcc / width), cx1 = (int)(((long long)(x + 1) * kCullOcc - 1) / width);
          const int cy0 = (int)((long long)y * kCullOcc / height), cy1 = (int)(((long long)(y + 1) * kCullOcc - 1) / height);
          for (int cy = cy0; cy <= cy1 && cy < kCullOcc; cy++)
            for (int cx = cx0; cx <= cx1 && cx < kCullOcc; cx++) ctx->cull_occ[cy] |= 1u << cx;
        }
    ctx->mask_generation++;
    ctx->spectrum_valid = false;
    ctx->lenscam_dirty = true;   // the stop mask is part of the lens camera's exposure calibration
    size_t rows = a.host_stats.max_y >= a.host_stats.min_y
                      ? (size_t)(a.host_stats.max_y - a.host_stats.min_y + 1) : 1;
    if ((st = dev_alloc(ctx, &ctx->spectrum, (size_t)width * width)) != LF_OK) return st;
    if ((st = dev_alloc(ctx, &ctx->twiddle, (size_t)width)) != LF_OK) return st;
    if ((st = dev_alloc(ctx, &ctx->dft_rows, rows * width)) != LF_OK) return st;
    if ((st = lfk_build_spectrum(ctx)) != LF_OK) return st;
  }
  return LF_OK;
}

lf_status lf_get_aperture_stats(lf_ctx* ctx, lf_aperture_slot slot, lf_aperture_stats* out) {
  if (!ctx || !out || (slot != 0 && slot != 1)) return LF_ERR_INVALID;
  if (!ctx->ap[slot].valid) return lf_fail(ctx, LF_ERR_STATE, "aperture not set");
  *out = ctx->ap[slot].host_stats;
  return LF_OK;
}

lf_status lf_set_paraxial_lens(lf_ctx* ctx, int n, int stop_index, const float* thickness,
                               const float* curvature, const float* ior_rgb) {
  if (!ctx) return LF_ERR_INVALID;
  if (!thickness || !curvature || !ior_rgb) {
    default_paraxial_lens(ctx->pl);
  } else {
    if (n < 2 || n > LF_MAX_SURFACES || stop_index < 0 || stop_index >= n)
      return lf_fail(ctx, LF_ERR_INVALID, "paraxial lens: bad n / stop index");
    LfParaxialLens L;
    default_paraxial_lens(L);
    L.n = n; L.stop = stop_index;
    for (int k = 0; k < n; k++) {
      L.thickness[k] = thickness[k];
      L.curvature[k] = curvature[k];
      for (int c = 0; c < 3; c++) L.ior[c][k] = ior_rgb[c * n + k];
    }
    ctx->pl = L;
  }
  ctx->ghost_valid = false;
  return upload_paraxial(ctx);
}

lf_status lf_set_camera(lf_ctx* ctx, const double c2w[9], const double pos[3], double hfov_deg,
                        double vfov_deg) {
  if (!ctx || !c2w || !pos) return LF_ERR_INVALID;
  std::memcpy(ctx->cam.c2w, c2w, 9 * sizeof(double));
  std::memcpy(ctx->cam.pos, pos, 3 * sizeof(double));
  ctx->cam.hfov_deg = hfov_deg;
  ctx->cam.vfov_deg = vfov_deg;
  ctx->cam_valid = true;
  return LF_OK;
}

lf_status lf_find_sun_pos(lf_ctx* ctx, const double* lights, int n_lights) {
  if (!ctx || n_lights < 0 || (n_lights > 0 && !lights)) return LF_ERR_INVALID;
  if (!ctx->cam_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_find_sun_pos before lf_set_camera");
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_find_sun_pos before lf_set_frame");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  // flare_origins.clear(); flare_radiance.clear() (raytraced_renderer.cpp:306-307); axis_ray and
  // angle_to_sun keep their previous values, exactly like the reference's members.
  // The lights travel as kernel arguments (no allocation, no copy, no synchronisation per frame);
  // only a scene with more than kMaxSunLightArgs directional lights takes the staged path.
  if (n_lights <= kMaxSunLightArgs) {
    LfSunLightArgs args;
    std::memset(&args, 0, sizeof(args));
    if (n_lights > 0) std::memcpy(args.v, lights, sizeof(double) * 6 * (size_t)n_lights);
    lf_status st = lfk_frame_setup(ctx, &args, nullptr, n_lights, true);
    if (st != LF_OK) return st;
  } else {
    if ((size_t)n_lights > ctx->sun_lights_cap) {
      LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->sun_lights_dev) (void)hipFree(ctx->sun_lights_dev);
      ctx->sun_lights_dev = nullptr; ctx->sun_lights_cap = 0;
      LF_HIP(ctx, hipMalloc((void**)&ctx->sun_lights_dev, sizeof(double) * 6 * (size_t)n_lights));
      ctx->sun_lights_cap = (size_t)n_lights;
    }
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the previous frame may still read the buffer
    LF_HIP(ctx, hipMemcpy(ctx->sun_lights_dev, lights, sizeof(double) * 6 * (size_t)n_lights,
                          hipMemcpyHostToDevice));
    lf_status st = lfk_frame_setup(ctx, nullptr, ctx->sun_lights_dev, n_lights, true);
    if (st != LF_OK) return st;
  }
  ctx->flares_valid = true;
  ctx->ghost_valid = false;
  return LF_OK;
}

lf_status lf_set_flares(lf_ctx* ctx, int n, const double* origins, const double* radiance,
                        const double axis_ray[2], float angle_to_sun) {
  if (!ctx || n < 0 || n > LF_MAX_FLARES || (n > 0 && (!origins || !radiance)) || !axis_ray)
    return LF_ERR_INVALID;
  LfFlares f;
  std::memset(&f, 0, sizeof(f));
  f.n_flares = n;
  f.angle_to_sun = angle_to_sun;
  for (int k = 0; k < n; k++) {
    f.origin[k][0] = origins[2 * k]; f.origin[k][1] = origins[2 * k + 1];
    for (int c = 0; c < 3; c++) f.radiance[k][c] = radiance[3 * k + c];
  }
  f.axis_ray[0] = axis_ray[0]; f.axis_ray[1] = axis_ray[1];
  LF_HIP(ctx, hipMemcpyAsync(ctx->flares, &f, sizeof(f), hipMemcpyHostToDevice, ctx->stream));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->flares_valid = true;
  ctx->ghost_valid = false;
  return LF_OK;
}

lf_status lf_get_flares(lf_ctx* ctx, int* n, double* origins, double* radiance, double axis_ray[2],
                        float* angle_to_sun) {
  if (!ctx || !n) return LF_ERR_INVALID;
  LfFlares f;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(&f, ctx->flares, sizeof(f), hipMemcpyDeviceToHost));
  *n = f.n_flares;
  for (int k = 0; k < f.n_flares; k++) {
    if (origins) { origins[2 * k] = f.origin[k][0]; origins[2 * k + 1] = f.origin[k][1]; }
    if (radiance) for (int c = 0; c < 3; c++) radiance[3 * k + c] = f.radiance[k][c];
  }
  if (axis_ray) { axis_ray[0] = f.axis_ray[0]; axis_ray[1] = f.axis_ray[1]; }
  if (angle_to_sun) *angle_to_sun = f.angle_to_sun;
  return LF_OK;
}

lf_status lf_set_jitter_mt19937(lf_ctx* ctx, uint32_t seed, const uint32_t* order, size_t n_order) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_jitter_mt19937 before lf_set_frame");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->W * ctx->H;
  std::vector<uint32_t> own;
  if (!order) {  // the reference's tile queue: 32x32 tiles, y-major; pixels y-major inside a tile
    own.reserve(n);
    const int T = 32;
    for (int ty = 0; ty < ctx->H; ty += T)
      for (int tx = 0; tx < ctx->W; tx += T)
        for (int y = ty; y < std::min(ty + T, ctx->H); y++)
          for (int x = tx; x < std::min(tx + T, ctx->W); x++)
            own.push_back((uint32_t)(x + y * ctx->W));
    order = own.data();
    n_order = own.size();
  }
  // Each visited pixel consumes 2*ns_aa draws (pixel jitter, pathtracer.cpp:844) and then the 32
  // draws of calculate_irradiance_falloff (:1050).  Only the latter reach the device, stored
  // pixel-major; a pixel visited twice keeps its last visit, like the reference's buffer.
  std::vector<uint32_t> table(n * 32, 0u);
  const size_t naa = 2 * (size_t)ctx->ns_aa;
  std::vector<uint32_t> aa(n * naa, 0u);
  Mt19937 mt(seed);
  for (size_t v = 0; v < n_order; v++) {
    if (order[v] >= n) return lf_fail(ctx, LF_ERR_INVALID, "visit order: pixel index out of range");
    uint32_t* da = aa.data() + (size_t)order[v] * naa;
    for (size_t k = 0; k < naa; k++) da[k] = mt.next();   // consumed by the scene-term kernel
    uint32_t* dst = table.data() + (size_t)order[v] * 32;
    for (int k = 0; k < 32; k++) dst[k] = mt.next();
  }
  lf_status st;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a running flare layer may read the old tables
  if ((st = dev_alloc(ctx, &ctx->jitter_raw, n * 32)) != LF_OK) return st;
  LF_HIP(ctx, hipMemcpy(ctx->jitter_raw, table.data(), table.size() * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
  if ((st = dev_alloc(ctx, &ctx->jitter_aa_raw, n * naa)) != LF_OK) return st;
  if (naa)
    LF_HIP(ctx, hipMemcpy(ctx->jitter_aa_raw, aa.data(), aa.size() * sizeof(uint32_t),
                          hipMemcpyHostToDevice));
  ctx->jitter_aa_ns = ctx->ns_aa;
  ctx->jitter_mode = 0;
  ctx->jitter_table_valid = true;
  return LF_OK;
}

lf_status lf_set_jitter_counter(lf_ctx* ctx, uint64_t key) {
  if (!ctx) return LF_ERR_INVALID;
  ctx->jitter_mode = 1;
  ctx->jitter_key = key;
  return LF_OK;
}

lf_status lf_set_scene_term(lf_ctx* ctx, const double* rgb) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_scene_term before lf_set_frame");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  size_t n = (size_t)ctx->W * ctx->H * 3;
  if (!rgb) {
    if (ctx->scene) { LF_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->scene); }
    ctx->scene = nullptr;
    return LF_OK;
  }
  lf_status st;
  if (!ctx->scene && (st = dev_alloc(ctx, &ctx->scene, n)) != LF_OK) return st;
  // the context's stream is non-blocking: a null-stream copy is NOT ordered behind a flare layer
  // that still reads the previous scene term
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(ctx->scene, rgb, n * sizeof(double), hipMemcpyHostToDevice));
  return LF_OK;
}

lf_status lf_generate_ghost_buffer(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_generate_ghost_buffer before lf_set_frame");
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "no flare state: call lf_find_sun_pos or lf_set_flares");
  if (!ctx->ap[LF_APERTURE_GHOST].valid) return lf_fail(ctx, LF_ERR_STATE, "ghost aperture not set");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  lf_status st;
  if ((st = lfk_frame_setup(ctx, nullptr, nullptr, 0, false)) != LF_OK) return st;
  if ((st = lfk_ghost_raster(ctx)) != LF_OK) return st;
  ctx->ghost_valid = true;
  return LF_OK;
}

lf_status lf_render_flare_layer(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_render_flare_layer before lf_set_frame");
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "no flare state: call lf_find_sun_pos or lf_set_flares");
  if (!ctx->ap[LF_APERTURE_STARBURST].valid || !ctx->spectrum_valid)
    return lf_fail(ctx, LF_ERR_STATE, "starburst aperture not set");
  if (ctx->jitter_mode == 0 && !ctx->jitter_table_valid)
    return lf_fail(ctx, LF_ERR_STATE, "MT19937 jitter selected but no table (frame was resized?)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  lf_status st;
  if ((st = lfk_flare_layer(ctx)) != LF_OK) return st;
  ctx->sample_valid = true;
  ctx->rgba_y0 = ctx->rgba_y1 = 0;  // the tonemapped copy is stale
  return LF_OK;
}

lf_status lf_read_tile(lf_ctx* ctx, int which, int x0, int y0, int x1, int y1, double* dst,
                       size_t pixel_stride) {
  if (!ctx || !dst || which < 0 || which > 3 || pixel_stride < 3) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_read_tile before lf_set_frame");
  if (which == 3 && !ctx->scene)
    return lf_fail(ctx, LF_ERR_STATE, "lf_read_tile(LF_SCENE_BUFFER) before lf_render_scene_term / lf_set_scene_term");
  if (x0 < 0 || y0 < 0 || x1 > ctx->W || y1 > ctx->H || x0 > x1 || y0 > y1)
    return lf_fail(ctx, LF_ERR_INVALID, "tile out of range");
  if (x0 == x1 || y0 == y1) return LF_OK;
  const double* src = which == 0 ? ctx->sample : which == 1 ? ctx->ghost : which == 2 ? ctx->star : ctx->scene;
  { const lf_status js = lf_comm_join(ctx); if (js != LF_OK) return js; }
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t tw = (size_t)(x1 - x0);
  if (pixel_stride == 3) {
    LF_HIP(ctx, hipMemcpy2D(dst, tw * 3 * sizeof(double), src + 3 * ((size_t)y0 * ctx->W + x0),
                            (size_t)ctx->W * 3 * sizeof(double), tw * 3 * sizeof(double),
                            (size_t)(y1 - y0), hipMemcpyDeviceToHost));
  } else {
    std::vector<double> tmp(tw * 3 * (size_t)(y1 - y0));
    LF_HIP(ctx, hipMemcpy2D(tmp.data(), tw * 3 * sizeof(double),
                            src + 3 * ((size_t)y0 * ctx->W + x0), (size_t)ctx->W * 3 * sizeof(double),
                            tw * 3 * sizeof(double), (size_t)(y1 - y0), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tw * (size_t)(y1 - y0); i++)
      for (int c = 0; c < 3; c++) dst[i * pixel_stride + c] = tmp[3 * i + c];
  }
  return LF_OK;
}

lf_status lf_read_pixel(lf_ctx* ctx, int which, int x, int y, double rgb[3]) {
  return lf_read_tile(ctx, which, x, y, x + 1, y + 1, rgb, 3);
}

lf_status lf_write_to_framebuffer(lf_ctx* ctx, int x0, int y0, int x1, int y1, uint32_t* dst,
                                  size_t row_stride) {
  if (!ctx || !dst) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_write_to_framebuffer before lf_set_frame");
  if (x0 < 0 || y0 < 0 || x1 > ctx->W || y1 > ctx->H || x0 > x1 || y0 > y1)
    return lf_fail(ctx, LF_ERR_INVALID, "tile out of range");
  if (row_stride < (size_t)(x1 - x0)) return lf_fail(ctx, LF_ERR_INVALID, "row_stride < tile width");
  if (x0 == x1 || y0 == y1) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  { const lf_status js = lf_comm_join(ctx); if (js != LF_OK) return js; }
  if (y0 < ctx->rgba_y0 || y1 > ctx->rgba_y1) {
    // tonemap the current band and whatever else the tile needs (the reference tonemaps exactly
    // the tile, image.h:208-223; whole rows keep the launch simple), remember the rows done
    const int ya = std::min(y0, ctx->y0 < ctx->y1 ? ctx->y0 : y0);
    const int yb = std::max(y1, ctx->y0 < ctx->y1 ? ctx->y1 : y1);
    lf_status st = lfk_tonemap(ctx, ya, yb);
    if (st != LF_OK) return st;
    ctx->rgba_y0 = ya; ctx->rgba_y1 = yb;
  }
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy2D(dst, row_stride * sizeof(uint32_t), ctx->rgba + (size_t)y0 * ctx->W + x0,
                          (size_t)ctx->W * sizeof(uint32_t), (size_t)(x1 - x0) * sizeof(uint32_t),
                          (size_t)(y1 - y0), hipMemcpyDeviceToHost));
  return LF_OK;
}

lf_status lf_save_image_rgba(lf_ctx* ctx, uint32_t* dst) {
  if (!ctx || !dst) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_save_image_rgba before lf_set_frame");
  if (!ctx->sample_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_save_image_rgba before lf_render_flare_layer");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  { const lf_status js = lf_comm_join(ctx); if (js != LF_OK) return js; }
  lf_status st = lfk_tonemap(ctx, 0, ctx->H);   // the saved image is always the whole frame
  if (st != LF_OK) return st;
  ctx->rgba_y0 = 0; ctx->rgba_y1 = ctx->H;
  const size_t n = (size_t)ctx->W * ctx->H;
  if (!ctx->rgba_flip) LF_HIP(ctx, hipMalloc((void**)&ctx->rgba_flip, n * sizeof(uint32_t)));
  st = lfk_flip_rows(ctx, ctx->rgba_flip);
  if (st != LF_OK) return st;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(dst, ctx->rgba_flip, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return LF_OK;
}

lf_status lf_device_buffer(lf_ctx* ctx, int which, void** dptr, size_t* bytes) {
  if (!ctx || !dptr || which < 0 || which > 2) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_device_buffer before lf_set_frame");
  *dptr = which == 0 ? ctx->sample : which == 1 ? ctx->ghost : ctx->star;
  if (bytes) *bytes = (size_t)ctx->W * ctx->H_alloc * 3 * sizeof(double);
  return LF_OK;
}

// ---------------------------------------------------------------- geometric lens -------------
lf_status lf_set_lens(lf_ctx* ctx, int n_surfaces, int stop_index, int n_lambda,
                      const float* radius, const float* thickness, const float* ior,
                      const float* semi_aperture, float sensor_width_mm) {
  if (!ctx || !radius || !thickness || !ior || !semi_aperture) return LF_ERR_INVALID;
  if (n_surfaces < 1 || n_surfaces > LF_MAX_SURFACES || n_lambda < 1 || n_lambda > LF_MAX_LAMBDA ||
      stop_index < -1 || stop_index >= n_surfaces || !(sensor_width_mm > 0))
    return lf_fail(ctx, LF_ERR_INVALID, "lens: bad sizes");
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_lens before lf_set_frame");
  for (int k = 0; k < n_surfaces; k++) {
    if (!(semi_aperture[k] > 0) || !std::isfinite(semi_aperture[k]))
      return lf_fail(ctx, LF_ERR_INVALID, "lens: semi-aperture <= 0 or not finite");
    if (!std::isfinite(radius[k]) || !std::isfinite(thickness[k]))
      return lf_fail(ctx, LF_ERR_INVALID, "lens: radius / thickness not finite (a flat surface is radius 0)");
    if (k == stop_index && radius[k] != 0) return lf_fail(ctx, LF_ERR_INVALID, "lens: the stop must be flat");
    for (int l = 0; l < n_lambda; l++)
      if ((!(ior[l * n_surfaces + k] >= 1.0f) || !std::isfinite(ior[l * n_surfaces + k])) && k != stop_index)
        return lf_fail(ctx, LF_ERR_INVALID, "lens: index of refraction < 1 or not finite");
  }
  // a pupil target belongs to the prescription it was computed for (lf_aim_at_exit_pupil): a new lens
  // starts from the default disc, the rear element's clear aperture
  ctx->pupil_target_h = 0.0f; ctx->pupil_target_z = 0.0f;
  lf_derive_lens(ctx, n_surfaces, stop_index, n_lambda, radius, thickness, ior, semi_aperture,
                 sensor_width_mm);
  ctx->raw_n = n_surfaces; ctx->raw_stop = stop_index;
  std::memcpy(ctx->raw_radius, radius, sizeof(float) * (size_t)n_surfaces);
  std::memcpy(ctx->raw_thickness, thickness, sizeof(float) * (size_t)n_surfaces);
  std::memcpy(ctx->raw_ior, ior, sizeof(float) * (size_t)n_surfaces * (size_t)n_lambda);
  ctx->lens_valid = true;
  ctx->events_dirty = true;
  // default pair set: every pair of glass surfaces + the primary path
  return lf_set_ghost_pairs(ctx, nullptr, 0, 1);
}

lf_status lf_set_lambda_rgb(lf_ctx* ctx, const float* weights) {
  if (!ctx || !weights) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_set_lambda_rgb before lf_set_lens");
  // the march's sums are UNSIGNED fixed point: a weight must be finite and >= 0 (a host whose colour matrix has
  // negative lobes renders the positive and the negative part as two frames and subtracts)
  for (int k = 0; k < 3 * ctx->lens.n_lambda; k++)
    if (!std::isfinite(weights[k]) || weights[k] < 0.0f)
      return lf_fail(ctx, LF_ERR_INVALID, "lf_set_lambda_rgb: weights must be finite and >= 0");
  for (int l = 0; l < ctx->lens.n_lambda; l++)
    for (int c = 0; c < 3; c++) ctx->lens.lambda_rgb[l][c] = weights[3 * l + c];
  return LF_OK;
}

lf_status lf_set_sun(lf_ctx* ctx, const float dir[3], const float radiance[3],
                     float angular_radius) {
  if (!ctx || !dir || !radiance) return LF_ERR_INVALID;
  double n = std::sqrt((double)dir[0] * dir[0] + (double)dir[1] * dir[1] + (double)dir[2] * dir[2]);
  if (!(n > 0) || !(dir[2] < 0) || !(angular_radius > 0) || angular_radius > 1.5f)
    return lf_fail(ctx, LF_ERR_INVALID, "sun: direction must have z < 0, 0 < angular radius <= 1.5");
  for (int c = 0; c < 3; c++)   // unsigned fixed-point sums: see lf_set_lambda_rgb; any finite magnitude is fine (lf_march_fix_bits)
    if (!std::isfinite(radiance[c]) || radiance[c] < 0.0f || !std::isfinite(dir[c]))
      return lf_fail(ctx, LF_ERR_INVALID, "sun: radiance must be finite and >= 0, the direction finite");
  for (int c = 0; c < 3; c++) {
    ctx->lens.sun_dir[c] = (float)(dir[c] / n);
    ctx->lens.sun_radiance[c] = radiance[c];
  }
  ctx->lens.sun_inv_one_minus_cos = (float)(1.0 / (1.0 - std::cos((double)angular_radius)));
  ctx->lens.sun_ss = (float)((double)ctx->lens.sun_dir[0] * ctx->lens.sun_dir[0] +
                             (double)ctx->lens.sun_dir[1] * ctx->lens.sun_dir[1] +
                             (double)ctx->lens.sun_dir[2] * ctx->lens.sun_dir[2]);
  ctx->sun_valid = true;
  return LF_OK;
}

lf_status lf_paraxial_efl(int n, int stop, const float* radius, const float* thickness,
                          const float* ior_row, double* efl_mm) {
  if (n < 1 || n > LF_MAX_SURFACES || !radius || !thickness || !ior_row || !efl_mm) return LF_ERR_INVALID;
  // ray (height, angle); T(d) = [[1, d], [0, 1]], R(c, n1, n2) = [[1, 0], [c (n1 - n2) / n2, n1 / n2]]
  double m00 = 1, m01 = 0, m10 = 0, m11 = 1, n1 = 1.0;
  for (int k = 0; k < n; k++) {
    if (k != stop) {
      const double c = radius[k] == 0.0f ? 0.0 : 1.0 / (double)radius[k], n2 = ior_row[k];
      if (!(n2 >= 1.0)) return LF_ERR_INVALID;
      const double r10 = c * (n1 - n2) / n2, r11 = n1 / n2;
      const double a10 = r10 * m00 + r11 * m10, a11 = r10 * m01 + r11 * m11;
      m10 = a10; m11 = a11;
      n1 = n2;
    }
    if (k + 1 < n) {  // the last thickness (to the sensor) does not change the power
      const double d = thickness[k];
      m00 += d * m10; m01 += d * m11;
    }
  }
  if (m10 == 0.0) return LF_ERR_INVALID;  // afocal
  *efl_mm = -1.0 / (m10 * n1);            // image-space medium n1 (air for a camera lens)
  return LF_OK;
}

// Where a collimated beam from field angle theta (small) lands on the sensor plane, per unit of theta: the
// height of its CHIEF ray (the one through the stop's centre) -- the centre of the beam's image whether the
// sensor sits in the focal plane (then this is the focal length) or not (a lens refocused by lf_focus_lens:
// focal length + sensor shift x the chief ray's exit slope).  Paraxial, the reference's T / R operators.
static bool paraxial_image_scale(int n, int stop, const float* radius, const float* thickness, const float* ior_row,
                                 double* scale) {
  // two rays from the first vertex, (height 1, slope 0) and (height 0, slope 1): everything is linear in them
  double y[2] = {1.0, 0.0}, u[2] = {0.0, 1.0}, ys[2] = {1.0, 0.0};
  double n1 = 1.0;
  for (int k = 0; k < n; k++) {
    if (k == stop) { ys[0] = y[0]; ys[1] = y[1]; }
    else {
      const double c = radius[k] == 0.0f ? 0.0 : 1.0 / (double)radius[k], n2 = ior_row[k];
      if (!(n2 >= 1.0)) return false;
      for (int r = 0; r < 2; r++) u[r] = c * (n1 - n2) / n2 * y[r] + n1 / n2 * u[r];
      n1 = n2;
    }
    for (int r = 0; r < 2; r++) y[r] += (double)thickness[k] * u[r];   // (the last thickness: to the sensor)
  }
  // chief ray of slope 1: height y0 at the first vertex such that it crosses the stop's centre
  const double y0 = (stop >= 0 && ys[0] != 0.0) ? -ys[1] / ys[0] : 0.0;
  *scale = y0 * y[0] + y[1];
  return std::isfinite(*scale) && *scale != 0.0;
}

lf_status lf_paraxial_image_scale(int n, int stop, const float* radius, const float* thickness, const float* ior_row,
                                  double* scale_mm) {
  if (n <= 0 || !radius || !thickness || !ior_row || !scale_mm) return LF_ERR_INVALID;
  return paraxial_image_scale(n, stop, radius, thickness, ior_row, scale_mm) ? LF_OK : LF_ERR_INVALID;
}

lf_status lf_set_sun_from_flares(lf_ctx* ctx, int flare, double efl_mm, float angular_radius) {
  if (!ctx || flare < 0 || flare >= LF_MAX_FLARES) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_set_sun_from_flares before lf_set_frame");
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_set_sun_from_flares before lf_set_lens");
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "no flare state: call lf_find_sun_pos or lf_set_flares");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LfFlares f;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(&f, ctx->flares, sizeof(f), hipMemcpyDeviceToHost));
  if (flare >= f.n_flares) return lf_fail(ctx, LF_ERR_INVALID, "lf_set_sun_from_flares: no such flare in the frame");
  if (!(efl_mm > 0.0)) {
    // the image scale of THIS sensor position: the focal length when the sensor sits in the focal plane (the
    // shipped prescriptions), the chief ray's landing height per unit field angle when lf_focus_lens has moved it
    if (!paraxial_image_scale(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                              ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, &efl_mm) || !(efl_mm > 0.0))
      return lf_fail(ctx, LF_ERR_INVALID, "lf_set_sun_from_flares: the prescription images no distant point on its sensor");
  }
  const double sw = ctx->sensor_w_mm, sh = sw * (double)ctx->H / (double)ctx->W;
  const float dir[3] = {(float)((f.origin[flare][0] - 0.5) * sw / efl_mm),
                        (float)((f.origin[flare][1] - 0.5) * sh / efl_mm), -1.0f};
  const float rad[3] = {(float)f.radiance[flare][0], (float)f.radiance[flare][1], (float)f.radiance[flare][2]};
  return lf_set_sun(ctx, dir, rad, angular_radius);
}

lf_status lf_set_ghost_pairs(lf_ctx* ctx, const int* pairs, int n_pairs, int include_primary) {
  if (!ctx || n_pairs < 0) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_set_ghost_pairs before lf_set_lens");
  LfPairsDev P;
  std::memset(&P, 0, sizeof(P));
  const int ns = ctx->lens.n_surf, stop = ctx->lens.stop;
  if (include_primary) { P.ij[P.n][0] = -1; P.ij[P.n][1] = -1; P.n++; }
  if (!pairs || n_pairs == 0) {
    for (int i = 0; i < ns; i++)
      for (int j = i + 1; j < ns; j++) {
        if (i == stop || j == stop) continue;
        if (P.n >= LF_MAX_PAIRS + 1) return lf_fail(ctx, LF_ERR_INVALID, "too many ghost pairs");
        P.ij[P.n][0] = i; P.ij[P.n][1] = j; P.n++;
      }
  } else {
    if (n_pairs > LF_MAX_PAIRS) return lf_fail(ctx, LF_ERR_INVALID, "too many ghost pairs");
    for (int q = 0; q < n_pairs; q++) {
      int i = pairs[2 * q], j = pairs[2 * q + 1];
      if (i == -1 && j == -1) {  // the primary path, listed explicitly
        if (P.n > 0 && P.ij[0][0] < 0) return lf_fail(ctx, LF_ERR_INVALID, "primary path listed twice");
        if (P.n > 0) {  // keep it first, like include_primary does
          for (int k = P.n; k > 0; k--) { P.ij[k][0] = P.ij[k - 1][0]; P.ij[k][1] = P.ij[k - 1][1]; }
        }
        P.ij[0][0] = -1; P.ij[0][1] = -1; P.n++;
        continue;
      }
      if (i < 0 || j <= i || j >= ns || i == stop || j == stop)
        return lf_fail(ctx, LF_ERR_INVALID, "ghost pair must satisfy 0 <= i < j < n, neither the stop");
      P.ij[P.n][0] = i; P.ij[P.n][1] = j; P.n++;
    }
  }
  if (P.n == 0) return lf_fail(ctx, LF_ERR_INVALID, "empty pair set");
  ctx->pairs = P;
  ctx->events_dirty = true;
  return LF_OK;
}

lf_status lf_set_pupil_subcells(lf_ctx* ctx, int bits) {
  if (!ctx) return LF_ERR_INVALID;
  if (bits < 0 || bits > 8) return lf_fail(ctx, LF_ERR_INVALID, "pupil sub-cell bits must be 0..8");
  ctx->march_sub_bits = bits;
  return LF_OK;
}

lf_status lf_set_tile_stride(lf_ctx* ctx, int stride) {
  if (!ctx) return LF_ERR_INVALID;
  if (stride != 1 && stride != 2 && stride != 4 && stride != 8)
    return lf_fail(ctx, LF_ERR_INVALID, "tile stride must be 1, 2, 4 or 8");
  ctx->march_xstride_log2 = stride == 1 ? 0 : stride == 2 ? 1 : stride == 4 ? 2 : 3;
  return LF_OK;
}

lf_status lf_trace_ghosts(lf_ctx* ctx, int spp, uint64_t key) {
  if (!ctx || spp <= 0) return LF_ERR_INVALID;
  if (!ctx->lens_valid || !ctx->sun_valid)
    return lf_fail(ctx, LF_ERR_STATE, "lf_trace_ghosts needs lf_set_lens and lf_set_sun");
  if (!ctx->ap[LF_APERTURE_STARBURST].valid)
    return lf_fail(ctx, LF_ERR_STATE, "aperture mask (LF_APERTURE_STARBURST slot) not set");
  if (ctx->pupil_target_h > 0.0f && ctx->lens.stop >= 0) {
    // a reduced disc (the stop's image) is an unbiased estimator only for paths that cross the stop before
    // their first reflection; a pair with both mirrors behind the stop reaches the sensor through parts of
    // the rear element the disc does not cover
    for (int q = 0; q < ctx->pairs.n; q++)
      if (ctx->pairs.ij[q][0] > ctx->lens.stop)
        return lf_fail(ctx, LF_ERR_STATE,
                       "a pupil target is set and the pair selection holds a pair with both mirrors behind the stop: "
                       "such pairs need the default disc (march them in a second launch, lf_set_ghost_accumulate)");
  }
  LF_HIP(ctx, hipSetDevice(ctx->device));
  lf_status st = lfk_march(ctx, spp, key);
  if (st != LF_OK) return st;
  ctx->ghost_valid = true;
  return LF_OK;
}

lf_status lf_set_march_culling(lf_ctx* ctx, int mode) {
  if (!ctx) return LF_ERR_INVALID;
  if (mode < 0 || mode > 2) return lf_fail(ctx, LF_ERR_INVALID, "march culling: 0 off, 1 on, 2 on and rebuilt at every launch");
  ctx->march_cull = mode;
  return LF_OK;
}

// ---- the pre-pass of a multi-GPU frame, shared ----------------------------------------------
lf_status lf_set_cull_share(lf_ctx* ctx, int rank, int nranks) {
  if (!ctx) return LF_ERR_INVALID;
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return lf_fail(ctx, LF_ERR_INVALID, "lf_set_cull_share: 0 <= rank < nranks <= 64");
  if (ctx->cull_share_how == 1) return lf_fail(ctx, LF_ERR_STATE, "lf_set_cull_share: the table is shared through the communicator (lf_comm_share_cull)");
  ctx->cull_share_how = nranks > 1 ? 2 : 0;
  ctx->cull_share_rank = nranks > 1 ? rank : 0;
  ctx->cull_share_n = nranks;
  ctx->cull_hash = 0; ctx->cull_hash_pending = 0; ctx->cull_fresh = false;
  return LF_OK;
}

lf_status lf_cull_prepare(lf_ctx* ctx, int spp) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->cull_share_how != 2) return lf_fail(ctx, LF_ERR_STATE, "lf_cull_prepare without lf_set_cull_share(rank, nranks > 1)");
  if (!ctx->lens_valid || !ctx->ap[LF_APERTURE_STARBURST].valid || ctx->W == 0)
    return lf_fail(ctx, LF_ERR_STATE, "lf_cull_prepare: frame, prescription and aperture mask come first");
  if (spp < 1) return LF_ERR_INVALID;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  return lfk_cull_prepare(ctx, spp);
}

lf_status lf_cull_table_view(lf_ctx* ctx, void** device_ptr, uint64_t* entries, uint64_t* entries_per_rank) {
  if (!ctx || !device_ptr || !entries || !entries_per_rank) return LF_ERR_INVALID;
  *device_ptr = nullptr; *entries = 0; *entries_per_rank = 0;
  if (ctx->cull_hash_pending == 0) return LF_OK;           // (this launch does not cull: nothing to exchange)
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the slab is written when the host's exchange reads it
  *entries_per_rank = (uint64_t)ctx->cull_share_nb * (uint64_t)(ctx->cull_cells + 1);
  *entries = *entries_per_rank * (uint64_t)ctx->cull_share_n_resident;
  *device_ptr = ctx->cull_dev;
  return LF_OK;
}

lf_status lf_cull_commit(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->cull_share_how != 2) return lf_fail(ctx, LF_ERR_STATE, "lf_cull_commit without lf_set_cull_share(rank, nranks > 1)");
  if (ctx->cull_hash_pending == 0) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t h = ctx->cull_hash_pending;
  ctx->cull_hash_pending = 0;
  const lf_status st = lfk_cull_finish(ctx, h);
  if (st != LF_OK) return st;
  ctx->cull_fresh = true;
  return LF_OK;
}

lf_status lf_get_cull_info(lf_ctx* ctx, int info[8]) {
  if (!ctx || !info) return LF_ERR_INVALID;
  info[0] = ctx->march_cull;
  info[1] = ctx->last_march_culled ? 1 : 0;
  info[2] = ctx->cull_bx; info[3] = ctx->cull_by; info[4] = ctx->cull_cells; info[5] = ctx->cull_G; info[6] = ctx->cull_P;
  info[7] = 1 << ctx->cull_blk_log2;
  return LF_OK;
}

lf_status lf_get_cull_reason(lf_ctx* ctx, int* reason) {
  if (!ctx || !reason) return LF_ERR_INVALID;
  *reason = ctx->cull_reason;
  return LF_OK;
}

lf_status lf_set_cull_audit(lf_ctx* ctx, int rays_per_dropped_box) {
  if (!ctx) return LF_ERR_INVALID;
  if (rays_per_dropped_box < 0 || rays_per_dropped_box > 255) return lf_fail(ctx, LF_ERR_INVALID, "cull audit: 0 (off) .. 255 rays per dropped box");
  if (rays_per_dropped_box != ctx->cull_audit_density) ctx->cull_hash = 0;     // (a resident table was audited at the other density)
  ctx->cull_audit_density = rays_per_dropped_box;
  return LF_OK;
}

lf_status lf_get_cull_audit(lf_ctx* ctx, uint64_t* rays, uint64_t* lit, int* launches_refuted) {
  if (!ctx) return LF_ERR_INVALID;
  if (rays) *rays = ctx->cull_audit_rays;
  if (lit) *lit = ctx->cull_audit_lit;
  if (launches_refuted) *launches_refuted = ctx->cull_audit_tripped;
  return LF_OK;
}

// TEST HOOK (lensflare.h): the switches of the test suite and of bench.py's A/B legs, by name
// (ctx == null: the value becomes the default of every context created afterwards -- a test that cannot reach the
// contexts a helper creates; value 0 of a 0/1 knob takes the default away again)
lf_status lf_test_knob(lf_ctx* ctx, const char* name, double value) {
  if (!name) return LF_ERR_INVALID;
  if (!ctx) {
    std::lock_guard<std::mutex> lock(g_knob_mu);
    for (size_t i = 0; i < g_knob_defaults.size(); i++)
      if (g_knob_defaults[i].first == name) { g_knob_defaults.erase(g_knob_defaults.begin() + (long)i); break; }
    if (value != 0.0) g_knob_defaults.emplace_back(name, value);
    return LF_OK;
  }
  const std::string n(name);
  const int iv = (int)value;
  lf_ctx::CullRules& R = ctx->cull_rules;
  bool rules = true;
  if (n == "cull_force") { ctx->cull_force = iv != 0; return LF_OK; }
  if (n == "cull_weights_first") { ctx->cull_weights_first = iv != 0; return LF_OK; }
  if (n == "cull_no_prefix") { ctx->cull_no_prefix = iv != 0; return LF_OK; }
  if (n == "scene_compact") { ctx->scene_compact = iv < 0 ? -1 : iv != 0; return LF_OK; }
  if (n == "comm_force_exchange") { ctx->comm_force_exchange = iv != 0; return LF_OK; }
  if (n == "scene_lens_strided") { ctx->scene_lens_strided = iv < 0 ? -1 : iv != 0; return LF_OK; }
  if (n == "bvh_median") { ctx->bvh_median = iv != 0; return LF_OK; }
  if (n == "bvh_leaf") { ctx->bvh_leaf_max = iv; return LF_OK; }
  if (n == "march_tail_tiles") { ctx->march_tail_tiles = iv; return LF_OK; }       // (-1: the default, one round of resident workgroups)
  if (n == "march_tail_groups") { ctx->march_tail_groups = iv; return LF_OK; }     // (1: no split tail)
  if (n == "cull_general_kernel") { if (!iv) R = lf_ctx::CullRules(); ctx->cull_rules_custom = iv != 0; }
  else if (n == "cull_strict") R.strict = iv;
  else if (n == "cull_strict_lost") R.strict_lost = iv;
  else if (n == "cull_slack") R.slack_mode = iv;
  else if (n == "cull_keep_partial") R.keep_partial = iv;
  else if (n == "cull_disable") R.disable = iv;
  else if (n == "cull_margin") R.margin = (float)value;
  else if (n == "cull_lobe_k") R.lobe_k = (float)value;
  else if (n == "cull_lost_rel") R.lost_rel = (float)value;
  else if (n == "cull_lost_abs") R.lost_abs = (float)value;
  else rules = false;
  if (!rules) return lf_fail(ctx, LF_ERR_INVALID, "lf_test_knob: unknown knob '" + n + "'");
  if (n != "cull_general_kernel") ctx->cull_rules_custom = true;
  ctx->cull_hash = 0;
  return LF_OK;
}

lf_status lf_get_cull_started_fraction(lf_ctx* ctx, double* fraction) {
  if (!ctx || !fraction) return LF_ERR_INVALID;
  *fraction = ctx->cull_started_fraction;
  return LF_OK;
}

lf_status lf_get_cull_table(lf_ctx* ctx, uint64_t* out, size_t n_entries) {
  if (!ctx || !out) return LF_ERR_INVALID;
  if (!ctx->last_march_culled || !ctx->cull_dev || ctx->cull_hash == 0)
    return lf_fail(ctx, LF_ERR_STATE, "lf_get_cull_table: the last lf_trace_ghosts did not cull (or none has run)");
  if (ctx->cull_chunks > 1)
    return lf_fail(ctx, LF_ERR_STATE, "lf_get_cull_table: the last lf_trace_ghosts marched more than 64 paths in two launches, each with its own table");
  const size_t n = (size_t)ctx->cull_bx * ctx->cull_by * (size_t)(ctx->cull_cells + 1);
  if (n_entries != n) return lf_fail(ctx, LF_ERR_INVALID, "lf_get_cull_table: size must be blocks_x * blocks_y * (cells + 1)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->cull_share_nb > 0 && ctx->cull_share_n_resident > 1) {
    // a shared table lies slab by slab (lf_cull_row_of_block): handed out in block order all the same
    const size_t re = (size_t)(ctx->cull_cells + 1), rows = (size_t)ctx->cull_share_nb * ctx->cull_share_n_resident;
    std::vector<uint64_t> raw(rows * re);
    LF_HIP(ctx, hipMemcpy(raw.data(), ctx->cull_dev, raw.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < ctx->cull_bx * ctx->cull_by; b++)
      std::memcpy(out + (size_t)b * re, raw.data() + lf_cull_row_of_block(b, ctx->cull_share_n_resident, ctx->cull_share_nb) * re, re * sizeof(uint64_t));
    return LF_OK;
  }
  LF_HIP(ctx, hipMemcpy(out, ctx->cull_dev, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return LF_OK;
}

lf_status lf_get_march_fix_bits(lf_ctx* ctx, int* bits) {
  if (!ctx || !bits) return LF_ERR_INVALID;
  *bits = ctx->march_fix_bits;
  return LF_OK;
}

lf_status lf_generate_lens_rays(lf_ctx* ctx, int lambda, size_t n, const float* sensor_xy_mm,
                                const float* pupil_uv, float* out) {
  if (!ctx || !sensor_xy_mm || !pupil_uv || !out) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_generate_lens_rays before lf_set_lens");
  if (!ctx->ap[LF_APERTURE_STARBURST].valid)
    return lf_fail(ctx, LF_ERR_STATE, "aperture mask (LF_APERTURE_STARBURST slot) not set");
  if (lambda < 0 || lambda >= ctx->lens.n_lambda || n > 0x7fffffffull)
    return lf_fail(ctx, LF_ERR_INVALID, "lf_generate_lens_rays: wavelength index / count out of range");
  if (n == 0) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  float *d_xy = nullptr, *d_uv = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc((void**)&d_xy, n * 2 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d_uv, n * 2 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, n * 8 * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(d_xy, sensor_xy_mm, n * 2 * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_uv, pupil_uv, n * 2 * sizeof(float), hipMemcpyHostToDevice);
  lf_status st = LF_OK;
  if (e == hipSuccess) st = lfk_lens_rays(ctx, lambda, (int)n, d_xy, d_uv, d_out);
  if (e == hipSuccess && st == LF_OK) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && st == LF_OK) e = hipMemcpy(out, d_out, n * 8 * sizeof(float), hipMemcpyDeviceToHost);
  if (d_xy) (void)hipFree(d_xy);
  if (d_uv) (void)hipFree(d_uv);
  if (d_out) (void)hipFree(d_out);
  if (st != LF_OK) return st;
  LF_HIP(ctx, e);
  return LF_OK;
}

lf_status lf_march_tables(int n_surfaces, int stop_index, int n_lambda, const float* radius,
                          const float* thickness, const float* ior, const float* semi_aperture,
                          const int* pairs, int n_pairs, int include_primary, int* info,
                          float* rows, size_t rows_cap, int* skip, size_t skip_cap) {
  if (!info) return LF_ERR_INVALID;
  lf_ctx ctx;          // plain host state: nothing below touches the device
  ctx.W = ctx.H = 64;
  lf_status st = lf_set_lens(&ctx, n_surfaces, stop_index, n_lambda, radius, thickness, ior, semi_aperture, 36.0f);
  if (st == LF_OK && (pairs || !include_primary)) st = lf_set_ghost_pairs(&ctx, pairs, n_pairs, include_primary);
  std::vector<LfEventRow> r;
  std::vector<int> sk;
  if (st == LF_OK) st = lf_build_march_tables(&ctx, r, sk);
  if (st != LF_OK) return st;
  const LfPairsDev& P = ctx.pairs;
  info[0] = P.n; info[1] = P.total_events; info[2] = P.prog_off; info[3] = P.prog_rows;
  info[4] = (int)r.size(); info[5] = (int)sk.size();
  for (int q = 0; q < P.n && q < LF_MAX_PAIRS + 1; q++) {
    info[8 + 4 * q] = P.ij[q][0]; info[8 + 4 * q + 1] = P.ij[q][1];
    info[8 + 4 * q + 2] = P.ev_off[q]; info[8 + 4 * q + 3] = P.ev_cnt[q];
  }
  if (rows) {
    if (rows_cap < r.size() * 8) return LF_ERR_INVALID;
    for (size_t i = 0; i < r.size(); i++) std::memcpy(rows + 8 * i, &r[i], 8 * sizeof(float));   // the documented fields
  }
  if (skip) {
    if (skip_cap < sk.size()) return LF_ERR_INVALID;
    std::memcpy(skip, sk.data(), sk.size() * sizeof(int));
  }
  return LF_OK;
}

namespace {
// what Application::load hands the renderer, flattened for lf_set_scene / lf_set_scene_lights;
// returns an empty string, or why the device scene term cannot render the file
extern "C++" std::string flatten_collada(const lfamd::ColladaScene& sc, std::vector<double>& sph, std::vector<int>& sph_m,
                            std::vector<double>& tp, std::vector<double>& tn, std::vector<int>& tri_m,
                            std::vector<double>& mats, std::vector<double>& lights, std::vector<double>& suns) {
  // DiffuseBSDF and EmissionBSDF as they are.  Mirror / Refraction / Glass / Microfacet are unfilled
  // stubs in the reference: their f() returns 0 and they emit nothing (advanced_bsdf.cpp:17-133,
  // bsdf.h:183-254), and the reference's integrator is zero_bounce + one_bounce only
  // (pathtracer.cpp:282-302) -- such a surface IS a black occluder there, so that is what it is here:
  // diffuse with reflectance 0.
  for (const auto& m : sc.materials) {
    const bool lit = m.kind == lfamd::BSDF_DIFFUSE || m.kind == lfamd::BSDF_EMISSION;
    mats.push_back(m.kind == lfamd::BSDF_EMISSION ? 1.0 : 0.0);
    mats.push_back(lit ? m.rgb.x : 0.0); mats.push_back(lit ? m.rgb.y : 0.0); mats.push_back(lit ? m.rgb.z : 0.0);
  }
  for (const auto& s : sc.spheres) {
    sph.insert(sph.end(), {s.o.x, s.o.y, s.o.z, s.r});
    sph_m.push_back(s.material);
  }
  for (const auto& t : sc.triangles) {
    for (int k = 0; k < 3; k++) { tp.push_back(t.p[k].x); tp.push_back(t.p[k].y); tp.push_back(t.p[k].z); }
    for (int k = 0; k < 3; k++) { tn.push_back(t.n[k].x); tn.push_back(t.n[k].y); tn.push_back(t.n[k].z); }
    tri_m.push_back(t.material);
  }
  for (const auto& l : sc.lights) {
    double row[16] = {0};
    row[1] = l.radiance.x; row[2] = l.radiance.y; row[3] = l.radiance.z;
    auto put = [&](int at, const lfamd::ColladaVec3& v) { row[at] = v.x; row[at + 1] = v.y; row[at + 2] = v.z; };
    if (l.type == lfamd::LIGHT_DIRECTIONAL) {
      row[0] = 0; put(4, l.direction);
      suns.insert(suns.end(), {l.position.x, l.position.y, l.position.z, l.radiance.x, l.radiance.y, l.radiance.z});
    } else if (l.type == lfamd::LIGHT_POINT) {
      row[0] = 1; put(4, l.position);
    } else if (l.type == lfamd::LIGHT_HEMISPHERE) {
      row[0] = 2;
    } else if (l.type == lfamd::LIGHT_AREA) {
      row[0] = 3; put(4, l.position); put(7, l.direction); put(10, l.dim_x); put(13, l.dim_y);
    } else {
      return "spot lights are a stub in the reference (light.cpp:64-72) and are not rendered";
    }
    lights.insert(lights.end(), row, row + 16);
  }
  return "";
}
}  // namespace

lf_status lf_collada_check(const char* path, char* msg, size_t msg_cap) {
  if (!path) return LF_ERR_INVALID;
  lfamd::ColladaScene sc;
  std::string err;
  if (!lfamd::load_collada(path, sc, err)) err = "lf_load_collada: " + err;
  else {
    std::vector<double> a, c, d, e, f, g;
    std::vector<int> b, h;
    err = flatten_collada(sc, a, b, c, d, h, e, f, g);
  }
  if (msg && msg_cap) { std::strncpy(msg, err.c_str(), msg_cap - 1); msg[msg_cap - 1] = 0; }
  return err.empty() ? LF_OK : LF_ERR_INVALID;
}

lf_status lf_load_collada(lf_ctx* ctx, const char* path, lf_collada_camera* camera, double* sun_lights,
                          int max_sun_lights, int* n_sun_lights) {
  if (!ctx || !path) return LF_ERR_INVALID;
  lfamd::ColladaScene sc;
  std::string err;
  if (!lfamd::load_collada(path, sc, err)) return lf_fail(ctx, LF_ERR_INVALID, "lf_load_collada: " + err);
  std::vector<double> sph, tp, tn, mats, lights, suns;
  std::vector<int> sph_m, tri_m;
  err = flatten_collada(sc, sph, sph_m, tp, tn, tri_m, mats, lights, suns);
  if (!err.empty()) return lf_fail(ctx, LF_ERR_INVALID, "lf_load_collada: " + err);
  const int n_sun = (int)(suns.size() / 6);
  for (int k = 0; k < n_sun && sun_lights && k < max_sun_lights; k++)
    std::memcpy(sun_lights + 6 * k, suns.data() + 6 * k, 6 * sizeof(double));
  if (n_sun_lights) *n_sun_lights = n_sun;
  if (camera) {
    std::memset(camera, 0, sizeof(*camera));
    const auto& c = sc.camera;
    camera->present = c.present ? 1 : 0;
    camera->hfov = c.hFov; camera->vfov = c.vFov; camera->nclip = c.nClip; camera->fclip = c.fClip;
    camera->pos[0] = c.pos.x; camera->pos[1] = c.pos.y; camera->pos[2] = c.pos.z;
    camera->dir[0] = c.dir.x; camera->dir[1] = c.dir.y; camera->dir[2] = c.dir.z;
    camera->up[0] = c.up.x; camera->up[1] = c.up.y; camera->up[2] = c.up.z;
  }
  lf_status st = lf_set_scene(ctx, (int)sph_m.size(), sph.data(), sph_m.data(), (int)tri_m.size(), tp.data(),
                              tn.data(), tri_m.data(), (int)sc.materials.size(), mats.data(), 0, nullptr);
  if (st != LF_OK) return st;
  return lf_set_scene_lights(ctx, (int)(lights.size() / 16), lights.data());
}

lf_status lf_set_starburst_spectrum(lf_ctx* ctx, int n, const double* scale, const double* rgb_weights) {
  if (!ctx || n < 0 || n > LF_MAX_LAMBDA || (n > 0 && (!scale || !rgb_weights))) return LF_ERR_INVALID;
  LfStarSpectrum sp{};
  sp.n = n;
  for (int l = 0; l < n; l++) {
    if (!(scale[l] > 0.0) || !(scale[l] < 1e6))
      return lf_fail(ctx, LF_ERR_INVALID, "lf_set_starburst_spectrum: scale must be positive and finite");
    sp.scale[l] = scale[l];
    for (int c = 0; c < 3; c++) sp.rgb[l][c] = rgb_weights[3 * l + c];
  }
  ctx->star_spec = sp;
  ctx->sample_valid = false;
  return LF_OK;
}

// ---------------------------------------------------------------- helper members ---------------
// The reference declares its ghost / starburst helpers as public members (pathtracer.h:45-57,
// :95-101); a drop-in defines every one of them, on the device like the frame-level calls.
static lf_status ghost_helper_ready(lf_ctx* ctx, const char* who) {
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, std::string(who) + " before lf_set_frame");
  if (!ctx->ap[LF_APERTURE_GHOST].valid) return lf_fail(ctx, LF_ERR_STATE, std::string(who) + ": ghost aperture not set");
  { const lf_status js = lf_comm_join(ctx); if (js != LF_OK) return js; }
  LF_HIP(ctx, hipSetDevice(ctx->device));
  return LF_OK;
}

lf_status lf_clear_ghost_buffer(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_clear_ghost_buffer before lf_set_frame");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  LF_HIP(ctx, hipMemsetAsync(ctx->ghost, 0, (size_t)ctx->W * ctx->H_alloc * 3 * sizeof(double), ctx->stream));
  ctx->ghost_valid = true;
  return LF_OK;
}

lf_status lf_draw_ghost(lf_ctx* ctx, int channel, float r1, float r2, int bbox[4]) {
  if (!ctx || channel < 0 || channel > 2) return LF_ERR_INVALID;
  lf_status st = ghost_helper_ready(ctx, "lf_draw_ghost");
  if (st != LF_OK) return st;
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_draw_ghost: no flare state (axis_ray)");
  return lfk_draw_ghost(ctx, channel, r1, r2, bbox);
}

lf_status lf_rasterize_textured_triangle(lf_ctx* ctx, const float v[12], const double colour[3], int bbox[4]) {
  if (!ctx || !v || !colour) return LF_ERR_INVALID;
  lf_status st = ghost_helper_ready(ctx, "lf_rasterize_textured_triangle");
  if (st != LF_OK) return st;
  return lfk_raster_triangle(ctx, v, colour, bbox);
}

lf_status lf_fill_textured_pixel(lf_ctx* ctx, const float v[12], int x, int y, const double colour[3]) {
  if (!ctx || !v || !colour) return LF_ERR_INVALID;
  lf_status st = ghost_helper_ready(ctx, "lf_fill_textured_pixel");
  if (st != LF_OK) return st;
  // "assumes in bounds" (pathtracer.cpp:307): the reference would write past its buffer
  if (x < 0 || y < 0 || x >= ctx->W || y >= ctx->H) return lf_fail(ctx, LF_ERR_INVALID, "lf_fill_textured_pixel: pixel outside the frame");
  return lfk_fill_pixel(ctx, v, x, y, colour);
}

lf_status lf_shift_vertex(lf_ctx* ctx, float x, float y, float scale, float shift_amount, double out_xy[2]) {
  if (!ctx || !out_xy) return LF_ERR_INVALID;
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_shift_vertex: no flare state (axis_ray)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  return lfk_shift_vertex(ctx, x, y, scale, shift_amount, out_xy);
}

lf_status lf_compute_phase(lf_ctx* ctx, int flare, double u, double v, double out_re_im[2], double screen_pos[2]) {
  if (!ctx || !out_re_im || flare < 0 || flare >= LF_MAX_FLARES) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_compute_phase before lf_set_frame");
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_compute_phase: no flare state");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  double o[4];
  lf_status st = lfk_compute_phase(ctx, flare, u, v, o);
  if (st != LF_OK) return st;
  out_re_im[0] = o[0]; out_re_im[1] = o[1];
  if (screen_pos) { screen_pos[0] = o[2]; screen_pos[1] = o[3]; }
  return LF_OK;
}

lf_status lf_irradiance_falloff(lf_ctx* ctx, int x, int y, double radius, double rgb[3]) {
  if (!ctx || !rgb) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_irradiance_falloff before lf_set_frame");
  if (x < 0 || y < 0 || x >= ctx->W || y >= ctx->H) return lf_fail(ctx, LF_ERR_INVALID, "lf_irradiance_falloff: pixel outside the frame");
  if (!ctx->flares_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_irradiance_falloff: no flare state");
  if (ctx->jitter_mode == 0 && !ctx->jitter_table_valid)
    return lf_fail(ctx, LF_ERR_STATE, "MT19937 jitter selected but no table (frame was resized?)");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  return lfk_irradiance_falloff(ctx, x, y, radius, rgb);
}

// ---------------------------------------------------------------- lens file ---------------------
lf_status lf_load_lens_file(lf_ctx* ctx, const char* path) {
  if (!ctx || !path) return LF_ERR_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) return lf_fail(ctx, LF_ERR_INVALID, std::string("lf_load_lens_file: cannot open ") + path);
  // rows: radius thickness n_1 ... n_L semi_aperture; '#' starts a comment; radius 0 with index 0 = the stop
  std::vector<std::vector<double>> rows;
  double sensor_w = 36.0;
  char line[1024];
  bool bad = false;
  while (std::fgets(line, sizeof(line), f)) {
    if (char* h = std::strchr(line, '#')) *h = 0;
    std::vector<double> v;
    char* p = line;
    bool keyword = false, skip_line = false;
    while (*p) {
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      if (!*p) break;
      if (v.empty() && !keyword && std::strncmp(p, "sensor_width_mm", 15) == 0) { keyword = true; p += 15; continue; }
      // (the wavelengths of the index columns: documentation for the host, the march only needs the indices)
      if (v.empty() && !keyword && std::strncmp(p, "lambda_nm", 9) == 0) { skip_line = true; break; }
      char* e = nullptr;
      const double d = std::strtod(p, &e);
      if (e == p) { bad = true; break; }
      v.push_back(d);
      p = e;
    }
    if (bad) break;
    if (skip_line) continue;
    if (keyword) { if (v.size() != 1) { bad = true; break; } sensor_w = v[0]; continue; }
    if (!v.empty()) rows.push_back(v);
  }
  std::fclose(f);
  const int n = (int)rows.size();
  if (bad || n < 1 || n > LF_MAX_SURFACES) return lf_fail(ctx, LF_ERR_INVALID, "lf_load_lens_file: malformed prescription");
  const int nl = (int)rows[0].size() - 3;
  if (nl < 1 || nl > LF_MAX_LAMBDA) return lf_fail(ctx, LF_ERR_INVALID, "lf_load_lens_file: a row is radius thickness n_1..n_L semi_aperture");
  float radius[LF_MAX_SURFACES], thick[LF_MAX_SURFACES], semi[LF_MAX_SURFACES], ior[LF_MAX_LAMBDA * LF_MAX_SURFACES];
  int stop = -1;
  for (int k = 0; k < n; k++) {
    if ((int)rows[k].size() != nl + 3) return lf_fail(ctx, LF_ERR_INVALID, "lf_load_lens_file: rows of different length");
    radius[k] = (float)rows[k][0]; thick[k] = (float)rows[k][1]; semi[k] = (float)rows[k][nl + 2];
    const bool is_stop = rows[k][0] == 0 && rows[k][2] == 0;
    if (is_stop && stop < 0) stop = k;
    for (int l = 0; l < nl; l++) ior[l * n + k] = (is_stop && stop == k) ? 1.0f : (float)rows[k][2 + l];
  }
  return lf_set_lens(ctx, n, stop, nl, radius, thick, ior, semi, (float)sensor_w);
}

lf_status lf_get_lens_info(lf_ctx* ctx, int* n_surfaces, int* stop_index, int* n_lambda, float* sensor_width_mm,
                           double* efl_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_get_lens_info before lf_set_lens");
  if (n_surfaces) *n_surfaces = ctx->raw_n;
  if (stop_index) *stop_index = ctx->raw_stop;
  if (n_lambda) *n_lambda = ctx->lens.n_lambda;
  if (sensor_width_mm) *sensor_width_mm = ctx->sensor_w_mm;
  if (efl_mm) {
    *efl_mm = 0.0;   // (an afocal prescription has none)
    (void)lf_paraxial_efl(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                          ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, efl_mm);
  }
  return LF_OK;
}

// ---------------------------------------------------------------- pupil target -------------------
lf_status lf_set_pupil_target(lf_ctx* ctx, float radius_mm, float z_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_set_pupil_target before lf_set_lens");
  if (radius_mm > 0.0f) {
    if (!(z_mm < ctx->lens.z_sensor) || !std::isfinite(z_mm) || !std::isfinite(radius_mm))
      return lf_fail(ctx, LF_ERR_INVALID, "pupil target: the disc must lie in front of the sensor plane");
    ctx->pupil_target_h = radius_mm; ctx->pupil_target_z = z_mm;
  } else {
    ctx->pupil_target_h = 0.0f; ctx->pupil_target_z = 0.0f;
  }
  lf_apply_pupil_target(ctx);
  return LF_OK;
}

lf_status lf_get_pupil_target(lf_ctx* ctx, float* radius_mm, float* z_mm, float* z_sensor_mm) {
  if (!ctx) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_get_pupil_target before lf_set_lens");
  if (radius_mm) *radius_mm = ctx->lens.pupil_h;
  if (z_mm) *z_mm = ctx->lens.pupil_z;
  if (z_sensor_mm) *z_sensor_mm = ctx->lens.z_sensor;
  return LF_OK;
}

lf_status lf_paraxial_exit_pupil(int n, int stop, const float* radius, const float* thickness, const float* ior_row,
                                 double* z_mm, double* magnification) {
  if (n < 1 || n > LF_MAX_SURFACES || stop < 0 || stop >= n || !radius || !thickness || !ior_row || !z_mm || !magnification)
    return LF_ERR_INVALID;
  // the stop's centre imaged by the interfaces behind it: ray (height y, angle u) from the stop plane,
  // T(d) = [[1, d], [0, 1]], R(c, n1, n2) = [[1, 0], [c (n1 - n2) / n2, n1 / n2]] (pathtracer.cpp:527-533)
  double A = 1, B = 0, Cc = 0, D = 1;   // system matrix stop plane -> rear vertex
  double n1 = 1.0, z = 0.0, z_stop = 0.0;
  for (int k = 0; k < n; k++) {
    if (k == stop) z_stop = z;
    if (k < stop) n1 = ior_row[k];      // the medium the stop sits in
    z += thickness[k];
  }
  double z_rear = z - thickness[n - 1];
  double nm = n1;
  for (int k = stop; k < n; k++) {
    if (k > stop) {
      const double c = radius[k] == 0.0f ? 0.0 : 1.0 / (double)radius[k], n2 = ior_row[k];
      if (!(n2 >= 1.0)) return LF_ERR_INVALID;
      const double r10 = c * (nm - n2) / n2, r11 = nm / n2;
      const double c2 = r10 * A + r11 * Cc, d2 = r10 * B + r11 * D;
      Cc = c2; D = d2;
      nm = n2;
    }
    if (k + 1 < n) {
      const double d = thickness[k];
      A += d * Cc; B += d * D;
    }
  }
  (void)z_stop;
  if (D == 0.0) return LF_ERR_INVALID;   // the stop is imaged at infinity (image-space telecentric)
  const double l = -B / D;               // image distance behind the rear vertex (+z = towards the sensor)
  *z_mm = z_rear + l;
  *magnification = A + l * Cc;
  return LF_OK;
}

lf_status lf_aim_at_exit_pupil(lf_ctx* ctx, float margin) {
  if (!ctx || !(margin > 0.0f)) return LF_ERR_INVALID;
  if (!ctx->lens_valid) return lf_fail(ctx, LF_ERR_STATE, "lf_aim_at_exit_pupil before lf_set_lens");
  if (ctx->raw_stop < 0) return lf_fail(ctx, LF_ERR_INVALID, "lf_aim_at_exit_pupil: the prescription has no stop");
  if (!ctx->ap[LF_APERTURE_STARBURST].valid)
    return lf_fail(ctx, LF_ERR_STATE, "aperture mask (LF_APERTURE_STARBURST slot) not set");
  double z = 0, m = 0;
  lf_status st = lf_paraxial_exit_pupil(ctx->raw_n, ctx->raw_stop, ctx->raw_radius, ctx->raw_thickness,
                                        ctx->raw_ior + (size_t)(ctx->lens.n_lambda / 2) * ctx->raw_n, &z, &m);
  if (st != LF_OK) return lf_fail(ctx, LF_ERR_INVALID, "lf_aim_at_exit_pupil: the stop has no finite paraxial image");
  // the part of the stop that is open: the circle around the mask's non-zero texels, never more than
  // the housing
  const double open = std::min(1.0, ctx->ap[LF_APERTURE_STARBURST].open_radius);
  const double r = (double)ctx->lens.stop_h * open * std::fabs(m) * (double)margin;
  return lf_set_pupil_target(ctx, (float)r, (float)z);
}

lf_status lf_set_ghost_accumulate(lf_ctx* ctx, int on) {
  if (!ctx) return LF_ERR_INVALID;
  ctx->ghost_accumulate = on != 0;
  return LF_OK;
}

// one float in, one float out, as a device instruction computes it (lf_native_sqrt, lf_native_rcp)
static lf_status native_unary(lf_ctx* ctx, const float* x, float* y, size_t n,
                              lf_status (*launch)(lf_ctx*, const float*, float*, size_t)) {
  if (!ctx || (n && (!x || !y))) return LF_ERR_INVALID;
  if (n == 0) return LF_OK;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  float *d_x = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc((void**)&d_x, n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d_y, n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(d_x, x, n * sizeof(float), hipMemcpyHostToDevice);
  lf_status st = LF_OK;
  if (e == hipSuccess) st = launch(ctx, d_x, d_y, n);
  if (e == hipSuccess && st == LF_OK) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && st == LF_OK) e = hipMemcpy(y, d_y, n * sizeof(float), hipMemcpyDeviceToHost);
  if (d_x) (void)hipFree(d_x);
  if (d_y) (void)hipFree(d_y);
  if (st != LF_OK) return st;
  LF_HIP(ctx, e);
  return LF_OK;
}

lf_status lf_native_sqrt(lf_ctx* ctx, const float* x, float* y, size_t n) {
  return native_unary(ctx, x, y, n, lfk_native_sqrt);
}

lf_status lf_native_rcp(lf_ctx* ctx, const float* x, float* y, size_t n) {
  return native_unary(ctx, x, y, n, lfk_native_rcp);
}

lf_status lf_get_counters(lf_ctx* ctx, lf_counters* out) {
  if (!ctx || !out) return LF_ERR_INVALID;
  unsigned long long c[8];
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(c, ctx->counters_dev, sizeof(c), hipMemcpyDeviceToHost));
  out->rays_launched = c[0]; out->surface_events = c[1]; out->rays_clipped_stop = c[2];
  out->rays_vignetted = c[3]; out->rays_tir = c[4]; out->rays_reached_scene = c[5];
  out->rays_hit_light = c[6];
  return LF_OK;
}

lf_status lf_get_executed_events(lf_ctx* ctx, uint64_t* out) {
  if (!ctx || !out) return LF_ERR_INVALID;
  unsigned long long c = 0;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(&c, ctx->counters_dev + 7, sizeof(c), hipMemcpyDeviceToHost));
  *out = c;
  return LF_OK;
}

lf_status lf_get_march_stats(lf_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return LF_ERR_INVALID;
  unsigned long long c[kMarchCounterSlots];
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LF_HIP(ctx, hipMemcpy(c, ctx->counters_dev, sizeof(c), hipMemcpyDeviceToHost));
  out[0] = c[7]; out[1] = c[8]; out[2] = c[9]; out[3] = 0;
#ifdef LF_EXPERIMENTS
  if (std::getenv("LF_MARCH_PRINT_HIST")) {   // instrumented builds only (LF_MARCH_LIVE_HIST / _PAIR_STATS): the slots stay 0 otherwise
    static const char* kinds[3] = {"refraction", "mirror_or_flat", "stop"};
    for (int k = 0; k < 3; k++) {
      std::fprintf(stderr, "LIVE_HIST %s", kinds[k]);
      for (int b = 0; b < 9; b++) std::fprintf(stderr, " %llu", c[kMarchHistSlot + 9 * k + b]);
      std::fprintf(stderr, "\n");
    }
    for (int q = 0; q < ctx->pairs.n && q < 64; q++)
      if (c[kMarchPairSlot + q] | c[kMarchPairSlot + 64 + q])
        std::fprintf(stderr, "PAIR_STAT %d %d %d %llu %llu %llu\n", q, ctx->pairs.ij[q][0], ctx->pairs.ij[q][1],
                     c[kMarchPairSlot + q], c[kMarchPairSlot + 64 + q], c[kMarchPairSlot + 128 + q]);
  }
#endif
  return LF_OK;
}

lf_status lf_reset_counters(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  LF_HIP(ctx, hipMemsetAsync(ctx->counters_dev, 0, kMarchCounterSlots * sizeof(unsigned long long), ctx->stream));
  ctx->cull_audit_rays = 0; ctx->cull_audit_lit = 0; ctx->cull_audit_tripped = 0;
  return LF_OK;
}

// ---------------------------------------------------------------- measurement ----------------
lf_status lf_timing_enable(lf_ctx* ctx, int on) {
  if (!ctx) return LF_ERR_INVALID;
  ctx->timing = on != 0;
  return LF_OK;
}

lf_status lf_timing_reset(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& t : ctx->timed) { ctx->event_pool.push_back(t.start); ctx->event_pool.push_back(t.stop); }
  ctx->timed.clear();
  for (int k = 0; k < LFK_COUNT; k++) { ctx->timed_ms[k] = 0.0; ctx->timed_n[k] = 0; }
  return LF_OK;
}

lf_status lf_timing_get(lf_ctx* ctx, const char* kernel, int* launches, double* total_ms) {
  if (!ctx || !kernel || !launches || !total_ms) return LF_ERR_INVALID;
  int id = -1;
  for (int k = 0; k < LFK_COUNT; k++) if (std::strcmp(kernel, kKernelNames[k]) == 0) id = k;
  if (id < 0) return lf_fail(ctx, LF_ERR_INVALID, "unknown kernel name");
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) LF_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));   // "exchange" runs there
  timing_fold(ctx, true);
  *launches = ctx->timed_n[id];
  *total_ms = ctx->timed_ms[id];
  return LF_OK;
}

}  // extern "C"
