// lf_group.hip -- multi-GPU inside the C ABI (SURVEY.md section 8 row e; north star: "sensor tiles
// shard naturally across the 8 GPUs of one node with a final RCCL gather over xGMI").
//
// The reference scales a frame over std::threads that pull 32x32 tiles from a mutex-guarded queue
// (src/pathtracer/raytraced_renderer.cpp:314-328, :352-354, :681-715; src/util/work_queue.h:11-51).
// Here the unit is the march's 8-row sensor tile row, dealt round-robin: tile row t belongs to rank
// t % n (lf_set_row_interleave), every rank renders its tile rows into its own full-frame buffer, and
// the ONLY data-path collective is the exchange of finished tile rows: the frame viewed as
// [groups][n][tile row] is completed everywhere by ONE ncclAllGather per frame (pack this rank's
// slots, gather, unpack -- two device copies of 1/n and 1 frame).  xGMI is point to point (7 links
// x ~153 GB/s per GPU), the ring all-gather of a 49.8 MB f64 1080p frame moves 43.6 MB into every
// GPU: well under a millisecond of wire time against an 18 ms share of the march at n = 8.
//
// Two shapes, one exchange:
//   lf_comm_*   one process per GPU (the driver's torchrun launch): rank 0 makes the id, the host
//               shares it by whatever it has (MPI, torch.distributed, a file), every rank attaches.
//   lf_group_*  one process, n devices (a C++ host application such as the CGL app): one context +
//               stream per device, ncclCommInitAll, work issued from one host thread per device.
// RCCL is resolved with dlopen at first use: the library has no link-time dependency on it, and a
// single-GPU host never loads it.  Devices listed twice in a group (a rehearsal on a one-GPU box)
// cannot form an RCCL communicator; the group then exchanges with peer copies on the streams.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>   // types and prototypes only: the entry points are resolved at run time

#include "lf_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;   // (group work runs on one host thread per device)
  std::call_once(once, []() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) { r.error = "librccl.so.1 not found"; return; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(r.handle, name); if (!p) ok = false; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.handle); r.handle = nullptr; r.error = "librccl lacks an entry point"; }
  });
  return r.handle ? &r : nullptr;
}

// frame viewed as [groups][world][e doubles]: slot (g, rank) -> send[g].  T = double: the exchange
// moves the sensor values as they are; T = float (lf_comm_set_exchange_precision(32)): half the bytes
// on the wire, the rows a rank RECEIVES are rounded to float (its own rows stay as rendered)
template <typename T>
__global__ void k_pack_rows(const double* __restrict__ frame, T* __restrict__ send, int rank, int world,
                            size_t groups, size_t e) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * e) return;
  const size_t g = i / e, k = i - g * e;
  send[i] = (T)frame[(g * world + rank) * e + k];
}
// recv = [world][groups][e] -> frame[g][r]; the rank's own slots are already in place
template <typename T>
__global__ void k_unpack_rows(const T* __restrict__ recv, double* __restrict__ frame, int rank, int world,
                              size_t groups, size_t e) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)world * groups * e) return;
  const size_t r = i / (groups * e), rem = i - r * groups * e, g = rem / e, k = rem - g * e;
  if ((int)r == rank) return;
  frame[(g * world + r) * e + k] = (double)recv[i];
}

// ... and the frame dealt by BLOCKS of 64 x 64 pixels (lf_set_block_deal): slot (g, rank) = block g * world + rank, its
// 64 x 64 x 3 values row by row (a block that reaches over the frame's edge -- or past its last block -- travels padded)
constexpr int kBlk = 1 << kDealBlockLog2;
constexpr size_t kBlkE = (size_t)kBlk * kBlk * 3;
__device__ __forceinline__ bool block_slot(size_t g, int r, int world, size_t k, int W, int H, int bx, int nblk, size_t& at) {
  const size_t b = g * (size_t)world + (size_t)r;
  if (b >= (size_t)nblk) return false;
  const int by = (int)(b / (size_t)bx), bxx = (int)(b - (size_t)by * bx);
  const int ry = (int)(k / ((size_t)kBlk * 3)), rem = (int)(k - (size_t)ry * kBlk * 3), rx = rem / 3, c = rem - 3 * rx;
  const int x = bxx * kBlk + rx, y = by * kBlk + ry;
  at = ((size_t)y * W + x) * 3 + c;
  return x < W && y < H;
}
template <typename T>
__global__ void k_pack_blocks(const double* __restrict__ frame, T* __restrict__ send, int rank, int world, size_t groups,
                              int W, int H, int bx, int nblk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * kBlkE) return;
  size_t at;
  send[i] = block_slot(i / kBlkE, rank, world, i % kBlkE, W, H, bx, nblk, at) ? (T)frame[at] : (T)0;
}
template <typename T>
__global__ void k_unpack_blocks(const T* __restrict__ recv, double* __restrict__ frame, int rank, int world, size_t groups,
                                int W, int H, int bx, int nblk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)world * groups * kBlkE) return;
  const size_t r = i / (groups * kBlkE), rem = i - r * groups * kBlkE;
  if ((int)r == rank) return;
  size_t at;
  if (block_slot(rem / kBlkE, (int)r, world, rem % kBlkE, W, H, bx, nblk, at)) frame[at] = (double)recv[i];
}

double* frame_buffer(lf_ctx* ctx, int which) { return which == 0 ? ctx->sample : which == 1 ? ctx->ghost : ctx->star; }

// the exchange's unit: a tile row (8 rows x W x 3) or a block (64 x 64 x 3); `groups` units per rank
struct Shape { size_t groups, e; };
Shape shape(const lf_ctx* ctx, int world) {
  if (ctx->deal_by_block) {
    const size_t nblk = (size_t)((ctx->W + kBlk - 1) / kBlk) * (size_t)((ctx->H + kBlk - 1) / kBlk);
    return Shape{(nblk + world - 1) / world, kBlkE};
  }
  const size_t ntrows = (size_t)(ctx->H + 7) / 8;
  return Shape{(ntrows + world - 1) / world, (size_t)8 * ctx->W * 3};
}

lf_status ensure_staging(lf_ctx* ctx, int world) {
  const Shape s = shape(ctx, world);
  const size_t need = (size_t)(world + 1) * s.groups * s.e;   // send [groups][e] + recv [world][groups][e]
  if (need <= ctx->comm_stage_cap) return LF_OK;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) LF_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
  if (ctx->comm_stage) (void)hipFree(ctx->comm_stage);
  ctx->comm_stage = nullptr; ctx->comm_stage_cap = 0;
  LF_HIP(ctx, hipMalloc((void**)&ctx->comm_stage, need * sizeof(double)));
  ctx->comm_stage_cap = need;
  return LF_OK;
}

// staging layout in elements of the exchange type: send [groups][e], then recv [world][groups][e]
// (sized for doubles, so the float exchange fits as well)
void* stage_recv(lf_ctx* ctx, size_t cnt) {
  return ctx->comm_f32 ? (void*)((float*)ctx->comm_stage + cnt) : (void*)(ctx->comm_stage + cnt);
}

lf_status launch_pack(lf_ctx* ctx, int which, int rank, int world, hipStream_t stream = nullptr) {
  const Shape s = shape(ctx, world);
  const size_t n = s.groups * s.e;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t q = stream ? stream : ctx->stream;
  if (ctx->deal_by_block) {
    const int bx = (ctx->W + kBlk - 1) / kBlk, nblk = bx * ((ctx->H + kBlk - 1) / kBlk);
    if (ctx->comm_f32)
      hipLaunchKernelGGL(k_pack_blocks<float>, grid, dim3(256), 0, q, frame_buffer(ctx, which), (float*)ctx->comm_stage, rank, world,
                         s.groups, ctx->W, ctx->H, bx, nblk);
    else
      hipLaunchKernelGGL(k_pack_blocks<double>, grid, dim3(256), 0, q, frame_buffer(ctx, which), ctx->comm_stage, rank, world, s.groups,
                         ctx->W, ctx->H, bx, nblk);
  } else if (ctx->comm_f32)
    hipLaunchKernelGGL(k_pack_rows<float>, grid, dim3(256), 0, q, frame_buffer(ctx, which), (float*)ctx->comm_stage,
                       rank, world, s.groups, s.e);
  else
    hipLaunchKernelGGL(k_pack_rows<double>, grid, dim3(256), 0, q, frame_buffer(ctx, which), ctx->comm_stage, rank,
                       world, s.groups, s.e);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

lf_status launch_unpack(lf_ctx* ctx, int which, int rank, int world, hipStream_t stream = nullptr) {
  const Shape s = shape(ctx, world);
  const size_t n = (size_t)world * s.groups * s.e;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t q = stream ? stream : ctx->stream;
  if (ctx->deal_by_block) {
    const int bx = (ctx->W + kBlk - 1) / kBlk, nblk = bx * ((ctx->H + kBlk - 1) / kBlk);
    if (ctx->comm_f32)
      hipLaunchKernelGGL(k_unpack_blocks<float>, grid, dim3(256), 0, q, (const float*)stage_recv(ctx, s.groups * s.e),
                         frame_buffer(ctx, which), rank, world, s.groups, ctx->W, ctx->H, bx, nblk);
    else
      hipLaunchKernelGGL(k_unpack_blocks<double>, grid, dim3(256), 0, q, (const double*)stage_recv(ctx, s.groups * s.e),
                         frame_buffer(ctx, which), rank, world, s.groups, ctx->W, ctx->H, bx, nblk);
  } else if (ctx->comm_f32)
    hipLaunchKernelGGL(k_unpack_rows<float>, grid, dim3(256), 0, q, (const float*)stage_recv(ctx, s.groups * s.e),
                       frame_buffer(ctx, which), rank, world, s.groups, s.e);
  else
    hipLaunchKernelGGL(k_unpack_rows<double>, grid, dim3(256), 0, q, (const double*)stage_recv(ctx, s.groups * s.e),
                       frame_buffer(ctx, which), rank, world, s.groups, s.e);
  LF_HIP(ctx, hipGetLastError());
  return LF_OK;
}

// the exchange changes rows of a frame buffer: a tonemapped copy of the sensor buffer is stale then
void invalidate_tonemap(lf_ctx* ctx, int which) {
  if (which == 0) ctx->rgba_y0 = ctx->rgba_y1 = 0;
}

lf_status check_gather_args(lf_ctx* ctx, int which, int world) {
  if (!ctx || which < 0 || which > 2) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "gather before lf_set_frame");
  if (ctx->row_period != world)
    return lf_fail(ctx, LF_ERR_STATE, "gather: the row interleave does not match the communicator (frame resized?)");
  const Shape s = shape(ctx, world);
  if (!ctx->deal_by_block && (size_t)ctx->H_alloc * ctx->W * 3 < s.groups * world * s.e)
    return lf_fail(ctx, LF_ERR_STATE, "gather: frame buffers are not padded for this world size");
  return LF_OK;
}

// What ONE rank hands to the collective, in one place: ncclAllGather(send, recv, count, type) delivers rank
// q's `count` elements at recv + q * count.  lf_comm_gather, lf_comm_gather_async, lf_group_gather's RCCL
// branch and its peer-copy stand-in (a group whose devices repeat cannot form a communicator) all take their
// pointers and counts from here, and lf_comm_exchange_plan shows them to the tests.
struct ExchangePlan {
  void* send;          // [groups][tile row]: this rank's tile rows, packed
  void* recv;          // [world][groups][tile row]
  size_t count;        // elements per rank = groups * 8 rows * W * 3
  size_t esz;          // bytes per element (double, or float under lf_comm_set_exchange_precision(32))
  ncclDataType_t type;
};
ExchangePlan exchange_plan(lf_ctx* ctx, int world) {
  const Shape s = shape(ctx, world);
  ExchangePlan p;
  p.count = s.groups * s.e;
  p.esz = ctx->comm_f32 ? sizeof(float) : sizeof(double);
  p.type = ctx->comm_f32 ? ncclFloat : ncclDouble;
  p.send = ctx->comm_stage;
  p.recv = stage_recv(ctx, p.count);
  return p;
}

// tests only: run the whole exchange (pack, all-gather, unpack) even with a single rank
bool force_exchange(const lf_ctx* ctx) { return ctx->comm_force_exchange; }   // (lf_test_knob)

}  // namespace

struct lf_group {
  std::vector<lf_ctx*> ctx;
  std::vector<int> devices;
  bool rccl = false;   // false: duplicate devices (rehearsal) -> peer copies instead of a communicator
  std::string err;
};

lf_status lf_comm_join(lf_ctx* ctx) {
  if (ctx && ctx->comm_pending) {
    LF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_ev_done, 0));
    ctx->comm_pending = false;
  }
  return LF_OK;
}

lf_status lf_comm_allgather_u64_inplace(lf_ctx* ctx, unsigned long long* base, size_t count_per_rank) {
  if (ctx->comm_poisoned.load()) return LF_ERR_STATE;
  if (!ctx->comm) return lf_fail(ctx, LF_ERR_STATE, "shared cull table: no communicator (lf_comm_init_rank)");
  Rccl* r = rccl();
  if (!r) return lf_fail(ctx, LF_ERR_STATE, "RCCL is not available");
  ctx->comm_busy.fetch_add(1);
  struct Leave { lf_ctx* c; ~Leave() { c->comm_busy.fetch_sub(1); } } leave{ctx};
  LF_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->comm_stream) {
    LF_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_main, hipEventDisableTiming));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_pack, hipEventDisableTiming));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_done, hipEventDisableTiming));
  }
  if (!ctx->comm_ev_table) {
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_table, hipEventDisableTiming));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_table_done, hipEventDisableTiming));
  }
  LF_HIP(ctx, hipEventRecord(ctx->comm_ev_table, ctx->stream));
  LF_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev_table, 0));
  // in place: this rank's slab already lies where the collective would put it
  const ncclResult_t rc = r->AllGather(base + (size_t)ctx->comm_rank * count_per_rank, base, count_per_rank, ncclUint64,
                                       (ncclComm_t)ctx->comm, ctx->comm_stream);
  if (ctx->comm_poisoned.load()) return LF_ERR_STATE;
  if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclAllGather (cull table): ") + r->GetErrorString(rc));
  LF_HIP(ctx, hipEventRecord(ctx->comm_ev_table_done, ctx->comm_stream));
  LF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_ev_table_done, 0));
  return LF_OK;
}

extern "C" {

// ---------------------------------------------------------------- one process per GPU -----------
lf_status lf_comm_get_unique_id(unsigned char id[LF_COMM_ID_BYTES]) {
  if (!id) return LF_ERR_INVALID;
  Rccl* r = rccl();
  if (!r) return LF_ERR_STATE;
  static_assert(LF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) return LF_ERR_HIP;
  std::memcpy(id, u.internal, LF_COMM_ID_BYTES);
  return LF_OK;
}

// (a call that may block: counted while inside; see lf_ctx::comm_poisoned)
struct CommBusy {
  lf_ctx* c;
  explicit CommBusy(lf_ctx* ctx) : c(ctx) { c->comm_busy.fetch_add(1); }
  ~CommBusy() { c->comm_busy.fetch_sub(1); }
};
#define LF_COMM_REFUSE_POISONED(ctx)                                                                             \
  do { if ((ctx)->comm_poisoned.load()) return LF_ERR_STATE; } while (0)   /* (no lf_fail: the error string is the blocked thread's too) */

lf_status lf_comm_init_rank(lf_ctx* ctx, int nranks, int rank, const unsigned char id[LF_COMM_ID_BYTES]) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return LF_ERR_INVALID;
  LF_COMM_REFUSE_POISONED(ctx);
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_comm_init_rank before lf_set_frame");
  Rccl* r = rccl();
  if (!r) return lf_fail(ctx, LF_ERR_STATE, "RCCL is not available: librccl.so.1 could not be loaded");
  if (ctx->comm) return lf_fail(ctx, LF_ERR_STATE, "this context already has a communicator");
  LF_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, LF_COMM_ID_BYTES);
  // the blocking call works on state THIS call owns; the context sees the communicator only if nobody has
  // given up on the call in the meantime
  ncclComm_t c = nullptr;
  ncclResult_t rc;
  {
    CommBusy busy(ctx);
    rc = r->CommInitRank(&c, nranks, u, rank);
    std::lock_guard<std::mutex> lock(ctx->comm_mu);
    if (ctx->comm_poisoned.load()) {
      if (rc == ncclSuccess && c) (void)r->CommAbort(c);   // it arrived late: never published, never leaked
      return LF_ERR_STATE;
    }
    if (rc == ncclSuccess) { ctx->comm = c; ctx->comm_nranks = nranks; ctx->comm_rank = rank; }
  }
  if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  return lf_set_row_interleave(ctx, rank, nranks);   // the deal: tile row t belongs to rank t % nranks
}

lf_status lf_comm_poison(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->comm_mu);
  ctx->comm_poisoned.store(true);
  return LF_OK;
}

int lf_comm_is_poisoned(lf_ctx* ctx) { return ctx && ctx->comm_poisoned.load() ? 1 : 0; }

lf_status lf_comm_gather(lf_ctx* ctx, int which) {
  if (!ctx) return LF_ERR_INVALID;
  LF_COMM_REFUSE_POISONED(ctx);
  CommBusy busy(ctx);
  if (!ctx->comm) return lf_fail(ctx, LF_ERR_STATE, "lf_comm_gather before lf_comm_init_rank");
  const int world = ctx->comm_nranks, rank = ctx->comm_rank;
  lf_status st = check_gather_args(ctx, which, world);
  if (st != LF_OK) return st;
  if (world == 1 && !force_exchange(ctx)) return LF_OK;
  Rccl* r = rccl();
  LF_HIP(ctx, hipSetDevice(ctx->device));
  if ((st = lf_comm_join(ctx)) != LF_OK) return st;     // an asynchronous exchange uses the same staging
  if ((st = ensure_staging(ctx, world)) != LF_OK) return st;
  invalidate_tonemap(ctx, which);
  hipEvent_t ev = lf_timing_begin(ctx, LFK_EXCHANGE);
  if ((st = launch_pack(ctx, which, rank, world)) != LF_OK) return st;
  const ExchangePlan x = exchange_plan(ctx, world);
  const ncclResult_t rc = r->AllGather(x.send, x.recv, x.count, x.type, (ncclComm_t)ctx->comm, ctx->stream);
  LF_COMM_REFUSE_POISONED(ctx);      // the host gave up on this call while it was blocked: touch nothing any more
  if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(rc));
  st = launch_unpack(ctx, which, rank, world);
  lf_timing_end(ctx, LFK_EXCHANGE, ev);
  return st;
}

// The exchange of the frame just rendered, on the context's SECOND stream: pack (after everything the
// main stream has queued so far), all-gather, unpack.  The main stream only waits for the pack -- the
// rows it is about to overwrite with the next frame have been copied out by then -- so the next
// frame's march overlaps this frame's exchange.  The rows the unpack writes belong to other ranks:
// nothing this rank renders touches them.  lf_comm_wait / lf_synchronize / the read functions join
// the two streams again.
lf_status lf_comm_gather_async(lf_ctx* ctx, int which) {
  if (!ctx) return LF_ERR_INVALID;
  LF_COMM_REFUSE_POISONED(ctx);
  CommBusy busy(ctx);
  if (!ctx->comm) return lf_fail(ctx, LF_ERR_STATE, "lf_comm_gather_async before lf_comm_init_rank");
  const int world = ctx->comm_nranks, rank = ctx->comm_rank;
  lf_status st = check_gather_args(ctx, which, world);
  if (st != LF_OK) return st;
  if (world == 1 && !force_exchange(ctx)) return LF_OK;
  Rccl* r = rccl();
  LF_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->comm_stream) {
    LF_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_main, hipEventDisableTiming));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_pack, hipEventDisableTiming));
    LF_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev_done, hipEventDisableTiming));
  }
  if ((st = ensure_staging(ctx, world)) != LF_OK) return st;
  invalidate_tonemap(ctx, which);
  const ExchangePlan x = exchange_plan(ctx, world);
  LF_HIP(ctx, hipEventRecord(ctx->comm_ev_main, ctx->stream));
  LF_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev_main, 0));
  hipEvent_t ev = lf_timing_begin(ctx, LFK_EXCHANGE, ctx->comm_stream);   // pack -> unpack, on the exchange's own stream
  if ((st = launch_pack(ctx, which, rank, world, ctx->comm_stream)) != LF_OK) return st;
  LF_HIP(ctx, hipEventRecord(ctx->comm_ev_pack, ctx->comm_stream));
  LF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_ev_pack, 0));
  const ncclResult_t rc = r->AllGather(x.send, x.recv, x.count, x.type, (ncclComm_t)ctx->comm, ctx->comm_stream);
  LF_COMM_REFUSE_POISONED(ctx);
  if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclAllGather: ") + r->GetErrorString(rc));
  if ((st = launch_unpack(ctx, which, rank, world, ctx->comm_stream)) != LF_OK) return st;
  lf_timing_end(ctx, LFK_EXCHANGE, ev, ctx->comm_stream);
  LF_HIP(ctx, hipEventRecord(ctx->comm_ev_done, ctx->comm_stream));
  ctx->comm_pending = true;
  return LF_OK;
}

// The cull pre-pass shared between the ranks (lf_cull.hip): every rank has built the slab of table rows that is its
// own; ONE in-place all-gather of equal slabs completes the table everywhere.  On the communicator's stream -- the only
// stream RCCL calls of this context are ever queued on, so that every rank issues them in one order (the previous
// frame's exchange, then this) -- after what the main stream has queued (the pre-pass), and the main stream goes on
// (the march) when it is done.
lf_status lf_comm_share_cull(lf_ctx* ctx, int on) {
  if (!ctx) return LF_ERR_INVALID;
  if (!on) { ctx->cull_share_how = 0; ctx->cull_share_n = 1; ctx->cull_share_rank = 0; return LF_OK; }
  LF_COMM_REFUSE_POISONED(ctx);
  if (!ctx->comm) return lf_fail(ctx, LF_ERR_STATE, "lf_comm_share_cull before lf_comm_init_rank");
  ctx->cull_share_how = 1; ctx->cull_share_n = ctx->comm_nranks; ctx->cull_share_rank = ctx->comm_rank;
  return LF_OK;
}

lf_status lf_comm_exchange_plan(lf_ctx* ctx, int world, uint64_t out[6]) {
  if (!ctx || !out || world < 1) return LF_ERR_INVALID;
  if (ctx->W == 0) return lf_fail(ctx, LF_ERR_STATE, "lf_comm_exchange_plan before lf_set_frame");
  const Shape s = shape(ctx, world);
  const size_t esz = ctx->comm_f32 ? sizeof(float) : sizeof(double);
  out[0] = s.groups * s.e;                       // sendcount (elements per rank) of the ncclAllGather
  out[1] = esz;                                  // bytes per element
  out[2] = s.groups * s.e * esz;                 // byte offset of the receive area behind the send area
  out[3] = (uint64_t)(world + 1) * s.groups * s.e * sizeof(double);   // staging bytes (sized for doubles)
  out[4] = s.groups;                             // groups of `world` consecutive tile rows
  out[5] = s.e;                                  // elements of one tile row: 8 rows x W x 3
  return LF_OK;
}

lf_status lf_comm_wait(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  return lf_comm_join(ctx);
}

lf_status lf_comm_available(void) { return rccl() ? LF_OK : LF_ERR_STATE; }

lf_status lf_comm_info(lf_ctx* ctx, int* nranks, int* rank) {
  if (!ctx) return LF_ERR_INVALID;
  int n = ctx->comm_nranks, k = ctx->comm_rank;
  if (ctx->comm) {   // what RCCL itself says about the communicator this context is attached to
    Rccl* r = rccl();
    ncclResult_t rc = r->CommCount((ncclComm_t)ctx->comm, &n);
    if (rc == ncclSuccess) rc = r->CommUserRank((ncclComm_t)ctx->comm, &k);
    if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclCommCount: ") + r->GetErrorString(rc));
  } else {
    n = 0; k = -1;   // no RCCL communicator (single GPU, or a rehearsal group exchanging with peer copies)
  }
  if (nranks) *nranks = n;
  if (rank) *rank = k;
  return LF_OK;
}

lf_status lf_comm_test(lf_ctx* ctx, int* done) {
  if (!ctx || !done) return LF_ERR_INVALID;
  LF_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = ctx->comm_pending ? hipEventQuery(ctx->comm_ev_done) : hipStreamQuery(ctx->stream);
  if (e == hipErrorNotReady) { *done = 0; return LF_OK; }
  LF_HIP(ctx, e);
  *done = 1;
  return LF_OK;
}

lf_status lf_comm_abort(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  void* comm = nullptr;
  {
    // (ncclCommAbort is what unblocks a collective stuck in another thread: it may run beside a blocked call, but
    // the communicator is taken out of the context under the lock, so that nobody publishes or reads it half-way)
    std::lock_guard<std::mutex> lock(ctx->comm_mu);
    comm = ctx->comm;
    ctx->comm = nullptr; ctx->comm_nranks = 1; ctx->comm_rank = 0;
    ctx->comm_pending = false;
    // (a table shared through this communicator: back to every rank building its own)
    if (ctx->cull_share_how == 1) { ctx->cull_share_how = 0; ctx->cull_share_n = 1; ctx->cull_share_rank = 0; ctx->cull_hash = 0; }
  }
  if (!comm) return LF_OK;
  Rccl* r = rccl();
  (void)hipSetDevice(ctx->device);
  const ncclResult_t rc = r->CommAbort((ncclComm_t)comm);   // ends the collectives in flight on this rank
  if (ctx->comm_busy.load() > 0) return LF_OK;   // a blocked call is on its way out: the streams are its to leave
  if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
  (void)hipStreamSynchronize(ctx->stream);
  if (rc != ncclSuccess) return lf_fail(ctx, LF_ERR_HIP, std::string("ncclCommAbort: ") + r->GetErrorString(rc));
  return LF_OK;
}

lf_status lf_comm_set_exchange_precision(lf_ctx* ctx, int bits) {
  if (!ctx || (bits != 32 && bits != 64)) return LF_ERR_INVALID;
  lf_status st = lf_comm_join(ctx);   // an exchange in flight keeps the layout it started with
  if (st != LF_OK) return st;
  LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->comm_f32 = bits == 32;
  return LF_OK;
}

lf_status lf_comm_destroy(lf_ctx* ctx) {
  if (!ctx) return LF_ERR_INVALID;
  if (ctx->comm_stream) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->comm_stream);
    (void)hipEventDestroy(ctx->comm_ev_main); (void)hipEventDestroy(ctx->comm_ev_pack);
    (void)hipEventDestroy(ctx->comm_ev_done);
    if (ctx->comm_ev_table) { (void)hipEventDestroy(ctx->comm_ev_table); (void)hipEventDestroy(ctx->comm_ev_table_done); }
    ctx->comm_ev_table = ctx->comm_ev_table_done = nullptr;
    (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr; ctx->comm_ev_main = ctx->comm_ev_pack = ctx->comm_ev_done = nullptr;
    ctx->comm_pending = false;
  }
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_nranks = 1; ctx->comm_rank = 0;
    if (ctx->cull_share_how == 1) { ctx->cull_share_how = 0; ctx->cull_share_n = 1; ctx->cull_share_rank = 0; ctx->cull_hash = 0; }
  }
  return LF_OK;
}

// ---------------------------------------------------------------- one process, n devices --------
lf_status lf_group_create(lf_group** out, int n, const int* devices) {
  if (!out || n < 1 || n > 64 || !devices) return LF_ERR_INVALID;
  *out = nullptr;
  lf_group* g = new lf_group();
  bool distinct = true;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) if (devices[i] == devices[j]) distinct = false;
  for (int i = 0; i < n; i++) {
    lf_ctx* c = nullptr;
    const lf_status st = lf_create(&c, devices[i]);
    if (st != LF_OK) { lf_group_destroy(g); return st; }
    g->ctx.push_back(c);
    g->devices.push_back(devices[i]);
  }
  if (n > 1 && distinct) {
    Rccl* r = rccl();
    if (!r) { lf_group_destroy(g); return LF_ERR_STATE; }
    std::vector<ncclComm_t> comms(n);
    if (r->CommInitAll(comms.data(), n, devices) != ncclSuccess) { lf_group_destroy(g); return LF_ERR_HIP; }
    for (int i = 0; i < n; i++) { g->ctx[i]->comm = comms[i]; g->ctx[i]->comm_nranks = n; g->ctx[i]->comm_rank = i; }
    g->rccl = true;
  } else {
    for (int i = 0; i < n; i++) { g->ctx[i]->comm_nranks = n; g->ctx[i]->comm_rank = i; }
  }
  *out = g;
  return LF_OK;
}

lf_status lf_group_destroy(lf_group* g) {
  if (!g) return LF_ERR_INVALID;
  for (lf_ctx* c : g->ctx) { (void)lf_comm_destroy(c); (void)lf_destroy(c); }
  delete g;
  return LF_OK;
}

int lf_group_size(const lf_group* g) { return g ? (int)g->ctx.size() : 0; }

lf_ctx* lf_group_ctx(lf_group* g, int rank) {
  return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr;
}

const char* lf_group_last_error(const lf_group* g) { return g ? g->err.c_str() : "null group"; }

lf_status lf_group_set_frame(lf_group* g, int width, int height) {
  if (!g) return LF_ERR_INVALID;
  const int n = (int)g->ctx.size();
  for (int r = 0; r < n; r++) {
    lf_status st = lf_set_frame(g->ctx[r], width, height);
    if (st == LF_OK) st = lf_set_row_interleave(g->ctx[r], r, n);
    if (st != LF_OK) { g->err = lf_last_error(g->ctx[r]); return st; }
  }
  return LF_OK;
}

// the group's frame dealt by blocks of 64 x 64 pixels (lf_set_block_deal on every context: on) or by tile rows (off, the default)
lf_status lf_group_set_block_deal(lf_group* g, int on) {
  if (!g) return LF_ERR_INVALID;
  const int n = (int)g->ctx.size();
  for (int r = 0; r < n; r++) {
    const lf_status st = on ? lf_set_block_deal(g->ctx[r], r, n) : lf_set_row_interleave(g->ctx[r], r, n);
    if (st != LF_OK) { g->err = lf_last_error(g->ctx[r]); return st; }
  }
  return LF_OK;
}

lf_status lf_group_for_each(lf_group* g, lf_group_fn fn, void* user) {
  if (!g || !fn) return LF_ERR_INVALID;
  const int n = (int)g->ctx.size();
  std::vector<lf_status> st(n, LF_OK);
  std::vector<std::thread> th;
  // one host thread per device, like the reference's worker threads (raytraced_renderer.cpp:352-354)
  for (int r = 1; r < n; r++) th.emplace_back([&, r]() { st[r] = fn(g->ctx[r], r, user); });
  st[0] = fn(g->ctx[0], 0, user);
  for (auto& t : th) t.join();
  for (int r = 0; r < n; r++)
    if (st[r] != LF_OK) { g->err = "rank " + std::to_string(r) + ": " + lf_last_error(g->ctx[r]); return st[r]; }
  return LF_OK;
}

lf_status lf_group_gather(lf_group* g, int which) {
  if (!g) return LF_ERR_INVALID;
  const int n = (int)g->ctx.size();
  if (n == 1) return LF_OK;
  auto hip_fail = [&](int r, const char* what, hipError_t e) {
    g->err = "rank " + std::to_string(r) + ": " + what + ": " + hipGetErrorString(e);
    return LF_ERR_HIP;
  };
#define LF_GROUP_HIP(r, expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(r, #expr, e_); } while (0)
  for (int r = 1; r < n; r++)
    if (g->ctx[r]->comm_f32 != g->ctx[0]->comm_f32) { g->err = "the contexts of a group must exchange at one precision"; return LF_ERR_STATE; }
  for (int r = 0; r < n; r++) {
    lf_ctx* c = g->ctx[r];
    lf_status st = check_gather_args(c, which, n);
    if (st == LF_OK) { LF_GROUP_HIP(r, hipSetDevice(c->device)); st = ensure_staging(c, n); }
    if (st == LF_OK) { invalidate_tonemap(c, which); st = launch_pack(c, which, r, n); }
    if (st != LF_OK) { g->err = lf_last_error(c); return st; }
  }
  if (g->rccl) {
    Rccl* rc = rccl();
    ncclResult_t e = rc->GroupStart();
    if (e != ncclSuccess) { g->err = std::string("ncclGroupStart: ") + rc->GetErrorString(e); return LF_ERR_HIP; }
    for (int r = 0; r < n; r++) {
      lf_ctx* c = g->ctx[r];
      const hipError_t he = hipSetDevice(c->device);
      if (he != hipSuccess) { (void)rc->GroupEnd(); return hip_fail(r, "hipSetDevice", he); }
      const ExchangePlan x = exchange_plan(c, n);
      e = rc->AllGather(x.send, x.recv, x.count, x.type, (ncclComm_t)c->comm, c->stream);
      if (e != ncclSuccess) { (void)rc->GroupEnd(); g->err = std::string("ncclAllGather: ") + rc->GetErrorString(e); return LF_ERR_HIP; }
    }
    e = rc->GroupEnd();
    if (e != ncclSuccess) { g->err = std::string("ncclGroupEnd: ") + rc->GetErrorString(e); return LF_ERR_HIP; }
  } else {
    // rehearsal (devices listed twice): every context pulls the other contexts' packed rows with
    // peer copies on its own stream, after they have been packed
    for (int r = 0; r < n; r++) {
      LF_GROUP_HIP(r, hipSetDevice(g->ctx[r]->device));
      LF_GROUP_HIP(r, hipStreamSynchronize(g->ctx[r]->stream));
    }
    // (what ncclAllGather(x.send, x.recv, x.count, x.type) does: rank q's x.count elements land at
    // x.recv + q * x.count -- the same plan, executed with copies)
    for (int r = 0; r < n; r++) {
      lf_ctx* c = g->ctx[r];
      LF_GROUP_HIP(r, hipSetDevice(c->device));
      const ExchangePlan x = exchange_plan(c, n);
      for (int q = 0; q < n; q++) {
        const ExchangePlan xq = exchange_plan(g->ctx[q], n);
        if (xq.count != x.count || xq.esz != x.esz) { g->err = "the contexts of a group disagree about the exchange's shape"; return LF_ERR_STATE; }
        LF_GROUP_HIP(r, hipMemcpyPeerAsync((char*)x.recv + (size_t)q * x.count * x.esz, c->device, xq.send,
                                           g->ctx[q]->device, x.count * x.esz, c->stream));
      }
    }
  }
  for (int r = 0; r < n; r++) {
    LF_GROUP_HIP(r, hipSetDevice(g->ctx[r]->device));
    const lf_status st = launch_unpack(g->ctx[r], which, r, n);
    if (st != LF_OK) { g->err = lf_last_error(g->ctx[r]); return st; }
  }
  // the staging buffers are reused by the next gather: order the streams against each other
  for (int r = 0; r < n; r++) {
    LF_GROUP_HIP(r, hipSetDevice(g->ctx[r]->device));
    LF_GROUP_HIP(r, hipStreamSynchronize(g->ctx[r]->stream));
  }
#undef LF_GROUP_HIP
  return LF_OK;
}

// The cull pre-pass of the group's next lf_trace_ghosts(spp), shared between its devices: every context builds its
// slab (one host thread per device), ONE in-place all-gather of slabs -- ncclAllGather inside a group call, or the
// peer-copy stand-in of a rehearsal group -- and every context takes the completed table over.  The host calls it
// before the lf_group_for_each that renders (same inputs on every context, as for any launch of a group).
lf_status lf_group_share_cull(lf_group* g, int spp) {
  if (!g || spp < 1) return LF_ERR_INVALID;
  const int n = (int)g->ctx.size();
  if (n == 1) return LF_OK;
  if (g->ctx[0]->deal_by_block) return LF_OK;       // dealt by blocks: every context builds the rows it reads, nothing to share
  for (int r = 0; r < n; r++) {
    const lf_status st = lf_set_cull_share(g->ctx[r], r, n);
    if (st != LF_OK) { g->err = lf_last_error(g->ctx[r]); return st; }
  }
  struct Arg { int spp; } arg{spp};
  lf_status st = lf_group_for_each(g, [](lf_ctx* c, int, void* u) { return lf_cull_prepare(c, ((Arg*)u)->spp); }, &arg);
  if (st != LF_OK) return st;
  std::vector<void*> dev(n, nullptr);
  uint64_t entries = 0, per = 0;
  for (int r = 0; r < n; r++) {
    uint64_t e = 0, p = 0;
    st = lf_cull_table_view(g->ctx[r], &dev[r], &e, &p);
    if (st != LF_OK) { g->err = lf_last_error(g->ctx[r]); return st; }
    if (r == 0) { entries = e; per = p; }
    else if (e != entries || p != per) { g->err = "the contexts of a group disagree about the cull table (different inputs?)"; return LF_ERR_STATE; }
  }
  if (entries != 0) {
    if (g->rccl) {
      Rccl* rc = rccl();
      ncclResult_t e = rc->GroupStart();
      if (e != ncclSuccess) { g->err = std::string("ncclGroupStart: ") + rc->GetErrorString(e); return LF_ERR_HIP; }
      for (int r = 0; r < n; r++) {
        lf_ctx* c = g->ctx[r];
        if (hipSetDevice(c->device) != hipSuccess) { (void)rc->GroupEnd(); g->err = "hipSetDevice"; return LF_ERR_HIP; }
        unsigned long long* base = (unsigned long long*)dev[r];
        e = rc->AllGather(base + (size_t)r * per, base, (size_t)per, ncclUint64, (ncclComm_t)c->comm, c->stream);
        if (e != ncclSuccess) { (void)rc->GroupEnd(); g->err = std::string("ncclAllGather (cull table): ") + rc->GetErrorString(e); return LF_ERR_HIP; }
      }
      e = rc->GroupEnd();
      if (e != ncclSuccess) { g->err = std::string("ncclGroupEnd: ") + rc->GetErrorString(e); return LF_ERR_HIP; }
    } else {
      // rehearsal: what the all-gather delivers, with peer copies (lf_cull_table_view has synchronised every stream)
      for (int r = 0; r < n; r++) {
        lf_ctx* c = g->ctx[r];
        if (hipSetDevice(c->device) != hipSuccess) { g->err = "hipSetDevice"; return LF_ERR_HIP; }
        for (int q = 0; q < n; q++) {
          if (q == r) continue;
          const hipError_t he = hipMemcpyPeerAsync((unsigned long long*)dev[r] + (size_t)q * per, c->device,
                                                   (unsigned long long*)dev[q] + (size_t)q * per, g->ctx[q]->device,
                                                   (size_t)per * sizeof(unsigned long long), c->stream);
          if (he != hipSuccess) { g->err = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(he); return LF_ERR_HIP; }
        }
      }
      for (int r = 0; r < n; r++) {      // (a slab must not be overwritten... none is: every context only READS the others' own slabs)
        (void)hipSetDevice(g->ctx[r]->device);
        if (hipStreamSynchronize(g->ctx[r]->stream) != hipSuccess) { g->err = "hipStreamSynchronize"; return LF_ERR_HIP; }
      }
    }
  }
  return lf_group_for_each(g, [](lf_ctx* c, int, void*) { return lf_cull_commit(c); }, nullptr);
}

}  // extern "C"
