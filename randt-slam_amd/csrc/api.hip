// Host side of the C ABI declared in include/randt.h: contexts, device-resident map batches and
// the entry points that enqueue the gfx950 kernels.  There is deliberately NO CPU fallback: without
// a visible HIP device every entry point fails with RANDT_ERR_NODEVICE / RANDT_ERR_HIP.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "randt_internal.h"

int randt_set_error(randt_ctx* ctx, int status, const char* what, hipError_t e) {
  DeviceGuard dev_guard__(ctx);
  if (ctx) {
    ctx->last_error = what ? what : "";
    if (e != hipSuccess) {
      ctx->last_error += ": ";
      ctx->last_error += hipGetErrorString(e);
    }
  }
  return status;
}

// ---------------------------------------------------------------- who else is using the device (RANDT_SOLVE_AUTO) ----------
namespace {
std::mutex g_ctx_mu;
std::vector<randt_ctx*> g_ctxs;  // every live context of this process
inline long long now_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
}  // namespace

void randt_note_enqueue(randt_ctx* ctx) { ctx->last_enqueue_ns.store(now_ns(), std::memory_order_relaxed); }

bool randt_device_shared(randt_ctx* ctx) {
  const long long now = now_ns();
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  if (g_ctxs.size() < 2) return false;
  // pass 1: somebody enqueued a moment ago (the pipelined caller: every decision ends here, a few loads)
  for (randt_ctx* o : g_ctxs) {
    if (o == ctx || o->device != ctx->device) continue;
    const long long t = o->last_enqueue_ns.load(std::memory_order_relaxed);
    if (t != 0 && now - t < 100000ll) return true;
  }
  // pass 2: older stamps -- ask the stream, once per stamp
  for (randt_ctx* o : g_ctxs) {
    if (o == ctx || o->device != ctx->device) continue;
    long long t = o->last_enqueue_ns.load(std::memory_order_relaxed);
    if (t == 0) continue;
    const hipError_t e = hipStreamQuery(o->stream);
    if (e == hipErrorNotReady) return true;
    (void)hipGetLastError();
    o->last_enqueue_ns.compare_exchange_strong(t, 0ll, std::memory_order_relaxed);  // drained (unless it has enqueued again meanwhile)
  }
  return false;
}

// ---------------------------------------------------------------- batches used by other contexts' streams ----------
void randt_note_foreign_user(randt_ctx* user, const randt_maps* m) {
  for (auto& slot : m->foreign) {
    randt_ctx* cur = slot.load(std::memory_order_relaxed);
    if (cur == user) return;
    if (!cur) {
      if (slot.compare_exchange_strong(cur, user, std::memory_order_relaxed) || cur == user) return;
    }
  }
  m->foreign_overflow.store(true, std::memory_order_relaxed);
}

// Before a library-owned batch gives its block back: the owner's stream waits (device side) for a marker recorded NOW on the
// stream of every other live context that has used the batch -- whatever they enqueued on it so far is ahead of the marker, and
// the block's next user is enqueued on the owner's stream behind the wait.  A context that no longer exists has synchronised
// its stream when it was destroyed.  Falls back to a host wait on that stream if an event cannot be had.
static void wait_for_foreign_users(randt_maps* m) {
  bool any = m->foreign_overflow.load(std::memory_order_relaxed);
  for (auto& slot : m->foreign) any = any || slot.load(std::memory_order_relaxed) != nullptr;
  if (!any) return;
  const bool all = m->foreign_overflow.load(std::memory_order_relaxed);
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  for (randt_ctx* o : g_ctxs) {
    if (o == m->ctx) continue;
    bool used = all;
    for (auto& slot : m->foreign) used = used || slot.load(std::memory_order_relaxed) == o;
    if (!used) continue;
    bool ordered = false;
    {
      DeviceGuard on_user(o);
      if (!o->marker_ev && hipEventCreateWithFlags(&o->marker_ev, hipEventDisableTiming) != hipSuccess) {
        o->marker_ev = nullptr;
        (void)hipGetLastError();
      }
      if (o->marker_ev && hipEventRecord(o->marker_ev, o->stream) == hipSuccess) {
        DeviceGuard on_owner(m->ctx);
        ordered = hipStreamWaitEvent(m->ctx->stream, o->marker_ev, 0) == hipSuccess;
      }
      if (!ordered) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(o->stream);
      }
    }
    ++m->ctx->stats.foreign_waits;
  }
}

// ---------------------------------------------------------------- storage pool, counters, pinned ring ----------
hipError_t randt_hip_malloc(randt_ctx* ctx, void** p, size_t bytes) {
  if (ctx) ++ctx->stats.device_allocs;
  return hipMalloc(p, bytes);
}
hipError_t randt_hip_free(randt_ctx* ctx, void* p) {
  if (ctx) ++ctx->stats.device_frees;
  return hipFree(p);
}
hipError_t randt_sync(randt_ctx* ctx) {
  ++ctx->stats.stream_syncs;
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  ctx->last_enqueue_ns.store(0, std::memory_order_relaxed);  // drained: other contexts need not count this one as busy
  return e;
}

// A parked block serves a request if it is large enough and at most twice as large (+ 4 KB): a 64 KB scan map does not
// take a 520 KB submap block.  Sizes are rounded to 256 bytes so that equal requests meet equal blocks.
hipError_t randt_dev_alloc(randt_ctx* ctx, void** p, size_t bytes, size_t* granted) {
  const size_t want = (bytes + 255) & ~(size_t)255;
  auto it = ctx->pool_free.lower_bound(want);
  if (it != ctx->pool_free.end() && it->first <= 2 * want + 4096) {
    *p = it->second;
    if (granted) *granted = it->first;
    ctx->pool_bytes -= it->first;
    ctx->pool_free.erase(it);
    ++ctx->stats.pool_hits;
    return hipSuccess;
  }
  const hipError_t e = randt_hip_malloc(ctx, p, want);
  if (granted) *granted = want;
  return e;
}

void randt_dev_release(randt_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (bytes > ctx->pool_cap || ctx->pool_bytes + bytes > ctx->pool_cap) {
    // really freed: the stream may still use the block
    (void)randt_sync(ctx);
    (void)randt_hip_free(ctx, p);
    return;
  }
  ctx->pool_free.emplace(bytes, p);
  ctx->pool_bytes += bytes;
}

void* randt_pin_take(randt_ctx* ctx, size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  if (bytes > randt_ctx::kPinSegBytes) return nullptr;
  if (!ctx->pin_ring) {
    void* r = nullptr;
    if (hipHostMalloc(&r, randt_ctx::kPinSegs * randt_ctx::kPinSegBytes, hipHostMallocDefault) != hipSuccess || !r) {
      (void)hipGetLastError();
      return nullptr;
    }
    for (int i = 0; i < randt_ctx::kPinSegs; ++i)
      if (hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(r);
        for (int j = 0; j < i; ++j) (void)hipEventDestroy(ctx->pin_ev[j]);
        return nullptr;
      }
    ctx->pin_ring = static_cast<char*>(r);
    ctx->pin_cur = 0;
    ctx->pin_off = 0;
  }
  if (ctx->pin_off + bytes > randt_ctx::kPinSegBytes) {
    // this segment is full: everything enqueued so far may still read it -> an event behind it; the next segment is
    // reusable once ITS event (recorded a whole ring revolution ago) has passed -- normally long ago
    (void)hipEventRecord(ctx->pin_ev[ctx->pin_cur], ctx->stream);
    ctx->pin_pending[ctx->pin_cur] = true;
    ctx->pin_cur = (ctx->pin_cur + 1) % randt_ctx::kPinSegs;
    ctx->pin_off = 0;
    if (ctx->pin_pending[ctx->pin_cur]) {
      if (hipEventQuery(ctx->pin_ev[ctx->pin_cur]) != hipSuccess) {
        (void)hipGetLastError();
        ++ctx->stats.stream_syncs;
        (void)hipEventSynchronize(ctx->pin_ev[ctx->pin_cur]);
      }
      ctx->pin_pending[ctx->pin_cur] = false;
    }
  }
  char* out = ctx->pin_ring + (size_t)ctx->pin_cur * randt_ctx::kPinSegBytes + ctx->pin_off;
  ctx->pin_off += bytes;
  return out;
}

namespace {

int ensure_ws(randt_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return RANDT_OK;
  if (ctx->ws) {
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));
    RANDT_HIP_CHECK(ctx, randt_hip_free(ctx, ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->ws, want));
  ctx->ws_bytes = want;
  return RANDT_OK;
}

void fill_view(randt_maps* m, int n_maps, const randt_map_params* p, int cap) {
  m->p = *p;
  MapView& v = m->v;
  v.n_maps = n_maps;
  v.cap = cap;
  v.size_x = p->size_x;
  v.size_y = p->size_y;
  v.n_slots = p->size_x * p->size_y;
  v.res = p->resolution;
  // Map::initialize (ndt_map.cpp:19-20)
  v.offset_x = -(double)(uint32_t)p->size_x / 2.0 * p->resolution + p->center_x;
  v.offset_y = -(double)(uint32_t)p->size_y / 2.0 * p->resolution + p->center_y;
  v.rmax = (int)(p->max_neighbour_dist / p->resolution);  // ndt_map.cpp:117
  v.min_points = p->min_points_per_cell;
  v.pad_ = 0;
}

bool bad_params(const randt_map_params* p, int n_maps, int cap) {
  return !p || n_maps <= 0 || cap <= 0 || p->size_x <= 0 || p->size_y <= 0 || !(p->resolution > 0.0) ||
         (long long)p->size_x * p->size_y > (1ll << 30);
}

__global__ void k_clear(MapView v, int first, int count) {
  const int map = first + blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) v.counts[map] = 0;
  if (v.grid) {
    int32_t* g = v.grid + (size_t)map * v.n_slots;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.n_slots; i += gridDim.x * blockDim.x) g[i] = -1;
  }
}

// Self-test behind randt_ctx::lds_atomics_lane_ordered, under the build kernel's own conditions: FOUR wavefronts of one
// workgroup add to the SAME 64-bit LDS words at once, each in its own 16-bit field (k_ndt_build's (bin, wave) counters),
// in several collision patterns (all lanes on one word, groups of 2..32 neighbours, strided groups, irregular group sizes),
// three dependent rounds per pattern, 32 repetitions with the wavefronts deliberately out of step; every lane checks that the
// value it got back equals the number of LOWER lanes of its group (plus its wavefront's earlier total).  out[0] = 1 if all
// hold.  (The kernel that relies on the order also re-checks it on a sample in every launch: ndt_build.hip.)
__global__ __launch_bounds__(256) void k_lds_atomic_order_probe(int32_t* out) {
  __shared__ unsigned long long w[64];
  __shared__ int fails;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sh = 16 * wave;
  if (threadIdx.x == 0) fails = 0;
  int ok = 1;
  for (int rep = 0; rep < 32; ++rep) {
    for (int pattern = 0; pattern < 8; ++pattern) {
      if (threadIdx.x < 64) w[threadIdx.x] = 0ull;
      __syncthreads();
      int grp, rank, size;
      if (pattern < 6) {  // contiguous groups of 64, 32, 16, 8, 4, 2 lanes
        size = 64 >> pattern;
        grp = lane / size;
        rank = lane % size;
      } else if (pattern == 6) {  // interleaved: lanes with equal (lane % 8) collide
        size = 8;
        grp = lane % 8;
        rank = lane / 8;
      } else {  // irregular group sizes 1, 2, 3, ...
        int start = 0, g = 0;
        while (start + g + 1 <= lane) {
          start += g + 1;
          ++g;
        }
        grp = g;
        rank = lane - start;
        size = g + 1;
        if (start + size > 64) size = 64 - start;
      }
      for (int skew = 0; skew < ((wave * 7 + rep) & 15); ++skew) __builtin_amdgcn_s_sleep(1);  // the wavefronts arrive out of step
      for (int round = 0; round < 3; ++round) {  // a wavefront's LDS operations complete in program order
        const unsigned long long old = atomicAdd(&w[grp], 1ull << sh);
        if ((int)((old >> sh) & 0xffff) != round * size + rank) ok = 0;
      }
      __syncthreads();
    }
  }
  if (__ballot(ok != 0) != ~0ull && lane == 0) atomicAdd(&fails, 1);
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fails == 0 ? 1 : 0;
}

}  // namespace

extern "C" {

int randt_version(void) { return RANDT_VERSION; }

const char* randt_status_string(int status) {
  switch (status) {
    case RANDT_OK: return "ok";
    case RANDT_ERR_INVALID: return "invalid argument";
    case RANDT_ERR_HIP: return "HIP runtime error";
    case RANDT_ERR_UNSUPPORTED: return "size not supported by the kernels";
    case RANDT_ERR_NOMEM: return "out of memory";
    case RANDT_ERR_NODEVICE: return "no HIP device (this library has no CPU fallback)";
    default: return "unknown status";
  }
}

const char* randt_last_error(const randt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

void randt_matcher_params_default(randt_matcher_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  // config/parameters_indoor.yaml:7,9,24-39 (loop-closure refinement values) + Ceres 2.1.0 defaults
  p->loss_scale = 1.5;
  p->mu_scale = 1.5;
  p->loss_alpha = -2.0;
  p->loss_weight = 1.0;
  p->gnc_divisor = 1.3;
  p->gnc_steps = 2;
  p->max_iterations = 200;
  p->n_neighbours = 4;
  p->lookup_mahalanobis = 1;
  p->use_intensity = 1;
  p->parameterization = RANDT_PARAM_AMBIENT4;
  p->max_consecutive_invalid_steps = 5;
  p->function_tolerance = 1e-6;
  p->gradient_tolerance = 1e-10;
  p->parameter_tolerance = 1e-8;
  p->initial_radius = 1e4;
  p->max_radius = 1e16;
  p->min_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
}

int randt_ctx_create(int device, void* stream, randt_ctx** out) {
  if (!out) return RANDT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return RANDT_ERR_NODEVICE;
  if (device < 0 || device >= n) return RANDT_ERR_INVALID;
  randt_ctx* ctx = new (std::nothrow) randt_ctx();
  if (!ctx) return RANDT_ERR_NOMEM;
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  if (hipSetDevice(device) != hipSuccess) {
    delete ctx;
    return RANDT_ERR_HIP;
  }
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
    ctx->lds_limit = lds;
  if (const char* e = getenv("RANDT_SOLVE_RPB")) {
    const int r = atoi(e);
    if (r == 1 || r == 2 || r == 4 || r == 8) ctx->solve_rpb = r;
  }
  if (const char* e = getenv("RANDT_SOLVE_GROUP")) ctx->solve_group = atoi(e) ? 1 : 0;  // (unset: chosen from the launch size)
  if (const char* e = getenv("RANDT_SOLVE_SPLIT")) {
    const int w = atoi(e);
    if (w >= 0 && w <= 8) ctx->solve_split = w;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
  }
  if (const char* e = getenv("RANDT_SOLVE_BLOCK")) {
    int b = atoi(e);
    if (b == 64 || b == 128) ctx->solve_block = b;
  }
  if (const char* e = getenv("RANDT_ASSOC_STAGE_GRID")) ctx->assoc_stage_grid = atoi(e) ? 1 : 0;
  if (const char* e = getenv("RANDT_ASSOC_TP_CH")) ctx->assoc_tp_ch = atoi(e) > 0 ? atoi(e) : ctx->assoc_tp_ch;
  if (const char* e = getenv("RANDT_ASSOC_TP_PPW")) ctx->assoc_tp_ppw = atoi(e) > 0 ? atoi(e) : ctx->assoc_tp_ppw;
  if (const char* e = getenv("RANDT_BUILD_TILED")) ctx->build_tiled = atoi(e) ? 1 : 0;
  if (const char* e = getenv("RANDT_POOL_MAX_BYTES")) {
    const long long v = atoll(e);
    if (v >= 0) ctx->pool_cap = (size_t)v;  // 0: nothing is parked (every destroy synchronises and frees, like before the pool)
  }
  {
    // does this device serve colliding LDS atomics in lane order?  (one 64-thread launch; if the probe cannot run or says
    // no, the build kernels keep the ballot ranking, which assumes nothing)
    int32_t* d_flag = nullptr;
    int32_t h_flag = 0;
    if (hipMalloc(&d_flag, sizeof(int32_t)) == hipSuccess) {
      bool good = true;
      for (int rep = 0; rep < 4 && good; ++rep) {
        hipLaunchKernelGGL(k_lds_atomic_order_probe, dim3(1), dim3(256), 0, ctx->stream, d_flag);
        good = hipMemcpyAsync(&h_flag, d_flag, sizeof(h_flag), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
               hipStreamSynchronize(ctx->stream) == hipSuccess && h_flag == 1;
      }
      ctx->lds_atomics_lane_ordered = good ? 1 : 0;
      (void)hipFree(d_flag);
    }
    (void)hipGetLastError();
    if (const char* e = getenv("RANDT_BUILD_ATOMIC_RANK")) ctx->lds_atomics_lane_ordered = (atoi(e) && ctx->lds_atomics_lane_ordered) ? 1 : 0;
    if (const char* e = getenv("RANDT_DEBUG_FORCE_MISRANK")) ctx->debug_force_misrank = atoi(e) ? 1 : 0;
    if (const char* e = getenv("RANDT_WINDOW_GENERAL")) ctx->window_general = atoi(e) ? 1 : 0;
    // the word the build kernel counts its in-kernel ranking fallbacks in: pinned host memory, read without a synchronisation
    // in front of every build launch; without it the atomic ranking is not used at all
    void* pin = nullptr;
    if (hipHostMalloc(&pin, 64, hipHostMallocDefault) == hipSuccess && pin) {
      memset(pin, 0, 64);
      ctx->misrank_word = static_cast<int32_t*>(pin);
      if (hipMalloc(reinterpret_cast<void**>(&ctx->d_misrank_count), 64) == hipSuccess) (void)hipMemset(ctx->d_misrank_count, 0, 64);
      else {
        ctx->d_misrank_count = nullptr;
        (void)hipGetLastError();
      }
    } else {
      (void)hipGetLastError();
      ctx->lds_atomics_lane_ordered = 0;
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    g_ctxs.push_back(ctx);
  }
  *out = ctx;
  return RANDT_OK;
}

// debug / test hook (not part of the ABI): the geometry the context's last pair-solve launch took: split width W >= 2
// (several wavefronts per registration: the latency placement) or 0 (one wavefront per registration)
int randt_debug_last_solve_placement(const randt_ctx* ctx) { return ctx ? ctx->last_placement : -1; }

// debug / test hook (not part of the ABI): did the LDS atomic ordering self-test pass on this context's device?
int randt_debug_lds_atomics_lane_ordered(const randt_ctx* ctx) { return ctx ? ctx->lds_atomics_lane_ordered : 0; }
// debug / test hook: workgroups of this context's NDT builds that found the atomic ranking out of order and re-ranked with
// ballots (synchronises the stream)
int randt_debug_build_rank_fallbacks(randt_ctx* ctx) {
  if (!ctx) return 0;
  DeviceGuard dev_guard__(ctx);
  (void)randt_sync(ctx);
  int32_t n = 0;
  if (ctx->d_misrank_count && hipMemcpy(&n, ctx->d_misrank_count, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) n = 0;
  return n;
}

int randt_ctx_pool_stats(const randt_ctx* ctx, randt_pool_stats* out) {
  if (!ctx || !out) return RANDT_ERR_INVALID;
  *out = ctx->stats;
  out->pool_bytes = (int64_t)ctx->pool_bytes;
  out->pool_blocks = (int64_t)ctx->pool_free.size();
  return RANDT_OK;
}

int randt_ctx_pool_trim(randt_ctx* ctx) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx) return RANDT_ERR_INVALID;
  if (ctx->pool_free.empty()) return RANDT_OK;
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  for (auto& kv : ctx->pool_free) (void)randt_hip_free(ctx, kv.second);
  ctx->pool_free.clear();
  ctx->pool_bytes = 0;
  return RANDT_OK;
}

int randt_ctx_destroy(randt_ctx* ctx) {
  if (!ctx) return RANDT_OK;
  {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), ctx), g_ctxs.end());
  }
  DeviceGuard dev_guard__(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->tmp_cluster) (void)randt_maps_destroy(ctx->tmp_cluster);
  for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
  ctx->pool_free.clear();
  if (ctx->pin_ring) {
    (void)hipHostFree(ctx->pin_ring);
    for (int i = 0; i < randt_ctx::kPinSegs; ++i)
      if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
  }
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->order_ws) (void)hipFree(ctx->order_ws);
  if (ctx->build_ws) (void)hipFree(ctx->build_ws);
  if (ctx->build_wide_ws) (void)hipFree(ctx->build_wide_ws);
  if (ctx->small) (void)hipFree(ctx->small);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->marker_ev) (void)hipEventDestroy(ctx->marker_ev);
  if (ctx->misrank_word) (void)hipHostFree(ctx->misrank_word);
  if (ctx->d_misrank_count) (void)hipFree(ctx->d_misrank_count);
  delete ctx;
  return RANDT_OK;
}

// 4 KB that the synchronous host-level entries carve their pose / index / result words from (they synchronise before
// returning, so one block per context is enough and nothing is allocated per call).
static int small_block(randt_ctx* ctx, char** out) {
  if (!ctx->small) RANDT_HIP_CHECK(ctx, randt_hip_malloc(ctx, &ctx->small, 4096));
  *out = static_cast<char*>(ctx->small);
  return RANDT_OK;
}

int randt_ctx_set_stream(randt_ctx* ctx, void* stream) {
  if (!ctx) return RANDT_ERR_INVALID;
  if ((hipStream_t)stream != ctx->stream) {
    // parked blocks, the workspace and the pinned ring are ordered by the OLD stream: let it drain before another one reuses them
    DeviceGuard dev_guard__(ctx);
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));
    for (int i = 0; i < randt_ctx::kPinSegs; ++i) ctx->pin_pending[i] = false;
  }
  ctx->stream = (hipStream_t)stream;
  return RANDT_OK;
}

int randt_ctx_synchronize(randt_ctx* ctx) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx) return RANDT_ERR_INVALID;
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  return RANDT_OK;
}

int randt_ctx_set_trace(randt_ctx* ctx, double* d_trace, int max_len) {
  if (!ctx) return RANDT_ERR_INVALID;
  ctx->d_trace = d_trace;
  ctx->trace_len = d_trace ? max_len : 0;
  return RANDT_OK;
}

int randt_ctx_set_solve_mode(randt_ctx* ctx, int mode) {
  if (!ctx || (mode != RANDT_SOLVE_AUTO && mode != RANDT_SOLVE_THROUGHPUT && mode != RANDT_SOLVE_LATENCY)) return RANDT_ERR_INVALID;
  ctx->solve_mode = mode;
  return RANDT_OK;
}

size_t randt_maps_cells_bytes(int n_maps, int cell_capacity) { return (size_t)n_maps * cell_capacity * sizeof(randt_cell); }

size_t randt_maps_grid_bytes(int n_maps, const randt_map_params* p) {
  return p ? (size_t)n_maps * p->size_x * p->size_y * sizeof(int32_t) : 0;
}

int randt_maps_create_external(randt_ctx* ctx, int n_maps, const randt_map_params* p, int cell_capacity, void* d_cells,
                               void* d_counts, void* d_grid, randt_maps** out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !out || bad_params(p, n_maps, cell_capacity) || !d_cells || !d_counts) return RANDT_ERR_INVALID;
  if (((size_t)d_cells & 15) != 0) return randt_set_error(ctx, RANDT_ERR_INVALID, "cell storage must be 16-byte aligned", hipSuccess);
  randt_maps* m = new (std::nothrow) randt_maps();
  if (!m) return RANDT_ERR_NOMEM;
  m->ctx = ctx;
  fill_view(m, n_maps, p, cell_capacity);
  m->v.cells = (randt_cell*)d_cells;
  m->v.counts = (int32_t*)d_counts;
  m->v.grid = (int32_t*)d_grid;
  m->owns = false;
  *out = m;
  return RANDT_OK;
}

// Layout of a library-owned batch inside its ONE pooled block: [cells | counts | deferred status (2 words) | grid].
static void block_layout(int n_maps, const randt_map_params* p, int cell_capacity, int with_grid, size_t* off_counts, size_t* off_grid,
                         size_t* total) {
  const size_t cb = (randt_maps_cells_bytes(n_maps, cell_capacity) + 255) & ~(size_t)255;
  const size_t nb = (sizeof(int32_t) * ((size_t)n_maps + 2) + 255) & ~(size_t)255;
  *off_counts = cb;
  *off_grid = cb + nb;
  *total = cb + nb + (with_grid ? ((randt_maps_grid_bytes(n_maps, p) + 255) & ~(size_t)255) : 0);
}

// a batch on pooled storage, contents undefined (create clears it, clone copies into it)
static int maps_alloc(randt_ctx* ctx, int n_maps, const randt_map_params* p, int cell_capacity, int with_grid, randt_maps** out) {
  size_t off_counts, off_grid, total, granted = 0;
  block_layout(n_maps, p, cell_capacity, with_grid, &off_counts, &off_grid, &total);
  void* blk = nullptr;
  const hipError_t e = randt_dev_alloc(ctx, &blk, total, &granted);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // a failed hipMalloc leaves a sticky last-error; nothing else was allocated
    return randt_set_error(ctx, e == hipErrorOutOfMemory ? RANDT_ERR_NOMEM : RANDT_ERR_HIP, "hipMalloc (map storage)", e);
  }
  char* b = static_cast<char*>(blk);
  const int rc = randt_maps_create_external(ctx, n_maps, p, cell_capacity, b, b + off_counts, with_grid ? b + off_grid : nullptr, out);
  if (rc) {
    randt_dev_release(ctx, blk, granted);
    return rc;
  }
  (*out)->owns = true;  // from here on randt_maps_destroy returns the block
  (*out)->block = blk;
  (*out)->block_bytes = granted;
  return RANDT_OK;
}

// zeroes the cell records, the counts and the deferred-status words and sets the index grids to -1: one launch
__global__ __launch_bounds__(256) void k_maps_init(MapView v, size_t cell_words /* 16-byte words */) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4* c = reinterpret_cast<uint4*>(v.cells);
  for (size_t i = t0; i < cell_words; i += stride) c[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = t0; i < (size_t)v.n_maps + 2; i += stride) v.counts[i] = 0;
  if (v.grid) {
    const size_t n = (size_t)v.n_maps * v.n_slots;
    for (size_t i = t0; i < n; i += stride) v.grid[i] = -1;
  }
}

int randt_maps_create(randt_ctx* ctx, int n_maps, const randt_map_params* p, int cell_capacity, int with_grid,
                      randt_maps** out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !out || bad_params(p, n_maps, cell_capacity)) return RANDT_ERR_INVALID;
  *out = nullptr;
  int rc = maps_alloc(ctx, n_maps, p, cell_capacity, with_grid, out);
  if (rc) return rc;
  const size_t words = randt_maps_cells_bytes(n_maps, cell_capacity) / 16;
  size_t work = words > (size_t)n_maps * (*out)->v.n_slots ? words : (size_t)n_maps * (*out)->v.n_slots;
  int bx = (int)((work + 255) / 256);
  bx = bx > 1024 ? 1024 : (bx < 1 ? 1 : bx);
  hipLaunchKernelGGL(k_maps_init, dim3(bx), dim3(256), 0, ctx->stream, (*out)->v, words);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rc = randt_set_error(ctx, RANDT_ERR_HIP, "k_maps_init", e);
    (void)randt_maps_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

int randt_maps_destroy(randt_maps* m) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!m) return RANDT_OK;
  // no synchronisation: the block's next owner is served by the same stream (randt_internal.h, storage pool), behind a
  // device-side wait for the other contexts that have used the batch
  if (m->owns) {
    wait_for_foreign_users(m);
    randt_dev_release(m->ctx, m->block, m->block_bytes);
  }
  delete m;
  return RANDT_OK;
}

int randt_maps_info(const randt_maps* m, int* n_maps, int* cell_capacity, int* n_slots, int* with_grid) {
  if (!m) return RANDT_ERR_INVALID;
  if (n_maps) *n_maps = m->v.n_maps;
  if (cell_capacity) *cell_capacity = m->v.cap;
  if (n_slots) *n_slots = m->v.n_slots;
  if (with_grid) *with_grid = m->v.grid ? 1 : 0;
  return RANDT_OK;
}

int randt_maps_device_ptrs(const randt_maps* m, void** d_cells, void** d_counts, void** d_grid) {
  if (!m) return RANDT_ERR_INVALID;
  if (d_cells) *d_cells = m->v.cells;
  if (d_counts) *d_counts = m->v.counts;
  if (d_grid) *d_grid = m->v.grid;
  return RANDT_OK;
}

static int deferred_status(randt_maps* m, const int32_t d[2]);
static bool range_ok(const randt_maps* m, int first, int count) {
  return m && first >= 0 && count >= 0 && first + count <= m->v.n_maps;
}

int randt_maps_clear(randt_maps* m, int first, int count) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, first, count)) return RANDT_ERR_INVALID;
  if (count == 0) return RANDT_OK;
  randt_ctx* ctx = m->ctx;
  int bx = (m->v.n_slots + 255) / 256;
  bx = bx > 32 ? 32 : (bx < 1 ? 1 : bx);
  hipLaunchKernelGGL(k_clear, dim3(bx, count), dim3(256), 0, ctx->stream, m->v, first, count);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int randt_maps_upload(randt_maps* m, int idx, const randt_cell* h_cells, int n_cells, const int32_t* h_grid) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, idx, 1) || n_cells < 0 || n_cells > m->v.cap || (n_cells > 0 && !h_cells)) return RANDT_ERR_INVALID;
  randt_ctx* ctx = m->ctx;
  if (n_cells)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(m->v.cells + (size_t)idx * m->v.cap, h_cells, sizeof(randt_cell) * n_cells,
                                        hipMemcpyHostToDevice, ctx->stream));
  int32_t n = n_cells;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(m->v.counts + idx, &n, sizeof(n), hipMemcpyHostToDevice, ctx->stream));
  if (h_grid && m->v.grid)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(m->v.grid + (size_t)idx * m->v.n_slots, h_grid, sizeof(int32_t) * m->v.n_slots,
                                        hipMemcpyHostToDevice, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  return RANDT_OK;
}

int randt_maps_download(randt_maps* m, int idx, randt_cell* h_cells, int max_cells, int* n_cells, int32_t* h_grid) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, idx, 1)) return RANDT_ERR_INVALID;
  randt_ctx* ctx = m->ctx;
  int32_t n = 0;
  int32_t deferred[2] = {0, 0};
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(&n, m->v.counts + idx, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  if (m->deferred_pending)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(deferred, m->v.counts + m->v.n_maps, sizeof(deferred), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  // the outputs first: randt.h promises the deferred status of earlier asynchronous inserts "with the outputs valid" (what
  // WAS placed is in the batch and is what the caller gets); the status is reported once, behind the completed download
  if (n_cells) *n_cells = n;
  int c = n < max_cells ? n : max_cells;
  if (h_cells && c > 0)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_cells, m->v.cells + (size_t)idx * m->v.cap, sizeof(randt_cell) * c,
                                        hipMemcpyDeviceToHost, ctx->stream));
  if (h_grid && m->v.grid)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_grid, m->v.grid + (size_t)idx * m->v.n_slots, sizeof(int32_t) * m->v.n_slots,
                                        hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  if (m->deferred_pending) {
    m->deferred_pending = false;
    return deferred_status(m, deferred);
  }
  return RANDT_OK;
}

int randt_maps_counts(randt_maps* m, int first, int count, int32_t* h_counts) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, first, count) || !h_counts) return RANDT_ERR_INVALID;
  randt_ctx* ctx = m->ctx;
  if (count)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_counts, m->v.counts + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, ctx->stream));
  int32_t deferred[2] = {0, 0};
  if (m->deferred_pending)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(deferred, m->v.counts + m->v.n_maps, sizeof(deferred), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  if (m->deferred_pending) {
    m->deferred_pending = false;
    return deferred_status(m, deferred);
  }
  return RANDT_OK;
}

// cells (whole capacity: the tail behind `count` stays whatever the source holds -- zero for library-owned batches), counts
// and index grids of `count` maps: blockIdx.y = map
__global__ __launch_bounds__(256) void k_maps_copy(MapView dst, int dst_first, MapView src, int src_first, int zero_deferred) {
  const int i = blockIdx.y;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* sc = reinterpret_cast<const uint4*>(src.cells + (size_t)(src_first + i) * src.cap);
  uint4* dc = reinterpret_cast<uint4*>(dst.cells + (size_t)(dst_first + i) * dst.cap);
  const size_t words = (size_t)src.cap * (sizeof(randt_cell) / 16);
  for (size_t w = t0; w < words; w += stride) dc[w] = sc[w];
  if (dst.grid && src.grid) {
    const int32_t* sg = src.grid + (size_t)(src_first + i) * src.n_slots;
    int32_t* dg = dst.grid + (size_t)(dst_first + i) * dst.n_slots;
    for (size_t w = t0; w < (size_t)src.n_slots; w += stride) dg[w] = sg[w];
  }
  if (t0 == 0) dst.counts[dst_first + i] = src.counts[src_first + i];
  if (zero_deferred && t0 == 0 && i == 0) dst.counts[dst.n_maps] = dst.counts[dst.n_maps + 1] = 0;  // a fresh library-owned batch
}

static int launch_maps_copy(randt_ctx* ctx, const MapView& dst, int dst_first, const MapView& src, int src_first, int count,
                            int zero_deferred = 0) {
  const size_t words = (size_t)src.cap * (sizeof(randt_cell) / 16);
  const size_t work = words > (size_t)src.n_slots ? words : (size_t)src.n_slots;
  int bx = (int)((work + 255) / 256);
  bx = bx > 128 ? 128 : (bx < 1 ? 1 : bx);
  hipLaunchKernelGGL(k_maps_copy, dim3(bx, count), dim3(256), 0, ctx->stream, dst, dst_first, src, src_first, zero_deferred);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int randt_maps_copy(randt_maps* dst, int dst_first, const randt_maps* src, int src_first, int count) {
  DeviceGuard dev_guard__(dst ? dst->ctx : nullptr);
  if (!range_ok(dst, dst_first, count) || !range_ok(src, src_first, count)) return RANDT_ERR_INVALID;
  if (dst->v.n_slots != src->v.n_slots || dst->v.cap < src->v.cap) return RANDT_ERR_INVALID;
  randt_ctx* ctx = dst->ctx;
  // the copied index grids and cell means only mean the same thing in a batch of the same geometry
  if (dst->p.size_x != src->p.size_x || dst->p.size_y != src->p.size_y || dst->p.resolution != src->p.resolution ||
      dst->p.center_x != src->p.center_x || dst->p.center_y != src->p.center_y)
    return randt_set_error(ctx, RANDT_ERR_INVALID, "randt_maps_copy: the two batches differ in map geometry (size, resolution or centre)", hipSuccess);
  if (count == 0) return RANDT_OK;
  randt_note_user(ctx, src);
  if (src->ctx->device != ctx->device) {
    // k_maps_copy dereferences both batches from dst's device: across GPUs that needs peer access, which nobody has promised
    // here -- the runtime's peer copy does not.  Whole capacity per map, like the kernel.
    for (int i = 0; i < count; ++i) {
      RANDT_HIP_CHECK(ctx, hipMemcpyPeerAsync(dst->v.cells + (size_t)(dst_first + i) * dst->v.cap, ctx->device,
                                              src->v.cells + (size_t)(src_first + i) * src->v.cap, src->ctx->device,
                                              sizeof(randt_cell) * (size_t)src->v.cap, ctx->stream));
      if (dst->v.grid && src->v.grid)
        RANDT_HIP_CHECK(ctx, hipMemcpyPeerAsync(dst->v.grid + (size_t)(dst_first + i) * dst->v.n_slots, ctx->device,
                                                src->v.grid + (size_t)(src_first + i) * src->v.n_slots, src->ctx->device,
                                                sizeof(int32_t) * (size_t)src->v.n_slots, ctx->stream));
    }
    RANDT_HIP_CHECK(ctx, hipMemcpyPeerAsync(dst->v.counts + dst_first, ctx->device, src->v.counts + src_first, src->ctx->device,
                                            sizeof(int32_t) * (size_t)count, ctx->stream));
    return RANDT_OK;
  }
  return launch_maps_copy(ctx, dst->v, dst_first, src->v, src_first, count);
}

int randt_maps_clone(const randt_maps* src, int first, int count, randt_maps** out) {
  DeviceGuard dev_guard__(src ? src->ctx : nullptr);
  if (!out || !range_ok(src, first, count) || count < 1) return RANDT_ERR_INVALID;
  *out = nullptr;
  randt_ctx* ctx = src->ctx;
  int rc = maps_alloc(ctx, count, &src->p, src->v.cap, src->v.grid ? 1 : 0, out);
  if (rc) return rc;
  rc = launch_maps_copy(ctx, (*out)->v, 0, src->v, first, count, 1);  // + the two deferred-status words behind the counts start at zero
  if (rc) {
    (void)randt_maps_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

int randt_ndt_build_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                              const int32_t* d_n_points, int stride_floats, int intensity_index,
                              const randt_cluster_params* cp, randt_maps* out, int first_map) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !cp || !range_ok(out, first_map, n_scans < 0 ? 0 : n_scans) || n_scans < 0 || pitch_points < 0 ||
      stride_floats < 3 || intensity_index < 0 || intensity_index >= stride_floats || cp->n_clusters <= 0 ||
      !(cp->max_range > 0.f))
    return RANDT_ERR_INVALID;
  if (n_scans == 0) return RANDT_OK;
  if (!d_points && pitch_points > 0) return RANDT_ERR_INVALID;
  if (pitch_points == 0) return randt_maps_clear(out, first_map, n_scans);
  randt_note_user(ctx, out);
  return launch_ndt_build(ctx, d_points, n_scans, pitch_points, d_n_points, stride_floats, intensity_index, cp, out->v, first_map);
}

int randt_ndt_build_pndt_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                                   const int32_t* d_n_points, int stride_floats, int intensity_index, const float* d_polar,
                                   const float* beam_cov9, const randt_cluster_params* cp, randt_maps* out, int first_map) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !cp || !range_ok(out, first_map, n_scans < 0 ? 0 : n_scans) || n_scans < 0 || pitch_points < 0 ||
      stride_floats < 3 || intensity_index < 0 || intensity_index >= stride_floats || cp->n_clusters <= 0 ||
      !(cp->max_range > 0.f) || !beam_cov9)
    return RANDT_ERR_INVALID;
  for (int i = 0; i < 9; ++i)
    if (!isfinite(beam_cov9[i])) return randt_set_error(ctx, RANDT_ERR_INVALID, "beam_cov must be finite", hipSuccess);
  if (n_scans == 0) return RANDT_OK;
  if ((!d_points || !d_polar) && pitch_points > 0) return RANDT_ERR_INVALID;
  if (pitch_points == 0) return randt_maps_clear(out, first_map, n_scans);
  randt_note_user(ctx, out);
  return launch_ndt_build(ctx, d_points, n_scans, pitch_points, d_n_points, stride_floats, intensity_index, cp, out->v, first_map,
                          d_polar, beam_cov9);
}

// host points -> the context's workspace without a synchronisation: through the pinned ring when they fit a segment (one
// memcpy + one async DMA; the host buffer is free on return), a pageable copy + wait otherwise
static int stage_points(randt_ctx* ctx, const float* h_points, size_t bytes, size_t ws_extra, const float** d_points) {
  int rc = ensure_ws(ctx, bytes + ws_extra);
  if (rc) return rc;
  if (void* pin = randt_pin_take(ctx, bytes)) {
    memcpy(pin, h_points, bytes);
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws, pin, bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws, h_points, bytes, hipMemcpyHostToDevice, ctx->stream));
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));  // a pageable source of this size may still be read after the call returns
  }
  *d_points = static_cast<const float*>(ctx->ws);
  return RANDT_OK;
}

int randt_ndt_build(randt_ctx* ctx, const float* h_points, int n_points, int stride_floats, int intensity_index,
                    const randt_cluster_params* cp, randt_maps* out, int map_idx) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || n_points < 0 || (n_points > 0 && !h_points) || stride_floats < 3) return RANDT_ERR_INVALID;
  if (n_points == 0) return randt_maps_clear(out, map_idx, 1);
  const float* d_points = nullptr;
  int rc = stage_points(ctx, h_points, sizeof(float) * (size_t)n_points * stride_floats, 0, &d_points);
  if (rc) return rc;
  return randt_ndt_build_batch_dev(ctx, d_points, 1, n_points, nullptr, stride_floats, intensity_index, cp, out, map_idx);
}

// ---------------------------------------------------------------- single-cell / single-cluster map edits --------
// The reference's Map can also be edited cell by cell (insertCluster, insertCell) and queried for one cell
// (getClosestCells); these host-level entries keep those calls available.  They are convenience paths (a few tiny
// launches and a synchronisation each), not the batched hot path.
static int append_from(randt_maps* m, int idx, const randt_maps* src, int set_grid, int* n_dropped, int* n_outside) {
  randt_ctx* ctx = m->ctx;
  char* d_blk = nullptr;
  int rc = small_block(ctx, &d_blk);
  if (rc) return rc;
  int32_t* d_status = reinterpret_cast<int32_t*>(d_blk + 512);
  rc = launch_maps_append(ctx, m->v, idx, src->v, 0, set_grid, d_status, 0);
  int32_t h_status[2] = {0, 0};
  if (!rc) {
    hipError_t e = hipMemcpyAsync(h_status, d_status, sizeof(h_status), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "append status read-back", e);
  } else {
    (void)randt_sync(ctx);
  }
  if (n_dropped) *n_dropped = h_status[0];
  if (n_outside) *n_outside = h_status[1];
  return rc;
}

// What asynchronous inserts (randt_maps_insert_cluster with accepted = NULL) could not report when they ran: the append
// kernel ADDS its dropped / outside counts to two words behind the batch's counts; the next synchronising read of the batch
// (randt_maps_counts / randt_maps_download) fetches them in the same round trip, reports once and clears them.
static int deferred_status(randt_maps* m, const int32_t d[2]) {
  if (!d[0] && !d[1]) return RANDT_OK;
  (void)hipMemsetAsync(m->v.counts + m->v.n_maps, 0, 2 * sizeof(int32_t), m->ctx->stream);
  if (d[1]) return randt_set_error(m->ctx, RANDT_ERR_INVALID, "an earlier asynchronous insert: cluster mean outside the map's index grid", hipSuccess);
  return randt_set_error(m->ctx, RANDT_ERR_UNSUPPORTED, "an earlier asynchronous insert: map cell capacity exhausted", hipSuccess);
}

int randt_maps_insert_cells(randt_maps* m, int idx, const randt_cell* h_cells, int n_cells, int set_grid) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, idx, 1) || n_cells < 0 || (n_cells > 0 && !h_cells)) return RANDT_ERR_INVALID;
  if (n_cells == 0) return RANDT_OK;
  randt_ctx* ctx = m->ctx;
  randt_maps* tmp = nullptr;
  int rc = maps_alloc(ctx, 1, &m->p, n_cells, 0, &tmp);  // pooled, fully overwritten by the upload
  if (rc) return rc;
  rc = randt_maps_upload(tmp, 0, h_cells, n_cells, nullptr);
  int dropped = 0, outside = 0;
  if (!rc) rc = append_from(m, idx, tmp, set_grid, &dropped, &outside);
  (void)randt_maps_destroy(tmp);
  if (rc) return rc;
  if (outside) return randt_set_error(ctx, RANDT_ERR_INVALID, "cell mean outside the map's index grid", hipSuccess);
  if (dropped) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "map cell capacity exhausted", hipSuccess);
  return RANDT_OK;
}

int randt_maps_insert_cluster(randt_maps* m, int idx, const float* h_points, int n_points, int stride_floats, int intensity_index,
                              int* accepted) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, idx, 1) || n_points < 0 || (n_points > 0 && !h_points)) return RANDT_ERR_INVALID;
  if (accepted) *accepted = 0;
  if (n_points == 0) return RANDT_OK;
  randt_ctx* ctx = m->ctx;
  // the one-cell scratch map lives as long as the context; only its geometry follows the target's
  if (!ctx->tmp_cluster) {
    int rc = randt_maps_create(ctx, 1, &m->p, 4, 0, &ctx->tmp_cluster);
    if (rc) return rc;
  }
  randt_maps* tmp = ctx->tmp_cluster;
  {
    void *c = tmp->v.cells, *n = tmp->v.counts;
    fill_view(tmp, 1, &m->p, 4);
    tmp->v.cells = static_cast<randt_cell*>(c);
    tmp->v.counts = static_cast<int32_t*>(n);
    tmp->v.grid = nullptr;
  }
  // one voxel that swallows every point: row = 1, res = 2 * max_range (grid.cpp:8-13) -> label 0 for |x|, |y| < res
  randt_cluster_params one;
  one.n_clusters = 1;
  one.max_range = 1.0e9f;
  int rc = randt_ndt_build(ctx, h_points, n_points, stride_floats, intensity_index, &one, tmp, 0);
  if (rc) return rc;
  if (!accepted && m->owns) {
    // asynchronous: nothing comes back to the host; an unplaceable cell is reported by the next synchronising read
    m->deferred_pending = true;
    return launch_maps_append(ctx, m->v, idx, tmp->v, 0, 1, m->v.counts + m->v.n_maps, 1);
  }
  int32_t cnt = 0;
  rc = randt_maps_counts(tmp, 0, 1, &cnt);
  int dropped = 0, outside = 0;
  if (!rc && cnt > 0) rc = append_from(m, idx, tmp, 1, &dropped, &outside);
  if (rc) return rc;
  if (outside) return randt_set_error(ctx, RANDT_ERR_INVALID, "cluster mean outside the map's index grid", hipSuccess);
  if (dropped) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "map cell capacity exhausted", hipSuccess);
  if (accepted) *accepted = cnt > 0 ? 1 : 0;
  return RANDT_OK;
}

int randt_maps_insert_clusters(randt_maps* m, int idx, const float* h_points, const int32_t* h_offsets, int n_clusters, int stride_floats,
                               int intensity_index, int* n_accepted) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, idx, 1) || n_clusters < 0 || stride_floats < 3 || intensity_index < 0 || intensity_index >= stride_floats) return RANDT_ERR_INVALID;
  if (n_accepted) *n_accepted = 0;
  if (n_clusters == 0) return RANDT_OK;
  if (!h_offsets || h_offsets[0] < 0) return RANDT_ERR_INVALID;
  for (int c = 0; c < n_clusters; ++c)
    if (h_offsets[c + 1] < h_offsets[c]) return randt_set_error(m->ctx, RANDT_ERR_INVALID, "cluster offsets must be non-decreasing", hipSuccess);
  const int n_points = h_offsets[n_clusters];
  if (n_points > 0 && !h_points) return RANDT_ERR_INVALID;
  randt_ctx* ctx = m->ctx;
  // points | offsets | status (2) | accepted (1) in the workspace: through the pinned ring when they fit a segment
  const size_t pb = (sizeof(float) * (size_t)n_points * stride_floats + 255) & ~(size_t)255, ob = (sizeof(int32_t) * ((size_t)n_clusters + 1) + 255) & ~(size_t)255;
  int rc = ensure_ws(ctx, pb + ob + 256);
  if (rc) return rc;
  char* ws = static_cast<char*>(ctx->ws);
  bool waited = false;
  if (char* pin = static_cast<char*>(randt_pin_take(ctx, pb + ob))) {
    if (n_points) memcpy(pin, h_points, sizeof(float) * (size_t)n_points * stride_floats);
    memcpy(pin + pb, h_offsets, sizeof(int32_t) * ((size_t)n_clusters + 1));
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws, pin, pb + ob, hipMemcpyHostToDevice, ctx->stream));
  } else {
    if (n_points) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws, h_points, sizeof(float) * (size_t)n_points * stride_floats, hipMemcpyHostToDevice, ctx->stream));
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + pb, h_offsets, sizeof(int32_t) * ((size_t)n_clusters + 1), hipMemcpyHostToDevice, ctx->stream));
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));  // pageable sources
    waited = true;
  }
  (void)waited;
  int32_t* d_tail = reinterpret_cast<int32_t*>(ws + pb + ob);
  const bool deferred = !n_accepted && m->owns;  // nothing comes back: the unplaceable clusters are reported by the next synchronising read
  rc = launch_maps_insert_clusters(ctx, m->v, idx, reinterpret_cast<const float*>(ws), reinterpret_cast<const int32_t*>(ws + pb), n_clusters, stride_floats,
                                   intensity_index, deferred ? m->v.counts + m->v.n_maps : d_tail, deferred ? 1 : 0, d_tail + 2);
  if (rc) return rc;
  if (deferred) {
    m->deferred_pending = true;
    return RANDT_OK;
  }
  int32_t h_tail[3] = {0, 0, 0};
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_tail, d_tail, sizeof(h_tail), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  if (n_accepted) *n_accepted = h_tail[2];
  if (h_tail[1]) return randt_set_error(ctx, RANDT_ERR_INVALID, "cluster mean outside the map's index grid", hipSuccess);
  if (h_tail[0]) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "map cell capacity exhausted", hipSuccess);
  return RANDT_OK;
}

int randt_closest_cells(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_cell* h_queries, int n_queries, int k,
                        int lookup_mahalanobis, int use_intensity, int32_t* h_out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !range_ok(fixed, fixed_idx, 1) || n_queries < 0 || k <= 0) return RANDT_ERR_INVALID;
  if (n_queries == 0) return RANDT_OK;
  if (!h_queries || !h_out) return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_maps* tmp = nullptr;
  int rc = maps_alloc(ctx, 1, &fixed->p, n_queries, 0, &tmp);  // pooled, fully overwritten by the upload
  if (rc) return rc;
  rc = randt_maps_upload(tmp, 0, h_queries, n_queries, nullptr);
  char* d_blk = nullptr;
  size_t blk_bytes = 0;
  const size_t corr_bytes = sizeof(int32_t) * (size_t)n_queries * k;
  if (!rc && randt_dev_alloc(ctx, reinterpret_cast<void**>(&d_blk), 256 + corr_bytes, &blk_bytes) != hipSuccess) {
    (void)hipGetLastError();
    d_blk = nullptr;
    rc = randt_set_error(ctx, RANDT_ERR_NOMEM, "hipMalloc", hipErrorOutOfMemory);
  }
  if (!rc) {
    const double ident[4] = {1.0, 0.0, 0.0, 0.0};
    double* d_pose = reinterpret_cast<double*>(d_blk);
    int32_t* d_fi = reinterpret_cast<int32_t*>(d_blk + 64);
    int32_t* d_corr = reinterpret_cast<int32_t*>(d_blk + 256);
    hipError_t e = hipMemcpyAsync(d_pose, ident, sizeof(ident), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_fi, &fixed_idx, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "hipMemcpyAsync", e);
    if (!rc) rc = launch_associate(ctx, fixed->v, d_fi, tmp->v, 0, 1, d_pose, k, lookup_mahalanobis, use_intensity, d_corr);
    if (!rc) {
      e = hipMemcpyAsync(h_out, d_corr, corr_bytes, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = randt_sync(ctx);
      if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "closest cells read-back", e);
    } else {
      (void)randt_sync(ctx);
    }
  }
  if (d_blk) randt_dev_release(ctx, d_blk, blk_bytes);
  (void)randt_maps_destroy(tmp);
  return rc;
}

// ---------------------------------------------------------------- single cells (facade Cell mutators) ---------
namespace {
// stage n cells (and optionally n more) in the workspace, run one cell kernel, read back what it produced
int cells_roundtrip(randt_ctx* ctx, int op, randt_cell* h_a, const randt_cell* h_b, int n, const double* h_pose4, double* h_out) {
  const size_t cb = sizeof(randt_cell) * (size_t)n;
  const size_t off_b = (cb + 255) & ~(size_t)255, off_pose = off_b + ((cb + 255) & ~(size_t)255), off_out = off_pose + 256;
  int rc = ensure_ws(ctx, off_out + sizeof(double) * (size_t)n + 256);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws, h_a, cb, hipMemcpyHostToDevice, ctx->stream));
  if (h_b) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + off_b, h_b, cb, hipMemcpyHostToDevice, ctx->stream));
  if (h_pose4) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + off_pose, h_pose4, sizeof(double) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_cells_op(ctx, op, (randt_cell*)ws, (const randt_cell*)(ws + off_b), n, (const double*)(ws + off_pose), (double*)(ws + off_out));
  if (rc) return rc;
  if (h_out) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_out, ws + off_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  else RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_a, ws, cb, hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  return RANDT_OK;
}
}  // namespace

int randt_cells_merge(randt_ctx* ctx, randt_cell* h_acc, const randt_cell* h_other, int n) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || n < 0 || (n > 0 && (!h_acc || !h_other))) return RANDT_ERR_INVALID;
  if (n == 0) return RANDT_OK;
  return cells_roundtrip(ctx, 0, h_acc, h_other, n, nullptr, nullptr);
}

int randt_cells_transform(randt_ctx* ctx, randt_cell* h_cells, int n, const double h_pose4[4]) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || n < 0 || !h_pose4 || (n > 0 && !h_cells)) return RANDT_ERR_INVALID;
  if (n == 0) return RANDT_OK;
  return cells_roundtrip(ctx, 1, h_cells, nullptr, n, h_pose4, nullptr);
}

int randt_points_transform(randt_ctx* ctx, float* h_points, int n_points, int stride_floats, const double h_pose4[4]) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || n_points < 0 || stride_floats < 3 || !h_pose4 || (n_points > 0 && !h_points)) return RANDT_ERR_INVALID;
  if (n_points == 0) return RANDT_OK;
  const size_t pb = sizeof(float) * (size_t)n_points * stride_floats, off_pose = (pb + 255) & ~(size_t)255;
  int rc = ensure_ws(ctx, off_pose + 64);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws, h_points, pb, hipMemcpyHostToDevice, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + off_pose, h_pose4, sizeof(double) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_points_transform(ctx, (float*)ws, n_points, stride_floats, (const double*)(ws + off_pose));
  if (rc) return rc;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_points, ws, pb, hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  return RANDT_OK;
}

int randt_cells_mahalanobis(randt_ctx* ctx, const randt_cell* h_self, const randt_cell* h_subtrahend, int n, int use_intensity,
                            double* h_out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || n < 0 || (n > 0 && (!h_self || !h_subtrahend || !h_out))) return RANDT_ERR_INVALID;
  if (n == 0) return RANDT_OK;
  return cells_roundtrip(ctx, use_intensity ? 2 : 3, const_cast<randt_cell*>(h_self), h_subtrahend, n, nullptr, h_out);
}

int randt_cell_add_points(randt_ctx* ctx, randt_cell* h_cell, const float* h_points, int n_points, int stride_floats,
                          int intensity_index, int min_points_per_cell, int* accepted) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !h_cell || n_points < 0 || (n_points > 0 && !h_points) || stride_floats < 3 || intensity_index < 0 ||
      intensity_index >= stride_floats)
    return RANDT_ERR_INVALID;
  if (accepted) *accepted = 0;
  if (n_points == 0) return RANDT_OK;
  const size_t pb = sizeof(float) * (size_t)n_points * stride_floats;
  const size_t off_cell = (pb + 255) & ~(size_t)255, off_acc = off_cell + 256;
  int rc = ensure_ws(ctx, off_acc + 64);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws, h_points, pb, hipMemcpyHostToDevice, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + off_cell, h_cell, sizeof(randt_cell), hipMemcpyHostToDevice, ctx->stream));
  rc = launch_cell_update(ctx, (randt_cell*)(ws + off_cell), (const float*)ws, n_points, stride_floats, intensity_index, min_points_per_cell,
                          (int32_t*)(ws + off_acc));
  if (rc) return rc;
  int32_t acc = 0;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_cell, ws + off_cell, sizeof(randt_cell), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(&acc, ws + off_acc, sizeof(acc), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  if (accepted) *accepted = acc;
  return RANDT_OK;
}

int randt_maps_reindex(randt_maps* m, int first, int count) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, first, count)) return RANDT_ERR_INVALID;
  if (!m->v.grid) return randt_set_error(m->ctx, RANDT_ERR_INVALID, "maps batch has no index grid", hipSuccess);
  return launch_maps_reindex(m->ctx, m->v, first, count);
}

// host poses for a kernel of this stream: the pinned ring (read by the device in place, nothing to wait for), else the
// workspace + a synchronisation
static int stage_poses(randt_ctx* ctx, const double* h_pose4, int count, const double** d_pose4, bool* must_sync) {
  const size_t bytes = sizeof(double) * 4 * (size_t)count;
  *must_sync = false;
  if (void* pin = randt_pin_take(ctx, bytes)) {
    memcpy(pin, h_pose4, bytes);
    *d_pose4 = static_cast<const double*>(pin);
    return RANDT_OK;
  }
  int rc = ensure_ws(ctx, bytes);
  if (rc) return rc;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws, h_pose4, bytes, hipMemcpyHostToDevice, ctx->stream));
  *d_pose4 = static_cast<const double*>(ctx->ws);
  *must_sync = true;  // ws is reused by the next call
  return RANDT_OK;
}

int randt_maps_transform(randt_maps* m, int first, int count, const double* h_pose4) {
  DeviceGuard dev_guard__(m ? m->ctx : nullptr);
  if (!range_ok(m, first, count) || (count > 0 && !h_pose4)) return RANDT_ERR_INVALID;
  if (count == 0) return RANDT_OK;
  randt_ctx* ctx = m->ctx;
  const double* d_pose4 = nullptr;
  bool must_sync = false;
  int rc = stage_poses(ctx, h_pose4, count, &d_pose4, &must_sync);
  if (rc) return rc;
  rc = launch_maps_transform(ctx, m->v, first, count, d_pose4);
  if (rc) return rc;
  return must_sync ? randt_ctx_synchronize(ctx) : RANDT_OK;
}

int randt_maps_merge(randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_first, int n_moving,
                     const double* h_pose4) {
  DeviceGuard dev_guard__(fixed ? fixed->ctx : nullptr);
  if (!range_ok(fixed, fixed_idx, 1) || !range_ok(moving, moving_first, n_moving) || (n_moving > 0 && !h_pose4) || !fixed->v.grid)
    return RANDT_ERR_INVALID;
  if (n_moving == 0) return RANDT_OK;
  randt_ctx* ctx = fixed->ctx;
  if (moving->ctx->device != ctx->device)
    return randt_set_error(ctx, RANDT_ERR_INVALID, "randt_maps_merge: the two batches live on different devices (copy the moving maps over first)", hipSuccess);
  randt_note_user(ctx, moving);
  const double* d_pose4 = nullptr;
  bool must_sync = false;
  int rc = stage_poses(ctx, h_pose4, n_moving, &d_pose4, &must_sync);
  if (rc) return rc;
  rc = launch_maps_merge(ctx, fixed->v, fixed_idx, moving->v, moving_first, n_moving, d_pose4);
  if (rc) return rc;
  return must_sync ? randt_ctx_synchronize(ctx) : RANDT_OK;
}

int randt_maps_merge_batch(randt_maps* fixed, int fixed_first, int n_fixed, const randt_maps* moving, int moving_first, int n_moving_each,
                           const double* h_pose4) {
  DeviceGuard dev_guard__(fixed ? fixed->ctx : nullptr);
  if (!range_ok(fixed, fixed_first, n_fixed) || n_moving_each < 0 || !range_ok(moving, moving_first, n_fixed * (n_moving_each > 0 ? n_moving_each : 0)) ||
      (n_fixed > 0 && n_moving_each > 0 && !h_pose4) || !fixed->v.grid)
    return RANDT_ERR_INVALID;
  if (n_fixed == 0 || n_moving_each == 0) return RANDT_OK;
  randt_ctx* ctx = fixed->ctx;
  if (moving->ctx->device != ctx->device)
    return randt_set_error(ctx, RANDT_ERR_INVALID, "randt_maps_merge_batch: the two batches live on different devices", hipSuccess);
  randt_note_user(ctx, moving);
  const double* d_pose4 = nullptr;
  bool must_sync = false;
  int rc = stage_poses(ctx, h_pose4, n_fixed * n_moving_each, &d_pose4, &must_sync);
  if (rc) return rc;
  rc = launch_maps_merge(ctx, fixed->v, fixed_first, moving->v, moving_first, n_moving_each, d_pose4, n_fixed);
  if (rc) return rc;
  return must_sync ? randt_ctx_synchronize(ctx) : RANDT_OK;
}

// The solve kernels run the reference's GNC / trust-region loops ON THE DEVICE (`do { ... mu /= divisor } while (mu > 1 /
// sqrt(divisor))`, ndt_matcher.cpp:382-397,466-483): a divisor <= 1 or a non-finite scale that merely hangs a CPU thread
// in the reference would hang a GPU queue here, so the ABI rejects such parameter sets before anything is enqueued.
static int check_matcher_params(randt_ctx* ctx, const randt_matcher_params* mp) {
  const char* bad = nullptr;
  auto pos = [](double v) { return isfinite(v) && v > 0.0; };
  if (mp->n_neighbours <= 0 || mp->n_neighbours > 64) bad = "n_neighbours must be in 1..64 (1..16 where the library associates itself)";
  else if (!(isfinite(mp->gnc_divisor) && mp->gnc_divisor > 1.0)) bad = "gnc_divisor must be finite and > 1 (the GNC loop divides mu by it until mu <= 1/sqrt(divisor))";
  else if (mp->gnc_steps < 1 || mp->gnc_steps > 64) bad = "gnc_steps must be in 1..64";
  else if (mp->max_iterations < 0) bad = "max_iterations must be >= 0";
  else if (mp->max_consecutive_invalid_steps < 1) bad = "max_consecutive_invalid_steps must be >= 1";
  else if (!pos(mp->loss_scale)) bad = "loss_scale must be finite and > 0";
  else if (!pos(mp->mu_scale)) bad = "mu_scale must be finite and > 0";
  else if (!pos(mp->loss_weight)) bad = "loss_weight must be finite and > 0";
  else if (!isfinite(mp->loss_alpha)) bad = "loss_alpha must be finite";
  else if (!isfinite(mp->function_tolerance) || !isfinite(mp->gradient_tolerance) || !isfinite(mp->parameter_tolerance) ||
           mp->function_tolerance < 0.0 || mp->gradient_tolerance < 0.0 || mp->parameter_tolerance < 0.0)
    bad = "tolerances must be finite and >= 0";
  else if (!pos(mp->initial_radius) || !pos(mp->max_radius) || !(isfinite(mp->min_radius) && mp->min_radius >= 0.0))
    bad = "trust-region radii must be finite and positive";
  else if (!(isfinite(mp->min_lm_diagonal) && mp->min_lm_diagonal >= 0.0) || !pos(mp->max_lm_diagonal) || mp->min_lm_diagonal > mp->max_lm_diagonal)
    bad = "LM diagonal bounds";
  else if (!isfinite(mp->min_relative_decrease)) bad = "min_relative_decrease";
  else if (mp->parameterization != RANDT_PARAM_MANIFOLD && mp->parameterization != RANDT_PARAM_AMBIENT4 && mp->parameterization != RANDT_PARAM_VECTOR &&
           mp->parameterization != RANDT_PARAM_ANALYTIC)
    bad = "unknown parameterization";
  if (bad) return randt_set_error(ctx, RANDT_ERR_INVALID, bad, hipSuccess);
  return RANDT_OK;
}

static int check_pairs(randt_ctx* ctx, const randt_maps* fixed, const randt_maps* moving, int moving_first, int n_pairs,
                       const randt_matcher_params* mp) {
  if (!ctx || !fixed || !mp || n_pairs < 0 || !range_ok(moving, moving_first, n_pairs)) return RANDT_ERR_INVALID;
  return check_matcher_params(ctx, mp);
}

int randt_associate_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx,
                              const randt_maps* moving, int moving_first, int n_pairs, const double* d_guess4,
                              const randt_matcher_params* mp, int32_t* d_corr) {
  DeviceGuard dev_guard__(ctx);
  // only the association's own fields are looked at (n_neighbours, lookup_mahalanobis, use_intensity): a caller that fills
  // nothing else must not be refused for solver parameters this entry never reads
  if (!ctx || !fixed || !mp || n_pairs < 0 || !range_ok(moving, moving_first, n_pairs)) return RANDT_ERR_INVALID;
  if (mp->n_neighbours < 1) return randt_set_error(ctx, RANDT_ERR_INVALID, "n_neighbours must be >= 1", hipSuccess);
  if (mp->n_neighbours > 16)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "n_neighbours > 16 not supported by the association kernel (it keeps at most sixteen candidates per cell)", hipSuccess);
  if (n_pairs == 0) return RANDT_OK;
  if (!d_guess4 || !d_corr) return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  return launch_associate(ctx, fixed->v, d_fixed_idx, moving->v, moving_first, n_pairs, d_guess4, mp->n_neighbours,
                          mp->lookup_mahalanobis, mp->use_intensity, d_corr);
}

int randt_solve_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx, const randt_maps* moving,
                          int moving_first, int n_pairs, const int32_t* d_corr, const randt_matcher_params* mp,
                          double* d_pose4, randt_result* d_results) {
  DeviceGuard dev_guard__(ctx);
  int rc = check_pairs(ctx, fixed, moving, moving_first, n_pairs, mp);
  if (rc) return rc;
  if (n_pairs == 0) return RANDT_OK;
  if (!d_corr || !d_pose4 || !d_results) return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  return launch_solve(ctx, fixed->v, d_fixed_idx, moving->v, moving_first, n_pairs, d_corr, mp, d_pose4, d_results);
}

int randt_register_batch_dev(randt_ctx* ctx, const randt_maps* fixed, const int32_t* d_fixed_idx,
                             const randt_maps* moving, int moving_first, int n_pairs, const randt_matcher_params* mp,
                             double* d_pose4, randt_result* d_results) {
  DeviceGuard dev_guard__(ctx);
  int rc = check_pairs(ctx, fixed, moving, moving_first, n_pairs, mp);
  if (rc) return rc;
  if (n_pairs == 0) return RANDT_OK;
  if (!d_pose4 || !d_results) return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  size_t corr_bytes = sizeof(int32_t) * (size_t)n_pairs * moving->v.cap * mp->n_neighbours;
  rc = ensure_ws(ctx, corr_bytes);
  if (rc) return rc;
  int32_t* d_corr = (int32_t*)ctx->ws;
  rc = launch_associate(ctx, fixed->v, d_fixed_idx, moving->v, moving_first, n_pairs, d_pose4, mp->n_neighbours,
                        mp->lookup_mahalanobis, mp->use_intensity, d_corr);
  if (rc) return rc;
  return launch_solve(ctx, fixed->v, d_fixed_idx, moving->v, moving_first, n_pairs, d_corr, mp, d_pose4, d_results);
}

int randt_scan_register_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int pitch_points,
                                  const int32_t* d_n_points, int stride_floats, int intensity_index,
                                  const randt_cluster_params* cp, const randt_maps* fixed, const int32_t* d_fixed_idx,
                                  randt_maps* scan_maps, const randt_matcher_params* mp, double* d_pose4,
                                  randt_result* d_results) {
  DeviceGuard dev_guard__(ctx);
  int rc = randt_ndt_build_batch_dev(ctx, d_points, n_scans, pitch_points, d_n_points, stride_floats, intensity_index, cp,
                                     scan_maps, 0);
  if (rc) return rc;
  return randt_register_batch_dev(ctx, fixed, d_fixed_idx, scan_maps, 0, n_scans, mp, d_pose4, d_results);
}

int randt_register_pair(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                        const randt_matcher_params* mp, double h_pose4[4], randt_result* h_result) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !h_pose4 || !range_ok(fixed, fixed_idx, 1) || !range_ok(moving, moving_idx, 1) || !mp) return RANDT_ERR_INVALID;
  // small staging block: [pose4 | result | fixed_idx]
  char* stage = nullptr;
  int rc0 = small_block(ctx, &stage);
  if (rc0) return rc0;
  stage += 1024;  // own region of the context's scratch block
  double* d_pose = (double*)stage;
  randt_result* d_res = (randt_result*)(d_pose + 4);
  int32_t* d_fi = (int32_t*)(d_res + 1);
  int32_t fi = fixed_idx;
  hipError_t e1 = hipMemcpyAsync(d_pose, h_pose4, sizeof(double) * 4, hipMemcpyHostToDevice, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(d_fi, &fi, sizeof(fi), hipMemcpyHostToDevice, ctx->stream);
  int rc = (e1 != hipSuccess || e2 != hipSuccess) ? randt_set_error(ctx, RANDT_ERR_HIP, "hipMemcpyAsync", e1 != hipSuccess ? e1 : e2) : RANDT_OK;
  if (!rc) rc = randt_register_batch_dev(ctx, fixed, d_fi, moving, moving_idx, 1, mp, d_pose, d_res);
  randt_result r;
  memset(&r, 0, sizeof(r));
  if (!rc) {
    hipError_t e3 = hipMemcpyAsync(h_pose4, d_pose, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e4 = hipMemcpyAsync(&r, d_res, sizeof(r), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e5 = randt_sync(ctx);
    if (e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess)
      rc = randt_set_error(ctx, RANDT_ERR_HIP, "download", e5 != hipSuccess ? e5 : (e3 != hipSuccess ? e3 : e4));
  } else {
    (void)randt_sync(ctx);
  }
  if (h_result) *h_result = r;
  return rc;
}


int randt_eval_cost_batch_dev(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                              const int32_t* d_corr, const randt_matcher_params* mp, double scale, const double* d_poses4,
                              int n_poses, double* d_cost, int32_t* d_n_res) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !mp || !range_ok(fixed, fixed_idx, 1) || !range_ok(moving, moving_idx, 1) || n_poses < 0) return RANDT_ERR_INVALID;
  if (n_poses == 0) return RANDT_OK;
  if (!d_corr || !d_poses4 || !d_cost || mp->n_neighbours <= 0 || !isfinite(mp->loss_alpha) || !(isfinite(scale) && scale > 0.0))
    return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  return launch_eval_cost(ctx, fixed->v, fixed_idx, moving->v, moving_idx, d_corr, mp->n_neighbours, mp->use_intensity, scale,
                          mp->loss_alpha, d_poses4, n_poses, d_cost, d_n_res);
}

int randt_cs_divergence_batch_dev(randt_ctx* ctx, const randt_maps* fixed, int fixed_first, int fixed_count,
                                  const int32_t* d_fixed_idx, const randt_maps* moving, int moving_first, int n_pairs,
                                  const double* d_pose4, double* d_out, double* d_terms) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !range_ok(fixed, fixed_first, fixed_count) || fixed_count < 1 || n_pairs < 0 || !range_ok(moving, moving_first, n_pairs))
    return RANDT_ERR_INVALID;
  if (n_pairs == 0) return RANDT_OK;
  if (!d_out) return RANDT_ERR_INVALID;
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  const int max_tiles = (fixed->v.cap + RANDT_CS_SELF_OUTER - 1) / RANDT_CS_SELF_OUTER;  // partial sums of the fixed maps' self terms (csdiv.hip)
  int rc = ensure_ws(ctx, sizeof(double) * (size_t)fixed_count * max_tiles + 256);
  if (rc) return rc;
  return launch_cs_divergence(ctx, fixed->v, fixed_first, fixed_count, d_fixed_idx, moving->v, moving_first, n_pairs, d_pose4,
                              (double*)ctx->ws, d_out, d_terms);
}

int randt_cs_divergence(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                        const double* h_pose4, double* out, double* h_terms) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !out || !range_ok(fixed, fixed_idx, 1) || !range_ok(moving, moving_idx, 1)) return RANDT_ERR_INVALID;
  // device scratch of this call: pose | fixed index | result | terms (kept apart from the workspace the batch entry uses)
  char* d_blk = nullptr;
  int rc = small_block(ctx, &d_blk);
  if (rc) return rc;
  double* d_pose = reinterpret_cast<double*>(d_blk);
  int32_t* d_fi = reinterpret_cast<int32_t*>(d_blk + 64);
  double* d_out = reinterpret_cast<double*>(d_blk + 128);
  double* d_terms = d_out + 1;
  hipError_t e = hipSuccess;
  if (h_pose4) e = hipMemcpyAsync(d_pose, h_pose4, sizeof(double) * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_fi, &fixed_idx, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "hipMemcpyAsync", e);
  if (!rc) rc = randt_cs_divergence_batch_dev(ctx, fixed, fixed_idx, 1, d_fi, moving, moving_idx, 1, h_pose4 ? d_pose : nullptr, d_out, d_terms);
  if (!rc) {
    e = hipMemcpyAsync(out, d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && h_terms) e = hipMemcpyAsync(h_terms, d_terms, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "cs divergence read-back", e);
  } else {
    (void)randt_sync(ctx);
  }
  return rc;
}

int randt_sc_make_batch_dev(randt_ctx* ctx, const float* d_points, int n_scans, int points_pitch, const int32_t* d_n_points,
                            int stride_floats, int intensity_index, const randt_sc_params* p, double* d_desc, double* d_ring_keys,
                            double* d_sector_keys) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !p || n_scans < 0 || points_pitch <= 0 || stride_floats < 3 || intensity_index < 0 || intensity_index >= stride_floats)
    return RANDT_ERR_INVALID;
  if (n_scans == 0) return RANDT_OK;
  if (!d_points || !d_desc || !d_ring_keys || !d_sector_keys) return RANDT_ERR_INVALID;
  return launch_sc_make(ctx, d_points, n_scans, points_pitch, d_n_points, stride_floats, intensity_index, p, d_desc, d_ring_keys,
                        d_sector_keys);
}

int randt_sc_detect_batch_dev(randt_ctx* ctx, const randt_sc_params* p, const double* d_desc, const double* d_ring_keys,
                              const double* d_pos, const double* d_dist, int n_db, const int32_t* d_query_ids, int n_queries,
                              int32_t* d_loop_id, float* d_yaw, double* d_min_dist) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !p || n_db < 0 || n_queries < 0) return RANDT_ERR_INVALID;
  if (n_queries == 0) return RANDT_OK;
  if (!d_desc || !d_ring_keys || !d_pos || !d_dist || !d_loop_id || !d_yaw) return RANDT_ERR_INVALID;
  int rc = ensure_ws(ctx, sc_detect_ws_bytes(n_queries, n_db, p->num_candidates));
  if (rc) return rc;
  return launch_sc_detect(ctx, p, d_desc, d_ring_keys, d_pos, d_dist, n_db, d_query_ids, n_queries, (float*)ctx->ws, d_loop_id, d_yaw,
                          d_min_dist);
}

// ---------------------------------------------------------------- Scan Context database -----------------
struct randt_sc_db {
  randt_ctx* ctx = nullptr;
  randt_sc_params p{};
  int n = 0, cap = 0;
  double *desc = nullptr, *ring = nullptr, *sector = nullptr, *pos = nullptr, *dist = nullptr;  // device
  int32_t* out_id = nullptr;  // device scratch: loop id | yaw | min dist
  float* out_yaw = nullptr;
  double* out_md = nullptr;
};

namespace {
int sc_db_reserve(randt_sc_db* db, int want) {
  if (want <= db->cap) return RANDT_OK;
  randt_ctx* ctx = db->ctx;
  int cap = db->cap > 0 ? db->cap : 64;
  while (cap < want) cap *= 2;
  const size_t nd = (size_t)db->p.num_ring * db->p.num_sector;
  struct Arr { double** p; size_t per; } arrs[5] = {{&db->desc, nd}, {&db->ring, (size_t)db->p.num_ring}, {&db->sector, (size_t)db->p.num_sector},
                                                   {&db->pos, 2}, {&db->dist, 1}};
  for (auto& a : arrs) {
    double* fresh = nullptr;
    RANDT_HIP_CHECK(ctx, hipMalloc(&fresh, sizeof(double) * a.per * cap));
    if (db->n > 0)
      RANDT_HIP_CHECK(ctx, hipMemcpyAsync(fresh, *a.p, sizeof(double) * a.per * db->n, hipMemcpyDeviceToDevice, ctx->stream));
    RANDT_HIP_CHECK(ctx, randt_sync(ctx));
    if (*a.p) (void)hipFree(*a.p);
    *a.p = fresh;
  }
  db->cap = cap;
  return RANDT_OK;
}
}  // namespace

int randt_sc_db_create(randt_ctx* ctx, const randt_sc_params* p, int initial_capacity, randt_sc_db** out) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !p || !out || p->num_ring < 1 || p->num_sector < 1) return RANDT_ERR_INVALID;
  randt_sc_db* db = new randt_sc_db();
  db->ctx = ctx;
  db->p = *p;
  hipError_t e = hipMalloc(&db->out_id, 64);
  if (e != hipSuccess) {
    delete db;
    return randt_set_error(ctx, RANDT_ERR_HIP, "hipMalloc", e);
  }
  db->out_yaw = reinterpret_cast<float*>(db->out_id + 2);
  db->out_md = reinterpret_cast<double*>(db->out_id + 4);
  int rc = sc_db_reserve(db, initial_capacity > 0 ? initial_capacity : 64);
  if (rc) {
    randt_sc_db_destroy(db);
    return rc;
  }
  *out = db;
  return RANDT_OK;
}

void randt_sc_db_destroy(randt_sc_db* db) {
  DeviceGuard dev_guard__(db ? db->ctx : nullptr);
  if (!db) return;
  for (double* p : {db->desc, db->ring, db->sector, db->pos, db->dist})
    if (p) (void)hipFree(p);
  if (db->out_id) (void)hipFree(db->out_id);
  delete db;
}

int randt_sc_db_size(const randt_sc_db* db) { return db ? db->n : 0; }

int randt_sc_db_append(randt_sc_db* db, const float* h_points, int n_points, int stride_floats, int intensity_index,
                       const double odom_position[2], double traversed_distance, int* node_id) {
  DeviceGuard dev_guard__(db ? db->ctx : nullptr);
  if (!db || n_points < 0 || (n_points > 0 && !h_points) || !odom_position || stride_floats < 3 || intensity_index < 0 ||
      intensity_index >= stride_floats)
    return RANDT_ERR_INVALID;
  randt_ctx* ctx = db->ctx;
  int rc = sc_db_reserve(db, db->n + 1);
  if (rc) return rc;
  const int pitch = n_points > 0 ? n_points : 1;
  const size_t bytes = sizeof(float) * (size_t)pitch * stride_floats;
  rc = ensure_ws(ctx, bytes + 64);
  if (rc) return rc;
  if (n_points > 0) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws, h_points, bytes, hipMemcpyHostToDevice, ctx->stream));
  int32_t* d_n = reinterpret_cast<int32_t*>((char*)ctx->ws + ((bytes + 15) & ~(size_t)15));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(d_n, &n_points, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  const size_t nd = (size_t)db->p.num_ring * db->p.num_sector;
  rc = launch_sc_make(ctx, (const float*)ctx->ws, 1, pitch, d_n, stride_floats, intensity_index, &db->p, db->desc + nd * db->n,
                      db->ring + (size_t)db->p.num_ring * db->n, db->sector + (size_t)db->p.num_sector * db->n);
  if (rc) return rc;
  const double pd[3] = {odom_position[0], odom_position[1], traversed_distance};
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(db->pos + 2 * (size_t)db->n, pd, sizeof(double) * 2, hipMemcpyHostToDevice, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(db->dist + db->n, pd + 2, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));  // the host buffers may go away
  if (node_id) *node_id = db->n;
  db->n += 1;
  return RANDT_OK;
}

int randt_sc_db_detect(randt_sc_db* db, int node_id, int* loop_id, float* yaw_diff_rad, double* min_dist) {
  DeviceGuard dev_guard__(db ? db->ctx : nullptr);
  if (!db || !loop_id || !yaw_diff_rad || node_id < 0 || node_id >= db->n) return RANDT_ERR_INVALID;
  randt_ctx* ctx = db->ctx;
  int rc = ensure_ws(ctx, sc_detect_ws_bytes(1, db->n, db->p.num_candidates));
  if (rc) return rc;
  int32_t* d_q = db->out_id + 1;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(d_q, &node_id, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  rc = launch_sc_detect(ctx, &db->p, db->desc, db->ring, db->pos, db->dist, db->n, d_q, 1, (float*)ctx->ws, db->out_id, db->out_yaw,
                        db->out_md);
  if (rc) return rc;
  int32_t id = -1;
  double md = 0;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(&id, db->out_id, sizeof(id), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(yaw_diff_rad, db->out_yaw, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(&md, db->out_md, sizeof(md), hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  *loop_id = id;
  if (min_dist) *min_dist = md;
  return RANDT_OK;
}

int randt_sc_db_download(const randt_sc_db* db, int node_id, double* h_desc, double* h_ring_key, double* h_sector_key) {
  DeviceGuard dev_guard__(db ? db->ctx : nullptr);
  if (!db || node_id < 0 || node_id >= db->n) return RANDT_ERR_INVALID;
  randt_ctx* ctx = db->ctx;
  const size_t nd = (size_t)db->p.num_ring * db->p.num_sector;
  if (h_desc) RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_desc, db->desc + nd * node_id, sizeof(double) * nd, hipMemcpyDeviceToHost, ctx->stream));
  if (h_ring_key)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_ring_key, db->ring + (size_t)db->p.num_ring * node_id, sizeof(double) * db->p.num_ring,
                                        hipMemcpyDeviceToHost, ctx->stream));
  if (h_sector_key)
    RANDT_HIP_CHECK(ctx, hipMemcpyAsync(h_sector_key, db->sector + (size_t)db->p.num_sector * node_id, sizeof(double) * db->p.num_sector,
                                        hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  return RANDT_OK;
}

int randt_filter_scan_batch_dev(randt_ctx* ctx, const float* d_raw, int n_scans, int n_azimuths, int n_bins,
                                int stride_floats, int intensity_index, const randt_filter_params* fp,
                                float* d_out_points, int pitch_out, int32_t* d_out_counts, float* d_out_polar,
                                float* d_peaks, int32_t* d_peak_counts, int32_t* d_status) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !fp || n_scans < 0 || n_azimuths <= 0 || n_bins <= 0 || stride_floats < 3 || intensity_index < 0 ||
      intensity_index >= stride_floats || pitch_out <= 0)
    return RANDT_ERR_INVALID;
  if (n_scans == 0) return RANDT_OK;
  if (!d_raw || !d_out_points || !d_out_counts || !d_status) return RANDT_ERR_INVALID;
  if (stride_floats == 4 && ((size_t)d_raw & 15) != 0) return randt_set_error(ctx, RANDT_ERR_INVALID, "packed xyzI input must be 16-byte aligned", hipSuccess);
  if ((long long)n_azimuths * n_bins > (1ll << 30)) return RANDT_ERR_UNSUPPORTED;
  int rc = ensure_ws(ctx, (size_t)n_scans * n_azimuths * (32 + 4 * 2 * 16) + 512);  // row records + FILT_STAGE staged points per row (filter.hip)
  if (rc) return rc;
  return launch_filter_scan(ctx, d_raw, n_scans, n_azimuths, n_bins, stride_floats, intensity_index, fp, d_out_points, pitch_out,
                            d_out_counts, d_out_polar, d_peaks, d_peak_counts, d_status, ctx->ws);
}

// one raw scan from the host into a pooled device block [raw | points | polar | peaks | counts, peak count, status]
namespace {
struct FilterBlock {
  char* base = nullptr;
  size_t bytes = 0;
  float *raw = nullptr, *pts = nullptr, *polar = nullptr, *peaks = nullptr;
  int32_t* tail = nullptr;  // [0] count, [1] peak count, [2] status
};
int filter_upload(randt_ctx* ctx, const float* h_raw, int n_az, int n_bins, int stride, int capacity, bool want_polar, bool want_peaks, FilterBlock* b) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t rb = up(sizeof(float) * (size_t)n_az * n_bins * stride), pb = up(sizeof(float) * 4 * (size_t)capacity),
               qb = want_polar ? up(sizeof(float) * 2 * (size_t)capacity) : 0, kb = want_peaks ? up(sizeof(float) * 3 * (size_t)n_az) : 0;
  void* blk = nullptr;
  hipError_t e = randt_dev_alloc(ctx, &blk, rb + pb + qb + kb + 256, &b->bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return randt_set_error(ctx, e == hipErrorOutOfMemory ? RANDT_ERR_NOMEM : RANDT_ERR_HIP, "hipMalloc (raw scan)", e);
  }
  b->base = static_cast<char*>(blk);
  b->raw = reinterpret_cast<float*>(b->base);
  b->pts = reinterpret_cast<float*>(b->base + rb);
  b->polar = want_polar ? reinterpret_cast<float*>(b->base + rb + pb) : nullptr;
  b->peaks = want_peaks ? reinterpret_cast<float*>(b->base + rb + pb + qb) : nullptr;
  b->tail = reinterpret_cast<int32_t*>(b->base + rb + pb + qb + kb);
  e = hipMemcpyAsync(b->raw, h_raw, sizeof(float) * (size_t)n_az * n_bins * stride, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = randt_sync(ctx);  // the caller's buffer (pageable or pinned) is free on return
  if (e != hipSuccess) {
    randt_dev_release(ctx, blk, b->bytes);
    b->base = nullptr;
    return randt_set_error(ctx, RANDT_ERR_HIP, "raw scan upload", e);
  }
  return RANDT_OK;
}
}  // namespace

int randt_filter_scan(randt_ctx* ctx, const float* h_raw, int n_azimuths, int n_bins, int stride_floats, int intensity_index,
                      const randt_filter_params* fp, float* h_out_points, int capacity, int* n_out, float* h_out_polar, float* h_peaks,
                      int* n_peaks, int* status) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !fp || !h_raw || !h_out_points || !n_out || !status || n_azimuths <= 0 || n_bins <= 0 || capacity <= 0 || stride_floats < 3 ||
      intensity_index < 0 || intensity_index >= stride_floats || (long long)n_azimuths * n_bins > (1ll << 30))
    return RANDT_ERR_INVALID;
  FilterBlock b;
  int rc = filter_upload(ctx, h_raw, n_azimuths, n_bins, stride_floats, capacity, h_out_polar != nullptr, h_peaks != nullptr, &b);
  if (rc) return rc;
  rc = randt_filter_scan_batch_dev(ctx, b.raw, 1, n_azimuths, n_bins, stride_floats, intensity_index, fp, b.pts, capacity, b.tail, b.polar, b.peaks,
                                   h_peaks ? b.tail + 1 : nullptr, b.tail + 2);
  int32_t h_tail[3] = {0, 0, 0};
  if (!rc) {
    hipError_t e = hipMemcpyAsync(h_tail, b.tail, sizeof(h_tail), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    const int n = h_tail[0] < capacity ? h_tail[0] : capacity;
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(h_out_points, b.pts, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && n > 0 && h_out_polar) e = hipMemcpyAsync(h_out_polar, b.polar, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && h_peaks && h_tail[1] > 0) e = hipMemcpyAsync(h_peaks, b.peaks, sizeof(float) * 3 * (size_t)h_tail[1], hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "filter read-back", e);
  } else {
    (void)randt_sync(ctx);
  }
  randt_dev_release(ctx, b.base, b.bytes);
  if (rc) return rc;
  *n_out = h_tail[0];
  if (n_peaks) *n_peaks = h_tail[1];
  *status = h_tail[2];
  return RANDT_OK;
}

int randt_filter_build(randt_ctx* ctx, const float* h_raw, int n_azimuths, int n_bins, int stride_floats, int intensity_index,
                       const randt_filter_params* fp, const randt_cluster_params* cp, int max_points, randt_maps* out, int map_idx, int* status) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !fp || !cp || !h_raw || !range_ok(out, map_idx, 1) || n_azimuths <= 0 || n_bins <= 0 || max_points <= 0 || stride_floats < 3 ||
      intensity_index < 0 || intensity_index >= stride_floats || (long long)n_azimuths * n_bins > (1ll << 30))
    return RANDT_ERR_INVALID;
  FilterBlock b;
  int rc = filter_upload(ctx, h_raw, n_azimuths, n_bins, stride_floats, max_points, false, false, &b);
  if (rc) return rc;
  rc = randt_filter_scan_batch_dev(ctx, b.raw, 1, n_azimuths, n_bins, stride_floats, intensity_index, fp, b.pts, max_points, b.tail, nullptr, nullptr,
                                   nullptr, b.tail + 2);
  if (!rc) rc = randt_ndt_build_batch_dev(ctx, b.pts, 1, max_points, b.tail, 4, 3, cp, out, map_idx);
  int32_t h_status = 0;
  if (!rc && status) {
    hipError_t e = hipMemcpyAsync(&h_status, b.tail + 2, sizeof(h_status), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "filter status read-back", e);
  }
  randt_dev_release(ctx, b.base, b.bytes);  // parked: whatever reuses it is enqueued behind the two kernels
  if (!rc && status) *status = h_status;
  return rc;
}

// ------------------------------------------------------------------ correlative search (f-3) ------
namespace {
struct BnbNode {
  double pose[4];
  int level;
};
void bnb_pose(double a, double tx, double ty, double* out);
void bnb_mul(const double* a, const double* b, double* out);
}  // namespace

int randt_search_global(randt_ctx* ctx, const randt_maps* fixed, int fixed_idx, const randt_maps* moving, int moving_idx,
                        const randt_matcher_params* mp, const randt_bnb_params* bp, double scale, double swl, double swa,
                        double h_trans4[4], double* min_cost_out, int* n_evals) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !mp || !bp || !h_trans4 || !range_ok(fixed, fixed_idx, 1) || !range_ok(moving, moving_idx, 1)) return RANDT_ERR_INVALID;
  // the reference's grid loops advance by these steps (`tx += initial_linear_step`, `a += angular_step`, :527-541,
  // :584-600): a non-positive or non-finite step never terminates (and the node list grows until memory runs out)
  if (!(isfinite(bp->csm_linear_step) && bp->csm_linear_step > 0.0) || !(isfinite(bp->csm_max_px_accurate_range) && bp->csm_max_px_accurate_range > 0.0) ||
      bp->csm_n_iter < 1 || bp->csm_n_iter > 16 || !isfinite(swl) || !isfinite(swa) || !isfinite(bp->csm_window_linear) ||
      !isfinite(bp->csm_window_angular) || !isfinite(bp->csm_cost_threshold) || !(isfinite(scale) && scale > 0.0) || !isfinite(mp->loss_alpha) ||
      bp->csm_linear_step >= 2.0 * bp->csm_max_px_accurate_range /* acos argument < -1: angular step NaN */)
    return randt_set_error(ctx, RANDT_ERR_INVALID, "correlative search: steps / ranges must be finite and positive, csm_n_iter in 1..16", hipSuccess);
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  swl = fmin(swl, bp->csm_window_linear);   // ndt_matcher.cpp:505-506
  swa = fmin(swa, bp->csm_window_angular);
  const int k = 4;                          // addNDTFactor(..., 4), :520
  const double linear_step = bp->csm_linear_step, max_range = bp->csm_max_px_accurate_range;
  const double angular_step = acos(1 - ((linear_step * linear_step) / (2 * max_range * max_range)));
  const size_t n_iter = (size_t)bp->csm_n_iter;
  const double initial_linear_step = pow(2, (double)n_iter - 1) * linear_step;
  std::vector<BnbNode> level_nodes, next_nodes;
  std::vector<float> keys;  // std::vector<std::vector<float>> calculated_points, 9 floats each
  auto key_of = [](const double* p, float* key) {
    key[0] = (float)p[0]; key[1] = (float)p[1]; key[2] = 0.f;
    key[3] = (float)(-p[1]); key[4] = (float)p[0]; key[5] = 0.f;
    key[6] = (float)p[2]; key[7] = (float)p[3]; key[8] = 1.f;
  };
  for (double tx = -swl / 2.0; tx <= swl / 2.0; tx += initial_linear_step)
    for (double ty = -swl / 2.0; ty <= swl / 2.0; ty += initial_linear_step)
      for (double a = -swa / 2.0; a < swa / 2.0; a += angular_step) {
        double d4[4];
        BnbNode nd;
        bnb_pose(a, tx, ty, d4);
        bnb_mul(h_trans4, d4, nd.pose);
        nd.level = 1;
        level_nodes.push_back(nd);
        float key[9];
        key_of(nd.pose, key);
        keys.insert(keys.end(), key, key + 9);
      }
  // association once at the guess (frozen for the whole search, quirk A.7-8)
  randt_matcher_params amp = *mp;
  amp.n_neighbours = k;
  const size_t corr_bytes = sizeof(int32_t) * (size_t)moving->v.cap * k;
  double min_cost = 100000.0;
  double best[4] = {1.0, 0.0, 0.0, 0.0};
  int evals = 0;
  void* d_blk = nullptr;  // [guess4 | idx | n_res] + corr live in a pooled block of their own (ws is used for poses / costs)
  size_t blk_bytes = 0;
  RANDT_HIP_CHECK(ctx, randt_dev_alloc(ctx, &d_blk, 256 + corr_bytes, &blk_bytes));
  double* d_guess = (double*)d_blk;
  int32_t* d_idx = (int32_t*)((char*)d_blk + 64);
  int32_t* d_nres = d_idx + 4;
  int32_t* d_corr = (int32_t*)((char*)d_blk + 256);
  int32_t h_idx[2] = {fixed_idx, moving_idx};
  int rc = RANDT_OK;
  hipError_t e = hipMemcpyAsync(d_guess, h_trans4, sizeof(double) * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, h_idx, sizeof(h_idx), hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) rc = randt_set_error(ctx, RANDT_ERR_HIP, "hipMemcpyAsync", e);
  if (!rc) rc = launch_associate(ctx, fixed->v, d_idx, moving->v, 0, 1, d_guess, k, mp->lookup_mahalanobis, mp->use_intensity, d_corr, d_idx + 1);
  while (!rc && !level_nodes.empty()) {
    const int P = (int)level_nodes.size();
    rc = ensure_ws(ctx, (size_t)P * 40 + 64);
    if (rc) break;
    std::vector<double> h_poses((size_t)P * 4), h_cost(P);
    for (int i = 0; i < P; ++i) memcpy(&h_poses[4 * (size_t)i], level_nodes[i].pose, sizeof(double) * 4);
    double* d_poses = (double*)ctx->ws;
    double* d_cost = d_poses + (size_t)P * 4;
    int32_t n_res = 0;
    e = hipMemcpyAsync(d_poses, h_poses.data(), sizeof(double) * 4 * P, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { rc = randt_set_error(ctx, RANDT_ERR_HIP, "hipMemcpyAsync", e); break; }
    rc = launch_eval_cost(ctx, fixed->v, fixed_idx, moving->v, moving_idx, d_corr, k, mp->use_intensity, scale, mp->loss_alpha, d_poses, P, d_cost, d_nres);
    if (rc) break;
    e = hipMemcpyAsync(h_cost.data(), d_cost, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&n_res, d_nres, sizeof(n_res), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = randt_sync(ctx);
    if (e != hipSuccess) { rc = randt_set_error(ctx, RANDT_ERR_HIP, "cost read-back", e); break; }
    evals += P;
    next_nodes.clear();
    for (int i = 0; i < P; ++i) {   // FIFO order of the reference's queue (:560-605)
      const double current_cost = h_cost[i] / (double)n_res;
      const size_t level = (size_t)level_nodes[i].level;
      if (current_cost < bp->csm_cost_threshold) {
        if (current_cost < min_cost) {
          memcpy(best, level_nodes[i].pose, sizeof(best));
          min_cost = current_cost;
        }
        if (level < n_iter) {
          const double cls = pow(2.0, (double)level) * linear_step, cas = angular_step;
          for (double tx = -cls; tx <= cls; tx += cls)
            for (double ty = -cls; ty <= cls; ty += cls)
              for (double a = -cas; a <= cas; a += cas) {
                double d4[4];
                BnbNode nd;
                bnb_pose(a, tx, ty, d4);
                bnb_mul(level_nodes[i].pose, d4, nd.pose);
                nd.level = (int)level + 1;
                float key[9];
                key_of(nd.pose, key);
                bool found = false;
                for (size_t t = 0; t + 9 <= keys.size() && !found; t += 9) {  // std::vector<float> ==: element-wise, -0.0f == 0.0f
                  bool same = true;
                  for (int c = 0; c < 9 && same; ++c) same = keys[t + c] == key[c];
                  found = same;
                }
                if (!found) {
                  keys.insert(keys.end(), key, key + 9);
                  next_nodes.push_back(nd);
                }
              }
        }
      }
    }
    level_nodes.swap(next_nodes);
  }
  randt_dev_release(ctx, d_blk, blk_bytes);  // every level ended with a synchronisation: nothing reads it any more
  if (rc) return rc;
  memcpy(h_trans4, best, sizeof(best));  // trans = best_trans (identity if nothing qualified), :606
  if (min_cost_out) *min_cost_out = min_cost;
  if (n_evals) *n_evals = evals;
  return RANDT_OK;
}

// ------------------------------------------------------------------ fixed-lag window (a16, a17) --
namespace {
void h_so2_normalize(double& c, double& s) {
  const double len = sqrt(c * c + s * s);
  c = c / len;
  s = s / len;
}
// Sophus SE2::exp and group product (se2.hpp, so2.hpp), host copies for the O(1) prediction step
void h_se2_exp(const double* xi, double* out) {
  const double theta = xi[2];
  double c = cos(theta), s = sin(theta);
  h_so2_normalize(c, s);
  double sbt, omcbt;
  if (fabs(theta) < 1e-10) {
    const double tsq = theta * theta;
    sbt = 1.0 - (1.0 / 6.0) * tsq;
    omcbt = 0.5 * theta - (1.0 / 24.0) * theta * tsq;
  } else {
    sbt = s / theta;
    omcbt = (1.0 - c) / theta;
  }
  out[0] = c;
  out[1] = s;
  out[2] = sbt * xi[0] - omcbt * xi[1];
  out[3] = omcbt * xi[0] + sbt * xi[1];
}
void h_se2_mul(const double* a, const double* b, double* out) {
  double re = a[0] * b[0] - a[1] * b[1];
  double im = a[0] * b[1] + a[1] * b[0];
  const double sq = re * re + im * im;
  if (sq != 1.0) {
    const double scale = 2.0 / (1.0 + sq);
    re *= scale;
    im *= scale;
  }
  h_so2_normalize(re, im);
  const double tx = a[2] + (a[0] * b[2] - a[1] * b[3]);
  const double ty = a[3] + (a[1] * b[2] + a[0] * b[3]);
  out[0] = re;
  out[1] = im;
  out[2] = tx;
  out[3] = ty;
}
// Sophus::SE2d(a, {tx, ty}) and the group product for the BNB pose grid
void bnb_pose(double a, double tx, double ty, double* out) {
  double c = cos(a), s = sin(a);
  h_so2_normalize(c, s);
  out[0] = c; out[1] = s; out[2] = tx; out[3] = ty;
}
void bnb_mul(const double* a, const double* b, double* out) { h_se2_mul(a, b, out); }
}  // namespace

int randt_predict_state(const randt_state* last, double stamp, randt_state* next) {
  if (!last || !next) return RANDT_ERR_INVALID;
  // predictSE2 with last_state.lin_acc = 0 (ndt_matcher.cpp:26,44-54; ceres_residuals.h:62-83)
  const double raw_dt = stamp - last->stamp;
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  const double screw[3] = {last->lin_vel[0] * dt, last->lin_vel[1] * dt, last->rot_vel * dt};
  double e[4];
  randt_state n;
  memset(&n, 0, sizeof(n));
  h_se2_exp(screw, e);
  h_se2_mul(last->pose, e, n.pose);
  n.lin_vel[0] = last->lin_vel[0];
  n.lin_vel[1] = last->lin_vel[1];
  n.rot_vel = last->rot_vel;
  n.lin_acc[0] = n.lin_acc[1] = 0.0;
  n.pos[0] = n.pose[2];
  n.pos[1] = n.pose[3];
  n.rot = atan2(n.pose[1], n.pose[0]);
  n.imu_bias = 0.0;  // X_next_.imu_bias keeps Matcher::initialize's value (ndt_matcher.cpp:15)
  n.stamp = stamp;
  *next = n;
  return RANDT_OK;
}

// NormalizeAngle (include/ndt_registration/state_manifold.h:17-23)
static double h_normalize_angle(double a) { return a - 2.0 * M_PI * floor((a + M_PI) / (2.0 * M_PI)); }

int randt_predict_state_param(const randt_state* last, double stamp, int parameterization, randt_state* next) {
  if (!last || !next) return RANDT_ERR_INVALID;
  if (parameterization == RANDT_PARAM_MANIFOLD) return randt_predict_state(last, stamp, next);
  if (parameterization != RANDT_PARAM_VECTOR && parameterization != RANDT_PARAM_ANALYTIC) return RANDT_ERR_INVALID;  // (pos, rot) forms
  // predict(...) with last_state.lin_acc = 0 (ndt_matcher.cpp:26-41; ceres_residuals.h:91-123)
  const double raw_dt = stamp - last->stamp;
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  randt_state n;
  memset(&n, 0, sizeof(n));
  double new_rot = last->rot;
  new_rot += dt * last->rot_vel;
  new_rot = h_normalize_angle(new_rot);
  const double rot = h_normalize_angle(last->rot + 0.5 * dt * last->rot_vel);
  const double sy = sin(rot), cy = cos(rot);
  const double half_dt2 = 0.5 * dt * dt;
  const double delta_x = last->lin_vel[0] * dt + 0.0 * half_dt2;
  const double delta_y = last->lin_vel[1] * dt + 0.0 * half_dt2;
  n.pos[0] = last->pos[0] + (cy * delta_x - sy * delta_y);
  n.pos[1] = last->pos[1] + (sy * delta_x + cy * delta_y);
  n.rot = new_rot;
  n.lin_vel[0] = last->lin_vel[0];
  n.lin_vel[1] = last->lin_vel[1];
  n.rot_vel = last->rot_vel;
  // X_next_.pose = Sophus::SE2d(X_next_.rot, X_next_.pos) (:41): SO2(theta) = (cos, sin)
  n.pose[0] = cos(new_rot);
  n.pose[1] = sin(new_rot);
  n.pose[2] = n.pos[0];
  n.pose[3] = n.pos[1];
  n.imu_bias = 0.0;
  n.stamp = stamp;
  *next = n;
  return RANDT_OK;
}

int randt_predict_state_batch(const randt_state* last, int n, double stamp, int parameterization, randt_state* next) {
  if (n < 0 || (n > 0 && (!last || !next))) return RANDT_ERR_INVALID;
  for (int i = 0; i < n; ++i) {
    const int rc = randt_predict_state_param(&last[i], stamp, parameterization, &next[i]);
    if (rc) return rc;
  }
  return RANDT_OK;
}

// n_windows fixed-lag windows of ONE shape (the same number of states and of fixed maps: replicas in lock-step) in one
// association launch + one solve launch (a workgroup per window), one pinned image up and one back, one synchronisation.
// Arrays are window-major: h_fixed_idx[w][n_fixed], h_moving_idx[w][S], h_states[w][n_states], h_imu[w][S], h_trans4[w][4].
static int register_windows(randt_ctx* ctx, int n_windows, const randt_maps* fixed, const int32_t* h_fixed_idx, int n_fixed,
                            const randt_maps* moving, const int32_t* h_moving_idx, randt_state* h_states, int n_states,
                            const double* h_imu, const randt_matcher_params* mp, const randt_window_params* wp, double* h_trans4,
                            int* rejected, randt_result* h_results) {
  DeviceGuard dev_guard__(ctx);
  if (!ctx || !fixed || !moving || !h_fixed_idx || !h_moving_idx || !h_states || !mp || !wp || !h_trans4 || n_windows < 0) return RANDT_ERR_INVALID;
  if (n_windows == 0) return RANDT_OK;
  const int S = n_states - 1;
  if (S < 1 || S > RANDT_WIN_MAX_STATES - 1 || n_fixed < 1 || n_fixed > 2)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "window: 1..12 optimised states, 1..2 fixed maps", hipSuccess);
  if (mp->parameterization != RANDT_PARAM_MANIFOLD && mp->parameterization != RANDT_PARAM_VECTOR && mp->parameterization != RANDT_PARAM_ANALYTIC)
    return randt_set_error(ctx, RANDT_ERR_INVALID, "window solve: parameterization must be RANDT_PARAM_MANIFOLD, _VECTOR or _ANALYTIC", hipSuccess);
  const bool vec = mp->parameterization != RANDT_PARAM_MANIFOLD;
  if (mp->n_neighbours <= 0) return randt_set_error(ctx, RANDT_ERR_INVALID, "n_neighbours must be >= 1", hipSuccess);
  if (mp->n_neighbours > 16)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "window solve: n_neighbours <= 16, like the pair registration (the association keeps at most sixteen candidates per cell; every shipped configuration uses 4)", hipSuccess);
  {
    const int prc = check_matcher_params(ctx, mp);
    if (prc) return prc;
    if (!(isfinite(wp->ndt_weight) && wp->ndt_weight > 0.0) || !isfinite(wp->weight_imu) || !isfinite(wp->weight_imu_bias))
      return randt_set_error(ctx, RANDT_ERR_INVALID, "window weights must be finite (ndt_weight > 0)", hipSuccess);
  }
  for (int w = 0; w < n_windows; ++w) {
    for (int f = 0; f < n_fixed; ++f)
      if (!range_ok(fixed, h_fixed_idx[(size_t)w * n_fixed + f], 1)) return RANDT_ERR_INVALID;
    for (int j = 0; j < S; ++j)
      if (!range_ok(moving, h_moving_idx[(size_t)w * S + j], 1)) return RANDT_ERR_INVALID;
  }
  randt_note_user(ctx, fixed);
  randt_note_user(ctx, moving);
  const int k = mp->n_neighbours;
  const int T = S * n_fixed;  // NDT terms per window
  const size_t NT = (size_t)n_windows * T;
  if (NT * moving->v.cap * k > 0x7fffffffull) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "window batch: correspondence tables beyond 2^31 entries", hipSuccess);

  // device workspace: corr | [ states | results | guess | fixed idx | moving idx | descriptors ] -- the bracketed span is ONE
  // host image, staged in pinned memory and moved with one copy per direction: all of it up, [ states | results ] back
  // (five small pageable copies cost ~40 us per scan)
  const size_t st_stride = 12 * RANDT_WIN_MAX_STATES;  // doubles per window (window.hip, ST_STRIDE x states)
  const size_t corr_stride = (size_t)T * moving->v.cap * k;
  const size_t corr_bytes = sizeof(int32_t) * corr_stride * n_windows;
  const size_t off_states = (corr_bytes + 255) & ~(size_t)255;
  const size_t res_bytes = sizeof(randt_result) * n_windows;
  const size_t off_res = off_states + sizeof(double) * st_stride * n_windows;
  const size_t off_guess = off_res + res_bytes;
  const size_t off_fidx = off_guess + sizeof(double) * 4 * NT;
  const size_t off_midx = off_fidx + ((sizeof(int32_t) * NT + 63) & ~(size_t)63);
  const size_t off_desc = (off_midx + sizeof(int32_t) * NT + 255) & ~(size_t)255;
  const size_t up_span = off_desc + sizeof(WinDesc) * n_windows - off_states;
  const size_t back_span = off_guess - off_states;  // states | results
  int rc = ensure_ws(ctx, off_states + up_span + 64);
  if (rc) return rc;
  const size_t pin_need = up_span + back_span + 512;
  if (pin_need > ctx->h_pin_bytes) {
    if (ctx->h_pin) {
      RANDT_HIP_CHECK(ctx, randt_sync(ctx));
      (void)hipHostFree(ctx->h_pin);
      ctx->h_pin = nullptr;
      ctx->h_pin_bytes = 0;
    }
    const size_t want = pin_need < 16384 ? 16384 : pin_need + pin_need / 2;
    RANDT_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
    ctx->h_pin_bytes = want;
  }
  char* ws = (char*)ctx->ws;
  char* img = (char*)ctx->h_pin;  // upload image; the download lands behind it
  char* back = img + ((up_span + 255) & ~(size_t)255);
  memset(img, 0, up_span);
  double* h_guess = reinterpret_cast<double*>(img + (off_guess - off_states));
  int32_t* h_fi = reinterpret_cast<int32_t*>(img + (off_fidx - off_states));
  int32_t* h_mi = reinterpret_cast<int32_t*>(img + (off_midx - off_states));
  WinDesc* h_desc = reinterpret_cast<WinDesc*>(img + (off_desc - off_states));
  for (int w = 0; w < n_windows; ++w) {
    const randt_state* st = h_states + (size_t)w * n_states;
    const int32_t* fidx = h_fixed_idx + (size_t)w * n_fixed;
    const int32_t* midx = h_moving_idx + (size_t)w * S;
    const double* imu = h_imu ? h_imu + (size_t)w * S : nullptr;
    WinDesc& W = h_desc[w];
    W.S = S;
    W.vec = vec ? 1 : 0;
    W.pad_ = mp->parameterization == RANDT_PARAM_ANALYTIC ? 1 : 0;  // the NDT functor's analytic rotation Jacobian (launch_solve_window picks the instantiation)
    W.k = k;
    W.d3 = mp->use_intensity ? 1 : 0;
    W.const_vel = wp->use_constant_velocity_model ? 1 : 0;
    W.use_imu = (wp->use_imu && imu) ? 1 : 0;
    W.w_imu = wp->weight_imu;
    W.w_bias = wp->weight_imu_bias;
    W.ndt_weight = wp->ndt_weight;
    memcpy(W.sqrtI, wp->motion_sqrtI, sizeof(W.sqrtI));
    // tangent / ambient layout in Ceres' parameter-block order (ndt_matcher.cpp:290-320)
    int a = 0, t = 0;
    for (int j = 0; j <= S; ++j) {
      if (j == 0) { W.off_amb[j][0] = W.off_tan[j][0] = -1; } else { W.off_amb[j][0] = a; W.off_tan[j][0] = t; a += vec ? 3 : 4; t += 3; }
      W.off_amb[j][1] = a; W.off_tan[j][1] = t; a += 2; t += 2;
      W.off_amb[j][2] = a; W.off_tan[j][2] = t; a += 1; t += 1;
      if (W.const_vel) { W.off_amb[j][3] = W.off_tan[j][3] = -1; } else { W.off_amb[j][3] = a; W.off_tan[j][3] = t; a += 2; t += 2; }
      if (W.use_imu && j > 0) { W.off_amb[j][4] = a; W.off_tan[j][4] = t; a += 1; t += 1; } else { W.off_amb[j][4] = W.off_tan[j][4] = -1; }
      if (j > 0) {
        W.raw_dt[j] = st[j].stamp - st[j - 1].stamp;
        W.imu[j - 1] = W.use_imu ? imu[j - 1] : 0.0;
      }
    }
    W.n_amb = a;
    W.n_tan = t;
    W.n_terms = 0;
    for (int j = 1; j <= S; ++j)
      for (int f = 0; f < n_fixed; ++f) {
        const int q = W.n_terms++;
        W.term_state[q] = j;
        W.term_moving[q] = midx[j - 1];
        W.term_fixed[q] = fidx[f];
        h_fi[(size_t)w * T + q] = fidx[f];
        h_mi[(size_t)w * T + q] = midx[j - 1];
        memcpy(h_guess + 4 * ((size_t)w * T + q), st[j].pose, sizeof(double) * 4);  // association at the state's own pose (:364)
      }
    double* h_packed = reinterpret_cast<double*>(img) + st_stride * w;
    for (int j = 0; j <= S; ++j) {  // 12 doubles per state (window.hip, ST_STRIDE)
      double* o = h_packed + 12 * j;
      if (vec) {  // the parameters are pos and rot; the pose slots carry cos / sin of rot for the NDT pass
        o[0] = cos(st[j].rot); o[1] = sin(st[j].rot); o[2] = st[j].pos[0]; o[3] = st[j].pos[1];
      } else {
        memcpy(o, st[j].pose, sizeof(double) * 4);
      }
      o[4] = st[j].lin_vel[0]; o[5] = st[j].lin_vel[1]; o[6] = st[j].rot_vel;
      o[7] = st[j].lin_acc[0]; o[8] = st[j].lin_acc[1]; o[9] = st[j].imu_bias;
      o[10] = st[j].rot; o[11] = 0.0;
    }
  }
  // (the window descriptors are indexed dynamically by the kernel: they live in device memory, not in kernel arguments)
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(ws + off_states, img, up_span, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_associate(ctx, fixed->v, (const int32_t*)(ws + off_fidx), moving->v, 0, (int)NT, (const double*)(ws + off_guess), k, mp->lookup_mahalanobis,
                        mp->use_intensity, (int32_t*)ws, (const int32_t*)(ws + off_midx));
  if (rc) return rc;
  rc = launch_solve_window(ctx, fixed->v, moving->v, h_desc[0], (const WinDesc*)(ws + off_desc), (const int32_t*)ws, mp, (double*)(ws + off_states),
                           (randt_result*)(ws + off_res), n_windows, (int)corr_stride, (int)st_stride);
  if (rc) return rc;
  RANDT_HIP_CHECK(ctx, hipMemcpyAsync(back, ws + off_states, back_span, hipMemcpyDeviceToHost, ctx->stream));
  RANDT_HIP_CHECK(ctx, randt_sync(ctx));
  for (int w = 0; w < n_windows; ++w) {
    randt_state* st = h_states + (size_t)w * n_states;
    double* trans4 = h_trans4 + 4 * (size_t)w;
    const double prior_t[2] = {trans4[2], trans4[3]};
    const double prior_rot = atan2(trans4[1], trans4[0]);
    const double* h_packed = reinterpret_cast<const double*>(back) + st_stride * w;
    for (int j = 0; j <= S; ++j) {
      const double* o = h_packed + 12 * j;
      st[j].lin_vel[0] = o[4]; st[j].lin_vel[1] = o[5]; st[j].rot_vel = o[6];
      st[j].lin_acc[0] = o[7]; st[j].lin_acc[1] = o[8]; st[j].imu_bias = o[9];
      // both pose representations (ndt_matcher.cpp:399-406, local_fuser.cpp:141-150)
      st[j].pos[0] = o[2];
      st[j].pos[1] = o[3];
      if (vec) {  // Sophus::SE2d(rot, pos)
        st[j].rot = o[10];
        st[j].pose[0] = cos(o[10]); st[j].pose[1] = sin(o[10]); st[j].pose[2] = o[2]; st[j].pose[3] = o[3];
      } else {
        memcpy(st[j].pose, o, sizeof(double) * 4);
        st[j].rot = atan2(o[1], o[0]);
      }
    }
    // rejection gate (ndt_matcher.cpp:411-422)
    int rej = 0;
    {
      randt_state* X = &st[S];
      const double pc = cos(prior_rot), ps = sin(prior_rot);
      const double re = X->pose[0] * pc + X->pose[1] * ps, im = X->pose[0] * ps - X->pose[1] * pc;
      const double dth = atan2(im, re);
      if (fabs(X->pose[2] - prior_t[0]) > wp->pose_reject_translation || fabs(X->pose[3] - prior_t[1]) > wp->pose_reject_translation ||
          fabs(dth) > wp->pose_reject_rotation) {
        printf("Rejected new estimated transform!\n");
        rej = 1;
        memcpy(X->pos, st[S - 1].pos, sizeof(X->pos));
        memcpy(X->pose, st[S - 1].pose, sizeof(X->pose));
        X->rot = st[S - 1].rot;
        X->lin_vel[0] = X->lin_vel[1] = 0.0;
        X->rot_vel = 0.0;
        X->lin_acc[0] = X->lin_acc[1] = 0.0;
        X->imu_bias = st[S - 1].imu_bias;
      }
    }
    memcpy(trans4, st[S].pose, sizeof(double) * 4);
    if (rejected) rejected[w] = rej;
    if (h_results) memcpy(&h_results[w], back + (off_res - off_states) + sizeof(randt_result) * w, sizeof(randt_result));
  }
  return RANDT_OK;
}

int randt_register_window(randt_ctx* ctx, const randt_maps* fixed, const int32_t* h_fixed_idx, int n_fixed,
                          const randt_maps* moving, const int32_t* h_moving_idx, randt_state* h_states, int n_states,
                          const double* h_imu, const randt_matcher_params* mp, const randt_window_params* wp,
                          double h_trans4[4], int* rejected, randt_result* h_result) {
  return register_windows(ctx, 1, fixed, h_fixed_idx, n_fixed, moving, h_moving_idx, h_states, n_states, h_imu, mp, wp, h_trans4, rejected, h_result);
}

int randt_register_window_batch(randt_ctx* ctx, int n_windows, const randt_maps* fixed, const int32_t* h_fixed_idx, int n_fixed,
                                const randt_maps* moving, const int32_t* h_moving_idx, randt_state* h_states, int n_states,
                                const double* h_imu, const randt_matcher_params* mp, const randt_window_params* wp,
                                double* h_trans4, int* rejected, randt_result* h_results) {
  return register_windows(ctx, n_windows, fixed, h_fixed_idx, n_fixed, moving, h_moving_idx, h_states, n_states, h_imu, mp, wp, h_trans4, rejected, h_results);
}

}  // extern "C"
