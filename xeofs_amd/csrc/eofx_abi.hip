// eofx_abi.hip -- C ABI (include/eofx.h) of the MI355X-native EOF / randomized-SVD engine:
// launch logic, the randomized-SVD drivers and the small host-side linear algebra.
// Kernels live in eofx_kernels.hpp.  gfx950 only.
#include <chrono>
#include "eofx_hfft.hpp"
#include "eofx_kernels.hpp"
#include "eofx_fit.hpp"
#include "eofx_gram.hpp"
#include "eofx_axb_dma.hpp"
#include "eofx_hosteig.hpp"
#ifndef EOFX_AXB_DMA_DEFAULT
#define EOFX_AXB_DMA_DEFAULT 1
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <mutex>
#include <shared_mutex>
#include <condition_variable>
#include <deque>
#include <memory>
#include <vector>

#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <hipfft/hipfft.h>

#include "eofx.h"

using namespace eofx;

struct ncclUniqueIdCompat {      // rccl.h: typedef struct { char internal[128]; } ncclUniqueId  (passed BY VALUE to ncclCommInitRank)
  char internal[128];
};

// ------------------------------------------------------------------------------------
// context, arena, helpers
// ------------------------------------------------------------------------------------
// Communicator of the feature-sharded entry points: one process per GPU, every rank holds its slice of the
// feature axis, the only traffic is all-reduces of sample-side panels (n x L float32), L x L float64 Gram matrices and a
// few scalars.  Two bindings: RCCL (ncclAllReduce enqueued on the context's own stream -- no host round trip, no second
// stream, no Python between the passes; the library is opened at run time, nothing links against it) and a host callback
// (tests: torch.distributed / gloo between two processes that share one GPU).
struct EofxComm {
  int world = 1, rank = 0;
  // RCCL binding
  void* lib = nullptr;
  void* comm = nullptr;    // ncclComm_t
  int (*p_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*p_destroy)(void*) = nullptr;
  const char* (*p_errstr)(int) = nullptr;
  // callback binding
  eofx_allreduce_fn fn = nullptr;
  void* user = nullptr;
  // bookkeeping (eofx_ctx_comm_stats): collectives issued, bytes, milliseconds (events, only while profiling is on)
  int64_t calls = 0, bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

struct eofx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char* arena = nullptr;
  size_t arena_size = 0;
  size_t arena_off = 0;
  std::string err;
  // arithmetic of the matrix passes (EOFX_PREC_*): power iterations / final basis+projection
  int prec_power = EOFX_PREC_F16X3;
  int prec_final = EOFX_PREC_F16X3;
  // released resident-matrix buffers, kept for reuse: hipMalloc/hipFree of tens of GB cost
  // ~1 s, far more than a fit.  Bounded by pool_cap bytes; eofx_ctx_trim() empties it.
  std::vector<std::pair<void*, size_t>> pool;
  size_t pool_bytes = 0;
  size_t pool_cap = (size_t)160 << 30;
  // cached setups of the Hilbert stage (kernel spectrum + correction vectors per (n, padding, decay))
  struct HilbertSetup {
    int64_t n = 0, P = 0;
    int padding = 0;
    double decay = 0.0;
    void* chat = nullptr;  // device cfloat[P/2+1]   (hipFFT route, P > 16384)
    float* hperm = nullptr;  // device float[P]: filter table of the one-kernel route, in its LDS order (P <= 16384)
    float* u = nullptr;    // device float[4 n]: [4][n] (hipFFT route) or [n][4] (one-kernel route)
  };
  std::vector<HilbertSetup> hsetups;
  // cached operators of the Hilbert stage as resident n x n matrices (eofx_rsvd_hilbert_c64): Im = Hc y along the samples
  struct HilbertOp {
    int64_t n = 0;
    int padding = 0;
    double decay = 0.0;
    eofx_mat* m = nullptr;
  };
  std::vector<HilbertOp> hops;
  // cached hipFFT plans of the Hilbert stage: key = (P, batch) -> (R2C plan, C2R plan)
  std::vector<std::pair<std::pair<int64_t, int64_t>, std::pair<void*, void*>>> fft_plans;
  // layout policy of the next preprocess / apply (eofx_ctx_set_layout): 0 = write both layouts, 1 = keep a reference to
  // the raw field instead of the feature-contiguous layout, 2 = in place: write nothing, both products stream the field
  int keep_raw = 0;
  // layout mode 3: in place, and a field with all-NaN grid points (land / sea mask) keeps them as ZERO columns instead of
  // being compacted (eofx_mat::masked); only for callers that compact / scatter the feature axis of the factors themselves
  bool allow_masked = false;
  bool want_rawT = false;          // eofx_ctx_set_sample_raw: the next in-place preprocess also writes the transposed raw field
  float* pending_rawT = nullptr;   // ... produced by run_colstats, adopted by the matrix (or returned to the pool)
  size_t pending_rawT_bytes = 0;
  // optional per-launch timing of the dominant kernel (atb_f32) with HIP events on `stream`
  bool profile = false;
  struct ProfEvent {
    hipEvent_t first, second;
    int kind;   // 0: atb kernels (incl. the statistics-carrying first pass), 1: axb (in-place row stream), 2: unused
    ProfEvent(hipEvent_t a, hipEvent_t b, int k = 0) : first(a), second(b), kind(k) {}
  };
  std::vector<ProfEvent> prof_events;
  int64_t prof_kind_launches[3] = {0, 0, 0};   // of the last eofx_ctx_profile_read
  double prof_kind_ms[3] = {0.0, 0.0, 0.0};
  double prof_flops = 0.0;  // 2*K*M*L summed over profiled launches (padded sizes)
  double prof_bytes = 0.0;  // K*M*4 (the A stream) summed over profiled launches
  // Panel maxima taken where a panel is WRITTEN (split-K reduction, panel_matmul, import, atb epilogue) instead of
  // by one more read of it before the split-fp16 pass that consumes it.  Live only inside an AmaxScope (the rSVD
  // drivers): a pool of zeroed device words, one per produced panel, looked up by the panel's address.
  unsigned* amax_slots = nullptr;
  int amax_next = 0, amax_depth = 0;
  std::vector<std::pair<const float*, const unsigned*>> amax_known;
  // page-locked host scratch for small asynchronous downloads (Gram matrices, flags)
  double* pinned = nullptr;
  // cached work lists of gram_nt_kernel (eofx_gram.hpp): host plan + its device copy, keyed by the tile grid and stage count
  struct GramPlanDev {
    int nti = 0, ntj = 0, nst = 0;
    bool sym = false;
    GramPlan pl;
    GramItem* items = nullptr;
    int2* tiles = nullptr;
  };
  std::vector<GramPlanDev*> gram_plans;
  // fp16 planes of the panel an in-place X Y product reads (eofx_axb_dma.hpp): one grow-only buffer per context
  _Float16* axb_planes = nullptr;
  size_t axb_planes_bytes = 0;
  int axb_dma = -1;   // -1 not decided, 0 off, 1 by size, 2 always (EOFX_AXB_DMA)
  // the last eofx_fit_f32: [0] 1 when the statistics rode on the first pass, [1] ms of the non-pass work of the
  // fused preprocessor (probe, finalize, correction; HIP events, only with profiling on), [2] fallback reason
  double fit_info[4] = {0.0, 0.0, 0.0, 0.0};
  int last_iters = 0;   // power iterations of the last eofx_rsvd_c64 (its adaptive rule decides the count)
  EofxComm* comm = nullptr;   // feature-sharded entry points (eofx_fit_sharded_f32)
};
constexpr int EOFX_AMAX_SLOTS = 1024;
constexpr int EOFX_HILBERT_OP_MAX_N = 16384;   // resident n x n Hilbert operator (two layouts): 2 GB at the limit
constexpr size_t EOFX_PINNED_DOUBLES = 2 * 256 * 256 + 64;

struct eofx_mat {
  int64_t n = 0, p = 0, n_pad = 0, p_pad = 0;
  float* X = nullptr;   // [n_pad x p_pad]; absent in raw mode until something needs it (ensure_X)
  float* Xt = nullptr;  // [p_pad x n_pad]; absent in in-place mode until something needs it (ensure_Xt)
  // raw mode (eofx_ctx_set_layout): the feature-contiguous layout is the caller's RAW field [n x p] (or the staged copy
  // of a host field, owned here) read through the affine preprocessing map aff = {shift[p_pad], scale[p_pad]}
  const float* raw = nullptr;
  float* raw_owned = nullptr;
  size_t raw_owned_bytes = 0;
  int64_t raw_ld = 0;
  float* aff = nullptr;      // [3][p_pad]: shift hi, shift lo, scale (aff_pack_kernel)
  unsigned* absmax_dev = nullptr;  // float bits of max |x| (device scalar, for the fp16-split scaling)
  float absmax = 0.f;              // host copy, valid once the matrix is built
  // in-place matrix of a field with all-NaN grid points (land / sea mask): the invalid features stay where they are as
  // ZERO columns (scale 0 in `aff`, bits ANDed to +0 by the MASK kernels) instead of being compacted away; p counts them,
  // p_valid does not.  The host shell compacts / scatters the feature axis of the factors.
  bool masked = false;
  int64_t p_valid = 0;
  // masked matrices: the 64-feature slab pairs of axb_f16 that hold at least one valid feature (device, ascending; built
  // on first use by ensure_active_pairs; n_act < 0: not built yet, act == nullptr afterwards: every pair is active)
  int* act = nullptr;
  int64_t n_act = -1;
  // in-place matrix about to be Hilbert-transformed (eofx_ctx_set_sample_raw): the RAW field in the sample-contiguous layout
  // [p_pad x n_pad], written by the statistics pass on its way (colstats4_tr_kernel); eofx_hilbert_f32 reads it through the
  // Scaler map and returns it to the pool.  Rows >= p and samples >= n are not initialised (the consumer never reads them).
  float* rawT = nullptr;
};

static int set_err(eofx_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return set_err(ctx, _e == hipErrorOutOfMemory ? EOFX_ERR_NOMEM : EOFX_ERR_HIP,           \
                     "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)           \
  do {                      \
    int _rc = (expr);       \
    if (_rc != EOFX_OK) return _rc; \
  } while (0)
#define KCHK() HIPCHK(hipGetLastError())

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
         attr.type == hipMemoryTypeUnified;
}

// Stack allocator over one grow-only device buffer.  Growing is only legal with an empty
// stack (nothing carved is live), which every public entry point guarantees by reserving first.
static int arena_reserve(eofx_ctx* ctx, size_t bytes) {
  bytes = (size_t)round_up((int64_t)bytes, 256);
  if (bytes <= ctx->arena_size) return EOFX_OK;
  if (ctx->arena_off != 0) return set_err(ctx, EOFX_ERR_ARG, "internal: arena grown while in use");
  if (ctx->arena) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_size = 0;
  }
  HIPCHK(hipMalloc((void**)&ctx->arena, bytes));
  ctx->arena_size = bytes;
  return EOFX_OK;
}
template <typename T>
static T* arena_alloc(eofx_ctx* ctx, size_t count) {
  size_t bytes = (size_t)round_up((int64_t)(count * sizeof(T)), 256);
  if (ctx->arena_off + bytes > ctx->arena_size) return nullptr;
  T* p = reinterpret_cast<T*>(ctx->arena + ctx->arena_off);
  ctx->arena_off += bytes;
  return p;
}
struct ArenaScope {
  eofx_ctx* ctx;
  size_t mark;
  explicit ArenaScope(eofx_ctx* c) : ctx(c), mark(c->arena_off) {}
  ~ArenaScope() { ctx->arena_off = mark; }
};
#define ARENA(T, var, count)                                                     \
  T* var = arena_alloc<T>(ctx, (size_t)(count));                                 \
  if (!var) return set_err(ctx, EOFX_ERR_NOMEM, "internal: arena exhausted (%s)", #var)

static void pool_trim(eofx_ctx* ctx);

static int set_device(eofx_ctx* ctx) {
  HIPCHK(hipSetDevice(ctx->device));
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// Fence between the contexts of ONE process on one GPU.  Measured on this platform: FFT-type kernels -- the
// engine's transform kernel and rocFFT alike -- return occasional wrong 64-byte pieces while the split-fp16 in-place streaming
// kernels of ANOTHER stream share the chip (tools/thread_probe2/3/5.py: 72 % of rocFFT calls beside the in-place X^T Z kernel, 0
// alone; rocSOLVER's eigh, elementwise kernels and sorts are never affected; LDS / register canaries stay intact).  The cause is
// below the engine; the engine's own transform kernel is fenced instead: every entry point holds this device's lock SHARED
// for its duration, the Hilbert entries hold it EXCLUSIVE -- they first wait for the kernels every other context of the device
// has queued, and finish their own before they let go.  Re-entrant per thread (entries call entries); uncontended cost: one
// shared-lock acquisition per entry (~50 ns).  Other processes on the same GPU, and FFT work the host application runs on its own
// streams, are outside this fence.
// ------------------------------------------------------------------------------------
constexpr int EOFX_MAX_DEVICES = 64;
struct DeviceFence {
  std::shared_mutex mu;
  std::mutex reg_mu;
  std::vector<eofx_ctx*> ctxs;      // live contexts on this device
};
static DeviceFence g_fence[EOFX_MAX_DEVICES];
static thread_local int t_fence_depth[EOFX_MAX_DEVICES];
struct FenceGuard {
  eofx_ctx* ctx;
  int dev = -1;
  bool took = false, excl = false;
  FenceGuard(eofx_ctx* c, bool exclusive) : ctx(c) {
    if (!c || c->device < 0 || c->device >= EOFX_MAX_DEVICES) return;
    dev = c->device;
    if (t_fence_depth[dev]++ > 0) return;        // nested entry on this thread: the outermost guard holds the lock
    took = true;
    excl = exclusive;
    DeviceFence& f = g_fence[dev];
    if (!exclusive) {
      f.mu.lock_shared();
      return;
    }
    f.mu.lock();                                  // no other entry of this process is running on the device now ...
    std::lock_guard<std::mutex> g(f.reg_mu);
    for (eofx_ctx* o : f.ctxs)                    // ... and what the other contexts left queued has finished
      if (o != c && o->stream != c->stream) (void)hipStreamSynchronize(o->stream);
  }
  ~FenceGuard() {
    if (dev < 0) return;
    const bool outermost = --t_fence_depth[dev] == 0;
    if (!took || !outermost) return;
    if (excl) {
      (void)hipStreamSynchronize(ctx->stream);    // the transform has finished before any other context may queue a pass
      g_fence[dev].mu.unlock();
    } else {
      g_fence[dev].mu.unlock_shared();
    }
  }
};
#define ENTER(c)          \
  FenceGuard _fence((c), false); \
  CHK(set_device(c))
#define ENTER_EXCLUSIVE(c)      \
  FenceGuard _fence((c), true); \
  CHK(set_device(c))

extern "C" int eofx_abi_version(void) { return EOFX_ABI_VERSION; }

extern "C" int eofx_ctx_create(int device, void* stream, eofx_ctx** out) {
  if (!out) return EOFX_ERR_ARG;
  eofx_ctx* ctx = new eofx_ctx();
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    delete ctx;
    *out = nullptr;
    return EOFX_ERR_HIP;
  }
  if (device >= 0 && device < EOFX_MAX_DEVICES) {
    std::lock_guard<std::mutex> g(g_fence[device].reg_mu);
    g_fence[device].ctxs.push_back(ctx);
  }
  *out = ctx;
  return EOFX_OK;
}
extern "C" int eofx_ctx_destroy(eofx_ctx* ctx) {
  if (!ctx) return EOFX_OK;
  {
    FenceGuard fence(ctx, false);
    if (ctx->device >= 0 && ctx->device < EOFX_MAX_DEVICES) {
      std::lock_guard<std::mutex> g(g_fence[ctx->device].reg_mu);
      auto& v = g_fence[ctx->device].ctxs;
      v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
    }
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->arena) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->arena);
  }
  pool_trim(ctx);
  if (ctx->amax_slots) (void)hipFree(ctx->amax_slots);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (auto& e : ctx->fft_plans) {
    hipfftDestroy((hipfftHandle)e.second.first);
    hipfftDestroy((hipfftHandle)e.second.second);
  }
  (void)eofx_ctx_comm_clear(ctx);
  if (ctx->axb_planes) (void)hipFree(ctx->axb_planes);
  for (auto* g : ctx->gram_plans) {
    if (g->items) (void)hipFree(g->items);
    if (g->tiles) (void)hipFree(g->tiles);
    delete g;
  }
  for (auto& h : ctx->hops) (void)eofx_mat_destroy(ctx, h.m);
  ctx->hops.clear();
  for (auto& h : ctx->hsetups) {
    if (h.chat) (void)hipFree(h.chat);
    if (h.hperm) (void)hipFree(h.hperm);
    if (h.u) (void)hipFree(h.u);
  }
  delete ctx;
  return EOFX_OK;
}
// Move the context to another HIP stream (everything already queued on the old one is awaited first).  The Python shell
// binds a context to torch's CURRENT stream, so that torch operations between engine calls (collectives of the sharded
// driver, the model classes' glue) are ordered with the engine's kernels on any stream, not only on the default one.
extern "C" int eofx_ctx_set_stream(eofx_ctx* ctx, void* stream) {
  if (!ctx) return EOFX_ERR_ARG;
  ENTER(ctx);
  if ((hipStream_t)stream == ctx->stream) return EOFX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)stream;
  // per-stream state cached in the context: the hipFFT plans of the Hilbert stage were bound to the old stream
  for (auto& e : ctx->fft_plans) {
    if (hipfftSetStream((hipfftHandle)e.second.first, ctx->stream) != HIPFFT_SUCCESS ||
        hipfftSetStream((hipfftHandle)e.second.second, ctx->stream) != HIPFFT_SUCCESS)
      return set_err(ctx, EOFX_ERR_HIP, "hipfftSetStream failed while moving the context to another stream");
  }
  return EOFX_OK;
}
extern "C" int eofx_ctx_synchronize(eofx_ctx* ctx) {
  if (!ctx) return EOFX_ERR_ARG;
  ENTER(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}
extern "C" const char* eofx_last_error(const eofx_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int eofx_ctx_profile(eofx_ctx* ctx, int enable) {
  if (!ctx) return EOFX_ERR_ARG;
  ctx->profile = enable != 0;
  return EOFX_OK;
}
extern "C" int eofx_ctx_profile_read(eofx_ctx* ctx, int64_t* launches, double* total_ms, double* flops,
                                     double* bytes) {
  if (!ctx) return EOFX_ERR_ARG;
  ENTER(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double ms = 0.0;
  for (int k = 0; k < 3; ++k) {
    ctx->prof_kind_launches[k] = 0;
    ctx->prof_kind_ms[k] = 0.0;
  }
  for (auto& pr : ctx->prof_events) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, pr.first, pr.second));
    ms += t;
    ctx->prof_kind_launches[pr.kind] += 1;
    ctx->prof_kind_ms[pr.kind] += t;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (launches) *launches = (int64_t)ctx->prof_events.size();
  if (total_ms) *total_ms = ms;
  if (flops) *flops = ctx->prof_flops;
  if (bytes) *bytes = ctx->prof_bytes;
  ctx->prof_events.clear();
  ctx->prof_flops = 0.0;
  ctx->prof_bytes = 0.0;
  return EOFX_OK;
}

// launches / summed milliseconds of the LAST eofx_ctx_profile_read by streaming kernel: [0] atb (the transposed-operand
// kernels, all variants), [1] axb_f16_kernel (the in-place row stream), [2] unused
extern "C" int eofx_ctx_profile_by_kernel(const eofx_ctx* ctx, int64_t* launches3, double* ms3) {
  if (!ctx) return EOFX_ERR_ARG;
  for (int k = 0; k < 3; ++k) {
    if (launches3) launches3[k] = ctx->prof_kind_launches[k];
    if (ms3) ms3[k] = ctx->prof_kind_ms[k];
  }
  return EOFX_OK;
}

// copy device -> user pointer (host or device)
static int copy_out(eofx_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
  if (!dst || bytes == 0) return EOFX_OK;
  HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, ctx->stream));
  return EOFX_OK;
}
static int copy_in(eofx_ctx* ctx, void* dst_dev, const void* src, size_t bytes) {
  if (bytes == 0) return EOFX_OK;
  HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyDefault, ctx->stream));
  return EOFX_OK;
}

// ---- panel maxima recorded by the producers (see eofx_ctx::amax_slots) -------------------------------------------
struct AmaxScope {
  eofx_ctx* ctx;
  explicit AmaxScope(eofx_ctx* c) : ctx(c) {
    if (ctx->amax_depth++ == 0) {
      ctx->amax_known.clear();
      ctx->amax_next = 0;
      if (!ctx->amax_slots && hipMalloc((void**)&ctx->amax_slots, sizeof(unsigned) * EOFX_AMAX_SLOTS) != hipSuccess) {
        (void)hipGetLastError();
        ctx->amax_slots = nullptr;
      }
      if (ctx->amax_slots &&
          hipMemsetAsync(ctx->amax_slots, 0, sizeof(unsigned) * EOFX_AMAX_SLOTS, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->amax_next = EOFX_AMAX_SLOTS;   // unusable: every lookup misses
      }
    }
  }
  ~AmaxScope() {
    if (--ctx->amax_depth == 0) ctx->amax_known.clear();
  }
};
static void amax_forget(eofx_ctx* ctx, const float* panel) {
  for (size_t i = 0; i < ctx->amax_known.size(); ++i)
    if (ctx->amax_known[i].first == panel) {
      ctx->amax_known.erase(ctx->amax_known.begin() + i);
      return;
    }
}
// a fresh zeroed word for the producer of `panel` (nullptr outside a scope / when the pool is used up)
static unsigned* amax_new(eofx_ctx* ctx, const float* panel) {
  if (ctx->amax_depth == 0) return nullptr;
  amax_forget(ctx, panel);
  if (!ctx->amax_slots || ctx->amax_next >= EOFX_AMAX_SLOTS) return nullptr;
  unsigned* slot = ctx->amax_slots + ctx->amax_next++;
  ctx->amax_known.emplace_back(panel, slot);
  return slot;
}
static const float* amax_get(const eofx_ctx* ctx, const float* panel) {
  if (ctx->amax_depth == 0) return nullptr;
  for (auto& e : ctx->amax_known)
    if (e.first == panel) return reinterpret_cast<const float*>(e.second);
  return nullptr;
}
static int pinned_scratch(eofx_ctx* ctx, double** out) {
  if (!ctx->pinned) HIPCHK(hipHostMalloc((void**)&ctx->pinned, sizeof(double) * EOFX_PINNED_DOUBLES, hipHostMallocDefault));
  *out = ctx->pinned;
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// small host linear algebra
// ------------------------------------------------------------------------------------
// Symmetric eigen-decomposition: Householder tridiagonalisation followed by the implicit-shift
// QL iteration, eigenvectors accumulated (the classic tred2/tql2 scheme), float64.
// sqrt(a^2 + b^2): the plain form unless it over- or underflows (std::hypot's care costs 20-40 ns a call, and the QL iteration
// makes two per rotation)
static inline double hypot_fast(double a, double b) {
  const double q = a * a + b * b;
  if (q > 1e-280 && q < 1e280) return std::sqrt(q);
  return std::hypot(a, b);
}
extern "C" int eofx_host_zheigh_top_f64(const double* Hr, const double* Hi, int m, int nev, double* w, double* Xr, double* Xi) {
  if (!Hr || !Hi || !w || !Xr || !Xi || m <= 0 || nev <= 0 || nev > m) return EOFX_ERR_ARG;
  return hosteig::zheigh_top(Hr, Hi, m, nev, w, Xr, Xi) == 0 ? EOFX_OK : EOFX_ERR_LINALG;
}
extern "C" int eofx_host_eigh_f64(const double* Ain, int n, double* w, double* Vec) {
  if (!Ain || !w || !Vec || n <= 0) return EOFX_ERR_ARG;
  std::vector<double> z((size_t)n * n), d(n), e(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) z[(size_t)i * n + j] = 0.5 * (Ain[(size_t)i * n + j] + Ain[(size_t)j * n + i]);
#define Z(i, j) z[(size_t)(i) * n + (j)]
  // --- Householder reduction to tridiagonal form, accumulating the transformation in z
  for (int i = n - 1; i >= 1; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(Z(i, k));
      if (scale == 0.0) {
        e[i] = Z(i, l);
      } else {
        for (int k = 0; k <= l; ++k) {
          Z(i, k) /= scale;
          h += Z(i, k) * Z(i, k);
        }
        double f = Z(i, l);
        double g = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        Z(i, l) = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          Z(j, i) = Z(i, j) / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g += Z(j, k) * Z(i, k);
          for (int k = j + 1; k <= l; ++k) g += Z(k, j) * Z(i, k);
          e[j] = g / h;
          f += e[j] * Z(i, j);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = Z(i, j);
          e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k) Z(j, k) -= (f * e[k] + g * Z(i, k));
        }
      }
    } else {
      e[i] = Z(i, l);
    }
    d[i] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += Z(i, k) * Z(k, j);
        for (int k = 0; k <= l; ++k) Z(k, j) -= g * Z(k, i);
      }
    }
    d[i] = Z(i, i);
    Z(i, i) = 1.0;
    for (int j = 0; j <= l; ++j) Z(j, i) = Z(i, j) = 0.0;
  }
  // --- implicit QL on the tridiagonal (d, e).  The rotations touch two COLUMNS of the accumulated transformation at a time: they
  // run on its transpose, where those are two contiguous rows (vectorised; 247 -> ~150 us for the 60 x 60 problem a fit waits for)
  std::vector<double> zt((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) zt[(size_t)j * n + i] = z[(size_t)i * n + j];
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) return EOFX_ERR_LINALG;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot_fast(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = (r = hypot_fast(f, g));
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          {
            double* __restrict__ ri = zt.data() + (size_t)i * n;
            double* __restrict__ rj = zt.data() + (size_t)(i + 1) * n;
            for (int k = 0; k < n; ++k) {
              const double fk = rj[k], zk = ri[k];
              rj[k] = s * zk + c * fk;
              ri[k] = c * zk - s * fk;
            }
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
#undef Z
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] > d[b]; });
  for (int j = 0; j < n; ++j) {
    w[j] = d[idx[j]];
    for (int i = 0; i < n; ++i) Vec[(size_t)i * n + j] = zt[(size_t)idx[j] * n + i];
  }
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// kernel launchers
// ------------------------------------------------------------------------------------
struct AtbPlan {
  int S;
  int64_t kps;
};
// split-K factor from a small cost model: 512 resident workgroups, ~10 GB/s of A per
// workgroup slot, partial-sum traffic at ~4 TB/s.
// 128-column tiles (atb_f16_kernel<4>, one workgroup per CU): the wide products (Gram matrices, PCA panels) and the
// two-matrix form of a 128-column complex panel (33-64 complex columns: Re and Im are then streamed ONCE per pass)
// (round 3: also the single-matrix passes of sketches of 65 columns and more -- EOF with 55+ modes: one 128-column launch at
// 4.3 TB/s beats two 64-column launches at 6.5 TB/s: 229.7 -> 205.6 ms per rSVD at k = 100, config-4 size)
static bool atb_wide(int L, bool two_matrix = false) {
  (void)two_matrix;
  if (const char* ev = std::getenv("EOFX_ATB_WIDE_MIN")) return L >= atoi(ev);   // tuning hook (tools/wide_sketch_probe.py)
  return L >= 96;       // (96 columns: one partial 128-column tile, see launch_atb)
}
static AtbPlan atb_plan(int64_t M, int64_t K, int L, bool wide = false) {
  const int bx = (int)(M / ATB_BM);
  const int bz = wide ? (L + 127) / 128 : (L + 63) / 64;
  const double resident = wide ? 256.0 : 512.0;
  AtbPlan best{1, K};
  double best_t = 1e30;
  for (int S = 1; S <= 128; ++S) {
    const int64_t kps = round_up((K + S - 1) / S, ATB_KG);
    const int s_eff = (int)((K + kps - 1) / kps);
    if (s_eff != S) continue;
    const double blocks = (double)bx * bz * s_eff;
    const double rounds = std::ceil(blocks / resident);
    double t = rounds * (double)kps * 2048.0 / 1.0e10;
    if (s_eff > 1) t += (2.0 * s_eff + 1.0) * (double)M * L * 4.0 / 4.0e12;
    if (t < best_t * 0.98) {
      best_t = t;
      best = {s_eff, kps};
    }
  }
  return best;
}
// axb_f16_kernel (the in-place sample-side product): rows_pad / 256 row tiles x S feature splits.  S = 1 when the row
// tiles alone fill the chip, otherwise a multiple of 8 (one split per XCD at a time) from the same kind of cost model.
static AtbPlan axb_plan(int64_t rows_pad, int64_t K) {
  const int64_t rt = rows_pad / AXB_BM;
  const int64_t units = K / AXB_KG;
  if (rt >= 1024 || units < 16) return {1, K};
  if (const char* ev = std::getenv("EOFX_AXB_SPLITS")) {   // tuning hook (tools/atb_probe.py)
    const int want = std::max(8, atoi(ev) / 8 * 8);
    const int64_t kps = round_up((units + want - 1) / want, 1) * AXB_KG;
    return {(int)((K + kps - 1) / kps), kps};
  }
  AtbPlan best{1, K};
  double best_t = 1e30;
  for (int s8 = 1; s8 <= 32 && 8 * s8 <= units; ++s8) {
    const int S = 8 * s8;
    const int64_t kps = (units + S - 1) / S * AXB_KG;
    const int s_eff = (int)((K + kps - 1) / kps);
    if (s_eff > S || s_eff <= S - 8) continue;
    const double rounds = std::ceil((double)rt * s8 / 64.0);          // 64 resident workgroups per XCD
    double t = rounds * (double)kps * 1024.0 / 1.25e10;               // 256 rows x 4 B per feature, ~12.5 GB/s per slot
    t += (2.0 * s_eff + 1.0) * (double)rows_pad * 64.0 * 4.0 / 4.0e12;
    if (t < best_t * 0.98) {
      best_t = t;
      best = {s_eff, kps};
    }
  }
  return best;
}
static size_t atb_scratch_bytes(int64_t M, int64_t K, int L) {
  const AtbPlan pl = atb_plan(M, K, L), plw = atb_plan(M, K, L, true);
  const int smax = std::max(pl.S, atb_wide(L, true) ? plw.S : 1);
  size_t b = smax > 1 ? (size_t)smax * M * L * (sizeof(float) + sizeof(double)) + 4096 : 4096;   // f32 or f64 partials
  const AtbPlan px = axb_plan(M, round_up(K, AXB_KG));      // in case M is the sample side of an in-place matrix
  if (px.S > 1) b = std::max(b, (size_t)px.S * M * round_up(L, 64) * sizeof(float) + 4096);
  return b;
}

// C[M x L] = A[K x M]^T B[K x L]; M multiple of 512, K multiple of 16, L multiple of 32.
// raw view of A (atb_f16_kernel<NB, true>): the raw field with the affine preprocessing map applied on the fly
struct AffView {
  const float* aff = nullptr;   // {shift hi, shift lo, scale}[ld]
  int64_t ld = 0;
  int rows = 0;        // valid rows of the raw field
  int64_t cols = 0;    // valid columns
  bool masked = false; // some features carry scale 0 = all-NaN grid points kept as zero columns (MASK kernels)
};
template <int NB>
static void launch_atb_variant(int prec, dim3 grid, hipStream_t st, const float* A, int64_t lda,
                               const float* B, int ldb, float* out, int L, int64_t M, int64_t K,
                               int64_t kps, int col_base, float a_scale, const float* b_absmax,
                               const AffView* aff = nullptr, const float* A2 = nullptr, const float* B2 = nullptr,
                               int s_half = 0, int sym = 0, unsigned* amax_out = nullptr, int l_valid = 128) {
  if (prec == EOFX_PREC_F16X3 && A2)
    hipLaunchKernelGGL(atb_f16_kernel<NB>, grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base,
                       a_scale, b_absmax, (const float*)nullptr, (int64_t)0, 0, (int64_t)0, A2, B2, s_half, 0, amax_out, l_valid);
  else if (prec == EOFX_PREC_F16X3 && aff && aff->masked)
    hipLaunchKernelGGL((atb_f16_kernel<NB, true, true>), grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base,
                       a_scale, b_absmax, aff->aff, aff->ld, aff->rows, aff->cols, (const float*)nullptr, (const float*)nullptr, 0, sym, amax_out,
                       l_valid);
  else if (prec == EOFX_PREC_F16X3 && aff)
    hipLaunchKernelGGL((atb_f16_kernel<NB, true>), grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base,
                       a_scale, b_absmax, aff->aff, aff->ld, aff->rows, aff->cols, (const float*)nullptr, (const float*)nullptr, 0, sym, amax_out,
                       l_valid);
  else if (prec == EOFX_PREC_F16X3)
    hipLaunchKernelGGL(atb_f16_kernel<NB>, grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base,
                       a_scale, b_absmax, (const float*)nullptr, (int64_t)0, 0, (int64_t)0, (const float*)nullptr, (const float*)nullptr, 0, sym, amax_out,
                       l_valid);
  else if constexpr (NB <= 2) {     // the 128-column tile exists for the split-fp16 kernel only
    if (prec == EOFX_PREC_BF16X3)
      hipLaunchKernelGGL((atb_bf16_kernel<NB, 2>), grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base);
    else if (prec == EOFX_PREC_BF16X6)
      hipLaunchKernelGGL((atb_bf16_kernel<NB, 3>), grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base);
    else
      hipLaunchKernelGGL(atb_f32_kernel<NB>, grid, dim3(256), 0, st, A, lda, B, ldb, out, L, M, K, kps, col_base);
  }
}

// a_absmax: max |a| over A (host); b_absmax_dev: device scalar with max |b| (nullptr: measured here).
// Both are only used by the fp16-split variant.
// sym: C = A^T A (B is A itself, L == M): tiles strictly below the diagonal are skipped and mirrored afterwards
static int launch_atb(eofx_ctx* ctx, const float* A, int64_t lda, int64_t K, int64_t M,
                      const float* B, int ldb, int L, float* C, int prec = EOFX_PREC_F32,
                      float a_absmax = 0.f, const float* b_absmax_dev = nullptr, const AffView* aff = nullptr,
                      const float* A2 = nullptr, const float* B2 = nullptr, int sym = 0) {
  if (aff && prec != EOFX_PREC_F16X3) return set_err(ctx, EOFX_ERR_ARG, "atb: the raw view needs the f16x3 kernel");
  if (A2 && (prec != EOFX_PREC_F16X3 || aff || !B2)) return set_err(ctx, EOFX_ERR_ARG, "atb: the two-matrix form needs the f16x3 kernel");
  if (M % ATB_BM || K % ATB_KG || L % 32 || L <= 0)
    return set_err(ctx, EOFX_ERR_ARG, "atb: bad geometry M=%lld K=%lld L=%d", (long long)M,
                   (long long)K, L);
  const int bx = (int)(M / ATB_BM);
  // wide products (Gram matrices, PCA panels) in 128-column tiles of the split-fp16 kernel, then 64 / 32-column rests
  const bool wide = atb_wide(L, A2 != nullptr) && prec == EOFX_PREC_F16X3;
  if (sym && !(wide && !aff && L == M)) sym = 0;
  // wide panels: full 128-column tiles, then -- when 96 columns are left (panels of 96 / 224 columns) -- ONE partial wide tile
  // instead of a 64- and a 32-column launch (each launch reads the field once)
  const int n4 = wide ? L / 128 : 0;
  const int part4 = (wide && L - 128 * n4 == 96) ? 96 : 0;
  const int cb4 = 128 * n4 + part4;
  const int nfull = (L - cb4) / 64, rem = (L - cb4) % 64;
  const AtbPlan plan = atb_plan(M, K, L, wide);
  int best_s = plan.S;
  int64_t best_kps = plan.kps;
  ArenaScope scope(ctx);
  float* out = C;
  const int s_half = best_s;            // two-matrix form: splits [0, s_half) stream A, [s_half, 2 s_half) stream A2
  if (A2) best_s *= 2;
  if (best_s > 1) {
    out = arena_alloc<float>(ctx, (size_t)best_s * M * L);
    if (!out && A2) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (two-matrix partials)");
    if (!out) {  // no room for partials: fall back to a single split (still correct)
      best_s = 1;
      best_kps = K;
      out = C;
    }
  }
  float a_scale = 1.f;
  if (prec == EOFX_PREC_F16X3) {
    if (a_absmax > 0.f && std::isfinite(a_absmax)) {
      int e;
      (void)std::frexp(a_absmax, &e);
      a_scale = std::ldexp(1.f, 14 - e);
    }
    if (!b_absmax_dev && ldb == L && !A2) b_absmax_dev = amax_get(ctx, B);   // recorded where the panel was written
    if (!b_absmax_dev) {
      unsigned* bm = arena_alloc<unsigned>(ctx, 1);
      if (!bm) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (panel max)");
      HIPCHK(hipMemsetAsync(bm, 0, sizeof(unsigned), ctx->stream));
      const int64_t total4 = K * (L / 4);
      hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))),
                         dim3(256), 0, ctx->stream, B, K, L, (int64_t)ldb, bm);
      KCHK();
      b_absmax_dev = reinterpret_cast<const float*>(bm);
    }
  }
  // the maximum of the panel this launch produces, for the pass that will consume it (split-fp16, whole-panel outputs)
  unsigned* amax_out = (prec == EOFX_PREC_F16X3 && !sym && !wide && !A2) ? amax_new(ctx, C) : (amax_forget(ctx, C), nullptr);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (ctx->profile) {
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    HIPCHK(hipEventRecord(ev0, ctx->stream));
  }
  if (prec == EOFX_PREC_F64) {   // float64 partials (the float32 ones above are not used)
    double* outd = nullptr;
    if (best_s > 1) {
      outd = arena_alloc<double>(ctx, (size_t)best_s * M * L);
      if (!outd) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (float64 partials)");
    }
    if (nfull > 0) {
      hipLaunchKernelGGL(atb_f64_kernel<2>, dim3(bx, best_s, nfull), dim3(512), 0, ctx->stream, A, lda, B, ldb, C, outd, L, M, K,
                         best_kps, 0);
      KCHK();
    }
    if (rem) {
      hipLaunchKernelGGL(atb_f64_kernel<1>, dim3(bx, best_s, 1), dim3(512), 0, ctx->stream, A, lda, B, ldb, C, outd, L, M, K,
                         best_kps, nfull * 64);
      KCHK();
    }
    if (ctx->profile) {
      HIPCHK(hipEventRecord(ev1, ctx->stream));
      ctx->prof_events.emplace_back(ev0, ev1);
      ctx->prof_flops += 2.0 * (double)K * (double)M * (double)L;
      ctx->prof_bytes += (double)K * (double)M * 4.0 * (nfull + (rem ? 1 : 0));
    }
    if (outd) {
      const int64_t count = M * L;
      hipLaunchKernelGGL(splitk_reduce_f64_kernel, dim3((int)std::min<int64_t>((count + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                         outd, C, count, best_s);
      KCHK();
    }
    return EOFX_OK;
  }
  unsigned* amax_direct = best_s == 1 ? amax_out : nullptr;   // single split: the kernels' own epilogues take the maximum
  if (n4 > 0) {
    dim3 grid(bx, best_s, n4);
    launch_atb_variant<4>(prec, grid, ctx->stream, A, lda, B, ldb, out, L, M, K, best_kps, 0, a_scale, b_absmax_dev, aff, A2, B2, s_half, sym, amax_direct);
    KCHK();
  }
  if (part4) {
    dim3 grid(bx, best_s, 1);
    launch_atb_variant<4>(prec, grid, ctx->stream, A, lda, B, ldb, out, L, M, K, best_kps, 128 * n4, a_scale, b_absmax_dev, aff, A2, B2, s_half, sym,
                          amax_direct, part4);
    KCHK();
  }
  if (nfull > 0) {
    dim3 grid(bx, best_s, nfull);
    launch_atb_variant<2>(prec, grid, ctx->stream, A, lda, B, ldb, out, L, M, K, best_kps, cb4, a_scale, b_absmax_dev, aff, A2, B2, s_half, sym, amax_direct);
    KCHK();
  }
  if (rem) {
    dim3 grid(bx, best_s, 1);
    launch_atb_variant<1>(prec, grid, ctx->stream, A, lda, B, ldb, out, L, M, K, best_kps, cb4 + nfull * 64, a_scale, b_absmax_dev, aff, A2, B2, s_half, sym, amax_direct);
    KCHK();
  }
  if (ctx->profile) {
    HIPCHK(hipEventRecord(ev1, ctx->stream));
    ctx->prof_events.emplace_back(ev0, ev1);
    ctx->prof_flops += 2.0 * (double)K * (double)M * (double)L * (A2 ? 2 : 1);
    ctx->prof_bytes += (double)K * (double)M * 4.0 * (n4 + (part4 ? 1 : 0) + nfull + (rem ? 1 : 0)) * (A2 ? 2 : 1);
  }
  if (best_s > 1) {
    const int64_t count4 = M * L / 4;
    const int blocks = (int)std::min<int64_t>((count4 + 255) / 256, amax_out ? 2048 : 4096);   // (one atomic per workgroup)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, out, C, count4,
                       best_s, amax_out);
    KCHK();
  }
  if (sym) {
    const int nt = (int)((M + 63) / 64);
    hipLaunchKernelGGL(symmetrize_lower_kernel, dim3(nt, nt), dim3(256), 0, ctx->stream, C, M, (int64_t)L);
    KCHK();
  }
  return EOFX_OK;
}


// W[n_pad x L] = X' Y with X' = the raw field [rows x cols] (ld) through the affine map, read in place (axb_f16_kernel).
// Y: [>= round_up(cols, 64) x L] panel whose rows >= cols are zero.  L a multiple of 32.
static int launch_axb(eofx_ctx* ctx, const float* raw, int64_t ld, int64_t rows, int64_t cols, int64_t rows_pad,
                      const float* aff, int64_t aff_ld, float a_absmax, const float* Y, int L, float* W, bool masked = false,
                      const int* act = nullptr, int64_t n_act = 0) {
  const int64_t K_all = round_up(cols, AXB_KG);
  // masked matrix with an active list: the kernel walks n_act slab pairs instead of K_all / 64 (see axb_f16_kernel)
  const int64_t K = (masked && act) ? n_act * AXB_KG : K_all;
  if (K == 0) {   // nothing but masked grid points
    HIPCHK(hipMemsetAsync(W, 0, sizeof(float) * (size_t)rows_pad * L, ctx->stream));
    amax_forget(ctx, W);
    return EOFX_OK;
  }
  if (L % 32 || L <= 0 || rows_pad % AXB_BM || rows >= ((int64_t)1 << 31) || K_all * L >= ((int64_t)1 << 30) ||
      64 * ld + K_all >= ((int64_t)1 << 30))     // the kernel's 32-bit offsets
    return set_err(ctx, EOFX_ERR_ARG, "axb: bad geometry rows=%lld cols=%lld L=%d", (long long)rows, (long long)cols, L);
  const AtbPlan plan = axb_plan(rows_pad, K);
  const int rt = (int)(rows_pad / AXB_BM);
  const int nfull = L / 64, rem = L % 64;
  ArenaScope scope(ctx);
  float* out = W;
  if (plan.S > 1) {
    out = arena_alloc<float>(ctx, (size_t)plan.S * rows_pad * L);
    if (!out) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (in-place product partials)");
  }
  float a_scale = 1.f;
  if (a_absmax > 0.f && std::isfinite(a_absmax)) {
    int e;
    (void)std::frexp(a_absmax, &e);
    a_scale = std::ldexp(1.f, 14 - e);
  }
  const float* bmax = amax_get(ctx, Y);      // recorded where the panel was written
  if (!bmax) {
    unsigned* bm = arena_alloc<unsigned>(ctx, 1);
    if (!bm) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (panel max)");
    HIPCHK(hipMemsetAsync(bm, 0, sizeof(unsigned), ctx->stream));
    const int64_t total4 = K_all * (L / 4);
    hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))),
                       dim3(256), 0, ctx->stream, Y, K_all, L, (int64_t)L, bm);
    KCHK();
    bmax = reinterpret_cast<const float*>(bm);
  }
  unsigned* amax_out = plan.S > 1 ? amax_new(ctx, W) : (amax_forget(ctx, W), nullptr);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (ctx->profile) {
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    HIPCHK(hipEventRecord(ev0, ctx->stream));
  }
  const int gx = plan.S > 1 ? 8 * rt * ((plan.S + 7) / 8) : rt;
  if (ctx->axb_dma < 0) {      // 0 never, 1 by size (the default), 2 always (EOFX_AXB_DMA=1: the tests run small shapes over it)
    const char* ev = std::getenv("EOFX_AXB_DMA");
    ctx->axb_dma = ev ? (atoi(ev) != 0 ? 2 : 0) : EOFX_AXB_DMA_DEFAULT;
  }
  // The LDS-DMA kernel costs a second launch and the split pass (~20 us together) and wins 3-5 % of the kernel: it pays from
  // about 16 GB of field per pass on (config 4 and 5; measured on one box each: 10000 x 1 036 800 -2.8 %, 10000 x 129 600
  // +0.9 %, 5000 x 259 200 +4.4 %, profiles/r04_axb_dma_ab.txt)
  bool dma = ctx->axb_dma == 2 || (ctx->axb_dma == 1 && (double)rows_pad * (double)K_all >= 4.0e9);
  const int ncb = nfull + (rem ? 1 : 0);
  if (dma) {            // the panel's fp16 planes: one grow-only buffer per context (K_all x 64 ncb x 4 bytes)
    const size_t need = (size_t)ncb * (size_t)(K_all / AXB_KG) * AXB_PAIR_BYTES;
    if (ctx->axb_planes_bytes < need) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (ctx->axb_planes) (void)hipFree(ctx->axb_planes);
      ctx->axb_planes = nullptr;
      ctx->axb_planes_bytes = 0;
      if (hipMalloc((void**)&ctx->axb_planes, need) != hipSuccess) {
        (void)hipGetLastError();
        ctx->axb_planes = nullptr;
        dma = false;    // no room for the planes: the register path needs none (same bits)
      } else {
        ctx->axb_planes_bytes = need;
      }
    }
  }
  if (dma) {   // the B panel as fp16 planes, moved by LDS-DMA (eofx_axb_dma.hpp); same bits in W
    hipLaunchKernelGGL(axb_bsplit_kernel, dim3((unsigned)((K_all + 31) / 32), ncb), dim3(256), 0, ctx->stream, Y, L, L, K_all, bmax,
                       ctx->axb_planes);
    KCHK();
    const int64_t pairs_all = K_all / AXB_KG;
    if (nfull > 0) {
      if (masked)
        hipLaunchKernelGGL((axb_f16_dma_kernel<4, 0, true>), dim3(gx, nfull), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff,
                           aff_ld, ctx->axb_planes, pairs_all, out, L, rows_pad, K, plan.kps, plan.S, rt, 0, a_scale, bmax, act);
      else
        hipLaunchKernelGGL((axb_f16_dma_kernel<4, 0, false>), dim3(gx, nfull), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff,
                           aff_ld, ctx->axb_planes, pairs_all, out, L, rows_pad, K, plan.kps, plan.S, rt, 0, a_scale, bmax,
                           (const int*)nullptr);
      KCHK();
    }
    if (rem) {
      if (masked)
        hipLaunchKernelGGL((axb_f16_dma_kernel<2, 0, true>), dim3(gx, 1), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff,
                           aff_ld, ctx->axb_planes, pairs_all, out, L, rows_pad, K, plan.kps, plan.S, rt, nfull * 64, a_scale, bmax, act);
      else
        hipLaunchKernelGGL((axb_f16_dma_kernel<2, 0, false>), dim3(gx, 1), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff,
                           aff_ld, ctx->axb_planes, pairs_all, out, L, rows_pad, K, plan.kps, plan.S, rt, nfull * 64, a_scale, bmax,
                           (const int*)nullptr);
      KCHK();
    }
  } else if (nfull > 0) {
    if (masked)
      hipLaunchKernelGGL((axb_f16_kernel<4, 0, true>), dim3(gx, nfull), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff, aff_ld, Y,
                         L, out, L, rows_pad, K, plan.kps, plan.S, rt, 0, a_scale, bmax, act);
    else
      hipLaunchKernelGGL(axb_f16_kernel<4>, dim3(gx, nfull), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff, aff_ld, Y, L, out,
                         L, rows_pad, K, plan.kps, plan.S, rt, 0, a_scale, bmax);
    KCHK();
  }
  if (!dma && rem) {
    if (masked)
      hipLaunchKernelGGL((axb_f16_kernel<2, 0, true>), dim3(gx, 1), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff, aff_ld, Y, L,
                         out, L, rows_pad, K, plan.kps, plan.S, rt, nfull * 64, a_scale, bmax, act);
    else
      hipLaunchKernelGGL(axb_f16_kernel<2>, dim3(gx, 1), dim3(256), 0, ctx->stream, raw, ld, (int)rows, cols, aff, aff_ld, Y, L, out,
                         L, rows_pad, K, plan.kps, plan.S, rt, nfull * 64, a_scale, bmax);
    KCHK();
  }
  if (ctx->profile) {
    HIPCHK(hipEventRecord(ev1, ctx->stream));
    ctx->prof_events.emplace_back(ev0, ev1, 1);
    ctx->prof_flops += 2.0 * (double)K * (double)rows_pad * (double)L;
    ctx->prof_bytes += (double)K * (double)rows * 4.0 * (nfull + (rem ? 1 : 0));
  }
  if (plan.S > 1) {
    const int64_t count4 = rows_pad * L / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<int64_t>((count4 + 255) / 256, amax_out ? 2048 : 4096)), dim3(256), 0,
                       ctx->stream, out, W, count4, plan.S, amax_out);
    KCHK();
  }
  return EOFX_OK;
}

// number of row-strided partial Gram matrices: >= 4 slabs of 32 rows per workgroup for the narrow
// (sketch-width) panels; wide panels already expose (L/64)^2 sub-blocks, so fewer partials (<= 64 MB)
static int gram_parts(int64_t rows, int L) {
  // workgroups along the rows; every workgroup writes one partial (L x L float64):
  // >= 8 k-steps (32 rows) per wave, <= 128 MB of partials (a 130 000 x 1536 panel, 300 sub-blocks: 3 row parts 8.9 ms,
  // 6 .. 24 parts 7.1 ms -- tools/wide_small_probe.py)
  // (one workgroup per CU: the products run at the fp64 MFMA rate either way -- ~45 TFLOP/s here -- and every further
  // partial is 8 L^2 bytes written and read again: 194 / 210 / 243 us for 256 / 512 / 1024 partials of a 1M x 64 panel)
  const int64_t by_rows = std::min<int64_t>((rows + 127) / 128, 256);
  const int64_t by_mem = ((int64_t)128 << 20) / ((int64_t)L * L * 8);
  if (const char* ev = std::getenv("EOFX_GRAM_PARTS")) return std::max(1, atoi(ev));   // tuning hook (tools/small_kernel_probe.py)
  return (int)std::max<int64_t>(1, std::min(by_rows, std::max<int64_t>(by_mem, 1)));
}

static int launch_gram(eofx_ctx* ctx, const float* P, int64_t rows, int L, double* G) {
  const int nb = (L + 63) / 64;
  const int nbx = gram_parts(rows, L);
  ArenaScope scope(ctx);
  ARENA(double, part, (size_t)nbx * L * L);
  hipLaunchKernelGGL(gram_mfma_kernel, dim3(nbx, nb * nb), dim3(256), 0, ctx->stream, P, rows, L, part);
  KCHK();
  const int64_t count = (int64_t)L * L;
  hipLaunchKernelGGL(f64_reduce_kernel, dim3((int)((count + 63) / 64)), dim3(256), 0, ctx->stream,
                     part, G, count, nbx);
  KCHK();
  return EOFX_OK;
}
static int launch_matmul(eofx_ctx* ctx, const float* P, int64_t rows, int L, const double* Mx, int Lo,
                         float* out, bool upper = false) {
  int KW = (int)std::min<int64_t>(round_up(L, 64), 256);            // rows of Mx held in LDS at a time
  if (L > 256) {                                                    // windowed form: tuning hook (tools/wide_small_probe.py)
    static const int kw_env = std::getenv("EOFX_PMM_KW") ? atoi(std::getenv("EOFX_PMM_KW")) : 0;
    if (kw_env == 128 || kw_env == 256) KW = kw_env;
  }
  const size_t smem = (size_t)KW * PMM_LD * sizeof(double);         // 33 .. 132 KB
  // opt in to more than 64 KB of dynamic LDS when a wide panel asks for it.  The attribute belongs to the (device, function) pair:
  // set whenever needed (a cheap host call), not remembered in a process-global -- contexts on several devices or threads
  if (smem > 64 * 1024)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_matmul_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int64_t units = (rows + 127) / 128;        // 4 waves x 32 rows
  const int64_t gx = L <= KW ? std::min<int64_t>(units, 1024) : units;   // windowed form: one group per wave
  dim3 grid((int)std::max<int64_t>(1, gx), (Lo + 63) / 64);
  hipLaunchKernelGGL(panel_matmul_kernel<false>, grid, dim3(256), smem, ctx->stream, P, rows, L, Mx, Lo, out, KW, amax_new(ctx, out),
                     (int64_t)0, (int64_t)0, 1, (const float*)nullptr, upper ? 1 : 0);
  KCHK();
  return EOFX_OK;
}
// out [rows x Lo] = (sub -) sum over the 64-column chunks c of P: chunk c at P + (c / cpb) * slab + (c % cpb) * 64, row stride ldp
// (see panel_matmul_kernel<true>); L = 64 x chunks.  out must not alias P (it may alias sub).
static int launch_matmul_gen(eofx_ctx* ctx, const float* P, int64_t ldp, int64_t slab, int cpb, int64_t rows, int L, const double* Mx, int Lo,
                             const float* sub, float* out) {
  const int KW = (int)std::min<int64_t>(round_up(L, 64), 256);
  const size_t smem = (size_t)KW * PMM_LD * sizeof(double);
  if (smem > 64 * 1024)      // (per device and function: set whenever needed, see launch_matmul)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_matmul_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int64_t units = (rows + 127) / 128;
  const int64_t gx = L <= KW ? std::min<int64_t>(units, 1024) : units;
  dim3 grid((int)std::max<int64_t>(1, gx), (Lo + 63) / 64);
  hipLaunchKernelGGL(panel_matmul_kernel<true>, grid, dim3(256), smem, ctx->stream, P, rows, L, Mx, Lo, out, KW, amax_new(ctx, out), ldp,
                     slab, cpb, sub);
  KCHK();
  return EOFX_OK;
}
// C [La x Lb] (float64, device) = Pa^T Pb over `rows` rows; La, Lb multiples of 64
static int launch_xgram(eofx_ctx* ctx, const float* Pa, int64_t lda, int La, const float* Pb, int64_t ldb, int Lb, int64_t rows, double* C) {
  const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 127) / 128, std::max<int64_t>(1, ((int64_t)32 << 20) / ((int64_t)La * Lb * 8))));
  ArenaScope scope(ctx);
  ARENA(double, part, (size_t)nbx * La * Lb);
  hipLaunchKernelGGL(xgram_mfma_kernel, dim3(nbx, (La / 64) * (Lb / 64)), dim3(256), 0, ctx->stream, Pa, lda, Pb, ldb, rows, La, Lb, part);
  KCHK();
  const int64_t count = (int64_t)La * Lb;
  hipLaunchKernelGGL(f64_reduce_kernel, dim3((int)((count + 63) / 64)), dim3(256), 0, ctx->stream, part, C, count, nbx);
  KCHK();
  return EOFX_OK;
}

// host fallback of chol_rinv for sketches wider than one wavefront's 64 columns: same algorithm,
// same dependent-column rule, float64.  G, Rinv are L x L row-major; only the l x l block is used.
static void host_chol_rinv(const double* G, int L, int l, double* Rinv, double tol) {
  std::vector<double> A((size_t)l * l, 0.0), X((size_t)l * l, 0.0), d0(l);
  std::vector<char> dead(l, 0);
  for (int r = 0; r < l; ++r) {
    d0[r] = G[(size_t)r * L + r];
    for (int c = r; c < l; ++c) A[(size_t)r * l + c] = G[(size_t)r * L + c];
  }
  for (int j = 0; j < l; ++j) {
    const double d = A[(size_t)j * l + j];
    const bool dj = !(d > tol * d0[j]) || !(d0[j] > 0.0);
    dead[j] = dj;
    const double rjj = dj ? 1.0 : std::sqrt(d);
    const double piv = dj ? 0.0 : 1.0 / rjj;
    A[(size_t)j * l + j] = rjj;
    double* rowj = &A[(size_t)j * l];
    for (int c = j + 1; c < l; ++c) rowj[c] *= piv;
    for (int r = j + 1; r < l; ++r) {
      const double f = rowj[r];
      if (f == 0.0) continue;
      double* rowr = &A[(size_t)r * l];
      for (int c = r; c < l; ++c) rowr[c] -= f * rowj[c];
    }
  }
  for (int c = 0; c < l; ++c) {
    if (dead[c]) continue;
    X[(size_t)c * l + c] = 1.0 / A[(size_t)c * l + c];
    for (int r = c - 1; r >= 0; --r) {
      double sum = 0.0;
      for (int t = r + 1; t <= c; ++t) sum += A[(size_t)r * l + t] * X[(size_t)t * l + c];
      X[(size_t)r * l + c] = -sum / A[(size_t)r * l + r];
    }
  }
  for (int r = 0; r < L; ++r)
    for (int c = 0; c < L; ++c) Rinv[(size_t)r * L + c] = (r < l && c < l) ? X[(size_t)r * l + c] : 0.0;
}

// out = P R^-1 with G = R^T R (leading l x l block)
static int launch_rinv(eofx_ctx* ctx, const double* G, int L, int l, double* Rinv);
static size_t rinv_blocked_bytes(int l);
static bool matmul_nt_ok(int64_t rows, int L, int Lo);
static size_t matmul_nt_scratch(eofx_ctx* ctx, int64_t rows, int L, int Lo);
static int launch_matmul_nt(eofx_ctx* ctx, const float* P, int64_t rows, int L, const double* Mx, int Lo, float* out);
static int launch_cholqr(eofx_ctx* ctx, const float* P, int64_t rows, int L, int l, const double* G,
                         float* out) {
  ArenaScope scope(ctx);
  ARENA(double, Rinv, (size_t)L * L);
  CHK(launch_rinv(ctx, G, L, l, Rinv));
  // (Also for wide sketches the product with R^-1 stays in the float64 kernel: through fp16 planes -- 22-bit operands against the
  // matrix' largest entry, and R^-1 spans orders of magnitude -- Q lost orthonormality, 1.5e-6 instead of 4e-9 at 1510 columns,
  // for 0.75 ms per factorisation.)
  return launch_matmul(ctx, P, rows, L, Rinv, L, out, /*upper=*/true);      // (R^-1 is upper triangular, zero below)
}

static int launch_colminmax(eofx_ctx* ctx, const float* P, int64_t rows, int L, float* mx, float* mn) {
  const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 127) / 128, 1024));
  ArenaScope scope(ctx);
  ARENA(float, pmx, (size_t)nparts * L);
  ARENA(float, pmn, (size_t)nparts * L);
  hipLaunchKernelGGL(colminmax_part_kernel, dim3(nparts, (L + 63) / 64), dim3(256), 0, ctx->stream, P,
                     rows, L, pmx, pmn);
  KCHK();
  hipLaunchKernelGGL(colminmax_final_kernel, dim3((L + 63) / 64), dim3(256), 0, ctx->stream, pmx, pmn,
                     nparts, L, mx, mn);
  KCHK();
  return EOFX_OK;
}

// dense [rows x k] export to a host|device destination, optional column signs (host doubles)
static int export_panel(eofx_ctx* ctx, const float* P, int64_t rows, int L, int k, const double* sign,
                        float* dst) {
  if (!dst) return EOFX_OK;
  ArenaScope scope(ctx);
  double* dsign = nullptr;
  if (sign) {
    dsign = arena_alloc<double>(ctx, k);
    if (!dsign) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (export sign)");
    CHK(copy_in(ctx, dsign, sign, sizeof(double) * k));
  }
  float* tmp = dst;
  const bool dev = is_device_ptr(dst);
  if (!dev) {
    tmp = arena_alloc<float>(ctx, (size_t)rows * k);
    if (!tmp) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (export staging)");
  }
  const int64_t total = rows * k;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(panel_export_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P, rows, L, k, dsign,
                     tmp);
  KCHK();
  if (!dev) {
    CHK(copy_out(ctx, dst, tmp, sizeof(float) * total));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return EOFX_OK;
}

static int import_panel(eofx_ctx* ctx, const float* src, int64_t rows, int l, float* P, int64_t rows_pad,
                        int L) {
  ArenaScope scope(ctx);
  const float* dsrc = src;
  if (!is_device_ptr(src)) {
    float* tmp = arena_alloc<float>(ctx, (size_t)rows * l);
    if (!tmp) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (import staging)");
    CHK(copy_in(ctx, tmp, src, sizeof(float) * rows * l));
    dsrc = tmp;
  }
  const int64_t total = rows_pad * (L / 4);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(panel_import_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dsrc, rows, l, P,
                     rows_pad, L, amax_new(ctx, P));
  KCHK();
  if (dsrc != src) HIPCHK(hipStreamSynchronize(ctx->stream));  // staging is released on return
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// resident matrix
// ------------------------------------------------------------------------------------
static void* pool_take(eofx_ctx* ctx, size_t bytes) {
  for (size_t i = 0; i < ctx->pool.size(); ++i)
    if (ctx->pool[i].second == bytes) {
      void* p = ctx->pool[i].first;
      ctx->pool_bytes -= bytes;
      ctx->pool.erase(ctx->pool.begin() + i);
      return p;
    }
  return nullptr;
}
static void pool_trim(eofx_ctx* ctx) {
  if (ctx->pool.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->pool) (void)hipFree(e.first);
  ctx->pool.clear();
  ctx->pool_bytes = 0;
}
static void pool_give(eofx_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (!ctx || bytes > ctx->pool_cap) {
    (void)hipFree(p);
    return;
  }
  // (exact-size reuse: a run of differently shaped fits would otherwise pile up buffers nobody asks for again -- 3.5 MB per fit
  // in tools/soak_probe.py, up to the byte cap; the entry cap keeps what one or two repeated shapes need)
  // Eviction is BATCHED: a workload whose steady state cycles through more shapes than the pool holds would
  // otherwise pay a stream synchronisation (and hipFree's own device synchronisation) on every give; when a cap is hit the
  // oldest quarter of the entries goes in one sweep behind ONE synchronisation, so at most one give in twelve stalls.
  constexpr size_t POOL_MAX_ENTRIES = 48, POOL_EVICT_BATCH = 12;
  if (!ctx->pool.empty() && (ctx->pool_bytes + bytes > ctx->pool_cap || ctx->pool.size() >= POOL_MAX_ENTRIES)) {
    (void)hipStreamSynchronize(ctx->stream);
    size_t evicted = 0;
    while (!ctx->pool.empty() && (evicted < POOL_EVICT_BATCH || ctx->pool_bytes + bytes > ctx->pool_cap)) {
      (void)hipFree(ctx->pool.front().first);
      ctx->pool_bytes -= ctx->pool.front().second;
      ctx->pool.erase(ctx->pool.begin());
      ++evicted;
    }
  }
  ctx->pool.emplace_back(p, bytes);
  ctx->pool_bytes += bytes;
}
static hipError_t pool_malloc(eofx_ctx* ctx, void** out, size_t bytes) {
  *out = pool_take(ctx, bytes);
  if (*out) return hipSuccess;
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess && !ctx->pool.empty()) {  // out of memory: drop the cache and retry
    (void)hipGetLastError();
    pool_trim(ctx);
    e = hipMalloc(out, bytes);
  }
  return e;
}
extern "C" int eofx_ctx_trim(eofx_ctx* ctx) {
  if (!ctx) return EOFX_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  pool_trim(ctx);
  if (ctx->axb_planes) {      // the panel planes of the in-place product are a cache as well
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->axb_planes);
    ctx->axb_planes = nullptr;
    ctx->axb_planes_bytes = 0;
  }
  return EOFX_OK;
}

static int mat_alloc(eofx_ctx* ctx, int64_t n, int64_t p, eofx_mat** out, bool want_x = true, bool want_xt = true) {
  eofx_mat* m = new eofx_mat();
  m->n = n;
  m->p = p;
  m->n_pad = round_up(n, ATB_BM);
  m->p_pad = round_up(p, ATB_BM);
  const size_t bytes = (size_t)m->n_pad * m->p_pad * sizeof(float);
  hipError_t e = want_x ? pool_malloc(ctx, (void**)&m->X, bytes) : hipSuccess;
  if (e == hipSuccess && want_xt) e = pool_malloc(ctx, (void**)&m->Xt, bytes);
  if (e == hipSuccess) e = pool_malloc(ctx, (void**)&m->absmax_dev, 256);
  if (e == hipSuccess) e = hipMemsetAsync(m->absmax_dev, 0, 256, ctx->stream);
  if (e != hipSuccess) {
    if (m->X) (void)hipFree(m->X);
    if (m->Xt) (void)hipFree(m->Xt);
    delete m;
    (void)hipGetLastError();
    return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate resident matrix %lld x %lld (2 x %.2f GB): %s",
                   (long long)n, (long long)p, bytes / 1e9, hipGetErrorString(e));
  }
  *out = m;
  return EOFX_OK;
}

extern "C" int eofx_mat_destroy(eofx_ctx* ctx, eofx_mat* m) {
  if (!m) return EOFX_OK;
  const size_t bytes = (size_t)m->n_pad * m->p_pad * sizeof(float);
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    // launch_apply queues a copy INTO this struct (m->absmax): nothing may still be in flight when it goes away
    (void)hipStreamSynchronize(ctx->stream);
    // same-stream reuse is ordered; nothing else touches these buffers
    if (m->X) pool_give(ctx, m->X, bytes);
    if (m->Xt) pool_give(ctx, m->Xt, bytes);
    if (m->act) pool_give(ctx, m->act, sizeof(int) * (size_t)(round_up(m->p, AXB_KG) / AXB_KG));
    pool_give(ctx, m->absmax_dev, 256);
    if (m->raw_owned) pool_give(ctx, m->raw_owned, m->raw_owned_bytes);
    if (m->rawT) pool_give(ctx, m->rawT, bytes);
  } else {
    if (m->X) (void)hipFree(m->X);
    if (m->Xt) (void)hipFree(m->Xt);
    if (m->act) (void)hipFree(m->act);
    if (m->raw_owned) (void)hipFree(m->raw_owned);
    if (m->rawT) (void)hipFree(m->rawT);
  }
  if (m->aff) {
    const size_t abytes = sizeof(float) * 3 * (size_t)m->p_pad;
    if (ctx) pool_give(ctx, m->aff, abytes);
    else (void)hipFree(m->aff);
  }
  delete m;
  return EOFX_OK;
}
extern "C" int eofx_mat_shape(const eofx_mat* m, int64_t* n, int64_t* p, int64_t* n_pad, int64_t* p_pad) {
  if (!m) return EOFX_ERR_ARG;
  if (n) *n = m->n;
  if (p) *p = m->p;
  if (n_pad) *n_pad = m->n_pad;
  if (p_pad) *p_pad = m->p_pad;
  return EOFX_OK;
}

// stage a host matrix on the device (hipMalloc'ed, caller frees); device input passes through
struct Staged {
  const float* dev = nullptr;
  float* owned = nullptr;
  ~Staged() {
    if (owned) (void)hipFree(owned);
  }
};
static int stage_input(eofx_ctx* ctx, const float* X, size_t count, Staged& st) {
  if (is_device_ptr(X)) {
    st.dev = X;
    return EOFX_OK;
  }
  HIPCHK(hipMalloc((void**)&st.owned, count * sizeof(float)));
  HIPCHK(hipMemcpyAsync(st.owned, X, count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  st.dev = st.owned;
  return EOFX_OK;
}

// raw mode over a host field: the staged device copy lives as long as the matrix that reads it
static void adopt_staged(eofx_mat** out, Staged& st, size_t bytes) {
  if (out && *out && (*out)->raw && (*out)->raw == st.owned) {
    (*out)->raw_owned = st.owned;
    (*out)->raw_owned_bytes = bytes;
    st.owned = nullptr;
  }
}

// absmax_src: device scalar holding max |transformed value| when the column statistics already know it
// (preprocess path); nullptr -> measured with one extra read of the written matrix.
static int launch_apply(eofx_ctx* ctx, const float* Xsrc, int64_t ld_src, const int64_t* row_map,
                        const int64_t* col_map, const double* shift, const double* scale, eofx_mat* m,
                        int* nan_flag, const unsigned* absmax_src = nullptr) {
  dim3 grid((int)(m->p_pad / 64), (int)(m->n_pad / 64));
  const bool vec = !col_map && (ld_src % 4 == 0) && ((uintptr_t)Xsrc % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(apply_kernel<true>, grid, dim3(256), 0, ctx->stream, Xsrc, ld_src, row_map, col_map,
                       shift, scale, m->n, m->p, m->X, m->p_pad, m->Xt, m->n_pad, nan_flag, (const float*)nullptr, (int64_t)0);
  else
    hipLaunchKernelGGL(apply_kernel<false>, grid, dim3(256), 0, ctx->stream, Xsrc, ld_src, row_map, col_map,
                       shift, scale, m->n, m->p, m->X, m->p_pad, m->Xt, m->n_pad, nan_flag, (const float*)nullptr, (int64_t)0);
  KCHK();
  if (absmax_src) {
    HIPCHK(hipMemcpyAsync(m->absmax_dev, absmax_src, sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    const int64_t total4 = m->n_pad * (m->p_pad / 4);   // same elements in either layout
    hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))), dim3(256), 0,
                       ctx->stream, m->Xt, m->p_pad, (int)m->n_pad, m->n_pad, m->absmax_dev);
    KCHK();
  }
  // host copy of max |x| (callers synchronise the stream before using the matrix)
  HIPCHK(hipMemcpyAsync(&m->absmax, m->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  return EOFX_OK;
}

// The sample-contiguous layout of an in-place matrix, built on demand: the raw field through the same affine map the
// streaming kernels apply (apply_kernel with the packed float triples), or a tiled transpose of X.
static int ensure_Xt(eofx_ctx* ctx, const eofx_mat* cm) {
  eofx_mat* m = const_cast<eofx_mat*>(cm);
  if (m->Xt) return EOFX_OK;
  if (!m->X && !(m->raw && m->aff)) return set_err(ctx, EOFX_ERR_ARG, "matrix holds no data (raw field released?)");
  const size_t bytes = (size_t)m->n_pad * m->p_pad * sizeof(float);
  if (pool_malloc(ctx, (void**)&m->Xt, bytes) != hipSuccess) {
    (void)hipGetLastError();
    m->Xt = nullptr;
    return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the sample-contiguous layout (%.2f GB)", bytes / 1e9);
  }
  if (m->X) {
    hipLaunchKernelGGL(transpose_kernel, dim3((int)(m->p_pad / 64), (int)(m->n_pad / 64)), dim3(256), 0, ctx->stream, m->X,
                       m->p_pad, m->Xt, m->n_pad);
  } else {
    hipLaunchKernelGGL(apply_kernel<true>, dim3((int)(m->p_pad / 64), (int)(m->n_pad / 64)), dim3(256), 0, ctx->stream, m->raw,
                       m->raw_ld, (const int64_t*)nullptr, (const int64_t*)nullptr, (const double*)nullptr, (const double*)nullptr,
                       m->n, m->p, (float*)nullptr, m->p_pad, m->Xt, m->n_pad, (int*)nullptr, (const float*)m->aff, m->p_pad);
  }
  KCHK();
  return EOFX_OK;
}
// The feature-contiguous layout of a raw-mode matrix, built on demand from the sample-contiguous one (tiled transpose).
static int ensure_X(eofx_ctx* ctx, const eofx_mat* cm) {
  eofx_mat* m = const_cast<eofx_mat*>(cm);
  if (m->X) return EOFX_OK;
  CHK(ensure_Xt(ctx, m));
  const size_t bytes = (size_t)m->n_pad * m->p_pad * sizeof(float);
  if (pool_malloc(ctx, (void**)&m->X, bytes) != hipSuccess) {
    (void)hipGetLastError();
    m->X = nullptr;
    return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the feature-contiguous layout (%.2f GB)", bytes / 1e9);
  }
  hipLaunchKernelGGL(transpose_kernel, dim3((int)(m->n_pad / 64), (int)(m->p_pad / 64)), dim3(256), 0, ctx->stream, m->Xt,
                     m->n_pad, m->X, m->p_pad);
  KCHK();
  return EOFX_OK;
}
extern "C" int eofx_ctx_set_layout(eofx_ctx* ctx, int keep_raw) {
  if (!ctx || keep_raw < 0 || keep_raw > 3) return EOFX_ERR_ARG;
  ctx->allow_masked = keep_raw == 3;
  if (keep_raw == 3) keep_raw = 2;
  ctx->keep_raw = keep_raw;
  return EOFX_OK;
}
extern "C" int eofx_ctx_set_sample_raw(eofx_ctx* ctx, int on) {
  if (!ctx) return EOFX_ERR_ARG;
  ctx->want_rawT = on != 0;
  return EOFX_OK;
}
extern "C" int eofx_mat_release_raw(eofx_ctx* ctx, eofx_mat* m) {
  if (!ctx || !m) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (m->raw && !m->Xt && !m->X) CHK(ensure_Xt(ctx, m));   // in place: the field is the only copy -- materialise first
  HIPCHK(hipStreamSynchronize(ctx->stream));   // passes still reading the raw field
  if (m->raw_owned) pool_give(ctx, m->raw_owned, m->raw_owned_bytes);
  m->raw_owned = nullptr;
  m->raw_owned_bytes = 0;
  m->raw = nullptr;
  return EOFX_OK;
}
// The sample-contiguous layout of a matrix, built ahead of the passes that would use it (an in-place matrix serves every
// pass from the raw field; repeated decompositions on the same matrix -- bootstrap members -- run their X Y passes
// 13 % faster over this layout).  only_if_room: skip, and report built = 0, unless HBM holds one more copy of the field
// with 8 GB to spare.  Masked in-place matrices keep their single layout.
extern "C" int eofx_mat_ensure_sample_layout(eofx_ctx* ctx, eofx_mat* m, int only_if_room, int* built) {
  if (!ctx || !m) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (built) *built = m->Xt != nullptr;
  if (m->Xt || m->masked) return EOFX_OK;
  if (only_if_room) {
    size_t free_b = 0, total_b = 0;
    const size_t need_b = (size_t)m->n_pad * m->p_pad * sizeof(float);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      return EOFX_OK;
    }
    if (free_b + ctx->pool_bytes <= need_b + need_b / 2 + ((size_t)8 << 30)) return EOFX_OK;
  }
  CHK(ensure_Xt(ctx, m));
  if (built) *built = 1;
  return EOFX_OK;
}
// Drop the sample-contiguous layout of a matrix that can rebuild it (in-place / raw mode: the raw field and its map stay).
// The memory goes back to the context's pool.  A matrix whose only data is that layout keeps it (EOFX_OK, nothing done).
extern "C" int eofx_mat_release_sample_layout(eofx_ctx* ctx, eofx_mat* m) {
  if (!ctx || !m) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (!m->Xt || !(m->X || (m->raw && m->aff))) return EOFX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));   // kernels that read it may still be in flight
  pool_give(ctx, m->Xt, (size_t)m->n_pad * m->p_pad * sizeof(float));
  m->Xt = nullptr;
  return EOFX_OK;
}
extern "C" int eofx_mat_masked(const eofx_mat* m, int* masked, int64_t* p_valid) {
  if (!m) return EOFX_ERR_ARG;
  if (masked) *masked = m->masked ? 1 : 0;
  if (p_valid) *p_valid = m->masked ? m->p_valid : m->p;
  return EOFX_OK;
}
extern "C" int eofx_mat_layout(const eofx_mat* m, int* has_x, int* has_raw) {
  if (!m) return EOFX_ERR_ARG;
  if (has_x) *has_x = (m->X != nullptr ? 1 : 0) | (m->Xt != nullptr ? 2 : 0);   // bit 0: feature-contiguous, bit 1: sample-contiguous
  if (has_raw) *has_raw = m->raw != nullptr;
  return EOFX_OK;
}

extern "C" int eofx_mat_from_dense_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t p, int64_t ld,
                                       eofx_mat** out) {
  if (!ctx || !X || !out || n <= 0 || p <= 0 || ld < p) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  Staged st;
  CHK(stage_input(ctx, X, (size_t)n * ld, st));
  eofx_mat* m = nullptr;
  CHK(mat_alloc(ctx, n, p, &m));
  CHK(arena_reserve(ctx, 1 << 20));
  ArenaScope scope(ctx);
  ARENA(int, flag, 1);
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
  int rc = launch_apply(ctx, st.dev, ld, nullptr, nullptr, nullptr, nullptr, m, flag);
  if (rc == EOFX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
  if (rc != EOFX_OK) {
    eofx_mat_destroy(ctx, m);
    return rc;
  }
  *out = m;
  return EOFX_OK;
}

extern "C" int eofx_mat_download_f32(eofx_ctx* ctx, const eofx_mat* m, float* dst) {
  if (!ctx || !m || !dst) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int64_t total = m->n * m->p;
  const bool dev = is_device_ptr(dst);
  float* tmp = dst;
  if (!dev) HIPCHK(hipMalloc((void**)&tmp, total * sizeof(float)));
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  CHK(ensure_X(ctx, m));
  hipLaunchKernelGGL(mat_download_kernel, dim3(blocks), dim3(256), 0, ctx->stream, m->X, m->p_pad, m->n,
                     m->p, tmp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !dev) e = hipMemcpyAsync(dst, tmp, total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (!dev) (void)hipFree(tmp);
  if (e != hipSuccess) return set_err(ctx, EOFX_ERR_HIP, "download failed: %s", hipGetErrorString(e));
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// fused preprocessor
// ------------------------------------------------------------------------------------
struct PreState {  // device arrays of length P
  int* cnt;
  double *mean, *stdv, *shift, *scale, *m2;
  unsigned* absmax;  // device scalar: max |transformed value| (float bits)
  float *vmin = nullptr, *vmax = nullptr;   // optional: per-feature extremes of the data (eofx_apply_f32)
};

static int run_colstats(eofx_ctx* ctx, const float* Xd, int64_t n, int64_t P, int center, int standardize,
                        const double* w_dev, PreState& ps, int64_t ld = 0, const int64_t* row_map = nullptr) {
  if (ld == 0) ld = P;
  const bool vec4 = P % 4 == 0 && ld % 4 == 0 && ((uintptr_t)Xd % 16) == 0;     // four features per thread, 16-byte loads
  const int gxs = (int)((P + 255) / 256);                    // workgroups of the finalize kernel (one thread per feature)
  const int gx = vec4 ? (int)((P / 4 + 255) / 256) : gxs;
  int64_t RS = std::max<int64_t>(1, (2048 + gx - 1) / gx);
  RS = std::min<int64_t>(RS, std::max<int64_t>(1, n / 64));
  int64_t rps = (n + RS - 1) / RS;
  if (ctx->want_rawT && ctx->keep_raw == 2 && !row_map) rps = round_up(rps, 64);   // (the transposing variant moves 64 x 64 tiles)
  RS = (n + rps - 1) / rps;
  ArenaScope scope(ctx);
  ARENA(int, cnt_p, (size_t)RS * P);
  ARENA(double, sum_p, (size_t)RS * P);
  ARENA(double, sq_p, (size_t)RS * P);
  ARENA(float, mn_p, (size_t)RS * P);
  ARENA(float, mx_p, (size_t)RS * P);
  // the Hilbert stage follows (eofx_ctx_set_sample_raw) and its one-kernel route takes this length two features at a time:
  // the pass also writes the raw field in the sample-contiguous layout (instead of a separate transposing copy behind the statistics pass)
  if (vec4 && !row_map && ctx->want_rawT && ctx->keep_raw == 2 && !ctx->pending_rawT && round_up(n, ATB_BM) <= 8192 && rps % 64 == 0) {
    const int64_t n_pad_t = round_up(n, ATB_BM), p_pad_t = round_up(P, ATB_BM);
    const size_t tb = (size_t)n_pad_t * p_pad_t * sizeof(float);
    if (pool_malloc(ctx, (void**)&ctx->pending_rawT, tb) == hipSuccess) ctx->pending_rawT_bytes = tb;
    else {
      (void)hipGetLastError();
      ctx->pending_rawT = nullptr;
    }
  }
  if (vec4 && ctx->pending_rawT && !row_map && ctx->want_rawT)
    hipLaunchKernelGGL(colstats_tr_kernel, dim3((unsigned)((P + 63) / 64), (int)RS), dim3(256), 0, ctx->stream, Xd, n, P, ld, rps, cnt_p,
                       sum_p, sq_p, mn_p, mx_p, ctx->pending_rawT, round_up(n, ATB_BM));
  else if (vec4)
    hipLaunchKernelGGL(colstats4_kernel, dim3(gx, (int)RS), dim3(256), 0, ctx->stream, Xd, n, P, ld, row_map, rps,
                       cnt_p, sum_p, sq_p, mn_p, mx_p);
  else
    hipLaunchKernelGGL(colstats_kernel, dim3(gx, (int)RS), dim3(256), 0, ctx->stream, Xd, n, P, ld, row_map, rps,
                       cnt_p, sum_p, sq_p, mn_p, mx_p);
  KCHK();
  HIPCHK(hipMemsetAsync(ps.absmax, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(colstats_finalize_kernel, dim3(gxs), dim3(256), 0, ctx->stream, cnt_p, sum_p, sq_p, mn_p, mx_p,
                     (int)RS, P, center, standardize, w_dev, (double)1.1920928955078125e-07, ps.cnt,
                     ps.mean, ps.stdv, ps.shift, ps.scale, ps.m2, ps.absmax, ps.vmin, ps.vmax);
  KCHK();
  return EOFX_OK;
}

struct FeatSummary {
  int64_t pv = 0;      // features with at least one non-NaN value
  int cmin = 0, cmax = 0;
  double tv = 0.0;     // total variance of the transformed valid features (ddof = 1)
};
static int run_feature_summary(eofx_ctx* ctx, const PreState& ps, int64_t P, FeatSummary& fs) {
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((P + 4095) / 4096, 256));
  ArenaScope scope(ctx);
  ARENA(double, tvp, nb);
  ARENA(int, ip, 3 * nb);
  hipLaunchKernelGGL(feature_summary_kernel, dim3(nb), dim3(256), 0, ctx->stream, ps.cnt, ps.m2, ps.scale, P, tvp, ip);
  KCHK();
  std::vector<double> htv(nb);
  std::vector<int> hip_(3 * nb);
  HIPCHK(hipMemcpyAsync(htv.data(), tvp, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(hip_.data(), ip, sizeof(int) * 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  fs = FeatSummary();
  fs.cmin = INT32_MAX;
  for (int b = 0; b < nb; ++b) {   // fixed order
    fs.tv += htv[b];
    fs.pv += hip_[3 * b];
    fs.cmin = std::min(fs.cmin, hip_[3 * b + 1]);
    fs.cmax = std::max(fs.cmax, hip_[3 * b + 2]);
  }
  return EOFX_OK;
}

static size_t colstats_scratch(int64_t n, int64_t P) {
  const int64_t gx = std::max<int64_t>(1, (P / 4 + 255) / 256);   // the four-features-per-thread kernel: fewer workgroups, more row splits
  int64_t RS = std::max<int64_t>(1, (2048 + gx - 1) / gx);      // (fewer than four features: gx was 0 -- a division by zero, round 5)
  RS = std::min<int64_t>(RS, std::max<int64_t>(1, n / 64));
  return (size_t)(RS + 1) * P * 32 + (size_t)P * 64 + (size_t)n * 16 + (1 << 20);
}

// shared tail of preprocess/apply: NaN policy, maps, allocation, apply kernel
static int sanitize_and_apply(eofx_ctx* ctx, const float* Xd, int64_t n, int64_t P, PreState& ps,
                              bool stats_absmax, const uint8_t* expect_valid, int check_nans, eofx_mat** out,
                              uint8_t* valid_feature, uint8_t* valid_sample, int64_t* n_out,
                              int64_t* p_out, std::vector<int>& hcnt, FeatSummary* summary = nullptr,
                              bool map_from_these_stats = false) {
  FeatSummary fs;
  CHK(run_feature_summary(ctx, ps, P, fs));
  if (summary) *summary = fs;
  int64_t pv = fs.pv;
  int cmax = fs.cmax, cmin = fs.cmin;
  hcnt.clear();
  if (pv == P) {   // every feature has data (the common case): no need for the P-sized count array on the host
    if (valid_feature) std::memset(valid_feature, 1, (size_t)P);
    if (expect_valid && check_nans)
      for (int64_t c = 0; c < P; ++c)
        if (!expect_valid[c])
          return set_err(ctx, EOFX_ERR_NAN_MISMATCH,
                         "Input data had NaN features in different locations than the original data.");
  } else {
    hcnt.resize(P);
    HIPCHK(hipMemcpyAsync(hcnt.data(), ps.cnt, sizeof(int) * P, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int64_t c = 0; c < P; ++c) {
      const int k = hcnt[c];
      if (valid_feature) valid_feature[c] = k > 0;
      if (expect_valid && check_nans && (expect_valid[c] != 0) != (k > 0))
        return set_err(ctx, EOFX_ERR_NAN_MISMATCH,
                       "Input data had NaN features in different locations than the original data.");
    }
  }
  if (pv == 0) return set_err(ctx, EOFX_ERR_ARG, "input has no valid (non-NaN) feature");
  static const char* kPartial =
      "Input data contains partial NaN entries, which will cause the the SVD to fail.";
  if (check_nans && cmin != cmax) return set_err(ctx, EOFX_ERR_PARTIAL_NAN, kPartial);
  // an infinity in the field: the Scaler runs BEFORE the Sanitizer (preprocessor.py), the mean of that feature is infinite and
  // x - mean leaves -inf and one NaN in its column -- the reference stops with the partial-NaN message; only an uncentred,
  // unstandardised fit carries the infinity into the decomposition (which then fails with its own message)
  if (check_nans && map_from_these_stats && !std::isfinite(fs.tv)) return set_err(ctx, EOFX_ERR_PARTIAL_NAN, kPartial);
  std::vector<int64_t> row_map;
  int64_t ns = n;
  if (check_nans && cmax < n) {  // some samples are missing entirely: find which
    ArenaScope scope(ctx);
    ARENA(int, rowcnt, (size_t)n);
    hipLaunchKernelGGL(rowcount_kernel, dim3((int)n), dim3(256), 0, ctx->stream, Xd, n, P, ps.cnt, rowcnt);
    KCHK();
    std::vector<int> hrow(n);
    HIPCHK(hipMemcpyAsync(hrow.data(), rowcnt, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ns = 0;
    for (int64_t r = 0; r < n; ++r) {
      if (hrow[r] != 0 && hrow[r] != pv) return set_err(ctx, EOFX_ERR_PARTIAL_NAN, kPartial);
      if (hrow[r] > 0) {
        row_map.push_back(r);
        ++ns;
      }
      if (valid_sample) valid_sample[r] = hrow[r] > 0;
    }
  } else if (valid_sample) {
    std::memset(valid_sample, 1, (size_t)n);
  }
  if (n_out) *n_out = ns;
  if (p_out) *p_out = pv;
  if (!out) return EOFX_OK;

  // raw mode (eofx_ctx_set_layout): nothing is dropped or reordered, so the raw field itself -- read through the
  // affine map -- is the feature-contiguous layout; only the sample-contiguous one is written.  P < 2^31 rows: int.
  const bool raw_ok = ctx->keep_raw && ns == n && P % 4 == 0 && ((uintptr_t)Xd % 16) == 0 && n < ((int64_t)1 << 31);
  // masked in place (layout mode 3): all-NaN grid points (sanitizer.py:80-126 would drop them) stay in the matrix as zero
  // columns -- scale 0 in the map, bits ANDed to +0 by the MASK kernels -- when they are a minority and the sketch lives
  // on the sample side; a zero column changes no product, Gram matrix or norm, so the factors are those of the compacted
  // matrix with zero rows in V at the masked features (the caller drops them).  1x the field in HBM instead of 3x.
  const bool masked_mode = raw_ok && ctx->keep_raw == 2 && ctx->allow_masked && stats_absmax && pv < P && 10 * pv >= 6 * P && n < pv;
  const bool raw_mode = (raw_ok && pv == P) || masked_mode;
  eofx_mat* m = nullptr;
  const bool in_place = raw_mode && ctx->keep_raw == 2;
  CHK(mat_alloc(ctx, ns, masked_mode ? P : pv, &m, !raw_mode, !in_place));
  m->p_valid = pv;
  m->masked = masked_mode;
  int rc = EOFX_OK;
  if (raw_mode) {
    const size_t abytes = sizeof(float) * 3 * (size_t)m->p_pad;
    if (pool_malloc(ctx, (void**)&m->aff, abytes) != hipSuccess) {
      (void)hipGetLastError();
      m->aff = nullptr;
      eofx_mat_destroy(ctx, m);
      return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the affine map (%zu bytes)", abytes);
    }
    hipLaunchKernelGGL(aff_pack_kernel, dim3((int)((m->p_pad + 255) / 256)), dim3(256), 0, ctx->stream, ps.shift, ps.scale, P,
                       m->p_pad, m->aff, masked_mode ? ps.cnt : (const int*)nullptr);
    if (hipGetLastError() != hipSuccess) {
      eofx_mat_destroy(ctx, m);
      return set_err(ctx, EOFX_ERR_HIP, "raw mode: affine map kernel failed");
    }
    m->raw = Xd;
    m->raw_ld = P;
  }
  {
    ArenaScope scope(ctx);
    int64_t* dcol = nullptr;
    int64_t* drow = nullptr;
    int* flag = arena_alloc<int>(ctx, 1);
    if (pv < P && !masked_mode) {
      std::vector<int64_t> col_map;
      col_map.reserve(pv);
      for (int64_t c = 0; c < P; ++c)
        if (hcnt[c] > 0) col_map.push_back(c);
      dcol = arena_alloc<int64_t>(ctx, pv);
      if (dcol && hipMemcpyAsync(dcol, col_map.data(), sizeof(int64_t) * pv, hipMemcpyHostToDevice,
                                 ctx->stream) != hipSuccess)
        rc = EOFX_ERR_HIP;
      if (rc == EOFX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
    }
    if (rc == EOFX_OK && !row_map.empty() && ns < n) {
      drow = arena_alloc<int64_t>(ctx, ns);
      if (drow && hipMemcpyAsync(drow, row_map.data(), sizeof(int64_t) * ns, hipMemcpyHostToDevice,
                                 ctx->stream) != hipSuccess)
        rc = EOFX_ERR_HIP;
      if (rc == EOFX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
    }
    if (!flag || (pv < P && !masked_mode && !dcol) || (ns < n && !drow)) rc = set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (maps)");
    if (rc == EOFX_OK && hipMemsetAsync(flag, 0, sizeof(int), ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
    if (rc == EOFX_OK && in_place && !stats_absmax) rc = set_err(ctx, EOFX_ERR_ARG, "in-place layout needs the column statistics");
    if (rc == EOFX_OK && in_place) {   // nothing to write: max |x'| comes from the column statistics
      if (hipMemcpyAsync(m->absmax_dev, ps.absmax, sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(&m->absmax, m->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = EOFX_ERR_HIP;
    } else if (rc == EOFX_OK)
      rc = launch_apply(ctx, Xd, P, drow, dcol, ps.shift, ps.scale, m, flag, stats_absmax ? ps.absmax : nullptr);
    if (rc == EOFX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
  }
  if (rc != EOFX_OK) {
    eofx_mat_destroy(ctx, m);
    if (rc == EOFX_ERR_HIP) set_err(ctx, rc, "HIP failure in apply: %s", hipGetErrorString(hipGetLastError()));
    return rc;
  }
  *out = m;
  return EOFX_OK;
}

extern "C" int eofx_preprocess_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t P, int center,
                                   int standardize, const double* feat_weights, int check_nans,
                                   eofx_mat** out, double* mean, double* std_, uint8_t* valid_feature,
                                   uint8_t* valid_sample, int64_t* n_out, int64_t* p_out,
                                   double* total_variance) {
  if (!ctx || !X || n <= 0 || P <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  Staged st;
  CHK(stage_input(ctx, X, (size_t)n * P, st));
  CHK(arena_reserve(ctx, colstats_scratch(n, P)));
  ArenaScope scope(ctx);
  PreState ps;
  ARENA(int, cnt, P);
  ARENA(double, dmean, P);
  ARENA(double, dstd, P);
  ARENA(double, dshift, P);
  ARENA(double, dscale, P);
  ARENA(double, dm2, P);
  ARENA(unsigned, dabsmax, 1);
  ps = {cnt, dmean, dstd, dshift, dscale, dm2, dabsmax};
  double* wdev = nullptr;
  if (feat_weights) {
    wdev = arena_alloc<double>(ctx, P);
    if (!wdev) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (weights)");
    CHK(copy_in(ctx, wdev, feat_weights, sizeof(double) * P));
  }
  {
    const int rc_cs = run_colstats(ctx, st.dev, n, P, center, standardize, wdev, ps);
    if (rc_cs != EOFX_OK) {
      if (ctx->pending_rawT) pool_give(ctx, ctx->pending_rawT, ctx->pending_rawT_bytes);
      ctx->pending_rawT = nullptr;
      ctx->pending_rawT_bytes = 0;
      return rc_cs;
    }
  }
  std::vector<int> hcnt;
  int64_t ns = 0, pv = 0;
  FeatSummary fs;
  const int rc_sa = sanitize_and_apply(ctx, st.dev, n, P, ps, true, nullptr, check_nans, out, valid_feature, valid_sample, &ns,
                                       &pv, hcnt, &fs, center || standardize);
  if (ctx->pending_rawT) {     // the transposed raw field of the statistics pass: to the in-place matrix, or back to the pool
    eofx_mat* m = (rc_sa == EOFX_OK && out) ? *out : nullptr;
    if (m && m->raw && m->aff && !m->X && !m->Xt && m->n == n && m->p == P &&
        (size_t)m->n_pad * m->p_pad * sizeof(float) == ctx->pending_rawT_bytes)
      m->rawT = ctx->pending_rawT;
    else
      pool_give(ctx, ctx->pending_rawT, ctx->pending_rawT_bytes);
    ctx->pending_rawT = nullptr;
    ctx->pending_rawT_bytes = 0;
  }
  CHK(rc_sa);
  adopt_staged(out, st, (size_t)n * P * sizeof(float));
  if (n_out) *n_out = ns;
  if (p_out) *p_out = pv;
  if (mean) HIPCHK(hipMemcpyAsync(mean, ps.mean, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
  if (std_) HIPCHK(hipMemcpyAsync(std_, ps.stdv, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
  if (total_variance) *total_variance = fs.tv;   // summed on the device (feature_summary_kernel)
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

extern "C" int eofx_apply_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t P, const double* mean,
                              const double* std_, const double* feat_weights, const uint8_t* valid_feature,
                              int check_nans, eofx_mat** out, uint8_t* valid_sample, int64_t* n_out) {
  if (!ctx || !X || !out || n <= 0 || P <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  Staged st;
  CHK(stage_input(ctx, X, (size_t)n * P, st));
  CHK(arena_reserve(ctx, colstats_scratch(n, P)));
  ArenaScope scope(ctx);
  PreState ps;
  ARENA(int, cnt, P);
  ARENA(double, dmean, P);
  ARENA(double, dstd, P);
  ARENA(double, dshift, P);
  ARENA(double, dscale, P);
  ARENA(double, dm2, P);
  ARENA(unsigned, dabsmax, 1);
  ARENA(float, dvmin, P);
  ARENA(float, dvmax, P);
  ps = {cnt, dmean, dstd, dshift, dscale, dm2, dabsmax, dvmin, dvmax};
  CHK(run_colstats(ctx, st.dev, n, P, 0, 0, nullptr, ps));
  // overwrite shift/scale with the fitted state
  std::vector<double> hshift(P, 0.0), hscale(P, 1.0);
  for (int64_t c = 0; c < P; ++c) {
    if (mean) hshift[c] = mean[c];
    double s = 1.0;
    if (std_) s /= std_[c];
    if (feat_weights) s *= feat_weights[c];
    hscale[c] = s;
    if (!(hshift[c] == hshift[c])) hshift[c] = 0.0;  // NaN stats belong to dropped features
    if (!(hscale[c] == hscale[c])) hscale[c] = 0.0;
  }
  CHK(copy_in(ctx, ps.shift, hshift.data(), sizeof(double) * P));
  CHK(copy_in(ctx, ps.scale, hscale.data(), sizeof(double) * P));
  // max |x'| under the FITTED map from the extremes of the new data: no extra read of the matrix for the fp16 scaling,
  // and the in-place layout (which writes nothing) becomes available for transform() too
  HIPCHK(hipMemsetAsync(ps.absmax, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(fitted_absmax_kernel, dim3((int)((P + 255) / 256)), dim3(256), 0, ctx->stream, ps.cnt, ps.vmin, ps.vmax,
                     ps.shift, ps.scale, P, ps.absmax);
  KCHK();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<int> hcnt;
  int64_t ns = 0, pv = 0;
  CHK(sanitize_and_apply(ctx, st.dev, n, P, ps, true, valid_feature, check_nans, out, nullptr, valid_sample, &ns,
                         &pv, hcnt));
  adopt_staged(out, st, (size_t)n * P * sizeof(float));
  if (n_out) *n_out = ns;
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// bootstrap resampling (validation/bootstrapper.py:78-91): rows of a resident matrix drawn with
// replacement, re-centred, as a new resident matrix -- a row gather inside the statistics and apply
// kernels instead of a host round trip.
// ------------------------------------------------------------------------------------
extern "C" int eofx_resample_f32(eofx_ctx* ctx, const eofx_mat* src, const int64_t* rows, int64_t n_rows,
                                 int center, eofx_mat** out, double* mean, double* total_variance) {
  if (!ctx || !src || !rows || !out || n_rows <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  for (int64_t i = 0; i < n_rows; ++i)
    if (rows[i] < 0 || rows[i] >= src->n)
      return set_err(ctx, EOFX_ERR_ARG, "row index %lld out of range [0, %lld)", (long long)rows[i], (long long)src->n);
  const int64_t P = src->p;
  CHK(arena_reserve(ctx, colstats_scratch(n_rows, P)));
  ArenaScope scope(ctx);
  PreState ps;
  ARENA(int, cnt, P);
  ARENA(double, dmean, P);
  ARENA(double, dstd, P);
  ARENA(double, dshift, P);
  ARENA(double, dscale, P);
  ARENA(double, dm2, P);
  ARENA(unsigned, dabsmax, 1);
  ARENA(int64_t, drows, n_rows);
  ps = {cnt, dmean, dstd, dshift, dscale, dm2, dabsmax};
  CHK(copy_in(ctx, drows, rows, sizeof(int64_t) * n_rows));
  CHK(ensure_X(ctx, src));
  CHK(run_colstats(ctx, src->X, n_rows, P, center, 0, nullptr, ps, src->p_pad, drows));
  eofx_mat* m = nullptr;
  CHK(mat_alloc(ctx, n_rows, P, &m));
  ARENA(int, flag, 1);
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
  int rc = launch_apply(ctx, src->X, src->p_pad, drows, nullptr, ps.shift, ps.scale, m, flag, ps.absmax);
  if (rc == EOFX_OK && mean)
    if (hipMemcpyAsync(mean, ps.mean, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
  if (rc == EOFX_OK && total_variance) {   // sum_c M2_c / (n_rows - 1), summed on the device
    FeatSummary fs;
    rc = run_feature_summary(ctx, ps, P, fs);
    if (rc == EOFX_OK) *total_variance = fs.tv;
  }
  if (rc == EOFX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = EOFX_ERR_HIP;
  if (rc != EOFX_OK) {
    eofx_mat_destroy(ctx, m);
    return rc == EOFX_ERR_HIP ? set_err(ctx, EOFX_ERR_HIP, "resample: HIP failure") : rc;
  }
  *out = m;
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// panel-level ABI
// ------------------------------------------------------------------------------------
static bool tmul_nt_ok(const eofx_mat* m, int L);
static size_t tmul_nt_scratch(eofx_ctx* ctx, const eofx_mat* m, int L);
static int mat_tmul_nt(eofx_ctx* ctx, const eofx_mat* m, const float* Zn, float* Yp, int L);
static int panel_tmul(eofx_ctx* ctx, const eofx_mat* m, const float* Zn, float* Yp, int L, int prec) {
  // panels of 512 columns and more (PCA pre-reduction): the MFMA-bound NT kernel over transposed fp16 planes (eofx_gram.hpp)
  if (prec == EOFX_PREC_F16X3 && tmul_nt_ok(m, L) && tmul_nt_scratch(ctx, m, L) > 0 &&
      ctx->arena_size - ctx->arena_off >= tmul_nt_scratch(ctx, m, L))   // (too little arena: the streaming tiles below, same result)
    return mat_tmul_nt(ctx, m, Zn, Yp, L);
  if (!m->X && m->raw && prec == EOFX_PREC_F16X3) {   // raw mode: stream the raw field through the affine map
    AffView av;
    av.aff = m->aff;
    av.ld = m->p_pad;
    av.rows = (int)m->n;
    av.cols = m->p;
    av.masked = m->masked;
    return launch_atb(ctx, m->raw, m->raw_ld, round_up(m->n, ATB_KG), m->p_pad, Zn, L, L, Yp, prec, m->absmax, nullptr, &av);
  }
  CHK(ensure_X(ctx, m));
  return launch_atb(ctx, m->X, m->p_pad, round_up(m->n, ATB_KG), m->p_pad, Zn, L, L, Yp, prec, m->absmax);
}
// The active slab pairs of a masked in-place matrix (eofx_mat::act): one look at the map's scales, once per matrix.
static int ensure_active_pairs(eofx_ctx* ctx, const eofx_mat* cm) {
  eofx_mat* m = const_cast<eofx_mat*>(cm);
  if (!m->masked || m->n_act >= 0 || !m->aff) return EOFX_OK;
  const int64_t npairs = round_up(m->p, AXB_KG) / AXB_KG;
  std::vector<float> sc((size_t)m->p_pad);
  HIPCHK(hipMemcpyAsync(sc.data(), m->aff + 2 * m->p_pad, sizeof(float) * (size_t)m->p_pad, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<int> list;
  list.reserve((size_t)npairs);
  for (int64_t q = 0; q < npairs; ++q) {
    bool any = false;
    for (int64_t c = q * AXB_KG; c < std::min<int64_t>((q + 1) * AXB_KG, m->p) && !any; ++c) any = sc[(size_t)c] != 0.f;
    if (any) list.push_back((int)q);
  }
  m->n_act = (int64_t)list.size();
  if (m->n_act == npairs || m->n_act == 0) return EOFX_OK;     // nothing to skip (or nothing at all): no list
  if (pool_malloc(ctx, (void**)&m->act, sizeof(int) * (size_t)npairs) != hipSuccess) {
    (void)hipGetLastError();
    m->act = nullptr;
    return EOFX_OK;                                              // (without the list every pair is read: still correct)
  }
  HIPCHK(hipMemcpyAsync(m->act, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));                     // `list` leaves scope
  return EOFX_OK;
}

static int panel_mul(eofx_ctx* ctx, const eofx_mat* m, const float* Yp, float* Wn, int L, int prec) {
  // Wide panels (96 columns and more: EOF with 55+ modes) on an in-place matrix: axb_f16 takes 64 columns per launch, so
  // every X Y pass would read the field L / 64 times, while the 128-column tile of atb_f16 over the sample-contiguous
  // layout reads it once per 128.  Where HBM has room for that layout (one more copy of the field), build it once --
  // 13 ms at config-4 size against ~5 ms saved in each of the 8 passes; the raw field stays the feature-side operand.
  if (!m->Xt && m->raw && m->aff && !m->masked && prec == EOFX_PREC_F16X3 && L >= 96 && !std::getenv("EOFX_NO_WIDE_XT")) {
    size_t free_b = 0, total_b = 0;
    const size_t need_b = (size_t)m->n_pad * m->p_pad * sizeof(float);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + ctx->pool_bytes > need_b + need_b / 2 + ((size_t)8 << 30))
      CHK(ensure_Xt(ctx, m));
    else
      (void)hipGetLastError();
  }
  if (!m->Xt && m->raw && m->aff && prec == EOFX_PREC_F16X3 &&
      round_up(m->p, AXB_KG) * (int64_t)L < ((int64_t)1 << 30) &&       // the kernel's 32-bit BYTE offsets into the panel
      64 * m->raw_ld + round_up(m->p, AXB_KG) < ((int64_t)1 << 30)) {   // in place: stream the raw field along its rows
    CHK(ensure_active_pairs(ctx, m));
    return launch_axb(ctx, m->raw, m->raw_ld, m->n, m->p, m->n_pad, m->aff, m->p_pad, m->absmax, Yp, L, Wn, m->masked, m->act,
                      m->n_act);
  }
  CHK(ensure_Xt(ctx, m));
  return launch_atb(ctx, m->Xt, m->n_pad, round_up(m->p, ATB_KG), m->n_pad, Yp, L, L, Wn, prec, m->absmax);
}
static bool valid_prec(int p) {
  return p == EOFX_PREC_F32 || p == EOFX_PREC_BF16X3 || p == EOFX_PREC_BF16X6 || p == EOFX_PREC_F16X3 || p == EOFX_PREC_F64;
}

extern "C" int eofx_ctx_set_precision(eofx_ctx* ctx, int power_passes, int final_passes) {
  if (!ctx || !valid_prec(power_passes) || !valid_prec(final_passes)) return set_err(ctx, EOFX_ERR_ARG, "bad precision");
  ctx->prec_power = power_passes;
  ctx->prec_final = final_passes;
  return EOFX_OK;
}

extern "C" int eofx_panel_tmul_f32(eofx_ctx* ctx, const eofx_mat* m, const float* Zn, float* Yp, int L,
                                   int prec) {
  if (!ctx || !m || !Zn || !Yp || !valid_prec(prec)) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  size_t need = atb_scratch_bytes(m->p_pad, round_up(m->n, ATB_KG), L);
  if (prec == EOFX_PREC_F16X3 && tmul_nt_ok(m, L)) need = std::max(need, tmul_nt_scratch(ctx, m, L));
  CHK(arena_reserve(ctx, need));
  return panel_tmul(ctx, m, Zn, Yp, L, prec);
}
extern "C" int eofx_panel_mul_f32(eofx_ctx* ctx, const eofx_mat* m, const float* Yp, float* Wn, int L,
                                  int prec) {
  if (!ctx || !m || !Wn || !Yp || !valid_prec(prec)) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, atb_scratch_bytes(m->n_pad, round_up(m->p, ATB_KG), L)));
  return panel_mul(ctx, m, Yp, Wn, L, prec);
}
extern "C" int eofx_panel_gram_f64(eofx_ctx* ctx, const float* P, int64_t rows_pad, int L, double* G) {
  if (!ctx || !P || !G || L % 32) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)(4 * gram_parts(rows_pad, L) + 1) * L * L * sizeof(double)));
  return launch_gram(ctx, P, rows_pad, L, G);
}
extern "C" int eofx_panel_cholqr_f32(eofx_ctx* ctx, const float* P, int64_t rows_pad, int L, int l,
                                     const double* G, float* out) {
  if (!ctx || !P || !G || !out || P == out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)2 * L * L * sizeof(double) + (l > 64 ? rinv_blocked_bytes(l) : 0)));
  return launch_cholqr(ctx, P, rows_pad, L, l, G, out);
}
extern "C" int eofx_panel_rinv_f64(eofx_ctx* ctx, const double* G, int L, int l, double* Rinv) {
  if (!ctx || !G || !Rinv || L <= 0 || l <= 0 || l > L || G == Rinv) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (l > 64) CHK(arena_reserve(ctx, rinv_blocked_bytes(l)));
  return launch_rinv(ctx, G, L, l, Rinv);
}
extern "C" int eofx_panel_matmul_f32(eofx_ctx* ctx, const float* P, int64_t rows_pad, int L,
                                     const double* M, int Lo, float* out) {
  if (!ctx || !P || !M || !out || P == out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  // a wide panel times a wide matrix (the PCA pre-reduction's V = B W: 129 600 x 1536 by 1536 x 1504, 0.6 TFLOP) belongs on the
  // matrix cores: fp16 planes of both operands and the tiled NT kernel of eofx_gram.hpp (2.4 ms instead of 22 ms in the float64
  // VALU kernel below, which is made for panels of up to 256 columns)
  if (matmul_nt_ok(rows_pad, L, Lo)) {
    const size_t need = matmul_nt_scratch(ctx, rows_pad, L, Lo);
    if (need && arena_reserve(ctx, need) == EOFX_OK) return launch_matmul_nt(ctx, P, rows_pad, L, M, Lo, out);
  }
  return launch_matmul(ctx, P, rows_pad, L, M, Lo, out);
}
extern "C" int eofx_panel_colminmax_f32(eofx_ctx* ctx, const float* P, int64_t rows, int L, float* mx,
                                        float* mn) {
  if (!ctx || !P || !mx || !mn) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)2 * 1024 * L * sizeof(float) + 4096));
  return launch_colminmax(ctx, P, rows, L, mx, mn);
}
extern "C" int eofx_panel_export_f32(eofx_ctx* ctx, const float* P, int64_t rows, int L, int k,
                                     const double* sign, float* dst) {
  if (!ctx || !P || !dst || k > L) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)rows * k * sizeof(float) + 8192));
  return export_panel(ctx, P, rows, L, k, sign, dst);
}
extern "C" int eofx_panel_import_f32(eofx_ctx* ctx, const float* src, int64_t rows, int l, float* P,
                                     int64_t rows_pad, int L) {
  if (!ctx || !P || !src || l > L || rows > rows_pad) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)rows * l * sizeof(float) + 8192));
  return import_panel(ctx, src, rows, l, P, rows_pad, L);
}

// ------------------------------------------------------------------------------------
// randomized SVD core on an abstract tall operator A (tall x small)
// ------------------------------------------------------------------------------------
struct LinOp {
  int64_t tall, small, tall_pad, small_pad;
  std::function<int(const float*, float*, int, int)> fwd;  // tall panel  = A   * small panel (.., L, prec)
  std::function<int(const float*, float*, int, int)> bwd;  // small panel = A^T * tall panel
  // rows of the tall side sharded over ranks (eofx_fit_sharded_f32): every L x L float64 Gram matrix of a tall panel is
  // summed over the ranks by this hook (count doubles, device, in place, stream order); bwd then includes its own reduction
  std::function<int(double*, int64_t)> reduce_tall_gram;
  // the same for the small side, when that is sharded as well (the cross-covariance operator X^T Y of two sharded fields)
  std::function<int(double*, int64_t)> reduce_small_gram;
  // padded row count of the WHOLE tall side over all ranks (0 = tall_pad): the rule that decides whether the tall panel is
  // re-normalised in every iteration must come out the same on every rank -- it decides which collectives are issued
  int64_t rule_tall_pad = 0;
};

struct RsvdOut {
  float* Tvec;  // [tall_pad x Lo]   singular vectors on the tall side
  float* Svec;  // [small_pad x Lo]  singular vectors on the small side
  int Lo;
  std::vector<double> s;  // singular values, descending (k of them)
  int64_t tall_pad = 0, small_pad = 0;   // padded row counts of the two panels
};

static int rsvd_auto_iters(int k, int64_t n, int64_t p) {
  // sklearn extmath._randomized_svd: n_iter = 7 if n_components < 0.1 * min(M.shape) else 4
  return ((double)k < 0.1 * (double)std::min(n, p)) ? 7 : 4;
}

// All panels are carved from the arena by the caller-visible drivers (reserve first).
constexpr size_t EOFX_ORTH_TALL_BYTES = (size_t)16 << 20;
constexpr double EOFX_PEAKED_RATIO = 30.0;
// Is the tall panel re-normalised between the two products of a power iteration?  scikit-learn normalises after EVERY
// product; leaving that step out is exact in exact arithmetic, but one iteration then squares sigma_1 / sigma_l inside
// the float32 panel and the Cholesky-QR that follows squares it again: modes more than ~500x below the leading one
// drift from the float64 reference (6e-5 at 1370x, lost beyond 4000x; with the step 8e-6 at 41000x, tests
// test_peaked_spectrum_*).  The step costs 0.85 ms per iteration at config 4 (4 % of a fit), so:
//   * always: small tall panels (<= 16 MiB: microseconds), the float64 mode, and the FIRST iteration of every fit;
//   * afterwards only where it matters: after the first iteration the small-side Gram matrix W^T W is a Rayleigh
//     quotient of X X^T on an orthonormal basis; if the square root of the ratio of its extreme eigenvalues
//     (~ sigma_1 / sigma_l) exceeds EOFX_PEAKED_RATIO the remaining iterations keep the step.
// ONE rule (eofx_orth_tall_rule / eofx_peaked_spectrum) for the C++ drivers and the panel-level (sharded) driver.
static bool orth_tall_rule(int64_t tall_pad, int L, int prec_power) {
  if (std::getenv("EOFX_FORCE_ORTH_TALL")) return true;    // experiments (tools/cond_study.py)
  return prec_power == EOFX_PREC_F64 || (size_t)tall_pad * L * sizeof(float) <= EOFX_ORTH_TALL_BYTES;
}
extern "C" int eofx_orth_tall_rule(int64_t tall_rows_pad, int L, int prec_power) {
  return orth_tall_rule(tall_rows_pad, L, prec_power) ? 1 : 0;
}
// G: leading l x l block (row stride ld) of the small-side float64 Gram matrix after the first iteration (host)
extern "C" int eofx_peaked_spectrum(const double* G, int ld, int l) {
  if (!G || l <= 0) return 0;
  std::vector<double> A((size_t)l * l), w(l), V((size_t)l * l);
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) A[(size_t)i * l + j] = 0.5 * (G[(size_t)i * ld + j] + G[(size_t)j * ld + i]);
  for (double v : A)
    if (!std::isfinite(v)) return 0;
  if (eofx_host_eigh_f64(A.data(), l, w.data(), V.data()) != EOFX_OK) return 1;
  const double hi = w[0], lo = w[l - 1];                 // descending
  if (!(hi > 0.0)) return 0;
  if (!(lo > 0.0)) return 1;
  return std::sqrt(hi / lo) > EOFX_PEAKED_RATIO ? 1 : 0;
}

// R^-1 (device, L x L float64, leading l x l block) of the Cholesky factor of G: the device kernel up to one
// wavefront's 64 columns, the host beyond
// Blocked right-looking Cholesky factorisation + triangular inverse on the device for sketches wider than one wavefront
// (the PCA pre-reduction's int(0.3 rank) + 10 columns: l = 1510): 64-column blocks, the diagonal blocks through
// chol_rinv_kernel (same pivots, same dependency rule against the ORIGINAL diagonal), row panels / trailing updates /
// the inverse's block columns through dgemm64_kernel.  ~5 launches per block, everything in float64, fixed order.
// Arena: 2 Lb^2 + Lb 64 + Lb doubles.
static size_t rinv_blocked_bytes(int l) {
  const size_t Lb = (size_t)round_up(l, 64);
  return (2 * Lb * Lb + Lb * Lb / 2 + 64 * 64 + Lb) * sizeof(double) + 8192;
}
static int launch_rinv_blocked(eofx_ctx* ctx, const double* G, int L, int l, double* Rinv) {
  const int Lb = (int)round_up(l, 64), nb = Lb / 64;
  ArenaScope scope(ctx);
  ARENA(double, S, (size_t)Lb * Lb);      // the matrix; R's off-diagonal blocks end up in its upper triangle
  ARENA(double, X, (size_t)Lb * Lb);      // R^-1
  ARENA(double, T, (size_t)Lb * Lb / 2 + 64 * 64);   // the inner products of one merge level of the inverse (<= half the matrix)
  ARENA(double, d0, Lb);
  hipLaunchKernelGGL(chol_blocked_init_kernel, dim3(512), dim3(256), 0, ctx->stream, G, L, l, S, Lb, d0);
  KCHK();
  HIPCHK(hipMemsetAsync(X, 0, sizeof(double) * (size_t)Lb * Lb, ctx->stream));
  for (int j = 0; j < nb; ++j) {
    const int64_t dj = (int64_t)64 * j * Lb + 64 * j;      // the diagonal block
    const int lj = std::min(64, l - 64 * j);
    hipLaunchKernelGGL(chol_rinv_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)(S + dj), Lb, lj, X + dj, 1e-13,
                       (const double*)(d0 + 64 * j));
    KCHK();
    const int rest = nb - 1 - j;
    if (rest > 0) {
      double* row = S + dj + 64;                            // S[j, j+1 ..] -> R[j, j+1 ..] = X_jj^T S[j, j+1 ..]
      hipLaunchKernelGGL(dgemm64_kernel<true>, dim3(rest, 1), dim3(256), 0, ctx->stream, (const double*)(X + dj), Lb,
                         (const double*)row, Lb, row, Lb, 64, 1.0, 0.0, 0);
      KCHK();
      double* trail = S + dj + (int64_t)64 * Lb + 64;       // S[j+1 .., j+1 ..] -= R[j, j+1 ..]^T R[j, j+1 ..]  (upper tiles)
      hipLaunchKernelGGL(dgemm64_kernel<true>, dim3(rest, rest), dim3(256), 0, ctx->stream, (const double*)row, Lb,
                         (const double*)row, Lb, trail, Lb, 64, -1.0, 1.0, 1);
      KCHK();
    }
  }
  // R^-1 by merging inverted diagonal segments pairwise, level by level: for [A B; 0 C] with A^-1, C^-1 at hand the block above
  // the diagonal is -A^-1 (B C^-1) -- two products with plenty of tiles each, all pairs of a level in ONE batched launch.
  // 24 blocks: 3 batched levels (1+1, 2+2, 4+4 blocks) and two single merges (8+8, 16+8) = 10 launches; the block column by
  // block column form this replaces took 46, the last ones with a K of 1472 on a handful of workgroups (2.5 of the 3.3 ms).
  {
    std::vector<int> seg((size_t)nb, 1);      // sizes (in 64-blocks) of the inverted segments along the diagonal
    while (seg.size() > 1) {
      std::vector<int> next;
      const size_t npairs = seg.size() / 2;
      bool uniform = true;
      for (size_t i = 0; i < 2 * npairs; ++i) uniform = uniform && seg[i] == seg[0];
      auto merge = [&](int64_t off, int m1, int m2, int batch, int64_t diag_stride) {   // off, m1, m2 in elements
        // T = R12 X22 ; X12 = -X11 T
        hipLaunchKernelGGL(dgemm64_kernel<false>, dim3(m2 / 64, m1 / 64, batch), dim3(256), 0, ctx->stream,
                           (const double*)(S + off * Lb + off + m1), Lb, (const double*)(X + (off + m1) * Lb + off + m1), Lb, T, m2, m2, 1.0,
                           0.0, 0, diag_stride, diag_stride, (int64_t)m1 * m2);
        hipLaunchKernelGGL(dgemm64_kernel<false>, dim3(m2 / 64, m1 / 64, batch), dim3(256), 0, ctx->stream,
                           (const double*)(X + off * Lb + off), Lb, (const double*)T, m2, X + off * Lb + off + m1, Lb, m1, -1.0, 0.0, 0,
                           diag_stride, (int64_t)m1 * m2, diag_stride);
      };
      if (uniform && npairs > 0) {
        const int m = 64 * seg[0];
        merge(0, m, m, (int)npairs, (int64_t)2 * m * (Lb + 1));
        KCHK();
        for (size_t i = 0; i < npairs; ++i) next.push_back(2 * seg[0]);
      } else {
        int64_t off = 0;
        for (size_t i = 0; i < npairs; ++i) {
          const int m1 = 64 * seg[2 * i], m2 = 64 * seg[2 * i + 1];
          merge(off, m1, m2, 1, 0);
          KCHK();
          off += m1 + m2;
          next.push_back(seg[2 * i] + seg[2 * i + 1]);
        }
      }
      if (seg.size() & 1) next.push_back(seg.back());
      seg.swap(next);
    }
  }
  hipLaunchKernelGGL(chol_blocked_export_kernel, dim3(512), dim3(256), 0, ctx->stream, (const double*)X, Lb, l, Rinv, L);
  KCHK();
  return EOFX_OK;
}
static int launch_rinv(eofx_ctx* ctx, const double* G, int L, int l, double* Rinv) {
  if (l <= 64) {
    hipLaunchKernelGGL(chol_rinv_kernel, dim3(1), dim3(256), 0, ctx->stream, G, L, l, Rinv, 1e-13, (const double*)nullptr);
    KCHK();
  } else if (ctx->arena_size - ctx->arena_off >= rinv_blocked_bytes(l) && !std::getenv("EOFX_HOST_RINV")) {
    CHK(launch_rinv_blocked(ctx, G, L, l, Rinv));
  } else {
    std::vector<double> hG((size_t)L * L), hR((size_t)L * L);
    HIPCHK(hipMemcpyAsync(hG.data(), G, sizeof(double) * L * L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    host_chol_rinv(hG.data(), L, l, hR.data(), 1e-13);
    HIPCHK(hipMemcpyAsync(Rinv, hR.data(), sizeof(double) * L * L, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return EOFX_OK;
}

// first_fwd (optional): replaces the FIRST product A * Zs of the run (the fused fit computes it together with the
// column statistics, eofx_fit.hpp).  It may return EOFX_FIT_FALLBACK (> 0), which is passed through to the caller.
constexpr int EOFX_FIT_FALLBACK = 1;
typedef std::function<int(const float*, float*, int)> FirstFwd;
// range (optional): replaces the power iterations AND the range product -- given the imported sketch Zs it returns
// Yt = A (A^T A)^n_iter Zs (any column scaling), computed however the caller likes (the cross-covariance driver iterates
// in sample space through the two Gram matrices, eofx_crosscov_rsvd_f32).
typedef std::function<int(const float*, float*, int)> RangeFinder;
static int rsvd_core(eofx_ctx* ctx, const LinOp& op, int k, int l, int n_iter, const float* omega,
                     RsvdOut& out, const FirstFwd* first_fwd = nullptr, const RangeFinder* range = nullptr) {
  const int L = (int)round_up(l, 32);
  const int Lo = (int)round_up(k, 32);
  AmaxScope amax_scope(ctx);
  ARENA(float, Zs, (size_t)op.small_pad * L);
  ARENA(float, Ws, (size_t)op.small_pad * L);
  ARENA(float, Yt, (size_t)op.tall_pad * L);
  ARENA(float, Qt, (size_t)op.tall_pad * L);
  ARENA(float, Tv, (size_t)op.tall_pad * Lo);
  ARENA(float, Sv, (size_t)op.small_pad * Lo);
  ARENA(double, G, (size_t)L * L);
  ARENA(double, G0, (size_t)L * L);
  ARENA(double, R2, (size_t)L * L);
  ARENA(double, Md, (size_t)L * Lo);
  ARENA(double, Md2, (size_t)L * Lo);
  double* pin = nullptr;
  CHK(pinned_scratch(ctx, &pin));
  if ((size_t)2 * L * L > EOFX_PINNED_DOUBLES) return set_err(ctx, EOFX_ERR_ARG, "internal: sketch too wide for the host scratch");

  CHK(import_panel(ctx, omega, op.small, l, Zs, op.small_pad, L));
  bool first_done = false;
  auto fwd = [&](const float* z, float* y, int prec) -> int {
    if (first_fwd && !first_done) {
      first_done = true;
      return (*first_fwd)(z, y, L);
    }
    first_done = true;
    return op.fwd(z, y, L, prec);
  };
  auto gram_tall = [&](const float* Pn, double* Gout) -> int {
    CHK(launch_gram(ctx, Pn, op.tall_pad, L, Gout));
    return op.reduce_tall_gram ? op.reduce_tall_gram(Gout, (int64_t)L * L) : EOFX_OK;
  };
  auto gram_small = [&](const float* Pn, double* Gout) -> int {
    CHK(launch_gram(ctx, Pn, op.small_pad, L, Gout));
    return op.reduce_small_gram ? op.reduce_small_gram(Gout, (int64_t)L * L) : EOFX_OK;
  };
  // power iterations: Z <- orth(A^T (A Z)).  Only the small-side panel is orthonormalised
  // (Cholesky-QR with a float64 Gram matrix); the tall panel is never factorised here.
  const int pp = ctx->prec_power, pf = ctx->prec_final;
  // scikit-learn re-normalises after EVERY product (LU).  Leaving the tall panel as it is between the two
  // products of an iteration is exact in exact arithmetic, but in float32 the unconverged noise-bulk modes of
  // small problems (k ~ n/3 on ~100 samples) then drift 1e-4 from the float64 reference instead of 2e-6.
  // Where the tall panel is small (<= EOFX_ORTH_TALL_BYTES) the extra Cholesky-QR costs microseconds and is
  // done; for large panels it made no measurable difference (tools/cond_study.py) and would cost 6 % of a
  // config-4 fit, so it is skipped there.
  const bool orth_always = orth_tall_rule(op.rule_tall_pad > 0 ? op.rule_tall_pad : op.tall_pad, L, pp);
  bool orth_rest = orth_always;
  // The peaked-spectrum question (one L x L download per fit) is asked after the first iteration and answered in the
  // MIDDLE of the second: the copy goes to page-locked memory behind an event, the next product is launched, and the
  // host solves its l x l eigen-problem while that product streams the matrix.
  hipEvent_t peaked_ev = nullptr;
  bool peaked_pending = false;
  int rc = EOFX_OK;
  for (int it = 0; it < (range ? 0 : n_iter) && rc == EOFX_OK; ++it) {
    rc = fwd(Zs, Yt, pp);
    if (rc != EOFX_OK) break;
    if (peaked_pending) {
      peaked_pending = false;
      if (hipEventSynchronize(peaked_ev) != hipSuccess) rc = set_err(ctx, EOFX_ERR_HIP, "event wait failed");
      else orth_rest = eofx_peaked_spectrum(pin, L, l) != 0;
      if (rc != EOFX_OK) break;
    }
    if (it == 0 || orth_rest) {
      if ((rc = gram_tall(Yt, G)) != EOFX_OK) break;
      if ((rc = launch_cholqr(ctx, Yt, op.tall_pad, L, l, G, Qt)) != EOFX_OK) break;
      rc = op.bwd(Qt, Ws, L, pp);
    } else {
      rc = op.bwd(Yt, Ws, L, pp);
    }
    if (rc != EOFX_OK) break;
    const bool ask = it == 0 && !orth_always && n_iter > 1;
    if ((rc = gram_small(Ws, ask ? G0 : G)) != EOFX_OK) break;
    if (ask) {   // peaked spectrum?
      if (hipMemcpyAsync(pin, G0, sizeof(double) * L * L, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          (!peaked_ev && hipEventCreateWithFlags(&peaked_ev, hipEventDisableTiming) != hipSuccess) ||
          hipEventRecord(peaked_ev, ctx->stream) != hipSuccess) {
        rc = set_err(ctx, EOFX_ERR_HIP, "asynchronous Gram download failed");
        break;
      }
      peaked_pending = true;
    }
    rc = launch_cholqr(ctx, Ws, op.small_pad, L, l, ask ? G0 : G, Zs);
  }
  if (peaked_ev) {
    if (peaked_pending) (void)hipEventSynchronize(peaked_ev);
    (void)hipEventDestroy(peaked_ev);
  }
  if (rc != EOFX_OK) return rc;
  // range basis on the tall side: Q = orth(A Z), CholeskyQR2.  The first factor is applied to the tall panel
  // (Q1 = Y R1^-1, float32); the second one, R2 = chol(Q1^T Q1) = I + O(eps cond^2), is not: B^T = A^T Q =
  // (A^T Q1) R2^-1 is a product on the SMALL side and U = Q Uh = Q1 (R2^-1 Uh) folds it into the final rotation --
  // one pass over the tall panel less, and Q itself is never rounded to float32.
  // The range-basis pass only fixes a subspace: power-pass precision is enough; the projection B^T = A^T Q below
  // decides the singular values and uses the final one.
  if (range) CHK((*range)(Zs, Yt, L));
  else CHK(fwd(Zs, Yt, pp));
  CHK(gram_tall(Yt, G));
  CHK(launch_cholqr(ctx, Yt, op.tall_pad, L, l, G, Qt));      // Q1
  CHK(gram_tall(Qt, G));
  CHK(launch_rinv(ctx, G, L, l, R2));                          // R2^-1
  // B^T = A^T Q  (small x l);  B B^T = (B^T)^T (B^T)
  CHK(op.bwd(Qt, Zs, L, pf));                                  // A^T Q1 (Zs is free now)
  CHK(launch_matmul(ctx, Zs, op.small_pad, L, R2, L, Ws));     // (A^T Q1) R2^-1
  CHK(gram_small(Ws, G));
  double* hG = pin;
  double* hR2 = pin + (size_t)L * L;
  HIPCHK(hipMemcpyAsync(hG, G, sizeof(double) * L * L, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(hR2, R2, sizeof(double) * L * L, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<double> Gl((size_t)l * l), w(l), Uh((size_t)l * l);
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) {
      const double v = 0.5 * (hG[(size_t)i * L + j] + hG[(size_t)j * L + i]);
      if (!std::isfinite(v))
        return set_err(ctx, EOFX_ERR_LINALG,
                       "SVD failed. This may be due to isolated NaN values in the data.");
      Gl[(size_t)i * l + j] = v;
    }
  eofx_host_eigh_f64(Gl.data(), l, w.data(), Uh.data());
  out.s.assign(k, 0.0);
  std::vector<double> M1((size_t)L * Lo, 0.0), M2((size_t)L * Lo, 0.0);
  std::vector<double> invs(k);
  for (int j = 0; j < k; ++j) {
    const double sv = std::sqrt(std::max(w[j], 0.0));
    out.s[j] = sv;
    invs[j] = sv > 0.0 ? 1.0 / sv : 0.0;
  }
  for (int i = 0; i < l; ++i) {                                // M1 = R2^-1 Uh[:, :k] (rows accumulated over q in the same order
    double* m1 = &M1[(size_t)i * Lo];                          // as before: contiguous in j), M2 = Uh[:, :k] / s
    for (int q = 0; q < l; ++q) {
      const double r = hR2[(size_t)i * L + q];
      const double* uq = &Uh[(size_t)q * l];
      for (int j = 0; j < k; ++j) m1[j] += r * uq[j];
    }
    for (int j = 0; j < k; ++j) M2[(size_t)i * Lo + j] = Uh[(size_t)i * l + j] * invs[j];
  }
  HIPCHK(hipMemcpyAsync(Md, M1.data(), sizeof(double) * L * Lo, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(Md2, M2.data(), sizeof(double) * L * Lo, hipMemcpyHostToDevice, ctx->stream));
  CHK(launch_matmul(ctx, Qt, op.tall_pad, L, Md, Lo, Tv));
  CHK(launch_matmul(ctx, Ws, op.small_pad, L, Md2, Lo, Sv));
  HIPCHK(hipStreamSynchronize(ctx->stream));      // M1 / M2 are host vectors of this frame
  out.Tvec = Tv;
  out.Svec = Sv;
  out.Lo = Lo;
  out.tall_pad = op.tall_pad;
  out.small_pad = op.small_pad;
  return EOFX_OK;
}

static size_t rsvd_scratch_bytes(int64_t tall_pad, int64_t small_pad, int l, int k) {
  const size_t L = (size_t)round_up(l, 32), Lo = (size_t)round_up(k, 32);
  size_t b = 0;
  b += 2 * small_pad * L * 4 + 2 * tall_pad * L * 4 + tall_pad * Lo * 4 + small_pad * Lo * 4;
  b += atb_scratch_bytes(small_pad, tall_pad, (int)L);        // split-K partials (small side)
  b += atb_scratch_bytes(tall_pad, small_pad, (int)L);        // split-K partials (tall side)
  b += atb_scratch_bytes(small_pad, tall_pad, (int)Lo);
  b += (size_t)(4 * gram_parts(std::max(tall_pad, small_pad), (int)L) + 8) * L * L * 8 + 2 * L * Lo * 8 + 4096;  // gram partials, Rinv, G0, R2, M1, M2
  b += (size_t)std::max(tall_pad, small_pad) * (Lo + L) * 4;  // export / import staging
  if (l > 64) b += rinv_blocked_bytes(l);                     // blocked device Cholesky of a wide sketch
  b += 4 << 20;
  return b;
}

// xeofs sign rule (xarray_utils.py:273-301) from the per-mode max/min of VT
static int comm_allreduce(eofx_ctx* ctx, void* buf, int64_t count, int dtype, int op);
// over_ranks: the rows of Vpanel are this rank's slice of the feature axis -- one all-reduce(max) of [max | -min] over the
// communicator of the context makes the extrema (and so the signs) global
static int sign_rule(eofx_ctx* ctx, const float* Vpanel, int64_t rows, int Lo, int k,
                     std::vector<double>& sign, bool over_ranks = false) {
  ArenaScope scope(ctx);
  ARENA(float, mx, 2 * (size_t)Lo);
  float* mn = mx + Lo;
  CHK(launch_colminmax(ctx, Vpanel, rows, Lo, mx, mn));
  if (over_ranks) {
    hipLaunchKernelGGL(negate_kernel, dim3(1), dim3(256), 0, ctx->stream, mn, Lo);
    KCHK();
    CHK(comm_allreduce(ctx, mx, 2 * (int64_t)Lo, 0, 1));
    hipLaunchKernelGGL(negate_kernel, dim3(1), dim3(256), 0, ctx->stream, mn, Lo);
    KCHK();
  }
  std::vector<float> hmx(Lo), hmn(Lo);
  HIPCHK(hipMemcpyAsync(hmx.data(), mx, sizeof(float) * Lo, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(hmn.data(), mn, sizeof(float) * Lo, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  sign.assign(k, 1.0);
  for (int j = 0; j < k; ++j) sign[j] = (std::fabs(hmx[j]) >= std::fabs(hmn[j])) ? 1.0 : -1.0;
  return EOFX_OK;
}

// sign rule, U / s / V to the caller (host|device)
// Numerically null modes (more modes asked for than the matrix has numerical rank -- exactly low-rank data, a constant
// field, k close to min(n, p)): the small-side vectors are B^T u / s with s at rounding level, the tall-side ones may be the
// zeroed "dead" columns of a Cholesky-QR -- noise or zeros where scikit-learn's randomized_svd (a QR and a dense SVD,
// extmath.py) returns orthonormal factors whatever the values.  Columns [first, k) of such a factor are re-orthonormalised
// against the columns before them, keeping their direction where they have one: block Gram-Schmidt through the float64 Gram
// matrix of the panel (N <- (N - G G^T N) R^-1, host algebra on Lo x Lo, two rounds); a column that is zero, not finite or
// inside the span of the others is first replaced by a fixed pseudo-random vector (null_fill_kernel).  The columns before
// `first` keep their bits.
// `reduce` (optional): the rows of P are sharded over ranks -- the Gram matrix is summed over them (Lo x Lo doubles, device, in
// place) and `rows_total` is the row count over all ranks; every rank then takes the same decisions from the same matrix.
typedef std::function<int(double*, int64_t)> GramReduce;
static int fix_null_columns(eofx_ctx* ctx, float* P, int64_t rows, int64_t rows_pad, int Lo, int k, int first,
                            const GramReduce* reduce = nullptr, int64_t rows_total = -1) {
  if (rows_total < 0) rows_total = rows;
  if (first >= k || rows_total <= first) return EOFX_OK;
  if (!reduce) {   // (the repair never turns a working call into an out-of-memory error: without room in the arena the columns stay)
    const size_t need = (size_t)rows_pad * Lo * 4 + (size_t)(gram_parts(rows_pad, Lo) + 4) * Lo * Lo * 8 + (64 << 10);
    if (ctx->arena_size - ctx->arena_off < need) return EOFX_OK;
  }
  ArenaScope scope(ctx);
  ARENA(double, G, (size_t)Lo * Lo);
  ARENA(double, Mx, (size_t)Lo * Lo);
  ARENA(float, tmp, (size_t)rows_pad * Lo);
  ARENA(int, dflag, Lo);
  const int m = k - first;
  std::vector<double> hG((size_t)Lo * Lo), hM((size_t)Lo * Lo);
  std::vector<int> flag(Lo, 0);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 255) / 256, 4096));
  auto gram = [&]() -> int {
    CHK(launch_gram(ctx, P, rows_pad, Lo, G));
    if (reduce && *reduce) CHK((*reduce)(G, (int64_t)Lo * Lo));
    HIPCHK(hipMemcpyAsync(hG.data(), G, sizeof(double) * Lo * Lo, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EOFX_OK;
  };
  auto refill = [&]() -> int {
    HIPCHK(hipMemcpyAsync(dflag, flag.data(), sizeof(int) * Lo, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(null_fill_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P, rows, Lo, first, k, (const int*)dflag);
    KCHK();
    HIPCHK(hipStreamSynchronize(ctx->stream));      // (flag is reused)
    return EOFX_OK;
  };
  int rounds_done = 0;
  for (int attempt = 0; attempt < 8 && rounds_done < 2; ++attempt) {
    CHK(gram());
    // columns that cannot be kept: zero (the dead columns of a Cholesky-QR) or not finite
    bool bad = false;
    std::fill(flag.begin(), flag.end(), 0);
    for (int j = first; j < k; ++j) {
      const double d = hG[(size_t)j * Lo + j];
      if (!std::isfinite(d) || !(d > 1e-30)) flag[j] = 1, bad = true;
      for (int i = 0; i < k && !flag[j]; ++i)
        if (!std::isfinite(hG[(size_t)i * Lo + j])) flag[j] = 1, bad = true;
    }
    if (bad && rounds_done == 0) {
      CHK(refill());
      continue;
    }
    // S = C_NN - C_GN^T C_GN (the columns before `first` are orthonormal), its Cholesky factor R, Mx = [I, -C_GN R^-1; 0, R^-1]
    std::vector<double> S((size_t)m * m), Ri((size_t)m * m, 0.0);
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) {
        double v = hG[(size_t)(first + a) * Lo + first + b];
        for (int g = 0; g < first; ++g) v -= hG[(size_t)g * Lo + first + a] * hG[(size_t)g * Lo + first + b];
        S[(size_t)a * m + b] = v;
      }
    // right-looking Cholesky with a pivot floor: a column left with < 1e-6 of its squared length lay inside the span before it
    bool dependent = false;
    std::vector<double> A(S);
    for (int j = 0; j < m; ++j) {
      const double d = A[(size_t)j * m + j];
      if (!(d > 1e-6 * std::max(S[(size_t)j * m + j], 1e-300)) || !std::isfinite(d)) {
        flag[first + j] = 1;          // (all such columns are found in one sweep: this one drops out of the factorisation)
        dependent = true;
        for (int c = j; c < m; ++c) A[(size_t)j * m + c] = 0.0;
        A[(size_t)j * m + j] = 1.0;
        continue;
      }
      const double rjj = std::sqrt(d);
      A[(size_t)j * m + j] = rjj;
      for (int c = j + 1; c < m; ++c) A[(size_t)j * m + c] /= rjj;
      for (int r = j + 1; r < m; ++r) {
        const double f = A[(size_t)j * m + r];
        for (int c = r; c < m; ++c) A[(size_t)r * m + c] -= f * A[(size_t)j * m + c];
      }
    }
    if (dependent) {
      if (rounds_done > 0) break;     // (cannot happen after a successful round; leave what the round produced)
      CHK(refill());
      continue;
    }
    for (int c = 0; c < m; ++c) {       // R^-1 (upper triangular)
      Ri[(size_t)c * m + c] = 1.0 / A[(size_t)c * m + c];
      for (int r = c - 1; r >= 0; --r) {
        double sum = 0.0;
        for (int t = r + 1; t <= c; ++t) sum += A[(size_t)r * m + t] * Ri[(size_t)t * m + c];
        Ri[(size_t)r * m + c] = -sum / A[(size_t)r * m + r];
      }
    }
    std::fill(hM.begin(), hM.end(), 0.0);
    for (int i = 0; i < first; ++i) hM[(size_t)i * Lo + i] = 1.0;
    for (int a = 0; a < m; ++a)
      for (int b = a; b < m; ++b) hM[(size_t)(first + a) * Lo + first + b] = Ri[(size_t)a * m + b];
    for (int g = 0; g < first; ++g)
      for (int b = 0; b < m; ++b) {
        double v = 0.0;
        for (int a = 0; a <= b; ++a) v += hG[(size_t)g * Lo + first + a] * Ri[(size_t)a * m + b];
        hM[(size_t)g * Lo + first + b] = -v;
      }
    HIPCHK(hipMemcpyAsync(Mx, hM.data(), sizeof(double) * Lo * Lo, hipMemcpyHostToDevice, ctx->stream));
    CHK(launch_matmul(ctx, P, rows_pad, Lo, Mx, Lo, tmp));
    // only the re-orthonormalised columns go back: the columns before `first` keep their bits
    HIPCHK(hipMemcpy2DAsync(P + first, sizeof(float) * Lo, tmp + first, sizeof(float) * Lo, sizeof(float) * (size_t)(k - first), (size_t)rows_pad,
                            hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));      // (hM is reused)
    ++rounds_done;
  }
  return EOFX_OK;
}

// both factors of a finished decomposition: modes whose value is below 1e-5 of the leading one (the level from which the
// eigen-solver's 1e-15 no longer keeps two such vectors orthogonal to 1e-5) go through fix_null_columns
static int fix_null_modes(eofx_ctx* ctx, const RsvdOut& ro, int64_t tall, int64_t small, int k,
                          const GramReduce* reduce_tall = nullptr, int64_t tall_total = -1,
                          const GramReduce* reduce_small = nullptr, int64_t small_total = -1) {
  int first_null = k;
  const double s0 = ro.s.empty() ? 0.0 : ro.s[0];
  for (int j = k - 1; j >= 0 && !(ro.s[j] > 1e-5 * s0); --j) first_null = j;
  // (a constant field: every value is zero, every column is replaced -- scikit-learn returns arbitrary orthonormal factors there too)
  if (first_null < k && s0 >= 0.0 && std::isfinite(s0) && ro.tall_pad >= tall && ro.small_pad >= small) {
    CHK(fix_null_columns(ctx, ro.Tvec, tall, ro.tall_pad, ro.Lo, k, first_null, reduce_tall, tall_total));
    CHK(fix_null_columns(ctx, ro.Svec, small, ro.small_pad, ro.Lo, k, first_null, reduce_small, small_total));
  }
  return EOFX_OK;
}

static int rsvd_finish(eofx_ctx* ctx, const RsvdOut& ro, bool transposed, int64_t n, int64_t p, int k, int flip,
                       float* U, float* s, float* V) {
  CHK(fix_null_modes(ctx, ro, transposed ? p : n, transposed ? n : p, k));
  const float* Vp = transposed ? ro.Tvec : ro.Svec;
  const float* Up = transposed ? ro.Svec : ro.Tvec;
  std::vector<double> sign;
  if (flip) CHK(sign_rule(ctx, Vp, p, ro.Lo, k, sign));
  CHK(export_panel(ctx, Up, n, ro.Lo, k, flip ? sign.data() : nullptr, U));
  CHK(export_panel(ctx, Vp, p, ro.Lo, k, flip ? sign.data() : nullptr, V));
  if (s) {
    std::vector<float> hs(k);
    for (int j = 0; j < k; ++j) hs[j] = (float)ro.s[j];
    HIPCHK(hipMemcpy(s, hs.data(), sizeof(float) * k, hipMemcpyDefault));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

extern "C" int eofx_rsvd_f32(eofx_ctx* ctx, const eofx_mat* m, int k, int n_oversamples, int n_iter,
                             const float* omega, int flip, float* U, float* s, float* V) {
  if (!ctx || !m || !omega || k <= 0 || n_oversamples < 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int64_t n = m->n, p = m->p, r = std::min(n, p);
  if (k > r)
    return set_err(ctx, EOFX_ERR_RANK,
                   "n_modes must be less than or equal to the rank of the dataset (rank = %lld).", (long long)r);
  const int l_req = k + n_oversamples;
  const int l = (int)std::min<int64_t>(l_req, r);
  if (l > EOFX_MAX_SKETCH)
    return set_err(ctx, EOFX_ERR_ARG, "sketch width k+n_oversamples = %d > %d is not supported", l, EOFX_MAX_SKETCH);
  if (n_iter < 0) n_iter = rsvd_auto_iters(k, n, p);
  const bool transposed = n < p;  // sklearn: transpose = n_samples < n_features
  LinOp op;
  if (transposed) {
    op = {p, n, m->p_pad, m->n_pad,
          [&](const float* z, float* y, int L, int pr) { return panel_tmul(ctx, m, z, y, L, pr); },
          [&](const float* y, float* w, int L, int pr) { return panel_mul(ctx, m, y, w, L, pr); }};
  } else {
    op = {n, p, m->n_pad, m->p_pad,
          [&](const float* z, float* y, int L, int pr) { return panel_mul(ctx, m, z, y, L, pr); },
          [&](const float* y, float* w, int L, int pr) { return panel_tmul(ctx, m, y, w, L, pr); }};
  }
  CHK(arena_reserve(ctx, rsvd_scratch_bytes(op.tall_pad, op.small_pad, l, k)));
  ArenaScope scope(ctx);
  // omega arrives as (small x l_req).  A sketch as wide as the rank spans everything, so the identity
  // is used instead of the Gaussian draw: a square Gaussian matrix is occasionally ill conditioned
  // (cond ~ 1e3..1e4) and a single pass would amplify its rounding error by that factor.
  std::vector<float> om_clamped;
  const float* om = omega;
  if (l == r) {
    om_clamped.assign((size_t)op.small * l, 0.f);
    for (int64_t i = 0; i < l; ++i) om_clamped[(size_t)i * l + i] = 1.f;
    om = om_clamped.data();
  } else if (l != l_req) {
    if (is_device_ptr(omega)) return set_err(ctx, EOFX_ERR_ARG, "omega must be a host pointer");
    om_clamped.resize((size_t)op.small * l);
    for (int64_t i = 0; i < op.small; ++i)
      for (int j = 0; j < l; ++j) om_clamped[(size_t)i * l + j] = omega[(size_t)i * l_req + j];
    om = om_clamped.data();
  }
  RsvdOut ro;
  CHK(rsvd_core(ctx, op, k, l, n_iter, om, ro));
  return rsvd_finish(ctx, ro, transposed, n, p, k, flip, U, s, V);
}

// ------------------------------------------------------------------------------------
// The fused fit: Scaler.fit + Sanitizer + Decomposer.fit with the column statistics taken during the FIRST pass of
// the randomized SVD (eofx_fit.hpp): 2 n_iter + 2 reads of the field instead of 2 n_iter + 3.
// ------------------------------------------------------------------------------------
// State of one statistics-carrying first pass.  prepare() carves its buffers from the arena (the caller has reserved
// fit_first_bytes() on top of its own needs and holds the ArenaScope), runs the probe and creates the in-place matrix;
// run() is the pass itself: Yt = X'^T Zs plus the Scaler's state.  Both may return EOFX_FIT_FALLBACK.
struct FitFirst {
  eofx_ctx* ctx = nullptr;
  const float* Xd = nullptr;
  int64_t n = 0, P = 0, p_pad = 0, n_pad = 0, K = 0;
  int center = 1, standardize = 0, l = 0, L = 0, S = 1;
  AtbPlan plan{1, 0};
  PreState ps{};
  double *dcorr = nullptr, *st_sq = nullptr, *wpart = nullptr, *wbar = nullptr, *wdev = nullptr;
  float *cshift = nullptr, *st_max = nullptr;
  int* dflags = nullptr;
  unsigned* hword = nullptr;
  float a_scale = 1.f;
  bool maybe_masked = false;
  bool sharded = false;      // a slice of a sharded field: the range rule of the masked layout is the caller's (global counts)
  eofx_mat* m = nullptr;
  FeatSummary fs;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  static constexpr int NPART = 64;
  ~FitFirst() {
    for (int i = 0; i < 4; ++i)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (m) eofx_mat_destroy(ctx, m);
  }
  static size_t bytes(int64_t n, int64_t P, int l) {
    const int L = (int)round_up(l, 32);
    const int64_t p_pad = round_up(P, ATB_BM);
    const int S = atb_plan(p_pad, round_up(n, ATB_KG), L).S;
    return (size_t)S * p_pad * (8 + 4) + (size_t)P * (4 + 6 * 8) + (size_t)p_pad * 4 + (size_t)P * 16 +
           (size_t)(NPART + 1) * L * 8 + (size_t)S * p_pad * L * 4 + (1 << 20);
  }
  int prepare(eofx_ctx* c, const float* X, int64_t n_, int64_t P_, int center_, int standardize_, const double* feat_weights,
              int l_) {
    ctx = c;
    Xd = X;
    n = n_;
    P = P_;
    center = center_;
    standardize = standardize_;
    l = l_;
    L = (int)round_up(l, 32);
    p_pad = round_up(P, ATB_BM);
    n_pad = round_up(n, ATB_BM);
    K = round_up(n, ATB_KG);
    plan = atb_plan(p_pad, K, L);
    S = plan.S;
    ARENA(int, cnt, P);
    ARENA(double, dmean, P);
    ARENA(double, dstd, P);
    ARENA(double, dshift, P);
    ARENA(double, dscale, P);
    ARENA(double, dm2, P);
    ARENA(unsigned, dabsmax, 4);
    ps = {cnt, dmean, dstd, dshift, dscale, dm2, dabsmax};
    ARENA(double, dcorr_, P);
    ARENA(float, cshift_, p_pad);
    ARENA(double, st_sq_, (size_t)S * p_pad);
    ARENA(float, st_max_, (size_t)S * p_pad);
    ARENA(double, wpart_, (size_t)NPART * L);
    ARENA(double, wbar_, L);
    ARENA(int, dflags_, 4);
    dcorr = dcorr_;
    cshift = cshift_;
    st_sq = st_sq_;
    st_max = st_max_;
    wpart = wpart_;
    wbar = wbar_;
    dflags = dflags_;
    if (feat_weights) {
      wdev = arena_alloc<double>(ctx, P);
      if (!wdev) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (weights)");
      CHK(copy_in(ctx, wdev, feat_weights, sizeof(double) * P));
    }
    double* pin = nullptr;
    CHK(pinned_scratch(ctx, &pin));
    hword = reinterpret_cast<unsigned*>(pin + EOFX_PINNED_DOUBLES - 8);   // 4 words beyond what rsvd_core uses
    if (ctx->profile) {
      for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&ev[i]));
      HIPCHK(hipEventRecord(ev[0], ctx->stream));
    }
    // provisional shift (mean of nine sampled rows) and scale (sampled max |x - c|, with 2^6 of headroom: the lo fp16
    // term keeps 11 bits down to 2^-17 of the largest value, so a generous scale costs nothing; an underestimate is
    // caught by the overflow flag and sends the fit back to the two-step path)
    HIPCHK(hipMemsetAsync(ps.absmax, 0, sizeof(unsigned) * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(dflags, 0, sizeof(int) * 4, ctx->stream));
    hipLaunchKernelGGL(fit_probe_kernel, dim3((int)((p_pad / 4 + 255) / 256)), dim3(256), 0, ctx->stream, Xd, n, P, P, p_pad, cshift,
                       ps.absmax + 1, dflags);
    KCHK();
    HIPCHK(hipMemcpyAsync(hword, ps.absmax + 1, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hword + 1, dflags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float est;
    std::memcpy(&est, hword, sizeof(float));
    // bit 1: some feature is NaN in all nine sampled rows -- an all-NaN grid point if the whole column is; the pass
    // itself verifies that (eofx_fit.hpp) and the matrix takes the masked in-place layout, where the caller allows it
    maybe_masked = (hword[1] & 2) != 0;
    if ((hword[1] & 1) || (maybe_masked && !ctx->allow_masked) || !(est > 0.f) || !std::isfinite(est)) {
      ctx->fit_info[2] = 1.0;    // NaN in the sampled rows / constant or non-finite sample
      return EOFX_FIT_FALLBACK;
    }
    int e2;
    (void)std::frexp(est, &e2);
    a_scale = std::ldexp(1.f, 14 - e2 - 6);
    CHK(mat_alloc(ctx, n, P, &m, false, false));
    if (pool_malloc(ctx, (void**)&m->aff, sizeof(float) * 3 * (size_t)p_pad) != hipSuccess) {
      (void)hipGetLastError();
      m->aff = nullptr;
      return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the affine map");
    }
    m->raw = Xd;
    m->raw_ld = P;
    m->p_valid = P;
    m->masked = maybe_masked;     // the MASK kernels from the second pass on (harmless if no feature turns out masked)
    return EOFX_OK;
  }
  int run(const float* Zs, float* Yt, int LL) {
    // sum_i Omega[i, :] of the rank-one correction
    hipLaunchKernelGGL(panel_colsum_part_kernel, dim3(NPART), dim3(256), 0, ctx->stream, Zs, n, LL, wpart);
    KCHK();
    hipLaunchKernelGGL(panel_colsum_final_kernel, dim3(1), dim3(256), 0, ctx->stream, wpart, NPART, LL, wbar);
    KCHK();
    const float* bmax = amax_get(ctx, Zs);
    if (!bmax) {
      unsigned* bm = ps.absmax + 2;
      const int64_t total4 = n_pad * (LL / 4);
      hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))),
                         dim3(256), 0, ctx->stream, Zs, n_pad, LL, (int64_t)LL, bm);
      KCHK();
      bmax = reinterpret_cast<const float*>(bm);
    }
    float* part = Yt;
    if (S > 1) {
      part = arena_alloc<float>(ctx, (size_t)S * p_pad * LL);
      if (!part) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (first-pass partials)");
    }
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (ctx->profile) {
      HIPCHK(hipEventRecord(ev[1], ctx->stream));
      HIPCHK(hipEventCreate(&p0));
      HIPCHK(hipEventCreate(&p1));
      HIPCHK(hipEventRecord(p0, ctx->stream));
    }
    const dim3 grid((int)(p_pad / ATB_BM), S, 1);
    if (LL == 64)
      hipLaunchKernelGGL(atb_f16_fit_kernel<2>, grid, dim3(256), 0, ctx->stream, Xd, P, (int)n, P, cshift, Zs, LL, part, LL, p_pad, K,
                         plan.kps, a_scale, bmax, st_sq, st_max, p_pad);
    else
      hipLaunchKernelGGL(atb_f16_fit_kernel<1>, grid, dim3(256), 0, ctx->stream, Xd, P, (int)n, P, cshift, Zs, LL, part, LL, p_pad, K,
                         plan.kps, a_scale, bmax, st_sq, st_max, p_pad);
    KCHK();
    if (ctx->profile) {
      HIPCHK(hipEventRecord(p1, ctx->stream));
      ctx->prof_events.emplace_back(p0, p1, 0);
      ctx->prof_flops += 2.0 * (double)K * (double)p_pad * (double)LL;
      ctx->prof_bytes += (double)n * (double)P * 4.0;
      HIPCHK(hipEventRecord(ev[2], ctx->stream));
    }
    hipLaunchKernelGGL(fit_finalize_kernel, dim3((int)((p_pad + 255) / 256)), dim3(256), 0, ctx->stream, part, p_pad, LL, st_sq, st_max,
                       p_pad, S, Xd, P, n, K - n, P, p_pad, cshift, a_scale, center, standardize, wdev,
                       (double)1.1920928955078125e-07, bmax, ps.cnt, ps.mean, ps.stdv, ps.shift, ps.scale, ps.m2, m->aff, dcorr,
                       ps.absmax, dflags + 1);
    KCHK();
    HIPCHK(hipMemcpyAsync(hword + 2, dflags + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hword + 3, ps.absmax, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(m->absmax_dev, ps.absmax, sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream));
    // the rank-one correction does not wait for the verdict on the statistics: it is queued behind them
    hipLaunchKernelGGL(fit_reduce_kernel, dim3((int)std::min<int64_t>((p_pad * (LL / 4) + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                       part, Yt, p_pad, LL, l, S, P, dcorr, ps.scale, wbar, amax_new(ctx, Yt),
                       maybe_masked ? ps.cnt : (const int*)nullptr);
    KCHK();
    if (ctx->profile) HIPCHK(hipEventRecord(ev[3], ctx->stream));
    CHK(run_feature_summary(ctx, ps, P, fs));       // total variance; synchronises
    if (hword[2] & 3) {
      ctx->fit_info[2] = 2.0 + (double)(hword[2] & 3);   // 3: NaN / inf in the field, 4: fp16 overflow of the provisional scale
      return EOFX_FIT_FALLBACK;                          //    (or a finite value in an all-NaN candidate), 5: both
    }
    if (hword[2] & 8) {
      ctx->fit_info[2] = 7.0;    // standardize with feature scales further apart than the first pass's one scale resolves
      return EOFX_FIT_FALLBACK;
    }
    if (fs.pv < P) {   // all-NaN grid points: the masked in-place layout, under the conditions of sanitize_and_apply
      if (!(ctx->allow_masked && (sharded || (10 * fs.pv >= 6 * P && n < fs.pv)))) {
        ctx->fit_info[2] = 6.0;                          // a mask outside the in-place range (too many points, n >= valid p)
        return EOFX_FIT_FALLBACK;
      }
      m->p_valid = fs.pv;
      m->masked = true;
    } else {
      m->masked = false;
    }
    std::memcpy(&m->absmax, hword + 3, sizeof(float));
    return EOFX_OK;
  }
  // which features hold data (host bytes), after run()
  int valid_features(uint8_t* valid_feature) {
    if (!valid_feature) return EOFX_OK;
    if (!m || !m->masked) {
      std::memset(valid_feature, 1, (size_t)P);
      return EOFX_OK;
    }
    std::vector<int> hc((size_t)P);
    HIPCHK(hipMemcpyAsync(hc.data(), ps.cnt, sizeof(int) * P, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int64_t c = 0; c < P; ++c) valid_feature[c] = hc[(size_t)c] > 0;
    return EOFX_OK;
  }
  // mean / std to the caller, the event times of the non-pass work, and the matrix itself
  int finish(double* mean, double* std_, double* total_variance, eofx_mat** out) {
    if (mean) HIPCHK(hipMemcpyAsync(mean, ps.mean, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
    if (std_) HIPCHK(hipMemcpyAsync(std_, ps.stdv, sizeof(double) * P, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (total_variance) *total_variance = fs.tv;
    if (ctx->profile) {
      float t01 = 0.f, t23 = 0.f;
      (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
      (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
      ctx->fit_info[1] = (double)t01 + (double)t23;
    }
    *out = m;
    m = nullptr;
    return EOFX_OK;
  }
};

static bool fit_first_eligible(const eofx_ctx* ctx, const float* Xdev, int64_t n, int64_t P, int l) {
  return ctx->keep_raw == 2 && ctx->prec_power == EOFX_PREC_F16X3 && n < P && l > 0 && l < n && round_up(l, 32) <= 64 &&
         l % 32 != 0 && P % 4 == 0 && ((uintptr_t)Xdev % 16) == 0 && n < ((int64_t)1 << 31) && !std::getenv("EOFX_NO_FUSED_FIT");
}

// ------------------------------------------------------------------------------------
// The fused fit: Scaler.fit + Sanitizer + Decomposer.fit with the column statistics taken during the FIRST pass of
// the randomized SVD (eofx_fit.hpp): 2 n_iter + 2 reads of the field instead of 2 n_iter + 3.
// ------------------------------------------------------------------------------------
static int fit_fused(eofx_ctx* ctx, const float* Xd, int64_t n, int64_t P, int center, int standardize,
                     const double* feat_weights, int k, int l, int n_iter, const float* omega, int flip,
                     eofx_mat** out, double* mean, double* std_, double* total_variance, float* U, float* s, float* V,
                     uint8_t* valid_feature, int64_t* p_valid) {
  const int64_t p_pad = round_up(P, ATB_BM), n_pad = round_up(n, ATB_BM);
  CHK(arena_reserve(ctx, rsvd_scratch_bytes(p_pad, n_pad, l, k) + FitFirst::bytes(n, P, l)));
  ArenaScope scope(ctx);
  FitFirst ff;
  int rc = ff.prepare(ctx, Xd, n, P, center, standardize, feat_weights, l);
  if (rc != EOFX_OK) return rc;
  FirstFwd first = [&](const float* Zs, float* Yt, int LL) -> int { return ff.run(Zs, Yt, LL); };
  const eofx_mat* m = ff.m;
  LinOp op = {P, n, p_pad, n_pad,
              [&](const float* z, float* y, int LL, int pr) { return panel_tmul(ctx, m, z, y, LL, pr); },
              [&](const float* y, float* w, int LL, int pr) { return panel_mul(ctx, m, y, w, LL, pr); }};
  RsvdOut ro;
  rc = rsvd_core(ctx, op, k, l, n_iter, omega, ro, &first);
  if (rc != EOFX_OK) return rc;
  CHK(rsvd_finish(ctx, ro, true, n, P, k, flip, U, s, V));
  CHK(ff.valid_features(valid_feature));
  if (p_valid) *p_valid = ff.m->masked ? ff.m->p_valid : P;
  return ff.finish(mean, std_, total_variance, out);
}

// The first pass on its own, for drivers that put collectives between the passes (the feature-sharded fit: the product
// X_g^T Z of a rank's shard needs no communication, so every rank takes its statistics while it computes it):
// Yp [p_pad x L] = X'^T Zn for the device panel Zn [n_pad x L] whose first l columns are in use (l < L: the spare column
// carries the ones), plus everything eofx_preprocess_f32 returns.  Falls back to eofx_preprocess_f32 + eofx_panel_tmul_f32
// by itself (NaN fields, other precisions, ...); *fused tells which.  Yp must hold round_up(P, 512) rows; after a fallback
// that compacted the field only the first (*out)->p_pad rows are meaningful.
extern "C" int eofx_fit_first_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t P, int center, int standardize,
                                  const double* feat_weights, int check_nans, const float* Zn, int L, int l, float* Yp,
                                  eofx_mat** out, double* mean, double* std_, uint8_t* valid_feature, uint8_t* valid_sample,
                                  int64_t* n_out, int64_t* p_out, double* total_variance, int* fused) {
  if (!ctx || !X || !out || !Zn || !Yp || n <= 0 || P <= 0 || L <= 0 || L % 32 || l <= 0 || l > L)
    return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (fused) *fused = 0;
  ctx->fit_info[0] = ctx->fit_info[1] = ctx->fit_info[2] = 0.0;
  Staged st;
  CHK(stage_input(ctx, X, (size_t)n * P, st));
  if (fit_first_eligible(ctx, st.dev, n, P, l) && round_up(l, 32) == L) {
    CHK(arena_reserve(ctx, FitFirst::bytes(n, P, l) + atb_scratch_bytes(round_up(P, ATB_BM), round_up(n, ATB_KG), L)));
    ArenaScope scope(ctx);
    FitFirst ff;
    int rc = ff.prepare(ctx, st.dev, n, P, center, standardize, feat_weights, l);
    if (rc == EOFX_OK) rc = ff.run(Zn, Yp, L);
    if (rc < 0) return rc;
    if (rc == EOFX_OK) {
      CHK(ff.valid_features(valid_feature));
      const int64_t pv_fused = ff.m->masked ? ff.m->p_valid : P;
      CHK(ff.finish(mean, std_, total_variance, out));
      adopt_staged(out, st, (size_t)n * P * sizeof(float));
      if (valid_sample) std::memset(valid_sample, 1, (size_t)n);
      if (n_out) *n_out = n;
      if (p_out) *p_out = pv_fused;
      if (fused) *fused = 1;
      ctx->fit_info[0] = 1.0;
      return EOFX_OK;
    }
  } else {
    ctx->fit_info[2] = -1.0;
  }
  eofx_mat* m = nullptr;
  int64_t ns = 0, pv = 0;
  CHK(eofx_preprocess_f32(ctx, st.dev, n, P, center, standardize, feat_weights, check_nans, &m, mean, std_, valid_feature,
                          valid_sample, &ns, &pv, total_variance));
  adopt_staged(&m, st, (size_t)n * P * sizeof(float));
  if (n_out) *n_out = ns;
  if (p_out) *p_out = pv;
  int rc = ns == n ? eofx_panel_tmul_f32(ctx, m, Zn, Yp, L, ctx->prec_power) : EOFX_OK;   // dropped samples: the caller re-imports Z
  if (rc != EOFX_OK) {
    const std::string keep = ctx->err;
    eofx_mat_destroy(ctx, m);
    ctx->err = keep;
    return rc;
  }
  *out = m;
  return EOFX_OK;
}

extern "C" int eofx_fit_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t P, int center, int standardize,
                            const double* feat_weights, int check_nans, int k, int n_oversamples, int n_iter,
                            const float* omega, int64_t omega_rows, int flip, eofx_mat** out, double* mean, double* std_,
                            uint8_t* valid_feature, uint8_t* valid_sample, int64_t* n_out, int64_t* p_out,
                            double* total_variance, float* U, float* s, float* V, int* fused) {
  if (!ctx || !X || !out || !omega || n <= 0 || P <= 0 || k <= 0 || n_oversamples < 0)
    return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (fused) *fused = 0;
  ctx->fit_info[0] = ctx->fit_info[1] = ctx->fit_info[2] = 0.0;
  Staged st;
  CHK(stage_input(ctx, X, (size_t)n * P, st));
  const int l_req = k + n_oversamples;
  const int l = (int)std::min<int64_t>(l_req, std::min(n, P));
  const int iters = n_iter < 0 ? rsvd_auto_iters(k, n, P) : n_iter;
  const bool eligible = fit_first_eligible(ctx, st.dev, n, P, l) && k <= n && l == l_req && omega_rows >= n && !is_device_ptr(omega);
  if (eligible) {
    int64_t pv_fused = P;
    const int rc = fit_fused(ctx, st.dev, n, P, center, standardize, feat_weights, k, l, iters, omega, flip, out, mean, std_,
                             total_variance, U, s, V, valid_feature, &pv_fused);
    if (rc < 0) return rc;
    if (rc == EOFX_OK) {
      adopt_staged(out, st, (size_t)n * P * sizeof(float));
      if (valid_sample) std::memset(valid_sample, 1, (size_t)n);
      if (n_out) *n_out = n;
      if (p_out) *p_out = pv_fused;
      if (fused) *fused = 1;
      ctx->fit_info[0] = 1.0;
      return EOFX_OK;
    }
  } else {
    ctx->fit_info[2] = -1.0;   // shape / precision / layout outside the fused path
  }
  // the two-step path: statistics pass, NaN policies of the Sanitizer, then the decomposition
  eofx_mat* m = nullptr;
  int64_t ns = 0, pv = 0;
  CHK(eofx_preprocess_f32(ctx, st.dev, n, P, center, standardize, feat_weights, check_nans, &m, mean, std_, valid_feature,
                          valid_sample, &ns, &pv, total_variance));
  adopt_staged(&m, st, (size_t)n * P * sizeof(float));
  if (n_out) *n_out = ns;
  if (p_out) *p_out = pv;
  const int64_t small = std::min(m->n, m->p);
  int rc = EOFX_OK;
  if (omega_rows < small)
    rc = set_err(ctx, EOFX_ERR_ARG, "omega has %lld rows, the compacted matrix needs %lld", (long long)omega_rows, (long long)small);
  // numpy fills the sketch row by row, so the (small x l) draw of the reference is the leading part of a taller one
  if (rc == EOFX_OK) rc = eofx_rsvd_f32(ctx, m, k, n_oversamples, n_iter, omega, flip, U, s, V);
  if (rc != EOFX_OK) {
    const std::string keep = ctx->err;
    eofx_mat_destroy(ctx, m);
    ctx->err = keep;
    return rc;
  }
  *out = m;
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// communicator of the feature-sharded entry (include/eofx.h)
// ------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, ncclUniqueIdCompat, int) = nullptr;
  int (*allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*errstr)(int) = nullptr;
};
static int rccl_open(eofx_ctx* ctx, RcclApi& api) {
  static RcclApi cached;
  if (!cached.lib) {
    void* h = nullptr;
    // an instance already in the process first (PyTorch ships its own librccl: two instances would not share communicators)
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (!h)
      if (const char* ev = std::getenv("EOFX_RCCL_LIB")) h = dlopen(ev, RTLD_NOW | RTLD_GLOBAL);
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_err(ctx, EOFX_ERR_HIP, "cannot open librccl (%s); set EOFX_RCCL_LIB", dlerror());
    cached.lib = h;
    cached.get_unique_id = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    cached.comm_init_rank = reinterpret_cast<int (*)(void**, int, ncclUniqueIdCompat, int)>(dlsym(h, "ncclCommInitRank"));
    cached.allreduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    cached.comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    cached.errstr = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!cached.get_unique_id || !cached.comm_init_rank || !cached.allreduce || !cached.comm_destroy) {
      cached = RcclApi();
      return set_err(ctx, EOFX_ERR_HIP, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    }
  }
  api = cached;
  return EOFX_OK;
}
extern "C" int eofx_comm_unique_id(char* id128) {
  if (!id128) return EOFX_ERR_ARG;
  RcclApi api;
  CHK(rccl_open(nullptr, api));
  ncclUniqueIdCompat id;
  if (api.get_unique_id(&id) != 0) return EOFX_ERR_HIP;
  std::memcpy(id128, id.internal, 128);
  return EOFX_OK;
}
extern "C" int eofx_ctx_comm_clear(eofx_ctx* ctx) {
  if (!ctx) return EOFX_ERR_ARG;
  if (!ctx->comm) return EOFX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->comm->events) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (ctx->comm->comm && ctx->comm->p_destroy) (void)ctx->comm->p_destroy(ctx->comm->comm);
  delete ctx->comm;
  ctx->comm = nullptr;
  return EOFX_OK;
}
extern "C" int eofx_ctx_comm_init_rccl(eofx_ctx* ctx, const char* id128, int world, int rank) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(eofx_ctx_comm_clear(ctx));
  RcclApi api;
  CHK(rccl_open(ctx, api));
  ncclUniqueIdCompat id;
  std::memcpy(id.internal, id128, 128);
  void* comm = nullptr;
  const int rc = api.comm_init_rank(&comm, world, id, rank);
  if (rc != 0 || !comm)
    return set_err(ctx, EOFX_ERR_HIP, "ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, api.errstr ? api.errstr(rc) : "?");
  auto* c = new EofxComm();
  c->world = world;
  c->rank = rank;
  c->lib = api.lib;
  c->comm = comm;
  c->p_allreduce = api.allreduce;
  c->p_destroy = api.comm_destroy;
  c->p_errstr = api.errstr;
  ctx->comm = c;
  return EOFX_OK;
}
extern "C" int eofx_ctx_comm_set_callback(eofx_ctx* ctx, eofx_allreduce_fn fn, void* user, int world, int rank) {
  if (!ctx || !fn || world < 1 || rank < 0 || rank >= world) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  CHK(eofx_ctx_comm_clear(ctx));
  auto* c = new EofxComm();
  c->world = world;
  c->rank = rank;
  c->fn = fn;
  c->user = user;
  ctx->comm = c;
  return EOFX_OK;
}
extern "C" int eofx_ctx_comm_stats(eofx_ctx* ctx, int64_t* calls, int64_t* bytes, double* ms) {
  if (!ctx) return EOFX_ERR_ARG;
  double t = 0.0;
  if (ctx->comm) {
    ENTER(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto& e : ctx->comm->events) {
      float dt = 0.f;
      (void)hipEventElapsedTime(&dt, e.first, e.second);
      t += dt;
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    ctx->comm->events.clear();
  }
  if (calls) *calls = ctx->comm ? ctx->comm->calls : 0;
  if (bytes) *bytes = ctx->comm ? ctx->comm->bytes : 0;
  if (ms) *ms = t;
  if (ctx->comm) ctx->comm->calls = ctx->comm->bytes = 0;
  return EOFX_OK;
}
// all-reduce `count` elements of a device buffer in place, in stream order.  dtype: 0 f32, 1 f64, 2 i32; op: 0 sum, 1 max, 2 min
static int comm_allreduce(eofx_ctx* ctx, void* buf, int64_t count, int dtype, int op) {
  EofxComm* c = ctx->comm;
  if (!c) return EOFX_OK;
  static const int esize[3] = {4, 8, 4};
  c->calls += 1;
  c->bytes += count * esize[dtype];
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profile) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, ctx->stream));
  }
  if (c->fn) {
    if (c->fn(c->user, buf, count, dtype, op, (void*)ctx->stream) != 0)
      return set_err(ctx, EOFX_ERR_HIP, "the all-reduce callback failed");
  } else {
    static const int nccl_type[3] = {7, 8, 2};     // ncclFloat32, ncclFloat64, ncclInt32 (rccl.h)
    static const int nccl_op[3] = {0, 2, 3};       // ncclSum, ncclMax, ncclMin
    const int rc = c->p_allreduce(buf, buf, (size_t)count, nccl_type[dtype], nccl_op[op], c->comm, ctx->stream);
    if (rc != 0) return set_err(ctx, EOFX_ERR_HIP, "ncclAllReduce failed: %s", c->p_errstr ? c->p_errstr(rc) : "?");
  }
  if (ctx->profile) {
    HIPCHK(hipEventRecord(e1, ctx->stream));
    c->events.emplace_back(e0, e1);
  }
  return EOFX_OK;
}
// the ranks agree on the worst of their local verdicts (0 go on, 1 fall back, 2 error); one int32 all-reduce + a host read
static int comm_vote(eofx_ctx* ctx, int local, int* global) {
  *global = local;
  if (!ctx->comm) return EOFX_OK;
  ArenaScope scope(ctx);
  ARENA(int, d, 4);
  HIPCHK(hipMemcpyAsync(d, &local, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  CHK(comm_allreduce(ctx, d, 1, 2, 1));
  HIPCHK(hipMemcpyAsync(global, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

// sum of one integer per rank (float64 carries integers below 2^53 exactly); one all-reduce + a host read
static int comm_sum_i64(eofx_ctx* ctx, int64_t local, int64_t* global) {
  *global = local;
  if (!ctx->comm) return EOFX_OK;
  ArenaScope scope(ctx);
  ARENA(double, d, 2);
  const double v = (double)local;
  double r = 0.0;
  HIPCHK(hipMemcpyAsync(d, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CHK(comm_allreduce(ctx, d, 1, 1, 0));
  HIPCHK(hipMemcpyAsync(&r, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *global = (int64_t)std::llround(r);
  return EOFX_OK;
}

// One round of each collective the sharded fit uses, on known values: sum / max / min of float32, sum of float64, max of
// int32 over {rank + 1}.  *ok = 1 when every result is what `world` ranks must produce.  Collective: every rank calls it.
extern "C" int eofx_ctx_comm_selftest(eofx_ctx* ctx, int* ok) {
  if (!ctx || !ok) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  *ok = 0;
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached");
  ENTER(ctx);
  CHK(arena_reserve(ctx, 4096));
  ArenaScope scope(ctx);
  ARENA(double, d, 8);
  const int world = ctx->comm->world, rank = ctx->comm->rank;
  const float vf = (float)(rank + 1);
  const double vd = (double)(rank + 1);
  const int vi = rank + 1;
  float* f = reinterpret_cast<float*>(d);          // f[0] sum, f[1] max, f[2] min
  double* dd = d + 2;                               // dd[0] sum
  int* ii = reinterpret_cast<int*>(d + 4);         // ii[0] max
  const float hf[3] = {vf, vf, vf};
  HIPCHK(hipMemcpyAsync(f, hf, sizeof(hf), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dd, &vd, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ii, &vi, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));       // the host sources are on this frame
  CHK(comm_allreduce(ctx, f, 1, 0, 0));
  CHK(comm_allreduce(ctx, f + 1, 1, 0, 1));
  CHK(comm_allreduce(ctx, f + 2, 1, 0, 2));
  CHK(comm_allreduce(ctx, dd, 1, 1, 0));
  CHK(comm_allreduce(ctx, ii, 1, 2, 1));
  float rf[3];
  double rd;
  int ri;
  HIPCHK(hipMemcpyAsync(rf, f, sizeof(rf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&rd, dd, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&ri, ii, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const double tri = 0.5 * world * (world + 1);
  *ok = rf[0] == (float)tri && rf[1] == (float)world && rf[2] == 1.f && rd == tri && ri == world;
  return EOFX_OK;
}

// What the first real multi-GPU run needs to diagnose itself: how many ranks the attached communicator
// really reduces over, and what each collective of a sharded fit costs there.  ranks_seen = all-reduce(sum) of 1.0 per rank;
// us[i] = mean time (HIP events on the context's stream, `reps` back-to-back calls after one warm-up) of an all-reduce(sum) of
// counts[i] elements of dtypes[i] (0 float32, 1 float64, 2 int32).  Collective: every rank calls it with the same arguments.
extern "C" int eofx_ctx_comm_probe(eofx_ctx* ctx, int ncases, const int64_t* counts, const int* dtypes, int reps, double* ranks_seen,
                                   double* us) {
  if (!ctx || !counts || !dtypes || !ranks_seen || !us || ncases < 0 || reps <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached");
  ENTER(ctx);
  int64_t most = 8;
  for (int i = 0; i < ncases; ++i) {
    if (counts[i] <= 0 || dtypes[i] < 0 || dtypes[i] > 2) return set_err(ctx, EOFX_ERR_ARG, "bad argument");    // (the same on every rank)
    most = std::max<int64_t>(most, counts[i] * 8);
  }
  {   // a rank-local failure before the first collective (growing the arena) becomes a vote: every rank leaves together
    const int rc_local = arena_reserve(ctx, (size_t)most + 4096);
    const std::string err_local = rc_local != EOFX_OK ? ctx->err : std::string();
    int verdict = 0;
    CHK(comm_vote(ctx, rc_local != EOFX_OK ? 2 : 0, &verdict));
    if (verdict != 0) return rc_local != EOFX_OK ? (ctx->err = err_local, rc_local) : set_err(ctx, EOFX_ERR_HIP, "the communicator probe failed on another rank");
  }
  ArenaScope scope(ctx);
  ARENA(char, buf, (size_t)most);
  HIPCHK(hipMemsetAsync(buf, 0, (size_t)most, ctx->stream));
  const float one = 1.f;
  HIPCHK(hipMemcpyAsync(buf, &one, sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  CHK(comm_allreduce(ctx, buf, 1, 0, 0));
  float seen = 0.f;
  HIPCHK(hipMemcpyAsync(&seen, buf, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *ranks_seen = (double)seen;
  HIPCHK(hipMemsetAsync(buf, 0, (size_t)most, ctx->stream));
  struct Events {        // destroyed on every path out of the function
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  HIPCHK(hipEventCreate(&ev.e0));
  HIPCHK(hipEventCreate(&ev.e1));
  for (int i = 0; i < ncases; ++i) {
    CHK(comm_allreduce(ctx, buf, counts[i], dtypes[i], 0));
    HIPCHK(hipEventRecord(ev.e0, ctx->stream));
    for (int r = 0; r < reps; ++r) CHK(comm_allreduce(ctx, buf, counts[i], dtypes[i], 0));
    HIPCHK(hipEventRecord(ev.e1, ctx->stream));
    HIPCHK(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    us[i] = 1e3 * (double)ms / reps;
  }
  return EOFX_OK;
}

extern "C" int eofx_fit_sharded_f32(eofx_ctx* ctx, const float* X, int64_t n, int64_t P, int64_t P_total, int center,
                                    int standardize, const double* feat_weights, int k, int n_oversamples, int n_iter,
                                    const float* omega, int64_t omega_rows, int flip, eofx_mat** out, double* mean,
                                    double* std_, uint8_t* valid_feature, double* total_variance, float* U, float* s,
                                    float* V) {
  if (!ctx || !X || !out || !omega || n <= 0 || P <= 0 || P_total < P || k <= 0 || n_oversamples < 0)
    return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached (eofx_ctx_comm_init_rccl / eofx_ctx_comm_set_callback)");
  if (!(n < P_total)) return set_err(ctx, EOFX_ERR_ARG, "the sharded fit needs the sketch on the sample side (n < P_total)");
  ENTER(ctx);
  ctx->fit_info[0] = ctx->fit_info[1] = ctx->fit_info[2] = 0.0;
  // A failure of THIS rank before the first vote (staging the slice, growing the arena) must not return here: the other ranks
  // would wait in the vote's all-reduce for ever.  It becomes this rank's verdict (2) and every rank leaves together.
  Staged st;
  int rc = stage_input(ctx, X, (size_t)n * P, st);
  const int l_req = k + n_oversamples;
  const int l = (int)std::min<int64_t>(l_req, n);
  const int iters = n_iter < 0 ? rsvd_auto_iters(k, n, P_total) : n_iter;
  const int64_t p_pad = round_up(P, ATB_BM), n_pad = round_up(n, ATB_BM);
  if (rc == EOFX_OK) {
    const size_t Lo_ = (size_t)round_up(k, 32);
    rc = arena_reserve(ctx, rsvd_scratch_bytes(p_pad, n_pad, l, k) + FitFirst::bytes(n, P, l) + (1 << 16) +
                                (size_t)p_pad * Lo_ * 4 + (size_t)(gram_parts(p_pad, (int)Lo_) + 4) * Lo_ * Lo_ * 8 + (64 << 10));
  }
  ArenaScope scope(ctx);
  FitFirst ff;
  ff.sharded = true;
  if (rc == EOFX_OK) {
    const bool eligible = fit_first_eligible(ctx, st.dev, n, P, l) && k <= n && l == l_req && omega_rows >= n && !is_device_ptr(omega);
    rc = eligible ? ff.prepare(ctx, st.dev, n, P, center, standardize, feat_weights, l) : EOFX_FIT_FALLBACK;
    if (!eligible) ctx->fit_info[2] = -1.0;
  }
  const std::string err_local = rc < 0 ? ctx->err : std::string();
  int verdict = 0;
  CHK(comm_vote(ctx, rc < 0 ? 2 : rc > 0 ? 1 : 0, &verdict));
  if (verdict == 2) return rc < 0 ? (ctx->err = err_local, rc) : set_err(ctx, EOFX_ERR_HIP, "the sharded fit failed on another rank");
  if (verdict == 1) return EOFX_FIT_FALLBACK;
  // the first product + the statistics of the slice; then the ranks agree on whether every slice passed
  FirstFwd first = [&](const float* Zs, float* Yt, int LL) -> int {
    const int r1 = ff.run(Zs, Yt, LL);
    const std::string e1 = r1 < 0 ? ctx->err : std::string();
    int v = 0;
    CHK(comm_vote(ctx, r1 < 0 ? 2 : r1 > 0 ? 1 : 0, &v));
    if (v == 2) return r1 < 0 ? (ctx->err = e1, r1) : set_err(ctx, EOFX_ERR_HIP, "the sharded fit failed on another rank");
    if (v == 1) return EOFX_FIT_FALLBACK;
    // all-NaN grid points (a land / sea mask) stay as zero columns of every slice: the decomposition needs more valid features
    // over ALL slices than samples (the sketch sits on the sample side); one int32 sum, the same verdict on every rank
    int64_t pv_total = 0;
    CHK(comm_sum_i64(ctx, ff.m->masked ? ff.m->p_valid : P, &pv_total));
    if (!(n < pv_total) || k > std::min<int64_t>(n, pv_total)) {
      ctx->fit_info[2] = 6.0;
      return EOFX_FIT_FALLBACK;
    }
    return EOFX_OK;
  };
  const eofx_mat* m = ff.m;
  LinOp op = {P, n, p_pad, n_pad,
              // feature-side panel of this slice: local
              [&](const float* z, float* y, int LL, int pr) { return panel_tmul(ctx, m, z, y, LL, pr); },
              // sample-side panel: the partial sum over this slice's features, then the sum over the slices
              [&](const float* y, float* w, int LL, int pr) {
                CHK(panel_mul(ctx, m, y, w, LL, pr));
                amax_forget(ctx, w);            // (the recorded maximum is that of the partial sum)
                return comm_allreduce(ctx, w, (int64_t)n_pad * LL, 0, 0);
              }};
  op.reduce_tall_gram = [&](double* G, int64_t count) { return comm_allreduce(ctx, G, count, 1, 0); };
  op.rule_tall_pad = round_up(P_total, ATB_BM);
  RsvdOut ro;
  rc = rsvd_core(ctx, op, k, l, iters, omega, ro, &first);
  if (rc != EOFX_OK) return rc;       // (EOFX_FIT_FALLBACK: every rank leaves here together)
  // numerically null modes (more modes than the field has rank): both factors stay orthonormal, as in eofx_fit_f32 -- the
  // feature-side factor through its all-reduced Gram matrix, the replicated sample-side one locally (same bits on every rank)
  CHK(fix_null_modes(ctx, ro, P, n, k, &op.reduce_tall_gram, P_total));
  // sign rule over all slices: one all-reduce(max) of [max | -min]
  std::vector<double> sign(k, 1.0);
  if (flip) {
    ARENA(float, ext, 2 * (size_t)ro.Lo);
    CHK(launch_colminmax(ctx, ro.Tvec, P, ro.Lo, ext, ext + ro.Lo));
    hipLaunchKernelGGL(negate_kernel, dim3(1), dim3(256), 0, ctx->stream, ext + ro.Lo, ro.Lo);
    KCHK();
    CHK(comm_allreduce(ctx, ext, 2 * (int64_t)ro.Lo, 0, 1));
    std::vector<float> h(2 * (size_t)ro.Lo);
    HIPCHK(hipMemcpyAsync(h.data(), ext, sizeof(float) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < k; ++j) sign[j] = (std::fabs(h[j]) >= std::fabs(-h[ro.Lo + j])) ? 1.0 : -1.0;
  }
  CHK(export_panel(ctx, ro.Svec, n, ro.Lo, k, flip ? sign.data() : nullptr, U));
  CHK(export_panel(ctx, ro.Tvec, P, ro.Lo, k, flip ? sign.data() : nullptr, V));
  if (s) {
    std::vector<float> hs(k);
    for (int j = 0; j < k; ++j) hs[j] = (float)ro.s[j];
    HIPCHK(hipMemcpy(s, hs.data(), sizeof(float) * k, hipMemcpyDefault));
  }
  CHK(ff.valid_features(valid_feature));
  double tv_local = 0.0;
  CHK(ff.finish(mean, std_, &tv_local, out));
  adopt_staged(out, st, (size_t)n * P * sizeof(float));
  {   // total variance = sum over the slices
    ARENA(double, dtv, 2);
    HIPCHK(hipMemcpyAsync(dtv, &tv_local, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CHK(comm_allreduce(ctx, dtv, 1, 1, 0));
    HIPCHK(hipMemcpyAsync(&tv_local, dtv, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (total_variance) *total_variance = tv_local;
  ctx->fit_info[0] = 1.0;
  return EOFX_OK;
}

// H W (transpose = 0) / H^T Z (transpose = 1) of a bootstrap member on an n x L sample-side panel (device pointers):
// idx [n] the draw, order [n] = stable argsort of idx, rowptr [n + 1] = first position in `order` of every source row
// (all int64, device).  See the kernels in eofx_fit.hpp; P_in and P_out are [rows_pad x L] panels and must differ.
extern "C" int eofx_panel_bootstrap_f32(eofx_ctx* ctx, const float* P_in, int64_t n, int64_t rows_pad, int L,
                                        const int64_t* idx, const int64_t* order, const int64_t* rowptr, int transpose,
                                        float* P_out) {
  if (!ctx || !P_in || !P_out || P_in == P_out || !idx || !order || !rowptr || n <= 0 || rows_pad < n || L <= 0 || L % 4)
    return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int NPART = 64;
  CHK(arena_reserve(ctx, (size_t)(NPART + 1) * L * 8 + 4096));
  ArenaScope scope(ctx);
  ARENA(double, wpart, (size_t)NPART * L);
  ARENA(double, wsum, L);
  const int blocks = (int)std::min<int64_t>((rows_pad * (L / 4) + 255) / 256, 4096);
  if (transpose) {
    hipLaunchKernelGGL(bst_segsum_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P_in, order, rowptr, n, rows_pad, L, P_out);
    KCHK();
    hipLaunchKernelGGL(panel_colsum_part_kernel, dim3(NPART), dim3(256), 0, ctx->stream, P_in, n, L, wpart);     // sum_i Z[i]
  } else {
    hipLaunchKernelGGL(bst_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P_in, idx, n, rows_pad, L, P_out);
    KCHK();
    hipLaunchKernelGGL(panel_colsum_part_kernel, dim3(NPART), dim3(256), 0, ctx->stream, P_out, n, L, wpart);    // c^T W
  }
  KCHK();
  hipLaunchKernelGGL(panel_colsum_final_kernel, dim3(1), dim3(256), 0, ctx->stream, wpart, NPART, L, wsum);
  KCHK();
  hipLaunchKernelGGL(bst_rankone_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P_out, wsum, transpose ? rowptr : (const int64_t*)nullptr, n, L);
  KCHK();
  return EOFX_OK;
}

// [0] 1 when the last eofx_fit_f32 took the fused path, [1] milliseconds of its non-pass work (probe, finalize,
// correction; measured with HIP events when profiling is on, else 0), [2] why not: 0 fused, -1 not eligible (shape,
// precision, layout), 1 NaN / constant data in the sampled rows, 3 NaN or inf in the field, 4 fp16 range of the
// provisional scale exceeded (or a finite value in a column whose sampled rows were all NaN), 5 both, 6 all-NaN grid
// points outside the range of the masked in-place layout
extern "C" int eofx_ctx_fit_info(const eofx_ctx* ctx, double* info3) {
  if (!ctx || !info3) return EOFX_ERR_ARG;
  for (int i = 0; i < 3; ++i) info3[i] = ctx->fit_info[i];
  return EOFX_OK;
}

extern "C" int eofx_ctx_last_iterations(const eofx_ctx* ctx, int* iterations) {
  if (!ctx || !iterations) return EOFX_ERR_ARG;
  *iterations = ctx->last_iters;
  return EOFX_OK;
}

extern "C" int eofx_project_f32(eofx_ctx* ctx, const eofx_mat* m, const float* V, int k, float* out) {
  if (!ctx || !m || !V || !out || k <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int Lo = (int)round_up(k, 32);
  CHK(arena_reserve(ctx, (size_t)(m->p_pad + m->n_pad) * Lo * 4 * 2 + atb_scratch_bytes(m->n_pad, round_up(m->p, ATB_KG), Lo) +
                             (size_t)(m->p + m->n) * k * 4 + (1 << 20)));
  ArenaScope scope(ctx);
  ARENA(float, Vp, (size_t)m->p_pad * Lo);
  ARENA(float, Sn, (size_t)m->n_pad * Lo);
  CHK(import_panel(ctx, V, m->p, k, Vp, m->p_pad, Lo));
  CHK(panel_mul(ctx, m, Vp, Sn, Lo, ctx->prec_final));
  CHK(export_panel(ctx, Sn, m->n, Lo, k, nullptr, out));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

extern "C" int eofx_reconstruct_f32(eofx_ctx* ctx, const float* S, const float* V, int64_t n, int64_t p,
                                    int k, float* out) {
  if (!ctx || !S || !V || !out || k <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  Staged ss, sv;
  CHK(stage_input(ctx, S, (size_t)n * k, ss));
  CHK(stage_input(ctx, V, (size_t)p * k, sv));
  const bool dev = is_device_ptr(out);
  float* tmp = out;
  if (!dev) HIPCHK(hipMalloc((void**)&tmp, (size_t)n * p * sizeof(float)));
  dim3 grid((int)((p + 63) / 64), (int)((n + 63) / 64));
  hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(256), 0, ctx->stream, ss.dev, sv.dev, n, p, k, tmp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !dev)
    e = hipMemcpyAsync(out, tmp, (size_t)n * p * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (!dev) (void)hipFree(tmp);
  if (e != hipSuccess) return set_err(ctx, EOFX_ERR_HIP, "reconstruct failed: %s", hipGetErrorString(e));
  return EOFX_OK;
}

// Gram matrix of a resident matrix through the atb kernel (float32 result):
// side 0: sample space, G[n_pad x n_pad] = X X^T;  side 1: feature space, G[p_pad x p_pad] = X^T X
static int mat_gram(eofx_ctx* ctx, const eofx_mat* m, int side, float* G) {
  if (side == 0) {
    const int64_t npad = m->n_pad;
    CHK(ensure_Xt(ctx, m));
    return launch_atb(ctx, m->Xt, npad, round_up(m->p, ATB_KG), npad, m->Xt, (int)npad, (int)npad, G, ctx->prec_final,
                      m->absmax, reinterpret_cast<const float*>(m->absmax_dev), nullptr, nullptr, nullptr, 1);
  }
  const int64_t ppad = m->p_pad;
  CHK(ensure_X(ctx, m));
  return launch_atb(ctx, m->X, ppad, round_up(m->n, ATB_KG), ppad, m->X, (int)ppad, (int)ppad, G, ctx->prec_final,
                    m->absmax, reinterpret_cast<const float*>(m->absmax_dev), nullptr, nullptr, nullptr, 1);
}
static int sample_gram(eofx_ctx* ctx, const eofx_mat* m, float* G) { return mat_gram(ctx, m, 0, G); }

// ---- sample-space Gram matrix on the fp16 matrix cores (eofx_gram.hpp): planes once, then the tiled NT product ------
constexpr int64_t EOFX_GRAM_MAX_SIDE = 32768;
static int gram_plan_get(eofx_ctx* ctx, int nti, int ntj, bool sym, int nst, const eofx_ctx::GramPlanDev** out) {
  for (auto* g : ctx->gram_plans)
    if (g->nti == nti && g->ntj == ntj && g->sym == sym && g->nst == nst) {
      *out = g;
      return EOFX_OK;
    }
  auto* g = new eofx_ctx::GramPlanDev();
  g->nti = nti;
  g->ntj = ntj;
  g->sym = sym;
  g->nst = nst;
  int S = 0;
  if (const char* ev = std::getenv("EOFX_GRAM_SPLITS")) S = atoi(ev);     // tuning hook (tools/gram_probe.py)
  gram_plan_build(nti, ntj, sym, nst, S, g->pl);
  if (hipMalloc((void**)&g->items, sizeof(GramItem) * g->pl.items.size()) != hipSuccess ||
      hipMalloc((void**)&g->tiles, sizeof(int2) * g->pl.tiles.size()) != hipSuccess ||
      hipMemcpy(g->items, g->pl.items.data(), sizeof(GramItem) * g->pl.items.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(g->tiles, g->pl.tiles.data(), sizeof(int2) * g->pl.tiles.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (g->items) (void)hipFree(g->items);
    if (g->tiles) (void)hipFree(g->tiles);
    delete g;
    return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the work list of the Gram kernel");
  }
  ctx->gram_plans.push_back(g);
  *out = g;
  return EOFX_OK;
}
static int64_t gram_kpad(const eofx_mat* m) { return round_up(m->p, 2 * GR_BK); }
// does the fast route take this matrix?  (sample side up to 32768; the 32-bit row offsets of the LDS-DMA loads)
static bool gram_fast_ok(const eofx_mat* m) {
  return m->n_pad % GR_BM == 0 && m->n_pad <= EOFX_GRAM_MAX_SIDE && (m->X || (m->raw && m->aff) || m->Xt) &&
         gram_kpad(m) * 4 * GR_BM < ((int64_t)1 << 32) && !std::getenv("EOFX_NO_FAST_GRAM");
}
static size_t gram_fast_scratch(eofx_ctx* ctx, const eofx_mat* m) {
  const eofx_ctx::GramPlanDev* g = nullptr;
  const int nt = (int)(m->n_pad / GR_BM);
  if (gram_plan_get(ctx, nt, nt, true, (int)(gram_kpad(m) / GR_BK), &g) != EOFX_OK) return 0;
  return (size_t)m->n_pad * gram_kpad(m) * 4 + (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM * sizeof(float) + 8192;
}
// the fp16 hi / lo planes of the preprocessed matrix (planes_split_kernel), from the raw field through the map (in-place and
// masked in-place matrices) or from the feature-contiguous layout; *a_scale: the exact power of two they are scaled by
static int mat_planes(eofx_ctx* ctx, const eofx_mat* m, _Float16* planes, int64_t kpad, float* a_scale) {
  float sc = 1.f;
  if (m->absmax > 0.f && std::isfinite(m->absmax)) {
    int e;
    (void)std::frexp(m->absmax, &e);
    sc = std::ldexp(1.f, 14 - e);
  }
  *a_scale = sc;
  const int rows_per_wg = 64;
  dim3 grid((unsigned)((kpad + 2047) / 2048), (unsigned)((m->n_pad + rows_per_wg - 1) / rows_per_wg));
  if (!m->X && m->raw && m->aff) {
    hipLaunchKernelGGL(planes_split_kernel, grid, dim3(256), 0, ctx->stream, m->raw, m->raw_ld, m->n, m->p, (const float*)m->aff,
                       m->p_pad, sc, planes, m->n_pad, kpad, rows_per_wg);
  } else {
    CHK(ensure_X(ctx, m));
    hipLaunchKernelGGL(planes_split_kernel, grid, dim3(256), 0, ctx->stream, (const float*)m->X, m->p_pad, m->n, m->p,
                       (const float*)nullptr, (int64_t)0, sc, planes, m->n_pad, kpad, rows_per_wg);
  }
  KCHK();
  return EOFX_OK;
}
// G[n_pad x n_pad] = X' X'^T (float32, full symmetric matrix).  Arena: gram_fast_scratch(m) bytes.
static int mat_gram_fast(eofx_ctx* ctx, const eofx_mat* m, float* G) {
  const int64_t kpad = gram_kpad(m);
  const int nt = (int)(m->n_pad / GR_BM);
  const eofx_ctx::GramPlanDev* g = nullptr;
  CHK(gram_plan_get(ctx, nt, nt, true, (int)(kpad / GR_BK), &g));
  ArenaScope scope(ctx);
  ARENA(_Float16, planes, (size_t)m->n_pad * kpad * 2);
  ARENA(float, part, (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM);
  float sc = 1.f;
  CHK(mat_planes(ctx, m, planes, kpad, &sc));
  hipLaunchKernelGGL(gram_nt_kernel, dim3(g->pl.grid), dim3(512), 0, ctx->stream, (const _Float16*)planes, (const _Float16*)planes,
                     kpad * 4, (const GramItem*)g->items, part, 1.f / (sc * sc));
  KCHK();
  hipLaunchKernelGGL(gram_finish_kernel, dim3(g->pl.T, 16), dim3(256), 0, ctx->stream, (const float*)part, (const int2*)g->tiles, g->pl.S, G,
                     m->n_pad);
  KCHK();
  return EOFX_OK;
}

// ---- wide feature-side products Yp [p_pad x L] = X'^T Zn through the same NT kernel (L in the hundreds: the PCA
// pre-reduction's 1500-column panel took 83 ms through the 128-column streaming tiles, the field re-read 12 times at a third of
// the matrix cores' rate; here: transposed planes of the field once, planes of Zn^T, one MFMA-bound product)
static bool tmul_nt_ok(const eofx_mat* m, int L) {
  const int64_t kpad = round_up(m->n, 2 * GR_BK);
  return L >= 512 && m->p_pad % GR_BM == 0 && (m->X || (m->raw && m->aff)) && kpad * 4 * GR_BM < ((int64_t)1 << 32) &&
         m->n >= 512 && !std::getenv("EOFX_NO_TMUL_NT");
}
// (the partial tiles follow the plan's split-K factor S -- as matmul_nt_scratch and gram_fast_scratch count them; the estimate
// p_pad Lp 4 of round 4 covered S = 1 only: 700 x 3000 with L = 512 plans S = 3, and a fresh context ran out of arena)
static size_t tmul_nt_scratch(eofx_ctx* ctx, const eofx_mat* m, int L) {
  const int64_t kpad = round_up(m->n, 2 * GR_BK), Lp = round_up(L, GR_BM);
  const eofx_ctx::GramPlanDev* g = nullptr;
  if (gram_plan_get(ctx, (int)(m->p_pad / GR_BM), (int)(Lp / GR_BM), false, (int)(kpad / GR_BK), &g) != EOFX_OK) return 0;
  return (size_t)(m->p_pad + Lp) * kpad * 4 + (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM * sizeof(float) + (1 << 16);
}
static int mat_tmul_nt(eofx_ctx* ctx, const eofx_mat* m, const float* Zn, float* Yp, int L) {
  const int64_t kpad = round_up(m->n, 2 * GR_BK), Lp = round_up(L, GR_BM);
  const int nti = (int)(m->p_pad / GR_BM), ntj = (int)(Lp / GR_BM);
  const eofx_ctx::GramPlanDev* g = nullptr;
  CHK(gram_plan_get(ctx, nti, ntj, false, (int)(kpad / GR_BK), &g));
  ArenaScope scope(ctx);
  ARENA(_Float16, pa, (size_t)m->p_pad * kpad * 2);
  ARENA(_Float16, pb, (size_t)Lp * kpad * 2);
  ARENA(float, part, (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM);
  ARENA(unsigned, zmax, 4);
  float sa = 1.f;
  if (m->absmax > 0.f && std::isfinite(m->absmax)) {
    int e;
    (void)std::frexp(m->absmax, &e);
    sa = std::ldexp(1.f, 14 - e);
  }
  // max |Zn| for its power-of-two scale (one small read + a host word)
  const float* zm = amax_get(ctx, Zn);
  if (!zm) {
    HIPCHK(hipMemsetAsync(zmax, 0, sizeof(unsigned), ctx->stream));
    const int64_t total4 = m->n_pad * (int64_t)(L / 4);
    hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))), dim3(256), 0,
                       ctx->stream, Zn, m->n_pad, L, (int64_t)L, zmax);
    KCHK();
    zm = reinterpret_cast<const float*>(zmax);
  }
  float hz = 0.f;
  HIPCHK(hipMemcpyAsync(&hz, zm, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float sb = 1.f;
  if (hz > 0.f && std::isfinite(hz)) {
    int e;
    (void)std::frexp(hz, &e);
    sb = std::ldexp(1.f, 14 - e);
  }
  const dim3 ga((unsigned)(m->p_pad / 64), (unsigned)(kpad / 64)), gb((unsigned)(Lp / 64), (unsigned)(kpad / 64));
  if (!m->X && m->raw && m->aff) {
    hipLaunchKernelGGL(planes_split_t_kernel, ga, dim3(256), 0, ctx->stream, m->raw, m->raw_ld, m->n, m->p, (const float*)m->aff, m->p_pad,
                       sa, pa, kpad);
  } else {
    CHK(ensure_X(ctx, m));
    hipLaunchKernelGGL(planes_split_t_kernel, ga, dim3(256), 0, ctx->stream, (const float*)m->X, m->p_pad, m->n, m->p, (const float*)nullptr,
                       (int64_t)0, sa, pa, kpad);
  }
  KCHK();
  hipLaunchKernelGGL(planes_split_t_kernel, gb, dim3(256), 0, ctx->stream, Zn, (int64_t)L, m->n, (int64_t)L, (const float*)nullptr, (int64_t)0,
                     sb, pb, kpad);
  KCHK();
  hipLaunchKernelGGL(gram_nt_kernel, dim3(g->pl.grid), dim3(512), 0, ctx->stream, (const _Float16*)pa, (const _Float16*)pb, kpad * 4,
                     (const GramItem*)g->items, part, 1.f / (sa * sb));
  KCHK();
  hipLaunchKernelGGL(nt_finish_kernel, dim3(g->pl.T, 16), dim3(256), 0, ctx->stream, (const float*)part, (const int2*)g->tiles, g->pl.S, Yp,
                     (int64_t)L, m->p_pad, L);
  KCHK();
  amax_forget(ctx, Yp);
  return EOFX_OK;
}

// out[rows x Lo] = P[rows x L] Mx[L x Lo] for WIDE operands, on the fp16 matrix cores: both operands as {hi, lo} fp16 planes
// (22 bits, exact power-of-two scales from their maxima), the tiled NT kernel, float32 accumulation -- the arithmetic of
// every other pass.  Mx (float64 on the device) is rounded to float32 first.
static bool matmul_nt_ok(int64_t rows, int L, int Lo) {
  const int64_t kpad = round_up(L, 2 * GR_BK);
  return L >= 512 && Lo >= 256 && Lo % 4 == 0 && L % 4 == 0 && rows >= 1024 && rows % GR_BM == 0 &&
         kpad * 4 * GR_BM < ((int64_t)1 << 32) && !std::getenv("EOFX_NO_MATMUL_NT");
}
static size_t matmul_nt_scratch(eofx_ctx* ctx, int64_t rows, int L, int Lo) {
  const int64_t kpad = round_up(L, 2 * GR_BK), Lop = round_up(Lo, GR_BM);
  const eofx_ctx::GramPlanDev* g = nullptr;
  if (gram_plan_get(ctx, (int)(rows / GR_BM), (int)(Lop / GR_BM), false, (int)(kpad / GR_BK), &g) != EOFX_OK) return 0;
  return (size_t)(rows + Lop) * kpad * 4 + (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM * sizeof(float) + (size_t)L * Lo * 4 + (1 << 16);
}
__global__ __launch_bounds__(256) void f64_to_f32_kernel(const double* __restrict__ a, float* __restrict__ b, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) b[i] = (float)a[i];
}
static int launch_matmul_nt(eofx_ctx* ctx, const float* P, int64_t rows, int L, const double* Mx, int Lo, float* out) {
  const int64_t kpad = round_up(L, 2 * GR_BK), Lop = round_up(Lo, GR_BM);
  const int nti = (int)(rows / GR_BM), ntj = (int)(Lop / GR_BM);
  const eofx_ctx::GramPlanDev* g = nullptr;
  CHK(gram_plan_get(ctx, nti, ntj, false, (int)(kpad / GR_BK), &g));
  ArenaScope scope(ctx);
  ARENA(_Float16, pa, (size_t)rows * kpad * 2);
  ARENA(_Float16, pb, (size_t)Lop * kpad * 2);
  ARENA(float, part, (size_t)g->pl.T * g->pl.S * GR_BM * GR_BM);
  ARENA(float, m32, (size_t)L * Lo);
  ARENA(unsigned, mx, 4);
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)L * Lo + 255) / 256, 4096)), dim3(256), 0, ctx->stream, Mx, m32,
                     (int64_t)L * Lo);
  KCHK();
  HIPCHK(hipMemsetAsync(mx, 0, 2 * sizeof(unsigned), ctx->stream));
  const float* pm = amax_get(ctx, P);
  if (!pm) {
    const int64_t total4 = rows * (int64_t)(L / 4);
    hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))), dim3(256), 0,
                       ctx->stream, P, rows, L, (int64_t)L, mx);
    KCHK();
    pm = reinterpret_cast<const float*>(mx);
  }
  {
    const int64_t total4 = (int64_t)L * (Lo / 4);
    hipLaunchKernelGGL(panel_absmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((total4 + 1023) / 1024, 1024))), dim3(256), 0,
                       ctx->stream, (const float*)m32, (int64_t)L, Lo, (int64_t)Lo, mx + 1);
    KCHK();
  }
  float hp = 0.f, hm = 0.f;
  HIPCHK(hipMemcpyAsync(&hp, pm, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&hm, mx + 1, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  auto pow2 = [](float v) {
    if (!(v > 0.f) || !std::isfinite(v)) return 1.f;
    int e;
    (void)std::frexp(v, &e);
    return std::ldexp(1.f, 14 - e);
  };
  const float sa = pow2(hp), sb = pow2(hm);
  const int rows_per_wg = 64;
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)((kpad + 2047) / 2048), (unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0,
                     ctx->stream, P, (int64_t)L, rows, (int64_t)L, (const float*)nullptr, (int64_t)0, sa, pa, rows, kpad, rows_per_wg);
  KCHK();
  hipLaunchKernelGGL(planes_split_t_kernel, dim3((unsigned)(Lop / 64), (unsigned)(kpad / 64)), dim3(256), 0, ctx->stream, (const float*)m32,
                     (int64_t)Lo, (int64_t)L, (int64_t)Lo, (const float*)nullptr, (int64_t)0, sb, pb, kpad);
  KCHK();
  hipLaunchKernelGGL(gram_nt_kernel, dim3(g->pl.grid), dim3(512), 0, ctx->stream, (const _Float16*)pa, (const _Float16*)pb, kpad * 4,
                     (const GramItem*)g->items, part, 1.f / (sa * sb));
  KCHK();
  hipLaunchKernelGGL(nt_finish_kernel, dim3(g->pl.T, 16), dim3(256), 0, ctx->stream, (const float*)part, (const int2*)g->tiles, g->pl.S, out,
                     (int64_t)Lo, rows, Lo);
  KCHK();
  amax_forget(ctx, out);
  return EOFX_OK;
}

// <a, b> over `count` floats, float64 accumulation, fixed reduction tree
static int device_dot(eofx_ctx* ctx, const float* a, const float* b, int64_t count, double* out) {
  const int nb = 1024;
  CHK(arena_reserve(ctx, nb * sizeof(double) + 4096));
  ArenaScope scope(ctx);
  ARENA(double, part, nb);
  hipLaunchKernelGGL(dotprod_part_kernel, dim3(nb), dim3(256), 0, ctx->stream, a, b, count, part);
  KCHK();
  std::vector<double> hp(nb);
  HIPCHK(hipMemcpyAsync(hp.data(), part, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double t = 0.0;
  for (int i = 0; i < nb; ++i) t += hp[i];
  *out = t;
  return EOFX_OK;
}

extern "C" int eofx_mat_gram_f32(eofx_ctx* ctx, const eofx_mat* m, int side, float* G) {
  if (!ctx || !m || !G || (side != 0 && side != 1)) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int64_t d = side ? m->p_pad : m->n_pad, o = side ? m->n_pad : m->p_pad;
  if (d > (1 << 16)) return set_err(ctx, EOFX_ERR_ARG, "Gram side of %lld is too large", (long long)d);
  // sample side, more features than samples: the MFMA-bound tiled kernel over the fp16 planes (eofx_gram.hpp)
  if (side == 0 && m->p >= m->n && gram_fast_ok(m) && ctx->prec_final == EOFX_PREC_F16X3) {
    CHK(arena_reserve(ctx, gram_fast_scratch(ctx, m) + (1 << 20)));
    return mat_gram_fast(ctx, m, G);
  }
  CHK(arena_reserve(ctx, atb_scratch_bytes(d, o, (int)d) + (1 << 20)));
  return mat_gram(ctx, m, side, G);
}

// G = A_a A_b^T (side 0, [n_pad x n_pad]) or A_a^T A_b (side 1, [p_pad x p_pad]) of two resident matrices of the same shape:
// the off-diagonal blocks of the Hermitian Gram matrix of a complex field Z = A + iB (Z Z^H = (A A^T + B B^T) + i (B A^T - A B^T)).
extern "C" int eofx_mat_cross_gram_f32(eofx_ctx* ctx, const eofx_mat* a, const eofx_mat* b, int side, float* G) {
  if (!ctx || !a || !b || !G || (side != 0 && side != 1)) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (a->n != b->n || a->p != b->p) return set_err(ctx, EOFX_ERR_SHAPE, "the two matrices must have the same shape");
  ENTER(ctx);
  const int64_t d = side ? a->p_pad : a->n_pad, o = side ? a->n_pad : a->p_pad;
  if (d > (1 << 16)) return set_err(ctx, EOFX_ERR_ARG, "Gram side of %lld is too large", (long long)d);
  CHK(arena_reserve(ctx, atb_scratch_bytes(d, o, (int)d) + (1 << 20)));
  const float amax = std::max(a->absmax, b->absmax);
  if (side == 0) {
    CHK(ensure_Xt(ctx, a));
    CHK(ensure_Xt(ctx, b));
    return launch_atb(ctx, a->Xt, d, round_up(a->p, ATB_KG), d, b->Xt, (int)d, (int)d, G, ctx->prec_final, amax,
                      reinterpret_cast<const float*>(b->absmax_dev));
  }
  CHK(ensure_X(ctx, a));
  CHK(ensure_X(ctx, b));
  return launch_atb(ctx, a->X, d, round_up(a->n, ATB_KG), d, b->X, (int)d, (int)d, G, ctx->prec_final, amax,
                    reinterpret_cast<const float*>(b->absmax_dev));
}

extern "C" int eofx_vec_dot_f64(eofx_ctx* ctx, const float* a, const float* b, int64_t count, double* out) {
  if (!ctx || !a || !b || !out || count < 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  return device_dot(ctx, a, b, count, out);
}

// ------------------------------------------------------------------------------------
// cross-covariance path (MCA): matrix-free rSVD of C = X^T Y / (n-1)
// ------------------------------------------------------------------------------------
// sh != nullptr (eofx_crosscov_rsvd_sharded_f32): x and y are this rank's slices of two fields whose feature axes are split
// over the ranks of the context's communicator -- p*_total features over all ranks, this rank's first one at p*_offset of the
// global axis.  Both sides of C = X^T Y are then sharded: every sample-side panel (Y Z, X W: partial sums over a rank's
// features) is all-reduced, every Gram matrix of a feature-side panel too, the two n x n sample-space Gram matrices once;
// everything on the sample side is replicated and computed redundantly (bit-identical on every rank).
struct CrossShard {
  int64_t p1_total, p1_offset, p2_total, p2_offset;
};
static int crosscov_impl(eofx_ctx* ctx, const eofx_mat* x, const eofx_mat* y, int k,
                         int n_oversamples, int n_iter, const float* omega, eofx_sketch_fn omega_fn, void* omega_user, int flip,
                         float* Q1, float* s, float* Q2, float* scores1, float* scores2,
                         float* norm1, float* norm2, double* tsc, const CrossShard* sh = nullptr) {
  if (!ctx || !x || !y || (!omega && !omega_fn) || k <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  if (x->n != y->n)
    return set_err(ctx, EOFX_ERR_SHAPE,
                   "Both data matrices must have the same number of samples but found %lld in the first and %lld in the second.",
                   (long long)x->n, (long long)y->n);
  const bool shd = sh != nullptr;
  const int64_t n = x->n, p1 = x->p, p2 = y->p;
  // the VALID feature counts decide the orientation, the rank and the iteration count, as in the reference (a masked in-place
  // matrix carries its all-NaN grid points as zero columns: p1 / p2 below are the physical widths of the panels)
  const int64_t P1 = shd ? sh->p1_total : (x->masked ? x->p_valid : p1), P2 = shd ? sh->p2_total : (y->masked ? y->p_valid : p2),
                r = std::min(P1, P2);
  auto reduce_panel = [&](float* Pn, int64_t count) -> int {      // a sample-side panel: sum of the ranks' partial sums
    if (!shd) return EOFX_OK;
    amax_forget(ctx, Pn);
    return comm_allreduce(ctx, Pn, count, 0, 0);
  };
  GramReduce reduce_gram;
  if (shd) reduce_gram = [&](double* Gm, int64_t count) { return comm_allreduce(ctx, Gm, count, 1, 0); };
  if (k > r)
    return set_err(ctx, EOFX_ERR_RANK,
                   "n_modes must be less than or equal to the rank of the dataset (rank = %lld).", (long long)r);
  const int l_req = k + n_oversamples;
  const int l = (int)std::min<int64_t>(l_req, r);
  if (l > EOFX_MAX_SKETCH) return set_err(ctx, EOFX_ERR_ARG, "sketch width %d > %d is not supported", l, EOFX_MAX_SKETCH);
  if (l != l_req) return set_err(ctx, EOFX_ERR_ARG, "sketch wider than rank not supported on the cross path");
  if (n_iter < 0) n_iter = rsvd_auto_iters(k, P1, P2);
  const int L = (int)round_up(l, 32);
  const bool transposed = P1 < P2;  // C is (p1 x p2): sklearn transposes when rows < cols
  const int64_t npad = x->n_pad;
  // The total squared covariance needs the two sample-space Gram matrices (below).  With them resident the power
  // iterations need not touch the fields at all: A = Ft^T Fs (Ft the field on the tall side), A Z = Ft^T (Fs Z) and
  // (A^T A) Z = Fs^T Gt (Fs Z), so in terms of T = Fs Z (n x l) one iteration is T <- Gs (Gt T) -- two n x n x l products
  // (tens of microseconds) instead of four passes over the fields.  Same subspace in exact arithmetic as scikit-learn's
  // iteration on C (range of (C C^T)^q C Omega); the range basis and the projection that decides the singular values are
  // still computed from the fields themselves.  Taken when the TSC is asked for and both fields are wider than long.
  bool gram_route = tsc && n < P1 && n < P2 && gram_fast_ok(x) && gram_fast_ok(y) && !std::getenv("EOFX_CROSS_NO_GRAM");
  size_t need = rsvd_scratch_bytes(std::max(x->p_pad, y->p_pad), std::max(x->p_pad, y->p_pad), l, k) +
                (size_t)npad * L * 4 * 2 + atb_scratch_bytes(npad, std::max(x->p_pad, y->p_pad), L);
  if (tsc) need += (size_t)npad * npad * 4 * 2 + (1 << 20);
  const size_t need_gram = gram_fast_ok(x) && gram_fast_ok(y)
                               ? std::max(gram_fast_scratch(ctx, x), gram_fast_scratch(ctx, y)) + (size_t)npad * L * 4 * 2 + atb_scratch_bytes(npad, npad, L)
                               : 0;
  const size_t need_plain = tsc ? atb_scratch_bytes(npad, std::max(x->p_pad, y->p_pad), (int)npad) : 0;
  if (shd) {
    // The Gram route is a collective decision (a rank's slice may not qualify), and so is an error of this rank before the first
    // data collective (growing the arena): one vote -- 0 go on with the Gram route, 1 without it, 2 some rank failed.  The arena is
    // sized for either route up front (+ room for the null-mode repair of a sharded factor).
    const size_t Lo_ = (size_t)round_up(k, 32), pp_ = (size_t)std::max(x->p_pad, y->p_pad);
    need += pp_ * Lo_ * 4 + (size_t)(gram_parts((int64_t)pp_, (int)Lo_) + 4) * Lo_ * Lo_ * 8 + (64 << 10) + std::max(need_gram, need_plain);
    const int rc_local = arena_reserve(ctx, need);
    const std::string err_local = rc_local != EOFX_OK ? ctx->err : std::string();
    int worst = 0;
    CHK(comm_vote(ctx, rc_local != EOFX_OK ? 2 : gram_route ? 0 : 1, &worst));
    if (worst == 2)
      return rc_local != EOFX_OK ? (ctx->err = err_local, rc_local) : set_err(ctx, EOFX_ERR_HIP, "the sharded cross-covariance fit failed on another rank");
    gram_route = worst == 0;
  } else {
    need += gram_route ? need_gram : need_plain;
    CHK(arena_reserve(ctx, need));
  }
  ArenaScope scope(ctx);
  ARENA(float, Tn, (size_t)npad * L);
  float *Gx = nullptr, *Gy = nullptr;
  if (tsc) {
    Gx = arena_alloc<float>(ctx, (size_t)npad * npad);
    Gy = arena_alloc<float>(ctx, (size_t)npad * npad);
    if (!Gx || !Gy) return set_err(ctx, EOFX_ERR_NOMEM, "internal: arena exhausted (Gram matrices)");
  }
  if (gram_route) {
    CHK(mat_gram_fast(ctx, x, Gx));
    if (shd) CHK(comm_allreduce(ctx, Gx, npad * npad, 0, 0));      // X X^T = sum over the slices
    CHK(mat_gram_fast(ctx, y, Gy));
    if (shd) CHK(comm_allreduce(ctx, Gy, npad * npad, 0, 0));
  }
  // C   Z = X^T (Y Z);   C^T W = Y^T (X W)     (scaling by 1/(n-1) is applied to s at the end)
  auto C_mul = [&](const float* z2, float* out1, int LL, int pr) {
    CHK(panel_mul(ctx, y, z2, Tn, LL, pr));
    CHK(reduce_panel(Tn, npad * LL));
    return panel_tmul(ctx, x, Tn, out1, LL, pr);
  };
  auto Ct_mul = [&](const float* z1, float* out2, int LL, int pr) {
    CHK(panel_mul(ctx, x, z1, Tn, LL, pr));
    CHK(reduce_panel(Tn, npad * LL));
    return panel_tmul(ctx, y, Tn, out2, LL, pr);
  };
  LinOp op;
  if (transposed)
    op = {p2, p1, y->p_pad, x->p_pad, Ct_mul, C_mul};  // A = C^T (p2 x p1)
  else
    op = {p1, p2, x->p_pad, y->p_pad, C_mul, Ct_mul};  // A = C   (p1 x p2)
  if (shd) {
    op.reduce_tall_gram = reduce_gram;
    op.reduce_small_gram = reduce_gram;
    op.rule_tall_pad = round_up(transposed ? P2 : P1, ATB_BM);
  }
  // the sketch is asked for only now: on the Gram route ~18 ms of matrix work are already queued behind which the caller's
  // generator (scikit-learn's legacy stream: ~5 ms for 129 600 x 30 deviates) finishes unnoticed
  if (!omega) {
    omega = omega_fn(omega_user);
    if (!omega) return set_err(ctx, EOFX_ERR_ARG, "the sketch callback returned no matrix");
  }
  std::vector<float> om_eye;
  if (l == r && !shd) {   // full-width sketch: identity (see eofx_rsvd_f32); a sharded caller hands over its rows of it
    if ((transposed ? x : y)->masked)
      return set_err(ctx, EOFX_ERR_ARG, "a sketch as wide as the rank on a masked in-place matrix is not supported (compact the field)");
    om_eye.assign((size_t)op.small * l, 0.f);
    for (int64_t i = 0; i < l; ++i) om_eye[(size_t)i * l + i] = 1.f;
    omega = om_eye.data();
  }
  RsvdOut ro;
  if (gram_route) {
    const eofx_mat* ft = transposed ? y : x;     // field on the tall side of A
    const eofx_mat* fs = transposed ? x : y;
    const float* Gt = transposed ? Gy : Gx;
    const float* Gs = transposed ? Gx : Gy;
    RangeFinder range = [&](const float* Zs, float* Yt, int LL) -> int {
      ArenaScope inner(ctx);
      ARENA(float, T2, (size_t)npad * LL);
      ARENA(double, Gd, (size_t)LL * LL);
      auto orth = [&](const float* in, float* outp) -> int {     // Cholesky-QR of an n x l panel (float64 Gram matrix)
        CHK(launch_gram(ctx, in, npad, LL, Gd));
        return launch_cholqr(ctx, in, npad, LL, l, Gd, outp);
      };
      // exact-f32 MFMA products with the symmetric Gram matrices (G^T T = G T): 100 MB each, tens of microseconds
      auto gmul = [&](const float* Gm, const float* in, float* outp) -> int {
        return launch_atb(ctx, Gm, npad, npad, npad, in, LL, LL, outp, EOFX_PREC_F32);
      };
      CHK(panel_mul(ctx, fs, Zs, Tn, LL, ctx->prec_power));      // T = Fs Z
      CHK(reduce_panel(Tn, npad * LL));
      for (int it = 0; it < n_iter; ++it) {
        CHK(orth(Tn, T2));
        CHK(gmul(Gt, T2, Tn));
        CHK(orth(Tn, T2));
        CHK(gmul(Gs, T2, Tn));
      }
      CHK(orth(Tn, T2));
      return panel_tmul(ctx, ft, T2, Yt, LL, ctx->prec_power);   // Yt = Ft^T T
    };
    CHK(rsvd_core(ctx, op, k, l, n_iter, omega, ro, nullptr, &range));
  } else {
    CHK(rsvd_core(ctx, op, k, l, n_iter, omega, ro));
  }
  // (more modes than the cross-covariance has rank)
  if (shd)
    CHK(fix_null_modes(ctx, ro, transposed ? p2 : p1, transposed ? p1 : p2, k, &reduce_gram, transposed ? P2 : P1, &reduce_gram,
                       transposed ? P1 : P2));
  else
    CHK(fix_null_modes(ctx, ro, transposed ? p2 : p1, transposed ? p1 : p2, k));
  const float* Q1p = transposed ? ro.Svec : ro.Tvec;  // left vectors of C  (p1)
  const float* Q2p = transposed ? ro.Tvec : ro.Svec;  // right vectors of C (p2)
  std::vector<double> sign;
  if (flip) CHK(sign_rule(ctx, Q2p, p2, ro.Lo, k, sign, shd));
  const double* sg = flip ? sign.data() : nullptr;
  CHK(export_panel(ctx, Q1p, p1, ro.Lo, k, sg, Q1));
  CHK(export_panel(ctx, Q2p, p2, ro.Lo, k, sg, Q2));
  if (s) {
    std::vector<float> hs(k);
    for (int j = 0; j < k; ++j) hs[j] = (float)(ro.s[j] / (double)(n - 1));
    HIPCHK(hipMemcpy(s, hs.data(), sizeof(float) * k, hipMemcpyDefault));
  }
  // scores and norms (cpcca.py:204-208)
  if (scores1 || scores2 || norm1 || norm2) {
    ARENA(float, Sn, (size_t)npad * ro.Lo);
    ARENA(double, Gs, (size_t)ro.Lo * ro.Lo);
    for (int which = 0; which < 2; ++which) {
      const eofx_mat* mm = which ? y : x;
      const float* Qp = which ? Q2p : Q1p;
      float* sc = which ? scores2 : scores1;
      float* nr = which ? norm2 : norm1;
      if (!sc && !nr) continue;
      CHK(panel_mul(ctx, mm, Qp, Sn, ro.Lo, ctx->prec_final));
      CHK(reduce_panel(Sn, npad * ro.Lo));
      CHK(export_panel(ctx, Sn, n, ro.Lo, k, sg, sc));
      if (nr) {
        CHK(launch_gram(ctx, Sn, npad, ro.Lo, Gs));
        std::vector<double> hg((size_t)ro.Lo * ro.Lo);
        HIPCHK(hipMemcpyAsync(hg.data(), Gs, sizeof(double) * hg.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<float> hn(k);
        for (int j = 0; j < k; ++j) hn[j] = (float)std::sqrt(hg[(size_t)j * ro.Lo + j]);
        HIPCHK(hipMemcpy(nr, hn.data(), sizeof(float) * k, hipMemcpyDefault));
      }
    }
  }
  // total squared covariance ||X^T Y||_F^2/(n-1)^2 = <X X^T, Y Y^T>/(n-1)^2 : two n x n Grams (cpcca.py:197,991-1000)
  if (tsc) {
    if (!gram_route) {
      CHK(sample_gram(ctx, x, Gx));
      if (shd) CHK(comm_allreduce(ctx, Gx, npad * npad, 0, 0));
      CHK(sample_gram(ctx, y, Gy));
      if (shd) CHK(comm_allreduce(ctx, Gy, npad * npad, 0, 0));
    }
    double t = 0.0;
    CHK(device_dot(ctx, Gx, Gy, npad * npad, &t));
    *tsc = t / ((double)(n - 1) * (double)(n - 1));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

extern "C" int eofx_crosscov_rsvd_f32(eofx_ctx* ctx, const eofx_mat* x, const eofx_mat* y, int k,
                                      int n_oversamples, int n_iter, const float* omega, int flip,
                                      float* Q1, float* s, float* Q2, float* scores1, float* scores2,
                                      float* norm1, float* norm2, double* tsc) {
  if (!omega) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  return crosscov_impl(ctx, x, y, k, n_oversamples, n_iter, omega, nullptr, nullptr, flip, Q1, s, Q2, scores1, scores2, norm1, norm2, tsc);
}
extern "C" int eofx_crosscov_rsvd_lazy_f32(eofx_ctx* ctx, const eofx_mat* x, const eofx_mat* y, int k,
                                           int n_oversamples, int n_iter, eofx_sketch_fn omega_fn, void* omega_user, int flip,
                                           float* Q1, float* s, float* Q2, float* scores1, float* scores2,
                                           float* norm1, float* norm2, double* tsc) {
  if (!omega_fn) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  return crosscov_impl(ctx, x, y, k, n_oversamples, n_iter, nullptr, omega_fn, omega_user, flip, Q1, s, Q2, scores1, scores2, norm1, norm2, tsc);
}

extern "C" int eofx_crosscov_rsvd_sharded_f32(eofx_ctx* ctx, const eofx_mat* x, const eofx_mat* y, int64_t p1_total,
                                              int64_t p1_offset, int64_t p2_total, int64_t p2_offset, int k, int n_oversamples,
                                              int n_iter, const float* omega, int flip, float* Q1, float* s, float* Q2,
                                              float* scores1, float* scores2, float* norm1, float* norm2, double* tsc) {
  if (!ctx || !x || !y || !omega) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached (eofx_ctx_comm_init_rccl / eofx_ctx_comm_set_callback)");
  // (a masked in-place slice counts its VALID features on the global axis; its zero columns are not part of it)
  if (p1_offset < 0 || p2_offset < 0 || p1_offset + (x->masked ? x->p_valid : x->p) > p1_total ||
      p2_offset + (y->masked ? y->p_valid : y->p) > p2_total)
    return set_err(ctx, EOFX_ERR_ARG, "the slice [offset, offset + p) lies outside the global feature axis");
  if (is_device_ptr(omega)) return set_err(ctx, EOFX_ERR_ARG, "omega must be a host pointer");
  const CrossShard sh{p1_total, p1_offset, p2_total, p2_offset};
  return crosscov_impl(ctx, x, y, k, n_oversamples, n_iter, omega, nullptr, nullptr, flip, Q1, s, Q2, scores1, scores2, norm1, norm2,
                       tsc, &sh);
}

// all-reduce of a small HOST vector over the context's communicator (the global facts of a sharded preprocess: feature counts,
// sample votes, variances): staged through the arena, in stream order with the kernels already queued.  op: 0 sum, 1 max, 2 min
extern "C" int eofx_ctx_comm_allreduce_f64(eofx_ctx* ctx, double* host_buf, int64_t count, int op) {
  if (!ctx || !host_buf || count <= 0 || op < 0 || op > 2) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)count * sizeof(double) + 4096));
  ArenaScope scope(ctx);
  ARENA(double, d, (size_t)count);
  HIPCHK(hipMemcpyAsync(d, host_buf, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  CHK(comm_allreduce(ctx, d, count, 1, op));
  HIPCHK(hipMemcpyAsync(host_buf, d, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// Hilbert transform stage (hilbert_transform.py:40-114) and complex-panel helpers
// ------------------------------------------------------------------------------------
#define FFTCHK(expr)                                                                            \
  do {                                                                                          \
    hipfftResult _r = (expr);                                                                   \
    if (_r != HIPFFT_SUCCESS) {                                                                 \
      rc = set_err(ctx, EOFX_ERR_HIP, "%s failed with hipfftResult %d (%s:%d)", #expr, (int)_r, \
                   __FILE__, __LINE__);                                                         \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)

// impulse response of Im(analytic-signal filter) of period N at integer lag d (0 at d = 0 mod N):
// N even: (2/N) cot(pi d/N) for odd d, 0 for even d;  N odd: (cot(pi d/N) - (-1)^d / sin(pi d/N)) / N
static double hilbert_kappa(int64_t N, int64_t d) {
  int64_t m = d % N;
  if (m < 0) m += N;
  if (m == 0) return 0.0;
  const double x = M_PI * (double)d / (double)N;
  if (N % 2 == 0) return (m % 2) ? (2.0 / (double)N) / std::tan(x) : 0.0;
  const double sgn = (std::llabs(d) % 2) ? -1.0 : 1.0;
  return (1.0 / std::tan(x) - sgn / std::sin(x)) / (double)N;
}

static int get_fft_plans(eofx_ctx* ctx, int64_t P, int64_t ldw, int64_t nh, int64_t batch, hipfftHandle& pf,
                         hipfftHandle& pb) {
  for (auto& e : ctx->fft_plans)
    if (e.first.first == P && e.first.second == batch) {
      pf = (hipfftHandle)e.second.first;
      pb = (hipfftHandle)e.second.second;
      return EOFX_OK;
    }
  int len = (int)P, rembed = (int)ldw, cembed = (int)nh;
  if (hipfftPlanMany(&pf, 1, &len, &rembed, 1, (int)ldw, &cembed, 1, (int)nh, HIPFFT_R2C, (int)batch) != HIPFFT_SUCCESS)
    return set_err(ctx, EOFX_ERR_HIP, "hipfftPlanMany(R2C, P=%lld, batch=%lld) failed", (long long)P, (long long)batch);
  if (hipfftPlanMany(&pb, 1, &len, &cembed, 1, (int)nh, &rembed, 1, (int)ldw, HIPFFT_C2R, (int)batch) != HIPFFT_SUCCESS) {
    hipfftDestroy(pf);
    return set_err(ctx, EOFX_ERR_HIP, "hipfftPlanMany(C2R, P=%lld, batch=%lld) failed", (long long)P, (long long)batch);
  }
  hipfftSetStream(pf, ctx->stream);
  hipfftSetStream(pb, ctx->stream);
  ctx->fft_plans.push_back({{P, batch}, {(void*)pf, (void*)pb}});
  return EOFX_OK;
}

// forward transform of a power-of-two length, float64, in place (host; filter table of the one-kernel Hilbert route)
static void host_fft_f64(std::vector<std::complex<double>>& a) {
  const size_t N = a.size();
  for (size_t i = 1, j = 0; i < N; ++i) {
    size_t bit = N >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= N; len <<= 1) {
    std::vector<std::complex<double>> w(len / 2);
    for (size_t k = 0; k < len / 2; ++k) {
      const double ang = -2.0 * M_PI * (double)k / (double)len;
      w[k] = std::complex<double>(std::cos(ang), std::sin(ang));
    }
    for (size_t i = 0; i < N; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> x = a[i + k], y = a[i + k + len / 2] * w[k];
        a[i + k] = x + y;
        a[i + k + len / 2] = x - y;
      }
  }
}

// circular length of the convolution behind the Hilbert stage: power of two >= 2 n, and >= 1024 so that half of it
// covers the padded series length (n_pad = round_up(n, 512) <= P / 2: the one-kernel route writes samples [0, P / 2));
// that route (eofx_hfft.hpp) holds 2^14 complex points in LDS: circular lengths up to 2^15
static int64_t hilbert_length(int64_t n, int* log2_out) {
  int L = 10;
  while (((int64_t)1 << L) < 2 * n) ++L;
  if (log2_out) *log2_out = L;
  return (int64_t)1 << L;
}
static const int HFFT_MAX_LOG2 = 15;   // 2^14 complex points in LDS: two features up to P = 2^14, one feature (even / odd samples) at P = 2^15

// The four correction vectors of the padded transform, float64 [4][n] (threaded):
// u1 = K_pre e_rev, u2 = K_pos e, u3 = (K_pre + K_pos) 1, u4 = K_pre (t - n) + K_pos (t + n)
// with K_pre[t][s] = kappa((n + t) - s), K_pos[t][s] = kappa((n + t) - (2n + s)), e[t] = exp(-t / n / decay)
static void hilbert_pad_vectors(int64_t n, int64_t N, double decay, std::vector<double>& u) {
  std::vector<double> e((size_t)n), kap((size_t)(4 * n + 1));
  for (int64_t t = 0; t < n; ++t) e[t] = std::exp(-(double)t / (double)n / decay);
  for (int64_t d = -2 * n; d <= 2 * n; ++d) kap[(size_t)(d + 2 * n)] = hilbert_kappa(N, d);
  u.assign((size_t)4 * n, 0.0);
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t t = lo; t < hi; ++t) {
      double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
      for (int64_t sidx = 0; sidx < n; ++sidx) {
        const double kp = kap[(size_t)((n + t - sidx) + 2 * n)];
        const double kq = kap[(size_t)((t - n - sidx) + 2 * n)];
        a1 += kp * e[n - 1 - sidx];
        a2 += kq * e[sidx];
        a3 += kp + kq;
        a4 += kp * (double)(sidx - n) + kq * (double)(sidx + n);
      }
      u[(size_t)t] = a1; u[(size_t)(n + t)] = a2; u[(size_t)(2 * n + t)] = a3; u[(size_t)(3 * n + t)] = a4;
    }
  };
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(16, n / 256));
  std::vector<std::thread> th;
  const int64_t step = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) th.emplace_back(work, t * step, std::min<int64_t>(n, (t + 1) * step));
  for (auto& t : th) t.join();
}

// kernel spectrum and correction vectors for series length n (cached per context)
static int get_hilbert_setup(eofx_ctx* ctx, int64_t n, int padding, double decay, int64_t P, bool fused,
                             const cfloat** chat_out, const float** hperm_out, const float** u_out) {
  for (auto& h : ctx->hsetups)
    if (h.n == n && h.padding == padding && h.decay == decay && h.P == P && (fused ? h.hperm != nullptr : h.chat != nullptr)) {
      *chat_out = (const cfloat*)h.chat;
      *hperm_out = h.hperm;
      *u_out = h.u;
      return EOFX_OK;
    }
  const int64_t N = padding ? 3 * n : n;  // period of the Hilbert kernel
  const int64_t nh = P / 2 + 1, ldw = P + 2;
  eofx_ctx::HilbertSetup hs;
  hs.n = n; hs.P = P; hs.padding = padding; hs.decay = decay;
  int L = 0;
  while (((int64_t)1 << L) < P) ++L;
  if (fused) {
    // the kernel is real and odd, so its spectrum is i h[k] with h real: one float per frequency, stored in the order
    // the forward stages leave the spectrum in LDS; 1/P of the unnormalised inverse folded in
    std::vector<std::complex<double>> c((size_t)P, 0.0);
    for (int64_t d = -(n - 1); d <= n - 1; ++d) c[(size_t)((d % P + P) % P)] = hilbert_kappa(N, d) / (double)P;
    host_fft_f64(c);
    std::vector<float> hp((size_t)P);
    if (L <= 14) {
      for (int64_t pos = 0; pos < P; ++pos) hp[(size_t)pos] = (float)c[(size_t)hfft::position_frequency(L, pos)].imag();
    } else {
      // one feature per workgroup through the half-length transform (M = P/2 positions): the two tables of the
      // split-filter-merge step, hm = h[k] - h[M-k] and hp2 = (h[k] + h[M-k]) / 2  (h[M] = 0)
      const int64_t M = P / 2;
      for (int64_t pos = 0; pos < M; ++pos) {
        const int64_t k = hfft::position_frequency(L - 1, pos);
        const double hk = c[(size_t)k].imag(), hkp = k ? c[(size_t)(M - k)].imag() : 0.0;
        hp[(size_t)pos] = (float)(hk - hkp);
        hp[(size_t)(M + pos)] = (float)(0.5 * (hk + hkp));
      }
    }
    HIPCHK(hipMalloc((void**)&hs.hperm, sizeof(float) * P));
    HIPCHK(hipMemcpy(hs.hperm, hp.data(), sizeof(float) * P, hipMemcpyHostToDevice));
  } else {
    // circular embedding of the Toeplitz kernel, lags -(n-1) .. n-1, scaled by 1/P (unnormalised inverse)
    std::vector<float> c((size_t)ldw, 0.f);
    for (int64_t d = -(n - 1); d <= n - 1; ++d)
      c[(size_t)((d % P + P) % P)] = (float)(hilbert_kappa(N, d) / (double)P);
    float* cdev = nullptr;
    HIPCHK(hipMalloc((void**)&cdev, sizeof(float) * ldw));
    HIPCHK(hipMalloc(&hs.chat, sizeof(cfloat) * nh));
    HIPCHK(hipMemcpy(cdev, c.data(), sizeof(float) * ldw, hipMemcpyHostToDevice));
    hipfftHandle pf, pb;
    CHK(get_fft_plans(ctx, P, ldw, nh, 1, pf, pb));
    if (hipfftExecR2C(pf, cdev, (hipfftComplex*)hs.chat) != HIPFFT_SUCCESS)
      return set_err(ctx, EOFX_ERR_HIP, "hipfftExecR2C failed (kernel spectrum)");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(cdev);
  }
  if (padding) {
    std::vector<double> ud;
    hilbert_pad_vectors(n, N, decay, ud);
    std::vector<float> hu((size_t)4 * n + 4, 0.f);   // (+ the four means over the samples, one-kernel route)
    for (int64_t t = 0; t < n; ++t)
      for (int k = 0; k < 4; ++k) {
        const float v = (float)ud[(size_t)k * n + t];
        if (fused) hu[(size_t)(4 * t + k)] = v;      // the one-kernel route reads the four vectors interleaved per sample
        else hu[(size_t)k * n + t] = v;
      }
    if (fused)       // means of the (float32-rounded) vectors: the kernel adds coefficient * mean to the output's mean
      for (int k = 0; k < 4; ++k) {
        double m = 0.0;
        for (int64_t t = 0; t < n; ++t) m += (double)hu[(size_t)(4 * t + k)];
        hu[(size_t)(4 * n + k)] = (float)(m / (double)n);
      }
    HIPCHK(hipMalloc((void**)&hs.u, sizeof(float) * (4 * n + 4)));
    HIPCHK(hipMemcpy(hs.u, hu.data(), sizeof(float) * (4 * n + 4), hipMemcpyHostToDevice));
  }
  ctx->hsetups.push_back(hs);
  *chat_out = (const cfloat*)hs.chat;
  *hperm_out = hs.hperm;
  *u_out = hs.u;
  return EOFX_OK;
}

constexpr int HILBERT_SQ_PARTIALS = 16384;
// launch of the one-kernel route for one plan (L = log2 of the circular length)
template <int L, int MODE>
static hipError_t launch_hilbert_fused(eofx_ctx* ctx, const float* Xt, int64_t n_pad, int64_t n, int64_t p, int padding,
                                       const float* hperm, const float* u, float* Bt, float* At, unsigned* bmax,
                                       unsigned* amax, const float* aff = nullptr, int64_t aff_ld = 0, const float* oscale = nullptr,
                                       double* sqpart = nullptr) {
  using PL = hfft::plan<L>;
  auto kern = hfft::hilbert_fft_kernel<L, MODE>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL::lds);
  if (e != hipSuccess) return e;
  const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>((int64_t)(163840 / PL::lds), 2048 / PL::WG));
  const int64_t groups = MODE ? p : (p + 1) / 2;
  static int cu_count = 0;
  if (cu_count <= 0) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, ctx->device);
    if (e != hipSuccess) return e;
    cu_count = prop.multiProcessorCount;
  }
  const int grid = (int)std::min<int64_t>(groups, (int64_t)cu_count * per_cu);
  // (sqpart: one float64 partial per wave, grid * NW <= CUs * 2048 / 64 of them -- HILBERT_SQ_PARTIALS bounds it)
  if (sqpart && (int64_t)grid * PL::NW > HILBERT_SQ_PARTIALS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PL::WG), PL::lds, ctx->stream, Xt, n_pad, (int)n, p, padding, hperm, u, Bt, At,
                     bmax, amax, aff, aff_ld, oscale, sqpart);
  return hipGetLastError();
}

extern "C" int eofx_hilbert_f32(eofx_ctx* ctx, const eofx_mat* a, int padding, double decay_factor,
                                eofx_mat** out_imag, eofx_mat** out_real) {
  if (!ctx || !a || !out_imag) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (padding && !(decay_factor > 0.0)) return set_err(ctx, EOFX_ERR_ARG, "decay_factor must be positive");
  ENTER_EXCLUSIVE(ctx);
  const int64_t n = a->n, p = a->p, n_pad = a->n_pad, p_pad = a->p_pad;
  int L = 0;
  const int64_t P = hilbert_length(n, &L);  // circular length: power of two >= 2n
  const bool fused = L <= HFFT_MAX_LOG2 && n_pad <= P / 2;   // the whole stage in one kernel (eofx_hfft.hpp); longer series: hipFFT route
  const int64_t nh = P / 2 + 1, ldw = P + 2;
  const cfloat* chat = nullptr;
  const float* hperm = nullptr;
  const float* u = nullptr;
  CHK(get_hilbert_setup(ctx, n, padding ? 1 : 0, padding ? decay_factor : 0.0, P, fused, &chat, &hperm, &u));
  // An in-place input (the raw field through its Scaler map, no written layout) keeps the stage lean: its
  // sample-contiguous copy exists only while the kernel runs, and the imaginary part is produced in that layout alone
  // (eofx_rsvd_c64 streams the pair [raw field, Im^T] directly; any other consumer gets the feature-contiguous layout on
  // demand through ensure_X).
  // (round 5: a MASKED in-place field -- all-NaN grid points kept as zero columns, sanitizer.py:80-126 -- takes the lean route
  // too: its series are zeros, so is their transform, and Im carries the same zero columns as a plain written matrix)
  const bool lean = a->raw && a->aff && !a->X;
  // round 5: the statistics pass of the in-place preprocess already wrote the RAW field in the sample-contiguous layout
  // (eofx_ctx_set_sample_raw, colstats4_tr_kernel); the one-kernel route reads it through the Scaler map -- the same
  // expression apply_kernel would have written, so the transform is bit-identical to the transient-copy route below
  const bool from_rawT = lean && a->rawT && !a->Xt && fused && L <= 14 && !out_real;
  const bool xt_transient = lean && !a->Xt && !from_rawT;
  if (!from_rawT) CHK(ensure_Xt(ctx, a));
  // features per FFT batch: real series + half spectrum of about 3 GB together
  int64_t Fc = std::max<int64_t>(1, std::min<int64_t>(p, (int64_t)(3.0e9 / (4.0 * (double)ldw + 8.0 * (double)nh))));
  eofx_mat *mi = nullptr, *mr = nullptr;
  CHK(mat_alloc(ctx, n, p, &mi, !lean));
  int rc = EOFX_OK;
  if (out_real) rc = mat_alloc(ctx, n, p, &mr, !lean);
  float* work = nullptr;
  cfloat* spec = nullptr;
  float* coef = nullptr;
  hipfftHandle plan_f = 0, plan_b = 0, tail_f = 0, tail_b = 0;
  const size_t work_bytes = (size_t)Fc * ldw * sizeof(float), spec_bytes = (size_t)Fc * nh * sizeof(cfloat);
  const size_t coef_bytes = (size_t)Fc * 4 * sizeof(float);
  if (rc != EOFX_OK) goto done;
  if (fused) {
    hipError_t e = hipSuccess;
    float* At = mr ? mr->Xt : nullptr;
    unsigned* amax = mr ? mr->absmax_dev : nullptr;
    switch (L) {
#define EOFX_HF(LL) \
  case LL: e = launch_hilbert_fused<LL, 0>(ctx, from_rawT ? a->rawT : a->Xt, n_pad, n, p, padding ? 1 : 0, hperm, u, mi->Xt, At, mi->absmax_dev, amax, \
                                           from_rawT ? a->aff : nullptr, a->p_pad, a->masked ? a->aff + 2 * a->p_pad : nullptr); break;
      EOFX_HF(10) EOFX_HF(11) EOFX_HF(12) EOFX_HF(13) EOFX_HF(14)
#undef EOFX_HF
      case 15:   // 8193 .. 16384 samples: one feature per workgroup, half-length transform of its even / odd samples
        e = launch_hilbert_fused<14, 1>(ctx, a->Xt, n_pad, n, p, padding ? 1 : 0, hperm, u, mi->Xt, At, mi->absmax_dev, amax);
        break;
      default: e = hipErrorInvalidValue;
    }
    if (e != hipSuccess) {
      rc = set_err(ctx, EOFX_ERR_HIP, "hilbert stage failed: %s", hipGetErrorString(e));
      goto done;
    }
  } else if (pool_malloc(ctx, (void**)&work, work_bytes) != hipSuccess ||
      pool_malloc(ctx, (void**)&spec, spec_bytes) != hipSuccess ||
      pool_malloc(ctx, (void**)&coef, coef_bytes) != hipSuccess) {
    rc = set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the FFT work buffers");
    goto done;
  }
  if (!fused) {
    rc = get_fft_plans(ctx, P, ldw, nh, Fc, plan_f, plan_b);
    if (rc != EOFX_OK) goto done;
    const int64_t tail = p % Fc;
    if (tail) {
      rc = get_fft_plans(ctx, P, ldw, nh, tail, tail_f, tail_b);
      if (rc != EOFX_OK) goto done;
    }
    for (int64_t f0 = 0; f0 < p; f0 += Fc) {
      const int64_t fc = std::min(Fc, p - f0);
      const bool full = fc == Fc;
      hipLaunchKernelGGL(hilbert_pack_kernel, dim3((int)fc), dim3(256), 0, ctx->stream, a->Xt, n_pad, n, f0,
                         padding ? 1 : 0, work, ldw, P, coef);
      FFTCHK(hipfftExecR2C(full ? plan_f : tail_f, work, (hipfftComplex*)spec));
      const int64_t total = fc * nh;
      hipLaunchKernelGGL(hilbert_filter_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 16384)),
                         dim3(256), 0, ctx->stream, spec, chat, nh, total);
      FFTCHK(hipfftExecC2R(full ? plan_b : tail_b, (hipfftComplex*)spec, work));
      hipLaunchKernelGGL(hilbert_unpack_kernel, dim3((int)fc), dim3(256), 0, ctx->stream, work, ldw, n, n_pad,
                         f0, padding ? 1 : 0, coef, u, a->Xt, mi->Xt, mr ? mr->Xt : nullptr, mi->absmax_dev,
                         mr ? mr->absmax_dev : nullptr);
    }
  }
  {
    // zero the padding feature rows, then build the feature-contiguous layout by transposition
    const size_t pad_bytes = (size_t)(p_pad - p) * n_pad * sizeof(float);
    if (pad_bytes) {
      if (hipMemsetAsync(mi->Xt + p * n_pad, 0, pad_bytes, ctx->stream) != hipSuccess ||
          (mr && hipMemsetAsync(mr->Xt + p * n_pad, 0, pad_bytes, ctx->stream) != hipSuccess)) {
        rc = set_err(ctx, EOFX_ERR_HIP, "memset failed");
        goto done;
      }
    }
    dim3 grid((int)(n_pad / 64), (int)(p_pad / 64));
    if (mi->X) hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, mi->Xt, n_pad, mi->X, p_pad);
    if (mr && mr->X) hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, mr->Xt, n_pad, mr->X, p_pad);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&mi->absmax, mi->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && mr)
      e = hipMemcpyAsync(&mr->absmax, mr->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_err(ctx, EOFX_ERR_HIP, "hilbert stage failed: %s", hipGetErrorString(e));
  }
done:
  pool_give(ctx, work, work_bytes);
  pool_give(ctx, spec, spec_bytes);
  pool_give(ctx, coef, coef_bytes);
  if (a->rawT) {                 // consumed (or not usable by this call): back to the pool either way
    eofx_mat* am = const_cast<eofx_mat*>(a);
    pool_give(ctx, am->rawT, (size_t)n_pad * p_pad * sizeof(float));
    am->rawT = nullptr;
  }
  if (xt_transient && a->Xt) {   // (the stage ends with a stream synchronisation; same-stream reuse is ordered anyway)
    eofx_mat* am = const_cast<eofx_mat*>(a);
    pool_give(ctx, am->Xt, (size_t)n_pad * p_pad * sizeof(float));
    am->Xt = nullptr;
  }
  if (rc != EOFX_OK) {
    if (mi) eofx_mat_destroy(ctx, mi);
    if (mr) eofx_mat_destroy(ctx, mr);
    return rc;
  }
  *out_imag = mi;
  if (out_real) *out_real = mr;
  return EOFX_OK;
}

// Sum of squares of the imaginary part eofx_hilbert_f32 would produce for `a` (so total variance of the analytic signal =
// (sumsq(a) + this) / (n - 1), reference single/eof.py:93 on the complex field) WITHOUT writing it: the one-kernel route with
// zero-sized output descriptors and one float64 partial per wave, summed on the host in a fixed order.  Consumes the
// transposed raw layout of an in-place matrix exactly as eofx_hilbert_f32 does.  Series longer than the one-kernel
// route (or its one-feature-per-workgroup form) materialise the part, sum it and return it to the pool.
extern "C" int eofx_hilbert_sumsq_f64(eofx_ctx* ctx, const eofx_mat* a, int padding, double decay_factor, double* out) {
  if (!ctx || !a || !out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (padding && !(decay_factor > 0.0)) return set_err(ctx, EOFX_ERR_ARG, "decay_factor must be positive");
  ENTER_EXCLUSIVE(ctx);
  const int64_t n = a->n, p = a->p, n_pad = a->n_pad, p_pad = a->p_pad;
  int L = 0;
  const int64_t P = hilbert_length(n, &L);
  const bool fused = L <= 14 && n_pad <= P / 2;
  if (!fused) {
    eofx_mat* mi = nullptr;
    CHK(eofx_hilbert_f32(ctx, a, padding, decay_factor, &mi, nullptr));
    const int rc = eofx_mat_sumsq_f64(ctx, mi, out);
    eofx_mat_destroy(ctx, mi);
    return rc;
  }
  const cfloat* chat = nullptr;
  const float *hperm = nullptr, *u = nullptr;
  CHK(get_hilbert_setup(ctx, n, padding ? 1 : 0, decay_factor, P, true, &chat, &hperm, &u));
  const bool lean = a->raw && a->aff && !a->X;
  const bool from_rawT = lean && a->rawT && !a->Xt;
  const bool xt_transient = lean && !a->Xt && !from_rawT;
  if (!from_rawT) CHK(ensure_Xt(ctx, a));
  int rc = EOFX_OK;
  double* sq = nullptr;
  unsigned* bmax = nullptr;
  std::vector<double> hsq((size_t)HILBERT_SQ_PARTIALS);
  const size_t sq_bytes = sizeof(double) * HILBERT_SQ_PARTIALS + 256;
  if (pool_malloc(ctx, (void**)&sq, sq_bytes) != hipSuccess) {
    (void)hipGetLastError();
    rc = set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the partial sums");
  }
  if (rc == EOFX_OK) {
    bmax = reinterpret_cast<unsigned*>(sq + HILBERT_SQ_PARTIALS);
    hipError_t e = hipMemsetAsync(sq, 0, sq_bytes, ctx->stream);
    if (e == hipSuccess) switch (L) {
#define EOFX_HF(LL) \
  case LL: e = launch_hilbert_fused<LL, 0>(ctx, from_rawT ? a->rawT : a->Xt, n_pad, n, p, padding ? 1 : 0, hperm, u, nullptr, nullptr, bmax, nullptr, \
                                           from_rawT ? a->aff : nullptr, a->p_pad, a->masked ? a->aff + 2 * a->p_pad : nullptr, sq); break;
      EOFX_HF(10) EOFX_HF(11) EOFX_HF(12) EOFX_HF(13) EOFX_HF(14)
#undef EOFX_HF
      default: e = hipErrorInvalidValue;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hsq.data(), sq, sizeof(double) * HILBERT_SQ_PARTIALS, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = set_err(ctx, EOFX_ERR_HIP, "hilbert stage failed: %s", hipGetErrorString(e));
  }
  pool_give(ctx, sq, sq_bytes);
  if (a->rawT) {                 // consumed (or not usable by this call): back to the pool either way
    eofx_mat* am = const_cast<eofx_mat*>(a);
    pool_give(ctx, am->rawT, (size_t)n_pad * p_pad * sizeof(float));
    am->rawT = nullptr;
  }
  if (xt_transient && a->Xt) {
    eofx_mat* am = const_cast<eofx_mat*>(a);
    pool_give(ctx, am->Xt, (size_t)n_pad * p_pad * sizeof(float));
    am->Xt = nullptr;
  }
  if (rc != EOFX_OK) return rc;
  double t = 0.0;
  for (double v : hsq) t += v;
  *out = t;
  return EOFX_OK;
}

extern "C" int eofx_cpanel_combine_f32(eofx_ctx* ctx, const float* P1, const float* P2, int conj_left,
                                       int64_t rows_pad, int L, float* out) {
  if (!ctx || !P1 || !P2 || !out || L % 2) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int64_t total = rows_pad * (L / 2);
  hipLaunchKernelGGL(cpanel_combine_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0,
                     ctx->stream, P1, P2, conj_left ? 1.f : -1.f, rows_pad, L, out);
  KCHK();
  return EOFX_OK;
}

static int panel_colargminmax(eofx_ctx* ctx, const float* P, int64_t rows, int L, int64_t* amax, int64_t* amin, const float* rowscale);
extern "C" int eofx_panel_colargminmax_f32(eofx_ctx* ctx, const float* P, int64_t rows, int L, int64_t* amax,
                                           int64_t* amin) {
  return panel_colargminmax(ctx, P, rows, L, amax, amin, nullptr);
}
static int panel_colargminmax(eofx_ctx* ctx, const float* P, int64_t rows, int L, int64_t* amax, int64_t* amin, const float* rowscale) {
  if (!ctx || !P || !amax || !amin) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((rows + 15) / 16, 2048));
  CHK(arena_reserve(ctx, (size_t)nparts * L * 24 + 8192));
  ArenaScope scope(ctx);
  ARENA(float, pmx, (size_t)nparts * L);
  ARENA(float, pmn, (size_t)nparts * L);
  ARENA(int64_t, imx, (size_t)nparts * L);
  ARENA(int64_t, imn, (size_t)nparts * L);
  hipLaunchKernelGGL(colargminmax_part_kernel, dim3(nparts, (L + 63) / 64), dim3(256), 0, ctx->stream, P, rows, L,
                     pmx, imx, pmn, imn, rowscale);
  KCHK();
  hipLaunchKernelGGL(colargminmax_final_kernel, dim3((L + 63) / 64), dim3(1024), 0, ctx->stream, pmx, imx, pmn, imn,
                     nparts, L, amax, amin);
  KCHK();
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// complex randomized SVD (rows R9 of SURVEY.md §8a): Z = A + iB as two resident real matrices.
// Replaces scipy.sparse.linalg.svds(lobpcg) at xeofs/linalg/decomposer.py:149-160.  A complex panel of h columns is
// a real panel [Re | Im] of 2 h columns; a pass over the data is ONE launch of the streaming kernel in its two-matrix
// form (Z^H W = A^T [Wr|Wi] + B^T [Wi|-Wr],  Z Y = A [Yr|Yi] + B [-Yi|Yr]); orthonormalisation is a complex
// Cholesky-QR on the Hermitian Gram matrix assembled from one real float64 Gram of the panel (host, l <= 64).
// ------------------------------------------------------------------------------------
namespace {
typedef std::complex<double> zdouble;

// Hermitian l x l Gram P^H P from the real LP x LP Gram of [Pr | Pi] (h = LP / 2)
static void hermitian_from_real(const std::vector<double>& G, int LP, int l, std::vector<zdouble>& H) {
  const int h = LP / 2;
  H.assign((size_t)l * l, zdouble(0.0, 0.0));
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) {
      const double rr = G[(size_t)i * LP + j], ii = G[(size_t)(h + i) * LP + h + j];
      const double ri = G[(size_t)i * LP + h + j], ir = G[(size_t)(h + i) * LP + j];
      H[(size_t)i * l + j] = zdouble(rr + ii, ri - ir);
    }
  for (int i = 0; i < l; ++i)
    for (int j = i; j < l; ++j) {
      const zdouble v = 0.5 * (H[(size_t)i * l + j] + std::conj(H[(size_t)j * l + i]));
      H[(size_t)i * l + j] = v;
      H[(size_t)j * l + i] = std::conj(v);
    }
}
// T (l x l upper triangular) with (P T)^H (P T) = I for H = P^H P; dependent columns -> zero columns (same rule as the
// real driver's chol_rinv: pivot below tol * original diagonal)
static void host_zchol_rinv(const std::vector<zdouble>& Hin, int l, std::vector<zdouble>& T, double tol,
                            std::vector<zdouble>* Rout = nullptr, int* n_live = nullptr, const double* dref = nullptr, double tolref = 0.0) {
  std::vector<zdouble> A(Hin);
  std::vector<double> d0(l);
  std::vector<char> dead(l, 0);
  for (int j = 0; j < l; ++j) d0[j] = Hin[(size_t)j * l + j].real();
  for (int j = 0; j < l; ++j) {        // H = R^H R, R upper triangular, stored in the upper part of A
    const double d = A[(size_t)j * l + j].real();
    const bool dj = !(d > tol * d0[j]) || !(d0[j] > 0.0) || (dref && !(d > tolref * dref[j]));   // (dref: see eofx_rsvd_c64)
    dead[j] = dj;
    const double rjj = dj ? 1.0 : std::sqrt(d);
    const double piv = dj ? 0.0 : 1.0 / rjj;
    A[(size_t)j * l + j] = rjj;
    for (int c = j + 1; c < l; ++c) A[(size_t)j * l + c] *= piv;
    for (int r = j + 1; r < l; ++r) {
      const zdouble f = std::conj(A[(size_t)j * l + r]);
      if (f == zdouble(0.0, 0.0)) continue;
      for (int c = r; c < l; ++c) A[(size_t)r * l + c] -= f * A[(size_t)j * l + c];
    }
  }
  if (Rout) {                          // P = Q R (a dependent column keeps its coefficients on the earlier columns of Q)
    Rout->assign((size_t)l * l, zdouble(0.0, 0.0));
    for (int r = 0; r < l; ++r)
      for (int c = r; c < l; ++c) (*Rout)[(size_t)r * l + c] = (r == c && dead[r]) ? zdouble(0.0, 0.0) : A[(size_t)r * l + c];
  }
  if (n_live) {
    *n_live = 0;
    for (int j = 0; j < l; ++j) *n_live += dead[j] ? 0 : 1;
  }
  T.assign((size_t)l * l, zdouble(0.0, 0.0));
  for (int c = 0; c < l; ++c) {        // T = R^-1 column by column
    if (dead[c]) continue;
    T[(size_t)c * l + c] = 1.0 / A[(size_t)c * l + c];
    for (int r = c - 1; r >= 0; --r) {
      zdouble sum(0.0, 0.0);
      for (int t = r + 1; t <= c; ++t) sum += A[(size_t)r * l + t] * T[(size_t)t * l + c];
      T[(size_t)r * l + c] = -sum / A[(size_t)r * l + r];
    }
  }
}
// Hermitian eigen-decomposition through the real symmetric embedding [[Hr, -Hi], [Hi, Hr]] (every eigenvalue twice,
// eigenvectors (x; y) <-> x + i y) and the real tridiagonal QL solver; complex Gram-Schmidt inside clusters removes the
// duplicates.  -> w descending, V columns (row-major l x l).
static int host_heigh(const std::vector<zdouble>& H, int l, std::vector<double>& w, std::vector<zdouble>& V) {
  const int m = 2 * l;
  std::vector<double> E((size_t)m * m), ew(m), ev((size_t)m * m);
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < l; ++j) {
      const zdouble v = H[(size_t)i * l + j];
      E[(size_t)i * m + j] = v.real();
      E[(size_t)(l + i) * m + l + j] = v.real();
      E[(size_t)i * m + l + j] = -v.imag();
      E[(size_t)(l + i) * m + j] = v.imag();
    }
  const int rc = eofx_host_eigh_f64(E.data(), m, ew.data(), ev.data());   // descending eigenvalues, columns
  if (rc != EOFX_OK) return rc;
  w.assign(l, 0.0);
  V.assign((size_t)l * l, zdouble(0.0, 0.0));
  int got = 0;
  const double scale = std::max(std::fabs(ew[0]), std::fabs(ew[m - 1]));
  for (int c = 0; c < m && got < l; ++c) {
    std::vector<zdouble> v(l);
    for (int i = 0; i < l; ++i) v[i] = zdouble(ev[(size_t)i * m + c], ev[(size_t)(l + i) * m + c]);
    for (int pass = 0; pass < 2; ++pass)
      for (int g = 0; g < got; ++g) {
        if (std::fabs(w[g] - ew[c]) > 1e-6 * scale + 1e-300) continue;    // other clusters are orthogonal already
        zdouble dot(0.0, 0.0);
        for (int i = 0; i < l; ++i) dot += std::conj(V[(size_t)i * l + g]) * v[i];
        for (int i = 0; i < l; ++i) v[i] -= dot * V[(size_t)i * l + g];
      }
    double nrm = 0.0;
    for (int i = 0; i < l; ++i) nrm += std::norm(v[i]);
    nrm = std::sqrt(nrm);
    if (nrm < 0.5) continue;             // the partner (i v) of an accepted vector
    for (int i = 0; i < l; ++i) V[(size_t)i * l + got] = v[i] / nrm;
    w[got] = ew[c];
    ++got;
  }
  return got == l ? EOFX_OK : EOFX_ERR_LINALG;
}
// real LP x Lo matrix E with [Pr|Pi] E = [Re(P M) | Im(P M)] for complex M (l x m), h = LP/2, ho = Lo/2
static void embed_right(const std::vector<zdouble>& M, int l, int mcols, int LP, int Lo, std::vector<double>& E) {
  const int h = LP / 2, ho = Lo / 2;
  E.assign((size_t)LP * Lo, 0.0);
  for (int i = 0; i < l; ++i)
    for (int j = 0; j < mcols; ++j) {
      const zdouble v = M[(size_t)i * mcols + j];
      E[(size_t)i * Lo + j] = v.real();
      E[(size_t)(h + i) * Lo + j] = -v.imag();
      E[(size_t)i * Lo + ho + j] = v.imag();
      E[(size_t)(h + i) * Lo + ho + j] = v.real();
    }
}
}  // namespace

// The Hilbert stage as ONE linear map along the samples: for a series y [n] (padding "exp": linear fit, exponential pads,
// transform of the 3n-long series, middle third -- reference utils/hilbert_transform.py:47-92; no padding: the circular
// transform of the series itself), minus the mean over the samples,
//   Im = Hc y,   Hc = C (T + u1 (e_0 - a0)^T + u2 (e_{n-1} - a0 - (n-1) a1)^T + u3 a0^T + u4 a1^T),   C = I - 1 1^T / n
// with T[t][s] = kappa(N, t - s), a0^T y = c0 and a1^T y = c1 the coefficients of the linear fit, and u1..u4 the
// correction vectors of get_hilbert_setup.  Built in float64 on the host, held as a resident n x n matrix (both layouts)
// per (n, padding, decay): eofx_rsvd_hilbert_c64 applies it to the SAMPLE-side panels instead of materialising Im.
static int build_hilbert_operator_host(eofx_ctx* ctx, int64_t n, int padding, double decay, std::vector<float>& hc);
static int get_hilbert_operator(eofx_ctx* ctx, int64_t n, int padding, double decay, const eofx_mat** out) {
  for (auto& h : ctx->hops)
    if (h.n == n && h.padding == padding && (!padding || h.decay == decay)) {
      *out = h.m;
      return EOFX_OK;
    }
  std::vector<float> hc;
  CHK(build_hilbert_operator_host(ctx, n, padding, decay, hc));
  eofx_mat* m = nullptr;
  CHK(eofx_mat_from_dense_f32(ctx, hc.data(), n, n, n, &m));
  eofx_ctx::HilbertOp h;
  h.n = n; h.padding = padding; h.decay = decay; h.m = m;
  while (ctx->hops.size() >= 2) {          // a small cache (an operator is up to 2 GB): the oldest one goes
    (void)eofx_mat_destroy(ctx, ctx->hops.front().m);
    ctx->hops.erase(ctx->hops.begin());
  }
  ctx->hops.push_back(h);
  *out = m;
  return EOFX_OK;
}
// Hc [n x n] row-major float32 on the host: Im = Hc A for the Hilbert stage of eofx_hilbert_f32 along the samples (built in float64)
static int build_hilbert_operator_host(eofx_ctx* ctx, int64_t n, int padding, double decay, std::vector<float>& hc) {
  const int64_t N = padding ? 3 * n : n;
  std::vector<double> kap((size_t)(2 * n + 1)), pre((size_t)(2 * n + 2), 0.0);   // kappa(N, d), d = -n .. n, and its prefix sums
  for (int64_t d = -n; d <= n; ++d) kap[(size_t)(d + n)] = hilbert_kappa(N, d);
  for (size_t i = 0; i < kap.size(); ++i) pre[i + 1] = pre[i] + kap[i];
  std::vector<double> u, rowv((size_t)4 * n, 0.0), ubar(4, 0.0);
  if (padding) {
    hilbert_pad_vectors(n, N, decay, u);
    const double tbar = 0.5 * (double)(n - 1), stt = (double)n * ((double)n * (double)n - 1.0) / 12.0;
    for (int64_t sidx = 0; sidx < n; ++sidx) {
      const double a1 = n > 1 ? ((double)sidx - tbar) / stt : 0.0, a0 = 1.0 / (double)n - tbar * a1;
      rowv[(size_t)sidx] = (sidx == 0 ? 1.0 : 0.0) - a0;                                      // amp_pre = y[0] - c0
      rowv[(size_t)(n + sidx)] = (sidx == n - 1 ? 1.0 : 0.0) - a0 - (double)(n - 1) * a1;    // amp_pos = y[n-1] - fit(n-1)
      rowv[(size_t)(2 * n + sidx)] = a0;
      rowv[(size_t)(3 * n + sidx)] = a1;
    }
    for (int k = 0; k < 4; ++k) {
      double m = 0.0;
      for (int64_t t = 0; t < n; ++t) m += u[(size_t)k * n + t];
      ubar[k] = m / (double)n;
    }
  }
  try {
    hc.resize((size_t)n * n);
  } catch (const std::bad_alloc&) {
    return set_err(ctx, EOFX_ERR_NOMEM, "cannot allocate the %lld x %lld Hilbert operator on the host", (long long)n, (long long)n);
  }
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t t = lo; t < hi; ++t) {
      float* row = hc.data() + (size_t)t * n;
      double ut[4] = {0, 0, 0, 0};
      if (padding)
        for (int k = 0; k < 4; ++k) ut[k] = u[(size_t)k * n + t] - ubar[k];
      for (int64_t sidx = 0; sidx < n; ++sidx) {
        // column mean of T: (1/n) sum_t kappa(t - s) = (pre[n - s + n] - pre[-s + n]) / n
        const double cm = (pre[(size_t)(2 * n - sidx)] - pre[(size_t)(n - sidx)]) / (double)n;
        double v = kap[(size_t)(t - sidx + n)] - cm;
        if (padding)
          v += ut[0] * rowv[(size_t)sidx] + ut[1] * rowv[(size_t)(n + sidx)] + ut[2] * rowv[(size_t)(2 * n + sidx)] +
               ut[3] * rowv[(size_t)(3 * n + sidx)];
        row[sidx] = (float)v;
      }
    }
  };
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(16, n / 256));
  std::vector<std::thread> th;
  const int64_t step = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) th.emplace_back(work, t * step, std::min<int64_t>(n, (t + 1) * step));
  for (auto& t : th) t.join();
  return EOFX_OK;
}
extern "C" int eofx_hilbert_operator_f32(eofx_ctx* ctx, int64_t n, int padding, double decay_factor, float* out) {
  if (!ctx || !out || n <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (padding && !(decay_factor > 0.0)) return set_err(ctx, EOFX_ERR_ARG, "decay_factor must be positive");
  if (n > EOFX_HILBERT_OP_MAX_N) return set_err(ctx, EOFX_ERR_ARG, "the Hilbert operator is limited to %d samples", EOFX_HILBERT_OP_MAX_N);
  if (is_device_ptr(out)) return set_err(ctx, EOFX_ERR_ARG, "out must be a host pointer");
  std::vector<float> hc;
  CHK(build_hilbert_operator_host(ctx, n, padding ? 1 : 0, decay_factor, hc));
  std::memcpy(out, hc.data(), sizeof(float) * hc.size());
  return EOFX_OK;
}

struct CplxOps {
  eofx_ctx* ctx;
  const eofx_mat *A, *B;
  int LP;
  float* rot;    // companion panel [max(n_pad, p_pad) x LP]
  float* tmp;    // second product of the two-launch path (other precisions than f16x3)
  float absmax;
  int rot_panel(const float* P, float sgn, int64_t rows) {
    const int64_t total = rows * (LP / 2);
    hipLaunchKernelGGL(cpanel_rot_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, ctx->stream, P, sgn,
                       rows, LP, rot);
    return hipGetLastError() == hipSuccess ? EOFX_OK : EOFX_ERR_HIP;
  }
  int combine(const float* P1, const float* P2, float sgn, int64_t rows, float* out) {
    const int64_t total = rows * (LP / 2);
    hipLaunchKernelGGL(cpanel_combine_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                       P1, P2, sgn, rows, LP, out);
    return hipGetLastError() == hipSuccess ? EOFX_OK : EOFX_ERR_HIP;
  }
  // Lean layout (any part that is not held in both written layouts: Re as the raw field in place, Im in the
  // sample-contiguous layout only -- what eofx_hilbert_f32 leaves for an in-place input -- or both parts of a complex
  // input in place): every pass streams each part once in a layout it has.  A part whose contraction axis is its
  // contiguous one goes through axb_f16 (rows streamed along their lines), the other way through atb_f16, and the two
  // real products meet in cpanel_combine.  `ident` is the identity map (0, 0, 1) for the rows of a written Im^T.
  bool lean = false;
  const float* ident = nullptr;
  int64_t ident_ld = 0;
  // Operator mode (eofx_rsvd_hilbert_c64): B is absent and Z = (I + i Hc) A with the n x n Hilbert operator Hc resident
  // (get_hilbert_operator).  Z^H W = A^T (W - i Hc^T W) and Z Y = (I + i Hc)(A Y): ONE real pass over the field per
  // product plus an n x n x LP product on the sample side -- the imaginary part is never written or read.
  const eofx_mat* Hop = nullptr;
  static bool axb_fits(int64_t ld, int64_t cols, int L) {   // the 32-bit offsets of axb_f16 (launch_axb checks them too)
    const int64_t K = round_up(cols, AXB_KG);
    return K * (int64_t)L < ((int64_t)1 << 30) && 64 * ld + K < ((int64_t)1 << 30);   // BYTE offsets in 32 bits
  }
  int t_part(const eofx_mat* M, const float* Wn, float* out, int prec) {      // M^T W  [p_pad x LP]
    if (!M->X && M->raw && M->aff && prec == EOFX_PREC_F16X3) {
      AffView av;
      av.aff = M->aff;
      av.ld = M->p_pad;
      av.rows = (int)M->n;
      av.cols = M->p;
      av.masked = M->masked;
      return launch_atb(ctx, M->raw, M->raw_ld, round_up(M->n, ATB_KG), M->p_pad, Wn, LP, LP, out, prec, M->absmax, nullptr, &av);
    }
    if (!M->X && M->Xt && prec == EOFX_PREC_F16X3 && ident && axb_fits(M->n_pad, M->n, LP))   // the rows of M^T times W
      return launch_axb(ctx, M->Xt, M->n_pad, M->p, M->n, M->p_pad, ident, ident_ld, M->absmax, Wn, LP, out);
    CHK(ensure_X(ctx, M));
    return launch_atb(ctx, M->X, M->p_pad, round_up(M->n, ATB_KG), M->p_pad, Wn, LP, LP, out, prec, M->absmax);
  }
  int n_part(const eofx_mat* M, const float* Yp, float* out, int prec) {      // M Y  [n_pad x LP]
    if (!M->Xt && M->raw && M->aff && prec == EOFX_PREC_F16X3 && axb_fits(M->raw_ld, M->p, LP))
      return launch_axb(ctx, M->raw, M->raw_ld, M->n, M->p, M->n_pad, M->aff, M->p_pad, M->absmax, Yp, LP, out, M->masked);
    CHK(ensure_Xt(ctx, M));
    return launch_atb(ctx, M->Xt, M->n_pad, round_up(M->p, ATB_KG), M->n_pad, Yp, LP, LP, out, prec, M->absmax);
  }
  // feature-side panel = Z^H W
  int zh_mul(const float* Wn, float* Yp, int prec) {
    const int64_t K = round_up(A->n, ATB_KG), M = A->p_pad;
    if (Hop) {
      CHK(launch_atb(ctx, Hop->X, Hop->p_pad, K, Hop->p_pad, Wn, LP, LP, tmp, prec, Hop->absmax));   // Hc^T [Wr | Wi]
      amax_forget(ctx, rot);
      CHK(combine(Wn, tmp, 1.f, A->n_pad, rot));                                                      // W - i Hc^T W
      return t_part(A, rot, Yp, prec);
    }
    if (lean) {
      CHK(t_part(A, Wn, Yp, prec));
      CHK(t_part(B, Wn, tmp, prec));
      amax_forget(ctx, Yp);
      return combine(Yp, tmp, 1.f, A->p_pad, Yp);
    }
    if (prec == EOFX_PREC_F16X3) {
      CHK(rot_panel(Wn, 1.f, A->n_pad));
      return launch_atb(ctx, A->X, M, K, M, Wn, LP, LP, Yp, prec, absmax, nullptr, nullptr, B->X, rot);
    }
    CHK(launch_atb(ctx, A->X, M, K, M, Wn, LP, LP, Yp, prec, absmax));
    CHK(launch_atb(ctx, B->X, M, K, M, Wn, LP, LP, tmp, prec, absmax));
    return combine(Yp, tmp, 1.f, A->p_pad, Yp);
  }
  // sample-side panel = Z Y
  int z_mul(const float* Yp, float* Wn, int prec) {
    const int64_t K = round_up(A->p, ATB_KG), M = A->n_pad;
    if (Hop) {
      CHK(n_part(A, Yp, tmp, prec));                                                                  // T = A [Yr | Yi]
      CHK(launch_atb(ctx, Hop->Xt, Hop->n_pad, round_up(A->n, ATB_KG), Hop->n_pad, tmp, LP, LP, rot, prec, Hop->absmax));   // Hc T
      amax_forget(ctx, Wn);
      return combine(tmp, rot, -1.f, A->n_pad, Wn);                                                   // T + i Hc T
    }
    if (lean) {
      CHK(n_part(A, Yp, Wn, prec));
      CHK(n_part(B, Yp, tmp, prec));
      amax_forget(ctx, Wn);
      return combine(Wn, tmp, -1.f, A->n_pad, Wn);
    }
    if (prec == EOFX_PREC_F16X3) {
      CHK(rot_panel(Yp, -1.f, A->p_pad));
      return launch_atb(ctx, A->Xt, M, K, M, Yp, LP, LP, Wn, prec, absmax, nullptr, nullptr, B->Xt, rot);
    }
    CHK(launch_atb(ctx, A->Xt, M, K, M, Yp, LP, LP, Wn, prec, absmax));
    CHK(launch_atb(ctx, B->Xt, M, K, M, Yp, LP, LP, tmp, prec, absmax));
    return combine(Wn, tmp, -1.f, A->n_pad, Wn);
  }
};

// lean layout (see CplxOps): some part is not held in both written layouts -- every part is streamed in a layout it has
// instead of writing the missing ones (64-column panels, split-fp16 passes)
static bool cplx_lean(const eofx_mat* A, const eofx_mat* B, int LP, int prec_power, int prec_final) {
  const bool all_written = A->X && A->Xt && B->X && B->Xt;
  // (a masked real part -- zero columns in place, the MASK forms of the streaming kernels -- is streamed where it lies as well;
  // the imaginary part of such a field is a written matrix whose columns there are zeros)
  return !all_written && !B->masked && LP == 64 && prec_power == EOFX_PREC_F16X3 && prec_final == EOFX_PREC_F16X3;
}
// the identity map for the rows of a written sample-contiguous part (arena memory of the caller's scope)
static int cplx_lean_setup(eofx_ctx* ctx, CplxOps& ops) {
  const int64_t n = ops.A->n, il = round_up(n, AXB_KG);
  float* ident = arena_alloc<float>(ctx, 3 * (size_t)il);
  if (!ident) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (identity map)");
  std::vector<float> h(3 * (size_t)il, 0.f);
  for (int64_t i = 0; i < n; ++i) h[2 * (size_t)il + i] = 1.f;     // (shift hi, shift lo, scale) = (0, 0, 1); scale 0 beyond n
  CHK(copy_in(ctx, ident, h.data(), sizeof(float) * h.size()));
  HIPCHK(hipStreamSynchronize(ctx->stream));                         // h leaves scope
  ops.lean = true;
  ops.ident = ident;
  ops.ident_ld = il;
  return EOFX_OK;
}

// One pass of the complex operator on a [Re | Im] panel (the step the feature-sharded driver all-reduces around):
// conj_left = 1: out [p_pad x L] = Z^H W for W [n_pad x L]; conj_left = 0: out [n_pad x L] = Z Y for Y [p_pad x L].
// In the default precision this is ONE launch of the streaming kernel over both parts (as inside eofx_rsvd_c64).
extern "C" int eofx_cmat_mul_f32(eofx_ctx* ctx, const eofx_mat* A, const eofx_mat* B, int conj_left, const float* Pin, int L,
                                 int final_pass, float* Pout) {
  if (!ctx || !A || !B || !Pin || !Pout) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (A->n != B->n || A->p != B->p) return set_err(ctx, EOFX_ERR_SHAPE, "real and imaginary parts must have the same shape");
  if (L != 64 && L != 128) return set_err(ctx, EOFX_ERR_ARG, "complex panels are 64 or 128 real columns wide");
  ENTER(ctx);
  const int prec = final_pass ? ctx->prec_final : ctx->prec_power;
  const bool lean = cplx_lean(A, B, L, prec, prec);
  if (!lean) {
    CHK(ensure_X(ctx, A));
    CHK(ensure_X(ctx, B));
  }
  const int64_t big = std::max(A->n_pad, A->p_pad);
  size_t need = (size_t)2 * big * L * 4 + (8 << 20);
  need += (size_t)2 * big * L * 4;   // the two partial results of a two-matrix launch that needs no split
  need += 2 * atb_scratch_bytes(A->p_pad, round_up(A->n, ATB_KG), L) + 2 * atb_scratch_bytes(A->n_pad, round_up(A->p, ATB_KG), L);
  CHK(arena_reserve(ctx, need));
  ArenaScope scope(ctx);
  ARENA(float, rot, (size_t)big * L);
  ARENA(float, tmp, (size_t)big * L);
  CplxOps ops{ctx, A, B, L, rot, tmp, std::max(A->absmax, B->absmax)};
  if (lean) CHK(cplx_lean_setup(ctx, ops));
  CHK(conj_left ? ops.zh_mul(Pin, Pout, prec) : ops.z_mul(Pin, Pout, prec));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}

// U [n x k] and V [p x k] are complex64, row-major, interleaved (re, im); s [k] float32; all host|device.
// omega: [min(n, p) x (k + n_oversamples)] REAL Gaussian start (host), as the reference's random_state would draw.
//
// Round 5: BLOCK KRYLOV.  The reference's complex branch is scipy's svds(solver="lobpcg") (linalg/decomposer.py:149-160): a
// block Krylov-class eigen-solver on Z^H Z.  Plain subspace iteration with scikit-learn's count (rounds 1-4) reads the field
// the same number of times but leaves modes inside a flat noise bulk 1e-3 .. 4e-2 short.  The passes over the field are now
// the steps of a block Lanczos recurrence on the small side, M = A_op^H A_op:
//     P_i = A_op Z_i (tall, KEPT),  W_i = A_op^H P_i = M Z_i,  Z_{i+1} = orth(W_i - K (K^H W_i)),  K = [Z_0 .. Z_i]
// (full re-orthogonalisation, twice: a few small-side kernels on min(n, p) x 64 panels), and after the same q products a
// Rayleigh-Ritz step over the WHOLE Krylov space K = [Z_0 .. Z_q] picks the sketch-width subspace: H = K^H M K is assembled
// from the products already made (K^H W_i; the last diagonal block is P_q^H P_q), its leading eigenvectors y come from the
// host (hosteig::zheigh_top, order (q + 1) l = 240 at config 5), and the tall basis A_op K y is a linear combination of the kept
// panels P_i -- no extra pass.  The final stage (CholeskyQR2, one projection pass, l x l Hermitian problem) is unchanged and
// still fixes the values, so the Rayleigh-Ritz matrix only has to identify the subspace.  Same 2 q + 2 passes as before;
// on the bench's config-5 sample the worst mode goes from 8.7e-3 to 6e-7 (tools/probes/krylov_rr_prototype.py).
//   n_iter >= 0: that many products;  -1: scikit-learn's count (7 if k < 0.1 min(n, p) else 4);
//   -2 ("converge"): restarted cycles of that count, each starting from the Ritz block of the previous one, until the leading k
//        values move by <= 1e-6 (relative, squared values) or 20 products have been made (lobpcg's own limit under svds).
// Sketches whose Krylov space would exceed order 384 (k + n_oversamples > 48 at q = 7) are THICK-RESTARTED (`compress`: the Ritz
// block + the newest block) so that the host's Rayleigh-Ritz problem stays below that order; sketches so wide that fewer than 3
// blocks fit (l > 128: not reachable, l <= 64) would keep the subspace iteration of rounds 1-4 (EOFX_C64_KRYLOV=0 forces it, for
// comparisons).  The panel-level driver (xeofs_amd/complex_svd.py) solves its Rayleigh-Ritz problem with numpy and keeps up to
// order 512 WITHOUT a restart: for 48 < l <= 64 at q = 7 the two drivers run different (both convergent) recurrences -- by design.
// B != nullptr: Z = A + i B (both resident).  B == nullptr: Z = (I + i Hc) A with the resident Hilbert operator Hop.
// p_total > 0 (eofx_rsvd_sharded_c64 / eofx_rsvd_hilbert_sharded_c64): A (and B) are this rank's slice of the feature axis of a
// field with p_total (valid) features over all ranks of the context's communicator, n < p_total.  The recurrence lives on the
// sample side, which is replicated: the only collectives are one all-reduce(sum) of the n x LP sample-side panel per product
// Z Y, one of the LP x LP float64 Gram matrix per feature-side panel that is factorised, and two small ones for the sign rule.
// The Hilbert operator acts on the replicated sample-side panel, so the operator route shards without further exchange.
static int rsvd_c64_impl(eofx_ctx* ctx, const eofx_mat* A, const eofx_mat* B, const eofx_mat* Hop, int k, int n_oversamples,
                         int n_iter, const float* omega, int flip_signs, float* U, float* s, float* V, int64_t p_total = 0) {
  if (!ctx || !A || (!B && !Hop) || !omega || !s || k <= 0 || n_oversamples < 0)
    return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (B && (A->n != B->n || A->p != B->p)) return set_err(ctx, EOFX_ERR_SHAPE, "real and imaginary parts must have the same shape");
  ENTER(ctx);
  const bool shd = p_total > 0;
  if (shd && !ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached (eofx_ctx_comm_init_rccl / eofx_ctx_comm_set_callback)");
  if (shd && !(A->n < p_total)) return set_err(ctx, EOFX_ERR_ARG, "the sharded complex decomposition needs the sketch on the sample side (n < p_total)");
  const int64_t n = A->n, p = A->p, r = std::min(n, shd ? p_total : (A->masked ? A->p_valid : p));
  if (k > r) return set_err(ctx, EOFX_ERR_RANK, "n_modes must be less than or equal to the rank of the dataset (rank = %lld).", (long long)r);
  const int l = (int)std::min<int64_t>(k + n_oversamples, r);
  if (l > 64) return set_err(ctx, EOFX_ERR_ARG, "complex sketch width %d > 64 is not supported (n_modes + n_oversamples <= 64)", l);
  if (!shd && A->masked && !(n < A->p_valid)) return set_err(ctx, EOFX_ERR_ARG, "masked in-place matrix with fewer valid features than samples");
  const bool adaptive = n_iter == -2;
  const int auto_count = k < 0.1 * (double)r ? 7 : 4;
  if (n_iter == -1) n_iter = auto_count;
  const int it_min = 2, it_cap = 20;
  if (adaptive) n_iter = it_cap;
  if (n_iter < 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  const int h = l <= 32 ? 32 : 64, LP = 2 * h;
  const int ko = (int)round_up(k, 16), Lo = 2 * ko;       // output panels [Re(ko) | Im(ko)]
  const bool lean = B && cplx_lean(A, B, LP, ctx->prec_power, ctx->prec_final);
  // (sharded: a rank-local failure of the steps before the first collective -- building a layout, growing the arena -- must not
  // leave the other ranks in an all-reduce: it becomes this rank's vote below and every rank returns together)
  int rc_local = EOFX_OK;
  if (B && !lean) {
    rc_local = ensure_X(ctx, A);   // (ensure_X builds the sample-contiguous layout first where that is missing too)
    if (rc_local == EOFX_OK) rc_local = ensure_X(ctx, B);
    if (!shd) CHK(rc_local);
  }
  const bool transposed = shd || n < p;     // A_op = Z^H: tall side = features
  const int64_t small = transposed ? n : p;
  const int64_t small_pad = transposed ? A->n_pad : A->p_pad, tall_pad = transposed ? A->p_pad : A->n_pad;
  const int64_t big = std::max(A->n_pad, A->p_pad);
  // block Krylov: products per cycle, blocks, order of the Rayleigh-Ritz problem
  // blocks kept before a thick restart: the Rayleigh-Ritz problem stays below order 384 (the host solves it: ~10 ms at 240)
  constexpr int KRYLOV_MAX_ORDER = 384;
  const char* kenv = std::getenv("EOFX_C64_KRYLOV");
  const int nb_fit = KRYLOV_MAX_ORDER / l;
  const bool krylov = n_iter >= 1 && nb_fit >= 3 && !(kenv && atoi(kenv) == 0);
  const int nbmax = krylov ? std::min(n_iter + 1, nb_fit) : 0;
  const int64_t ldk = (int64_t)nbmax * LP;
  size_t need = (size_t)(2 * small_pad + 2 * tall_pad + 2 * big) * LP * 4 + (size_t)(small_pad + tall_pad) * Lo * 4;
  need += 2 * atb_scratch_bytes(A->p_pad, round_up(n, ATB_KG), LP) + 2 * atb_scratch_bytes(A->n_pad, round_up(p, ATB_KG), LP);
  need += (size_t)(4 * gram_parts(big, LP) + 8) * LP * LP * 8 + (size_t)big * (Lo + LP) * 4 + (8 << 20);
  if (krylov)
    need += (size_t)nbmax * (2 * small_pad + tall_pad) * LP * 4 + (size_t)small_pad * LP * 4 + (size_t)(3 + nbmax) * nbmax * LP * LP * 8 +
            ((size_t)48 << 20);
  if (rc_local == EOFX_OK) rc_local = arena_reserve(ctx, need);
  if (shd) {
    const std::string err_local = rc_local != EOFX_OK ? ctx->err : std::string();
    int verdict = 0;
    CHK(comm_vote(ctx, rc_local != EOFX_OK ? 2 : 0, &verdict));
    if (verdict != 0)
      return rc_local != EOFX_OK ? (ctx->err = err_local, rc_local) : set_err(ctx, EOFX_ERR_HIP, "the sharded complex decomposition failed on another rank");
  } else {
    CHK(rc_local);
  }
  ArenaScope scope(ctx);
  ARENA(float, Zs, (size_t)small_pad * LP);
  ARENA(float, Ws, (size_t)small_pad * LP);
  ARENA(float, Yt, (size_t)tall_pad * LP);
  ARENA(float, Qt, (size_t)tall_pad * LP);
  ARENA(float, rot, (size_t)big * LP);
  ARENA(float, tmp, (size_t)big * LP);
  ARENA(float, Tv, (size_t)tall_pad * Lo);
  ARENA(float, Sv, (size_t)small_pad * Lo);
  ARENA(double, G, (size_t)LP * LP);
  ARENA(double, Ed, (size_t)LP * LP);
  float *Kw = nullptr, *Ww = nullptr, *Pt = nullptr, *Vs = nullptr;
  double *Cd = nullptr, *Ecd = nullptr, *Eall = nullptr, *Cf = nullptr;
  if (krylov) {
    Kw = arena_alloc<float>(ctx, (size_t)small_pad * ldk);               // the Krylov blocks Z_0 .. Z_q side by side
    Ww = arena_alloc<float>(ctx, (size_t)small_pad * ldk);               // their products W_i = A_op^H (A_op Z_i)
    Pt = arena_alloc<float>(ctx, (size_t)nbmax * tall_pad * LP);         // the tall panels P_i = A_op Z_i (or their Q factors)
    Vs = arena_alloc<float>(ctx, (size_t)small_pad * LP);
    Cd = arena_alloc<double>(ctx, (size_t)nbmax * LP * LP);
    Ecd = arena_alloc<double>(ctx, (size_t)nbmax * LP * LP);
    Eall = arena_alloc<double>(ctx, (size_t)nbmax * LP * LP);
    Cf = arena_alloc<double>(ctx, (size_t)nbmax * nbmax * LP * LP);
    if (!Kw || !Ww || !Pt || !Vs || !Cd || !Ecd || !Eall || !Cf) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (block Krylov panels)");
  }
  CplxOps ops{ctx, A, B, LP, rot, tmp, B ? std::max(A->absmax, B->absmax) : A->absmax};
  ops.Hop = B ? nullptr : Hop;
  if (lean) CHK(cplx_lean_setup(ctx, ops));
  const int pp = ctx->prec_power, pf = ctx->prec_final;
  auto fwd = [&](const float* in, float* out, int prec) { return transposed ? ops.zh_mul(in, out, prec) : ops.z_mul(in, out, prec); };
  auto bwd = [&](const float* in, float* out, int prec) -> int {
    if (!transposed) return ops.zh_mul(in, out, prec);
    CHK(ops.z_mul(in, out, prec));
    if (shd) {                       // the partial sum over this rank's features -> the sum over all of them
      amax_forget(ctx, out);
      CHK(comm_allreduce(ctx, out, small_pad * LP, 0, 0));
    }
    return EOFX_OK;
  };
  std::vector<double> hG((size_t)LP * LP), hE;
  std::vector<zdouble> H, T;
  auto gram_h = [&](const float* P, int64_t rows_pad, bool tall_side = false) -> int {
    CHK(launch_gram(ctx, P, rows_pad, LP, G));
    if (shd && tall_side) CHK(comm_allreduce(ctx, G, (int64_t)LP * LP, 1, 0));     // rows sharded over the ranks
    HIPCHK(hipMemcpyAsync(hG.data(), G, sizeof(double) * LP * LP, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hermitian_from_real(hG, LP, l, H);
    for (const zdouble& v : H)
      if (!std::isfinite(v.real()) || !std::isfinite(v.imag()))
        return set_err(ctx, EOFX_ERR_LINALG, "SVD failed. This may be due to isolated NaN values in the data.");
    return EOFX_OK;
  };
  auto right_mul = [&](const float* P, int64_t rows_pad, const std::vector<zdouble>& M, int mcols, int Lout, float* out) -> int {
    embed_right(M, l, mcols, LP, Lout, hE);
    HIPCHK(hipMemcpyAsync(Ed, hE.data(), sizeof(double) * LP * Lout, hipMemcpyHostToDevice, ctx->stream));
    CHK(launch_matmul(ctx, P, rows_pad, LP, Ed, Lout, out));
    HIPCHK(hipStreamSynchronize(ctx->stream));    // hE / Ed are reused
    return EOFX_OK;
  };
  auto orth = [&](const float* P, int64_t rows_pad, float* out, bool tall_side = false) -> int {
    CHK(gram_h(P, rows_pad, tall_side));
    host_zchol_rinv(H, l, T, 1e-13);
    return right_mul(P, rows_pad, T, l, LP, out);
  };
  // start panel [Omega | 0]: a real Gaussian (or the identity for a full-width sketch, which the caller passes as omega)
  {
    std::vector<float> host((size_t)small * LP, 0.f);
    for (int64_t i = 0; i < small; ++i)
      for (int j = 0; j < l; ++j) host[(size_t)i * LP + j] = omega[(size_t)i * (k + n_oversamples) + j];
    CHK(import_panel(ctx, host.data(), small, LP, Zs, small_pad, LP));
  }
  // (adaptive subspace iteration: the tall panel is orthonormalised in every iteration -- only then is W^H W the Rayleigh
  // quotient whose eigenvalues are compared)
  const bool orth_always = (adaptive && !krylov) || orth_tall_rule(shd ? round_up(p_total, ATB_BM) : tall_pad, LP, pp);
  const bool trace = std::getenv("EOFX_C64_TRACE") != nullptr;
  ctx->last_iters = 0;

  // ---- block Lanczos state: blocks Z_0 .. Z_{nb-1} side by side in Kw (orthonormal, dead columns zero); block b < nW has
  //      been multiplied: slot b of Pt holds A_op Z_b (or its Q factor, then Rf[b] is the triangular factor) and block b of Ww
  //      holds W_b with M Z_b = W_b Rf[b]
  auto cplx_block = [&](const double* g, int64_t ldc, std::vector<zdouble>& out) {   // complex l x l block of a real LP x LP cross-Gram block
    out.assign((size_t)l * l, zdouble(0.0, 0.0));
    for (int i = 0; i < l; ++i)
      for (int j = 0; j < l; ++j)
        out[(size_t)i * l + j] = zdouble(g[(size_t)i * ldc + j] + g[(size_t)(h + i) * ldc + h + j], g[(size_t)i * ldc + h + j] - g[(size_t)(h + i) * ldc + j]);
  };
  auto zmatmul = [&](const std::vector<zdouble>& X, const std::vector<zdouble>& Y) {    // l x l
    std::vector<zdouble> Z((size_t)l * l, zdouble(0.0, 0.0));
    for (int i = 0; i < l; ++i)
      for (int t = 0; t < l; ++t) {
        const zdouble x = X[(size_t)i * l + t];
        if (x == zdouble(0.0, 0.0)) continue;
        for (int j = 0; j < l; ++j) Z[(size_t)i * l + j] += x * Y[(size_t)t * l + j];
      }
    return Z;
  };
  auto copy_block = [&](float* wide, const float* src, int blk) -> int {
    HIPCHK(hipMemcpy2DAsync(wide + (size_t)blk * LP, sizeof(float) * ldk, src, sizeof(float) * LP, sizeof(float) * LP, (size_t)small_pad,
                            hipMemcpyDeviceToDevice, ctx->stream));
    return EOFX_OK;
  };
  std::vector<double> hC;
  // out = Win - K (K^H Win) over the first nbk blocks; the real cross-Gram K^T Win stays in Cd (and in hC when asked for)
  auto project = [&](const float* Win, int nbk, float* out, bool to_host) -> int {
    CHK(launch_xgram(ctx, Kw, ldk, nbk * LP, Win, LP, LP, small_pad, Cd));
    if (to_host) {
      hC.resize((size_t)nbk * LP * LP);
      HIPCHK(hipMemcpyAsync(hC.data(), Cd, sizeof(double) * hC.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    hipLaunchKernelGGL(cproj_embed_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)nbk * h * h + 255) / 256, 1024)), dim3(256), 0, ctx->stream,
                       (const double*)Cd, nbk, LP, Ecd);
    KCHK();
    return launch_matmul_gen(ctx, Kw, ldk, 64, 1 << 20, small_pad, nbk * LP, Ecd, LP, Win, out);
  };
  int nb = 0, nW = 0;
  bool exhausted = false, orth_rest = orth_always;
  std::vector<std::vector<zdouble>> Rf(nbmax);      // (empty = identity: the tall panel was not orthonormalised)
  std::vector<zdouble> Hqq, blk;
  std::vector<double> dref(l), ones(l, 1.0);
  // multiply the newest block (b = nb - 1 = nW): slot b, W_b; then the next block = what is left of W_b after two rounds of
  // (project on all blocks, Cholesky-QR).  A column of the new block dies when what is left of it after the columns before it
  // falls below 1e-13 of its own squared norm (as everywhere), or -- first round -- below 1e-10 of its squared norm BEFORE the
  // projection (a residual below 1e-5 of the product is the rounding noise of a converged direction), or -- second round --
  // when the re-projected unit column kept less than half its length (it lay inside the blocks already there).
  auto lanczos_step = [&]() -> int {
    const int b = nW;
    float* slot = Pt + (size_t)b * tall_pad * LP;
    Rf[b].clear();
    if (ctx->last_iters == 0 || orth_rest) {
      CHK(fwd(Zs, Yt, pp));
      CHK(gram_h(Yt, tall_pad, true));
      host_zchol_rinv(H, l, T, 1e-13, &Rf[b]);
      CHK(right_mul(Yt, tall_pad, T, l, LP, slot));
    } else {
      CHK(fwd(Zs, slot, pp));
    }
    CHK(bwd(slot, Ws, pp));
    const bool first = ctx->last_iters == 0;
    ++ctx->last_iters;
    CHK(copy_block(Ww, Ws, b));
    nW = b + 1;
    CHK(project(Ws, nb, Vs, true));
    CHK(gram_h(Vs, small_pad));                      // (synchronises: hC is on the host)
    for (int j = 0; j < l; ++j) dref[j] = H[(size_t)j * l + j].real();      // |W_j|^2 = |V_j|^2 + |K^H W_j|^2
    for (int c = 0; c < nb; ++c) {
      cplx_block(&hC[(size_t)c * LP * LP], LP, blk);
      for (int i = 0; i < l; ++i)
        for (int j = 0; j < l; ++j) dref[j] += std::norm(blk[(size_t)i * l + j]);
    }
    if (first && !orth_always && n_iter > 1) {       // peaked spectrum?  H_00 = Z_0^H M Z_0 is on the host
      cplx_block(hC.data(), LP, blk);
      std::vector<zdouble> H00 = Rf[0].empty() ? blk : zmatmul(blk, Rf[0]), V0;
      for (int i = 0; i < l; ++i)
        for (int j = i; j < l; ++j) {
          const zdouble v = 0.5 * (H00[(size_t)i * l + j] + std::conj(H00[(size_t)j * l + i]));
          H00[(size_t)i * l + j] = v;
          H00[(size_t)j * l + i] = std::conj(v);
        }
      std::vector<double> w0;
      orth_rest = host_heigh(H00, l, w0, V0) != EOFX_OK || !(w0[l - 1] > 0.0) || std::sqrt(w0[0] / w0[l - 1]) > EOFX_PEAKED_RATIO;
    }
    int live = 0;
    host_zchol_rinv(H, l, T, 1e-13, nullptr, &live, dref.data(), 1e-10);
    CHK(right_mul(Vs, small_pad, T, l, LP, Zs));
    if (live > 0) {
      CHK(project(Zs, nb, Vs, false));
      CHK(gram_h(Vs, small_pad));
      host_zchol_rinv(H, l, T, 1e-13, nullptr, &live, ones.data(), 0.25);
      CHK(right_mul(Vs, small_pad, T, l, LP, Zs));
    }
    if (trace) fprintf(stderr, "[eofx_rsvd_c64] product %d: %d live columns in the next block (%d blocks)\n", ctx->last_iters, live, nb + (live > 0));
    if (live == 0) {
      exhausted = true;        // the Krylov space is invariant: every product of its blocks is known
      return EOFX_OK;
    }
    CHK(copy_block(Kw, Zs, nb));
    ++nb;
    return EOFX_OK;
  };
  // Rayleigh-Ritz over the first nbr blocks: H = K^H M K from the columns K^H W_i Rf[i] (i < nW) and, when the newest block has no
  // product yet, P^H P of its tall panel (Hqq).  -> leading l eigenvectors X (order nbr l), values wv; res[j] (with_res): the
  // norm of the part of M K y_j outside the first nbr blocks, read off the coupling to block nbr.
  std::vector<double> Xr, Xi, wv, hCf;
  auto rayleigh_ritz = [&](int nbr, bool with_last, std::vector<double>* res) -> int {
    const int nWr = std::min(nW, nbr);
    const int nrow = res ? std::min(nb, nbr + 1) : nbr;         // one more block row: the coupling
    CHK(launch_xgram(ctx, Kw, ldk, nrow * LP, Ww, ldk, nWr * LP, small_pad, Cf));
    hCf.resize((size_t)nrow * LP * nWr * LP);
    HIPCHK(hipMemcpyAsync(hCf.data(), Cf, sizeof(double) * hCf.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int64_t ldc = (int64_t)nWr * LP;
    const int m = nbr * l;
    std::vector<std::vector<zdouble>> raw((size_t)nrow * nbr);
    for (int i = 0; i < nWr; ++i)
      for (int j = 0; j < nrow; ++j) {
        cplx_block(&hCf[(size_t)j * LP * ldc + (size_t)i * LP], ldc, blk);
        raw[(size_t)j * nbr + i] = Rf[i].empty() ? blk : zmatmul(blk, Rf[i]);
      }
    if (with_last) raw[(size_t)(nbr - 1) * nbr + nbr - 1] = Hqq;
    std::vector<double> Hr((size_t)m * m, 0.0), Hi((size_t)m * m, 0.0);
    for (int a = 0; a < nbr; ++a)
      for (int c = a; c < nbr; ++c) {
        const std::vector<zdouble>& u = raw[(size_t)a * nbr + c];     // block (a, c)
        const std::vector<zdouble>& v = raw[(size_t)c * nbr + a];     // block (c, a): its conjugate transpose is another reading of (a, c)
        if (u.empty() && v.empty()) continue;
        const double wu = u.empty() ? 0.0 : (v.empty() ? 1.0 : 0.5), wvv = v.empty() ? 0.0 : (u.empty() ? 1.0 : 0.5);
        for (int i = 0; i < l; ++i)
          for (int j = 0; j < l; ++j) {
            zdouble val(0.0, 0.0);
            if (!u.empty()) val += wu * u[(size_t)i * l + j];
            if (!v.empty()) val += wvv * std::conj(v[(size_t)j * l + i]);
            const size_t ij = (size_t)(a * l + i) * m + c * l + j, ji = (size_t)(c * l + j) * m + a * l + i;
            Hr[ij] = val.real();
            Hi[ij] = val.imag();
            if (a != c) {
              Hr[ji] = val.real();
              Hi[ji] = -val.imag();
            }
          }
      }
    for (double v : Hr)
      if (!std::isfinite(v)) return set_err(ctx, EOFX_ERR_LINALG, "SVD failed. This may be due to isolated NaN values in the data.");
    wv.assign(l, 0.0);
    Xr.assign((size_t)m * l, 0.0);
    Xi.assign((size_t)m * l, 0.0);
    const auto t_rr0 = std::chrono::steady_clock::now();
    const int rc_rr = hosteig::zheigh_top(Hr.data(), Hi.data(), m, l, wv.data(), Xr.data(), Xi.data());
    if (trace) fprintf(stderr, "[eofx_rsvd_c64] host Rayleigh-Ritz solve, order %d: %.2f ms\n", m,
                       1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rr0).count());
    if (rc_rr != 0) {
      // general-purpose route (real symmetric embedding): slower, no assumptions
      std::vector<zdouble> Hz((size_t)m * m), Vz;
      for (size_t e = 0; e < Hz.size(); ++e) Hz[e] = zdouble(Hr[e], Hi[e]);
      std::vector<double> wz;
      if (host_heigh(Hz, m, wz, Vz) != EOFX_OK) return set_err(ctx, EOFX_ERR_LINALG, "complex SVD: Rayleigh-Ritz eigen-solver failed");
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < l; ++j) {
          Xr[(size_t)i * l + j] = Vz[(size_t)i * m + j].real();
          Xi[(size_t)i * l + j] = Vz[(size_t)i * m + j].imag();
        }
      std::copy(wz.begin(), wz.begin() + l, wv.begin());
    }
    if (res) {
      res->assign(l, 0.0);
      if (nrow > nbr)
        for (int j = 0; j < l; ++j) {
          double r2 = 0.0;
          for (int i2 = 0; i2 < l; ++i2) {           // row i2 of block nbr of M K y_j
            zdouble acc(0.0, 0.0);
            for (int c = 0; c < nWr; ++c) {
              const std::vector<zdouble>& cb = raw[(size_t)nbr * nbr + c];
              if (cb.empty()) continue;
              for (int t = 0; t < l; ++t) acc += cb[(size_t)i2 * l + t] * zdouble(Xr[(size_t)(c * l + t) * l + j], Xi[(size_t)(c * l + t) * l + j]);
            }
            r2 += std::norm(acc);
          }
          (*res)[j] = std::sqrt(r2);
        }
    }
    return EOFX_OK;
  };
  // coefficient stack for a linear combination of per-block panels: block b gets Rf[b] y_b (with_rf) or y_b
  std::vector<double> hEall;
  auto coeff_stack = [&](int nbr, bool with_rf) -> int {
    hEall.assign((size_t)nbr * LP * LP, 0.0);
    std::vector<double> eb;
    std::vector<zdouble> yb((size_t)l * l);
    for (int b = 0; b < nbr; ++b) {
      for (int i = 0; i < l; ++i)
        for (int j = 0; j < l; ++j) yb[(size_t)i * l + j] = zdouble(Xr[(size_t)(b * l + i) * l + j], Xi[(size_t)(b * l + i) * l + j]);
      const std::vector<zdouble> cb = (with_rf && !Rf[b].empty()) ? zmatmul(Rf[b], yb) : yb;
      embed_right(cb, l, l, LP, LP, eb);
      std::copy(eb.begin(), eb.end(), hEall.begin() + (size_t)b * LP * LP);
    }
    HIPCHK(hipMemcpyAsync(Eall, hEall.data(), sizeof(double) * hEall.size(), hipMemcpyHostToDevice, ctx->stream));
    return EOFX_OK;
  };
  // thick restart when the space is full: blocks 0 .. nb-2 (all multiplied) collapse to their leading Ritz block X = K y, with
  // A_op X and M X as the same combination of the kept panels; the newest block (not yet multiplied) follows it
  auto compress = [&]() -> int {
    const int nbr = nb - 1;
    CHK(rayleigh_ritz(nbr, false, nullptr));
    if (trace) fprintf(stderr, "[eofx_rsvd_c64] thick restart after %d products: %d blocks -> Ritz block + newest block\n", ctx->last_iters, nb);
    CHK(coeff_stack(nbr, false));
    CHK(launch_matmul_gen(ctx, Kw, ldk, 64, 1 << 20, small_pad, nbr * LP, Eall, LP, nullptr, Vs));        // X
    HIPCHK(hipStreamSynchronize(ctx->stream));
    CHK(coeff_stack(nbr, true));
    CHK(launch_matmul_gen(ctx, Ww, ldk, 64, 1 << 20, small_pad, nbr * LP, Eall, LP, nullptr, Ws));        // M X
    CHK(launch_matmul_gen(ctx, Pt, LP, tall_pad * LP, LP / 64, tall_pad, nbr * LP, Eall, LP, nullptr, Yt)); // A_op X
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpyAsync(Pt, Yt, sizeof(float) * (size_t)tall_pad * LP, hipMemcpyDeviceToDevice, ctx->stream));
    CHK(copy_block(Kw, Vs, 0));
    CHK(copy_block(Ww, Ws, 0));
    CHK(copy_block(Kw, Zs, 1));
    for (auto& rf : Rf) rf.clear();
    nb = 2;
    nW = 1;
    return EOFX_OK;
  };

  std::vector<double> w;
  std::vector<zdouble> Uh;
  if (krylov) {
    CHK(orth(Zs, small_pad, Vs));                    // Z_0: the orthonormalised start panel
    std::swap(Zs, Vs);
    CHK(copy_block(Kw, Zs, 0));
    nb = 1;
    // "converge" (round 6): continue until every WANTED singular value is good to 2e-6 and every gap-separated wanted vector to
    // |cos| >= 1 - 5e-6 -- inside the parity tolerances (1e-5 on the values against the float64 oracle, |cos| >= 1 - 1e-5 for modes
    // separated by 2 %) -- or 20 products have been made (lobpcg's own limit under svds).  The error of a Ritz value theta_j of
    // M = A_op^H A_op (theta = sigma^2) is estimated from its OWN history: Ritz values of a growing Krylov space rise monotonically
    // towards their eigenvalues and, product after product, geometrically; with D_c the rise between two checks d products apart and
    // rho = D_c / D_{c-1} the ratio of two successive rises, the distance still to go is D_c rho / (1 - rho) (first estimate,
    // without a ratio: D_c).  Round 5 asked for a residual |M x - theta x| <= 1e-5 theta instead: values were then good to 1e-8 and
    // a field whose last wanted modes sit a per cent above a flat bulk always paid the full 20 products; a residual bound with
    // the gap to the nearest Ritz value (tried first this round) is 250 times too pessimistic there
    // (profiles/r06_r9_evidence*.txt).  A check is a host Rayleigh-Ritz solve (0.6 ms at order 120, 5 at 240, 16 at 360 ~ one
    // product): every third product from three before scikit-learn's count on (a field that "auto" would have served stops there at
    // the price of two small solves), and none any more once the observed rate says the limit of 20 products comes first (modes
    // inside a flat bulk: the reference's lobpcg runs into its iteration limit on those as well).
    const double val_tol = 4e-6, vec_tol = 1e-5, sep_rel = 0.04;
    const int check_every = 3;
    int next_check = std::max(std::max(auto_count, it_min) - check_every, it_min);
    double worst_prev = -1.0;
    std::vector<double> th_prev, rise_prev;
    while (ctx->last_iters < n_iter && !exhausted) {
      if (nb == nbmax) CHK(compress());
      CHK(lanczos_step());
      if (adaptive && !exhausted && ctx->last_iters >= next_check && ctx->last_iters < n_iter) {
        CHK(rayleigh_ritz(nb - 1, false, nullptr));
        double worst = th_prev.empty() ? 1e300 : 0.0;       // largest (estimate / tolerance) over the wanted modes: <= 1 = converged
        std::vector<double> rise(k, 0.0);
        for (int j = 0; j < k && !th_prev.empty(); ++j) {
          const double th = std::max(wv[j], 1e-300);
          rise[j] = std::fabs(wv[j] - th_prev[j]) / th;
          double rho = 0.5;                                 // no ratio yet: the rise itself is the estimate
          if (!rise_prev.empty() && rise_prev[j] > 0.0) rho = std::min(0.7, std::max(0.02, rise[j] / rise_prev[j]));
          const double est = rise[j] * rho / (1.0 - rho);
          double score = est / val_tol;
          double gap = 1e300;                               // relative gap to the nearest other Ritz value
          if (j > 0) gap = std::min(gap, (wv[j - 1] - wv[j]) / th);
          if (j + 1 < l) gap = std::min(gap, (wv[j] - wv[j + 1]) / th);
          if (gap >= sep_rel && gap < 1e300) score = std::max(score, est / gap / vec_tol);     // sin^2 of the vector's angle ~ error / gap
          worst = std::max(worst, score);
        }
        if (trace) fprintf(stderr, "[eofx_rsvd_c64] after %d products: worst (error estimate / tolerance) over the leading %d Ritz values %.3e\n", ctx->last_iters, k, worst);
        if (worst <= 1.0) break;
        next_check = ctx->last_iters + check_every;
        if (!rise_prev.empty() && worst_prev > 0.0 && worst < 1e299) {      // two estimates: will the limit come first?
          const double f = worst / worst_prev;                               // factor per check interval
          const double checks_needed = f < 1.0 ? std::log(worst) / std::log(1.0 / f) : 1e9;
          if ((double)ctx->last_iters + checks_needed * check_every > (double)n_iter + check_every) {
            next_check = n_iter + 1;
            if (trace) fprintf(stderr, "[eofx_rsvd_c64] at this rate (x %.3g per %d products) the limit of %d products comes first: no further checks\n", f, check_every, n_iter);
          }
        }
        if (!th_prev.empty()) {
          rise_prev = rise;
          worst_prev = worst;
        }
        th_prev.assign(wv.begin(), wv.begin() + k);
      }
    }
    bool with_last = false;
    if (!exhausted) {          // the newest block's panel: its diagonal block of H is P^H P
      const int b = nb - 1;
      float* slot = Pt + (size_t)b * tall_pad * LP;
      Rf[b].clear();
      if (orth_rest || ctx->last_iters == 0) {
        CHK(fwd(Zs, Yt, pp));
        CHK(gram_h(Yt, tall_pad, true));
        Hqq = H;
        host_zchol_rinv(H, l, T, 1e-13, &Rf[b]);
        CHK(right_mul(Yt, tall_pad, T, l, LP, slot));
      } else {
        CHK(fwd(Zs, slot, pp));
        CHK(gram_h(slot, tall_pad, true));
        Hqq = H;
      }
      with_last = true;
    }
    CHK(rayleigh_ritz(nb, with_last, nullptr));
    if (trace) fprintf(stderr, "[eofx_rsvd_c64] Rayleigh-Ritz over %d blocks (order %d) after %d products: leading Ritz values %.6e %.6e ... %.6e\n", nb, nb * l, ctx->last_iters, wv[0], l > 1 ? wv[1] : 0.0, wv[l - 1]);
    CHK(coeff_stack(nb, true));
    CHK(launch_matmul_gen(ctx, Pt, LP, tall_pad * LP, LP / 64, tall_pad, nb * LP, Eall, LP, nullptr, Yt));   // A_op K y
    CHK(orth(Yt, tall_pad, Qt, true));
    CHK(orth(Qt, tall_pad, Yt, true));                     // Q in Yt (CholeskyQR2)
    CHK(bwd(Yt, Ws, pf));                            // B^H, B = Q^H A_op
    CHK(gram_h(Ws, small_pad));                      // B B^H
    if (host_heigh(H, l, w, Uh) != EOFX_OK) return set_err(ctx, EOFX_ERR_LINALG, "complex SVD: Hermitian eigen-solver failed");
  } else {
    bool orth_rest = orth_always;
    std::vector<double> ritz_prev;
    int calm = 0;
    for (int it = 0; it < n_iter; ++it) {
      CHK(fwd(Zs, Yt, pp));
      if (it == 0 || orth_rest) {
        CHK(orth(Yt, tall_pad, Qt, true));
        CHK(bwd(Qt, Ws, pp));
      } else {
        CHK(bwd(Yt, Ws, pp));
      }
      CHK(gram_h(Ws, small_pad));
      ctx->last_iters = it + 1;
      if (it == 0 && !orth_always && n_iter > 1) {   // peaked spectrum?  the Hermitian Gram matrix is on the host already
        std::vector<zdouble> V0;
        std::vector<double> w0;
        orth_rest = host_heigh(H, l, w0, V0) != EOFX_OK || !(w0[l - 1] > 0.0) || std::sqrt(w0[0] / w0[l - 1]) > EOFX_PEAKED_RATIO;
      }
      bool done = false;
      if (adaptive) {
        std::vector<zdouble> V0;
        std::vector<double> w0;
        if (host_heigh(H, l, w0, V0) == EOFX_OK) {
          double worst = 0.0;
          if (ritz_prev.size() == (size_t)k)
            for (int j = 0; j < k; ++j) worst = std::max(worst, std::fabs(w0[j] - ritz_prev[j]) / std::max(w0[j], 1e-300));
          else
            worst = 1.0;
          ritz_prev.assign(w0.begin(), w0.begin() + k);
          calm = worst <= 1e-6 ? calm + 1 : 0;
          if (trace) fprintf(stderr, "[eofx_rsvd_c64] iteration %d: max relative change of the leading %d Ritz values %.3e\n", it + 1, k, worst);
          done = calm >= 2 && it + 1 >= it_min;
        }
      }
      host_zchol_rinv(H, l, T, 1e-13);
      CHK(right_mul(Ws, small_pad, T, l, LP, Zs));
      if (done) break;
    }
    CHK(fwd(Zs, Yt, pp));
    CHK(orth(Yt, tall_pad, Qt, true));
    CHK(orth(Qt, tall_pad, Yt, true));                     // Q in Yt (CholeskyQR2)
    CHK(bwd(Yt, Ws, pf));                            // B^H, B = Q^H A_op
    CHK(gram_h(Ws, small_pad));                      // B B^H
    if (host_heigh(H, l, w, Uh) != EOFX_OK) return set_err(ctx, EOFX_ERR_LINALG, "complex SVD: Hermitian eigen-solver failed");
  }
  std::vector<zdouble> M1((size_t)l * k), M2((size_t)l * k);
  std::vector<float> hs(k);
  for (int j = 0; j < k; ++j) {
    const double sv = std::sqrt(std::max(w[j], 0.0));
    hs[j] = (float)sv;
    const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
    for (int i = 0; i < l; ++i) {
      M1[(size_t)i * k + j] = Uh[(size_t)i * l + j];
      M2[(size_t)i * k + j] = Uh[(size_t)i * l + j] * inv;
    }
  }
  CHK(right_mul(Yt, tall_pad, M1, k, Lo, Tv));     // A_op = Tall diag(s) Small^H
  CHK(right_mul(Ws, small_pad, M2, k, Lo, Sv));
  // Numerically null modes (more modes asked for than the matrix has rank: the analytic signal of a short series has about
  // n / 2 + 1 independent rows): B^H u / s is rounding noise there.  The reference's solver ends with a dense SVD of A V
  // (scipy svds, _svds.py), whose left vectors are orthonormal whatever the values; here such columns of the small-side factor
  // are re-orthonormalised on the host against all the columns before them (two rounds of Gram-Schmidt in float64; a column
  // that lay inside their span is replaced by the first unit vector that does not).  Values and the tall side are untouched.
  {
    int first_null = k;
    for (int j = k - 1; j >= 0 && (double)hs[j] <= 3e-6 * (double)hs[0]; --j) first_null = j;
    if (first_null < k && hs[0] > 0.f) {
      std::vector<float> hp((size_t)small * Lo);
      HIPCHK(hipMemcpy2DAsync(hp.data(), sizeof(float) * Lo, Sv, sizeof(float) * Lo, sizeof(float) * Lo, (size_t)small, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      auto col = [&](int j, std::vector<zdouble>& v) {
        v.resize((size_t)small);
        for (int64_t r = 0; r < small; ++r) v[(size_t)r] = zdouble(hp[(size_t)r * Lo + j], hp[(size_t)r * Lo + ko + j]);
      };
      std::vector<std::vector<zdouble>> basis((size_t)k);
      for (int j = 0; j < first_null; ++j) col(j, basis[(size_t)j]);
      auto project_out = [&](std::vector<zdouble>& v, int upto) {
        for (int round = 0; round < 2; ++round)
          for (int c = 0; c < upto; ++c) {
            zdouble dot(0.0, 0.0);
            for (int64_t r = 0; r < small; ++r) dot += std::conj(basis[(size_t)c][(size_t)r]) * v[(size_t)r];
            for (int64_t r = 0; r < small; ++r) v[(size_t)r] -= dot * basis[(size_t)c][(size_t)r];
          }
        double nn = 0.0;
        for (int64_t r = 0; r < small; ++r) nn += std::norm(v[(size_t)r]);
        return std::sqrt(nn);
      };
      int64_t next_unit = 0;
      for (int j = first_null; j < k; ++j) {
        std::vector<zdouble> v;
        col(j, v);
        double nn0 = 0.0;
        for (const zdouble& x : v) nn0 += std::norm(x);
        double nn = std::isfinite(nn0) && nn0 > 0.0 ? project_out(v, j) / std::sqrt(nn0) : 0.0;
        while (!(nn > 1e-3) && next_unit < small) {          // inside the span of the others (or not finite): a unit vector instead
          v.assign((size_t)small, zdouble(0.0, 0.0));
          v[(size_t)next_unit++] = zdouble(1.0, 0.0);
          nn = project_out(v, j);
          if (nn > 0.1) break;
          nn = 0.0;
        }
        double nrm = 0.0;
        for (const zdouble& x : v) nrm += std::norm(x);
        nrm = std::sqrt(nrm);
        for (int64_t r = 0; r < small; ++r) {
          const zdouble x = nrm > 0.0 ? v[(size_t)r] / nrm : zdouble(0.0, 0.0);
          v[(size_t)r] = x;
          hp[(size_t)r * Lo + j] = (float)x.real();
          hp[(size_t)r * Lo + ko + j] = (float)x.imag();
        }
        basis[(size_t)j] = v;
      }
      HIPCHK(hipMemcpy2DAsync(Sv, sizeof(float) * Lo, hp.data(), sizeof(float) * Lo, sizeof(float) * Lo, (size_t)small, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  const float* Vp = transposed ? Tv : Sv;
  const float* Up = transposed ? Sv : Tv;
  std::vector<double> sign(k, 1.0);
  if (flip_signs) {
    // VT = conj(V)^T; numpy's max / min of complex numbers are lexicographic (real part, then imaginary part)
    ARENA(int64_t, amax, Lo);
    ARENA(int64_t, amin, Lo);
    CHK(panel_colargminmax(ctx, Vp, p, Lo, amax, amin, A->masked ? A->aff + 2 * A->p_pad : nullptr));   // (scale plane: 0 at masked features)
    ARENA(float, picks, 4 * (size_t)k);
    hipLaunchKernelGGL(cpanel_pick_kernel, dim3((k + 63) / 64), dim3(64), 0, ctx->stream, Vp, Lo, k, amax, amin, picks);
    KCHK();
    std::vector<float> c(4 * (size_t)k);
    HIPCHK(hipMemcpyAsync(c.data(), picks, sizeof(float) * 4 * k, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<double> mag(2 * (size_t)k);
    for (int j = 0; j < k; ++j) {
      mag[j] = std::hypot((double)c[4 * j], (double)c[4 * j + 1]);
      mag[k + j] = std::hypot((double)c[4 * j + 2], (double)c[4 * j + 3]);
    }
    if (shd) {
      // global lexicographic extrema over the slices: the largest / smallest real part wins (all-reduce(max) of [re_max | -re_min]),
      // the rank that holds it contributes the magnitude of its entry (all-reduce(max) of float64, -1 from the others)
      ARENA(float, dre, 2 * (size_t)k);
      ARENA(double, dmag, 2 * (size_t)k);
      std::vector<float> re(2 * (size_t)k), gre(2 * (size_t)k);
      for (int j = 0; j < k; ++j) {
        re[j] = p > 0 ? c[4 * j] : -INFINITY;
        re[k + j] = p > 0 ? -c[4 * j + 2] : -INFINITY;
      }
      HIPCHK(hipMemcpyAsync(dre, re.data(), sizeof(float) * 2 * k, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      CHK(comm_allreduce(ctx, dre, 2 * (int64_t)k, 0, 1));
      HIPCHK(hipMemcpyAsync(gre.data(), dre, sizeof(float) * 2 * k, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      for (int j = 0; j < 2 * k; ++j)
        if (!(p > 0 && re[j] == gre[j])) mag[j] = -1.0;
      HIPCHK(hipMemcpyAsync(dmag, mag.data(), sizeof(double) * 2 * k, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      CHK(comm_allreduce(ctx, dmag, 2 * (int64_t)k, 1, 1));
      HIPCHK(hipMemcpyAsync(mag.data(), dmag, sizeof(double) * 2 * k, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (int j = 0; j < k; ++j)   // |max| >= |min| in numpy's lexicographic complex order (real part decides)
      sign[j] = mag[j] >= mag[k + j] ? 1.0 : -1.0;
  }
  // interleaved complex64 exports
  auto export_c = [&](const float* P, int64_t rows, float* dst) -> int {
    if (!dst) return EOFX_OK;
    ArenaScope sc(ctx);
    double* dsign = arena_alloc<double>(ctx, k);
    if (!dsign) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (export sign)");
    CHK(copy_in(ctx, dsign, sign.data(), sizeof(double) * k));
    float* t = dst;
    const bool dev = is_device_ptr(dst);
    if (!dev) {
      t = arena_alloc<float>(ctx, (size_t)rows * k * 2);
      if (!t) return set_err(ctx, EOFX_ERR_NOMEM, "arena exhausted (export staging)");
    }
    const int64_t total = rows * k;
    hipLaunchKernelGGL(cpanel_export_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, ctx->stream, P,
                       rows, Lo, k, dsign, t);
    KCHK();
    if (!dev) CHK(copy_out(ctx, dst, t, sizeof(float) * total * 2));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EOFX_OK;
  };
  CHK(export_c(Up, n, U));
  CHK(export_c(Vp, p, V));
  if (is_device_ptr(s)) HIPCHK(hipMemcpy(s, hs.data(), sizeof(float) * k, hipMemcpyHostToDevice));
  else std::memcpy(s, hs.data(), sizeof(float) * k);
  return EOFX_OK;
}

extern "C" int eofx_rsvd_c64(eofx_ctx* ctx, const eofx_mat* A, const eofx_mat* B, int k, int n_oversamples, int n_iter,
                             const float* omega, int flip_signs, float* U, float* s, float* V) {
  if (!B) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  return rsvd_c64_impl(ctx, A, B, nullptr, k, n_oversamples, n_iter, omega, flip_signs, U, s, V);
}

// The decomposition of the ANALYTIC SIGNAL of a resident real matrix without writing its imaginary part:
// Z = A + i H(A) = (I + i Hc) A with Hc the n x n operator of the Hilbert stage along the samples (padding, decay as in
// eofx_hilbert_f32; get_hilbert_operator).  Same arguments, rules and outputs as eofx_rsvd_c64 on (A, eofx_hilbert_f32(A));
// every product streams the real field once and applies Hc on the sample side (CplxOps operator mode), so a pass moves
// half the bytes of the two-part form and the field needs no second resident copy.
extern "C" int eofx_rsvd_hilbert_c64(eofx_ctx* ctx, const eofx_mat* A, int padding, double decay_factor, int k,
                                     int n_oversamples, int n_iter, const float* omega, int flip_signs, float* U, float* s,
                                     float* V) {
  if (!ctx || !A) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (padding && !(decay_factor > 0.0)) return set_err(ctx, EOFX_ERR_ARG, "decay_factor must be positive");
  if (A->n > EOFX_HILBERT_OP_MAX_N)
    return set_err(ctx, EOFX_ERR_ARG, "the resident Hilbert operator is limited to %d samples (got %lld): use eofx_hilbert_f32 + eofx_rsvd_c64",
                   EOFX_HILBERT_OP_MAX_N, (long long)A->n);
  ENTER(ctx);
  const eofx_mat* Hop = nullptr;
  CHK(get_hilbert_operator(ctx, A->n, padding ? 1 : 0, decay_factor, &Hop));
  return rsvd_c64_impl(ctx, A, nullptr, Hop, k, n_oversamples, n_iter, omega, flip_signs, U, s, V);
}

extern "C" int eofx_rsvd_sharded_c64(eofx_ctx* ctx, const eofx_mat* A, const eofx_mat* B, int64_t p_total, int k, int n_oversamples,
                                     int n_iter, const float* omega, int flip_signs, float* U, float* s, float* V) {
  if (!B || p_total <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  return rsvd_c64_impl(ctx, A, B, nullptr, k, n_oversamples, n_iter, omega, flip_signs, U, s, V, p_total);
}

extern "C" int eofx_rsvd_hilbert_sharded_c64(eofx_ctx* ctx, const eofx_mat* A, int64_t p_total, int padding, double decay_factor,
                                             int k, int n_oversamples, int n_iter, const float* omega, int flip_signs, float* U,
                                             float* s, float* V) {
  if (!ctx || !A || p_total <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  if (padding && !(decay_factor > 0.0)) return set_err(ctx, EOFX_ERR_ARG, "decay_factor must be positive");
  if (A->n > EOFX_HILBERT_OP_MAX_N)
    return set_err(ctx, EOFX_ERR_ARG, "the resident Hilbert operator is limited to %d samples (got %lld): use eofx_hilbert_f32 + eofx_rsvd_sharded_c64",
                   EOFX_HILBERT_OP_MAX_N, (long long)A->n);
  if (!ctx->comm) return set_err(ctx, EOFX_ERR_ARG, "no communicator attached (eofx_ctx_comm_init_rccl / eofx_ctx_comm_set_callback)");
  ENTER(ctx);
  const eofx_mat* Hop = nullptr;
  {   // building the operator (host memory, an upload) can fail on one rank alone: the ranks agree before the first collective
    const int rc_local = get_hilbert_operator(ctx, A->n, padding ? 1 : 0, decay_factor, &Hop);
    const std::string err_local = rc_local != EOFX_OK ? ctx->err : std::string();
    int verdict = 0;
    CHK(comm_vote(ctx, rc_local != EOFX_OK ? 2 : 0, &verdict));
    if (verdict != 0)
      return rc_local != EOFX_OK ? (ctx->err = err_local, rc_local) : set_err(ctx, EOFX_ERR_HIP, "the sharded Hilbert decomposition failed on another rank");
  }
  return rsvd_c64_impl(ctx, A, nullptr, Hop, k, n_oversamples, n_iter, omega, flip_signs, U, s, V, p_total);
}

// sum of squares of the resident matrix in float64 (fixed tree); for zero-mean columns
// total variance = sumsq / (n - 1)
extern "C" int eofx_mat_sumsq_f64(eofx_ctx* ctx, const eofx_mat* m, double* out) {
  if (!ctx || !m || !out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  const int nb = 2048;
  CHK(arena_reserve(ctx, nb * sizeof(double) + 4096));
  ArenaScope scope(ctx);
  ARENA(double, part, nb);
  if (!m->X) CHK(ensure_Xt(ctx, m));
  const float* base = m->X ? m->X : m->Xt;   // the same elements (raw mode holds the sample-contiguous layout only)
  hipLaunchKernelGGL(dotprod_part_kernel, dim3(nb), dim3(256), 0, ctx->stream, base, base, m->n_pad * m->p_pad, part);
  KCHK();
  std::vector<double> hp(nb);
  HIPCHK(hipMemcpyAsync(hp.data(), part, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double t = 0.0;
  for (int i = 0; i < nb; ++i) t += hp[i];
  *out = t;
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// Gaussian sketch matrix, bit-identical to numpy's legacy stream
//   np.random.RandomState(seed).normal(size=(rows, cols)).astype(np.float32)
// which is what scikit-learn draws for randomized_svd (extmath.py _randomized_range_finder), so
// `random_state` keeps its meaning.  MT19937 (init_genrand seeding), 53-bit doubles, Marsaglia
// polar method with the cached second deviate (numpy legacy_gauss).  The uniform stream and the
// accept/reject decisions are inherently sequential; the log/sqrt transforms of the accepted pairs
// are independent and run on worker threads (same libm, same results).
// ------------------------------------------------------------------------------------
namespace {
struct MT19937 {
  uint32_t mt[624];
  int idx;
  explicit MT19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void refill() {
    int k = 0;
    for (; k < 624 - 397; ++k) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
      mt[k] = mt[k + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
    }
    for (; k < 623; ++k) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
      mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
    }
    const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ (-(int32_t)(y & 1u) & 0x9908b0dfu);
    idx = 0;
  }
  inline uint32_t next() {
    if (idx >= 624) refill();
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  inline double next_double() {
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};
}  // namespace

// Parallel, still bit-identical.  Every candidate pair of the polar method consumes exactly four MT19937 words whether
// it is accepted or not, so candidate i always sits at words 4i..4i+3 of the stream.  Only the raw MT recurrence is
// sequential -- written as ONE linear recurrence x[i] = x[i-227] ^ f(x[i-624], x[i-623]) over the whole stream (no
// 624-word refill blocks, no copies), eight words per AVX2 step where the host has it: 0.55 ns per word on the GPU box,
// bound by one core's store bandwidth (40 MB for a 129 600 x 30 sketch).  Tempering, the conversion to doubles, the accept test and the log / sqrt of the accepted
// pairs are data-parallel over candidates: every worker thread turns its contiguous share of the candidates into its
// own list of deviates in ONE pass, and the lists are concatenated in order.
namespace {
#define EOFX_MT_FILL_BODY                                                                                     \
  for (int64_t i = from; i < count; ++i) {                                                                     \
    const uint32_t y = (x[i - 624] & 0x80000000u) | (x[i - 623] & 0x7fffffffu);                               \
    x[i] = x[i - 227] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & 0x9908b0dfu);                            \
  }
#if !defined(__HIP_DEVICE_COMPILE__)
// eight words per step: every load lies at least 220 words behind the eight words being written.  [from, count): the
// words [0, from) are final (from >= 624), so the stream can be extended chunk by chunk.
__attribute__((target("avx2"))) void mt_fill_avx2_range(uint32_t* __restrict__ x, int64_t from, int64_t count) {
  const __m256i upper = _mm256_set1_epi32((int)0x80000000u), lower = _mm256_set1_epi32(0x7fffffff);
  const __m256i one = _mm256_set1_epi32(1), matrix = _mm256_set1_epi32((int)0x9908b0dfu), zero = _mm256_setzero_si256();
  int64_t i = from;
  for (; i + 8 <= count; i += 8) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(x + i - 624));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(x + i - 623));
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(x + i - 227));
    const __m256i y = _mm256_or_si256(_mm256_and_si256(a, upper), _mm256_and_si256(b, lower));
    const __m256i mag = _mm256_and_si256(_mm256_sub_epi32(zero, _mm256_and_si256(y, one)), matrix);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(x + i), _mm256_xor_si256(_mm256_xor_si256(c, _mm256_srli_epi32(y, 1)), mag));
  }
  for (; i < count; ++i) {
    const uint32_t y = (x[i - 624] & 0x80000000u) | (x[i - 623] & 0x7fffffffu);
    x[i] = x[i - 227] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & 0x9908b0dfu);
  }
}
#else
inline void mt_fill_avx2_range(uint32_t*, int64_t, int64_t) {}   // (device pass of the single-source compile: host code only)
#endif
void mt_fill_generic_range(uint32_t* __restrict__ x, int64_t from, int64_t count) { EOFX_MT_FILL_BODY }
#undef EOFX_MT_FILL_BODY
inline uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
}  // namespace

namespace {
// A small persistent pool for the data-parallel half of the sketch generator (starting a std::thread costs ~40 us; a
// 10000 x 60 sketch is ~1.5 ms of work in total, so eighteen fresh threads per call were a third of its wall time).
struct SketchPool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  std::deque<std::function<void()>> tasks;
  int pending = 0;
  bool stop = false;
  void ensure(int n) {
    std::lock_guard<std::mutex> lk(mu);
    while ((int)threads.size() < n) threads.emplace_back([this] { loop(); });
  }
  // Keep the workers on the cores that share the calling thread's L3 (EOFX_SKETCH_PIN=1; Linux): the stream buffer then moves
  // between the producer and its consumers inside one cache instead of across the socket.
  int pinned_cpu = -1;
  int domain_threads = 0;     // hardware threads that share the caller's L3 (0: unknown / not pinned)
  void pin_near_caller() {
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
    const int cpu = sched_getcpu();
    if (cpu < 0 || cpu == pinned_cpu) return;
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE* f = fopen(path, "r");
    if (!f) return;
    char buf[512] = {0};
    const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!ok) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    int count = 0;
    for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
      int a = 0, b = 0;
      const int got = sscanf(tok, "%d-%d", &a, &b);
      if (got == 1) b = a;
      if (got < 1) continue;
      for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); ++count; }
    }
    if (count < 2) return;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& t : threads) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    pinned_cpu = cpu;
    domain_threads = count;
#endif
  }
  void loop() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [this] { return stop || !tasks.empty(); });
        if (stop && tasks.empty()) return;
        job = std::move(tasks.front());
        tasks.pop_front();
      }
      job();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      tasks.push_back(std::move(f));
      ++pending;
    }
    cv_work.notify_one();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [this] { return pending == 0; });
  }
  ~SketchPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
};
SketchPool& sketch_pool() {
  static SketchPool* pool = new SketchPool();     // leaked on purpose: worker threads must not be joined at exit order
  return *pool;
}
std::mutex g_sketch_call;    // one generation at a time (the pool's completion counter is per call)
// spin politely: a few thousand pauses (the waits here are microseconds), then give the core away -- on a host with fewer free
// cores than workers a pure spin would keep the producer off its core
struct SpinWait {
  int spins = 0;
  void operator()() {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (++spins < 4000) _mm_pause();
    else std::this_thread::yield();
#endif
  }
};
}  // namespace

extern "C" int eofx_sketch_gaussian_f32(uint32_t seed, int64_t rows, int64_t cols, float* out) {
  if (!out || rows < 0 || cols < 0) return EOFX_ERR_ARG;
  const int64_t total = rows * cols;
  if (total == 0) return EOFX_OK;
  const int64_t npairs = (total + 1) / 2;
  const int hw = (int)std::thread::hardware_concurrency();
  // eight workers: the producer thread is the bound from four on (profiles/r04_sketch_probe.txt), and eight fit the physical cores
  // of one L3 domain, where the workers are kept (SketchPool::pin_near_caller)
  int max_threads = std::max(1, std::min(8, hw > 1 ? hw - 1 : 1));
  if (const char* ev = getenv("EOFX_SKETCH_THREADS")) max_threads = std::max(1, atoi(ev));
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  std::lock_guard<std::mutex> call_lock(g_sketch_call);
  // x[0..623]: the generator state (init_genrand seeding); x[624 + w]: raw (untempered) output word w of the stream
  uint32_t state[624];
  {
    const MT19937 g(seed);
    std::memcpy(state, g.mt, sizeof(state));
  }
  int64_t done_pairs = 0;
  while (done_pairs < npairs) {
    const int64_t need = npairs - done_pairs;
    // acceptance probability pi/4: ask for slightly more candidates than expected, at least a few
    const int64_t ncand = (int64_t)((double)need / 0.7853981633974483 * 1.01) + 64;
    const int64_t nwords = 4 * ncand;
    // ONE grow-only buffer for the process (calls are serialised by g_sketch_call): 40 MB of fresh pages per call cost
    // more than the recurrence itself -- and a buffer per calling THREAD meant exactly that for callers that draw on a
    // short-lived worker thread (engine.SketchFuture: 12 ms instead of 5.8 for the 129 600 x 30 sketch of config 3)
    static std::unique_ptr<uint32_t[]> xbuf;
    static size_t xcap = 0;
    if (xcap < (size_t)(624 + nwords)) {
      xbuf.reset(new uint32_t[(size_t)(624 + nwords)]);   // uninitialised on purpose
      xcap = (size_t)(624 + nwords);
    }
    uint32_t* x = xbuf.get();
    std::memcpy(x, state, sizeof(state));
    const uint32_t* raw = x + 624;
    // The raw recurrence is sequential and runs on this thread, chunk by chunk.  The workers are woken ONCE per call and
    // then claim chunks from an atomic counter, spinning (politely) for the few microseconds until the chunk they hold has
    // been produced: a mutex + condition-variable hand-over per 16 K-word chunk cost more than the chunk's recurrence.
    // Deviates go to one grow-only buffer at their chunk's place; once every chunk is done the same workers copy them to
    // their final offsets.
    const int64_t chunk = 4096;                                     // candidates per chunk (16 K words): the deviate work of the LAST chunk is the tail of the call
    const int64_t nchunks = (ncand + chunk - 1) / chunk;
    static std::unique_ptr<float[]> dbuf;                          // [nchunks][2 chunk]: (f b, f a) per accepted pair, in stream order
    static size_t dcap = 0;
    if (dcap < (size_t)(2 * chunk * nchunks)) {
      dbuf.reset(new float[(size_t)(2 * chunk * nchunks)]);
      dcap = (size_t)(2 * chunk * nchunks);
    }
    float* const dev = dbuf.get();
    std::vector<int64_t> cnt((size_t)nchunks, 0), off((size_t)nchunks + 1, 0);
    std::atomic<int64_t> produced{0}, claim{0}, finished{0}, claim2{0};
    std::atomic<int> phase2{0};
    auto work = [&](int64_t c) {
      const int64_t lo = c * chunk, hi = std::min<int64_t>(ncand, lo + chunk);
      float* d = dev + 2 * chunk * c;
      int64_t q = 0;
      for (int64_t i = lo; i < hi; ++i) {
        const uint32_t w0 = mt_temper(raw[4 * i]) >> 5, w1 = mt_temper(raw[4 * i + 1]) >> 6;
        const uint32_t w2 = mt_temper(raw[4 * i + 2]) >> 5, w3 = mt_temper(raw[4 * i + 3]) >> 6;
        const double a = 2.0 * ((w0 * 67108864.0 + w1) / 9007199254740992.0) - 1.0;
        const double b = 2.0 * ((w2 * 67108864.0 + w3) / 9007199254740992.0) - 1.0;
        const double r = a * a + b * b;
        if (r >= 1.0 || r == 0.0) continue;
        const double f = std::sqrt(-2.0 * std::log(r) / r);
        d[q++] = (float)(f * b);      // returned first
        d[q++] = (float)(f * a);      // the cached deviate
      }
      cnt[(size_t)c] = q;
    };
    const int64_t obase = 2 * done_pairs;
    auto copy_out = [&](int64_t c) {     // deviates beyond `total` belong to later draws of the stream and are dropped
      const int64_t o = obase + off[(size_t)c];
      if (o >= total) return;
      const int64_t take = std::min<int64_t>(cnt[(size_t)c], total - o);
      std::memcpy(out + o, dev + 2 * chunk * c, sizeof(float) * (size_t)take);
    };
    auto worker = [&] {
      for (;;) {
        const int64_t c = claim.fetch_add(1, std::memory_order_relaxed);
        if (c >= nchunks) break;
        for (SpinWait sw; produced.load(std::memory_order_acquire) <= c;) sw();
        work(c);
        finished.fetch_add(1, std::memory_order_release);
      }
      for (SpinWait sw; !phase2.load(std::memory_order_acquire);) sw();
      for (;;) {
        const int64_t c = claim2.fetch_add(1, std::memory_order_relaxed);
        if (c >= nchunks) break;
        copy_out(c);
      }
    };
    const int nworkers = (max_threads > 1 && nchunks > 1) ? (int)std::min<int64_t>(max_threads, nchunks) : 0;
    SketchPool& pool = sketch_pool();
    if (nworkers) {
      pool.ensure(nworkers);
      static const bool pin = !(getenv("EOFX_SKETCH_PIN") && atoi(getenv("EOFX_SKETCH_PIN")) == 0);
      if (pin) pool.pin_near_caller();
      for (int w = 0; w < nworkers; ++w) pool.submit(worker);
    }
    int64_t filled = 624;                                            // words of x that hold final values
    for (int64_t c = 0; c < nchunks; ++c) {
      const int64_t upto = 624 + 4 * std::min<int64_t>(ncand, (c + 1) * chunk);
      // the recurrence reads up to 624 words back and writes only beyond `filled`: extending the buffer in place
      if (have_avx2) mt_fill_avx2_range(x, filled, upto);
      else mt_fill_generic_range(x, filled, upto);
      filled = upto;
      produced.store(c + 1, std::memory_order_release);
    }
    for (;;) {                                                        // this thread takes chunks too once the stream stands
      const int64_t c = claim.fetch_add(1, std::memory_order_relaxed);
      if (c >= nchunks) break;
      work(c);
      finished.fetch_add(1, std::memory_order_release);
    }
    for (SpinWait sw; finished.load(std::memory_order_acquire) < nchunks;) sw();
    for (int64_t c = 0; c < nchunks; ++c) off[(size_t)c + 1] = off[(size_t)c] + cnt[(size_t)c];
    phase2.store(1, std::memory_order_release);
    for (;;) {
      const int64_t c = claim2.fetch_add(1, std::memory_order_relaxed);
      if (c >= nchunks) break;
      copy_out(c);
    }
    if (nworkers) pool.wait();
    std::memcpy(state, x + nwords, sizeof(state));     // the state after these words (a rare second round continues here)
    const int64_t o = std::min<int64_t>(total, obase + off[(size_t)nchunks]);
    done_pairs = (o + 1) / 2;
    if (o >= total) break;
  }
  return EOFX_OK;
}

// ------------------------------------------------------------------------------------
// rotation of loadings (Varimax / Promax), panel-level steps
// ------------------------------------------------------------------------------------
// row norms of a panel (host|device float64 output of `rows` entries)
static int launch_rownorm(eofx_ctx* ctx, const float* P, int64_t rows, int64_t L, int64_t ld, double* out) {
  ArenaScope scope(ctx);
  ARENA(double, tmp, rows);
  hipLaunchKernelGGL(rownorm_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, ctx->stream, P, rows, L, ld, tmp);
  KCHK();
  HIPCHK(hipMemcpyAsync(out, tmp, sizeof(double) * rows, hipMemcpyDefault, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return EOFX_OK;
}
extern "C" int eofx_panel_rownorm_f64(eofx_ctx* ctx, const float* P, int64_t rows, int L, double* out) {
  if (!ctx || !P || !out || rows <= 0 || L <= 0) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)rows * sizeof(double) + 4096));
  return launch_rownorm(ctx, P, rows, L, L, out);
}
extern "C" int eofx_mat_feature_norms_f64(eofx_ctx* ctx, const eofx_mat* m, double* out) {
  if (!ctx || !m || !out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)m->p * sizeof(double) + 4096));
  CHK(ensure_Xt(ctx, m));
  return launch_rownorm(ctx, m->Xt, m->p, m->n, m->n_pad, out);   // rows of X^T = features
}

// Euclidean norm of every sample (row) of the resident matrix, out[n] float64 host|device: without building a layout
// for an in-place matrix (the raw field goes through the Scaler map on the fly)
extern "C" int eofx_mat_sample_norms_f64(eofx_ctx* ctx, const eofx_mat* m, double* out) {
  if (!ctx || !m || !out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  CHK(arena_reserve(ctx, (size_t)m->n * sizeof(double) + 4096));
  if (!m->X && m->raw && m->aff) {
    ArenaScope scope(ctx);
    ARENA(double, tmp, m->n);
    hipLaunchKernelGGL(rownorm_aff_kernel, dim3((int)((m->n + 3) / 4)), dim3(256), 0, ctx->stream, m->raw, m->n, m->p, m->raw_ld,
                       m->aff, m->p_pad, tmp);
    KCHK();
    HIPCHK(hipMemcpyAsync(out, tmp, sizeof(double) * m->n, hipMemcpyDefault, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return EOFX_OK;
  }
  CHK(ensure_X(ctx, m));
  return launch_rownorm(ctx, m->X, m->n, m->p, m->p_pad, out);
}

extern "C" int eofx_panel_row_normalize_f32(eofx_ctx* ctx, const float* P, int64_t rows_pad, int L, float* out) {
  if (!ctx || !P || !out) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  hipLaunchKernelGGL(row_normalize_kernel, dim3((int)((rows_pad + 3) / 4)), dim3(256), 0, ctx->stream, P, rows_pad, L,
                     2.220446049250313e-16, out);
  KCHK();
  return EOFX_OK;
}

// max |column| over the rows of a complex [Re | Im] panel (L = 2 h columns, h a power of two <= 128) -> out[h] (device)
extern "C" int eofx_cpanel_colabsmax_f32(eofx_ctx* ctx, const float* P, int64_t rows, int L, float* out) {
  if (!ctx || !P || !out || L < 2 || L > 256 || (L & (L - 1))) return set_err(ctx, EOFX_ERR_ARG, "bad argument");
  ENTER(ctx);
  HIPCHK(hipMemsetAsync(out, 0, sizeof(float) * (L / 2), ctx->stream));
  const int rstep = 256 / (L / 2);
  hipLaunchKernelGGL(cpanel_colabsmax_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>((rows + rstep - 1) / rstep, 2048))),
                     dim3(256), 0, ctx->stream, P, rows, L, reinterpret_cast<unsigned*>(out));
  KCHK();
  return EOFX_OK;
}

template <int LW>
static int launch_rot_step_wide(eofx_ctx* ctx, const float* X, int64_t rows_pad, const double* R, const double* aux, int mode,
                                double power, double* G) {
  constexpr size_t lds = sizeof(float) * 32 * (LW + 4) + sizeof(double) * 32 * (LW + 2) + sizeof(double) * 32 * 66;
  auto kern = rot_step_wide_kernel<LW>;
  HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // one workgroup (two waves per SIMD) per CU: 256 / 512 workgroups in all
  const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((rows_pad + 31) / 32, LW == 256 ? 64 : 256));
  CHK(arena_reserve(ctx, (size_t)(nbx + 1) * LW * LW * sizeof(double) + 4096));
  ArenaScope scope(ctx);
  ARENA(double, part, (size_t)nbx * LW * LW);
  hipLaunchKernelGGL(kern, dim3(nbx, LW / 64), dim3(512), lds, ctx->stream, X, rows_pad, R, aux, mode, power, part);
  KCHK();
  const int64_t count = (int64_t)LW * LW;
  hipLaunchKernelGGL(f64_reduce_kernel, dim3((int)((count + 63) / 64)), dim3(256), 0, ctx->stream, part, G, count, nbx);
  KCHK();
  return EOFX_OK;
}

extern "C" int eofx_panel_rot_step_f64(eofx_ctx* ctx, const float* X, int64_t rows_pad, int L, const double* R,
                                       const double* aux, int mode, double power, double* G) {
  if (!ctx || !X || !R || !aux || !G || (L != 32 && L != 64 && L != 128 && L != 256) || mode < 0 || mode > 3 ||
      (mode >= 2 && L < 64))
    return set_err(ctx, EOFX_ERR_ARG, "bad argument (L = 32, 64, 128 or 256; the complex modes 2 / 3 need a [Re | Im] panel of L >= 64)");
  ENTER(ctx);
  if (L == 128) return launch_rot_step_wide<128>(ctx, X, rows_pad, R, aux, mode, power, G);
  if (L == 256) return launch_rot_step_wide<256>(ctx, X, rows_pad, R, aux, mode, power, G);
  const int nbx = (int)std::max<int64_t>(1, std::min<int64_t>((rows_pad + 31) / 32, 512));
  CHK(arena_reserve(ctx, (size_t)(nbx + 1) * L * L * sizeof(double) + 4096));
  ArenaScope scope(ctx);
  ARENA(double, part, (size_t)nbx * L * L);
  hipLaunchKernelGGL(rot_step_kernel, dim3(nbx), dim3(256), 0, ctx->stream, X, rows_pad, L, R, aux, mode, power, part);
  KCHK();
  const int64_t count = (int64_t)L * L;
  hipLaunchKernelGGL(f64_reduce_kernel, dim3((int)((count + 63) / 64)), dim3(256), 0, ctx->stream, part, G, count, nbx);
  KCHK();
  return EOFX_OK;
}
