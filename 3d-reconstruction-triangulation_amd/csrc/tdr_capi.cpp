// tdr_capi.cpp — implementation of the C ABI declared in include/tdr_amd.h.
// Host side of the engine: context (camera set resident on one GPU), H2D/D2H staging for the
// host-buffer entry points, error mapping.  No CPU compute path exists here: every solve is a
// kernel launch, and without a gfx950 device tdr_create fails with TDR_E_NO_DEVICE.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <functional>
#include <mutex>
#include <new>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "tdr_amd.h"
#include "tdr_kernels.h"
#include "worker_pool.h"

using namespace tdr_amd;

namespace {

thread_local std::string g_last_error;

int hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu ||
      e == hipErrorInsufficientDriver || e == hipErrorSharedObjectInitFailed)
    return TDR_E_NO_DEVICE;
  return TDR_E_HIP;
}

#define TDR_HIP(call)                                   \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #call);   \
  } while (0)

// Nothing may throw across the C boundary: entry points that allocate host memory or start threads run their
// body through this (std::bad_alloc, std::system_error from std::thread, ...) -> TDR_E_HIP + tdr_last_error().
template <class F>
int guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::exception& e) {
    g_last_error = std::string("host exception: ") + e.what();
  } catch (...) {
    g_last_error = "host exception";
  }
  return TDR_E_HIP;
}

// joins every thread of a group on scope exit (after telling them to stop), so that an exception or an early
// return between two thread creations can never leave a joinable std::thread behind
struct ThreadGroup {
  std::vector<std::thread> threads;
  std::function<void()> stop;
  ~ThreadGroup() {
    if (stop) stop();
    for (auto& t : threads)
      if (t.joinable()) t.join();
  }
};

constexpr unsigned long long kNoBad = ~0ull;

struct EventPair {  // two HIP events, destroyed on every exit path
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

// RCCL is bound lazily (dlopen) so that the single-GPU path has no load-time dependency on it.
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) return a;
    auto sym = [&](const char* n) { return dlsym(a.handle, n); };
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.CommInitAll && a.CommDestroy && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
    return a;
  }();
  return api;
}
constexpr int64_t kChunkSystems = int64_t(1) << 24;
constexpr size_t kSlotBytes = 64 + sizeof(double) * 2 * kMaxCam;  // tdr_ctx::h_slot
constexpr int kPipeSlots = 4;  // device (and pinned) slots of the pipelined host-buffer path
constexpr int kBlockThreads = 256;  // kBlock of tdr_kernels.hip (fma_peak_kernel)  // host-buffer path: systems per staged chunk

}  // namespace

struct tdr_ctx {
  int device = 0;
  DeviceCams cams;
  hipStream_t stream = nullptr;
  unsigned long long* d_first_bad = nullptr;
  // [0] scratch first_bad, [1] survivor count of the scoring kernel, [2 .. 2+W) LM launch records (W = kLmCounters *
  // kLmCounterWords: the queue head of each LM launch in flight, tdr_kernels.h), [2+W] rejected-value count of the int16
  // packer, [3+W] overflow flag of the tree sweep, [4+W] number of flagged cuts of the tree sweep
  unsigned long long* d_scalars = nullptr;
  double* d_P = nullptr;
  RayCam* d_ray = nullptr;
  CfCam* d_cf = nullptr;  // constants of the closed-form ray kernel, derived from the same aux records
  // staging workspace for the host-buffer entry points (grow-only)
  char* d_ws = nullptr;
  size_t ws_bytes = 0;
  // pinned host staging block for small host-buffer batches (grow-only, <= pinned_max())
  char* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  // small workspace for the single-system entry points
  double* d_point = nullptr;  // [kMaxCam*12 | kMaxCam*2 | 4] doubles (RayCam is 11 doubles <= 12)
  // slot of the single-system entry points: host-coherent pinned memory the kernels read and write in place —
  // [x y z err | sequence word | pad to 64 B | uv of up to kMaxCam observations] — so that a call needs no copy in
  // either direction (tdr_matrix_triangulate_point, tdr_ray_triangulate_point, tdr_dist_from_ray); d_slot is its
  // device address
  double* h_slot = nullptr;
  double* d_slot = nullptr;
  unsigned long long slot_seq = 0;
  // pipelined host-buffer path (run_host_pipelined): upload / download streams next to `stream` (compute),
  // per-slot events, pinned staging slots (grow-only)
  hipStream_t s_up = nullptr, s_down = nullptr;
  std::unique_ptr<WorkerPool> pool;  // helper threads of run_host_pipelined, created on first use
  hipStream_t s_comp2 = nullptr;  // second compute stream: consecutive LM chunks alternate, so that one chunk's drain
                                  // (waves finishing their last systems) runs beside the next chunk's start
  hipEvent_t ev_up[kPipeSlots] = {}, ev_comp[kPipeSlots] = {}, ev_down[kPipeSlots] = {};
  char* h_pipe = nullptr;
  size_t h_pipe_bytes = 0;
  // device tree sweep (tdr_score_combinations_levelwise): grow-only device buffers
  struct DevBuf {
    char* p = nullptr;
    size_t cap = 0;
  };
  DevBuf sweep[14];
  int64_t sweep_result_n = -1;  // survivors of the last device sweep still held in sweep[B_OUT] (-1: none)
  int pipe_mode = -1;       // -1: TDR_HOST_PIPE / default; see tdr_set_host_pipeline
  int64_t pipe_chunk = 0;   //  0: TDR_HOST_CHUNK / default
  int matrix_variant = 0;
  std::mutex mu;
  // device == -1 at creation: one sub-context per visible GPU; host-buffer batches are block-sharded
  // over them by host threads (single-process multi-GPU for C++ callers)
  std::vector<tdr_ctx*> shards;
  std::vector<ncclComm_t> comms;  // one per shard, created on first RCCL gather (distinct devices only)
};

namespace {

int ensure_ws(tdr_ctx* ctx, size_t bytes) {
  if (ctx->ws_bytes >= bytes) return TDR_OK;
  if (ctx->d_ws) {
    TDR_HIP(hipFree(ctx->d_ws));
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
  }
  TDR_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_ws), bytes));
  ctx->ws_bytes = bytes;
  return TDR_OK;
}

// Batches whose input + output fit this many bytes go through one pinned staging block: two truly
// asynchronous copies and one synchronisation per call instead of one blocking pageable copy per camera row.
// That is the reference's own operating point (a few thousand frames per recording).  TDR_HOST_PINNED_MAX=0
// switches the route off.
size_t pinned_max() {
  static const size_t v = [] {
    const char* e = std::getenv("TDR_HOST_PINNED_MAX");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)4 << 20;
  }();
  return v;
}

int ensure_stage(tdr_ctx* ctx, size_t bytes) {
  if (ctx->h_stage_bytes >= bytes) return TDR_OK;
  if (ctx->h_stage) {
    TDR_HIP(hipHostFree(ctx->h_stage));
    ctx->h_stage = nullptr;
    ctx->h_stage_bytes = 0;
  }
  TDR_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), bytes, hipHostMallocDefault));
  ctx->h_stage_bytes = bytes;
  return TDR_OK;
}

hipStream_t pick_stream(tdr_ctx* ctx, void* stream) {
  return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
}

int read_first_bad(tdr_ctx* ctx, hipStream_t st, int64_t* first_bad) {
  unsigned long long v = kNoBad;
  TDR_HIP(hipMemcpyAsync(&v, ctx->d_first_bad, sizeof(v), hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  if (v != kNoBad) {
    unsigned long long reset = kNoBad;
    TDR_HIP(hipMemcpyAsync(ctx->d_first_bad, &reset, sizeof(reset), hipMemcpyHostToDevice, st));
    TDR_HIP(hipStreamSynchronize(st));
  }
  if (first_bad) *first_bad = (v == kNoBad) ? -1 : (int64_t)v;
  return TDR_OK;
}

// Host-buffer batches.  path 0 = matrix, 1 = ray (LM), 3 = ray closed form (2 = the int16 matrix format, device entry
// points only).
struct HostOut {  // the caller's result arrays (err / iters / status optional; the last two: ray path only)
  double* xyz = nullptr;
  double* err = nullptr;
  int32_t* iters = nullptr;
  int32_t* status = nullptr;
  HostOut at(int64_t s0) const {
    return HostOut{xyz + 3 * s0, err ? err + s0 : nullptr, iters ? iters + s0 : nullptr,
                   status ? status + s0 : nullptr};
  }
};

int run_host_batch(tdr_ctx* ctx, int path, const double* pts, int n_cam_pts, int64_t src_stride,
                   int64_t n_sys, HostOut out, int64_t* first_bad, const double* const* rows = nullptr,
                   int64_t row_offset = 0);

// block-partition [0, n_sys) over the sub-contexts; every shard works on an in-place slice of the
// caller's [cam][sys] array and writes its block of the result; lowest failing index wins
int run_host_batch_sharded(tdr_ctx* ctx, int path, const double* pts, int n_cam_pts, int64_t n_sys,
                           HostOut out, int64_t* first_bad, const double* const* rows) {
  const int64_t G = (int64_t)ctx->shards.size();
  std::vector<int> rcs((size_t)G, TDR_OK);
  std::vector<int64_t> fbs((size_t)G, -1);
  ThreadGroup group;
  auto& threads = group.threads;
  for (int64_t g = 0; g < G; ++g) {
    const int64_t s0 = n_sys * g / G, s1 = n_sys * (g + 1) / G;
    threads.emplace_back([=, &rcs, &fbs]() {
      rcs[(size_t)g] = run_host_batch(ctx->shards[(size_t)g], path, pts ? pts + 2 * s0 : nullptr, n_cam_pts,
                                      n_sys, s1 - s0, out.at(s0), &fbs[(size_t)g], rows, s0);
      if (fbs[(size_t)g] >= 0) fbs[(size_t)g] += s0;
    });
  }
  for (auto& th : threads) th.join();
  for (int64_t g = 0; g < G; ++g)  // shards are in index order: the first failing shard holds the first failing system
    if (rcs[(size_t)g] != TDR_OK) {
      if (first_bad) *first_bad = fbs[(size_t)g];
      return rcs[(size_t)g];
    }
  return TDR_OK;
}

hipError_t launch_path(tdr_ctx* ctx, int path, const double* d_pts, int n_used, int64_t cam_stride, int64_t n,
                       double* d_xyz, double* d_err, int* d_iters, int* d_status, unsigned long long* d_fb,
                       hipStream_t st) {
  if (path == 3) return launch_ray_closest_point(ctx->cams, d_pts, n_used, cam_stride, n, d_xyz, d_err, d_fb, st);
  return path == 0 ? launch_matrix(ctx->cams, ctx->matrix_variant, d_pts, n_used, cam_stride, n, d_xyz, d_err,
                                   d_fb, st)
                   : launch_ray(ctx->cams, d_pts, n_used, cam_stride, n, d_xyz, d_err, d_iters, d_status, d_fb, st);
}

// Layout of one batch (or one pipeline slot) in device memory, and — for the pinned routes — the identical
// layout in pinned host memory:  [cam rows | first_bad word (16 B) | xyz | err | iters | status]
struct SlotLayout {
  size_t pts_bytes, o_fb, o_xyz, o_err, o_iters, o_status, o_i16, total;
  // with_i16: room for the camera rows as int16 pairs behind the outputs (narrow transport of the LM path, whose
  // kernel reads doubles: the rows are unpacked on the device into the front of the slot)
  SlotLayout(int n_used, int64_t cap, const HostOut& out, bool with_i16 = false) {
    pts_bytes = (size_t)n_used * cap * 16;
    o_fb = pts_bytes;
    o_xyz = o_fb + 16;
    o_err = o_xyz + (size_t)cap * 24;
    o_iters = o_err + (out.err ? (size_t)cap * 8 : 0);
    o_status = o_iters + (out.iters ? (size_t)cap * 4 : 0);
    o_i16 = (o_status + (out.status ? (size_t)cap * 4 : 0) + 255) & ~(size_t)255;
    total = (o_i16 + (with_i16 ? (size_t)n_used * cap * 4 : 0) + 255) & ~(size_t)255;
  }
};

// ---- large batches: chunked and pipelined --------------------------------------------------------------
// The drop-in call hands over pageable host memory and wants the result in pageable host memory
// (src/MatrixTriangulator.cpp:70-100: arguments by value, result by value), and the kernel is two orders of
// magnitude faster than the link, so the call's duration is what crosses PCIe.  The batch is cut into chunks
// that run through kPipeSlots device slots on three streams:
//     upload(k+1)  ||  kernel(k)  ||  download(k-1)
// so both directions of the link work at the same time and the kernel hides behind them.  Measured on the
// MI355X box (tools/pcie_probe.cpp, tools/host_path_rate.py): a pageable copy costs ~0.15 ms of
// set-up on top of its 56 GB/s and blocks its calling thread; true DMA from pinned memory runs at 57.6 GB/s.
//   uploads    mode 1: hipMemcpyAsync per camera row straight from the caller's memory (issuing thread blocks)
//              mode 2: worker threads copy the rows into pinned slots ahead of the link, one DMA per chunk
//   downloads  always straight into the caller's arrays, issued by a helper thread (the runtime pins the fresh
//              destination pages in the kernel; copying out of a pinned slot from user space would fault
//              them in one by one)
//   TDR_HOST_PIPE   0 = serial (upload, kernel, download, synchronise per chunk: the round-1 path), 1, 2
//   TDR_HOST_CHUNK  systems per chunk          TDR_HOST_COPY_THREADS  staging workers of mode 2
int host_pipe_mode() {
  static const int v = [] {
    const char* e = std::getenv("TDR_HOST_PIPE");
    const int m = e ? std::atoi(e) : 3;
    return m < 0 || m > 3 ? 3 : m;
  }();
  return v;
}
// systems per pipeline chunk: TDR_HOST_CHUNK, else as many as make one slot's detection rows 64 MiB (2^19 at 8
// cameras — the measured optimum on C3 — 2^17 at 32 cameras, so that the four pinned slots stay at 256 MiB)
int64_t host_pipe_chunk_env() {
  static const int64_t env = [] {
    const char* e = std::getenv("TDR_HOST_CHUNK");
    const long long c = e ? std::atoll(e) : 0;
    return (int64_t)(c > 0 && c < 1024 ? 1024 : c);
  }();
  return env;
}
bool host_pipe_chunk_is_default() { return host_pipe_chunk_env() <= 0; }
int64_t host_pipe_chunk(int n_used) {
  const int64_t env = host_pipe_chunk_env();
  if (env > 0) return env;
  const int64_t c = ((int64_t)64 << 20) / ((int64_t)(n_used > 0 ? n_used : 1) * 16);
  return c < 4096 ? 4096 : c;
}
// staging / copy-out workers of the pinned route: TDR_HOST_COPY_THREADS, else 4 for plain copies (more only add
// contention: 26.2 ms with 4, 29.7 ms with 8 on C3) and 8 when they also convert to int16 (14.1 ms with 4, 11.8 ms
// with 8, 12.6 ms with 16)
int host_pipe_threads(bool narrow) {
  static const int env = [] {
    const char* e = std::getenv("TDR_HOST_COPY_THREADS");
    const int t = e ? std::atoi(e) : 0;
    return t < 0 ? 0 : (t > 64 ? 64 : t);
  }();
  return env > 0 ? env : (narrow ? 8 : 4);
}

bool host_pipe_trace() {  // TDR_HOST_PIPE_TRACE=1: per-chunk timestamps on stderr (diagnosis only)
  static const bool v = std::getenv("TDR_HOST_PIPE_TRACE") != nullptr;
  return v;
}
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- narrow transport (mode 3) ---------------------------------------------------------------------------
// The reference's detections are integer pixel coordinates (std::stoi, src/DetectionsContainer.cpp:36,57-60) held
// in doubles.  The staging workers of the pinned route already touch every observation once; in mode 3 they
// convert it to the int16 pair of the narrow observation format on the way (4 B instead of 16 B across PCIe) and
// the chunk is solved by the int16 kernel, whose results are bit-identical to the double kernel's on the same
// values (tests: test_matrix_int16_format_is_bit_identical).  A chunk holding any value that is not an integer
// in [-32768, 32767] is sent as doubles instead.  Matrix path only (the LM kernel has no narrow form).
// pack_obs_i16: n (x, y) double pairs -> n int16 pairs; false if a value does not convert exactly.
#if defined(__x86_64__)
__attribute__((target("avx2"))) bool pack_obs_i16_avx2(const double* src, int32_t* dst, size_t n) {
  const __m256d lo = _mm256_set1_pd(-32768.0), hi = _mm256_set1_pd(32767.0);
  __m256d ok = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {  // 4 observations = 8 doubles per iteration
    const __m256d a = _mm256_loadu_pd(src + 2 * i), b = _mm256_loadu_pd(src + 2 * i + 4);
    const __m128i ia = _mm256_cvttpd_epi32(a), ib = _mm256_cvttpd_epi32(b);
    ok = _mm256_and_pd(ok, _mm256_and_pd(_mm256_cmp_pd(_mm256_cvtepi32_pd(ia), a, _CMP_EQ_OQ),
                                          _mm256_cmp_pd(_mm256_cvtepi32_pd(ib), b, _CMP_EQ_OQ)));
    ok = _mm256_and_pd(ok, _mm256_and_pd(_mm256_and_pd(_mm256_cmp_pd(a, lo, _CMP_GE_OQ), _mm256_cmp_pd(a, hi, _CMP_LE_OQ)),
                                          _mm256_and_pd(_mm256_cmp_pd(b, lo, _CMP_GE_OQ), _mm256_cmp_pd(b, hi, _CMP_LE_OQ))));
    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm_packs_epi32(ia, ib));  // x0 y0 x1 y1 | x2 y2 x3 y3
  }
  bool good = _mm256_movemask_pd(ok) == 0xf;
  for (; i < n; ++i) {
    const double x = src[2 * i], y = src[2 * i + 1];
    const bool g = x >= -32768.0 && x <= 32767.0 && y >= -32768.0 && y <= 32767.0 && x == (double)(int)x && y == (double)(int)y;
    good = good && g;
    dst[i] = g ? (int32_t)(((uint32_t)(uint16_t)(int16_t)(int)x) | ((uint32_t)(uint16_t)(int16_t)(int)y << 16)) : 0;
  }
  return good;
}
#endif
bool pack_obs_i16(const double* src, int32_t* dst, size_t n) {
#if defined(__x86_64__)
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  if (have_avx2) return pack_obs_i16_avx2(src, dst, n);
  __m128d ok = _mm_castsi128_pd(_mm_set1_epi32(-1));
  const __m128d lo = _mm_set1_pd(-32768.0), hi = _mm_set1_pd(32767.0);
  for (size_t i = 0; i < n; ++i) {  // SSE2: one observation per iteration
    const __m128d a = _mm_loadu_pd(src + 2 * i);
    const __m128i ia = _mm_cvttpd_epi32(a);
    ok = _mm_and_pd(ok, _mm_and_pd(_mm_cmpeq_pd(_mm_cvtepi32_pd(ia), a), _mm_and_pd(_mm_cmpge_pd(a, lo), _mm_cmple_pd(a, hi))));
    dst[i] = _mm_cvtsi128_si32(_mm_packs_epi32(ia, ia));
  }
  return _mm_movemask_pd(ok) == 3;
#else
  bool good = true;
  for (size_t i = 0; i < n; ++i) {
    const double x = src[2 * i], y = src[2 * i + 1];
    const bool g = x >= -32768.0 && x <= 32767.0 && y >= -32768.0 && y <= 32767.0 && x == (double)(int)x && y == (double)(int)y;
    good = good && g;
    dst[i] = g ? (int32_t)(((uint32_t)(uint16_t)(int16_t)(int)x) | ((uint32_t)(uint16_t)(int16_t)(int)y << 16)) : 0;
  }
  return good;
#endif
}

// systems per pipeline chunk of this call.  LM path: the kernel, not the link, bounds the call, and its launches
// have a fixed cost (first fill, drain); 131072 systems — one resident set of waves — measured best (C5: 5.5 ms
// against 6.1 at 262144 and 6.5 at 65536)
int64_t pipe_chunk_for(const tdr_ctx* ctx, int path, int n_used) {
  if (ctx->pipe_chunk > 0) return ctx->pipe_chunk;
  const int64_t c = host_pipe_chunk(n_used);
  return (path == 1 && host_pipe_chunk_is_default() && c > 131072) ? 131072 : c;
}

int ensure_pipe(tdr_ctx* ctx, size_t pinned_bytes) {
  if (!ctx->s_up) TDR_HIP(hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
  if (!ctx->s_down) TDR_HIP(hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
  if (!ctx->s_comp2) TDR_HIP(hipStreamCreateWithFlags(&ctx->s_comp2, hipStreamNonBlocking));
  for (int i = 0; i < kPipeSlots; ++i) {
    if (!ctx->ev_up[i]) TDR_HIP(hipEventCreateWithFlags(&ctx->ev_up[i], hipEventDisableTiming));
    if (!ctx->ev_comp[i]) TDR_HIP(hipEventCreateWithFlags(&ctx->ev_comp[i], hipEventDisableTiming));
    if (!ctx->ev_down[i]) TDR_HIP(hipEventCreateWithFlags(&ctx->ev_down[i], hipEventDisableTiming));
  }
  if (ctx->h_pipe_bytes < pinned_bytes) {
    if (ctx->h_pipe) {
      TDR_HIP(hipHostFree(ctx->h_pipe));
      ctx->h_pipe = nullptr;
      ctx->h_pipe_bytes = 0;
    }
    TDR_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pipe), pinned_bytes, hipHostMallocDefault));
    ctx->h_pipe_bytes = pinned_bytes;
  }
  return TDR_OK;
}

int run_host_pipelined(tdr_ctx* ctx, int path, int n_used, const double* pts, int64_t src_stride, int64_t n_sys,
                       HostOut out, int64_t* first_bad, const double* const* rows, int64_t row_offset, int mode) {
  constexpr int S = kPipeSlots;
  const bool narrow = mode == 3;  // int16 transport where the values allow it (see pack_obs_i16)
  // the matrix path has a kernel that reads the int16 rows directly; the LM path unpacks them on the device
  const bool unpack = narrow && path != 0;
  if (mode == 3) mode = 2;                     // otherwise mode 3 is the pinned route
  int64_t chunk = pipe_chunk_for(ctx, path, n_used);
  if (narrow) chunk += chunk & 1;  // the int16 kernel loads observation pairs: even row stride
  const int64_t C = narrow ? chunk : (n_sys < chunk ? n_sys : chunk);
  // Equal chunks.  A ramp (C/8, C/4, C/2 first, the last C as C/2, C/4, C/4) was tried to shorten the time before
  // the link starts and after it stops: the first DMAs of a call take ~1 ms whatever their size and every chunk
  // costs ~0.2 ms of hand-overs, so it lost (C3 doubles 25.9 -> 29.8 ms, LM path 7.9 -> 11.1 ms).
  const int64_t K = (n_sys + C - 1) / C;
  const SlotLayout L(n_used, C, out, unpack);
  const size_t i16_at = unpack ? L.o_i16 : 0;  // where a slot keeps its int16 rows
  // mode 2: every pinned slot mirrors the device slot's layout ([cam rows | first_bad | xyz | err | iters | status])
  const size_t pin_slot = L.total;
  const size_t out_tail = L.o_status + (out.status ? (size_t)C * 4 : 0) - L.o_fb;  // bytes from first_bad on
  int rc = ensure_ws(ctx, (size_t)S * L.total);
  if (rc == TDR_OK) rc = ensure_pipe(ctx, mode == 2 ? (size_t)S * pin_slot : 0);
  if (rc != TDR_OK) return rc;
  const int device = ctx->device;
  hipStream_t s_up = ctx->s_up, s_comp = ctx->stream, s_down = ctx->s_down;
  static const bool two_compute_streams = [] {  // TDR_HOST_LM_STREAMS=1: A/B, all LM chunks on one stream
    const char* e = std::getenv("TDR_HOST_LM_STREAMS");
    return !(e && std::atoi(e) == 1);
  }();
  // TDR_HOST_DOWN_KERNEL=1 (A/B only): download through a copy kernel that stores into the mapped pinned slot
  // instead of hipMemcpyAsync — measured slower (28.2 vs 26.5 ms on C3), so off by default
  static const bool down_kernel_env = [] {
    const char* e = std::getenv("TDR_HOST_DOWN_KERNEL");
    return e && std::atoi(e) != 0;
  }();
  char* h_pipe_dev = nullptr;
  if (mode == 2 && down_kernel_env &&
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h_pipe_dev), ctx->h_pipe, 0) != hipSuccess)
    h_pipe_dev = nullptr;
  const bool down_kernel = h_pipe_dev != nullptr;
  const bool trace = host_pipe_trace();
  const double t_base = now_ms();
  auto row_src = [&](int c, int64_t s0) {
    return reinterpret_cast<const char*>(rows ? rows[c] + 2 * (row_offset + s0) : pts + 2 * ((size_t)c * src_stride + s0));
  };
  auto chunk_len = [&](int64_t k) { return (n_sys - k * C) < C ? (n_sys - k * C) : C; };

  // hand-over between the host threads (device-side ordering is carried by the events)
  std::mutex mu;
  std::condition_variable cv;
  int64_t launched = 0;     // chunks whose kernel is enqueued
  int64_t downloaded = 0;   // chunks whose results are in the caller's arrays
  int64_t stage_upto = S;   // staging workers may fill the pinned slots of chunks < stage_upto
  std::vector<int> staged((size_t)K, 0);  // per chunk: workers that have finished their share
  std::vector<char> wide((size_t)K, 0);   // narrow transport: chunk holds a value that does not fit (set by workers)
  int64_t dma_down = 0;                   // mode 2: chunks whose results have arrived in their pinned slot
  std::vector<int> copied((size_t)K, 0);  // mode 2: per chunk, workers that have copied their share out
  bool abort_all = false;
  int dl_rc = TDR_OK;
  int64_t bad_global = -1;
  auto set_abort = [&] {
    std::lock_guard<std::mutex> lk(mu);
    abort_all = true;
    cv.notify_all();
  };

  // ---- downloads: straight into the caller's arrays.  The runtime pins the destination pages in the kernel
  // (much cheaper than faulting fresh pages in from user space), and a pageable copy blocks its caller — hence
  // the separate thread.
  // every helper thread of this call; on any exit path they are told to stop and joined.  (T is declared before the
  // group: the workers capture it by reference, and on an exceptional unwind ~TaskGroup joins them first.)
  const int T = mode == 2 ? host_pipe_threads(narrow) : 0;
  if (!ctx->pool) ctx->pool.reset(new WorkerPool());
  TaskGroup group(*ctx->pool);
  group.stop = [&] {
    std::lock_guard<std::mutex> lk(mu);
    abort_all = true;
    stage_upto = K + S;
    cv.notify_all();
  };
  group.run([&] {
    auto fail = [&](int code) {
      std::lock_guard<std::mutex> lk(mu);
      dl_rc = code;
      abort_all = true;
      cv.notify_all();
    };
    if (hipSetDevice(device) != hipSuccess) return fail(TDR_E_HIP);
    for (int64_t k = 0; k < K; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return launched > k || abort_all; });
        if (abort_all) return;
      }
      const int slot = (int)(k % S);
      const int64_t s0 = k * C, n = chunk_len(k);
      char* d = ctx->d_ws + (size_t)slot * L.total;
      unsigned long long fb = kNoBad;
      const double td0 = now_ms();
      // TDR_HOST_DOWN_SYNC=1 (A/B): wait for the kernel on the host instead of inside the download stream
      static const bool down_sync = std::getenv("TDR_HOST_DOWN_SYNC") != nullptr;
      hipError_t e = down_sync ? hipEventSynchronize(ctx->ev_comp[slot]) : hipStreamWaitEvent(s_down, ctx->ev_comp[slot], 0);
      if (mode == 2) {
        // true DMA into the pinned slot (a pageable download was observed to queue behind the uploads' DMA
        // instead of running beside it); the workers below copy the results out to the caller's arrays
        char* h = ctx->h_pipe + (size_t)slot * pin_slot;
        if (down_kernel) {
          // A/B variant: the results leave through a copy KERNEL that stores straight into the pinned slot (mapped
          // into the device's address space), so that the DMA engine carries only the uploads
          if (e == hipSuccess)
            e = launch_stream(1, 8, reinterpret_cast<const double*>(d + L.o_fb),
                              reinterpret_cast<double*>(h_pipe_dev + (size_t)slot * pin_slot + L.o_fb),
                              (int64_t)((out_tail + 15) / 16), s_down);
        } else if (e == hipSuccess) {
          e = hipMemcpyAsync(h + L.o_fb, d + L.o_fb, out_tail, hipMemcpyDeviceToHost, s_down);
        }
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_down[slot], s_down);
        if (e == hipSuccess) e = hipStreamSynchronize(s_down);
        if (e != hipSuccess) return fail(hip_fail(e, "pipelined download"));
        fb = *reinterpret_cast<const unsigned long long*>(h + L.o_fb);
        if (trace) std::fprintf(stderr, "  down %lld: %.2f .. %.2f ms\n", (long long)k, td0 - t_base, now_ms() - t_base);
        std::lock_guard<std::mutex> lk(mu);
        if (fb != kNoBad) {
          bad_global = s0 + (int64_t)fb;
          abort_all = true;
          cv.notify_all();
          return;
        }
        dma_down = k + 1;
        cv.notify_all();
        continue;
      }
      if (e == hipSuccess) e = hipMemcpyAsync(out.xyz + 3 * s0, d + L.o_xyz, (size_t)n * 24, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess && out.err)
        e = hipMemcpyAsync(out.err + s0, d + L.o_err, (size_t)n * 8, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess && out.iters)
        e = hipMemcpyAsync(out.iters + s0, d + L.o_iters, (size_t)n * 4, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess && out.status)
        e = hipMemcpyAsync(out.status + s0, d + L.o_status, (size_t)n * 4, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipMemcpyAsync(&fb, d + L.o_fb, sizeof(fb), hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipEventRecord(ctx->ev_down[slot], s_down);
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      if (e != hipSuccess) return fail(hip_fail(e, "pipelined download"));
      if (trace) std::fprintf(stderr, "  down %lld: %.2f .. %.2f ms\n", (long long)k, td0 - t_base, now_ms() - t_base);
      std::lock_guard<std::mutex> lk(mu);
      if (fb != kNoBad) {  // the reference throws at the first such frame; chunks arrive in order
        bad_global = s0 + (int64_t)fb;
        abort_all = true;
        cv.notify_all();
        return;
      }
      downloaded = k + 1;
      cv.notify_all();
    }
  });

  // ---- mode 2: staging workers.  Worker t copies its 1/T share of every camera row of chunk k from the
  // caller's memory into pinned slot k % S, as far ahead as free slots allow; the uploads are then true DMA.
  for (int t = 0; t < T; ++t)
    group.run([&, t] {
      for (int64_t k = 0; k < K; ++k) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return k < stage_upto || abort_all; });
          if (abort_all) return;
        }
        char* h = ctx->h_pipe + (size_t)(k % S) * pin_slot;
        const int64_t s0 = k * C, n = chunk_len(k);
        const size_t a = (size_t)n * 16 * (size_t)t / (size_t)T & ~(size_t)63, b = t + 1 == T ? (size_t)n * 16 : ((size_t)n * 16 * (size_t)(t + 1) / (size_t)T & ~(size_t)63);
        bool fits = true;
        if (narrow) {  // rows of int16 pairs, 4 B per observation, row stride C
          for (int c = 0; c < n_used; ++c)
            fits = pack_obs_i16(reinterpret_cast<const double*>(row_src(c, s0) + a),
                                reinterpret_cast<int32_t*>(h + i16_at + (size_t)c * C * 4) + a / 16, (b - a) / 16) && fits;
        } else {
          for (int c = 0; c < n_used; ++c) std::memcpy(h + (size_t)c * C * 16 + a, row_src(c, s0) + a, b - a);
        }
        std::lock_guard<std::mutex> lk(mu);
        if (!fits) wide[(size_t)k] = 1;
        if (++staged[(size_t)k] == T) cv.notify_all();
      }
    });

  // ---- mode 2: copy-out workers.  Worker t copies its 1/T share of chunk k's results from the pinned slot into
  // the caller's arrays — freshly allocated result memory is faulted in page by page here, which is why this
  // is spread over several threads.  The last worker to finish a chunk marks it downloaded.
  for (int t = 0; t < T; ++t)
    group.run([&, t] {
      for (int64_t k = 0; k < K; ++k) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return dma_down > k || abort_all; });
          if (dma_down <= k) return;  // aborted before this chunk arrived
        }
        const char* h = ctx->h_pipe + (size_t)(k % S) * pin_slot;
        const int64_t s0 = k * C, n = chunk_len(k);
        const int64_t a = n * t / T, b = n * (t + 1) / T;
        std::memcpy(out.xyz + 3 * (s0 + a), h + L.o_xyz + (size_t)a * 24, (size_t)(b - a) * 24);
        if (out.err) std::memcpy(out.err + s0 + a, h + L.o_err + (size_t)a * 8, (size_t)(b - a) * 8);
        if (out.iters) std::memcpy(out.iters + s0 + a, h + L.o_iters + (size_t)a * 4, (size_t)(b - a) * 4);
        if (out.status) std::memcpy(out.status + s0 + a, h + L.o_status + (size_t)a * 4, (size_t)(b - a) * 4);
        std::lock_guard<std::mutex> lk(mu);
        if (++copied[(size_t)k] == T) {
          downloaded = k + 1;  // chunks complete in order: every worker walks them in order
          cv.notify_all();
        }
      }
    });

  int up_rc = TDR_OK;
  const unsigned long long no_bad = kNoBad;
  for (int64_t k = 0; k < K && up_rc == TDR_OK; ++k) {
    const int slot = (int)(k % S);
    const int64_t n = chunk_len(k), s0 = k * C;
    char* d = ctx->d_ws + (size_t)slot * L.total;
    hipError_t e = hipSuccess;
    {  // device slot free: its previous chunk's results are in the caller's arrays (which implies its kernel ran)
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return ((k < S || downloaded > k - S) && (T == 0 || staged[(size_t)k] == T)) || abort_all; });
      if (abort_all) break;
    }
    const double tu0 = now_ms();
    bool chunk_narrow = narrow;
    if (mode == 2) {
      char* h = ctx->h_pipe + (size_t)slot * pin_slot;
      *reinterpret_cast<unsigned long long*>(h + L.o_fb) = no_bad;
      if (narrow && wide[(size_t)k]) {  // rare: a value that is not an int16 integer — this chunk travels as doubles
        chunk_narrow = false;
        for (int c = 0; c < n_used; ++c) std::memcpy(h + (size_t)c * C * 16, row_src(c, s0), (size_t)n * 16);
      }
      if (chunk_narrow) {
        e = hipMemcpyAsync(d + i16_at, h + i16_at, (size_t)n_used * C * 4, hipMemcpyHostToDevice, s_up);
        if (e == hipSuccess) e = hipMemcpyAsync(d + L.o_fb, h + L.o_fb, 16, hipMemcpyHostToDevice, s_up);
      } else {
        e = hipMemcpyAsync(d, h, L.pts_bytes + 16, hipMemcpyHostToDevice, s_up);
      }
    } else {
      for (int c = 0; c < n_used && e == hipSuccess; ++c)
        e = hipMemcpyAsync(d + (size_t)c * C * 16, row_src(c, s0), (size_t)n * 16, hipMemcpyHostToDevice, s_up);
      if (e == hipSuccess) e = hipMemcpyAsync(d + L.o_fb, &no_bad, sizeof(no_bad), hipMemcpyHostToDevice, s_up);
    }
    // LM chunks alternate between two compute streams: a chunk's kernel ends in a drain (its waves finish their
    // last systems at falling lane counts, ~0.4 ms), during which the next chunk's waves can already take the
    // freed slots — on one stream the chunks' drains add up
    hipStream_t sc = (path == 1 && two_compute_streams && (k & 1)) ? ctx->s_comp2 : s_comp;
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_up[slot], s_up);
    if (e == hipSuccess) e = hipStreamWaitEvent(sc, ctx->ev_up[slot], 0);
    if (e == hipSuccess && chunk_narrow && unpack)
      e = launch_unpack_i16(d + i16_at, n_used, C, n, reinterpret_cast<double*>(d), C, sc);
    if (e == hipSuccess && chunk_narrow && !unpack)
      e = launch_matrix_i16(ctx->cams, d, n_used, C, n, reinterpret_cast<double*>(d + L.o_xyz),
                            out.err ? reinterpret_cast<double*>(d + L.o_err) : nullptr,
                            reinterpret_cast<unsigned long long*>(d + L.o_fb), sc);
    else if (e == hipSuccess)
      e = launch_path(ctx, path, reinterpret_cast<const double*>(d), n_used, C, n,
                      reinterpret_cast<double*>(d + L.o_xyz), out.err ? reinterpret_cast<double*>(d + L.o_err) : nullptr,
                      out.iters ? reinterpret_cast<int*>(d + L.o_iters) : nullptr,
                      out.status ? reinterpret_cast<int*>(d + L.o_status) : nullptr,
                      reinterpret_cast<unsigned long long*>(d + L.o_fb), sc);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_comp[slot], sc);
    if (e != hipSuccess) up_rc = hip_fail(e, "pipelined upload / launch");
    {
      std::lock_guard<std::mutex> lk(mu);
      if (up_rc != TDR_OK) abort_all = true;
      else launched = k + 1;
      cv.notify_all();
    }
    if (mode == 2 && up_rc == TDR_OK && k >= 1) {
      // the DMA out of the previous chunk's pinned slot is done (this one is queued behind it): that slot can
      // take chunk k - 1 + S, so the workers run up to S - 1 chunks ahead of the link
      e = hipEventSynchronize(ctx->ev_up[(k - 1) % S]);
      if (e != hipSuccess) {
        up_rc = hip_fail(e, "pipelined upload");
        set_abort();
      } else {
        std::lock_guard<std::mutex> lk(mu);
        stage_upto = k + S;
        cv.notify_all();
      }
    }
    if (trace) std::fprintf(stderr, "  up   %lld: %.2f .. %.2f ms\n", (long long)k, tu0 - t_base, now_ms() - t_base);
  }
  if (up_rc != TDR_OK) set_abort();
  {  // an early exit (too-few system, error) must release the workers
    std::lock_guard<std::mutex> lk(mu);
    stage_upto = K + S;
    cv.notify_all();
  }
  ctx->pool->wait(group.tickets[0]);  // the downloader: returns when the last chunk has arrived (or on abort)
  if (dl_rc != TDR_OK || bad_global >= 0) set_abort();
  group.wait_all();
  group.stop = nullptr;  // finished normally: nothing left to stop
  if (group.failed.load() && dl_rc == TDR_OK) {  // a helper thread threw: everybody was released, the call fails
    g_last_error = "an exception in a helper thread of the pipelined host path";
    dl_rc = TDR_E_HIP;
  }
  // drain whatever is still in flight before the slots can be reused
  (void)hipStreamSynchronize(s_up);
  (void)hipStreamSynchronize(s_comp);
  (void)hipStreamSynchronize(ctx->s_comp2);
  (void)hipStreamSynchronize(s_down);
  if (up_rc != TDR_OK) return up_rc;
  if (dl_rc != TDR_OK) return dl_rc;
  if (bad_global >= 0) {
    if (first_bad) *first_bad = bad_global;
    return TDR_E_TOO_FEW;
  }
  return TDR_OK;
}

int run_host_batch_impl(tdr_ctx* ctx, int path, const double* pts, int n_cam_pts, int64_t src_stride, int64_t n_sys,
                        HostOut out, int64_t* first_bad, const double* const* rows, int64_t row_offset);

// Input is either one [cam][sys] block (`pts`, camera rows `src_stride` systems apart) or `rows`: one
// pointer per camera row (the storage of each std::vector<cv::Point2d>), read from system `row_offset` on.
int run_host_batch(tdr_ctx* ctx, int path, const double* pts, int n_cam_pts, int64_t src_stride,
                   int64_t n_sys, HostOut out, int64_t* first_bad, const double* const* rows,
                   int64_t row_offset) {
  return guarded([&] {
    return run_host_batch_impl(ctx, path, pts, n_cam_pts, src_stride, n_sys, out, first_bad, rows, row_offset);
  });
}

int run_host_batch_impl(tdr_ctx* ctx, int path, const double* pts, int n_cam_pts, int64_t src_stride, int64_t n_sys,
                        HostOut out, int64_t* first_bad, const double* const* rows, int64_t row_offset) {
  if (first_bad) *first_bad = -1;
  if (!ctx || (!pts && !rows) || !out.xyz) return TDR_E_ARG;
  if (path != 1 && (out.iters || out.status)) return TDR_E_ARG;
  if (rows)
    for (int c = 0; c < n_cam_pts; ++c)
      if (!rows[c] && n_sys > 0) return TDR_E_ARG;
  if (n_cam_pts < 1 || n_sys < 0 || src_stride < n_sys) return TDR_E_SHAPE;
  if (!ctx->shards.empty()) {
    if (src_stride != n_sys) return TDR_E_ARG;
    if (path != 0 && (n_cam_pts > ctx->cams.n_cam || !ctx->cams.has_ray))
      return n_cam_pts > ctx->cams.n_cam ? TDR_E_SHAPE : TDR_E_ARG;
    if (n_sys == 0) return TDR_OK;
    return run_host_batch_sharded(ctx, path, pts, n_cam_pts, n_sys, out, first_bad, rows);
  }
  int n_used;
  if (path == 0) {
    n_used = n_cam_pts < ctx->cams.n_cam ? n_cam_pts : ctx->cams.n_cam;  // MatrixTriangulator.cpp:84
  } else {
    if (n_cam_pts > ctx->cams.n_cam) return TDR_E_SHAPE;  // UB in the reference (RayTriangulator.cpp:65,69)
    if (!ctx->cams.has_ray) return TDR_E_ARG;
    n_used = n_cam_pts;
  }
  if (n_sys == 0) return TDR_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  const SlotLayout whole(n_used, n_sys, out);
  if (whole.total <= pinned_max()) {
    // small batch (the reference's own operating point): one pinned block, ONE copy up, the kernel, ONE copy
    // down (results and the too-few word together) and one synchronisation
    const SlotLayout& L = whole;
    int rc = ensure_ws(ctx, L.total);
    if (rc == TDR_OK) rc = ensure_stage(ctx, L.total);
    if (rc != TDR_OK) return rc;
    char* h = ctx->h_stage;
    for (int c = 0; c < n_used; ++c)
      std::memcpy(h + (size_t)c * n_sys * 16,
                  rows ? rows[c] + 2 * row_offset : pts + 2 * (size_t)c * src_stride, (size_t)n_sys * 16);
    unsigned long long* h_fb = reinterpret_cast<unsigned long long*>(h + L.o_fb);
    *h_fb = kNoBad;
    hipStream_t st = ctx->stream;
    char* d = ctx->d_ws;
    TDR_HIP(hipMemcpyAsync(d, h, L.pts_bytes + 16, hipMemcpyHostToDevice, st));
    hipError_t e = launch_path(ctx, path, reinterpret_cast<const double*>(d), n_used, n_sys, n_sys,
                               reinterpret_cast<double*>(d + L.o_xyz),
                               out.err ? reinterpret_cast<double*>(d + L.o_err) : nullptr,
                               out.iters ? reinterpret_cast<int*>(d + L.o_iters) : nullptr,
                               out.status ? reinterpret_cast<int*>(d + L.o_status) : nullptr,
                               reinterpret_cast<unsigned long long*>(d + L.o_fb), st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    const size_t tail = L.o_status + (out.status ? (size_t)n_sys * 4 : 0) - L.o_fb;
    TDR_HIP(hipMemcpyAsync(h + L.o_fb, d + L.o_fb, tail, hipMemcpyDeviceToHost, st));
    TDR_HIP(hipStreamSynchronize(st));
    if (*h_fb != kNoBad) {
      if (first_bad) *first_bad = (int64_t)*h_fb;
      return TDR_E_TOO_FEW;
    }
    std::memcpy(out.xyz, h + L.o_xyz, (size_t)n_sys * 24);
    if (out.err) std::memcpy(out.err, h + L.o_err, (size_t)n_sys * 8);
    if (out.iters) std::memcpy(out.iters, h + L.o_iters, (size_t)n_sys * 4);
    if (out.status) std::memcpy(out.status, h + L.o_status, (size_t)n_sys * 4);
    return TDR_OK;
  }
  const int mode = ctx->pipe_mode >= 0 ? ctx->pipe_mode : host_pipe_mode();
  if (mode != 0 && n_sys > pipe_chunk_for(ctx, path, n_used))
    return run_host_pipelined(ctx, path, n_used, pts, src_stride, n_sys, out, first_bad, rows, row_offset, mode);
  // serial route: per chunk upload, kernel, download, synchronise
  const int64_t chunk = n_sys < kChunkSystems ? n_sys : kChunkSystems;
  const SlotLayout L(n_used, chunk, out);
  int rc = ensure_ws(ctx, L.total);
  if (rc != TDR_OK) return rc;
  char* d = ctx->d_ws;
  hipStream_t st = ctx->stream;
  for (int64_t s0 = 0; s0 < n_sys; s0 += chunk) {
    const int64_t n = (n_sys - s0) < chunk ? (n_sys - s0) : chunk;
    for (int c = 0; c < n_used; ++c)
      TDR_HIP(hipMemcpyAsync(d + (size_t)c * chunk * 16,
                             rows ? rows[c] + 2 * (row_offset + s0) : pts + 2 * ((size_t)c * src_stride + s0),
                             (size_t)n * 16, hipMemcpyHostToDevice, st));
    hipError_t e = launch_path(ctx, path, reinterpret_cast<const double*>(d), n_used, chunk, n,
                               reinterpret_cast<double*>(d + L.o_xyz),
                               out.err ? reinterpret_cast<double*>(d + L.o_err) : nullptr,
                               out.iters ? reinterpret_cast<int*>(d + L.o_iters) : nullptr,
                               out.status ? reinterpret_cast<int*>(d + L.o_status) : nullptr, ctx->d_first_bad, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    TDR_HIP(hipMemcpyAsync(out.xyz + 3 * s0, d + L.o_xyz, (size_t)n * 24, hipMemcpyDeviceToHost, st));
    if (out.err) TDR_HIP(hipMemcpyAsync(out.err + s0, d + L.o_err, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    if (out.iters) TDR_HIP(hipMemcpyAsync(out.iters + s0, d + L.o_iters, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (out.status) TDR_HIP(hipMemcpyAsync(out.status + s0, d + L.o_status, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    int64_t fb = -1;
    rc = read_first_bad(ctx, st, &fb);
    if (rc != TDR_OK) return rc;
    if (fb >= 0) {  // the reference throws at the first such frame; nothing is returned
      if (first_bad) *first_bad = s0 + fb;
      return TDR_E_TOO_FEW;
    }
  }
  return TDR_OK;
}

int run_device_batch(tdr_ctx* ctx, int path, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                     int64_t n_sys, double* d_xyz, double* d_err, void* stream, int32_t* d_iters = nullptr,
                     int32_t* d_status = nullptr) {
  if (!ctx || !d_pts || !d_xyz) return TDR_E_ARG;
  if (!ctx->shards.empty()) return TDR_E_ARG;  // device pointers belong to one GPU: use a per-device context
  if (n_cam_pts < 1 || n_sys < 0 || cam_stride < n_sys) return TDR_E_SHAPE;
  if (reinterpret_cast<uintptr_t>(d_pts) & (path == 2 ? 3 : 15)) return TDR_E_ARG;  // 16-B / 4-B (x,y) loads
  if (path != 1 && (d_iters || d_status)) return TDR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(d_iters) | reinterpret_cast<uintptr_t>(d_status)) & 3) return TDR_E_ARG;
  int n_used;
  if (path == 0 || path == 2) {
    n_used = n_cam_pts < ctx->cams.n_cam ? n_cam_pts : ctx->cams.n_cam;
  } else {
    if (n_cam_pts > ctx->cams.n_cam) return TDR_E_SHAPE;
    if (!ctx->cams.has_ray) return TDR_E_ARG;
    n_used = n_cam_pts;
  }
  TDR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = pick_stream(ctx, stream);
  hipError_t e = path == 0   ? launch_matrix(ctx->cams, ctx->matrix_variant, d_pts, n_used, cam_stride, n_sys,
                                             d_xyz, d_err, ctx->d_first_bad, st)
                 : path == 2 ? launch_matrix_i16(ctx->cams, d_pts, n_used, cam_stride, n_sys, d_xyz, d_err,
                                                 ctx->d_first_bad, st)
                 : path == 3 ? launch_ray_closest_point(ctx->cams, d_pts, n_used, cam_stride, n_sys, d_xyz, d_err,
                                                        ctx->d_first_bad, st)
                             : launch_ray(ctx->cams, d_pts, n_used, cam_stride, n_sys, d_xyz, d_err, d_iters,
                                          d_status, ctx->d_first_bad, st);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return TDR_OK;
}

// Wait for a single-system kernel that publishes `seq` behind its result in the context's pinned slot: the host polls
// the sequence word (the kernel is a few microseconds long; hipStreamSynchronize alone costs as much again in its own
// bookkeeping) and falls back to the stream synchronisation if the word has not arrived after ~50 us — a busy stream,
// a preempted process — so the wait never spins for long.  TDR_POINT_SPIN=0: always synchronise (A/B).
int wait_slot(tdr_ctx* ctx, unsigned long long seq) {
  static const bool spin = [] {
    const char* e = std::getenv("TDR_POINT_SPIN");
    return !(e && std::atoi(e) == 0);
  }();
  volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(ctx->h_slot + 4);
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
      if (*word == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return TDR_OK;
      }
      if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
    }
  }
  TDR_HIP(hipStreamSynchronize(ctx->stream));
  if (*word != seq) {
    g_last_error = "single-system kernel finished without publishing its result";
    return TDR_E_HIP;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return TDR_OK;
}

// One system through the [constants kMaxCam*12 | uv kMaxCam*2 | out 4] workspace: the inputs are assembled in the
// pinned staging block and go up in ONE copy, the result comes back in one (pageable copies are blocking, and
// a call had three of them).  `fill(h)` writes n_obs constant records at h and the detections at h + kMaxCam*12.
// result_in_slot: the kernel writes its four doubles into the context's pinned slot (no copy back); else into the
// device workspace, copied back (round 4's route, knob TDR_POINT_ARGS=0).
template <typename Fill, typename Launch>
int run_single(tdr_ctx* ctx, Fill fill, Launch launch, double out_xyz[3], double* out_err, bool result_in_slot) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  constexpr size_t kIn = sizeof(double) * (kMaxCam * 12 + kMaxCam * 2), kAll = kIn + 4 * sizeof(double);
  int rc = ensure_stage(ctx, kAll);
  if (rc != TDR_OK) return rc;
  double* h = reinterpret_cast<double*>(ctx->h_stage);
  fill(h);
  double* d_c = ctx->d_point;
  double* d_uv = d_c + kMaxCam * 12;
  double* d_out = result_in_slot ? ctx->d_slot : d_uv + kMaxCam * 2;
  hipStream_t st = ctx->stream;
  TDR_HIP(hipMemcpyAsync(d_c, h, kIn, hipMemcpyHostToDevice, st));
  hipError_t e = launch(d_c, d_uv, d_out, st);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  double* h_out = result_in_slot ? ctx->h_slot : h + kMaxCam * 14;
  if (!result_in_slot) TDR_HIP(hipMemcpyAsync(h_out, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  out_xyz[0] = h_out[0];
  out_xyz[1] = h_out[1];
  out_xyz[2] = h_out[2];
  if (out_err) *out_err = h_out[3];
  return TDR_OK;
}

}  // namespace

extern "C" {

const char* tdr_strerror(int status) {
  switch (status) {
    case TDR_OK: return "ok";
    case TDR_E_SHAPE: return "Every camera should have the same number of points";
    case TDR_E_TOO_FEW: return "Too few rays are found";
    case TDR_E_CAMERA: return "Set width, height, fx, fovx, fovy, cx and cy first.";
    case TDR_E_HIP: return "HIP runtime error";
    case TDR_E_NO_DEVICE: return "no usable gfx950 HIP device (there is no CPU fallback)";
    case TDR_E_RCCL: return "RCCL error";
    case TDR_E_ARG: return "invalid argument";
    default: return "unknown tdr status";
  }
}

const char* tdr_last_error(void) { return g_last_error.c_str(); }

/* ---------------------------------------------------------------------------- cameras */
int tdr_camera_derive(int id, int width, int height, double focal_length, const double tvec[3],
                      const double rquat[4], tdr_camera* out) {
  if (!tvec || !rquat || !out) return TDR_E_ARG;
  constexpr double kRadToDeg = 57.29577951308232087679;  // src/Camera.h:23-24
  constexpr double kDegToRad = 0.01745329251994329576;   // src/Camera.h:25-26
  tdr_camera cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.id = id;
  cam.width = width;
  cam.height = height;
  for (int i = 0; i < 3; ++i) cam.tvec[i] = tvec[i];
  for (int i = 0; i < 4; ++i) cam.rquat[i] = rquat[i];

  {  // Camera::toRotVec (src/Camera.h:258-269) as called at src/utils.cpp:102; x,y,z,w = q[0..3]
    const double nq = std::sqrt(rquat[0] * rquat[0] + rquat[1] * rquat[1] + rquat[2] * rquat[2] +
                                rquat[3] * rquat[3]);
    const double angle = 2 * std::acos(rquat[3] / nq);
    const double tmp = nq * std::sin(angle * 0.5);
    for (int i = 0; i < 3; ++i) cam.rvec[i] = angle * (rquat[i] / tmp);
  }

  // compCxCy (src/Camera.h:78-83)
  if (width == 0 || height == 0) return TDR_E_CAMERA;
  cam.cx = (int)std::round(width / 2.0);
  cam.cy = (int)std::round(height / 2.0);
  // compFovx (src/Camera.h:89-92): note the truncated 57.2958
  if (focal_length == 0) return TDR_E_CAMERA;
  cam.fovx = 2 * std::atan(width / (2 * focal_length)) * 57.2958;
  // compFovy (src/Camera.h:98-104)
  if (cam.fovx == 0) return TDR_E_CAMERA;
  cam.fovy = 2.0 *
             std::atan(std::tan(cam.fovx * 0.5 * kDegToRad) / (double(width) / double(height))) *
             kRadToDeg;
  // compFxFy (src/Camera.h:111-116): focal lengths are re-derived from the fovs
  if (cam.fovy == 0) return TDR_E_CAMERA;
  cam.fx = (width / 2.0) / (std::tan((cam.fovx / 2.0) * kDegToRad));
  cam.fy = (height / 2.0) / (std::tan((cam.fovy / 2.0) * kDegToRad));

  // toRotMatrix (src/Camera.h:274-287): w-first formula on the quaternion as stored
  const double a = rquat[0], b = rquat[1], c = rquat[2], d = rquat[3];
  const double R[3][3] = {
      {1 - 2 * (c * c + d * d), 2 * (b * c - a * d), 2 * (b * d + a * c)},
      {2 * (b * c + a * d), 1 - 2 * (b * b + d * d), 2 * (c * d - a * b)},
      {2 * (b * d - a * c), 2 * (c * d + a * b), 1 - 2 * (b * b + c * c)}};
  // .inv() (src/Camera.h:134,168): cofactor expansion / determinant, zeros when singular
  double Ri[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  {
    const double co00 = R[1][1] * R[2][2] - R[1][2] * R[2][1];
    const double co01 = R[1][0] * R[2][2] - R[1][2] * R[2][0];
    const double co02 = R[1][0] * R[2][1] - R[1][1] * R[2][0];
    double det = R[0][0] * co00 - R[0][1] * co01 + R[0][2] * co02;
    if (det != 0.0) {
      det = 1.0 / det;
      Ri[0][0] = co00 * det;
      Ri[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) * det;
      Ri[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) * det;
      Ri[1][0] = (R[1][2] * R[2][0] - R[1][0] * R[2][2]) * det;
      Ri[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) * det;
      Ri[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) * det;
      Ri[2][0] = co02 * det;
      Ri[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) * det;
      Ri[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) * det;
    }
  }
  // compCamPos (src/Camera.h:167-170): camPos = -R^-1 * tvec
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (-Ri[i][k]) * tvec[k];
    cam.campos[i] = s;
  }
  // createCamMat (src/Camera.h:122-127)
  if (cam.fx == 0 || cam.fy == 0 || cam.cx == 0 || cam.cy == 0) return TDR_E_CAMERA;
  const double K[3][3] = {{cam.fx, 0, (double)cam.cx}, {0, cam.fy, (double)cam.cy}, {0, 0, 1}};
  // createExtricsicMat (src/Camera.h:133-153) and createPerspectiveMat (:159-161)
  double E[3][4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) E[i][j] = Ri[i][j];
    E[i][3] = cam.campos[i];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) cam.K[i * 3 + j] = K[i][j];
    for (int j = 0; j < 4; ++j) {
      cam.E[i * 4 + j] = E[i][j];
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[i][k] * E[k][j];
      cam.P[i * 4 + j] = s;
    }
  }
  cam.diagonal = (int)std::sqrt((double)(width * width + height * height));  // src/Camera.h:186
  *out = cam;
  return TDR_OK;
}

int tdr_camera_pack(const tdr_camera* cams, int n_cam, double* P_out, double* aux_out) {
  if (!cams || n_cam < 1) return TDR_E_ARG;
  for (int c = 0; c < n_cam; ++c) {
    if (P_out) std::memcpy(P_out + 12 * c, cams[c].P, 12 * sizeof(double));
    if (aux_out) {
      double* a = aux_out + TDR_CAM_AUX_DOUBLES * c;
      std::memcpy(a, cams[c].rquat, 4 * sizeof(double));
      std::memcpy(a + 4, cams[c].tvec, 3 * sizeof(double));
      a[7] = cams[c].fovy;
      a[8] = (double)cams[c].width;
      a[9] = (double)cams[c].height;
    }
  }
  return TDR_OK;
}

/* ---------------------------------------------------------------------------- context */
int tdr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tdr_create(const double* P, const double* cam_aux, int n_cam, int device, tdr_ctx** out) {
  if (!out) return TDR_E_ARG;
  *out = nullptr;
  if (!P || n_cam < 1 || n_cam > TDR_MAX_CAMERAS) return TDR_E_ARG;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev < 1) {
    g_last_error = "hipGetDeviceCount: no HIP device";
    return TDR_E_NO_DEVICE;
  }
  if (device == -1) return tdr_create_multi(P, cam_aux, n_cam, nullptr, n_dev, out);
  if (device < 0 || device >= n_dev) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TDR_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("device arch ") + prop.gcnArchName + " is not gfx950";
    return TDR_E_NO_DEVICE;
  }
  tdr_ctx* ctx = new (std::nothrow) tdr_ctx;
  if (!ctx) return TDR_E_ARG;
  ctx->device = device;
  std::memset(&ctx->cams.P, 0, sizeof(ctx->cams.P));
  std::memset(&ctx->cams.ray, 0, sizeof(ctx->cams.ray));
  ctx->cams.n_cam = n_cam;
  ctx->cams.has_ray = cam_aux != nullptr;
  for (int c = 0; c < n_cam; ++c) {
    std::memcpy(ctx->cams.P.p[c], P + 12 * c, 12 * sizeof(double));
    if (cam_aux) ctx->cams.ray.c[c] = make_raycam(cam_aux + TDR_CAM_AUX_DOUBLES * c);
  }
  int rc = TDR_OK;
  auto fail = [&](hipError_t err, const char* what) {
    rc = hip_fail(err, what);
    tdr_destroy(ctx);
    return rc;
  };
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return fail(e, "hipStreamCreate");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_first_bad), sizeof(unsigned long long))) !=
      hipSuccess)
    return fail(e, "hipMalloc");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_P), sizeof(CamSetP))) != hipSuccess)
    return fail(e, "hipMalloc");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_ray), sizeof(CamSetRay))) != hipSuccess)
    return fail(e, "hipMalloc");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_cf), sizeof(CfCam) * kMaxCam)) != hipSuccess)
    return fail(e, "hipMalloc");
  {
    CfCam cf[kMaxCam];
    std::memset(cf, 0, sizeof(cf));
    for (int c = 0; c < kMaxCam; ++c) cf[c].cz = 1.0;  // unused records: e = (0, 0, 1), D = 0 (ray_closest_point_kernel)
    if (cam_aux)
      for (int c = 0; c < n_cam; ++c) cf[c] = make_cfcam(ctx->cams.ray.c[c]);
    if ((e = hipMemcpy(ctx->d_cf, cf, sizeof(cf), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
  }
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_slot), kSlotBytes, hipHostMallocMapped)) != hipSuccess)
    return fail(e, "hipHostMalloc");
  std::memset(ctx->h_slot, 0, kSlotBytes);
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_slot), ctx->h_slot, 0)) != hipSuccess)
    return fail(e, "hipHostGetDevicePointer");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_point),
                     sizeof(double) * (kMaxCam * 12 + kMaxCam * 2 + 4))) != hipSuccess)
    return fail(e, "hipMalloc");
  unsigned long long nb = kNoBad;
  if ((e = hipMemcpy(ctx->d_first_bad, &nb, sizeof(nb), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy");
  if ((e = hipMemcpy(ctx->d_P, &ctx->cams.P, sizeof(CamSetP), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy");
  if ((e = hipMemcpy(ctx->d_ray, &ctx->cams.ray, sizeof(CamSetRay), hipMemcpyHostToDevice)) !=
      hipSuccess)
    return fail(e, "hipMemcpy");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_scalars), (5 + kLmCounters * kLmCounterWords) * sizeof(unsigned long long))) !=
      hipSuccess)
    return fail(e, "hipMalloc");
  if ((e = hipMemset(ctx->d_scalars, 0xff, (5 + kLmCounters * kLmCounterWords) * sizeof(unsigned long long))) != hipSuccess)
    return fail(e, "hipMemset");
  ctx->cams.d_P = ctx->d_P;
  ctx->cams.d_ray = ctx->d_ray;
  ctx->cams.d_cf = ctx->d_cf;
  ctx->cams.d_first_bad_scratch = ctx->d_scalars;
  ctx->cams.d_work_counter = ctx->d_scalars + 2;
  *out = ctx;
  return TDR_OK;
}

int tdr_create_multi(const double* P, const double* cam_aux, int n_cam, const int* devices, int n_shards,
                     tdr_ctx** out) {
  if (!out) return TDR_E_ARG;
  *out = nullptr;
  if (!P || n_cam < 1 || n_cam > TDR_MAX_CAMERAS || n_shards < 1 || n_shards > 64) return TDR_E_ARG;
  tdr_ctx* ctx = new (std::nothrow) tdr_ctx;
  if (!ctx) return TDR_E_ARG;
  std::memset(&ctx->cams.P, 0, sizeof(ctx->cams.P));
  std::memset(&ctx->cams.ray, 0, sizeof(ctx->cams.ray));
  ctx->cams.n_cam = n_cam;
  ctx->cams.has_ray = cam_aux != nullptr;
  ctx->cams.d_P = nullptr;
  ctx->cams.d_ray = nullptr;
  ctx->cams.d_cf = nullptr;
  ctx->cams.d_work_counter = ctx->cams.d_first_bad_scratch = nullptr;
  ctx->device = -1;
  for (int g = 0; g < n_shards; ++g) {
    tdr_ctx* sub = nullptr;
    int rc = tdr_create(P, cam_aux, n_cam, devices ? devices[g] : g, &sub);
    if (rc != TDR_OK) {
      tdr_destroy(ctx);
      return rc;
    }
    ctx->shards.push_back(sub);
  }
  *out = ctx;
  return TDR_OK;
}

void tdr_destroy(tdr_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->shards.empty() || ctx->device == -1) {
    for (ncclComm_t cm : ctx->comms)
      if (cm && rccl().ok) (void)rccl().CommDestroy(cm);
    for (tdr_ctx* s : ctx->shards) tdr_destroy(s);
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  if (ctx->d_first_bad) (void)hipFree(ctx->d_first_bad);
  if (ctx->d_P) (void)hipFree(ctx->d_P);
  if (ctx->d_ray) (void)hipFree(ctx->d_ray);
  if (ctx->d_cf) (void)hipFree(ctx->d_cf);
  if (ctx->d_point) (void)hipFree(ctx->d_point);
  if (ctx->h_slot) (void)hipHostFree(ctx->h_slot);
  if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
  if (ctx->d_ws) (void)hipFree(ctx->d_ws);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_pipe) (void)hipHostFree(ctx->h_pipe);
  for (auto& b : ctx->sweep)
    if (b.p) (void)hipFree(b.p);
  for (int i = 0; i < kPipeSlots; ++i) {
    if (ctx->ev_up[i]) (void)hipEventDestroy(ctx->ev_up[i]);
    if (ctx->ev_comp[i]) (void)hipEventDestroy(ctx->ev_comp[i]);
    if (ctx->ev_down[i]) (void)hipEventDestroy(ctx->ev_down[i]);
  }
  if (ctx->s_up) (void)hipStreamDestroy(ctx->s_up);
  if (ctx->s_down) (void)hipStreamDestroy(ctx->s_down);
  if (ctx->s_comp2) (void)hipStreamDestroy(ctx->s_comp2);
  delete ctx;
}

int tdr_n_cameras(const tdr_ctx* ctx) { return ctx ? ctx->cams.n_cam : 0; }

int tdr_set_matrix_variant(tdr_ctx* ctx, int variant) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || variant < 0 || variant > 7) return TDR_E_ARG;  // 5-7: A/B-only store policies (8 cameras)
  ctx->matrix_variant = variant;
  return TDR_OK;
}

int tdr_set_ray_jacobian(tdr_ctx* ctx, int literal) {
  if (!ctx || (literal != 0 && literal != 1)) return TDR_E_ARG;
  for (tdr_ctx* sh : ctx->shards) sh->cams.lm_literal.store(literal, std::memory_order_relaxed);
  ctx->cams.lm_literal.store(literal, std::memory_order_relaxed);
  return TDR_OK;
}

int tdr_set_host_pipeline(tdr_ctx* ctx, int mode, int64_t chunk_systems) {
  if (!ctx || mode < -1 || mode > 3 || chunk_systems < 0) return TDR_E_ARG;
  if (chunk_systems > 0 && chunk_systems < 64) return TDR_E_ARG;
  for (tdr_ctx* sh : ctx->shards) {
    sh->pipe_mode = mode;
    sh->pipe_chunk = chunk_systems;
  }
  ctx->pipe_mode = mode;
  ctx->pipe_chunk = chunk_systems;
  return TDR_OK;
}

/* ---------------------------------------------------------------------------- batch, host */
int tdr_matrix_triangulate_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys,
                                  double* out_xyz, double* out_err, int64_t* first_bad) {
  return run_host_batch(ctx, 0, pts, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad);
}

int tdr_ray_triangulate_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys,
                               double* out_xyz, double* out_err, int64_t* first_bad) {
  return run_host_batch(ctx, 1, pts, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad);
}

int tdr_ray_triangulate_points_ex(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys, double* out_xyz,
                                  double* out_err, int32_t* out_iters, int32_t* out_status, int64_t* first_bad) {
  return run_host_batch(ctx, 1, pts, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err, out_iters, out_status},
                        first_bad);
}

int tdr_matrix_triangulate_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts,
                                       int64_t n_sys, double* out_xyz, double* out_err,
                                       int64_t* first_bad) {
  return run_host_batch(ctx, 0, nullptr, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad, rows, 0);
}

int tdr_ray_triangulate_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts, int64_t n_sys,
                                    double* out_xyz, double* out_err, int64_t* first_bad) {
  return run_host_batch(ctx, 1, nullptr, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad, rows, 0);
}

/* ---------------------------------------------------------------------------- single system */
int tdr_matrix_triangulate_point(tdr_ctx* ctx, const double* P_obs, const double* uv, int n_obs,
                                 double out_xyz[3], double* out_err) {
  if (!ctx || !P_obs || !uv || !out_xyz) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];  // single systems run on the first GPU
  if (n_obs < 1 || n_obs > kMaxCam) return TDR_E_SHAPE;
  for (int i = 0; i < n_obs; ++i)
    if (uv[2 * i] == -1 || uv[2 * i + 1] == -1) return TDR_E_SHAPE;
  if (knob(K_POINT_ARGS, 1, 0, 1)) {
    // constants and pixels travel in the kernel argument, the result comes back through the pinned slot: one launch,
    // one wait, no copy (csrc/tdr_kernels.hip: matrix_point_arg_kernel)
    return guarded([&] {
      std::lock_guard<std::mutex> lock(ctx->mu);
      TDR_HIP(hipSetDevice(ctx->device));
      PointArgsHost args;
      std::memcpy(args.P, P_obs, sizeof(double) * 12 * n_obs);
      std::memcpy(args.uv, uv, sizeof(double) * 2 * n_obs);
      const unsigned long long seq = ++ctx->slot_seq;
      hipError_t e = launch_matrix_point_args(args, n_obs, ctx->d_slot,
                                              reinterpret_cast<unsigned long long*>(ctx->d_slot + 4), seq, ctx->stream);
      if (e != hipSuccess) return hip_fail(e, "kernel launch");
      int rc = wait_slot(ctx, seq);
      if (rc != TDR_OK) return rc;
      out_xyz[0] = ctx->h_slot[0];
      out_xyz[1] = ctx->h_slot[1];
      out_xyz[2] = ctx->h_slot[2];
      if (out_err) *out_err = ctx->h_slot[3];
      return (int)TDR_OK;
    });
  }
  return run_single(
      ctx,
      [&](double* h) {
        std::memcpy(h, P_obs, sizeof(double) * 12 * n_obs);
        std::memcpy(h + kMaxCam * 12, uv, sizeof(double) * 2 * n_obs);
      },
      [&](double* d_P, double* d_uv, double* d_out, hipStream_t st) {
        return launch_matrix_point(d_P, d_uv, n_obs, d_out, st);
      },
      out_xyz, out_err, false);
}

int tdr_ray_triangulate_point(tdr_ctx* ctx, const double* aux_obs, const double* uv, int n_obs,
                              double out_xyz[3], double* out_err) {
  if (!ctx || !aux_obs || !uv || !out_xyz) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];  // single systems run on the first GPU
  if (n_obs < 1 || n_obs > kMaxCam) return TDR_E_SHAPE;
  for (int i = 0; i < n_obs; ++i)
    if (uv[2 * i] == -1 || uv[2 * i + 1] == -1) return TDR_E_SHAPE;
  static_assert(sizeof(RayCam) <= 12 * sizeof(double), "RayCam must fit the point workspace");
  if (n_obs <= 16 && knob(K_POINT_ARGS, 1, 0, 1)) {
    // camera-per-lane kernel, constants in its argument, pixels and result in the pinned slot: no copy either way
    return guarded([&] {
      std::lock_guard<std::mutex> lock(ctx->mu);
      TDR_HIP(hipSetDevice(ctx->device));
      RayCam rc[16];
      for (int i = 0; i < n_obs; ++i) rc[i] = make_raycam(aux_obs + TDR_CAM_AUX_DOUBLES * i);
      std::memcpy(ctx->h_slot + 8, uv, sizeof(double) * 2 * n_obs);
      hipError_t e = launch_ray_point_coop(ctx->cams, rc, ctx->d_slot + 8, n_obs, ctx->d_slot, ctx->stream);
      if (e != hipSuccess) return hip_fail(e, "kernel launch");
      TDR_HIP(hipStreamSynchronize(ctx->stream));
      out_xyz[0] = ctx->h_slot[0];
      out_xyz[1] = ctx->h_slot[1];
      out_xyz[2] = ctx->h_slot[2];
      if (out_err) *out_err = ctx->h_slot[3];
      return (int)TDR_OK;
    });
  }
  return run_single(
      ctx,
      [&](double* h) {
        for (int i = 0; i < n_obs; ++i) {
          const RayCam rc = make_raycam(aux_obs + TDR_CAM_AUX_DOUBLES * i);
          std::memcpy(reinterpret_cast<char*>(h) + sizeof(RayCam) * i, &rc, sizeof(RayCam));
        }
        std::memcpy(h + kMaxCam * 12, uv, sizeof(double) * 2 * n_obs);
      },
      [&](double* d_rc, double* d_uv, double* d_out, hipStream_t st) {
        return launch_ray_point(ctx->cams, d_rc, d_uv, n_obs, d_out, st);
      },
      // the LM kernel writes point and error into the pinned slot itself: no copy back (its inputs still go up in one
      // copy — 30 LM passes of a lone wave, not the copy, are this call's time)
      out_xyz, out_err, knob(K_POINT_ARGS, 1, 0, 1) != 0);
}

/* ---------------------------------------------------------------------------- batch, device */
int tdr_matrix_triangulate_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts,
                                         int64_t cam_stride, int64_t n_sys, double* d_xyz,
                                         double* d_err, void* stream) {
  return run_device_batch(ctx, 0, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, stream);
}

int tdr_ray_triangulate_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts,
                                      int64_t cam_stride, int64_t n_sys, double* d_xyz,
                                      double* d_err, void* stream) {
  return run_device_batch(ctx, 1, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, stream);
}

int tdr_ray_triangulate_points_device_ex(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                         int64_t n_sys, double* d_xyz, double* d_err, int32_t* d_iters,
                                         int32_t* d_status, void* stream) {
  return run_device_batch(ctx, 1, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, stream, d_iters, d_status);
}

/* ---------------------------------------------------------------------------- ray path, closed form */
int tdr_ray_closest_point_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys, double* out_xyz,
                                 double* out_err, int64_t* first_bad) {
  return run_host_batch(ctx, 3, pts, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad);
}

int tdr_ray_closest_point_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                        int64_t n_sys, double* d_xyz, double* d_err, void* stream) {
  return run_device_batch(ctx, 3, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, stream);
}

int tdr_ray_closest_point_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts, int64_t n_sys,
                                      double* out_xyz, double* out_err, int64_t* first_bad) {
  return run_host_batch(ctx, 3, nullptr, n_cam_pts, n_sys, n_sys, HostOut{out_xyz, out_err}, first_bad, rows, 0);
}

/* ---------------------------------------------------------------------------- protected static helpers (host side) */
int tdr_create_ray(const double* cam_aux, const double uv[2], double origin[3], double dir[3]) {
  if (!cam_aux || !uv || !origin || !dir) return TDR_E_ARG;
  const Ray r = make_ray(make_raycam(cam_aux), uv[0], uv[1]);
  origin[0] = r.ox; origin[1] = r.oy; origin[2] = r.oz;
  dir[0] = r.dx; dir[1] = r.dy; dir[2] = r.dz;
  return TDR_OK;
}

int tdr_ray_direction_for_pixel(const double* cam_aux, const double uv[2], double dir_cam[3]) {
  if (!cam_aux || !uv || !dir_cam) return TDR_E_ARG;
  pixel_direction(make_raycam(cam_aux), uv[0], uv[1], dir_cam[0], dir_cam[1], dir_cam[2]);
  return TDR_OK;
}

int tdr_rotate_point_by_quaternion(const double quat[4], const double point[3], double out[3]) {
  if (!quat || !point || !out) return TDR_E_ARG;
  double rx, ry, rz;  // (out may alias point)
  rotate_by_quaternion(quat[0], quat[1], quat[2], quat[3], point[0], point[1], point[2], rx, ry, rz);
  out[0] = rx; out[1] = ry; out[2] = rz;
  return TDR_OK;
}

double tdr_dist_to_ray(const double origin[3], const double dir[3], const double point[3], int squared) {
  if (!origin || !dir || !point) return std::nan("");
  const Ray r{origin[0], origin[1], origin[2], dir[0], dir[1], dir[2]};
  const double d2 = dist2_to_ray(r, point[0], point[1], point[2]);
  return squared ? d2 : std::sqrt(d2);
}

/* ---------------------------------------------------------------------------- int16 observation format */
int tdr_matrix_triangulate_points_i16_device(tdr_ctx* ctx, const int16_t* d_pts16, int n_cam_pts,
                                             int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                                             void* stream) {
  return run_device_batch(ctx, 2, reinterpret_cast<const double*>(d_pts16), n_cam_pts, cam_stride, n_sys, d_xyz,
                          d_err, stream);
}

int tdr_pack_detections_i16_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                   int64_t n_sys, int16_t* d_out, int64_t out_stride, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_pts || !d_out) return TDR_E_ARG;
  if (n_cam_pts < 1 || n_sys < 0 || cam_stride < n_sys || out_stride < n_sys) return TDR_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(d_pts) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 3)) return TDR_E_ARG;
  if (n_sys == 0) return TDR_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = pick_stream(ctx, stream);
  unsigned long long* d_bad = ctx->d_scalars + 2 + kLmCounters * kLmCounterWords;
  TDR_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), st));
  hipError_t e = launch_pack_i16(d_pts, n_cam_pts, cam_stride, n_sys, d_out, out_stride, d_bad, st);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  unsigned long long bad = 0;
  TDR_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  if (bad != 0) {
    g_last_error = std::to_string(bad) + " observation(s) are not integer pixel coordinates in [-32768, 32767]";
    return TDR_E_ARG;
  }
  return TDR_OK;
}

int tdr_matrix_triangulate_points_i16(tdr_ctx* ctx, const int16_t* pts16, int n_cam_pts, int64_t n_sys,
                                      double* out_xyz, double* out_err, int64_t* first_bad) {
  if (first_bad) *first_bad = -1;
  if (!ctx || !pts16 || !out_xyz) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];
  if (n_cam_pts < 1 || n_sys < 0) return TDR_E_SHAPE;
  if (n_sys == 0) return TDR_OK;
  const int n_used = n_cam_pts < ctx->cams.n_cam ? n_cam_pts : ctx->cams.n_cam;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  const int64_t chunk = n_sys < kChunkSystems ? n_sys : kChunkSystems;
  const size_t pts_bytes = (size_t)n_used * chunk * 4, xyz_bytes = (size_t)chunk * 24;
  const size_t err_bytes = out_err ? (size_t)chunk * 8 : 0;
  int rc = ensure_ws(ctx, pts_bytes + 16 + xyz_bytes + err_bytes);
  if (rc != TDR_OK) return rc;
  char* d_in = ctx->d_ws;
  double* d_xyz = reinterpret_cast<double*>(ctx->d_ws + ((pts_bytes + 15) & ~(size_t)15));
  double* d_err = out_err ? d_xyz + 3 * chunk : nullptr;
  hipStream_t st = ctx->stream;
  for (int64_t s0 = 0; s0 < n_sys; s0 += chunk) {
    const int64_t n = (n_sys - s0) < chunk ? (n_sys - s0) : chunk;
    for (int c = 0; c < n_used; ++c)
      TDR_HIP(hipMemcpyAsync(d_in + (size_t)c * chunk * 4, pts16 + 2 * ((size_t)c * n_sys + s0), (size_t)n * 4,
                             hipMemcpyHostToDevice, st));
    hipError_t e = launch_matrix_i16(ctx->cams, d_in, n_used, chunk, n, d_xyz, d_err, ctx->d_first_bad, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    TDR_HIP(hipMemcpyAsync(out_xyz + 3 * s0, d_xyz, (size_t)n * 24, hipMemcpyDeviceToHost, st));
    if (out_err) TDR_HIP(hipMemcpyAsync(out_err + s0, d_err, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    int64_t fb = -1;
    rc = read_first_bad(ctx, st, &fb);
    if (rc != TDR_OK) return rc;
    if (fb >= 0) {
      if (first_bad) *first_bad = s0 + fb;
      return TDR_E_TOO_FEW;
    }
  }
  return TDR_OK;
}

int tdr_first_bad(tdr_ctx* ctx, void* stream, int64_t* first_bad) {
  if (!ctx || !ctx->shards.empty()) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  return read_first_bad(ctx, pick_stream(ctx, stream), first_bad);
}

int tdr_dist_from_ray_device(tdr_ctx* ctx, const int32_t* d_cam_idx, const double* d_uv,
                             const double* d_point, int64_t n, double* d_dist, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_cam_idx || !d_uv || !d_point || !d_dist) return TDR_E_ARG;
  if (n < 0) return TDR_E_SHAPE;
  if (!ctx->cams.has_ray) return TDR_E_ARG;
  if (reinterpret_cast<uintptr_t>(d_uv) & 15) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  hipError_t e = launch_dist_from_ray(ctx->cams, d_cam_idx, d_uv, d_point, n, d_dist,
                                      pick_stream(ctx, stream));
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return TDR_OK;
}

int tdr_dist_from_ray_batch(tdr_ctx* ctx, const int32_t* cam_idx, const double* uv, const double* point,
                            int64_t n, double* out_dist) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !cam_idx || !uv || !point || !out_dist) return TDR_E_ARG;
  if (n < 0) return TDR_E_SHAPE;
  if (!ctx->cams.has_ray) return TDR_E_ARG;
  for (int64_t i = 0; i < n; ++i)
    if (cam_idx[i] < 0 || cam_idx[i] >= ctx->cams.n_cam) return TDR_E_SHAPE;
  if (n == 0) return TDR_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  // block layout (host and device alike): [uv n*16 | point n*24 | cam_idx n*4, padded to 8 | dist n*8]
  const size_t o_pt = (size_t)n * 16, o_idx = o_pt + (size_t)n * 24;
  const size_t o_dist = (o_idx + (size_t)n * 4 + 7) & ~(size_t)7, total = o_dist + (size_t)n * 8;
  int rc = ensure_ws(ctx, total);
  if (rc == TDR_OK) rc = ensure_stage(ctx, total);
  if (rc != TDR_OK) return rc;
  char* h = ctx->h_stage;
  std::memcpy(h, uv, (size_t)n * 16);
  std::memcpy(h + o_pt, point, (size_t)n * 24);
  std::memcpy(h + o_idx, cam_idx, (size_t)n * 4);
  hipStream_t st = ctx->stream;
  // Small batches (the classifier issues one per frame, a few hundred distances): the kernel reads its inputs from
  // and writes its results to the pinned block directly — one launch and one synchronisation instead of copy,
  // launch, copy, synchronisation (TDR_DIST_ZERO_COPY=0: the copying form, A/B).
  static const bool zero_copy = [] {
    const char* e = std::getenv("TDR_DIST_ZERO_COPY");
    return !(e && std::atoi(e) == 0);
  }();
  char* hd = nullptr;
  if (zero_copy && n <= 4096 && hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0) == hipSuccess && hd) {
    hipError_t ez = launch_dist_from_ray(ctx->cams, reinterpret_cast<const int32_t*>(hd + o_idx),
                                         reinterpret_cast<const double*>(hd), reinterpret_cast<const double*>(hd + o_pt),
                                         n, reinterpret_cast<double*>(hd + o_dist), st);
    if (ez != hipSuccess) return hip_fail(ez, "kernel launch");
    TDR_HIP(hipStreamSynchronize(st));
    std::memcpy(out_dist, h + o_dist, (size_t)n * 8);
    return TDR_OK;
  }
  TDR_HIP(hipMemcpyAsync(ctx->d_ws, h, o_idx + (size_t)n * 4, hipMemcpyHostToDevice, st));
  hipError_t e = launch_dist_from_ray(ctx->cams, reinterpret_cast<const int32_t*>(ctx->d_ws + o_idx),
                                      reinterpret_cast<const double*>(ctx->d_ws),
                                      reinterpret_cast<const double*>(ctx->d_ws + o_pt), n,
                                      reinterpret_cast<double*>(ctx->d_ws + o_dist), st);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  TDR_HIP(hipMemcpyAsync(h + o_dist, ctx->d_ws + o_dist, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  std::memcpy(out_dist, h + o_dist, (size_t)n * 8);
  return TDR_OK;
}

int tdr_dist_from_ray(tdr_ctx* ctx, const double* cam_aux, const double uv[2], const double point[3],
                      double* out_dist) {
  if (!ctx || !cam_aux || !uv || !point || !out_dist) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];
  const RayCam rc = make_raycam(cam_aux);
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  if (knob(K_POINT_ARGS, 1, 0, 1)) {  // result through the pinned slot: one launch, one wait
    const unsigned long long seq = ++ctx->slot_seq;
    hipError_t e = launch_dist_from_ray_point(rc, uv[0], uv[1], point, ctx->d_slot,
                                              reinterpret_cast<unsigned long long*>(ctx->d_slot + 4), seq, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    const int rcw = wait_slot(ctx, seq);
    if (rcw != TDR_OK) return rcw;
    *out_dist = ctx->h_slot[0];
    return TDR_OK;
  }
  double* d_out = ctx->d_point + kMaxCam * 12 + kMaxCam * 2;
  hipError_t e = launch_dist_from_ray_point(rc, uv[0], uv[1], point, d_out, nullptr, 0, ctx->stream);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  TDR_HIP(hipMemcpyAsync(out_dist, d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TDR_HIP(hipStreamSynchronize(ctx->stream));
  return TDR_OK;
}

int tdr_synth_detections_device(tdr_ctx* ctx, uint64_t seed, int n_drones, int64_t sys_offset,
                                int64_t n_sys, int64_t cam_stride, double noise_px,
                                double dropout_p, int round_to_int, int ray_model, double* d_pts,
                                double* d_truth, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_pts) return TDR_E_ARG;
  if (n_drones < 1 || n_sys < 0 || cam_stride < n_sys || sys_offset < 0) return TDR_E_SHAPE;
  if (ctx->cams.n_cam < 2) return TDR_E_SHAPE;
  if (ray_model && !ctx->cams.has_ray) return TDR_E_ARG;
  if (reinterpret_cast<uintptr_t>(d_pts) & 15) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  hipError_t e = launch_synth(ctx->cams, seed, n_drones, sys_offset, n_sys, cam_stride, noise_px,
                              dropout_p, round_to_int, ray_model, d_pts, d_truth,
                              pick_stream(ctx, stream));
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return TDR_OK;
}

/* ---------------------------------------------------------------------------- candidate scoring */
int tdr_decode_combination(int n_cam, const int32_t* det_count_frame, int64_t leaf, int32_t* digits) {
  if (!det_count_frame || !digits || n_cam < 1 || leaf < 0) return TDR_E_ARG;
  for (int c = n_cam - 1; c >= 0; --c) {
    const int64_t size = (int64_t)det_count_frame[c] + 1;
    digits[c] = (int32_t)(leaf % size);
    leaf /= size;
  }
  return leaf == 0 ? TDR_OK : TDR_E_ARG;
}

static int matrix_score_combinations_impl(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                                          double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                                          int64_t* n_out);

int tdr_matrix_score_combinations(tdr_ctx* ctx, int n_frames, const int32_t* det_count,
                                  const double* dets, double error_max, int min_cameras,
                                  tdr_candidate* out, int64_t cap, int64_t* n_out) {
  return guarded([&] {
    return matrix_score_combinations_impl(ctx, n_frames, det_count, dets, error_max, min_cameras, out, cap, n_out);
  });
}

static int matrix_score_combinations_impl(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                                          double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                                          int64_t* n_out) {
  static_assert(sizeof(tdr_candidate) == 48, "tdr_candidate layout");
  if (n_out) *n_out = 0;
  if (!ctx || !det_count || !n_out || n_frames < 0 || cap < 0 || (cap > 0 && !out)) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];  // one launch scores a whole recording: first GPU
  ctx->sweep_result_n = -1;
  if (n_frames == 0) return TDR_OK;
  const int n_cam = ctx->cams.n_cam;
  // host-side index: detection offsets (frame-major, camera-major) and leaf offsets per frame
  std::vector<long long> det_off((size_t)n_frames * n_cam), leaf_off((size_t)n_frames + 1);
  long long total_det = 0, total_leaf = 0;
  // Above this many full combinations the brute-force leaf kernel (which visits every leaf) hands over to
  // the breadth-first sweep (which, like the reference, never enters a cut subtree): 8 cameras x 4 drones is
  // 3.9e5 leaves per frame, 16 cameras x 4 drones 1.5e11.  Measured (tools/time_matrix_score.py,
  // profiles/r02_matrix_score_crossover.jsonl): level at 1e7 leaves in total (1.86 vs 1.90 ms), the sweep 13 %
  // ahead at 5e7, 1.6x at 8e8 (8 cameras x 4 drones x 3000 frames: 28.5 vs 17.4 ms), 3.7x at 3e8 leaves of a
  // 6-drone recording; the reference's S09_D6 recording (3.4e9 leaves): 189 vs 83 ms.
  static const long long kLeafKernelLimit = [] {
    const char* e = getenv("TDR_SCORE_LEAF_LIMIT");
    return e ? atoll(e) : (1ll << 25);
  }();
  bool too_many = false;
  for (int f = 0; f < n_frames && !too_many; ++f) {
    leaf_off[f] = total_leaf;
    long long leaves = 1;
    for (int c = 0; c < n_cam; ++c) {
      const int32_t k = det_count[(size_t)f * n_cam + c];
      if (k < 0) return TDR_E_SHAPE;
      det_off[(size_t)f * n_cam + c] = total_det;
      total_det += k;
      if (leaves > kLeafKernelLimit / (k + 1)) {
        too_many = true;
        break;
      }
      leaves *= (k + 1);
    }
    total_leaf += leaves;
    too_many |= total_leaf > kLeafKernelLimit;
  }
  if (too_many)
    return tdr_score_combinations_levelwise(ctx, 0, n_frames, det_count, dets, error_max, min_cameras, out, cap,
                                            n_out);
  leaf_off[n_frames] = total_leaf;
  if (total_det > 0 && !dets) return TDR_E_ARG;

  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t sort_bytes = 0;
  if (cap > 0) {
    hipError_t es = sort_candidates(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, cap, ctx->stream);
    if (es != hipSuccess) return hip_fail(es, "hipcub sort size");
  }
  const size_t b_leaf = up(leaf_off.size() * 8), b_doff = up(det_off.size() * 8),
               b_cnt = up((size_t)n_frames * n_cam * 4), b_det = up((size_t)total_det * 16 + 16),
               b_out = up((size_t)cap * sizeof(tdr_candidate) + 16), b_key = up((size_t)cap * 8 + 16),
               b_tmp = up(sort_bytes + 16);
  int rc = ensure_ws(ctx, b_leaf + b_doff + b_cnt + b_det + 2 * b_out + 2 * b_key + b_tmp);
  if (rc != TDR_OK) return rc;
  char* w = ctx->d_ws;
  long long* d_leaf = reinterpret_cast<long long*>(w);            w += b_leaf;
  long long* d_doff = reinterpret_cast<long long*>(w);            w += b_doff;
  int* d_cnt = reinterpret_cast<int*>(w);                         w += b_cnt;
  double* d_det = reinterpret_cast<double*>(w);                   w += b_det;
  void* d_out = w;                                                w += b_out;
  void* d_sorted = w;                                             w += b_out;
  unsigned long long* d_keys = reinterpret_cast<unsigned long long*>(w);      w += b_key;
  unsigned long long* d_keys_out = reinterpret_cast<unsigned long long*>(w);  w += b_key;
  void* d_tmp = w;
  hipStream_t st = ctx->stream;
  unsigned long long* d_n = ctx->d_scalars + 1;
  TDR_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), st));
  TDR_HIP(hipMemcpyAsync(d_leaf, leaf_off.data(), leaf_off.size() * 8, hipMemcpyHostToDevice, st));
  TDR_HIP(hipMemcpyAsync(d_doff, det_off.data(), det_off.size() * 8, hipMemcpyHostToDevice, st));
  TDR_HIP(hipMemcpyAsync(d_cnt, det_count, (size_t)n_frames * n_cam * 4, hipMemcpyHostToDevice, st));
  if (total_det > 0)
    TDR_HIP(hipMemcpyAsync(d_det, dets, (size_t)total_det * 16, hipMemcpyHostToDevice, st));
  hipError_t e = launch_score_leaves(ctx->cams, n_frames, d_leaf, d_cnt, d_doff, d_det, error_max,
                                     min_cameras, total_leaf, d_out, d_keys, cap, d_n, st);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  unsigned long long n = 0;
  TDR_HIP(hipMemcpyAsync(&n, d_n, sizeof(n), hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  *n_out = (int64_t)n;
  const int64_t got = (int64_t)n < cap ? (int64_t)n : cap;
  if (got > 0) {  // ascending global leaf index == the reference's push order, frame by frame
    size_t tb = sort_bytes;
    e = sort_candidates(d_tmp, tb, d_keys, d_keys_out, d_out, d_sorted, got, st);
    if (e != hipSuccess) return hip_fail(e, "hipcub sort");
    TDR_HIP(hipMemcpyAsync(out, d_sorted, (size_t)got * sizeof(tdr_candidate), hipMemcpyDeviceToHost, st));
    TDR_HIP(hipStreamSynchronize(st));
  }
  return (int64_t)n > cap ? TDR_E_ARG : TDR_OK;
}

/* Level-wise scoring: the reference's prefix-tree walk (src/DroneClassifier.cpp:43-85,156-198) turned into
 * a breadth-first sweep — all nodes of one camera depth that need a triangulation form ONE batch for the
 * existing batch kernel (unchosen cameras are (-1,-1) observations), dead nodes are dropped before the
 * next depth, so exactly the nodes the reference visits are evaluated, each once.  This is the scoring
 * route of the ray triangulator (an LM solve per node is too expensive for the brute-force leaf kernel) and a
 * cross-check of the leaf kernel for the matrix triangulator. */
}  // extern "C"

namespace {

double g_grow_ms = 0;  // TDR_SWEEP_TRACE: time spent re-allocating sweep buffers
int grow(tdr_ctx::DevBuf& b, size_t need) {
  if (b.cap >= need) return TDR_OK;
  const double t0 = now_ms();
  struct Acc {
    double t0;
    ~Acc() { g_grow_ms += now_ms() - t0; }
  } acc{t0};
  if (b.p) {
    TDR_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = need + need / 2 + 4096;
  TDR_HIP(hipMalloc(reinterpret_cast<void**>(&b.p), want));
  b.cap = want;
  return TDR_OK;
}

// The sweep on the device: node lists, batches and results stay in HBM; the host reads three 8-byte counters per
// camera depth (children, nodes to solve, survivors) to size the next launches.  Kernels: tdr_kernels.hip.
constexpr int kSweepOutBuf = 11;  // index of the survivors' buffer among tdr_ctx::sweep (B_OUT below)

int levelwise_device(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count, const double* dets,
                     double error_max, int min_cameras, tdr_candidate* out, int64_t cap, int64_t* n_out,
                     tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts) {
  const int n_cam = ctx->cams.n_cam;
  static const bool sweep_trace = std::getenv("TDR_SWEEP_TRACE") != nullptr;
  const double t_sweep0 = now_ms();
  g_grow_ms = 0;
  struct SweepTrace {
    bool on;
    double t0;
    ~SweepTrace() {
      if (on) std::fprintf(stderr, "  device sweep: %.1f ms in the call, %.1f ms of it re-allocating buffers\n", now_ms() - t0, g_grow_ms);
    }
  } sweep_trace_guard{sweep_trace, t_sweep0};
  std::vector<int64_t> det_off((size_t)n_frames * n_cam);
  int64_t total_det = 0;
  for (int f = 0; f < n_frames; ++f)
    for (int c = 0; c < n_cam; ++c) {
      const int32_t k = det_count[(size_t)f * n_cam + c];
      if (k < 0) return TDR_E_SHAPE;
      det_off[(size_t)f * n_cam + c] = total_det;
      total_det += k;
    }
  if (total_det > 0 && !dets) return TDR_E_ARG;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  enum { B_CNT, B_OFF, B_DET, B_CUR, B_CHILD, B_A, B_B, B_PTS, B_XYZ, B_ERR, B_TMP, B_OUT, B_ST, B_CUTS };
  static_assert(B_OUT == kSweepOutBuf, "tdr_fetch_scored_combinations reads sweep[kSweepOutBuf]");
  // the ray walk tracks the LM status of every node it solves (TDR_CANDIDATE_*, include/tdr_amd.h)
  const bool track = path == 1;
  unsigned long long* d_n_cuts = ctx->d_scalars + 4 + kLmCounters * kLmCounterWords;
  const int64_t cuts_room = (cuts && cuts_cap > 0) ? cuts_cap : 0;
  if (track) {
    TDR_HIP(hipMemsetAsync(d_n_cuts, 0, sizeof(unsigned long long), ctx->stream));
    int rcc = grow(ctx->sweep[B_CUTS], (size_t)(cuts_room > 0 ? cuts_room : 1) * sizeof(tdr_candidate));
    if (rcc != TDR_OK) return rcc;
  }
  auto& W = ctx->sweep;
  const size_t n_fc = (size_t)n_frames * n_cam;
  int rc = grow(W[B_CNT], n_fc * 4);
  if (rc == TDR_OK) rc = grow(W[B_OFF], n_fc * 8);
  if (rc == TDR_OK) rc = grow(W[B_DET], (size_t)total_det * 16 + 16);
  if (rc == TDR_OK) rc = grow(W[B_CUR], (size_t)n_frames * sizeof(tdr_candidate));
  if (rc != TDR_OK) return rc;
  TDR_HIP(hipMemcpyAsync(W[B_CNT].p, det_count, n_fc * 4, hipMemcpyHostToDevice, st));
  TDR_HIP(hipMemcpyAsync(W[B_OFF].p, det_off.data(), n_fc * 8, hipMemcpyHostToDevice, st));
  if (total_det > 0) TDR_HIP(hipMemcpyAsync(W[B_DET].p, dets, (size_t)total_det * 16, hipMemcpyHostToDevice, st));
  int* d_overflow = reinterpret_cast<int*>(ctx->d_scalars + 3 + kLmCounters * kLmCounterWords);
  TDR_HIP(hipMemsetAsync(d_overflow, 0, sizeof(unsigned long long), st));
  const int* d_cnt = reinterpret_cast<const int*>(W[B_CNT].p);
  const long long* d_off = reinterpret_cast<const long long*>(W[B_OFF].p);
  const double* d_det = reinterpret_cast<const double*>(W[B_DET].p);
  hipError_t e = launch_sweep_roots(W[B_CUR].p, n_frames, st);
  if (e != hipSuccess) return hip_fail(e, "sweep roots");
  const long long kMaxList = 0x7ffffff0ll;  // hipCUB scans take an int count
  auto read_total = [&](const long long* d_scan, long long n, long long* total) -> int {
    TDR_HIP(hipMemcpyAsync(total, d_scan + n, sizeof(long long), hipMemcpyDeviceToHost, st));
    TDR_HIP(hipStreamSynchronize(st));
    return TDR_OK;
  };
  auto scan = [&](const long long* d_in, long long* d_out, long long n) -> int {  // n elements incl. the terminator
    size_t bytes = 0;
    hipError_t es = sweep_scan(nullptr, bytes, d_in, d_out, n, st);
    if (es != hipSuccess) return hip_fail(es, "scan size");
    int r = grow(W[B_TMP], bytes + 16);
    if (r != TDR_OK) return r;
    es = sweep_scan(W[B_TMP].p, bytes, d_in, d_out, n, st);
    return es == hipSuccess ? TDR_OK : hip_fail(es, "scan");
  };
  long long n_cur = n_frames;
  for (int L = 0; L < n_cam; ++L) {
    if (n_cur == 0) break;
    // sizes -> offsets
    rc = grow(W[B_A], (size_t)(n_cur + 1) * 8);
    if (rc == TDR_OK) rc = grow(W[B_B], (size_t)(n_cur + 1) * 8);
    if (rc != TDR_OK) return rc;
    long long* d_a = reinterpret_cast<long long*>(W[B_A].p);
    long long* d_b = reinterpret_cast<long long*>(W[B_B].p);
    e = launch_sweep_sizes(W[B_CUR].p, n_cur, d_cnt, n_cam, L, d_a, st);
    if (e != hipSuccess) return hip_fail(e, "sweep sizes");
    rc = scan(d_a, d_b, n_cur + 1);
    long long n_children = 0;
    if (rc == TDR_OK) rc = read_total(d_b, n_cur, &n_children);
    if (rc != TDR_OK) return rc;
    if (n_children > kMaxList) return TDR_E_SHAPE;
    // expand; d_b = offsets (kept until expand is done), marks -> slots in fresh arrays
    rc = grow(W[B_CHILD], (size_t)n_children * sizeof(tdr_candidate));
    if (rc == TDR_OK) rc = grow(W[B_XYZ], (size_t)(n_children + 1) * 8);  // marks (re-used as scratch before the solve)
    if (rc != TDR_OK) return rc;
    long long* d_marks = reinterpret_cast<long long*>(W[B_XYZ].p);
    e = launch_sweep_expand(W[B_CUR].p, n_cur, d_b, d_cnt, n_cam, L, W[B_CHILD].p, d_marks, d_overflow, st);
    if (e != hipSuccess) return hip_fail(e, "sweep expand");
    rc = grow(W[B_A], (size_t)(n_children + 1) * 8);  // offsets are no longer needed: d_a becomes the slot array
    if (rc != TDR_OK) return rc;
    long long* d_slot = reinterpret_cast<long long*>(W[B_A].p);
    rc = scan(d_marks, d_slot, n_children + 1);
    long long n_eval = 0;
    if (rc == TDR_OK) rc = read_total(d_slot, n_children, &n_eval);
    if (rc != TDR_OK) return rc;
    // one batch for every node the reference would triangulate at this depth
    double *d_xyz = nullptr, *d_err = nullptr;
    int* d_lm_status = nullptr;
    if (n_eval > 0) {
      rc = grow(W[B_PTS], (size_t)n_cam * (size_t)n_eval * 16);
      if (rc == TDR_OK) rc = grow(W[B_XYZ], (size_t)n_eval * 24 + 16);
      if (rc == TDR_OK) rc = grow(W[B_ERR], (size_t)n_eval * 8 + 16);
      if (rc == TDR_OK && track) rc = grow(W[B_ST], (size_t)n_eval * 4 + 16);
      if (rc != TDR_OK) return rc;
      if (track) d_lm_status = reinterpret_cast<int*>(W[B_ST].p);
      double* d_pts = reinterpret_cast<double*>(W[B_PTS].p);
      d_xyz = reinterpret_cast<double*>(W[B_XYZ].p);
      d_err = reinterpret_cast<double*>(W[B_ERR].p);
      e = launch_sweep_gather(W[B_CHILD].p, n_children, d_slot, d_cnt, d_off, d_det, n_cam, L, n_eval, d_pts, st);
      if (e != hipSuccess) return hip_fail(e, "sweep gather");
      e = path == 0 ? launch_matrix(ctx->cams, ctx->matrix_variant, d_pts, n_cam, n_eval, n_eval, d_xyz, d_err,
                                    ctx->d_scalars, st)
                    : launch_ray(ctx->cams, d_pts, n_cam, n_eval, n_eval, d_xyz, d_err, nullptr, d_lm_status,
                                 ctx->d_scalars, st);
      if (e != hipSuccess) return hip_fail(e, "sweep solve");
    }
    // prune, then compact the survivors into the next depth's list
    rc = grow(W[B_B], (size_t)(n_children + 1) * 8);
    if (rc != TDR_OK) return rc;
    long long* d_alive = reinterpret_cast<long long*>(W[B_B].p);
    e = launch_sweep_prune(W[B_CHILD].p, n_children, d_slot, d_xyz, d_err, error_max, d_lm_status, L,
                           track ? W[B_CUTS].p : nullptr, cuts_room, track ? d_n_cuts : nullptr, d_alive, st);
    if (e != hipSuccess) return hip_fail(e, "sweep prune");
    rc = scan(d_alive, d_slot, n_children + 1);  // d_slot now holds the survivors' positions
    long long n_alive = 0;
    if (rc == TDR_OK) rc = read_total(d_slot, n_children, &n_alive);
    if (rc != TDR_OK) return rc;
    rc = grow(W[B_CUR], (size_t)(n_alive > 0 ? n_alive : 1) * sizeof(tdr_candidate));
    if (rc != TDR_OK) return rc;
    e = launch_sweep_compact(W[B_CHILD].p, n_children, d_slot, W[B_CUR].p, st);
    if (e != hipSuccess) return hip_fail(e, "sweep compact");
    n_cur = n_alive;
  }
  // push rule on the full combinations, back to the host in (frame, leaf) order
  long long n_push = 0;
  if (n_cur > 0) {
    rc = grow(W[B_A], (size_t)(n_cur + 1) * 8);
    if (rc == TDR_OK) rc = grow(W[B_B], (size_t)(n_cur + 1) * 8);
    if (rc != TDR_OK) return rc;
    long long* d_keep = reinterpret_cast<long long*>(W[B_A].p);
    long long* d_pos = reinterpret_cast<long long*>(W[B_B].p);
    e = launch_sweep_final(W[B_CUR].p, n_cur, min_cameras, d_keep, st);
    if (e != hipSuccess) return hip_fail(e, "sweep final");
    rc = scan(d_keep, d_pos, n_cur + 1);
    if (rc == TDR_OK) rc = read_total(d_pos, n_cur, &n_push);
    if (rc != TDR_OK) return rc;
    if (n_push > 0) {
      rc = grow(W[B_OUT], (size_t)n_push * sizeof(tdr_candidate));
      if (rc != TDR_OK) return rc;
      e = launch_sweep_compact(W[B_CUR].p, n_cur, d_pos, W[B_OUT].p, st);
      if (e != hipSuccess) return hip_fail(e, "sweep compact");
    }
  }
  int overflow = 0;
  TDR_HIP(hipMemcpyAsync(&overflow, d_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
  if (n_cuts) *n_cuts = 0;
  if (track) {
    unsigned long long nc = 0;
    TDR_HIP(hipMemcpyAsync(&nc, d_n_cuts, sizeof(nc), hipMemcpyDeviceToHost, st));
    TDR_HIP(hipStreamSynchronize(st));
    if (n_cuts) *n_cuts = (int64_t)nc;
    const int64_t take = (int64_t)nc < cuts_room ? (int64_t)nc : cuts_room;
    if (take > 0)
      TDR_HIP(hipMemcpyAsync(cuts, W[B_CUTS].p, (size_t)take * sizeof(tdr_candidate), hipMemcpyDeviceToHost, st));
  }
  const int64_t got = n_push < cap ? n_push : cap;
  TDR_HIP(hipStreamSynchronize(st));
  const double t_copy0 = now_ms();
  if (got > 0) TDR_HIP(hipMemcpyAsync(out, W[B_OUT].p, (size_t)got * sizeof(tdr_candidate), hipMemcpyDeviceToHost, st));
  TDR_HIP(hipStreamSynchronize(st));
  if (sweep_trace)
    std::fprintf(stderr, "  device sweep: %lld survivors, %.1f MB copied out in %.1f ms\n", (long long)got,
                 (double)got * sizeof(tdr_candidate) / 1e6, now_ms() - t_copy0);
  if (overflow) return TDR_E_SHAPE;
  *n_out = n_push;
  ctx->sweep_result_n = n_push;  // tdr_fetch_scored_combinations
  return n_push > cap ? TDR_E_ARG : TDR_OK;
}

int levelwise_host(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count, const double* dets,
                   double error_max, int min_cameras, tdr_candidate* out, int64_t cap, int64_t* n_out,
                   tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts);

int score_levelwise(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count, const double* dets,
                    double error_max, int min_cameras, tdr_candidate* out, int64_t cap, int64_t* n_out,
                    tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts) {
  if (n_out) *n_out = 0;
  if (n_cuts) *n_cuts = 0;
  if (!ctx || !det_count || !n_out || n_frames < 0 || cap < 0 || (cap > 0 && !out)) return TDR_E_ARG;
  if (cuts_cap < 0 || (cuts_cap > 0 && !cuts)) return TDR_E_ARG;
  if (path != 0 && path != 1) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];
  ctx->sweep_result_n = -1;
  if (path == 1 && !ctx->cams.has_ray) return TDR_E_ARG;
  if (n_frames == 0) return TDR_OK;
  // TDR_SWEEP_HOST=1: the round-1 sweep (nodes expanded on the host, one H2D -> launch -> D2H per depth); kept
  // as the cross-check of the device sweep (tests/test_gpu_score_combinations.py) and for A/B timing
  const bool host_sweep = knob_sweep_on_host();  // tests flip it through tdr_debug_set_knob
  return guarded([&] {
    return (host_sweep ? levelwise_host : levelwise_device)(ctx, path, n_frames, det_count, dets, error_max,
                                                            min_cameras, out, cap, n_out, cuts, cuts_cap, n_cuts);
  });
}

}  // namespace

extern "C" {

int tdr_score_combinations_levelwise(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count,
                                     const double* dets, double error_max, int min_cameras,
                                     tdr_candidate* out, int64_t cap, int64_t* n_out) {
  return score_levelwise(ctx, path, n_frames, det_count, dets, error_max, min_cameras, out, cap, n_out, nullptr, 0,
                         nullptr);
}

int tdr_ray_score_combinations_ex(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                                  double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                                  int64_t* n_out, tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts) {
  return score_levelwise(ctx, 1, n_frames, det_count, dets, error_max, min_cameras, out, cap, n_out, cuts, cuts_cap,
                         n_cuts);
}

int tdr_fetch_scored_combinations(tdr_ctx* ctx, tdr_candidate* out, int64_t cap, int64_t* n_out) {
  if (n_out) *n_out = 0;
  if (!ctx || !n_out || cap < 0 || (cap > 0 && !out)) return TDR_E_ARG;
  if (!ctx->shards.empty()) ctx = ctx->shards[0];
  std::lock_guard<std::mutex> lock(ctx->mu);
  const int64_t n = ctx->sweep_result_n;
  if (n < 0) return TDR_E_ARG;  // nothing held: the last scoring call did not go through the device sweep
  *n_out = n;
  if (n > cap) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  if (n > 0) {
    const double t0 = now_ms();
    TDR_HIP(hipMemcpyAsync(out, ctx->sweep[kSweepOutBuf].p, (size_t)n * sizeof(tdr_candidate), hipMemcpyDeviceToHost,
                           ctx->stream));
    TDR_HIP(hipStreamSynchronize(ctx->stream));
    if (std::getenv("TDR_SWEEP_TRACE"))
      std::fprintf(stderr, "  fetch: %.1f MB copied out in %.1f ms\n", (double)n * sizeof(tdr_candidate) / 1e6, now_ms() - t0);
  }
  return TDR_OK;
}

}  // extern "C"

namespace {

int levelwise_host(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count, const double* dets,
                   double error_max, int min_cameras, tdr_candidate* out, int64_t cap, int64_t* n_out,
                   tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts) {
  const int n_cam = ctx->cams.n_cam;
  int64_t cuts_seen = 0;
  std::vector<int64_t> det_off((size_t)n_frames * n_cam);
  int64_t total_det = 0;
  for (int f = 0; f < n_frames; ++f)
    for (int c = 0; c < n_cam; ++c) {
      const int32_t k = det_count[(size_t)f * n_cam + c];
      if (k < 0) return TDR_E_SHAPE;
      det_off[(size_t)f * n_cam + c] = total_det;
      total_det += k;
    }
  if (total_det > 0 && !dets) return TDR_E_ARG;
  struct Node {
    int32_t frame;
    int16_t count, flags;
    int64_t prefix;
    double x, y, z, err;
  };
  const int64_t kMaxNodes = int64_t(1) << 26;
  std::vector<Node> cur((size_t)n_frames), next, pending;
  for (int f = 0; f < n_frames; ++f) cur[(size_t)f] = Node{f, 0, 0, 0, 0, 0, 0, 0};
  std::vector<double> pts, xyz, err;
  std::vector<int32_t> lm_status;
  std::vector<int32_t> digits((size_t)n_cam);
  for (int L = 0; L < n_cam; ++L) {
    next.clear();
    pending.clear();
    // children in (parent, digit) order keep the list sorted by (frame, prefix); evaluated children get a
    // placeholder slot now and are filled / dropped after the batch
    std::vector<int64_t> pending_slot;
    for (const Node& nd : cur) {
      const int64_t size = (int64_t)det_count[(size_t)nd.frame * n_cam + L] + 1;
      if (nd.prefix > (INT64_MAX - size) / size) return TDR_E_SHAPE;
      for (int64_t d = 0; d < size; ++d) {
        Node ch = nd;
        ch.prefix = nd.prefix * size + d;
        if (d > 0) {
          ch.count = (int16_t)(nd.count + 1);
          if (ch.count >= 2) {  // the reference triangulates this node
            pending_slot.push_back((int64_t)next.size());
            pending.push_back(ch);
          }
        }
        next.push_back(ch);
        if ((int64_t)next.size() > kMaxNodes) return TDR_E_SHAPE;
      }
    }
    const int64_t n_eval = (int64_t)pending.size();
    std::vector<char> dead(next.size(), 0);
    if (n_eval > 0) {
      pts.assign((size_t)n_cam * n_eval * 2, -1.0);
      xyz.resize((size_t)n_eval * 3);
      err.resize((size_t)n_eval);
      for (int64_t i = 0; i < n_eval; ++i) {
        const Node& ch = pending[(size_t)i];
        int64_t p = ch.prefix;
        for (int c = L; c >= 0; --c) {  // digits of cameras 0..L
          const int64_t size = (int64_t)det_count[(size_t)ch.frame * n_cam + c] + 1;
          digits[(size_t)c] = (int32_t)(p % size);
          p /= size;
        }
        for (int c = 0; c <= L; ++c)
          if (digits[(size_t)c] > 0) {
            const double* d = dets + 2 * (det_off[(size_t)ch.frame * n_cam + c] + digits[(size_t)c] - 1);
            pts[2 * ((size_t)c * n_eval + i)] = d[0];
            pts[2 * ((size_t)c * n_eval + i) + 1] = d[1];
          }
      }
      int64_t fb = -1;
      HostOut ho{xyz.data(), err.data()};
      if (path == 1) {
        lm_status.resize((size_t)n_eval);
        ho.status = lm_status.data();
      }
      int rc = run_host_batch(ctx, path, pts.data(), n_cam, n_eval, n_eval, ho, &fb);
      if (rc != TDR_OK) return rc;
      for (int64_t i = 0; i < n_eval; ++i) {
        Node& ch = next[(size_t)pending_slot[(size_t)i]];
        const bool flagged = path == 1 && (lm_status[(size_t)i] & (TDR_LM_CAP | TDR_LM_NOT_CONVERGED)) != 0;
        if (err[(size_t)i] > error_max) {  // cut (src/DroneClassifier.cpp:185-187): the reference's predicate, a NaN is not cut
          dead[(size_t)pending_slot[(size_t)i]] = 1;
          if (flagged) {
            if (cuts && cuts_seen < cuts_cap) {
              tdr_candidate& c = cuts[cuts_seen];
              c.frame = ch.frame;
              c.n_used = ch.count;
              c.flags = (int16_t)(TDR_CANDIDATE_FLAGGED_CUT | (L << 8));
              c.leaf = ch.prefix;
              c.x = xyz[3 * (size_t)i]; c.y = xyz[3 * (size_t)i + 1]; c.z = xyz[3 * (size_t)i + 2];
              c.err = err[(size_t)i];
            }
            ++cuts_seen;
          }
        } else {
          if (flagged) ch.flags |= TDR_CANDIDATE_FLAGGED;
          ch.x = xyz[3 * (size_t)i];
          ch.y = xyz[3 * (size_t)i + 1];
          ch.z = xyz[3 * (size_t)i + 2];
          ch.err = err[(size_t)i];
        }
      }
    }
    cur.clear();
    for (size_t i = 0; i < next.size(); ++i)
      if (!dead[i]) cur.push_back(next[i]);
  }
  int64_t n = 0;
  for (const Node& nd : cur) {
    if (nd.count < 2 || nd.count < min_cameras) continue;  // push rule (src/DroneClassifier.cpp:190-197)
    if (n < cap) {
      tdr_candidate& o = out[n];
      o.frame = nd.frame;
      o.n_used = nd.count;
      o.flags = nd.flags;
      o.leaf = nd.prefix;
      o.x = nd.x; o.y = nd.y; o.z = nd.z; o.err = nd.err;
    }
    ++n;
  }
  *n_out = n;
  if (n_cuts) *n_cuts = cuts_seen;
  return n > cap ? TDR_E_ARG : TDR_OK;
}

}  // namespace

extern "C" {

int tdr_ray_score_combinations(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                               double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                               int64_t* n_out) {
  return tdr_score_combinations_levelwise(ctx, 1, n_frames, det_count, dets, error_max, min_cameras, out, cap,
                                          n_out);
}

/* ---------------------------------------------------------------------------- multi-GPU, device-resident */
int tdr_multi_shard_count(const tdr_ctx* ctx) { return ctx ? (int)ctx->shards.size() : 0; }

int tdr_multi_shard(tdr_ctx* ctx, int g, tdr_ctx** shard) {
  if (!ctx || !shard || g < 0 || g >= (int)ctx->shards.size()) return TDR_E_ARG;
  *shard = ctx->shards[(size_t)g];
  return TDR_OK;
}

int tdr_multi_matrix_triangulate_gather(tdr_ctx* ctx, const double* const* d_pts, int n_cam_pts,
                                        const int64_t* cam_stride, const int64_t* n_sys,
                                        double* const* d_xyz, double* d_xyz_root, int transport,
                                        int64_t* first_bad) {
  if (first_bad) *first_bad = -1;
  if (!ctx || ctx->shards.empty() || !d_pts || !cam_stride || !n_sys || !d_xyz) return TDR_E_ARG;
  if (transport < 0 || transport > 2) return TDR_E_ARG;
  const int G = (int)ctx->shards.size();
  bool distinct = true;
  for (int a = 0; a < G; ++a)
    for (int b = a + 1; b < G; ++b) distinct &= ctx->shards[(size_t)a]->device != ctx->shards[(size_t)b]->device;
  if (transport == 0) transport = (distinct && rccl().ok) ? 1 : 2;  // auto: RCCL when it can be used
  if (transport == 1 && (!distinct || !rccl().ok)) {
    g_last_error = !rccl().ok ? "librccl.so not loadable" : "RCCL needs one shard per distinct device";
    return TDR_E_RCCL;
  }
  // The shards' slices are cut into pieces; piece c of every shard is solved on the shard's compute stream and
  // its [n][3] block travels to shard 0's device on the shard's copy stream while piece c + 1 is being solved
  // ("pipeline the gather in chunks behind compute").  No collective on the data path itself.
  static const int kGatherChunks = [] {
    const char* e = std::getenv("TDR_GATHER_CHUNKS");
    const int v = e ? std::atoi(e) : 4;
    return v < 1 ? 1 : (v > kPipeSlots ? kPipeSlots : v);
  }();
  const int NC = d_xyz_root ? kGatherChunks : 1;
  std::vector<int64_t> off((size_t)G + 1, 0);
  for (int g = 0; g < G; ++g) off[(size_t)g + 1] = off[(size_t)g] + n_sys[g];
  RcclApi& R = rccl();
  if (d_xyz_root && transport == 1 && ctx->comms.empty()) {
    std::vector<int> devs((size_t)G);
    for (int g = 0; g < G; ++g) devs[(size_t)g] = ctx->shards[(size_t)g]->device;
    ctx->comms.assign((size_t)G, nullptr);
    ncclResult_t r = R.CommInitAll(ctx->comms.data(), G, devs.data());
    if (r != ncclSuccess) {
      g_last_error = std::string("ncclCommInitAll: ") + (R.GetErrorString ? R.GetErrorString(r) : "error");
      ctx->comms.clear();
      return TDR_E_RCCL;
    }
  }
  if (d_xyz_root)
    for (int g = 0; g < G; ++g) {
      TDR_HIP(hipSetDevice(ctx->shards[(size_t)g]->device));
      int rc = ensure_pipe(ctx->shards[(size_t)g], 0);  // copy stream + events
      if (rc != TDR_OK) return rc;
    }
  int status = TDR_OK;
  for (int c = 0; c < NC && status == TDR_OK; ++c) {
    // 1. piece c of every shard on its own GPU / compute stream
    for (int g = 0; g < G && status == TDR_OK; ++g) {
      tdr_ctx* sh = ctx->shards[(size_t)g];
      const int64_t c0 = n_sys[g] * c / NC, c1 = n_sys[g] * (c + 1) / NC;
      if (c1 == c0) continue;
      status = run_device_batch(sh, 0, d_pts[g] + 2 * c0, n_cam_pts, cam_stride[g], c1 - c0, d_xyz[g] + 3 * c0, nullptr,
                                nullptr);
      if (status == TDR_OK && d_xyz_root) {
        hipError_t e = hipEventRecord(sh->ev_comp[c], sh->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(sh->s_down, sh->ev_comp[c], 0);
        if (e != hipSuccess) status = hip_fail(e, "gather event");
      }
    }
    if (!d_xyz_root || status != TDR_OK) continue;
    // 2. its blocks onto shard 0's device, on the copy streams
    if (transport == 1) {
      ncclResult_t r = R.GroupStart();
      const bool started = r == ncclSuccess;
      for (int g = 0; g < G && r == ncclSuccess; ++g) {
        tdr_ctx* sh = ctx->shards[(size_t)g];
        const int64_t c0 = n_sys[g] * c / NC, c1 = n_sys[g] * (c + 1) / NC;
        if (c1 == c0) continue;
        if (hipSetDevice(sh->device) != hipSuccess) { status = TDR_E_HIP; break; }
        r = R.Send(d_xyz[g] + 3 * c0, (size_t)(c1 - c0) * 3, ncclDouble, 0, ctx->comms[(size_t)g], sh->s_down);
        if (r != ncclSuccess) break;
        if (hipSetDevice(ctx->shards[0]->device) != hipSuccess) { status = TDR_E_HIP; break; }
        r = R.Recv(d_xyz_root + 3 * (off[(size_t)g] + c0), (size_t)(c1 - c0) * 3, ncclDouble, g, ctx->comms[0],
                   ctx->shards[0]->s_down);
      }
      // a group that was opened is ALWAYS closed, whatever happened inside it
      const ncclResult_t r2 = started ? R.GroupEnd() : ncclSuccess;
      if (r != ncclSuccess || r2 != ncclSuccess) {
        g_last_error = std::string("rccl gather: ") +
                       (R.GetErrorString ? R.GetErrorString(r != ncclSuccess ? r : r2) : "error");
        status = TDR_E_RCCL;
      }
    } else {  // peer copies over xGMI (also the only option when shards share a device)
      for (int g = 0; g < G && status == TDR_OK; ++g) {
        tdr_ctx* sh = ctx->shards[(size_t)g];
        const int64_t c0 = n_sys[g] * c / NC, c1 = n_sys[g] * (c + 1) / NC;
        if (c1 == c0) continue;
        hipError_t e = hipSetDevice(sh->device);
        if (e == hipSuccess)
          e = hipMemcpyPeerAsync(d_xyz_root + 3 * (off[(size_t)g] + c0), ctx->shards[0]->device, d_xyz[g] + 3 * c0,
                                 sh->device, (size_t)(c1 - c0) * 24, sh->s_down);
        if (e != hipSuccess) status = hip_fail(e, "hipMemcpyPeerAsync");
      }
    }
  }
  // 3. drain every stream that was used (also on the error paths: nothing may stay in flight), and the globally
  // first failing system (8 bytes per shard, host side)
  int64_t base = 0;
  bool any_bad = false;
  for (int g = 0; g < G; ++g) {
    tdr_ctx* sh = ctx->shards[(size_t)g];
    (void)hipSetDevice(sh->device);
    if (sh->s_down) (void)hipStreamSynchronize(sh->s_down);
    int64_t fb = -1;
    int rc = tdr_first_bad(sh, nullptr, &fb);
    if (rc != TDR_OK && status == TDR_OK) status = rc;
    any_bad |= fb >= 0;
  }
  if (status != TDR_OK) return status;
  if (any_bad) {
    // A piece reports its failing system relative to its own start.  Failure is the rare path (the reference
    // throws): find the globally first one by solving the shards once more, unchunked, in order.
    for (int g = 0; g < G; ++g) {
      tdr_ctx* sh = ctx->shards[(size_t)g];
      int rc = run_device_batch(sh, 0, d_pts[g], n_cam_pts, cam_stride[g], n_sys[g], d_xyz[g], nullptr, nullptr);
      int64_t fb = -1;
      if (rc == TDR_OK) rc = tdr_first_bad(sh, nullptr, &fb);
      if (rc != TDR_OK) return rc;
      if (fb >= 0) {
        if (first_bad) *first_bad = base + fb;
        return TDR_E_TOO_FEW;
      }
      base += n_sys[g];
    }
  }
  return TDR_OK;
}

/* ---------------------------------------------------------------------------- memory helpers */
int tdr_device_alloc(tdr_ctx* ctx, int64_t bytes, void** d_ptr) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];  // memory helpers of a multi-GPU context: first GPU
  if (!ctx || !d_ptr || bytes < 0) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  TDR_HIP(hipMalloc(d_ptr, (size_t)(bytes > 0 ? bytes : 1)));
  return TDR_OK;
}

int tdr_device_free(tdr_ctx* ctx, void* d_ptr) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];  // memory helpers of a multi-GPU context: first GPU
  if (!ctx) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  TDR_HIP(hipFree(d_ptr));
  return TDR_OK;
}

int tdr_memcpy_h2d(tdr_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];  // memory helpers of a multi-GPU context: first GPU
  if (!ctx || bytes < 0) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  TDR_HIP(hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
  return TDR_OK;
}

int tdr_memcpy_d2h(tdr_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];  // memory helpers of a multi-GPU context: first GPU
  if (!ctx || bytes < 0) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  TDR_HIP(hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return TDR_OK;
}

int tdr_stream_synchronize(tdr_ctx* ctx, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];  // memory helpers of a multi-GPU context: first GPU
  if (!ctx) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  TDR_HIP(hipStreamSynchronize(pick_stream(ctx, stream)));
  return TDR_OK;
}

int tdr_pack_detections_i16_host(const double* pts, int64_t n_obs, int16_t* out, int* all_exact) {
  if (!pts || !out || n_obs < 0) return TDR_E_ARG;
  if ((reinterpret_cast<uintptr_t>(out) & 3) != 0) return TDR_E_ARG;
  const bool ok = n_obs == 0 || pack_obs_i16(pts, reinterpret_cast<int32_t*>(out), (size_t)n_obs);
  if (all_exact) *all_exact = ok ? 1 : 0;
  return TDR_OK;
}

int tdr_host_alloc(int64_t bytes, void** h_ptr) {
  if (!h_ptr || bytes < 0) return TDR_E_ARG;
  *h_ptr = nullptr;
  TDR_HIP(hipHostMalloc(h_ptr, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault));
  return TDR_OK;
}

int tdr_host_free(void* h_ptr) {
  if (!h_ptr) return TDR_OK;
  TDR_HIP(hipHostFree(h_ptr));
  return TDR_OK;
}

int tdr_time_launches(tdr_ctx* ctx, int path, const double* d_pts, int n_cam_pts,
                      int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err, int warmup,
                      int iters, void* stream, double* ms_per_launch) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !ms_per_launch || iters < 1 || warmup < 0) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = pick_stream(ctx, stream);
  for (int i = 0; i < warmup; ++i) {
    int rc = run_device_batch(ctx, path, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, st);
    if (rc != TDR_OK) return rc;
  }
  EventPair ev;
  TDR_HIP(hipEventCreate(&ev.e0));
  TDR_HIP(hipEventCreate(&ev.e1));
  TDR_HIP(hipEventRecord(ev.e0, st));
  for (int i = 0; i < iters; ++i) {
    int rc = run_device_batch(ctx, path, d_pts, n_cam_pts, cam_stride, n_sys, d_xyz, d_err, st);
    if (rc != TDR_OK) return rc;
  }
  TDR_HIP(hipEventRecord(ev.e1, st));
  TDR_HIP(hipEventSynchronize(ev.e1));
  float ms = 0;
  TDR_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *ms_per_launch = (double)ms / iters;
  return TDR_OK;
}


int tdr_debug_sym3_eig_device(tdr_ctx* ctx, const double* d_in, int64_t n, double* d_out, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_in || !d_out) return TDR_E_ARG;
  if (n < 0) return TDR_E_SHAPE;
  TDR_HIP(hipSetDevice(ctx->device));
  hipError_t e = launch_eig3_debug(d_in, n, d_out, pick_stream(ctx, stream));
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return TDR_OK;
}

int tdr_debug_lm_cascade_device(tdr_ctx* ctx, const double* d_in, int64_t n, int n_rays, double* d_out, void* stream) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_in || !d_out || n_rays < 2) return TDR_E_ARG;
  if (n < 0) return TDR_E_SHAPE;
  TDR_HIP(hipSetDevice(ctx->device));
  hipError_t e = launch_cascade_debug(d_in, n, n_rays, d_out, pick_stream(ctx, stream));
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  return TDR_OK;
}

int tdr_debug_set_knob(const char* name, int value, int unset) {
  return knob_set(name, value, unset) == 0 ? TDR_OK : TDR_E_ARG;
}

int tdr_time_fp64_fma(tdr_ctx* ctx, int waves_per_simd, int iters, double* tflops, double* ns_per_wave_inst) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !tflops || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1) return TDR_E_ARG;
  std::lock_guard<std::mutex> lock(ctx->mu);
  TDR_HIP(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  TDR_HIP(hipGetDeviceProperties(&prop, ctx->device));
  const int simds = prop.multiProcessorCount * 4;
  const int blocks = simds * waves_per_simd / (kBlockThreads / 64);
  hipStream_t st = ctx->stream;
  double* d_out = ctx->d_point;
  hipError_t e = launch_fma_peak(d_out, blocks, iters, st);  // warm-up
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  EventPair ev;
  TDR_HIP(hipEventCreate(&ev.e0));
  TDR_HIP(hipEventCreate(&ev.e1));
  TDR_HIP(hipEventRecord(ev.e0, st));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) {
    e = launch_fma_peak(d_out, blocks, iters, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
  }
  TDR_HIP(hipEventRecord(ev.e1, st));
  TDR_HIP(hipEventSynchronize(ev.e1));
  float ms = 0;
  TDR_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  const double s_per = (double)ms * 1e-3 / reps;
  const double wave_insts_per_simd = (double)waves_per_simd * 16.0 * iters;  // every SIMD issues this many
  *tflops = (double)blocks * kBlockThreads * 16.0 * iters * 2.0 / s_per / 1e12;
  if (ns_per_wave_inst) *ns_per_wave_inst = s_per / wave_insts_per_simd * 1e9;
  return TDR_OK;
}

int tdr_time_stream(tdr_ctx* ctx, int mode, int streams, const double* d_src, double* d_dst,
                    int64_t n_lanes, int warmup, int iters, double* ms_per_launch) {
  if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
  if (!ctx || !d_src || !d_dst || !ms_per_launch || iters < 1 || warmup < 0 || n_lanes < 1) return TDR_E_ARG;
  if (mode < 0 || mode > 3) return TDR_E_ARG;
  if (mode != 1 && streams != 4 && streams != 8 && streams != 16) return TDR_E_ARG;
  TDR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  for (int i = 0; i < warmup; ++i) {
    hipError_t e = launch_stream(mode, streams, d_src, d_dst, n_lanes, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
  }
  EventPair ev;
  TDR_HIP(hipEventCreate(&ev.e0));
  TDR_HIP(hipEventCreate(&ev.e1));
  TDR_HIP(hipEventRecord(ev.e0, st));
  for (int i = 0; i < iters; ++i) {
    hipError_t e = launch_stream(mode, streams, d_src, d_dst, n_lanes, st);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
  }
  TDR_HIP(hipEventRecord(ev.e1, st));
  TDR_HIP(hipEventSynchronize(ev.e1));
  float ms = 0;
  TDR_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *ms_per_launch = (double)ms / iters;
  return TDR_OK;
}

}  // extern "C"
