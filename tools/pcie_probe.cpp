// pcie_probe.cpp — measurement tool (not product): what the host<->device link of this box delivers for the
// shapes the host-buffer entry points move.  Build: hipcc -O2 -o pcie_probe pcie_probe.cpp -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char* dst, const char* src, size_t n, int threads) {
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) {
    size_t a = n * t / threads, b = n * (t + 1) / threads;
    th.emplace_back([=] { std::memcpy(dst + a, src + a, b - a); });
  }
  for (auto& x : th) x.join();
}

// zero-copy probes: a kernel that streams a (host-pinned) input with 16-byte loads and writes 24 bytes per 128 read
// to a (host-pinned) output — the byte ratio of the matrix path at 8 cameras
typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ void zc_kernel(const dbl2* __restrict__ in, double* __restrict__ out, long n_sys, int n_cam) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_sys) return;
  double ax = 0, ay = 0;
  for (int c = 0; c < n_cam; ++c) {
    const dbl2 v = __builtin_nontemporal_load(&in[(long)c * n_sys + i]);
    ax += v.x;
    ay += v.y;
  }
  __builtin_nontemporal_store(ax, &out[3 * i]);
  __builtin_nontemporal_store(ay, &out[3 * i + 1]);
  __builtin_nontemporal_store(ax - ay, &out[3 * i + 2]);
}

int main(int argc, char** argv) {
  const size_t in_bytes = (size_t)(argc > 1 ? atof(argv[1]) : 1.28e9), out_bytes = in_bytes * 3 / 16;
  CK(hipSetDevice(0));
  char *d_in, *d_out;
  CK(hipMalloc((void**)&d_in, in_bytes));
  CK(hipMalloc((void**)&d_out, out_bytes));
  char* page_in = (char*)malloc(in_bytes);
  char* page_out = (char*)malloc(out_bytes);
  memset(page_in, 1, in_bytes);
  memset(page_out, 1, out_bytes);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    CK(hipMemcpy(d_in, page_in, in_bytes, hipMemcpyHostToDevice));
    double t1 = now();
    CK(hipMemcpy(page_out, d_out, out_bytes, hipMemcpyDeviceToHost));
    double t2 = now();
    std::printf("pageable H2D %.1f GB/s (%.1f ms)   pageable D2H %.1f GB/s (%.1f ms)\n", in_bytes / (t1 - t0) / 1e9,
                (t1 - t0) * 1e3, out_bytes / (t2 - t1) / 1e9, (t2 - t1) * 1e3);
  }
  // pageable, chunked, uploads and downloads from two host threads on two streams
  for (size_t chunk : {(size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
    double t0 = now();
    std::thread up([&] {
      for (size_t o = 0; o < in_bytes; o += chunk)
        (void)hipMemcpyAsync(d_in + o, page_in + o, std::min(chunk, in_bytes - o), hipMemcpyHostToDevice, s1);
      (void)hipStreamSynchronize(s1);
    });
    std::thread down([&] {
      size_t c2 = chunk * 3 / 16;
      for (size_t o = 0; o < out_bytes; o += c2)
        (void)hipMemcpyAsync(page_out + o, d_out + o, std::min(c2, out_bytes - o), hipMemcpyDeviceToHost, s2);
      (void)hipStreamSynchronize(s2);
    });
    up.join();
    down.join();
    double t1 = now();
    std::printf("pageable duplex, chunk %zu MiB: %.1f ms  (%.1f GB/s of input)\n", chunk >> 20, (t1 - t0) * 1e3,
                in_bytes / (t1 - t0) / 1e9);
  }
  // pinned
  char *pin_in, *pin_out;
  double t0 = now();
  CK(hipHostMalloc((void**)&pin_in, in_bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin_out, out_bytes, hipHostMallocDefault));
  double t1 = now();
  std::printf("hipHostMalloc %.2f GB: %.1f ms\n", (in_bytes + out_bytes) / 1e9, (t1 - t0) * 1e3);
  memset(pin_in, 2, in_bytes);
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    CK(hipMemcpyAsync(d_in, pin_in, in_bytes, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    t1 = now();
    CK(hipMemcpyAsync(pin_out, d_out, out_bytes, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s2));
    double t2 = now();
    CK(hipMemcpyAsync(d_in, pin_in, in_bytes, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(pin_out, d_out, out_bytes, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    double t3 = now();
    std::printf("pinned H2D %.1f GB/s (%.1f ms)  D2H %.1f GB/s (%.1f ms)  duplex %.1f ms\n", in_bytes / (t1 - t0) / 1e9,
                (t1 - t0) * 1e3, out_bytes / (t2 - t1) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  // chunked pinned duplex: 10 upload DMAs on s1 and 10 download DMAs on s2 with nothing between them, issued
  // (a) all uploads first, then all downloads; (b) interleaved up/down per chunk; (c) downloads issued from a
  // second host thread while the first issues the uploads
  {
    const int K = 10;
    const size_t cu = in_bytes / K, cd = out_bytes / K;
    for (int variant = 0; variant < 3; ++variant) {
      CK(hipDeviceSynchronize());
      t0 = now();
      if (variant == 0) {
        for (int k = 0; k < K; ++k) CK(hipMemcpyAsync(d_in + k * cu, pin_in + k * cu, cu, hipMemcpyHostToDevice, s1));
        for (int k = 0; k < K; ++k) CK(hipMemcpyAsync(pin_out + k * cd, d_out + k * cd, cd, hipMemcpyDeviceToHost, s2));
      } else if (variant == 1) {
        for (int k = 0; k < K; ++k) {
          CK(hipMemcpyAsync(d_in + k * cu, pin_in + k * cu, cu, hipMemcpyHostToDevice, s1));
          CK(hipMemcpyAsync(pin_out + k * cd, d_out + k * cd, cd, hipMemcpyDeviceToHost, s2));
        }
      } else {
        std::thread dn([&] {
          for (int k = 0; k < K; ++k) (void)hipMemcpyAsync(pin_out + k * cd, d_out + k * cd, cd, hipMemcpyDeviceToHost, s2);
        });
        for (int k = 0; k < K; ++k) CK(hipMemcpyAsync(d_in + k * cu, pin_in + k * cu, cu, hipMemcpyHostToDevice, s1));
        dn.join();
      }
      CK(hipStreamSynchronize(s2));
      double t_down = now();
      CK(hipStreamSynchronize(s1));
      t1 = now();
      std::printf("chunked pinned duplex, variant %d: downloads done at %.1f ms, all done at %.1f ms\n", variant,
                  (t_down - t0) * 1e3, (t1 - t0) * 1e3);
    }
    // the same with each download waiting (stream-side) on an event recorded after the matching upload
    hipEvent_t ev[10];
    for (int k = 0; k < K; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    t0 = now();
    for (int k = 0; k < K; ++k) {
      CK(hipMemcpyAsync(d_in + k * cu, pin_in + k * cu, cu, hipMemcpyHostToDevice, s1));
      CK(hipEventRecord(ev[k], s1));
      CK(hipStreamWaitEvent(s2, ev[k], 0));
      CK(hipMemcpyAsync(pin_out + k * cd, d_out + k * cd, cd, hipMemcpyDeviceToHost, s2));
    }
    CK(hipStreamSynchronize(s2));
    double t_down = now();
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("chunked pinned duplex, download k waits for upload k: downloads done at %.1f ms, all at %.1f ms\n",
                (t_down - t0) * 1e3, (t1 - t0) * 1e3);
  }
  // host-side wait: a second thread waits for upload k (hipEventSynchronize) and then issues download k
  {
    const int K = 10;
    const size_t cu = in_bytes / K, cd = out_bytes / K;
    hipEvent_t ev[10];
    for (int k = 0; k < K; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    t0 = now();
    std::thread dn([&] {
      for (int k = 0; k < K; ++k) {
        while (hipEventQuery(ev[k]) != hipSuccess) {}
        (void)hipMemcpyAsync(pin_out + k * cd, d_out + k * cd, cd, hipMemcpyDeviceToHost, s2);
      }
    });
    // events must have been recorded before they are queried: record them all up front is not possible, so the
    // downloader spins on hipErrorNotReady only after the record below has been issued (K small: issue is fast)
    for (int k = 0; k < K; ++k) {
      CK(hipMemcpyAsync(d_in + k * cu, pin_in + k * cu, cu, hipMemcpyHostToDevice, s1));
      CK(hipEventRecord(ev[k], s1));
    }
    dn.join();
    CK(hipStreamSynchronize(s2));
    double t_down = now();
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("chunked pinned duplex, host thread waits for upload k: downloads done at %.1f ms, all at %.1f ms\n",
                (t_down - t0) * 1e3, (t1 - t0) * 1e3);
  }
  // zero-copy kernels
  {
    const long n_sys = (long)(in_bytes / 128);
    const int bs = 256;
    const unsigned grid = (unsigned)((n_sys + bs - 1) / bs);
    struct { const char* name; const void* in; void* out; } cases[] = {
        {"device in, device out", d_in, d_out},
        {"pinned in, device out", pin_in, d_out},
        {"device in, pinned out", d_in, pin_out},
        {"pinned in, pinned out", pin_in, pin_out},
    };
    for (auto& c : cases)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        t0 = now();
        hipLaunchKernelGGL(zc_kernel, dim3(grid), dim3(bs), 0, s1, (const dbl2*)c.in, (double*)c.out, n_sys, 8);
        CK(hipStreamSynchronize(s1));
        t1 = now();
        if (rep) std::printf("zero-copy kernel, %s: %.2f ms (%.1f GB/s of input)\n", c.name, (t1 - t0) * 1e3,
                             in_bytes / (t1 - t0) / 1e9);
      }
    // chunked: 10 launches over slices of the pinned input, as the pipeline would issue them
    CK(hipDeviceSynchronize());
    t0 = now();
    const int K = 10;
    const long per = n_sys / K;
    for (int k = 0; k < K; ++k) {
      // a slice of a [cam][sys] block is not contiguous: treat each chunk as its own [8][per] block
      const dbl2* in_k = (const dbl2*)pin_in + (size_t)k * per * 8;
      hipLaunchKernelGGL(zc_kernel, dim3((unsigned)((per + bs - 1) / bs)), dim3(bs), 0, s1, in_k,
                         (double*)pin_out + (size_t)k * per * 3, per, 8);
    }
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("zero-copy kernel, pinned in/out, 10 chunk launches: %.2f ms\n", (t1 - t0) * 1e3);
  }
  // host memcpy pageable -> pinned with T threads
  for (int T : {1, 2, 4, 8, 16}) {
    t0 = now();
    par_memcpy(pin_in, page_in, in_bytes, T);
    t1 = now();
    std::printf("memcpy pageable->pinned, %2d threads: %.1f GB/s (%.1f ms)\n", T, in_bytes / (t1 - t0) / 1e9,
                (t1 - t0) * 1e3);
  }
  // register in place
  t0 = now();
  hipError_t e = hipHostRegister(page_in, in_bytes, hipHostRegisterDefault);
  t1 = now();
  std::printf("hipHostRegister %.2f GB: %s, %.1f ms\n", in_bytes / 1e9, hipGetErrorString(e), (t1 - t0) * 1e3);
  if (e == hipSuccess) {
    t0 = now();
    CK(hipMemcpyAsync(d_in, page_in, in_bytes, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("registered H2D %.1f GB/s (%.1f ms)\n", in_bytes / (t1 - t0) / 1e9, (t1 - t0) * 1e3);
    t0 = now();
    CK(hipHostUnregister(page_in));
    t1 = now();
    std::printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
  }
  // staged pipeline: memcpy (T threads) chunk k+1 || DMA chunk k
  for (int T : {2, 4, 8}) {
    const size_t chunk = (size_t)32 << 20;
    hipEvent_t ev[2];
    CK(hipEventCreate(&ev[0]));
    CK(hipEventCreate(&ev[1]));
    t0 = now();
    int k = 0;
    for (size_t o = 0; o < in_bytes; o += chunk, ++k) {
      size_t n = std::min(chunk, in_bytes - o);
      char* stage = pin_in + (size_t)(k & 1) * chunk;
      if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
      par_memcpy(stage, page_in + o, n, T);
      CK(hipMemcpyAsync(d_in + o, stage, n, hipMemcpyHostToDevice, s1));
      CK(hipEventRecord(ev[k & 1], s1));
    }
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("staged pipeline (32 MiB chunks, %d memcpy threads): %.1f ms (%.1f GB/s)\n", T, (t1 - t0) * 1e3,
                in_bytes / (t1 - t0) / 1e9);
  }
  return 0;
}
