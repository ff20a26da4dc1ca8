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
