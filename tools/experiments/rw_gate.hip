// rw_gate.hip — chip-wide write phases WITHOUT a barrier: every wave holds its three result stores until the device-wide
// constant clock (s_memrealtime, 100 MHz, the same counter on every XCD) is inside a window that recurs every PERIOD
// ticks and stays open for OPEN ticks.  rw_mix*.hip: a write stream costs the read stream ~0.024 ms for existing at all
// (the DRAM drains its write queues about every 0.4 us whatever they hold); if all writes of the chip arrive in one
// short window per period, the drains outside the window have nothing to do.  The price: parked waves hold their
// registers (the kernel's plateau runs from 16 to 24 waves per CU, so some can park for free).
// hipcc --offload-arch=gfx950 -O3 rw_gate.hip -o rw_gate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int PERIOD, int OPEN, bool READS>
__global__ void __launch_bounds__(64) rw(const dbl2* __restrict__ src, long n, double* __restrict__ dst) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  double a = (double)i, b = 1, c = 2;
  if (READS) {
    dbl2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = __builtin_nontemporal_load(src + r * n + i);
    a = b = c = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a += v[r].x; b += v[r].y; c += v[r].x * v[r].y; }
  }
  if (PERIOD > 0) {
    // wave-uniform wait (scalar clock reads); the data dependence on `a` keeps it behind the loads
    asm volatile("" ::"v"(a));
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    for (int spin = 0; spin < 2048 && (int)(t % (unsigned)PERIOD) >= OPEN; ++spin) {  // bounded: every wave gets out
      __builtin_amdgcn_s_sleep(2);
      t = __builtin_amdgcn_s_memrealtime();
    }
  }
  __builtin_nontemporal_store(a, dst + i * 3);
  __builtin_nontemporal_store(b, dst + i * 3 + 1);
  __builtin_nontemporal_store(c, dst + i * 3 + 2);
}
template <int PERIOD, int OPEN, bool READS>
double run(const dbl2* d, long n, double* dst, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const dim3 grid((unsigned)(n / 64));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((rw<PERIOD, OPEN, READS>), grid, dim3(64), 0, 0, d, n, dst);
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((rw<PERIOD, OPEN, READS>), grid, dim3(64), 0, 0, d, n, dst);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters;
}
int main() {
  const long n = 10000000 / 1024 * 1024;
  dbl2* d; double* dst;
  if (hipMalloc(&d, n * 8 * 16) != hipSuccess || hipMalloc(&dst, n * 24 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0x3f, n * 8 * 16);
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  struct Case { const char* name; std::vector<double> ms; };
  std::vector<Case> cases;
  for (int round = 0; round < 7; ++round) {
    size_t c = 0;
#define RUN(P, O, NAME) { if (cases.size() <= c) cases.push_back({NAME, {}}); cases[c++].ms.push_back(run<P, O, true>(d, n, dst, 20)); }
    RUN(0, 0, "ungated")
    RUN(200, 50, "period_2us_open_0.5us")
    RUN(200, 100, "period_2us_open_1us")
    RUN(400, 50, "period_4us_open_0.5us")
    RUN(400, 100, "period_4us_open_1us")
    RUN(400, 200, "period_4us_open_2us")
    RUN(800, 100, "period_8us_open_1us")
    RUN(800, 200, "period_8us_open_2us")
    RUN(800, 400, "period_8us_open_4us")
    RUN(1600, 400, "period_16us_open_4us")
    RUN(100, 25, "period_1us_open_0.25us")
    RUN(100, 50, "period_1us_open_0.5us")
  }
  printf("{\"what\": \"8 rows x 16 B nontemporal reads + three nontemporal 8-B stores per lane (%ld lanes), the stores gated by the device-wide 100 MHz clock (wall clock rate reported: %d kHz); median of 7 interleaved rounds x 20 launches, ms\"", n, rate_khz);
  for (auto& cs : cases) { std::sort(cs.ms.begin(), cs.ms.end()); printf(", \"%s\": %.4f", cs.name, cs.ms[3]); }
  printf("}\n");
  return 0;
}
