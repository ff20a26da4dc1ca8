// rw_phase.hip — can the matrix kernel's reads and writes be separated in TIME?  rw_mix*.hip: the 1.28 GB read stream
// alone runs at 7.08 TB/s (0.182 ms), the 240 MB of results alone at 6.2 TB/s (0.039 ms), together they take 0.241 ms —
// 9 % more than the sum, the price of mixing reads and writes at the DRAM.  Here a PERSISTENT grid (one wave per
// resident slot) alternates phases without any barrier: every wave reads K tiles (next tile's loads in flight under
// the current one's arithmetic), keeps the K results in registers, then writes them; all waves start together and do
// identical work per phase, so the phases of different waves line up as far as fair bandwidth sharing keeps them.
// hipcc --offload-arch=gfx950 -O3 rw_phase.hip -o rw_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st3(double* p, double a, double b, double c) {
  __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1); __builtin_nontemporal_store(c, p + 2);
}
// baseline: one tile per wave, ordinary grid
__global__ void __launch_bounds__(64) base(const dbl2* __restrict__ src, long n, double* __restrict__ dst) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  dbl2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = __builtin_nontemporal_load(src + r * n + i);
  double a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) { a += v[r].x; b += v[r].y; c += v[r].x * v[r].y; }
  st3(dst + i * 3, a, b, c);
}
// persistent, phased.  G waves; phase p, step k: tile (p * K + k) * G + wave.  WRITE = false: the read side alone.
template <int K, bool WRITE>
__global__ void __launch_bounds__(64) phased(const dbl2* __restrict__ src, long n, double* __restrict__ dst, int phases) {
  const long G = gridDim.x;
  const long lane = threadIdx.x;
  for (int p = 0; p < phases; ++p) {
    double ra[K], rb[K], rc[K];
    dbl2 v[2][8];
    long t = ((long)p * K) * G + blockIdx.x;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[0][r] = __builtin_nontemporal_load(src + r * n + t * 64 + lane);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k + 1 < K) {
        const long t1 = ((long)p * K + k + 1) * G + blockIdx.x;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[(k + 1) & 1][r] = __builtin_nontemporal_load(src + r * n + t1 * 64 + lane);
      }
      double a = 0, b = 0, c = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) { a += v[k & 1][r].x; b += v[k & 1][r].y; c += v[k & 1][r].x * v[k & 1][r].y; }
      ra[k] = a; rb[k] = b; rc[k] = c;
    }
    if (WRITE) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const long tk = ((long)p * K + k) * G + blockIdx.x;
        st3(dst + (tk * 64 + lane) * 3, ra[k], rb[k], rc[k]);
      }
    } else {
      double s = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) s += ra[k] + rb[k] + rc[k];
      if (s == 1.2345e301) dst[0] = s;
    }
  }
}
template <typename F>
double timed(F launch, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) launch();
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters;
}
int main() {
  // n = 10 485 760 systems = 163 840 tiles: divisible by G * K for every (waves per CU, K) below
  const long n = 10485760, tiles = n / 64;
  dbl2* d; double* dst;
  if (hipMalloc(&d, n * 8 * 16) != hipSuccess || hipMalloc(&dst, n * 24 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0x3f, n * 8 * 16);
  struct Case { const char* name; std::vector<double> ms; };
  std::vector<Case> cases;
  auto add = [&](const char* name, double v, size_t idx) { if (cases.size() <= idx) cases.push_back({name, {}}); cases[idx].ms.push_back(v); };
  for (int round = 0; round < 7; ++round) {
    size_t c = 0;
    add("baseline_one_tile_per_wave", timed([&] { hipLaunchKernelGGL(base, dim3((unsigned)tiles), dim3(64), 0, 0, d, n, dst); }, 20), c++);
#define PH(WPC, K, W, NAME) { const long G = 256L * WPC; const int phases = (int)(tiles / (G * K)); \
    add(NAME, timed([&] { hipLaunchKernelGGL((phased<K, W>), dim3((unsigned)G), dim3(64), 0, 0, d, n, dst, phases); }, 20), c++); }
    PH(16, 8, true, "phased_16wpc_K8")
    PH(16, 4, true, "phased_16wpc_K4")
    PH(16, 2, true, "phased_16wpc_K2")
    PH(8, 8, true, "phased_8wpc_K8")
    PH(8, 16, true, "phased_8wpc_K16")
    PH(20, 8, true, "phased_20wpc_K8")
    PH(16, 8, false, "phased_16wpc_K8_reads_only")
    PH(8, 16, false, "phased_8wpc_K16_reads_only")
  }
  printf("{\"what\": \"8 rows x 16 B nontemporal reads + 24 B results per lane, %ld systems; persistent phased grid (wpc = waves per CU, K tiles per phase in registers) against the ordinary one-tile-per-wave grid; median of 7 interleaved rounds x 20 launches, ms\"", n);
  for (auto& cs : cases) { std::sort(cs.ms.begin(), cs.ms.end()); printf(", \"%s\": %.4f", cs.name, cs.ms[3]); }
  printf("}\n");
  return 0;
}
