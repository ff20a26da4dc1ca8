// rw_mix2.hip — second look at the store side (see rw_mix.hip): where does the ~0.024 ms that ANY store adds to the
// 0.181 ms read stream come from — the store path / wave lifetime, or DRAM read<->write turnarounds?
// hipcc --offload-arch=gfx950 -O3 rw_mix2.hip -o rw_mix2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
enum { NONE, ST24_NT, ST24_WRAP16M, ST24_WRAP768K, ST24_NT_WRAP64M, DUAL_TILE, HALF_WAVES_48, QUARTER_WAVES_96, ST24_NT_B256, DUAL_TILE_B128, N_MODES };
__device__ __forceinline__ void st3(double* p, double a, double b, double c, bool nt) {
  if (nt) { __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1); __builtin_nontemporal_store(c, p + 2); }
  else { p[0] = a; p[1] = b; p[2] = c; }
}
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) rw(const dbl2* __restrict__ src, long n, double* __restrict__ dst) {
  if (MODE == DUAL_TILE || MODE == DUAL_TILE_B128) {  // a block handles two tiles BLOCK lanes apart: 16 loads in flight, then both stores
    const long i = (long)blockIdx.x * (2 * BLOCK) + threadIdx.x;
    dbl2 v[16];
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r] = __builtin_nontemporal_load(src + r * n + i); v[8 + r] = __builtin_nontemporal_load(src + r * n + i + BLOCK); }
    double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a += v[r].x; b += v[r].y; c += v[r].x * v[r].y; d += v[8 + r].x; e += v[8 + r].y; f += v[8 + r].x * v[8 + r].y; }
    st3(dst + i * 3, a, b, c, true);
    st3(dst + (i + BLOCK) * 3, d, e, f, true);
    return;
  }
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  dbl2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = __builtin_nontemporal_load(src + r * n + i);
  double a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) { a += v[r].x; b += v[r].y; c += v[r].x * v[r].y; }
  if (MODE == NONE) { if (a == 1.2345e301) dst[0] = a; }
  if (MODE == ST24_NT || MODE == ST24_NT_B256) st3(dst + i * 3, a, b, c, true);
  if (MODE == ST24_WRAP16M) st3(dst + (i % 699050) * 3, a, b, c, false);
  if (MODE == ST24_WRAP768K) st3(dst + (i % 32768) * 3, a, b, c, false);
  if (MODE == ST24_NT_WRAP64M) st3(dst + (i % 2796202) * 3, a, b, c, true);
  if (MODE == HALF_WAVES_48) {  // the same 240 MB leave from HALF the waves (48 B per lane of every second block)
    if ((blockIdx.x & 1) == 0) { st3(dst + i * 3, a, b, c, true); st3(dst + (i + BLOCK) * 3, b, c, a, true); }
    else if (a == 1.2345e301) dst[0] = a;  // keeps the loads of the non-storing blocks alive (the compiler sank them into the branch)
  }
  if (MODE == QUARTER_WAVES_96) {
    if ((blockIdx.x & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) st3(dst + (i + k * BLOCK) * 3, a + k, b, c, true);
    } else if (a == 1.2345e301) dst[0] = a;
  }
}
// write only: 24 B per lane as three nontemporal 8-B stores / 16 B per lane, nothing read
template <int BYTES>
__global__ void __launch_bounds__(64) wr(double* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  const double a = (double)i;
  if (BYTES == 24) st3(dst + i * 3, a, a + 1, a + 2, true);
  else { dbl2 t; t.x = a; t.y = a + 1; __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(dst) + i); }
}
template <int BYTES>
double run_wr(double* dst, long n, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((wr<BYTES>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, dst, n);
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((wr<BYTES>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, dst, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters;
}
template <int MODE, int BLOCK>
double run(const dbl2* d, long n, double* dst, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long per_block = (MODE == DUAL_TILE || MODE == DUAL_TILE_B128) ? 2 * BLOCK : BLOCK;
  const dim3 grid((unsigned)(n / per_block));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((rw<MODE, BLOCK>), grid, dim3(BLOCK), 0, 0, d, n, dst);
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((rw<MODE, BLOCK>), grid, dim3(BLOCK), 0, 0, d, n, dst);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters;
}
int main() {
  const long n = 10000000 / 1024 * 1024;
  dbl2* d; double* dst;
  if (hipMalloc(&d, n * 8 * 16) != hipSuccess || hipMalloc(&dst, n * 24 + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0x3f, n * 8 * 16);
  const char* names[N_MODES] = {"none", "st24_nt", "st24_regular_wrapped_16MB", "st24_regular_wrapped_768KB", "st24_nt_wrapped_64MB",
                                "dual_tile_16_loads_then_2x_st24_nt", "st48_nt_from_half_the_waves", "st96_nt_from_a_quarter_of_the_waves",
                                "st24_nt_block256", "dual_tile_block128"};
  std::vector<std::vector<double>> ms(N_MODES);
  std::vector<double> w24, w16;
  for (int round = 0; round < 7; ++round) {
    w24.push_back(run_wr<24>(dst, n, 20));
    w16.push_back(run_wr<16>(dst, n * 3 / 2, 20));
    ms[NONE].push_back(run<NONE, 64>(d, n, dst, 20));
    ms[ST24_NT].push_back(run<ST24_NT, 64>(d, n, dst, 20));
    ms[ST24_WRAP16M].push_back(run<ST24_WRAP16M, 64>(d, n, dst, 20));
    ms[ST24_WRAP768K].push_back(run<ST24_WRAP768K, 64>(d, n, dst, 20));
    ms[ST24_NT_WRAP64M].push_back(run<ST24_NT_WRAP64M, 64>(d, n, dst, 20));
    ms[DUAL_TILE].push_back(run<DUAL_TILE, 64>(d, n, dst, 20));
    ms[HALF_WAVES_48].push_back(run<HALF_WAVES_48, 64>(d, n, dst, 20));
    ms[QUARTER_WAVES_96].push_back(run<QUARTER_WAVES_96, 64>(d, n, dst, 20));
    ms[ST24_NT_B256].push_back(run<ST24_NT_B256, 256>(d, n, dst, 20));
    ms[DUAL_TILE_B128].push_back(run<DUAL_TILE_B128, 128>(d, n, dst, 20));
  }
  printf("{\"what\": \"8 rows x 16 B nontemporal reads per lane (1.28 GB) + one store form; median of 7 interleaved rounds x 20 launches\"");
  for (int k = 0; k < N_MODES; ++k) {
    std::sort(ms[k].begin(), ms[k].end());
    printf(", \"%s\": %.4f", names[k], ms[k][3]);
  }
  std::sort(w24.begin(), w24.end()); std::sort(w16.begin(), w16.end());
  printf(", \"write_only_240MB_as_3x8B_nt\": %.4f, \"write_only_240MB_as_16B_nt\": %.4f", w24[3], w16[3]);
  printf("}\n");
  return 0;
}
