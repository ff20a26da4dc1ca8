// rw_mix.hip — what does the WRITE side of the matrix kernel's traffic cost on MI355X?  Every case reads the same 1.28 GB
// as 8 rows x 16 B per lane (nontemporal, 64-thread blocks: the matrix kernel's pattern at C3, pure read = 7.0 TB/s) and
// adds one store form.  hipcc --offload-arch=gfx950 -O3 rw_mix.hip -o rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
enum { NONE, ST8_NT, ST8, ST24_NT, ST24, ST16_NT, ST24_NT_SMALL, ST24_STAGED_NT, ST32_NT, N_MODES };
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) rw(const dbl2* __restrict__ src, long n, double* __restrict__ dst, long wrap) {
  __shared__ double lds[MODE == ST24_STAGED_NT ? BLOCK * 3 : 1];
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;  // n is a multiple of BLOCK in main()
  dbl2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = __builtin_nontemporal_load(src + r * n + i);
  double a = 0, b = 0, c = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) { a += v[r].x; b += v[r].y; c += v[r].x * v[r].y; }
  if (MODE == NONE) { if (a == 1.2345e301) dst[0] = a; }
  if (MODE == ST8_NT) __builtin_nontemporal_store(a, dst + i);
  if (MODE == ST8) dst[i] = a;
  if (MODE == ST24_NT || MODE == ST24_NT_SMALL) {
    const long o = MODE == ST24_NT_SMALL ? (i % wrap) * 3 : i * 3;
    __builtin_nontemporal_store(a, dst + o); __builtin_nontemporal_store(b, dst + o + 1); __builtin_nontemporal_store(c, dst + o + 2);
  }
  if (MODE == ST24) { dst[i * 3] = a; dst[i * 3 + 1] = b; dst[i * 3 + 2] = c; }
  if (MODE == ST16_NT) { dbl2 t; t.x = a; t.y = b; __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(dst) + i); }
  if (MODE == ST32_NT) { dbl2 t; t.x = a; t.y = b; dbl2 u; u.x = c; u.y = a; __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(dst) + 2 * i); __builtin_nontemporal_store(u, reinterpret_cast<dbl2*>(dst) + 2 * i + 1); }
  if (MODE == ST24_STAGED_NT) {  // the block's BLOCK*24 contiguous bytes leave as 16-B-per-lane stores
    lds[threadIdx.x * 3] = a; lds[threadIdx.x * 3 + 1] = b; lds[threadIdx.x * 3 + 2] = c;
    __syncthreads();
    dbl2* out = reinterpret_cast<dbl2*>(dst + (long)blockIdx.x * BLOCK * 3);
    for (int k = threadIdx.x; k < BLOCK * 3 / 2; k += BLOCK) { dbl2 t; t.x = lds[2 * k]; t.y = lds[2 * k + 1]; __builtin_nontemporal_store(t, out + k); }
  }
}
template <int MODE, int BLOCK>
double run(const dbl2* d, long n, double* dst, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const dim3 grid((unsigned)(n / BLOCK));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((rw<MODE, BLOCK>), grid, dim3(BLOCK), 0, 0, d, n, dst, 32768L);
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((rw<MODE, BLOCK>), grid, dim3(BLOCK), 0, 0, d, n, dst, 32768L);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters;
}
int main() {
  const long n = 10000000 / 256 * 256;
  dbl2* d; double* dst;
  if (hipMalloc(&d, n * 8 * 16) != hipSuccess || hipMalloc(&dst, n * 32 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0x3f, n * 8 * 16);
  const char* names[N_MODES] = {"none", "st8_nt", "st8", "st24_nt", "st24", "st16_nt", "st24_nt_into_768KB", "st24_staged16_nt", "st32_nt"};
  const double wbytes[N_MODES] = {0, 8, 8, 24, 24, 16, 24, 24, 32};
  std::vector<std::vector<double>> ms(N_MODES);
  for (int round = 0; round < 7; ++round) {
    ms[NONE].push_back(run<NONE, 64>(d, n, dst, 20));
    ms[ST8_NT].push_back(run<ST8_NT, 64>(d, n, dst, 20));
    ms[ST8].push_back(run<ST8, 64>(d, n, dst, 20));
    ms[ST24_NT].push_back(run<ST24_NT, 64>(d, n, dst, 20));
    ms[ST24].push_back(run<ST24, 64>(d, n, dst, 20));
    ms[ST16_NT].push_back(run<ST16_NT, 64>(d, n, dst, 20));
    ms[ST24_NT_SMALL].push_back(run<ST24_NT_SMALL, 64>(d, n, dst, 20));
    ms[ST24_STAGED_NT].push_back(run<ST24_STAGED_NT, 256>(d, n, dst, 20));
    ms[ST32_NT].push_back(run<ST32_NT, 64>(d, n, dst, 20));
  }
  printf("{\"what\": \"8 rows x 16 B nontemporal reads per lane (1.28 GB) + one store form, 64-thread blocks (staged: 256), median of 7 interleaved rounds x 20 launches\"");
  for (int k = 0; k < N_MODES; ++k) {
    std::sort(ms[k].begin(), ms[k].end());
    const double m = ms[k][3];
    printf(", \"%s\": {\"ms\": %.4f, \"TBps_total\": %.3f, \"write_B_per_lane\": %.0f}", names[k], m, n * (128.0 + wbytes[k]) / (m * 1e-3) / 1e12, wbytes[k]);
  }
  printf("}\n");
  return 0;
}
