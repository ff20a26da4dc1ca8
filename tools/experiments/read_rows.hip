// read_rows.hip — how does a pure nontemporal READ stream on MI355X depend on the number of concurrent rows a wave
// touches?  Same 1.28 GB in every case, cut into R rows (lane i of a block reads 16 B from each of the R rows at the
// same index: R = 8 is exactly the matrix kernel's access pattern at 8 cameras; R = 1 is one contiguous stream).
// Nothing is written (a never-true test keeps the loads alive).  hipcc --offload-arch=gfx950 -O3 read_rows.hip -o read_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int R, int BLOCK>
__global__ void __launch_bounds__(BLOCK) read_rows(const dbl2* __restrict__ src, long row_stride, long n, double* sink) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  dbl2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(src + r * row_stride + i);
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += v[r].x * v[r].y;
  if (s == 1.2345e301) sink[0] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int R, int BLOCK>
double run(const dbl2* d, long total16, double* sink, int iters) {
  const long n = total16 / R;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((read_rows<R, BLOCK>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, d, n, n, sink);
  hipEventRecord(a);
  for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((read_rows<R, BLOCK>), dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, d, n, n, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}
int main() {
  const long total16 = 80000000;  // 16-B elements = 1.28 GB
  dbl2* d; double* sink;
  CK(hipMalloc(&d, total16 * 16));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(d, 0x3f, total16 * 16));
  std::vector<std::pair<const char*, std::vector<double>>> res;
  const char* names[] = {"R1_b64", "R2_b64", "R4_b64", "R8_b64", "R16_b64", "R1_b256", "R8_b256", "R4_b256"};
  std::vector<std::vector<double>> ms(8);
  for (int round = 0; round < 7; ++round) {
    ms[0].push_back(run<1, 64>(d, total16, sink, 20));
    ms[1].push_back(run<2, 64>(d, total16, sink, 20));
    ms[2].push_back(run<4, 64>(d, total16, sink, 20));
    ms[3].push_back(run<8, 64>(d, total16, sink, 20));
    ms[4].push_back(run<16, 64>(d, total16, sink, 20));
    ms[5].push_back(run<1, 256>(d, total16, sink, 20));
    ms[6].push_back(run<8, 256>(d, total16, sink, 20));
    ms[7].push_back(run<4, 256>(d, total16, sink, 20));
  }
  printf("{\"what\": \"pure nontemporal read of 1.28 GB as R rows x 16 B per lane, median of 7 interleaved rounds x 20 launches\"");
  for (int k = 0; k < 8; ++k) {
    std::sort(ms[k].begin(), ms[k].end());
    const double m = ms[k][3];
    printf(", \"%s\": {\"ms\": %.4f, \"TBps\": %.3f}", names[k], m, total16 * 16.0 / (m * 1e-3) / 1e12);
  }
  printf("}\n");
  return 0;
}
