// tdr_lm_kernels.hip — the LM kernels of the ray path (RayTriangulator::triangulatePoints,
// src/RayTriangulator.cpp:51-107): instantiations of tdr_lm_device.h for every camera-count class, the convergence
// certificate, and their launchers.  A separate file so that it compiles beside tdr_kernels.hip (tdr_device_common.h).
#include "tdr_lm_device.h"

namespace tdr_amd {

namespace {

// ------------------------------------------------------------------------------------------
// Convergence certificate of the LM results (TDR_LM_NOT_CONVERGED, include/tdr_amd.h), a separate pass that
// runs only when the caller asked for the status words.  The solver's objective sum_i |d_i x (x - o_i)|^2 is
// an exact quadratic, so the distance of the returned point from its minimiser is x - x* = H^-1 g with
//   H = sum_i (|d_i|^2 I - d_i d_i^T),   g = sum_i (|d_i|^2 I - d_i d_i^T)(x - o_i)
// — one streaming pass over the detections (n_cam x 16 B + 24 B + 4 B per system), one lane per system.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    lm_status_kernel(const RayCam* __restrict__ cams, int n_cam, const double2* __restrict__ pts,
                     int64_t cam_stride, int64_t n_sys, const double* __restrict__ xyz,
                     int* __restrict__ status) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_sys) return;
  int st = status[s];
  if (st & kLmTooFew) return;
  const double x = xyz[3 * s], y = xyz[3 * s + 1], z = xyz[3 * s + 2];
  double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0;
  for (int c = 0; c < n_cam; ++c) {
    const double2 uv = pts[(int64_t)c * cam_stride + s];
    if (is_missing(uv.x, uv.y)) continue;
    const Ray r = make_ray(cams[c], uv.x, uv.y);
    const double dd = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const double wx = x - r.ox, wy = y - r.oy, wz = z - r.oz;
    const double dw = r.dx * wx + r.dy * wy + r.dz * wz;
    h00 += dd - r.dx * r.dx; h01 -= r.dx * r.dy; h02 -= r.dx * r.dz;
    h11 += dd - r.dy * r.dy; h12 -= r.dy * r.dz; h22 += dd - r.dz * r.dz;
    g0 += dd * wx - r.dx * dw; g1 += dd * wy - r.dy * dw; g2 += dd * wz - r.dz * dw;
  }
  double e0, e1, e2;
  const bool ok = solve_spd3(h00, h01, h02, h11, h12, h22, g0, g1, g2, e0, e1, e2);
  const double dist = fmax(fabs(e0), fmax(fabs(e1), fabs(e2)));
  const double xs = fmax(fabs(x - e0), fmax(fabs(y - e1), fabs(z - e2)));
  // NaN-safe: anything that is not provably within the tolerance is flagged (parallel rays: H singular)
  if (!(ok && dist <= kLmConvRel * fmax(xs, 1.0))) st |= kLmNotConverged;
  status[s] = st;
}

}  // namespace

hipError_t launch_lm_status(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride, int64_t n_sys,
                            const double* d_xyz, int* d_status, hipStream_t st) {
  hipLaunchKernelGGL(lm_status_kernel, dim3((unsigned)((n_sys + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                     cams.d_ray, n_cam_used, reinterpret_cast<const double2*>(d_pts), cam_stride, n_sys, d_xyz,
                     d_status);
  return hipGetLastError();
}

// the same dispatch over the literal central-difference build (tdr_lm_literal.hip)
hipError_t launch_ray_literal(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride,
                              int64_t n_sys, double* d_xyz, double* d_err, int* d_iters, int* d_status,
                              unsigned long long* fb, hipStream_t st);

hipError_t launch_ray(const DeviceCams& cams, const double* d_pts, int n_cam_used,
                      int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                      int* d_iters, int* d_status, unsigned long long* fb, hipStream_t st) {
  if (n_sys == 0) return hipSuccess;
  // TDR_LM_LITERAL_FD = 1 (diagnostic, tdr_debug_set_knob): every ray takes the reference's six-square-root central
  // difference (src/RayTriangulator.cpp:28-44) instead of the exact gradient above 0.1 mm — a separate build of the
  // same kernels, so the default ones are not touched by the switch
  if (knob(K_LM_LITERAL_FD, cams.lm_literal.load(std::memory_order_relaxed) ? 1 : 0, 0, 1))
    return launch_ray_literal(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_iters, d_status, fb, st);
  hipError_t e = hipErrorInvalidValue;
#define TDR_CASE(N)                                                                                       \
  if (e == hipErrorInvalidValue && n_cam_used <= N)                                                       \
    e = launch_ray_t<N>(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_iters, d_status, fb, st);
  TDR_CASE(2) TDR_CASE(3) TDR_CASE(4) TDR_CASE(6) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16)
  TDR_CASE(24) TDR_CASE(32)
#undef TDR_CASE
  if (e != hipSuccess || !d_status) return e;
  return launch_lm_status(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_status, st);
}

// One system through the camera-per-lane kernel (ray_coop_kernel: lane k evaluates observation k, the group runs the
// LM update redundantly): the per-observation constants travel in the kernel argument (CamR<N> built on the host from
// the CamPointPair's own cameras), the pixels are read where the caller put them (d_uv: a [cam][1] block — the C ABI
// hands in the device address of its pinned slot, so nothing is copied), point and error go to d_out4.  Same
// arithmetic in the same order as the batch kernels, hence the batch path's bits; a pass costs one camera term instead
// of the 32 masked ones of the generic kernel below (round 5: 122 -> see profiles/r05_single_call.json).
// Up to 16 observations; hipErrorInvalidValue beyond (the caller falls back to launch_ray_point).
hipError_t launch_ray_point_coop(const DeviceCams& dc, const RayCam* rc_host, const double* d_uv, int n_obs,
                                 double* d_out4, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_obs < 1 || n_obs > kCoopMaxCam) return hipErrorInvalidValue;
#define TDR_CASE(N)                                                                                                   \
  if (n_obs <= N) {                                                                                                   \
    CamR<N> cams;                                                                                                     \
    for (int c = 0; c < N; ++c) cams.c[c] = rc_host[c < n_obs ? c : 0];                                               \
    hipLaunchKernelGGL((ray_coop_kernel<N, false>), dim3(1), dim3(kWave), 0, stream, cams, n_obs,                     \
                       reinterpret_cast<const double2*>(d_uv), (int64_t)1, (int64_t)1, d_out4, d_out4 + 3,            \
                       (int*)nullptr, (int*)nullptr, dc.d_first_bad_scratch);                                         \
    return hipGetLastError();                                                                                         \
  }
  TDR_CASE(2) TDR_CASE(3) TDR_CASE(4) TDR_CASE(6) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16)
#undef TDR_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_ray_point(const DeviceCams& dc, const double* d_auxobs, const double* d_uv,
                            int n_obs, double* d_out4, hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  // d_auxobs already holds RayCam records (converted on the host by the C ABI); uv[n_obs][2] is a
  // [cam][sys] detection block with one system, so the batch kernel runs it unchanged.
  unsigned long long* counter = dc.d_work_counter + (dc.lm_launches.fetch_add(1) % kLmCounters) * kLmCounterWords;
  hipError_t e = hipMemsetAsync(counter, 0, kLmCounterWords * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  CamRDyn cams{reinterpret_cast<const RayCam*>(d_auxobs)};
  hipLaunchKernelGGL((ray_kernel<kMaxCam, CamRDyn, false, false, 0>), dim3(1), dim3(kWave), 0, stream, cams, n_obs,
                     reinterpret_cast<const double2*>(d_uv), (int64_t)1, (int64_t)1, d_out4,
                     d_out4 + 3, (int*)nullptr, (int*)nullptr, dc.d_first_bad_scratch, counter, 1, 0, 0);
  return hipGetLastError();
}

}  // namespace tdr_amd
