// tdr_lm_literal.hip — the LM kernels with RayClosestPoint::compute taken LITERALLY for every ray
// (src/RayTriangulator.cpp:19-44), knob TDR_LM_LITERAL_FD:
//   * the central-difference Jacobian from SIX distance evaluations at the shifted points (eps = 1e-4), each distance as
//     Triangulator::distToRay computes it without FMA (rounded products, plain sums, IEEE square root) — including the
//     rounding NOISE of that difference (~2.5e-9 per Jacobian entry), which the linearity form of the default kernels'
//     < 0.1 mm branch does not have;
//   * r_i = the distance and S = sum r_i * r_i (square root, then square), as cv::LMSolver forms them;
//   * (nearly) rank-deficient solves through the restated Jacobi sweeps of DECOMP_EIG (round 3's form) instead of the
//     closed-form rank cascade.
// The default kernels evaluate the exact gradient (and the linearity form of the central difference below 0.1 mm).
// This build exists to ATTRIBUTE the one measured directional residue of the status words on the device (VERDICT r4
// item 2; profiles/r05_lm_literal_fd.json): -6.8e-5 per two-ray system against the oracle with the default build.
// A diagnostic: same entry points, same schedule, 2-4 x the time; never the default.
#include "tdr_lm_device.h"

namespace tdr_amd {

hipError_t launch_lm_status(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride, int64_t n_sys,
                            const double* d_xyz, int* d_status, hipStream_t st);

hipError_t launch_ray_literal(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride,
                              int64_t n_sys, double* d_xyz, double* d_err, int* d_iters, int* d_status,
                              unsigned long long* fb, hipStream_t st) {
  hipError_t e = hipErrorInvalidValue;
#define TDR_CASE(N)                                                 \
  if (e == hipErrorInvalidValue && n_cam_used <= N)                 \
    e = launch_ray_t<N, true>(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_iters, d_status, fb, st);
  TDR_CASE(2) TDR_CASE(3) TDR_CASE(4) TDR_CASE(6) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16)
  TDR_CASE(24) TDR_CASE(32)
#undef TDR_CASE
  if (e != hipSuccess || !d_status) return e;
  return launch_lm_status(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_status, st);
}

}  // namespace tdr_amd
