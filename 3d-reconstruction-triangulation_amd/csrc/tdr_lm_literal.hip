// tdr_lm_literal.hip — the LM kernels with the reference's central-difference Jacobian taken LITERALLY for every ray
// (src/RayTriangulator.cpp:28-44: six distance evaluations, eps = 1e-4), knob TDR_LM_LITERAL_FD.
//
// The default kernels evaluate that form only for rays closer than 0.1 mm and the exact gradient elsewhere (the two
// differ by <= 5e-9 / r[mm]^2 relative, tdr_lm_device.h: lm_camera_term).  This build forces the 0.1 mm test true
// (template parameter LIT), so that the one measured directional residue of the status words (two-ray systems,
// profiles/r04_lm_direction.json) can be attributed ON THE DEVICE, not only on the CPU oracle.
// A diagnostic: same entry points, same schedule, about 1.5-2 x the time; never the default.
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
