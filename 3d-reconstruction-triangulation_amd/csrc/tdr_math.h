// tdr_math.h — small fp64 building blocks shared by the HIP kernels and (for the scalar
// static helper getDistFromRay) the host adapter.  Product code; independent of oracle/.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define TDR_HD __host__ __device__ __forceinline__
#else
#define TDR_HD inline
#endif

namespace tdr_amd {

struct Ray {
  double ox, oy, oz;
  double dx, dy, dz;
};

// Per-camera constants of the ray model that do not depend on the pixel, precomputed once on the
// host (tdr_create): aux = [qw qx qy qz  tx ty tz  fovy width height].
//   d      = 1 / tan(fovy * 0.0174533 / 2)                 (src/Triangulator.cpp:34)
//   aspect = width / height                                  (src/Triangulator.cpp:37)
struct RayCam {
  double qw, qx, qy, qz;
  double ox, oy, oz;
  double d, aspect, width, height;
};

// Triangulator::calculateRayDirectionForPixel + rotatePointByQuaternion + createRayForPoint
// (src/Triangulator.cpp:15-55).  The quaternion is used as stored (never normalised).
TDR_HD Ray make_ray(const RayCam& c, double u, double v) {
  double px = u + 0.5, py = v + 0.5;
  double x = c.aspect * ((2 * px / c.width) - 1);
  double y = (2 * py / c.height) - 1;
  double inv = 1.0 / sqrt(x * x + y * y + c.d * c.d);
  double vx = x * inv, vy = y * inv, vz = c.d * inv;
  // t = q * (0, v)
  double tw = -c.qx * vx - c.qy * vy - c.qz * vz;
  double tx = c.qw * vx + c.qy * vz - c.qz * vy;
  double ty = c.qw * vy - c.qx * vz + c.qz * vx;
  double tz = c.qw * vz + c.qx * vy - c.qy * vx;
  // r = t * conj(q)
  Ray r;
  r.dx = -tw * c.qx + tx * c.qw - ty * c.qz + tz * c.qy;
  r.dy = -tw * c.qy + tx * c.qz + ty * c.qw - tz * c.qx;
  r.dz = -tw * c.qz - tx * c.qy + ty * c.qx + tz * c.qw;
  r.ox = c.ox;
  r.oy = c.oy;
  r.oz = c.oz;
  return r;
}

// The two halves of make_ray() on their own, for the host-side helpers of the plugin surface
// (tdr_ray_direction_for_pixel / tdr_rotate_point_by_quaternion, include/tdr_amd.h): the same expressions in the same
// order, so make_ray(c, u, v).d == rotate_by_quaternion(q, pixel_direction(c, u, v)) bit for bit
// (tests/test_capi_cpu.py).
// Triangulator::calculateRayDirectionForPixel (src/Triangulator.cpp:27-44)
TDR_HD void pixel_direction(const RayCam& c, double u, double v, double& vx, double& vy, double& vz) {
  double px = u + 0.5, py = v + 0.5;
  double x = c.aspect * ((2 * px / c.width) - 1);
  double y = (2 * py / c.height) - 1;
  double inv = 1.0 / sqrt(x * x + y * y + c.d * c.d);
  vx = x * inv;
  vy = y * inv;
  vz = c.d * inv;
}
// Triangulator::rotatePointByQuaternion (src/Triangulator.cpp:15-25): q (0, v) conj(q), q as stored (w first)
TDR_HD void rotate_by_quaternion(double qw, double qx, double qy, double qz, double vx, double vy, double vz,
                                 double& rx, double& ry, double& rz) {
  double tw = -qx * vx - qy * vy - qz * vz;
  double tx = qw * vx + qy * vz - qz * vy;
  double ty = qw * vy - qx * vz + qz * vx;
  double tz = qw * vz + qx * vy - qy * vx;
  rx = -tw * qx + tx * qw - ty * qz + tz * qy;
  ry = -tw * qy + tx * qz + ty * qw - tz * qx;
  rz = -tw * qz - tx * qy + ty * qx + tz * qw;
}

// Triangulator::distToRay, squared form (src/Triangulator.cpp:3-9)
TDR_HD double dist2_to_ray(const Ray& r, double px, double py, double pz) {
  double wx = px - r.ox, wy = py - r.oy, wz = pz - r.oz;
  double cx = r.dy * wz - r.dz * wy;
  double cy = r.dz * wx - r.dx * wz;
  double cz = r.dx * wy - r.dy * wx;
  return cx * cx + cy * cy + cz * cz;
}
TDR_HD double dist_to_ray(const Ray& r, double px, double py, double pz) {
  return sqrt(dist2_to_ray(r, px, py, pz));
}

// Symmetric positive-definite 3x3 solve M x = v by LDL^T (M given by its upper triangle).
// Returns false when a pivot falls below kRankTol of the diagonal entry it was eliminated from: the normal
// equations M = A^T A then carry a relative error of about DBL_EPSILON / kRankTol (2e-8) or worse — or have
// no solution at all (rank-deficient geometry) — and the caller redoes the solve on A itself
// (pinv_minimum_norm, tdr_kernels.hip), which is what the reference's SVD pseudo-inverse does for every system.
constexpr double kRankTol = 1e-8;
// 1 / b for the three pivots.  On the device: v_rcp_f64 + two Newton steps (<= 1 ulp, 5 instructions) instead of the
// IEEE division expansion (11): the int16 format's kernel is VALU-bound, and the three divisions were 5 % of its
// instructions.  A zero pivot gives NaN instead of inf; either way the rank test below fails and the caller redoes the
// solve (pinv_minimum_norm).
TDR_HD double spd3_rcp(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(b);
  y = fma(fma(-b, y, 1.0), y, y);
  y = fma(fma(-b, y, 1.0), y, y);
  return y;
#else
  return 1.0 / b;
#endif
}
TDR_HD bool solve_spd3(double m00, double m01, double m02, double m11, double m12, double m22,
                       double v0, double v1, double v2, double& x0, double& x1, double& x2) {
  double i0 = spd3_rcp(m00);
  double l10 = m01 * i0, l20 = m02 * i0;
  double d1 = m11 - l10 * m01;
  double t12 = m12 - l20 * m01;
  double i1 = spd3_rcp(d1);
  double l21 = t12 * i1;
  double d2 = m22 - l20 * m02 - l21 * t12;
  double i2 = spd3_rcp(d2);
  double y0 = v0;
  double y1 = v1 - l10 * y0;
  double y2 = v2 - l20 * y0 - l21 * y1;
  x2 = y2 * i2;
  x1 = y1 * i1 - l21 * x2;
  x0 = y0 * i0 - l10 * x1 - l20 * x2;
  // (the first pivot has nothing it was eliminated from: it is measured against the other two diagonal entries,
  // all three being squared column norms of A in the same unit)
  return (m00 > kRankTol * (m11 + m22)) && (d1 > kRankTol * m11) && (d2 > kRankTol * m22);
}

// Inverse of a symmetric 3x3 by LDL^T-free adjugate; used for the diag(A^-1) step of LM.
TDR_HD void inv_sym3_diag(double m00, double m01, double m02, double m11, double m12, double m22,
                          double& i00, double& i11, double& i22) {
  // explicit fma: inlined into several kernels that must agree bit for bit
  double c00 = fma(m11, m22, -(m12 * m12));
  double c01 = fma(m02, m12, -(m01 * m22));
  double c02 = fma(m01, m12, -(m02 * m11));
  double det = fma(m00, c00, fma(m01, c01, m02 * c02));
  double id = 1.0 / det;
  i00 = c00 * id;
  i11 = fma(m00, m22, -(m02 * m02)) * id;
  i22 = fma(m00, m11, -(m01 * m01)) * id;
}

}  // namespace tdr_amd
