// tdr_lm_device.h — device code of the ray path's Levenberg-Marquardt kernels (templates; instantiated by
// tdr_lm_kernels.hip and tdr_lm_literal.hip) and the small fp64 building blocks the other kernels borrow from it
// (sqrt_pos, mul_rn, rcp_nr, lm_cascade, sym3_eig_pinv).  Internal; see tdr_device_common.h for the file layout.
#pragma once
#include "tdr_device_common.h"

namespace tdr_amd {
namespace {

// ------------------------------------------------------------------------------------------
// Ray path: Levenberg-Marquardt on r_i = |dir_i x (x - origin_i)| (src/RayTriangulator.cpp:8-47, :83-107) with the
// update schedule of cv::LMSolver::run as created by LMSolver::create(cb, 1000): epsx = epsf = FLT_EPSILON,
// lambda0 = 1, lc = 0.75, Rlo = .25, Rhi = .75.
//
// One system per lane.  Camera constants are wave-uniform (LDS broadcast reads); per-lane state is the 3
// direction doubles per camera, a validity bitmask and the 3x3 LM state.
//
// Structural choices:
//  * The solver consumes S = sum r^2, J^T J, J^T r and max r, never a single residual.  With c = d x w and
//    p = c x d the exact gradient of the distance is p / r, so a ray adds p p^T / r^2, p and |c|^2: no square
//    root in the loop (lm_camera_term).  The reference's central difference (eps = 1e-4) differs from the exact
//    gradient by <= 5e-9 / r[mm]^2 relative and is evaluated literally only for rays closer than 0.1 mm, through
//    the linearity of the cross product: dist(x +- eps e_k)^2 = |c|^2 +- 2 eps p_k + eps^2 |d x e_k|^2.
//  * Lane refill.  LM iteration counts vary 15..200+ between systems; a wave that simply loops until
//    its slowest lane stops idles half its lanes.  Each wave instead runs a state machine and, once
//    kRefill lanes are idle, pulls that many new systems from a global work counter.
//  * Speculative passes: every busy lane evaluates S and the normal equations at its trial point in one fused
//    pass; a rejected step wastes its J^T J (the schedule alternates accept / reject, lanes in random phase).
//  * Cooperative drain once the queue is empty and few systems are left (lm_coop_finish).
// ------------------------------------------------------------------------------------------
constexpr double kFdEps = 1e-4;    // THRESHOLD, src/RayTriangulator.h:9
constexpr int kMaxLmIters = 1000;  // MAX_ITERATIONS, src/RayTriangulator.h:10
constexpr int kLmCap = 1, kLmNotConverged = 2, kLmTooFew = 4;  // TDR_LM_* (include/tdr_amd.h)
constexpr double kLmConvRel = 5e-8;                            // TDR_LM_CONVERGED_REL
constexpr double kFltEps = 1.1920928955078125e-07;
constexpr int kRefillDefault = 8;   // idle lanes that trigger a refill (sweep: 8 best of 1..64)
constexpr int kLmChunk = 32;        // systems claimed per atomic while the batch is far from its end
#ifndef TDR_LM_WAVES
#define TDR_LM_WAVES 1  // min waves per SIMD the LM kernel is register-allocated for (A/B knob)
#endif

// sqrt for the LM loop: v_rsq_f64 seed (~23 bits) + one Goldschmidt step (~45 bits) + one residual
// correction (faithfully rounded: <= 1 ulp), i.e. the compiler's own fp64 sqrt expansion without its range
// scaling / class handling and without the second correction that only buys correct rounding (the operands
// here are squared distances in mm^2, far from the denormal and overflow ranges).  The seed is taken from
// max(a, tiny): for a == 0 every product below is then an exact 0 and no select is needed.
constexpr double kRsqFloor = 1e-300;
#ifndef TDR_LM_FUSED_CROSS
#define TDR_LM_FUSED_CROSS 0  // A/B: 1 = round-2 form of the distance cross product (see ray_dist2)
#endif
#ifndef TDR_LM_INTERLEAVE
#define TDR_LM_INTERLEAVE 2
#endif
constexpr int kLmInterleave = TDR_LM_INTERLEAVE;  // cameras whose evaluation chains may overlap (A/B knob)

__device__ __forceinline__ double sqrt_pos(double a) {
  const double y = __builtin_amdgcn_rsq(fmax(a, kRsqFloor));
  double g = a * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, a);
  return fma(d, h, g);
}

// sqrt(a) and 1/sqrt(a) together (same seed; one extra Newton step for the reciprocal)
__device__ __forceinline__ void sqrt_rsqrt_pos(double a, double& s, double& rs) {
  const double y = __builtin_amdgcn_rsq(fmax(a, kRsqFloor));
  double g = a * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, a);
  g = fma(d, h, g);
  double inv = h + h;
  inv = fma(inv, fma(-g, inv, 1.0), inv);
  s = g;
  rs = inv;  // ~1e150 when a == 0 (s == 0): callers test before use
}

// A product that is rounded on its own.  The library is built with -ffp-contract=fast, under which the backend fuses
// a * b - c whatever the source says (neither __dmul_rn nor `#pragma clang fp contract(off)` survives it): the
// product goes through an asm the optimiser cannot look into.
__device__ __forceinline__ double mul_rn(double a, double b) {
  double r;
  asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// v_max_f64 on operands known not to be NaN: fmax() makes the compiler canonicalise both inputs first
// (two extra v_max_f64 per use in the per-camera loop)
__device__ __forceinline__ double max_nn(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

typedef __attribute__((address_space(3))) volatile double lds_vdouble;

// Ray directions of one system (one lane).  The first NCAM - NLDS cameras live in VGPRs, the last NLDS in the
// block's LDS array [NLDS * 3][64] (this lane's column: conflict-free 8-B accesses), read through a volatile
// pointer like the camera constants so the loads stay next to their use.  Only the 24- and 32-camera kernels
// use it: their 144 / 192 direction VGPRs otherwise push the kernel to one wave per SIMD plus AGPR copies.
template <int NCAM, int NLDS = 0>
struct RaySet {
  static constexpr int NREG = NCAM - NLDS;
  static constexpr int kInLds = NLDS;
  double dx[NREG], dy[NREG], dz[NREG];
  lds_vdouble* l;
  unsigned valid;  // bit c = observation c present
  __device__ __forceinline__ void get(int c, double& x, double& y, double& z) const {
    if (c < NREG) {
      x = dx[c]; y = dy[c]; z = dz[c];
    } else {
      const int k = (c - NREG) * 3;
      x = l[(k + 0) * kWave]; y = l[(k + 1) * kWave]; z = l[(k + 2) * kWave];
    }
  }
  __device__ __forceinline__ void set(int c, double x, double y, double z) {
    if (c < NREG) {
      dx[c] = x; dy[c] = y; dz[c] = z;
    } else {
      const int k = (c - NREG) * 3;
      l[(k + 0) * kWave] = x; l[(k + 1) * kWave] = y; l[(k + 2) * kWave] = z;
    }
  }
};

struct CamRDyn {  // per-observation ray constants in device memory (single-system entry point)
  const RayCam* c;
};

struct LmEval {
  double S, r2max;
};

// |d x w|^2, the squared distance of a point from a ray (w = point - origin); every evaluator goes through
// this one expression so that they round alike
__device__ __forceinline__ double ray_dist2(double dx, double dy, double dz, double wx, double wy, double wz,
                                            double& cx, double& cy, double& cz) {
#if TDR_LM_FUSED_CROSS
  cx = fma(dy, wz, -(dz * wy));
  cy = fma(dz, wx, -(dx * wz));
  cz = fma(dx, wy, -(dy * wx));
#else
  // Both products rounded before the subtraction, as the reference's Point3d::cross compiles for baseline x86-64
  // (src/Triangulator.cpp:3-9; CMakeLists.txt sets no -march, so no FMA).  |w| is metres, |c| the ray distance in
  // millimetres: the subtraction cancels 3-4 digits, and a fused form (one product exact) carries about half the
  // rounding noise in S = sum |c|^2.  That is not neutral: where two nearly parallel rays make the solver creep, the
  // accept test Sd < S ends up comparing values that differ by less than their rounding error, the run stalls where
  // the noise says so, and the quieter S stalls later — 2-ray systems then end a few iterations closer to the
  // minimiser than the reference's, on one side of the TDR_LM_NOT_CONVERGED threshold more often than on the other
  // (tests/tools/lm_solve_form.py reproduces it on the CPU oracle with this one change: 599 against 877 one-sided
  // flags in 240 000 two-ray systems).  Three more VALU instructions per camera.
  cx = mul_rn(dy, wz) - mul_rn(dz, wy);
  cy = mul_rn(dz, wx) - mul_rn(dx, wz);
  cz = mul_rn(dx, wy) - mul_rn(dy, wx);
#endif
  return fma(cz, cz, fma(cy, cy, cx * cx));
}

// residuals only (RayClosestPoint::compute without J, src/RayTriangulator.cpp:19-26)
// Camera constants live in LDS as [n_cam][12] doubles (RayCam + pad) and are read through a volatile
// address_space(3) pointer: they are loop-invariant and wave-uniform, and the compiler otherwise
// hoists them into SGPRs it does not have (the first version spilled them to VGPR lanes: 1244
// v_readlane + 430 v_writelane in the loop).  Broadcast ds_read: every lane reads the same address.
struct LdsCams {
  lds_vdouble* p;  // [n_cam][12]
  __device__ __forceinline__ double ox(int c) const { return p[12 * c + 4]; }
  __device__ __forceinline__ double oy(int c) const { return p[12 * c + 5]; }
  __device__ __forceinline__ double oz(int c) const { return p[12 * c + 6]; }
  __device__ __forceinline__ double dd(int c) const { return p[12 * c + 11]; }
  // Ray direction for pixel (u, v) of camera c — make_ray() (tdr_math.h, src/Triangulator.cpp:15-55) with
  // the two divisions by width / height turned into multiplications by the per-block constants 2/width,
  // 2/height (slots 9, 10) and the normalisation done with the loop's own rsq-based reciprocal root: the
  // refill runs with only a few lanes active, and the IEEE divisions and sqrt were 60 % of its instructions.
  // Differs from make_ray() by <= 2 ulp per component.
  __device__ __forceinline__ void ray_dir(int c, double u, double v, double& dx, double& dy, double& dz) const {
    const double qw = p[12 * c + 0], qx = p[12 * c + 1], qy = p[12 * c + 2], qz = p[12 * c + 3];
    const double d = p[12 * c + 7];
    const double x = p[12 * c + 8] * fma(u + 0.5, p[12 * c + 9], -1.0);
    const double y = fma(v + 0.5, p[12 * c + 10], -1.0);
    double nrm, inv;
    sqrt_rsqrt_pos(fma(x, x, fma(y, y, d * d)), nrm, inv);
    const double vx = x * inv, vy = y * inv, vz = d * inv;
    // explicit fma chains: this function is instantiated in several kernels whose results must agree bit
    // for bit, so nothing is left to the contraction pass
    const double tw = -fma(qx, vx, fma(qy, vy, qz * vz));
    const double tx = fma(qw, vx, fma(qy, vz, -(qz * vy)));
    const double ty = fma(qw, vy, fma(qz, vx, -(qx * vz)));
    const double tz = fma(qw, vz, fma(qx, vy, -(qy * vx)));
    dx = fma(tx, qw, fma(tz, qy, -fma(tw, qx, ty * qz)));
    dy = fma(tx, qz, fma(ty, qw, -fma(tw, qy, tz * qx)));
    dz = fma(ty, qx, fma(tz, qw, -fma(tw, qz, tx * qy)));
  }
};

// one camera's record of the LDS table read by LdsCams
__device__ __forceinline__ void lds_store_camera(double* o, const RayCam& rc) {
  o[0] = rc.qw; o[1] = rc.qx; o[2] = rc.qy; o[3] = rc.qz;
  o[4] = rc.ox; o[5] = rc.oy; o[6] = rc.oz;
  o[7] = rc.d; o[8] = rc.aspect; o[9] = 2.0 / rc.width; o[10] = 2.0 / rc.height;
  const double nq = rc.qw * rc.qw + rc.qx * rc.qx + rc.qy * rc.qy + rc.qz * rc.qz;
  o[11] = nq * nq;  // |direction|^2 of every ray of this camera (unit vector rotated by the raw quaternion)
}

template <int NCAM, typename RS>
__device__ __forceinline__ LmEval lm_residuals(LdsCams org, const RS& rs,
                                               int n_cam, double x, double y, double z) {
  LmEval e{0, 0};
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam && ((rs.valid >> c) & 1u)) {
      double dx, dy, dz, cx, cy, cz;
      rs.get(c, dx, dy, dz);
      const double r2 = ray_dist2(dx, dy, dz, x - org.ox(c), y - org.oy(c), z - org.oz(c), cx, cy, cz);
      e.S += r2;
      e.r2max = max_nn(e.r2max, r2);
    }
  }
  return e;
}

// ---- LIT (knob TDR_LM_LITERAL_FD, tdr_lm_literal.hip): RayClosestPoint::compute taken literally for every ray.
// The distance exactly as Triangulator::distToRay evaluates it (src/Triangulator.cpp:3-9, built without FMA): both
// products of each cross-product component rounded, plain sums, IEEE square root.
__device__ __forceinline__ double dist_literal(double dx, double dy, double dz, double wx, double wy, double wz) {
  const double cx = mul_rn(dy, wz) - mul_rn(dz, wy);
  const double cy = mul_rn(dz, wx) - mul_rn(dx, wz);
  const double cz = mul_rn(dx, wy) - mul_rn(dy, wx);
  return __builtin_sqrt((mul_rn(cx, cx) + mul_rn(cy, cy)) + mul_rn(cz, cz));
}
// sum of the distances from the point (x, y, z) to the rays, in camera order: RayClosestPoint's `error`
// accumulator (src/RayTriangulator.cpp:21-26), evaluated once per system — for the point of the solver's last
// compute() call — when its outputs are written
template <int NCAM, bool LIT = false, typename RS>
__device__ __forceinline__ double lm_sum_dist(LdsCams org, const RS& rs, int n_cam, double x, double y, double z) {
  double sum = 0.0;
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam && ((rs.valid >> c) & 1u)) {
      double dx, dy, dz, cx, cy, cz;
      rs.get(c, dx, dy, dz);
      if constexpr (LIT) sum += dist_literal(dx, dy, dz, x - org.ox(c), y - org.oy(c), z - org.oz(c));
      else sum += sqrt_pos(ray_dist2(dx, dy, dz, x - org.ox(c), y - org.oy(c), z - org.oz(c), cx, cy, cz));
    }
  }
  return sum;
}

struct LmNormal {
  double a00, a01, a02, a11, a12, a22, v0, v1, v2;
  double S, r2max;
};

// One ray's contribution to compute(x, r, J) as the solver consumes it.  cv::LMSolver never looks at a single
// residual: it uses S = sum r_i^2, J^T J, J^T r and max r_i.  With w = x - origin,
//   c = d x w,   r^2 = |c|^2,   p = c x d = ((d x w) x d),   grad r = p / r      (exact gradient of the distance)
//   row of J^T J:  (p p^T) / r^2        row of J^T r:  (p / r) r = p        S:  r^2
// so the loop needs no square root at all: one v_rcp_f64 + one Newton step for 1 / r^2 (relative error
// ~(2^-23)^2 ~ 1.4e-14 in J^T J — which only shapes the step — while J^T r and S are exact sums of products).
// Round 2's first form went through r and 1/r (v_rsq_f64 + a Goldschmidt step, 2e-14 in all three): 44 VALU
// instructions per camera against 37 now.
//
// The reference differentiates numerically, (dist(x + eps e_k) - dist(x - eps e_k)) / (2 eps) with eps = 1e-4.
// With dist(x +- eps e_k)^2 = n_k +- t_k, n_k = r^2 + eps^2 |d x e_k|^2, t_k = 2 eps p_k this is
//   j_k = 2 p_k / (sqrt(n_k + t_k) + sqrt(n_k - t_k)) = g_k (1 - (eps^2 / 2 r^2)(|d x e_k|^2 - g_k^2)) + O(eps^4 / r^4).
// What is evaluated:
//   r >= 0.1 mm  the exact gradient.  The dropped truncation term is <= 5e-9 / r[mm]^2 relative, while the
//                reference's own subtraction carries ~1e-12 r[mm] of rounding noise; round 1 carried the term.
//                tests/tools/lm_jacobian_form.py runs the ORACLE with either form on 130 k systems: points agree
//                to 1.7e-8 relative, iteration counts are within 5 of each other on 97-99.7 % and identical on
//                19-33 % — exactly the spread between the oracle and the round-1 kernel: near convergence the
//                accept / reject decisions are rounding noise of S, not of J (profiles/r02_lm_jacobian_form.json;
//                the A/B switch TDR_LM_FD_TRUNCATION lived until commit fbcdc55).
//   r <  0.1 mm  (noise-free data near convergence) flagged `small`; with EXACT the literal six-sqrt central
//                difference j and the distance r are evaluated.
// Both cases leave the term in one shape, J^T J += wa j j^T, J^T r += wu j, S += r2:
//   gradient form  j = p, wa = 1 / r^2, wu = 1        central difference  j = j_fd, wa = 1, wu = r
// (multiplying by 1.0 is exact, so the generic accumulation below gives the same bits as code specialised for
// either case).  Every multiply-add is an explicit fma(): all instantiations must round identically, because
// which one a system runs through depends on its neighbours in the wave.
struct CamTerm {
  double j0, j1, j2, wa, wu, r2;
  unsigned key;  // lm_small_key(r2)
  bool small;
};

// A ray closer than 0.1 mm (and not exactly on the point: r2 == 0 is a missing observation's zero direction, or
// p == 0 anyway): decided on the high word of r2 with two integer instructions — r2 == 0 wraps to the largest
// key.  The threshold is the double whose high word is that of 1e-2 mm^2 (0.00999999977648258).
constexpr unsigned kSmallHi = 0x3F847AE1u;
__device__ __forceinline__ unsigned lm_small_key(double r2) { return (unsigned)__double2hiint(r2) - 1u; }
__device__ __forceinline__ bool lm_key_small(unsigned key) { return key < kSmallHi - 1u; }

// get_D() = |d|^2, the same for every pixel of a camera (|quaternion|^4, kept with the camera constants), so
// |d x e_k|^2 = D - d_k^2 costs one fma; it only feeds the central difference and is only fetched there.
// A missing observation (d = 0) has c = p = 0, r2 = 0 and contributes exact zeros.
// The reference's central difference (src/RayTriangulator.cpp:28-44) for a ray closer than 0.1 mm, in place over the
// gradient form of lm_camera_term (t.j* = p = c x d on entry).  One definition for every evaluator: the forms agree
// bit for bit.
__device__ __forceinline__ void lm_exact_patch(CamTerm& t, double dx, double dy, double dz, double D) {
  const double px = t.j0, py = t.j1, pz = t.j2;
  const double qx2 = fma(-dx, dx, D), qy2 = fma(-dy, dy, D), qz2 = fma(-dz, dz, D);
  const double e2 = kFdEps * kFdEps, two_e = 2 * kFdEps, inv2e = 1.0 / (2 * kFdEps);
  const double nx = fma(e2, qx2, t.r2), ny = fma(e2, qy2, t.r2), nz = fma(e2, qz2, t.r2);
  // n_k -+ t_k is a squared distance: >= 0 up to rounding
  t.j0 = (sqrt_pos(fmax(fma(two_e, px, nx), 0.0)) - sqrt_pos(fmax(fma(-two_e, px, nx), 0.0))) * inv2e;
  t.j1 = (sqrt_pos(fmax(fma(two_e, py, ny), 0.0)) - sqrt_pos(fmax(fma(-two_e, py, ny), 0.0))) * inv2e;
  t.j2 = (sqrt_pos(fmax(fma(two_e, pz, nz), 0.0)) - sqrt_pos(fmax(fma(-two_e, pz, nz), 0.0))) * inv2e;
  t.wa = 1.0;
  t.wu = sqrt_pos(t.r2);
}

// The central difference of src/RayTriangulator.cpp:28-44 with its SIX distance evaluations at the shifted points —
// not through the linearity of the cross product (lm_exact_patch): that form has the reference's truncation term but
// not its rounding NOISE.  Each shifted distance carries ~|w| eps = 5e-13 mm of cross-product rounding, which the
// division by 2 eps = 2e-4 turns into ~2.5e-9 of noise in every Jacobian entry (entries are <= 1) — five orders above
// the exact gradient's.  Round 5 found that this noise, not the truncation term, is what separates the default
// kernels' status words from the reference's on two-ray systems (profiles/r05_lm_literal_fd.json): the reference's
// noisier J^T r makes its creeping two-ray runs stop on the step test slightly more often.
// r and S follow cv::LMSolver: r_i = the distance, S = sum r_i * r_i (square root first, then the square).
__device__ __forceinline__ void lm_literal_patch(CamTerm& t, double dx, double dy, double dz, double px, double py,
                                                 double pz, double ox, double oy, double oz) {
  const double e = kFdEps;
  const double wx = px - ox, wy = py - oy, wz = pz - oz;
  t.j0 = (dist_literal(dx, dy, dz, (px + e) - ox, wy, wz) - dist_literal(dx, dy, dz, (px - e) - ox, wy, wz)) / (2 * e);
  t.j1 = (dist_literal(dx, dy, dz, wx, (py + e) - oy, wz) - dist_literal(dx, dy, dz, wx, (py - e) - oy, wz)) / (2 * e);
  t.j2 = (dist_literal(dx, dy, dz, wx, wy, (pz + e) - oz) - dist_literal(dx, dy, dz, wx, wy, (pz - e) - oz)) / (2 * e);
  const double r = dist_literal(dx, dy, dz, wx, wy, wz);
  t.wa = 1.0;
  t.wu = r;
  t.r2 = mul_rn(r, r);
}

// (ptx, pty, ptz): the point; (ox, oy, oz): the ray's origin — w = p - o is formed here (the same subtraction the callers
// made until round 5; the literal build needs the two apart)
template <bool EXACT, bool LIT = false, typename DF>
__device__ __forceinline__ CamTerm lm_camera_term(double dx, double dy, double dz, DF get_D, double ptx, double pty,
                                                  double ptz, double ox, double oy, double oz) {
  CamTerm t;
  const double wx = ptx - ox, wy = pty - oy, wz = ptz - oz;
  double cx, cy, cz;
  t.r2 = ray_dist2(dx, dy, dz, wx, wy, wz, cx, cy, cz);
  const double px = fma(cy, dz, -(cz * dy)), py = fma(cz, dx, -(cx * dz)), pz = fma(cx, dy, -(cy * dx));
  t.key = lm_small_key(t.r2);
  t.small = LIT || lm_key_small(t.key);
  // the gradient form first, unconditionally: the rare central-difference branch below overwrites it in place, so
  // the common path carries no merge of the two forms
  {
    // r2 == 0: p = 0, and the clamp keeps 1 / r2 finite, so every product below is an exact 0
    const double b = fmax(t.r2, kRsqFloor);  // r2 comes out of an fma: no canonicalisation, SGPR operand
    const double y = __builtin_amdgcn_rcp(b);
    t.wa = fma(fma(-b, y, 1.0), y, y);
    t.wu = 1.0;
    t.j0 = px;
    t.j1 = py;
    t.j2 = pz;
  }
  if constexpr (LIT) {
    lm_literal_patch(t, dx, dy, dz, ptx, pty, ptz, ox, oy, oz);
  } else {
    if (EXACT && t.small) lm_exact_patch(t, dx, dy, dz, get_D());
  }
  return t;
}

__device__ __forceinline__ void lm_accumulate(LmNormal& q, const CamTerm& t) {
  const double h0 = t.j0 * t.wa, h1 = t.j1 * t.wa, h2 = t.j2 * t.wa;
  q.a00 = fma(h0, t.j0, q.a00); q.a01 = fma(h0, t.j1, q.a01); q.a02 = fma(h0, t.j2, q.a02);
  q.a11 = fma(h1, t.j1, q.a11); q.a12 = fma(h1, t.j2, q.a12); q.a22 = fma(h2, t.j2, q.a22);
  q.v0 = fma(t.j0, t.wu, q.v0); q.v1 = fma(t.j1, t.wu, q.v1); q.v2 = fma(t.j2, t.wu, q.v2);
  q.S += t.r2;
}

// J^T J, J^T r, S and max r^2 at x for any subset of the cameras: one exec-masked block per camera, central
// difference where a ray is closer than 0.1 mm
template <int NCAM, bool LIT = false, typename RS>
__device__ __forceinline__ LmNormal lm_normal(LdsCams org, const RS& rs,
                                              int n_cam, double x, double y, double z) {
  LmNormal q{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam && ((rs.valid >> c) & 1u)) {
      double dx, dy, dz;
      rs.get(c, dx, dy, dz);
      const CamTerm t = lm_camera_term<true, LIT>(dx, dy, dz, [&] { return org.dd(c); }, x, y, z, org.ox(c),
                                                  org.oy(c), org.oz(c));
      lm_accumulate(q, t);
      q.r2max = max_nn(q.r2max, t.r2);
    }
  }
  return q;
}

// The same evaluation for a full camera set (n_cam == NCAM) without any per-camera control flow, so that the
// scheduler can interleave the cameras' dependent chains (rcp -> Newton -> scale -> accumulate) instead of
// running them one basic block after the other.  Missing observations carry the direction (0,0,0) (set at
// refill): c = d x w = 0, r2 = 0, p = 0, so every term they add is an exact 0 and no mask is needed.  The
// gradient form is used unconditionally; `need_exact` reports a present ray closer than 0.1 mm (noise-free data
// near convergence), for which the caller must repeat the pass through lm_normal() — the smallest key of the
// pass, one v_min_u32 per camera.  When no ray is that close, max r^2 is either >= 1e-2 or every r^2 is an exact
// zero, so the solver's stop test max r < FLT_EPSILON reads the same from S: r2max is set to S here and the
// per-camera maximum is left to lm_normal().
// The camera origins of the NEXT block are requested from LDS before this block's arithmetic starts, into the
// other of two register sets (compile-time parity after unrolling, no copies): the ds_read round trip then runs
// under the arithmetic instead of in front of every block (3.9 % of C5, A/B in alternating processes).  From 8
// cameras up — at 6 the twelve extra registers would cost the third wave per SIMD — and not in the 24 / 32-camera
// kernels, which have no register to spare.
// EXACT (round 3): the evaluator of a wave that holds a ray closer than 0.1 mm — about one system in a thousand on
// noisy data, i.e. one wave in sixteen at any time, for the rest of that system's run.  Same blocks of kLmInterleave
// cameras; the lanes with such a ray replace the gradient form by the reference's central difference in a sub-block
// that is entered only when some lane of the wave needs it for one of the block's cameras (lm_exact_patch, the code of
// lm_normal()), and max r^2 is kept per camera.  Until then such a wave ran through lm_normal(): one camera per basic
// block, every dependent chain (ds_read -> cross product -> rcp -> Newton -> accumulate) exposed at two waves per
// SIMD — 2.3 times the duration of a pass, 7 % of C5 (A/B against a build that never leaves the gradient form:
// 3.04 -> 2.83 ms).  No prefetch of the next block's origins here: with the patch's temporaries the twelve
// registers would cost the second wave per SIMD.
template <int NCAM, bool EXACT = false, bool LIT = false, typename RS>
__device__ __forceinline__ LmNormal lm_normal_full(LdsCams org, const RS& rs, int n_cam, double x,
                                                   double y, double z, bool& need_exact) {
  LmNormal q{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned kmin = 0xFFFFFFFFu;
  int nc = n_cam;
  constexpr bool kPrefetch = !EXACT && NCAM >= 8 && RS::kInLds == 0;
  double pox[2][kLmInterleave], poy[2][kLmInterleave], poz[2][kLmInterleave];
  if constexpr (kPrefetch) {
#pragma unroll
    for (int j = 0; j < kLmInterleave; ++j) {
      pox[0][j] = org.ox(j); poy[0][j] = org.oy(j); poz[0][j] = org.oz(j);
    }
  }
#pragma unroll
  for (int c0 = 0; c0 < NCAM; c0 += kLmInterleave) {
    // kLmInterleave cameras per basic block.  The test is always true here (n_cam == NCAM is a kernel
    // argument the compiler cannot see through) and costs one scalar compare: it only fences the instruction
    // selection and scheduling, which would otherwise spread all cameras over one huge block and keep
    // every camera's cross products alive at once (one wave per SIMD instead of two).  The empty asm makes the
    // operand opaque, so the compare is redone per block (s_cmp + s_cbranch_scc) instead of being hoisted into
    // NCAM / 2 lane masks that overflow the SGPR file (v_readlane spills in the loop).
    asm volatile("" : "+s"(nc));
    if (c0 < nc) {
      const int cur = (c0 / kLmInterleave) & 1, nxt = cur ^ 1;
#pragma unroll
      for (int j = 0; j < kLmInterleave; ++j) {
        const int cn = kPrefetch ? c0 + kLmInterleave + j : c0 + j;  // without prefetch: this block's own origins
        if (cn < NCAM) {
          pox[kPrefetch ? nxt : cur][j] = org.ox(cn); poy[kPrefetch ? nxt : cur][j] = org.oy(cn);
          poz[kPrefetch ? nxt : cur][j] = org.oz(cn);
        }
      }
      if constexpr (EXACT) {
        CamTerm t[kLmInterleave];
        bool any_small = false;
#pragma unroll
        for (int c = c0; c < c0 + kLmInterleave && c < NCAM; ++c) {
          double dx, dy, dz;
          rs.get(c, dx, dy, dz);
          t[c - c0] = lm_camera_term<false, LIT>(dx, dy, dz, [&] { return org.dd(c); }, x, y, z, pox[cur][c - c0],
                                                 poy[cur][c - c0], poz[cur][c - c0]);
          any_small = any_small || t[c - c0].small;
        }
        if (!LIT && __ballot(any_small) != 0ull) {  // (the literal build's terms are complete as they come)
#pragma unroll
          for (int c = c0; c < c0 + kLmInterleave && c < NCAM; ++c) {
            if (t[c - c0].small) {
              double dx, dy, dz;
              rs.get(c, dx, dy, dz);
              lm_exact_patch(t[c - c0], dx, dy, dz, org.dd(c));
            }
          }
        }
#pragma unroll
        for (int c = c0; c < c0 + kLmInterleave && c < NCAM; ++c) {
          lm_accumulate(q, t[c - c0]);
          q.r2max = max_nn(q.r2max, t[c - c0].r2);
        }
      } else {
#pragma unroll
        for (int c = c0; c < c0 + kLmInterleave && c < NCAM; ++c) {
          double dx, dy, dz;
          rs.get(c, dx, dy, dz);
          const CamTerm t = lm_camera_term<false>(dx, dy, dz, [&] { return org.dd(c); }, x, y, z, pox[cur][c - c0],
                                                  poy[cur][c - c0], poz[cur][c - c0]);
          kmin = kmin < t.key ? kmin : t.key;
          lm_accumulate(q, t);
        }
      }
    }
  }
  if constexpr (EXACT) {
    need_exact = true;
  } else {
    need_exact = lm_key_small(kmin);
    q.r2max = q.S;
  }
  return q;
}

// Reciprocal for the LM update: v_rcp_f64 seed + two Newton steps (<= 1 ulp, 5 instructions, a dependent
// chain of 5) instead of the IEEE division expansion (div_scale x2, rcp, 5 fma, div_fmas, div_fixup: 11
// instructions in one chain).  The divisions of a pass sit on the loop's critical path — solve, gain ratio,
// schedule — where nothing else of the lane can overlap them.  A zero operand gives NaN instead of inf: in
// the three pivots and two ratios that use it, both end in a NaN trial point / ratio that the accept test
// rejects the same way.  The two divisions of the lambda restart (1/det, 1/max|A^-1_ii|) keep the IEEE form:
// there inf and 0 are meaningful values (singular J^T J).
__device__ __forceinline__ double rcp_nr(double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = fma(fma(-b, y, 1.0), y, y);
  y = fma(fma(-b, y, 1.0), y, y);
  return y;
}

// a / b with one residual correction on top of rcp_nr (faithfully rounded)
__device__ __forceinline__ double div_nr(double a, double b) {
  const double y = rcp_nr(b);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
}

// solve_spd3 (tdr_math.h) with rcp_nr pivots.  Returns the smallest pivot (the caller's rank test; a NaN pivot
// makes every later one NaN, and v_min_f64 returns the other operand, so a broken factorisation reads as its
// last sound pivot or as the zero that broke it — both fail `> kEigTol * trace`).
__device__ __forceinline__ double lm_solve3(double m00, double m01, double m02, double m11, double m12, double m22,
                                            double v0, double v1, double v2, double& x0, double& x1, double& x2) {
  const double i0 = rcp_nr(m00);
  const double l10 = m01 * i0, l20 = m02 * i0;
  const double d1 = fma(-l10, m01, m11);
  const double t12 = fma(-l20, m01, m12);
  const double i1 = rcp_nr(d1);
  const double l21 = t12 * i1;
  const double d2 = fma(-l21, t12, fma(-l20, m02, m22));
  const double i2 = rcp_nr(d2);
  const double y1 = fma(-l10, v0, v1);
  const double y2 = fma(-l21, y1, fma(-l20, v0, v2));
  x2 = y2 * i2;
  x1 = fma(y1, i1, -(l21 * x2));
  x0 = fma(v0, i0, -fma(l10, x1, l20 * x2));
  return fmin(fmin(m00, d1), d2);
}

// ---- cv::solve / cv::invert with DECOMP_EIG on the solver's symmetric 3x3 (src/RayTriangulator.cpp:100-104 ->
// cv::LMSolver::run: `solve(Ap, v, d, DECOMP_EIG)` for the trial step, `invert(A, Ap, DECOMP_EIG)` for the lambda
// restart).  OpenCV takes the Jacobi eigen-decomposition A = V diag(w) V^T and back-substitutes with every
// eigenvalue |w_i| <= 2 DBL_EPSILON sum|w| DROPPED: on a rank-deficient J^T J — two rays always, three as the run
// converges, coplanar ones, lambda = 0 — the step is the minimum-norm (pseudo-inverse) one, not the NaN / huge step of a
// factorisation.  Round 2 solved with LDL^T only and biased the status words of the 2- and 3-ray systems towards
// TDR_LM_CAP (tests/tools/lm_solve_form.py shows the same bias on the CPU oracle with its solves
// swapped for LDL^T, and none from the Jacobian form).  For matrices that are safely full rank the two solves agree to
// rounding, so LDL^T stays on the hot path and a lane leaves it only when one of its pivots is below kEigTol of the
// trace: any matrix whose smallest eigenvalue is within the cut has such a pivot (the pivots multiply to det = w0 w1 w2
// and none exceeds the trace, so min pivot <= (2 eps)^(1/3) tr = 7.6e-6 tr).
// sym3_eig_pinv below is round 3's restatement of the Jacobi sweeps themselves (rotation order, formulas and the cut of
// the CPU restatement, oracle/tdr_oracle.c sym3_eig_solve, product by product) — bit-identical to it on every matrix
// tried.  Since round 4 the LM kernels use lm_cascade instead (further down) and this function is reached only through
// the test hook tdr_debug_sym3_eig_device, as the device-side yardstick for the cascade.
constexpr double kEigTol = 1e-5;

// One function for both uses so that the rotation loop exists once per kernel.  WANT_DIAG == false: x = pinv(M) v;
// true: x0 = max_k |pinv(M)_kk|.  Written on scalars (9 + 9 doubles of state, nothing copied) to keep its register
// footprint below what the hot loop already needs.
template <bool WANT_DIAG>
__device__ __forceinline__ void sym3_eig_pinv(double m00, double m01, double m02, double m11, double m12,
                                              double m22, double v0, double v1, double v2, double& x0, double& x1,
                                              double& x2) {
  // every product through mul_rn(): rounded on its own like the CPU restatement's (built with -ffp-contract=off).
  // The quotients and square roots of the rotation use the loop's rcp / rsq based forms (<= 1 ulp) instead of the
  // IEEE expansions (round 3: a two-ray run came here on every lambda = 0 pass, up to 1000 times, and the launch that
  // solved such systems was bound by this chain's latency — the expansions were two thirds of it).
  double a[3][3] = {{m00, m01, m02}, {m01, m11, m12}, {m02, m12, m22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma nounroll
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (!(off > mul_rn(kDblEps, diag))) break;  // (OpenCV's Jacobi stops at |pivot| <= eps); a NaN leaves at once
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] != 0.0) {
          // |theta| clamped: beyond 1e150 the rotation angle is below 1e-150 either way, and the square stays finite
          const double th = div_nr(a[q][q] - a[p][p], mul_rn(2.0, a[p][q]));
          const double theta = fmax(fmin(th, 1e150), -1e150);
          const double t = (theta >= 0 ? 1.0 : -1.0) * rcp_nr(fabs(theta) + sqrt_pos(mul_rn(theta, theta) + 1));
          double nrm, c;
          sqrt_rsqrt_pos(mul_rn(t, t) + 1, nrm, c);
          const double sn = mul_rn(t, c);
#pragma unroll
          for (int k = 0; k < 3; ++k) {  // A <- A G
            const double akp = a[k][p], akq = a[k][q];
            a[k][p] = mul_rn(c, akp) - mul_rn(sn, akq);
            a[k][q] = mul_rn(sn, akp) + mul_rn(c, akq);
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {  // A <- G^T A
            const double apk = a[p][k], aqk = a[q][k];
            a[p][k] = mul_rn(c, apk) - mul_rn(sn, aqk);
            a[q][k] = mul_rn(sn, apk) + mul_rn(c, aqk);
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double vkp = V[k][p], vkq = V[k][q];
            V[k][p] = mul_rn(c, vkp) - mul_rn(sn, vkq);
            V[k][q] = mul_rn(sn, vkp) + mul_rn(c, vkq);
          }
        }
      }
  }
  // back-substitution with the 2 eps sum|w| cut (cv::SVBkSb): M = sum_kept V_i (1 / w_i) V_i^T, row by row
  const double thr = mul_rn(2 * kDblEps, fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]));
  double iw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) iw[i] = fabs(a[i][i]) <= thr ? 0.0 : rcp_nr(a[i][i]);  // 0: the term is skipped below
  double out[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double mr[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (iw[i] != 0.0) {
        const double vi = mul_rn(V[r][i], iw[i]);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) mr[cc] += mul_rn(vi, V[cc][i]);
      }
    }
    out[r] = WANT_DIAG ? fabs(mr[r]) : mul_rn(mr[0], v0) + mul_rn(mr[1], v1) + mul_rn(mr[2], v2);
  }
  if (WANT_DIAG) {
    x0 = fmax(fmax(out[0], out[1]), out[2]);
  } else {
    x0 = out[0]; x1 = out[1]; x2 = out[2];
  }
}

__device__ __forceinline__ void sym3_eig_solve(double m00, double m01, double m02, double m11, double m12,
                                               double m22, double v0, double v1, double v2, double& x0,
                                               double& x1, double& x2) {
  sym3_eig_pinv<false>(m00, m01, m02, m11, m12, m22, v0, v1, v2, x0, x1, x2);
}

// max_k |pinv(M)_kk|
__device__ __forceinline__ double sym3_eig_max_inv_diag(double m00, double m01, double m02, double m11, double m12,
                                                        double m22) {
  double mx, u1, u2;
  sym3_eig_pinv<true>(m00, m01, m02, m11, m12, m22, 0.0, 0.0, 0.0, mx, u1, u2);
  return mx;
}

// ---- cv::solve / cv::invert(DECOMP_EIG) WITHOUT an eigen-decomposition (round 4) -------------------------------------
// What the reference's DECOMP_EIG returns for the solver's symmetric PSD 3x3 is the pseudo-inverse with every
// eigenvalue |w_i| <= 2 eps sum|w| dropped.  Which of them are dropped — the RANK — can be read off the matrix's own
// invariants, tr = sum w, e2 = the sum of the principal 2x2 minors (= w0 w1 + w0 w2 + w1 w2, ~ w1 w2 once w0 is
// small) and det, and each rank has a closed form in the matrix's entries:
//   tr <= 0                zero matrix: pinv = 0
//   e2 <= 2 eps tr^2       w1 within the cut: rank 1, pinv = M / tr^2
//   det <= 2 eps tr e2     w0 = det / e2 within the cut: rank 2.  With adj = the cofactor matrix (= w1 w2 n n^T for the
//                          unit null vector n):  y = (tr v - M v) / e2 inverts on the range (on an eigenvector of w1:
//                          (tr - w1) / (w1 w2) = 1 / w1) but sends n to (tr / e2) n.  With TWO rays v = J^T r has no
//                          null component by construction and y is the answer — near convergence J^T J tends to rank 1
//                          there, n becomes noise, and a projection along it biases the runs (measured on the oracle:
//                          62 652 : 494).  With more rays v has a small genuine null component (<= |r| sqrt(w0)) that
//                          the reference drops: y is cleared of its component along n (the largest column of adj;
//                          clearing v BEFORE the product instead costs eps x cond^2).
//                          pinv_kk = (tr (1 - adj_kk / e2) - M_kk) / e2.
//   otherwise              full rank: LDL^T, max_k |cof_kk| / det
// This is where two- and three-ray systems spend their runs: J^T r = 0 at the minimiser makes the gradients linearly
// dependent there, so J^T J of a two-ray system tends to rank 1 and of a three-ray system to rank 2 as the run
// converges, and every one of the (on average 244) lambda = 0 passes of a two-ray system is a rank-2 solve.  Round 3
// restated OpenCV's Jacobi sweeps for them (sym3_eig_pinv below: ~1000 dependent instructions per solve, 55-70 registers)
// and ran them in a second, latency-bound launch for "parked" systems — 4 % to 28 % of the systems of a 4- or 3-camera
// rig, 3.7 of the 6.6 ms of 1 M four-camera systems.  The closed forms are ~60 operations without a loop and fit the
// throughput kernels: there is no second launch any more.
// Not the reference's roundings, and within a few eps of the cuts the rank decision is noise (e2 and det come out of
// cancelling products) where the Jacobi form decides on eigenvalues of high relative accuracy — so it was measured on
// the CPU oracle before it was built (tests/tools/lm_hybrid_rule.py, oracle switch oracle_lm_hybrid_rule = 3,
// profiles/r04_lm_hybrid_rule.json: 5.56 M two- to four-ray systems; status flags variant-only : reference-only
// 490 : 491 with two rays, 90 : 101 with three, 20 : 16 with four, against 480 : 511 / 102 : 94 / 16 : 18 for an input
// moved by one ulp; against the Jacobi solve on 20 000 random matrices of every rank: the same rank on every one, the
// solution within 4 eps x cond).  The oracle's sym3_cascade evaluates the same operations in the same order —
// bit-identical through the test hook tdr_debug_lm_cascade_device — and the device keeps the Jacobi restatement as a
// second yardstick (tdr_debug_sym3_eig_device).
// Cold path: a lane comes here only when a pivot of the hot LDL^T is below kEigTol of the trace (any matrix within a
// cut has such a pivot, see kEigTol).  Every product that is followed by an addition goes through mul_rn / an
// explicit fma: the CPU restatement is built with -ffp-contract=off.
// Returns the rank.  WANT_DIAG: x0 = max_k |pinv(M)_kk|; else (x0, x1, x2) = pinv(M) v.
template <bool WANT_DIAG>
__device__ __forceinline__ int lm_cascade(double a00, double a01, double a02, double a11, double a12, double a22,
                                          double v0, double v1, double v2, int n_rays, double& x0, double& x1,
                                          double& x2) {
  const double tr = (a00 + a11) + a22;
  const double c00 = fma(a11, a22, -(a12 * a12)), c11 = fma(a00, a22, -(a02 * a02)), c22 = fma(a00, a11, -(a01 * a01));
  const double e2 = (c00 + c11) + c22;
  const double cut = 2 * kDblEps;
  if (!(tr > 0)) {  // zero (or NaN) matrix
    x0 = x1 = x2 = 0.0;
    return 0;
  }
  if (!(e2 > cut * (tr * tr))) {  // rank 1
    const double it2 = 1.0 / (tr * tr);
    if constexpr (WANT_DIAG) {
      x0 = fmax(fabs(a00), fmax(fabs(a11), fabs(a22))) * it2;
    } else {
      x0 = fma(a00, v0, fma(a01, v1, a02 * v2)) * it2;
      x1 = fma(a01, v0, fma(a11, v1, a12 * v2)) * it2;
      x2 = fma(a02, v0, fma(a12, v1, a22 * v2)) * it2;
    }
    return 1;
  }
  // det as the product of the unpivoted LDL^T pivots (any pivot <= 0: singular to working precision)
  const double l10 = a01 / a00, l20 = a02 / a00;
  const double p1 = a11 - mul_rn(l10, a01);
  const double t12 = a12 - mul_rn(l20, a01);
  const double p2 = (a22 - mul_rn(l20, a02)) - mul_rn(t12 / p1, t12);
  const double det = (a00 > 0 && p1 > 0 && p2 > 0) ? (a00 * p1) * p2 : 0.0;
  if (!(det > (cut * tr) * e2)) {  // rank 2
    const double ie2 = 1.0 / e2;
    if constexpr (WANT_DIAG) {
      const double q0 = fma(tr, fma(-c00, ie2, 1.0), -a00) * ie2;
      const double q1 = fma(tr, fma(-c11, ie2, 1.0), -a11) * ie2;
      const double q2 = fma(tr, fma(-c22, ie2, 1.0), -a22) * ie2;
      x0 = fmax(fabs(q0), fmax(fabs(q1), fabs(q2)));
    } else {
      const double w0 = fma(a00, v0, fma(a01, v1, a02 * v2));
      const double w1 = fma(a01, v0, fma(a11, v1, a12 * v2));
      const double w2 = fma(a02, v0, fma(a12, v1, a22 * v2));
      const double y0 = fma(tr, v0, -w0) * ie2, y1 = fma(tr, v1, -w1) * ie2, y2 = fma(tr, v2, -w2) * ie2;
      x0 = y0; x1 = y1; x2 = y2;
      if (n_rays > 2) {  // see the table above: never with two rays
        const double c01 = fma(a02, a12, -(a01 * a22)), c02 = fma(a01, a12, -(a02 * a11)),
                     c12 = fma(a01, a02, -(a00 * a12));
        double n0, n1, n2;
        if (c00 >= c11 && c00 >= c22) {
          n0 = c00; n1 = c01; n2 = c02;
        } else if (c11 >= c22) {
          n0 = c01; n1 = c11; n2 = c12;
        } else {
          n0 = c02; n1 = c12; n2 = c22;
        }
        const double nn = fma(n0, n0, fma(n1, n1, n2 * n2));
        const double t = fma(n0, y0, fma(n1, y1, n2 * y2)) / nn;
        x0 = fma(-t, n0, y0); x1 = fma(-t, n1, y1); x2 = fma(-t, n2, y2);
      }
    }
    return 2;
  }
  if constexpr (WANT_DIAG) {
    x0 = fmax(fabs(c00), fmax(fabs(c11), fabs(c22))) / det;
  } else {  // LDL^T with IEEE reciprocals (the CPU restatement's sym3_ldlt_solve)
    const double i0 = 1.0 / a00;
    const double m10 = a01 * i0, m20 = a02 * i0;
    const double d1 = a11 - mul_rn(m10, a01);
    const double s12 = a12 - mul_rn(m20, a01);
    const double i1 = 1.0 / d1;
    const double m21 = s12 * i1;
    const double d2 = (a22 - mul_rn(m20, a02)) - mul_rn(m21, s12);
    const double i2 = 1.0 / d2;
    const double y1 = v1 - mul_rn(m10, v0);
    const double y2 = (v2 - mul_rn(m20, v0)) - mul_rn(m21, y1);
    x2 = y2 * i2;
    x1 = mul_rn(y1, i1) - mul_rn(m21, x2);
    x0 = mul_rn(v0, i0) - (mul_rn(m10, x1) + mul_rn(m20, x2));
  }
  return 3;
}

// per-lane LM state
struct LmState {
  double x, y, z;
  double S, r2max;    // sum r^2 and max r^2 at (x, y, z)
  double tx, ty, tz;  // point of the last compute() call (the last trial, accepted or not): `err` is reported there
  double lambda, lc, D0, D1, D2;
  LmNormal A;
  int iter;
  int n_valid;
  bool need_J;
};

// The reference loop body (cv::LMSolver::run) in two halves around the evaluation of the trial point:
//   lm_trial  : solve (A + lambda D) d = v, trial point x - d
//   lm_update : lambda schedule from R, accept test, stop test.  Returns true when the lane has finished.
// LIT: (nearly) rank-deficient solves through the restated Jacobi sweeps of cv::solve / cv::invert(DECOMP_EIG)
// (sym3_eig_pinv, round 3's form) instead of the closed-form rank cascade.
template <bool LIT = false>
__device__ __forceinline__ void lm_trial(const LmState& st, double& d0, double& d1, double& d2) {
  const LmNormal& A = st.A;
  const double m00 = fma(st.lambda, st.D0, A.a00), m11 = fma(st.lambda, st.D1, A.a11),
               m22 = fma(st.lambda, st.D2, A.a22);
  const double pmin = lm_solve3(m00, A.a01, A.a02, m11, A.a12, m22, A.v0, A.v1, A.v2, d0, d1, d2);
  // (nearly) rank-deficient: the reference's DECOMP_EIG answer by rank, in closed form (lm_cascade; cold branch)
  if (!(pmin > kEigTol * (m00 + m11 + m22))) {
    double r0, r1, r2;
    if constexpr (LIT) sym3_eig_solve(m00, A.a01, A.a02, m11, A.a12, m22, A.v0, A.v1, A.v2, r0, r1, r2);
    else lm_cascade<false>(m00, A.a01, A.a02, m11, A.a12, m22, A.v0, A.v1, A.v2, st.n_valid, r0, r1, r2);
    d0 = r0; d1 = r1; d2 = r2;
  }
}

template <bool LIT = false>
__device__ __forceinline__ bool lm_update(LmState& st, double d0, double d1, double d2, double Sd, double r2max_d,
                                          bool& accepted) {
  const LmNormal& A = st.A;
  const double Rlo = 0.25, Rhi = 0.75;
  const double lambda_in = st.lambda, lc_in = st.lc;
  // the callback's `error` is overwritten by every compute(): remember where the last one was
  st.tx = st.x - d0; st.ty = st.y - d1; st.tz = st.z - d2;
  // dS = d . (2 v - A d)
  // (explicit fma chains here and below: the function is inlined into several kernels whose results must
  // agree bit for bit, so nothing is left to the contraction pass)
  const double t0 = fma(2.0, A.v0, -fma(A.a00, d0, fma(A.a01, d1, A.a02 * d2)));
  const double t1 = fma(2.0, A.v1, -fma(A.a01, d0, fma(A.a11, d1, A.a12 * d2)));
  const double t2 = fma(2.0, A.v2, -fma(A.a02, d0, fma(A.a12, d1, A.a22 * d2)));
  const double dS = fma(d0, t0, fma(d1, t1, d2 * t2));
  // The gain ratio R = (S - Sd) / dS is only ever compared with Rlo and Rhi: the two tests are made on the
  // products instead of on a quotient (a NaN anywhere fails both, as it does in the reference).
  const double num = st.S - Sd;
  const double den = fabs(dS) > kDblEps ? dS : 1.0;
  const double thi = Rhi * den, tlo = Rlo * den;
  const bool den_pos = den > 0.0;
  const bool r_hi = den_pos ? num > thi : num < thi;
  const bool r_lo = den_pos ? num < tlo : num > tlo;
  if (r_hi) {
    st.lambda *= 0.5;
    if (st.lambda < st.lc) st.lambda = 0;
  } else if (r_lo) {
    const double t = fma(d0, A.v0, fma(d1, A.v1, d2 * A.v2));
    double nu = div_nr(Sd - st.S, fabs(t) > kDblEps ? t : 1.0) + 2;
    nu = fmin(fmax(nu, 2.), 10.);
    if (st.lambda == 0) {
      // lambda restart: 1 / max(eps, max_k |(A^-1)_kk|) with (A^-1)_kk = cof_kk / det, i.e.
      // min(1 / eps, |det| / max_k |cof_kk|) — one quotient instead of 1 / det, three products and 1 / maxval.
      const double c00 = fma(A.a11, A.a22, -(A.a12 * A.a12));
      const double c01 = fma(A.a02, A.a12, -(A.a01 * A.a22));
      const double c02 = fma(A.a01, A.a12, -(A.a02 * A.a11));
      const double det = fma(A.a00, c00, fma(A.a01, c01, A.a02 * c02));
      const double c11 = fma(A.a00, A.a22, -(A.a02 * A.a02));
      const double c22 = fma(A.a00, A.a11, -(A.a01 * A.a01));
      const double q = fmax(fabs(c00), fmax(fabs(c11), fabs(c22)));
      double lam = fmin(1.0 / kDblEps, div_nr(fabs(det), q));
      // |det| / q = 1 / max_k (A^-1)_kk lies in [w_min, 3 w_min]: where it is within kEigTol of the trace the
      // reference's invert(A, DECOMP_EIG) may have dropped an eigenvalue — the pseudo-inverse's diagonal by rank
      // (lm_cascade; a zero matrix gives 0 there and 1 / eps here, as the reference's max(eps, .) does)
      if (!(fabs(det) > kEigTol * (A.a00 + A.a11 + A.a22) * q)) {
        double mx, u1 = 0, u2 = 0;
        (void)u1; (void)u2;
        if constexpr (LIT) mx = sym3_eig_max_inv_diag(A.a00, A.a01, A.a02, A.a11, A.a12, A.a22);
        else lm_cascade<true>(A.a00, A.a01, A.a02, A.a11, A.a12, A.a22, 0.0, 0.0, 0.0, st.n_valid, mx, u1, u2);
        lam = 1.0 / fmax(kDblEps, mx);
      }
      st.lambda = st.lc = lam;
      nu *= 0.5;
    }
    st.lambda *= nu;
  }
  accepted = Sd < st.S;
  if (accepted) {  // r(x_new) == rd; the caller installs J^T J, J^T r of the new point
    st.S = Sd;
    st.x -= d0; st.y -= d1; st.z -= d2;
    st.r2max = r2max_d;
  }
  ++st.iter;
  const double dinf = fmax(fabs(d0), fmax(fabs(d1), fabs(d2)));
  // Fixed point of the reference loop: a rejected step that leaves lambda and lc unchanged reproduces
  // itself exactly (same A, v, lambda -> same d, trial point, R, branch) until the 1000-iteration cap.
  // Stopping here returns bit for bit what iteration 1000 would.
  const bool spinning = !accepted && st.lambda == lambda_in && st.lc == lc_in;
  const bool go_on = st.iter < kMaxLmIters && dinf >= kFltEps && st.r2max >= kFltEps * kFltEps;
  if (spinning && go_on) st.iter = kMaxLmIters;  // what the reference's loop counter reads when it gets out
  return spinning || !go_on;
}

// outputs of a finished system.  status (optional) receives TDR_LM_CAP here; lm_status_kernel adds
// TDR_LM_NOT_CONVERGED afterwards (include/tdr_amd.h)
struct LmOut {
  double* __restrict__ xyz;
  double* __restrict__ err;
  int* __restrict__ iters;
  int* __restrict__ status;
  // sum_dist: lm_sum_dist() at (st.tx, st.ty, st.tz); only read when err is requested
  __device__ __forceinline__ void write(int64_t s, const LmState& st, double sum_dist) const {
    xyz[s * 3 + 0] = st.x;
    xyz[s * 3 + 1] = st.y;
    xyz[s * 3 + 2] = st.z;
    if (err) err[s] = sum_dist / (double)st.n_valid;
    if (iters) iters[s] = st.iter;
    if (status) status[s] = st.iter >= kMaxLmIters ? kLmCap : 0;
  }
  __device__ __forceinline__ void write_too_few(int64_t s) const {  // the reference throws for this frame
    const double nan = __builtin_nan("");
    xyz[s * 3 + 0] = nan; xyz[s * 3 + 1] = nan; xyz[s * 3 + 2] = nan;
    if (err) err[s] = nan;
    if (iters) iters[s] = 0;
    if (status) status[s] = kLmTooFew;
  }
};

// ---- cooperative drain --------------------------------------------------------------------------------
// Once the work queue is empty a wave keeps running until its slowest system stops — up to several hundred
// passes with a handful of lanes busy, each pass costing the full 16-camera evaluation (0.75 ms of the 4.35 ms
// of C5, and most of a small batch).  When few enough systems are left (4 at 9-16 cameras, 8 at 5-8, 16 up to
// 4), the wave therefore changes layout: each system gets a group of 16 / 8 / 4 lanes, lane k of the group
// evaluates camera k, the five per-camera results are exchanged inside the group (ds_bpermute) and every lane
// of the group accumulates them in camera order and runs the LM update redundantly.  A pass then costs one
// camera term + NCAM x 12 accumulation fma + the update instead of NCAM camera terms.  The accumulation order and every formula are those of the lane-per-system
// form (same lm_camera_term / lm_accumulate / lm_trial / lm_update; missing observations add exact zeros), so
// the results do not depend on where a system finishes.
constexpr int kCoopMaxCam = 16;
// lanes per system: the next power of two >= NCAM (4, 8 or 16), so 16, 8 or 4 systems fit a wave
__host__ __device__ constexpr int coop_group(int ncam) { return ncam <= 4 ? 4 : (ncam <= 8 ? 8 : 16); }

__device__ __forceinline__ double shfl_d(double v, int src_lane) { return __shfl(v, src_lane, kWave); }

// The cooperative LM loop of one wave: up to 64 / G systems, lane k of a group holds camera k's direction
// (mdx, mdy, mdz; zero for a missing observation or a lane without a camera) and a full copy of its system's
// LM state.  Runs until every active group has finished and written its result (lane 0 of the group).
template <int NCAM, bool LIT = false>
__device__ __forceinline__ void lm_coop_run(LdsCams org, LmState& st, double mdx, double mdy, double mdz,
                                            int64_t s, bool active, int n_cam, const LmOut& out) {
  constexpr int kCoopGroup = coop_group(NCAM);
  const int lane = threadIdx.x & (kWave - 1);
  const int k = lane % kCoopGroup;
  const bool has_cam = k < n_cam;
  const int kc = has_cam ? k : 0;  // lanes without a camera read camera 0's constants and contribute zeros
  if (!has_cam) mdx = mdy = mdz = 0.0;
  const double ox = org.ox(kc), oy = org.oy(kc), oz = org.oz(kc), dd = org.dd(kc);
  const int group_base = lane - k;
  while (__ballot(active) != 0ull) {
    // every lane runs the pass (the exchange needs all lanes of a group); finished groups just idle along
    const bool init = st.iter < 0;
    double d0 = 0, d1 = 0, d2 = 0;
    if (!init) lm_trial<LIT>(st, d0, d1, d2);  // every lane of a group holds the same state and takes the same step
    const double x = st.x - d0, y = st.y - d1, z = st.z - d2;
    const CamTerm t = lm_camera_term<true, LIT>(mdx, mdy, mdz, [&] { return dd; }, x, y, z, ox, oy, oz);
    LmNormal E{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCAM; ++c) {
      CamTerm tc;
      tc.j0 = shfl_d(t.j0, group_base + c); tc.j1 = shfl_d(t.j1, group_base + c);
      tc.j2 = shfl_d(t.j2, group_base + c); tc.wa = shfl_d(t.wa, group_base + c);
      tc.wu = shfl_d(t.wu, group_base + c); tc.r2 = shfl_d(t.r2, group_base + c);
      lm_accumulate(E, tc);
      E.r2max = max_nn(E.r2max, tc.r2);
    }
    bool finished = false;
    if (init) {  // LMSolver::run prologue
      st.A = E;
      st.S = E.S; st.r2max = E.r2max;
      st.tx = x; st.ty = y; st.tz = z;
      st.D0 = E.a00; st.D1 = E.a11; st.D2 = E.a22;
      st.lambda = 1; st.lc = 0.75; st.iter = 0;
    } else {
      bool accepted;
      finished = lm_update<LIT>(st, d0, d1, d2, E.S, E.r2max, accepted);
      if (accepted) st.A = E;
    }
    if (__ballot(active && finished) != 0ull) {
      // the reported error: distances at the point just evaluated — it is the last compute() of a system that
      // finishes in this pass — summed over the group in camera order (lm_sum_dist's order and operands)
      double sum_dist = 0.0;
      if (out.err) {
        const double r = LIT ? t.wu : sqrt_pos(t.r2);  // (the literal build's term carries the distance itself)
#pragma unroll
        for (int c = 0; c < NCAM; ++c) sum_dist += shfl_d(r, group_base + c);
      }
      if (active && finished) {
        if (k == 0) out.write(s, st, sum_dist);
        active = false;
      }
    }
    if (!active) st.iter = 0, st.lambda = 1.0;  // keep an idle group's arithmetic tame
  }
}

template <int NCAM, bool LIT = false, typename RS>
__device__ __forceinline__ void lm_coop_finish(LdsCams org, const RS& rs, const LmState& st_in,
                                               int64_t s_in, int n_cam, const LmOut& out) {
  static_assert(NCAM <= kCoopMaxCam, "one lane per camera");
  constexpr int kCoopGroup = coop_group(NCAM), kCoopSystems = kWave / kCoopGroup;
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane / kCoopGroup, k = lane % kCoopGroup;
  // the g-th busy lane feeds group g (all lanes are active here: wave-uniform call site)
  unsigned long long busy = __ballot(s_in >= 0);
  const int n_groups = __popcll(busy);
  int my_src = 0;
#pragma unroll
  for (int i = 0; i < kCoopSystems; ++i) {
    const int src = busy ? (int)__ffsll((long long)busy) - 1 : 0;
    busy &= busy - 1ull;
    if (g == i) my_src = src;
  }
  // ---- gather: the system's LM state to all lanes of its group, camera k's direction to lane k
  LmState st;
  st.x = shfl_d(st_in.x, my_src); st.y = shfl_d(st_in.y, my_src); st.z = shfl_d(st_in.z, my_src);
  st.S = shfl_d(st_in.S, my_src); st.r2max = shfl_d(st_in.r2max, my_src);
  st.tx = shfl_d(st_in.tx, my_src); st.ty = shfl_d(st_in.ty, my_src); st.tz = shfl_d(st_in.tz, my_src);
  st.lambda = shfl_d(st_in.lambda, my_src); st.lc = shfl_d(st_in.lc, my_src);
  st.D0 = shfl_d(st_in.D0, my_src); st.D1 = shfl_d(st_in.D1, my_src); st.D2 = shfl_d(st_in.D2, my_src);
  st.A.a00 = shfl_d(st_in.A.a00, my_src); st.A.a01 = shfl_d(st_in.A.a01, my_src);
  st.A.a02 = shfl_d(st_in.A.a02, my_src); st.A.a11 = shfl_d(st_in.A.a11, my_src);
  st.A.a12 = shfl_d(st_in.A.a12, my_src); st.A.a22 = shfl_d(st_in.A.a22, my_src);
  st.A.v0 = shfl_d(st_in.A.v0, my_src); st.A.v1 = shfl_d(st_in.A.v1, my_src);
  st.A.v2 = shfl_d(st_in.A.v2, my_src);
  st.A.S = st.A.r2max = 0.0;  // not read by lm_trial / lm_update
  st.iter = __shfl(st_in.iter, my_src, kWave);
  st.n_valid = __shfl(st_in.n_valid, my_src, kWave);
  st.need_J = false;
  const int s_lo = __shfl((int)(unsigned)(unsigned long long)s_in, my_src, kWave);
  const int s_hi = __shfl((int)(unsigned)((unsigned long long)s_in >> 32), my_src, kWave);
  const int64_t s = (int64_t)(((unsigned long long)(unsigned)s_hi << 32) | (unsigned long long)(unsigned)s_lo);
  double mdx = 0.0, mdy = 0.0, mdz = 0.0;
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    double rx, ry, rz;
    rs.get(c, rx, ry, rz);
    const double tx = shfl_d(rx, my_src), ty = shfl_d(ry, my_src), tz = shfl_d(rz, my_src);
    if (k == c) { mdx = tx; mdy = ty; mdz = tz; }
  }
  lm_coop_run<NCAM, LIT>(org, st, mdx, mdy, mdz, s, g < n_groups, n_cam, out);
}

// Camera-per-lane kernel for small batches: a batch that cannot fill the chip in the lane-per-system layout
// (the reference's own recordings are a few thousand frames) is latency-bound — every pass of a lone wave
// costs NCAM camera terms — so it runs in the cooperative layout from the start: 64 / G systems per wave, lane
// k of a group loads and sets up camera k's observation, then lm_coop_run.  Same arithmetic in the same order
// as ray_kernel (the start point is summed in camera order too), hence the same bits.
template <int NCAM, bool LIT = false>
__global__ void __launch_bounds__(kWave)
    ray_coop_kernel(const CamR<NCAM> cams, int n_cam, const double2* __restrict__ pts, int64_t cam_stride,
                    int64_t n_sys, double* __restrict__ xyz, double* __restrict__ err,
                    int* __restrict__ iters, int* __restrict__ status,
                    unsigned long long* __restrict__ first_bad) {
  constexpr int kCoopGroup = coop_group(NCAM), kCoopSystems = kWave / kCoopGroup;
  __shared__ __attribute__((aligned(16))) double cam_lds[NCAM * 12];
  const LmOut out{xyz, err, iters, status};
  const int lane = threadIdx.x;
  if (lane < n_cam) lds_store_camera(cam_lds + 12 * lane, cams.c[lane]);
  __syncthreads();
  const LdsCams org{(lds_vdouble*)cam_lds};
  const int g = lane / kCoopGroup, k = lane % kCoopGroup, group_base = lane - k;
  const bool has_cam = k < n_cam;
  const int kc = has_cam ? k : 0;
  for (int64_t base = (int64_t)blockIdx.x * kCoopSystems; base < n_sys;
       base += (int64_t)gridDim.x * kCoopSystems) {
    const int64_t s = base + g;
    const bool have = s < n_sys;
    bool ok = false;
    double mdx = 0.0, mdy = 0.0, mdz = 0.0;
    if (have && has_cam) {
      const double2 uv = pts[(int64_t)k * cam_stride + s];
      ok = !is_missing(uv.x, uv.y);
      org.ray_dir(kc, uv.x, uv.y, mdx, mdy, mdz);
      if (!ok) mdx = mdy = mdz = 0.0;
    }
    // start point = mean of the valid rays' origins, summed in camera order (src/RayTriangulator.cpp:90-98)
    int n_valid = 0;
    double x = 0, y = 0, z = 0;
#pragma unroll
    for (int c = 0; c < NCAM; ++c) {
      const bool okc = __shfl((int)ok, group_base + c, kWave) != 0;
      n_valid += okc ? 1 : 0;
      x += okc ? org.ox(c) : 0.0;
      y += okc ? org.oy(c) : 0.0;
      z += okc ? org.oz(c) : 0.0;
    }
    bool active = have;
    if (have && n_valid < 2) {  // the reference throws for this frame; record it and move on
      if (k == 0) {
        atomicMin(first_bad, (unsigned long long)s);
        out.write_too_few(s);
      }
      active = false;
    }
    LmState st;
    st.x = x / n_valid; st.y = y / n_valid; st.z = z / n_valid;
    st.S = st.r2max = 0.0;
    st.tx = st.x; st.ty = st.y; st.tz = st.z;
    st.lambda = 1.0; st.lc = 0.75; st.D0 = st.D1 = st.D2 = 1.0;
    st.A = LmNormal{1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0};
    st.iter = -1;  // "initial compute pending"
    st.n_valid = n_valid;
    st.need_J = false;
    if (!active) st.x = st.y = st.z = 0.0;
    lm_coop_run<NCAM, LIT>(org, st, mdx, mdy, mdz, s, active, n_cam, out);
  }
}

// NLDS: directions of the last NLDS cameras in LDS (24 / 32 cameras), register-allocated for two waves per SIMD.
// counters[0] is the head of the work queue.  (Round 3 had three forms of this kernel — a throughput form that parked
// what needed the eigen solve, a second launch that adopted the parked systems with the Jacobi sweeps inlined, and a
// standalone form — since round 4 the rank-deficient solves are closed forms in every kernel, lm_cascade.)
// WAVES: waves per SIMD the kernel is register-allocated for; 0 = the rule of rounds 1-4 (two with directions in LDS,
// else whatever the registers allow)
__host__ __device__ constexpr int lm_min_waves(int nlds, int waves) {
  return waves > 0 ? waves : (nlds > 0 ? 2 : TDR_LM_WAVES);
}
template <int NCAM, typename Cams, bool SPEC, bool FULL = false, int NLDS = 0, bool LIT = false, int WAVES = 0>
__global__ void __launch_bounds__(kWave, lm_min_waves(NLDS, WAVES))
    ray_kernel(const Cams cams, int n_cam, const double2* __restrict__ pts, int64_t cam_stride,
               int64_t n_sys, double* __restrict__ xyz, double* __restrict__ err,
               int* __restrict__ iters, int* __restrict__ status, unsigned long long* __restrict__ first_bad,
               unsigned long long* __restrict__ counters, int refill_threshold, int max_wait,
               int j_quarters) {
  unsigned long long* const work_counter = counters;
  __shared__ __attribute__((aligned(16))) double cam_lds[NCAM * 12];
  const LmOut out{xyz, err, iters, status};
  const int lane = threadIdx.x & (kWave - 1);
  if (lane < n_cam) lds_store_camera(cam_lds + 12 * lane, cams.c[lane]);
  __syncthreads();
  const LdsCams org{(lds_vdouble*)cam_lds};
  // SPEC kernels have no J phase: its two arguments carry the cooperative-drain switch and the claim size
  const bool coop = SPEC && j_quarters != 0;
  const int lm_chunk = SPEC && max_wait > 0 ? max_wait : kLmChunk;
  __shared__ __attribute__((aligned(16))) double dir_lds[NLDS > 0 ? NLDS * 3 * kWave : 1];
  RaySet<NCAM, NLDS> rs;
  rs.l = (lds_vdouble*)dir_lds + lane;
  // scratch of the cooperative refill (below): directions of up to kCoopRefill new systems on their way to the
  // owners (4.6 KB at 12 cameras, 6.1 KB at 16)
  // 0: every new lane sets up its own rays.  Past 12 cameras only the full-rig kernel (13-15 cameras on the masked
  // 16-camera kernel measured 1 % slower with it)
  constexpr int kCoopRefill = (NCAM <= 12 || (NCAM <= 16 && FULL)) ? 16 : 0;
  __shared__ __attribute__((aligned(16))) double refill_dir[kCoopRefill > 0 ? kCoopRefill * NCAM * 3 : 1];
  __shared__ long long refill_sys[kCoopRefill > 0 ? kCoopRefill : 1];
  __shared__ unsigned refill_ok[kCoopRefill > 0 ? kCoopRefill : 1];
  LmState st;
  int64_t s = -1;  // system this lane works on; -1 = idle
  // A finished system's outputs are written when the wave next refills (or at once when no refill will come):
  // the write-out block — address arithmetic, the division for the mean distance, four stores — otherwise runs
  // on almost every pass with a single lane active (a wave of 64 lanes sees about one finish per pass).  The
  // lane keeps its final state untouched meanwhile: idle lanes execute nothing that changes it.
  int64_t s_done = -1;
  bool finish_coop = false;
  bool exhausted = false;       // no system will arrive any more: global counter past the end, reservation empty
  bool claimed_all = false;     // the global counter has passed n_sys
  int64_t res_next = 0, res_end = 0, last_base = 0;  // this wave's reservation; last counter value seen
  bool wave_exact = LIT;  // FULL: this wave currently evaluates through the central-difference evaluator (see below)
  int waited = 0;
  st.need_J = false;
  // outputs of the lane's finished system s_done; its rays are still in place (idle lanes change nothing)
  auto write_done = [&] {
    out.write(s_done, st, out.err ? lm_sum_dist<NCAM, LIT>(org, rs, n_cam, st.tx, st.ty, st.tz) : 0.0);
  };
  for (;;) {
    // ---- refill idle lanes from the work counter (wave-uniform control flow)
    const unsigned long long idle_mask = __ballot(s < 0);
    const int n_idle = __popcll(idle_mask);
    if (!exhausted && n_idle >= refill_threshold) {
      // Systems are claimed from the global counter in chunks and handed out from the wave's reservation
      // [res_next, res_end): the device-scope atomic (a round trip to memory, and one address for every wave)
      // is then paid once per kLmChunk systems instead of once per refill.  In the last quarter of the batch
      // the chunk shrinks to what is needed, so no wave sits on a reservation while others run dry.
      const int64_t avail = res_end - res_next;
      int64_t first_new = 0;  // first system of a newly claimed chunk (lanes beyond the old reservation)
      int64_t new_next = res_next + n_idle, new_end = res_end;
      if (avail < n_idle) {
        const int64_t want = n_idle - avail;
        const int64_t chunk = (!claimed_all && 4 * last_base < 3 * n_sys && want < lm_chunk) ? lm_chunk : want;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)chunk);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
               (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)base);
        last_base = (int64_t)base;
        if (base + (unsigned long long)chunk >= (unsigned long long)n_sys) claimed_all = true;
        first_new = (int64_t)base;
        // the old reservation is used up by this refill; what remains is the rest of the new chunk
        new_end = (int64_t)base + chunk < n_sys ? (int64_t)base + chunk : n_sys;
        new_next = (int64_t)base + want < new_end ? (int64_t)base + want : new_end;
      }
      wave_exact = LIT;  // new systems arrive: try the branch-free evaluator again
      const bool took_new = avail < n_idle;
      if (s_done >= 0) {  // results of the systems that finished since the last refill
        write_done();
        s_done = -1;
      }
      // which idle lane adopts which system
      bool take = false;
      int64_t cand = n_sys;
      if (s < 0) {
        const int rank = __popcll(idle_mask & ((1ull << lane) - 1ull));
        cand = rank < avail ? res_next + rank : (took_new ? first_new + (rank - avail) : n_sys);
        take = cand < n_sys;
      }
      const unsigned long long new_mask = __ballot(take);
      const int n_new = __popcll(new_mask);
      int n_valid = 0;
      double x = 0, y = 0, z = 0;
      if constexpr (kCoopRefill > 0) {
        // ---- cooperative set-up of the new systems (round 3; up to 16 cameras).  A steady-state refill brings in a
        // handful of systems, and with every new lane setting up its own NCAM rays the wave executes the whole
        // set-up (47 instructions per camera) with those few lanes active — 7 % of the instructions at 12 % lane
        // utilisation.  Here, kCoopRefill new systems at a time, the wave spreads their rays over all 64 lanes (each
        // loads one detection and runs ray_dir), the directions travel through the block's LDS, and each owner
        // collects its NCAM directions.  Same function on the same inputs, start point summed by the owner in camera
        // order: nothing changes in the results (test_ray_kernel_forms_agree_bitwise).  The first fill of a wave (64
        // systems) takes four such rounds — the same ray work plus the collection.  1 M systems: 12 cameras 1.93 ->
        // 1.84 ms, 8 cameras 1.35 -> 1.31 ms.  At 16 cameras it first cost the second wave per SIMD (256 + 32
        // registers: 3.1 -> 4.1 ms) — the collection's origin reads, see below; with those from the kernel argument the
        // kernel stays at 250 registers.  Not in the 24 / 32-camera kernels, which have no register to spare.
        const int j_new = __popcll(new_mask & ((1ull << lane) - 1ull));
#pragma nounroll
        for (int j0 = 0; j0 < n_new; j0 += kCoopRefill) {  // wave-uniform
          const int m = n_new - j0 < kCoopRefill ? n_new - j0 : kCoopRefill;
          const bool mine = take && j_new >= j0 && j_new < j0 + m;
          if (mine) {
            refill_sys[j_new - j0] = cand;
            refill_ok[j_new - j0] = 0u;
          }
          __syncthreads();
          const int total = m * NCAM;
#pragma nounroll
          for (int it = lane; it < total; it += kWave) {
            const int jj = it / NCAM, c = it - jj * NCAM;
            double dx = 0.0, dy = 0.0, dz = 0.0;
            if (c < n_cam) {
              const double2 uv = pts[(int64_t)c * cam_stride + refill_sys[jj]];
              if (!is_missing(uv.x, uv.y)) {
                org.ray_dir(c, uv.x, uv.y, dx, dy, dz);
                atomicOr(&refill_ok[jj], 1u << c);
              }
            }
            double* o = refill_dir + 3 * it;
            o[0] = dx; o[1] = dy; o[2] = dz;
          }
          __syncthreads();
          if (mine) {
            s = cand;
            const unsigned valid = refill_ok[j_new - j0];
            rs.valid = valid;
            // (volatile reads: left to itself the compiler fetches all 3 NCAM doubles before it stores the first.
            // Past 12 cameras the origins come from the kernel argument — scalar loads — not from the LDS table: the
            // same doubles, but as 3 NCAM more volatile reads they were all issued before the first was consumed, and
            // with them the 16-camera kernel needed 256 + 32 registers, one wave per SIMD.  Up to 12 cameras the LDS
            // reads stay: the scalar form costs the 6-camera kernel its 168th register and with it the third wave.)
            const lds_vdouble* o = (const lds_vdouble*)refill_dir + 3 * NCAM * (j_new - j0);
#pragma unroll
            for (int c = 0; c < NCAM; ++c) {
              const double dx = o[3 * c], dy = o[3 * c + 1], dz = o[3 * c + 2];
              rs.set(c, dx, dy, dz);  // missing observation: (0, 0, 0), contributes exact zeros (lm_normal_full)
              const bool ok = (valid >> c) & 1u;
              n_valid += ok ? 1 : 0;
              // initial guess = mean of the valid rays' origins (src/RayTriangulator.cpp:90-98)
              if constexpr (NCAM > 12) {
                x += ok ? cams.c[c].ox : 0.0;
                y += ok ? cams.c[c].oy : 0.0;
                z += ok ? cams.c[c].oz : 0.0;
              } else {
                x += ok ? org.ox(c) : 0.0;
                y += ok ? org.oy(c) : 0.0;
                z += ok ? org.oz(c) : 0.0;
              }
            }
          }
          __syncthreads();  // the next round reuses the scratch
        }
      } else if (take) {
        s = cand;
        rs.valid = 0;
        // All detection loads of the system are issued before the first one is used.  With the load inside
        // the per-camera block below, every camera waited for its own HBM round trip (s_waitcnt vmcnt(0) per
        // camera: 16 serial round trips per refill).  Absent cameras (c >= n_cam) re-read row 0 — cheaper than
        // a branch that would split the loads over basic blocks again.  The registers are free here: the
        // evaluation's temporaries are not live during a refill — up to 16 cameras; the 24- and 32-camera
        // kernels load in batches of 16 (64 + 64 registers on top of the directions would spill).
        constexpr int kBatch = NCAM > 16 ? 16 : NCAM;
#pragma unroll
        for (int cb = 0; cb < NCAM; cb += kBatch) {
          double2 uvs[kBatch];
#pragma unroll
          for (int c = cb; c < cb + kBatch && c < NCAM; ++c)
            uvs[c - cb] = pts[(int64_t)(c < n_cam ? c : 0) * cam_stride + s];
#pragma unroll
          for (int c = cb; c < cb + kBatch && c < NCAM; ++c) {
            if (c < n_cam) {
              const double2 uv = uvs[c - cb];
              const bool ok = !is_missing(uv.x, uv.y);
              double dx, dy, dz;
              org.ray_dir(c, uv.x, uv.y, dx, dy, dz);
              if (!ok) dx = dy = dz = 0.0;  // contributes exact zeros (lm_normal_full)
              rs.set(c, dx, dy, dz);
              rs.valid |= ok ? (1u << c) : 0u;
              n_valid += ok ? 1 : 0;
              // initial guess = mean of the valid rays' origins (src/RayTriangulator.cpp:90-98)
              x += ok ? org.ox(c) : 0.0;
              y += ok ? org.oy(c) : 0.0;
              z += ok ? org.oz(c) : 0.0;
            } else {
              rs.set(c, 0.0, 0.0, 0.0);
            }
          }
        }
      }
      if (take) {
        st.n_valid = n_valid;
        if (n_valid < 2) {  // the reference throws for this frame; record it and move on
          atomicMin(first_bad, (unsigned long long)s);
          out.write_too_few(s);
          s = -1;
        } else {
          st.x = x / n_valid; st.y = y / n_valid; st.z = z / n_valid;
          st.iter = -1;  // marks "initial compute pending"
          st.need_J = true;
        }
      }
      res_next = new_next;
      res_end = new_end;
      exhausted = claimed_all && res_next >= res_end;
    }
    if (exhausted && s_done >= 0) {  // no refill will come: flush what is pending (wave-uniform condition first)
      write_done();
      s_done = -1;
    }
    if (__ballot(s >= 0) == 0ull) {
      if (exhausted) break;
      continue;  // everything this wave fetched was rejected; fetch again
    }
    if (SPEC && NCAM <= kCoopMaxCam && coop && exhausted &&
        __popcll(__ballot(s >= 0)) <= kWave / coop_group(NCAM)) {
      finish_coop = true;  // the cooperative drain runs AFTER the loop: called from inside, its by-reference use
      break;               // of the lane state made the compiler shuttle the state between two register sets
    }                      // on every pass (about 60 v_mov_b64 per pass)
    if (SPEC) {
      // ---- speculative form: every busy lane evaluates residuals AND the Jacobian at its trial point in
      // one fused pass (they share the cross product and the sqrt), so all lanes do identical work every
      // pass and nobody waits for a J phase; a rejected step wastes its Jacobian.
      if (s >= 0) {
        const bool init = st.iter < 0;
        double d0 = 0, d1 = 0, d2 = 0;
        if (!init) lm_trial<LIT>(st, d0, d1, d2);
        LmNormal E;
        bool redo = false;
        if (FULL && !LIT && !wave_exact) {
          bool small;
          E = lm_normal_full<NCAM>(org, rs, n_cam, st.x - d0, st.y - d1, st.z - d2, small);
          // some ray of some lane is closer than 0.1 mm: this wave repeats the pass, and runs the following
          // ones, through the evaluator that holds the exact central difference.  Wave-uniform, and the two
          // evaluators sit in sibling branches so neither lengthens the other's live ranges.
          if (__ballot(small) != 0ull) wave_exact = redo = true;
        } else if (FULL && NCAM >= 8) {  // (6 cameras: 168 -> 182 registers would cost the third wave per SIMD)
          bool unused;
          E = lm_normal_full<NCAM, true, LIT>(org, rs, n_cam, st.x - d0, st.y - d1, st.z - d2, unused);
        } else {
          E = lm_normal<NCAM, LIT>(org, rs, n_cam, st.x - d0, st.y - d1, st.z - d2);
        }
        bool finished = false, install = false;  // install: J^T J, J^T r of the evaluated point become the state's
        if (!redo) {
          if (init) {  // LMSolver::run prologue
            st.S = E.S; st.r2max = E.r2max;
            st.tx = st.x; st.ty = st.y; st.tz = st.z;
            st.D0 = E.a00; st.D1 = E.a11; st.D2 = E.a22;
            st.lambda = 1; st.lc = 0.75; st.iter = 0;
            install = true;
          } else {
            finished = lm_update<LIT>(st, d0, d1, d2, E.S, E.r2max, install);
          }
        }
        if (install) st.A = E;  // one install site: two made the compiler copy the state back and forth
        if (finished) {
          s_done = s;  // written at the next refill, or right below the loop head once the queue is exhausted
          s = -1;
        }
      }
      continue;
    }
    // ---- compute(x, r, J) where needed: first evaluation of a new system, or after an accept.
    // The reference's lambda schedule settles into a strict accept/reject alternation, so a lane
    // needs this (the expensive part) only every other pass, and with lanes in random phase the J pass
    // ran half empty (rocprofv3: 51 % of lanes active per VALU instruction).  Lanes that need J
    // therefore wait — at most max_wait passes — while the others take their cheap trial steps, until
    // >= j_quarters/4 of the busy lanes need J: the wave falls into phase.
    {
      const int n_busy = __popcll(__ballot(s >= 0));
      const int n_need = __popcll(__ballot(s >= 0 && st.need_J));
      const bool do_J = n_need > 0 && (4 * n_need >= j_quarters * n_busy || waited >= max_wait);
      if (do_J) {
        waited = 0;
        if (s >= 0 && st.need_J) {
          st.A = lm_normal<NCAM>(org, rs, n_cam, st.x, st.y, st.z);
          st.need_J = false;
          if (st.iter < 0) {  // LMSolver::run prologue
            st.S = st.A.S;
            st.r2max = st.A.r2max;
            st.tx = st.x; st.ty = st.y; st.tz = st.z;
            st.D0 = st.A.a00; st.D1 = st.A.a11; st.D2 = st.A.a22;
            st.lambda = 1;
            st.lc = 0.75;
            st.iter = 0;
          }
        }
      } else if (n_need > 0) {
        ++waited;
      }
    }
    // ---- one LM iteration for every busy lane whose J^T J is current
    if (s >= 0 && !st.need_J) {
      double d0, d1, d2;
      lm_trial(st, d0, d1, d2);
      const LmEval ed = lm_residuals<NCAM>(org, rs, n_cam, st.x - d0, st.y - d1, st.z - d2);
      bool accepted;
      bool finished = lm_update(st, d0, d1, d2, ed.S, ed.r2max, accepted);
      st.need_J = accepted;
      if (finished) {
        s_done = s;
        write_done();
        s_done = -1;
        s = -1;
        st.need_J = false;
      }
    }
  }
  if constexpr (SPEC && NCAM <= kCoopMaxCam) {
    if (finish_coop) lm_coop_finish<NCAM, LIT>(org, rs, st, s, n_cam, out);
  }
}

inline int lm_refill_threshold() { return knob(K_LM_REFILL, kRefillDefault, 1, 64); }

template <int NCAM, bool LIT = false>
hipError_t launch_ray_t(const DeviceCams& dc, const double* d_pts, int n_cam, int64_t cam_stride,
                        int64_t n_sys, double* d_xyz, double* d_err, int* d_iters, int* d_status,
                        unsigned long long* fb, hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  CamR<NCAM> cams;
  for (int c = 0; c < NCAM; ++c) cams.c[c] = dc.ray.c[c < n_cam ? c : 0];
  if constexpr (NCAM <= kCoopMaxCam) {
    // small batches: camera-per-lane layout from the start (ray_coop_kernel) while the systems, at 64 / G per
    // wave, fit about two waves per SIMD; above that the lane-per-system kernel's throughput wins
    const int64_t per_wave = kWave / coop_group(NCAM);
    // (the crossover sits at about 1024 waves — re-measured in round 4, when the kernel lost the inlined Jacobi sweeps
    // and went from two to three waves per SIMD, tools/ab_ray_small.py: 16 cameras 345 vs 412 us at 2000 systems,
    // 605 vs 463 us at 8000; 8 cameras 163 vs 204 us at 2000, level at 8000, 318 vs 281 us at 16000; 4 cameras level
    // at 16000, 1377 vs 1342 us at 32000)
    const int64_t coop_max = knob(K_LM_COOP_MAX, (int)(1024 * per_wave), 0, 1 << 30);
    if (n_sys <= coop_max && knob(K_LM_COOP, 1, 0, 1)) {
      const int64_t blocks = (n_sys + per_wave - 1) / per_wave;
      hipLaunchKernelGGL((ray_coop_kernel<NCAM, LIT>), dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kWave), 0,
                         stream, cams, n_cam, reinterpret_cast<const double2*>(d_pts), cam_stride, n_sys, d_xyz,
                         d_err, d_iters, d_status, fb);
      return hipGetLastError();
    }
  }
  unsigned long long* counter = dc.d_work_counter + (dc.lm_launches.fetch_add(1) % kLmCounters) * kLmCounterWords;
  hipError_t e = hipMemsetAsync(counter, 0, kLmCounterWords * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  // enough waves to fill the chip several times over, never more than the work needs
  const int64_t waves_needed = (n_sys + kWave - 1) / kWave;
  const int64_t max_waves = 256 * 4 * 8;
  const unsigned grid = (unsigned)(waves_needed < max_waves ? waves_needed : max_waves);
  const int chunk = knob(K_LM_CHUNK, kLmChunk, 1, 4096);  // systems claimed per atomic (A/B knob)
  const int coop = knob(K_LM_COOP, 1, 0, 1);  // cooperative drain (lm_coop_finish), <= 16 cameras
  // branch-free evaluator for full camera sets: 3-4 % faster from 6 cameras up, 2.5 % slower at 4 (A/B on
  // 1 M systems), so the small sets keep the masked form
  // 24 / 32 cameras: directions of the last kNlds cameras in LDS so that two waves fit a SIMD
  constexpr int kNlds = NCAM == 32 ? 12 : (NCAM == 24 ? 4 : 0);
  const double2* pts2 = reinterpret_cast<const double2*>(d_pts);
#define TDR_LM_LAUNCH_W(SPEC_, FULL_, NLDS_, WAVES_, A0, A1, A2)                                                  \
  hipLaunchKernelGGL((ray_kernel<NCAM, CamR<NCAM>, SPEC_, FULL_, NLDS_, LIT, WAVES_>), dim3(grid), dim3(kWave), 0,   \
                     stream, cams,                                                                                 \
                     n_cam, pts2, cam_stride, n_sys, d_xyz, d_err, d_iters, d_status, fb, counter, A0, A1, A2)
#define TDR_LM_LAUNCH(SPEC_, FULL_, NLDS_, A0, A1, A2) TDR_LM_LAUNCH_W(SPEC_, FULL_, NLDS_, 0, A0, A1, A2)
  // A/B, knob TDR_LM_NLDS != 0: the full-rig 8-camera kernel with the directions of its last FOUR
  // cameras in LDS, register-allocated for THREE waves per SIMD (168 VGPRs; 196 and two waves otherwise).  Measured in
  // round 5 (profiles/r05_lm8_nlds.json, 1 M systems with 5 % drop-outs): 1.209-1.241 ms against 1.255-1.268 ms
  // interleaved in one process, level under rocprofv3 (1.319 against 1.315 ms) — the third wave raises the VALU issue
  // fraction only from 0.48 to 0.50 because the build issues 4.7 % more VALU and 42 % more LDS instructions and waits
  // 2.6 x as long on LDS (the direction reads sit in front of every camera block).  Five or six cameras in LDS: 1.22 /
  // 1.25 ms.  Twelve cameras: six directions in LDS reach 168 registers only with 112 B of scratch, 2.70 against
  // 1.80 ms; with eight the block's LDS share caps the kernel at two waves anyway.  So the default stays in registers
  // and only the four-camera variant is kept as the A/B.  Same arithmetic, same bits
  // (tests/test_gpu_lm_builds.py).  Not in the literal build.
  if constexpr (NCAM == 8 && !LIT) {
    if (knob(K_LM_NLDS, 0, 0, 8) != 0 && n_cam == NCAM && knob(K_LM_FULL, 1, 0, 1)) {
      TDR_LM_LAUNCH_W(true, true, 4, 3, lm_refill_threshold(), chunk, coop);
      return hipGetLastError();
    }
  }
  // ONE launch: rank-deficient solves are closed forms inside the kernel (lm_cascade).  Round 3 parked such systems
  // for a second, eigen-solve launch here; a two-camera rig went straight to a standalone form of that kernel.
  if (kNlds > 0) {
    if (n_cam == NCAM && knob(K_LM_FULL, 1, 0, 1))
      TDR_LM_LAUNCH(true, true, kNlds, lm_refill_threshold(), chunk, coop);
    else
      TDR_LM_LAUNCH(true, false, kNlds, lm_refill_threshold(), chunk, coop);
  } else if (n_cam == NCAM && knob(K_LM_FULL, NCAM >= 6 ? 1 : 0, 0, 1)) {
    TDR_LM_LAUNCH(true, true, 0, lm_refill_threshold(), chunk, coop);
  } else {
    TDR_LM_LAUNCH(true, false, 0, lm_refill_threshold(), chunk, coop);
  }
#undef TDR_LM_LAUNCH
#undef TDR_LM_LAUNCH_W
  return hipGetLastError();
}

}  // namespace
}  // namespace tdr_amd
