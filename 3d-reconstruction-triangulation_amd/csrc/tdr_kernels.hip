// tdr_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the triangulation hot path.
//
// Mapping: one system (frame x drone) per lane, system index = global lane
// id.  The reference input layout pts[cam][sys] of 16-B (x,y) doubles
// (src/DetectionsContainer.cpp:145-171) makes lane i's read of pts[cam][s0+i] one fully
// coalesced global_load_dwordx4 (1 KiB per wave instruction) with no transposition.  Camera
// constants are wave-uniform: the matrix kernel stages P in LDS (broadcast ds_read_b128), the LM
// kernel takes the ray model from the kernarg segment (scalar loads).  The 3x3 normal-equation /
// LM state lives in registers; no MFMA (3x3 is not a dense contraction).
// Matrix path is HBM-bound: n_cam*16 B read + 24 B written per system.
#include "tdr_lm_device.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <hipcub/hipcub.hpp>

namespace tdr_amd {

// ---- the knob table (one per process; see tdr_device_common.h)
namespace {
constexpr const char* kKnobNames[K_COUNT] = {"TDR_LM_COOP", "TDR_LM_COOP_MAX", "TDR_LM_FULL", "TDR_LM_CHUNK",
                                              "TDR_LM_REFILL", "TDR_I16_X2", "TDR_SWEEP_HOST", "TDR_LM_LITERAL_FD",
                                              "TDR_LM_NLDS", "TDR_POINT_ARGS", "TDR_MATRIX_PAD_LDS"};
struct KnobTable {
  std::atomic<int> v[K_COUNT];
  KnobTable() {
    for (int k = 0; k < K_COUNT; ++k) {
      const char* e = getenv(kKnobNames[k]);
      v[k].store(e ? atoi(e) : kKnobUnset, std::memory_order_relaxed);
    }
  }
};
inline KnobTable& knob_table() {
  static KnobTable t;
  return t;
}
}  // namespace
int knob_raw(int k) { return knob_table().v[k].load(std::memory_order_relaxed); }

namespace {

// ------------------------------------------------------------------------------------------
// Matrix path: accumulate A^T A (6 unique) and A^T b (3) of the 2N x 3 inhomogeneous system
// (src/MatrixTriangulator.cpp:16-49) over the valid cameras.
// ------------------------------------------------------------------------------------------
struct NormalEq {
  double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
  double v0 = 0, v1 = 0, v2 = 0;
  int n = 0;
};

// Stage one wave's 64 x (x,y,z) through LDS so that the 1536 contiguous output bytes leave as
// 16-B-per-lane stores (1 KiB + 512 B wave instructions) instead of three 8-B stores at a 24-B
// lane stride.  Wave-private LDS region, no workgroup barrier.
template <bool NT = false>
__device__ __forceinline__ void store_xyz_staged(double* lds_wave, double* __restrict__ xyz,
                                                 int64_t wave_base, int64_t n_sys, int lane,
                                                 double x0, double x1, double x2) {
  lds_wave[lane * 3 + 0] = x0;
  lds_wave[lane * 3 + 1] = x1;
  lds_wave[lane * 3 + 2] = x2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int64_t left = n_sys - wave_base;
  int nd = (left >= kWave ? kWave : (int)left) * 3;  // doubles this wave owns
  double* out = xyz + wave_base * 3;
  const double2* l2 = reinterpret_cast<const double2*>(lds_wave);
#pragma unroll
  for (int i = lane; i < 96; i += kWave) {
    if (2 * i + 1 < nd) {
      if (NT) {
        dbl2 t;
        t.x = l2[i].x;
        t.y = l2[i].y;
        __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(out) + i);
      } else {
        reinterpret_cast<double2*>(out)[i] = l2[i];
      }
    } else if (2 * i < nd) {
      out[2 * i] = lds_wave[2 * i];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Matrix kernel.  Structure chosen from the first version's ISA dump: with
// P in SGPRs (kernarg) every row FMA had two scalar operands, and gfx950 allows one per VALU
// instruction, so the compiler emitted ~20 v_mov_b32 per camera; the accumulation compiled to
// mul+fma+add instead of FMA chains; and a 0/1 validity weight cost 6 extra fp64 multiplies per
// camera (787 instructions, ~2000 VALU cycles per wave).  Here (626 instructions, ~1230 cycles)
//   - the block stages P once in LDS; each camera's 12 doubles come back as six broadcast
//     ds_read_b128 (same address in every lane: conflict-free) on the LDS pipe, so all FMA
//     operands are VGPRs and no SGPR->VGPR copies are issued on the VALU;
//   - a missing observation is skipped by the EXEC mask (if (ok) {...}) instead of multiplied by 0;
//   - every accumulator is a pure fma chain.
// OUT_MODE 0: three 8-B stores per lane; 1: LDS-staged 16-B stores.
// NT_LD / NT_ST: nontemporal (streaming) cache policy — every byte is touched exactly once, and
// `nt` loads alone took the 8-camera kernel from 5.5 to 6.2 TB/s on MI355X.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_rows(NormalEq& q, const double* __restrict__ p, double u,
                                                  double v) {
  const double a0 = fma(-u, p[8], p[0]), a1 = fma(-u, p[9], p[1]), a2 = fma(-u, p[10], p[2]);
  const double c0 = fma(-v, p[8], p[4]), c1 = fma(-v, p[9], p[5]), c2 = fma(-v, p[10], p[6]);
  const double b0 = fma(u, p[11], -p[3]), b1 = fma(v, p[11], -p[7]);
  q.m00 = fma(a0, a0, q.m00); q.m00 = fma(c0, c0, q.m00);
  q.m01 = fma(a0, a1, q.m01); q.m01 = fma(c0, c1, q.m01);
  q.m02 = fma(a0, a2, q.m02); q.m02 = fma(c0, c2, q.m02);
  q.m11 = fma(a1, a1, q.m11); q.m11 = fma(c1, c1, q.m11);
  q.m12 = fma(a1, a2, q.m12); q.m12 = fma(c1, c2, q.m12);
  q.m22 = fma(a2, a2, q.m22); q.m22 = fma(c2, c2, q.m22);
  q.v0 = fma(a0, b0, q.v0); q.v0 = fma(c0, b1, q.v0);
  q.v1 = fma(a1, b0, q.v1); q.v1 = fma(c1, b1, q.v1);
  q.v2 = fma(a2, b0, q.v2); q.v2 = fma(c2, b1, q.v2);
}

__device__ __forceinline__ double residual_rows(const double* __restrict__ p, double u, double v,
                                                  double x0, double x1, double x2) {
  const double a0 = fma(-u, p[8], p[0]), a1 = fma(-u, p[9], p[1]), a2 = fma(-u, p[10], p[2]);
  const double c0 = fma(-v, p[8], p[4]), c1 = fma(-v, p[9], p[5]), c2 = fma(-v, p[10], p[6]);
  const double b0 = fma(u, p[11], -p[3]), b1 = fma(v, p[11], -p[7]);
  const double r0 = fma(a2, x2, fma(a1, x1, fma(a0, x0, -b0)));
  const double r1 = fma(c2, x2, fma(c1, x1, fma(c0, x0, -b1)));
  return fma(r1, r1, r0 * r0);
}

// ------------------------------------------------------------------------------------------
// Rank-deficient and ill-conditioned systems: the reference's cv::invert(A, DECOMP_SVD)
// (src/MatrixTriangulator.cpp:52-54) returns the Moore-Penrose solution for ANY A — singular values
// w_i <= 2 DBL_EPSILON sum(w) are dropped, so a duplicate camera, two cameras on one line of sight or
// collinear centres through the point give the minimum-norm point — while the normal equations above have no
// solution there, and lose eps * cond(A)^2 before that.  Lanes whose LDL^T pivots fall below kRankTol
// (tdr_math.h) of their diagonal therefore redo the solve here with the accuracy of an SVD of A itself:
// one-sided (Hestenes) Jacobi, the scheme OpenCV's JacobiSVD uses, but with A never stored — the rotations
// are accumulated in V (9 registers) and the rotated columns A V are re-generated row by row from P (LDS) and
// the detections (registers) whenever their inner products are needed.  Working on A V (not on A^T A)
// resolves a null direction down to w ~ eps |A|, where the Gram matrix stops at sqrt(eps) |A|.
//   rotation test  |p| <= 10 eps sqrt(a b), sweeps until one passes without a rotation (<= 40)
//   X = sum_{w_i > 2 eps sum w} v_i (u_i . b) / w_i,   (A V)_i = w_i u_i
// Cold path: only flagged lanes run it (one wave-uniform branch on the hot path).
// ------------------------------------------------------------------------------------------
// Register discipline: this code sits inside the streaming kernels, whose occupancy (six waves per SIMD at 8
// cameras) must not pay for it.  It therefore keeps only V (18 registers) and a handful of accumulators live:
// the camera loops are NOT unrolled, P rows come from memory (LDS) by a run-time index and the detections are
// re-read from global memory through `get_uv` instead of being held in the caller's register array.
struct Col3 {
  double x, y, z;
};

template <typename GetUV>
__device__ __forceinline__ void pinv_minimum_norm(const double* P_all, int n_cam, const GetUV& get_uv,
                                                   double& x0, double& x1, double& x2) {
  Col3 v0{1, 0, 0}, v1{0, 1, 0}, v2{0, 0, 1};  // right singular vectors (columns of V)
  const double kTol = 10.0 * kDblEps;
  // one Jacobi rotation of the rotated columns (A vi, A vj): a = |A vi|^2, b = |A vj|^2, p = (A vi).(A vj)
  auto rotate = [&](Col3& vi, Col3& vj) -> bool {
    double a = 0, b = 0, p = 0;
#pragma unroll 1
    for (int c = 0; c < n_cam; ++c) {
      double u, v;
      if (!get_uv(c, u, v)) continue;
      const double* P = P_all + 12 * c;
      const double r0 = fma(-u, P[8], P[0]), r1 = fma(-u, P[9], P[1]), r2 = fma(-u, P[10], P[2]);
      const double s0 = fma(-v, P[8], P[4]), s1 = fma(-v, P[9], P[5]), s2 = fma(-v, P[10], P[6]);
      const double ti = fma(r2, vi.z, fma(r1, vi.y, r0 * vi.x)), tj = fma(r2, vj.z, fma(r1, vj.y, r0 * vj.x));
      const double wi = fma(s2, vi.z, fma(s1, vi.y, s0 * vi.x)), wj = fma(s2, vj.z, fma(s1, vj.y, s0 * vj.x));
      a = fma(ti, ti, fma(wi, wi, a));
      b = fma(tj, tj, fma(wj, wj, b));
      p = fma(ti, tj, fma(wi, wj, p));
    }
    if (!(fabs(p) > kTol * sqrt(a * b))) return false;
    p *= 2;
    const double beta = a - b, gamma = hypot(p, beta);
    double cs, sn;
    if (beta < 0) {
      const double delta = (gamma - beta) * 0.5;
      sn = sqrt(delta / gamma);
      cs = p / (gamma * sn * 2);
    } else {
      cs = sqrt((gamma + beta) / (gamma * 2));
      sn = p / (gamma * cs * 2);
    }
    const Col3 ni{cs * vi.x + sn * vj.x, cs * vi.y + sn * vj.y, cs * vi.z + sn * vj.z};
    const Col3 nj{-sn * vi.x + cs * vj.x, -sn * vi.y + cs * vj.y, -sn * vi.z + cs * vj.z};
    vi = ni;
    vj = nj;
    return true;
  };
#pragma unroll 1
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool changed = rotate(v0, v1);
    changed |= rotate(v0, v2);
    changed |= rotate(v1, v2);
    if (!changed) break;
  }
  // w_i^2 = |A v_i|^2 and h_i = (A v_i) . b, one more pass over the rows
  double q0 = 0, q1 = 0, q2 = 0, h0 = 0, h1 = 0, h2 = 0;
#pragma unroll 1
  for (int c = 0; c < n_cam; ++c) {
    double u, v;
    if (!get_uv(c, u, v)) continue;
    const double* P = P_all + 12 * c;
    const double r0 = fma(-u, P[8], P[0]), r1 = fma(-u, P[9], P[1]), r2 = fma(-u, P[10], P[2]);
    const double s0 = fma(-v, P[8], P[4]), s1 = fma(-v, P[9], P[5]), s2 = fma(-v, P[10], P[6]);
    const double b0 = fma(u, P[11], -P[3]), b1 = fma(v, P[11], -P[7]);
    double t, w;
    t = fma(r2, v0.z, fma(r1, v0.y, r0 * v0.x)); w = fma(s2, v0.z, fma(s1, v0.y, s0 * v0.x));
    q0 = fma(t, t, fma(w, w, q0)); h0 = fma(t, b0, fma(w, b1, h0));
    t = fma(r2, v1.z, fma(r1, v1.y, r0 * v1.x)); w = fma(s2, v1.z, fma(s1, v1.y, s0 * v1.x));
    q1 = fma(t, t, fma(w, w, q1)); h1 = fma(t, b0, fma(w, b1, h1));
    t = fma(r2, v2.z, fma(r1, v2.y, r0 * v2.x)); w = fma(s2, v2.z, fma(s1, v2.y, s0 * v2.x));
    q2 = fma(t, t, fma(w, w, q2)); h2 = fma(t, b0, fma(w, b1, h2));
  }
  const double w0 = sqrt(q0), w1 = sqrt(q1), w2 = sqrt(q2);
  const double thr = 2 * kDblEps * (w0 + w1 + w2);  // SVD::backSubst's cut
  const double g0 = w0 > thr ? h0 / q0 : 0.0;
  const double g1 = w1 > thr ? h1 / q1 : 0.0;
  const double g2 = w2 > thr ? h2 / q2 : 0.0;
  x0 = fma(v2.x, g2, fma(v1.x, g1, v0.x * g0));
  x1 = fma(v2.y, g2, fma(v1.y, g1, v0.y * g0));
  x2 = fma(v2.z, g2, fma(v1.z, g1, v0.z * g0));
}

// IN16: the observations are two int16 (4 B, the narrow format of SURVEY.md §8d: the reference's detections
// are integer pixel coordinates), converted to double in registers; everything after the load is the same
// code, so the results are bit-identical to the double format on the same values.
template <int NCAM, bool WANT_ERR, int OUT_MODE, bool NT_LD, bool NT_ST, int BLOCK, bool IN16 = false>
__global__ void __launch_bounds__(BLOCK)
    matrix_kernel(const double* __restrict__ Pg, int n_cam, const void* __restrict__ pts_v,
                     int64_t cam_stride, int64_t n_sys, double* __restrict__ xyz,
                     double* __restrict__ err, unsigned long long* __restrict__ first_bad, int64_t sys_base) {
  // sys_base: index of this launch's first system in the caller's batch (a large batch is cut into several launches on
  // slices of the same arrays, launch_matrix_t); only the first-failing index needs it
  const double2* __restrict__ pts = reinterpret_cast<const double2*>(pts_v);
  __shared__ __attribute__((aligned(16))) double ldsP[NCAM * 12];
  __shared__ __attribute__((aligned(16))) double lds[OUT_MODE == 1 ? BLOCK * 3 : 2];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool active = s < n_sys;
  const int64_t sl = active ? s : n_sys - 1;  // clamp: every lane loads mapped memory
  // Up to 16 cameras every detection load is in flight before anything else.  The 24- and 32-camera kernels
  // would need 96 / 128 registers for that (three / two waves per SIMD, with no load in flight while both
  // compute): they keep a window of 16 detections and, as soon as camera c's has been read out, request camera
  // c + 16's into the same registers (kRotate; not for the variant that revisits the detections for the error
  // value, nor for the int16 format).
  constexpr bool kRotate = NCAM > 16 && !WANT_ERR && !IN16;
  constexpr int KB = kRotate ? 16 : NCAM;
  auto load_uv = [&](int c) -> double2 {
    if (IN16) {
      const int* src16 = reinterpret_cast<const int*>(pts_v) + ((int64_t)c * cam_stride + sl);
      const int w = NT_LD ? __builtin_nontemporal_load(src16) : *src16;
      return make_double2((double)(short)(w & 0xffff), (double)(short)(w >> 16));
    }
    const double2* src = pts + ((int64_t)c * cam_stride + sl);
    if (NT_LD) {
      dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(src));
      return make_double2(t.x, t.y);
    }
    return *src;
  };
  double2 uv[KB];
#pragma unroll
  for (int c = 0; c < KB; ++c)
    if (c < n_cam) uv[c] = load_uv(c);
  for (int i = threadIdx.x; i < n_cam * 12; i += BLOCK) ldsP[i] = Pg[i];
  __syncthreads();
  NormalEq q;
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam) {
      double p[12];
      const double2* lp = reinterpret_cast<const double2*>(ldsP + 12 * c);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double2 t = lp[k];
        p[2 * k] = t.x;
        p[2 * k + 1] = t.y;
      }
      const double u = uv[c % KB].x, v = uv[c % KB].y;
      if (kRotate && c + KB < NCAM) {
        if (c + KB < n_cam) uv[c % KB] = load_uv(c + KB);
      }
      if (!is_missing(u, v)) {
        q.n += 1;
        accumulate_rows(q, p, u, v);
      }
    }
  }
  double x0, x1, x2;
  const bool full_rank =
      solve_spd3(q.m00, q.m01, q.m02, q.m11, q.m12, q.m22, q.v0, q.v1, q.v2, x0, x1, x2);
  if (q.n < 2) {
    x0 = x1 = x2 = __builtin_nan("");
    if (active) atomicMin(first_bad, (unsigned long long)(s + sys_base));
  }
  // rank-deficient / ill-conditioned lanes: the reference's pseudo-inverse answer (pinv_minimum_norm above)
  const bool slow = q.n >= 2 && !full_rank;
  if (__ballot(slow) != 0ull) {
    if (slow)
      pinv_minimum_norm(ldsP, n_cam,
                        [&](int c, double& u, double& v) {
                          if (IN16) {
                            const int w = reinterpret_cast<const int*>(pts_v)[(int64_t)c * cam_stride + sl];
                            u = (double)(short)(w & 0xffff);
                            v = (double)(short)(w >> 16);
                          } else {
                            const double2 t = pts[(int64_t)c * cam_stride + sl];
                            u = t.x;
                            v = t.y;
                          }
                          return !is_missing(u, v);
                        },
                        x0, x1, x2);
  }
  if (WANT_ERR) {
    double ss = 0;
#pragma unroll
    for (int c = 0; c < NCAM; ++c) {
      if (c < n_cam) {
        double p[12];
        const double2* lp = reinterpret_cast<const double2*>(ldsP + 12 * c);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double2 t = lp[k];
          p[2 * k] = t.x;
          p[2 * k + 1] = t.y;
        }
        const double u = uv[c % KB].x, v = uv[c % KB].y;  // WANT_ERR: KB == NCAM
        if (!is_missing(u, v)) ss += residual_rows(p, u, v, x0, x1, x2);
      }
    }
    const double e = sqrt(ss / (double)(2 * q.n));
    if (active) {
      if (NT_ST) __builtin_nontemporal_store(e, err + s); else err[s] = e;
    }
  }
  if (OUT_MODE == 1) {
    const int64_t wave_base = (int64_t)blockIdx.x * BLOCK + wave * kWave;
    if (wave_base < n_sys)
      store_xyz_staged<NT_ST>(lds + wave * (kWave * 3), xyz, wave_base, n_sys, lane, x0, x1, x2);
  } else if (active) {
    if (NT_ST) {
      __builtin_nontemporal_store(x0, xyz + s * 3 + 0);
      __builtin_nontemporal_store(x1, xyz + s * 3 + 1);
      __builtin_nontemporal_store(x2, xyz + s * 3 + 2);
    } else {
      xyz[s * 3 + 0] = x0;
      xyz[s * 3 + 1] = x1;
      xyz[s * 3 + 2] = x2;
    }
  }
}

// A/B only (knob TDR_MATRIX_PAD_LDS, bytes): unused dynamic LDS per block, which caps the blocks resident on a CU — an
// occupancy sweep of the streaming kernels without touching their code (tools/ab_occupancy.py).  0 = none (default).
inline unsigned matrix_pad_lds() { return (unsigned)knob(K_MATRIX_PAD_LDS, 0, 0, 160 * 1024 - 4096); }

// systems per launch of a large batch: a multiple of every block size, and above C3's 10 M (which stays one launch)
constexpr int64_t kMatrixSlice = 10ll << 20;

// one slice of a large batch: the streaming configuration of variant 0 (nontemporal, 64-thread blocks from 6 cameras up)
template <int NCAM, bool WANT_ERR>
hipError_t launch_matrix_slice(const DeviceCams& dc, const double* d_pts, int n_cam, int64_t cam_stride, int64_t n_sys,
                               double* d_xyz, double* d_err, unsigned long long* d_first_bad, int64_t sys_base,
                               hipStream_t stream) {
  const double2* pts = reinterpret_cast<const double2*>(d_pts);
  constexpr int BLK = NCAM <= 5 ? 256 : 64;
  hipLaunchKernelGGL((matrix_kernel<NCAM, WANT_ERR, 0, true, true, BLK>), dim3((unsigned)((n_sys + BLK - 1) / BLK)),
                     dim3(BLK), matrix_pad_lds(), stream, dc.d_P, n_cam, pts, cam_stride, n_sys, d_xyz, d_err,
                     d_first_bad, sys_base);
  return hipGetLastError();
}

template <int NCAM, bool WANT_ERR>
hipError_t launch_matrix_t(const DeviceCams& dc, int variant, const double* d_pts, int n_cam,
                           int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                           unsigned long long* d_first_bad, hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  const double2* pts = reinterpret_cast<const double2*>(d_pts);
  // staged 16-B stores need a 16-B aligned result base
  if (variant == 2 && (reinterpret_cast<uintptr_t>(d_xyz) & 15) != 0) variant = 0;
  // Batches beyond kMatrixSlice systems go out as several launches on consecutive slices of the same arrays (in
  // place: cam_stride stays the caller's).  C4 — 40 M systems, 6.08 GB — ran the kernel 2-3 points of the HBM peak
  // below C3 as ONE launch (0.770 against 0.794 on one box, profiles/r04_c4_probe.json); cut into launches of ~10 M
  // systems it gains about 2 % (tools/c4_slices.py: 1.017 ms as one launch, 0.997 / 0.996 ms as four / eight,
  // interleaved in one process).  What remains below C3 is the device, not the launch: 0.794 next to C3's 0.808 in
  // profiles/r05_bench_default.json and 0.778 next to 0.800 in the driver's BENCH_r04.json, while the compute-free
  // stream of the same footprint loses 3-6 % from 1.3 GB to 5.4 GB in those very runs (the kernel stays above it).
  if (variant == 0 && n_sys > kMatrixSlice) {
    for (int64_t off = 0; off < n_sys; off += kMatrixSlice) {
      const int64_t m = n_sys - off < kMatrixSlice ? n_sys - off : kMatrixSlice;
      hipError_t e = launch_matrix_slice<NCAM, WANT_ERR>(dc, d_pts + 2 * off, n_cam, cam_stride, m, d_xyz + 3 * off,
                                                         d_err ? d_err + off : nullptr, d_first_bad, off, stream);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
#define TDR_MK(OUT, NTL, NTS, BLK)                                                               \
  hipLaunchKernelGGL((matrix_kernel<NCAM, WANT_ERR, OUT, NTL, NTS, BLK>),                        \
                     dim3((unsigned)((n_sys + BLK - 1) / BLK)), dim3(BLK), matrix_pad_lds(), stream, dc.d_P, \
                     n_cam, pts, cam_stride, n_sys, d_xyz, d_err, d_first_bad, (int64_t)0)
  switch (variant) {
    case 1: TDR_MK(0, false, false, 256); break;  // default cache policy
    case 2: TDR_MK(1, true, true, 256); break;    // LDS-staged 16-B result stores
    case 3: TDR_MK(0, true, true, 64); break;
    case 4: TDR_MK(0, true, true, 128); break;
    // A/B only (round 5, tools/ab_variants.py), 8 cameras without err — anything else takes the default below:
    // 5: nontemporal loads, DEFAULT-policy stores (lets the L2 merge the three 8-B stores of a 24-B row);
    // 6: LDS-staged 16-B nontemporal stores in 64-thread blocks (variant 2 has 256); 7: staged, default-policy stores
    case 5: case 6: case 7:
      if constexpr (NCAM == 8 && !WANT_ERR) {
        if (variant == 5) TDR_MK(0, true, false, 64);
        else if ((reinterpret_cast<uintptr_t>(d_xyz) & 15) != 0) TDR_MK(0, true, true, 64);
        else if (variant == 6) TDR_MK(1, true, true, 64);
        else TDR_MK(1, true, false, 64);
        break;
      }
      [[fallthrough]];
    default:  // 0: policy and block size picked from the measured A/B (MI355X)
      // - a batch that fits the 256 MiB Infinity Cache (producer -> consumer pipelines, repeated launches)
      //   is served from it under the default cache policy: 1M x 8 cameras 6 % faster than with `nt`;
      // - beyond that every byte is touched once: nontemporal loads + stores (+12 % at 10M systems), with
      //   64-thread blocks from 6 cameras up (+2-4 %; 256 stays better at 4 cameras).
      if ((uint64_t)n_sys * (uint64_t)(n_cam * 16 + 24) <= (192ull << 20)) TDR_MK(0, false, false, 256);
      else if (NCAM <= 5) TDR_MK(0, true, true, 256);
      else TDR_MK(0, true, true, 64);
      break;
  }
#undef TDR_MK
  return hipGetLastError();
}

// int16 format, K consecutive systems per lane (the launcher uses K = 2).  At 4 B per observation the one-system
// kernel is no longer HBM-bound (2.8 TB/s): every wave re-reads P from LDS (48 broadcast ds_read_b128 per wave at 8
// cameras) and the LDS pipe and the VALU each run at about half their ceilings with nothing to overlap them.  Here a
// lane loads 4 K bytes per camera — the observations of systems K s .. K s + K - 1 — and each P row fetched from LDS
// serves all of them, which divides the LDS traffic per system by K (+21 % at K = 2).
// Same accumulate_rows / solve_spd3 as matrix_kernel, hence the same bits.
// Requires cam_stride to be a multiple of K (4 K-byte aligned groups) and readable pad elements behind a ragged end.
// Round-3 measurements at 8 cameras x 10 M systems (tools/ab_i16.py; DESIGN.md §4.4, HISTORY.md §4.3): K = 4 with direct stores
// 0.29-0.30 ms (six 16-B stores at a 96-B lane stride: partial lines), K = 4 with STAGED stores 0.142, K = 2
// 0.139-0.145 either way; a grid-stride loop with the next group prefetched under the arithmetic 0.147-0.150 at every
// occupancy from two to five waves per SIMD, so latency is not what is left: 377 wave-VALU instructions per system
// (rocprofv3 SQ_INSTS_VALU) are 0.75-0.8 of the fp64 issue rate the chip sustains, with HBM at 0.5 of its peak.
template <int NCAM, bool WANT_ERR, bool NT, int BLOCK, int K, bool STAGED>
__global__ void __launch_bounds__(BLOCK)
    matrix_i16xk_kernel(const double* __restrict__ Pg, int n_cam, const int* __restrict__ pts,
                        int64_t group_stride, int64_t n_sys, double* __restrict__ xyz,
                        double* __restrict__ err, unsigned long long* __restrict__ first_bad) {
  // K (2 or 4) consecutive systems per lane: one 4 K-byte load per camera, each P row fetched from LDS serves all K
  // systems, 3 K / 2 sixteen-byte result stores.  group_stride: 4 K-byte groups per camera row.
  static_assert(K == 2 || K == 4, "two or four systems per lane");
  typedef int intk __attribute__((ext_vector_type(K)));
  __shared__ __attribute__((aligned(16))) double ldsP[NCAM * 12];
  const int64_t n_groups = (n_sys + K - 1) / K;
  const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;  // group index
  const bool active = s < n_groups;
  const int64_t sl = active ? s : n_groups - 1;
  intk w[NCAM];
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam) {
      const intk* src = reinterpret_cast<const intk*>(pts) + ((int64_t)c * group_stride + sl);
      if (NT) w[c] = __builtin_nontemporal_load(src);
      else w[c] = *src;
    }
  }
  for (int i = threadIdx.x; i < n_cam * 12; i += BLOCK) ldsP[i] = Pg[i];
  __syncthreads();
  NormalEq q[K];
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    if (c < n_cam) {
      double p[12];
      const double2* lp = reinterpret_cast<const double2*>(ldsP + 12 * c);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double2 t = lp[k];
        p[2 * k] = t.x;
        p[2 * k + 1] = t.y;
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const double u = (double)(short)(w[c][j] & 0xffff), v = (double)(short)(w[c][j] >> 16);
        if (!is_missing(u, v)) {
          q[j].n += 1;
          accumulate_rows(q[j], p, u, v);
        }
      }
    }
  }
  double x[3 * K];
  const double nan = __builtin_nan("");
  bool slow[K];
  bool any_slow = false, all_there = true;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool have = K * s + j < n_sys;  // the last lane's trailing systems may be padding
    all_there = all_there && have;
    const bool r = solve_spd3(q[j].m00, q[j].m01, q[j].m02, q[j].m11, q[j].m12, q[j].m22, q[j].v0, q[j].v1, q[j].v2,
                              x[3 * j], x[3 * j + 1], x[3 * j + 2]);
    if (q[j].n < 2) {
      x[3 * j] = x[3 * j + 1] = x[3 * j + 2] = nan;
      if (active && have) atomicMin(first_bad, (unsigned long long)(K * s + j));
    }
    slow[j] = q[j].n >= 2 && !r;  // rank-deficient / ill-conditioned: same cold path as matrix_kernel
    any_slow = any_slow || slow[j];
  }
  if (__ballot(any_slow) != 0ull) {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (slow[j])
        pinv_minimum_norm(ldsP, n_cam,
                          [&](int c, double& u, double& v) {
                            const int t = pts[((int64_t)c * group_stride + sl) * K + j];
                            u = (double)(short)(t & 0xffff);
                            v = (double)(short)(t >> 16);
                            return !is_missing(u, v);
                          },
                          x[3 * j], x[3 * j + 1], x[3 * j + 2]);
  }
  if (WANT_ERR) {
    double ss[K];
#pragma unroll
    for (int j = 0; j < K; ++j) ss[j] = 0;
#pragma unroll
    for (int c = 0; c < NCAM; ++c) {
      if (c < n_cam) {
        double p[12];
        const double2* lp = reinterpret_cast<const double2*>(ldsP + 12 * c);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double2 t = lp[k];
          p[2 * k] = t.x;
          p[2 * k + 1] = t.y;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const double u = (double)(short)(w[c][j] & 0xffff), v = (double)(short)(w[c][j] >> 16);
          if (!is_missing(u, v)) ss[j] += residual_rows(p, u, v, x[3 * j], x[3 * j + 1], x[3 * j + 2]);
        }
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < K; ++j)
        if (K * s + j < n_sys) err[K * s + j] = sqrt(ss[j] / (double)(2 * q[j].n));
    }
  }
  if (STAGED) {
    // one wave's 64 x 3 K doubles through its own LDS region, leaving as 1-KiB store instructions
    __shared__ __attribute__((aligned(16))) double stage[BLOCK * 3 * K];
    const int lane = threadIdx.x & (kWave - 1);
    double* mine = stage + (threadIdx.x - lane) * 3 * K;
    const int64_t wave_first = (s - lane) * K;  // first system of this wave
    if (wave_first < n_sys) {  // wave-uniform
#pragma unroll
      for (int k = 0; k < 3 * K / 2; ++k) {
        dbl2 t;
        t.x = x[2 * k];
        t.y = x[2 * k + 1];
        reinterpret_cast<dbl2*>(mine + lane * 3 * K)[k] = t;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t left = n_sys - wave_first;
      const int nd = (int)(left >= kWave * K ? kWave * K : left) * 3;  // doubles this wave owns
      double* out = xyz + wave_first * 3;
      if ((reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 3 * K / 2; ++k) {
          const int i = k * kWave + lane;
          if (2 * i + 1 < nd) {
            const dbl2 t = reinterpret_cast<const dbl2*>(mine)[i];
            if (NT) __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(out) + i);
            else reinterpret_cast<dbl2*>(out)[i] = t;
          } else if (2 * i < nd) {
            out[2 * i] = mine[2 * i];
          }
        }
      } else {
        for (int i = lane; i < nd; i += kWave) out[i] = mine[i];
      }
    }
  } else if (active) {
    double* out = xyz + 3 * K * s;  // 24 K bytes per lane, 16-B aligned when xyz is
    if (all_there && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 3 * K / 2; ++k) {
        dbl2 t;
        t.x = x[2 * k];
        t.y = x[2 * k + 1];
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(out) + k);
        else reinterpret_cast<dbl2*>(out)[k] = t;
      }
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j)
        if (K * s + j < n_sys) {
          out[3 * j] = x[3 * j]; out[3 * j + 1] = x[3 * j + 1]; out[3 * j + 2] = x[3 * j + 2];
        }
    }
  }
}

// int16 observation format: default / nontemporal policy by batch size, as above with 4 B per observation
template <int NCAM, bool WANT_ERR>
hipError_t launch_matrix_i16_t(const DeviceCams& dc, const void* d_pts16, int n_cam, int64_t cam_stride,
                               int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* d_first_bad,
                               hipStream_t stream) {
  (void)hipGetLastError();
  const bool small = (uint64_t)n_sys * (uint64_t)(n_cam * 4 + 24) <= (192ull << 20);
  // two systems per lane when the rows allow 8-B pair loads (even stride, a readable pad element behind an odd
  // n_sys, 8-B aligned base); knob TDR_I16_X2=0 forces one system per lane.  Results leave through LDS as 16-B-per-
  // lane stores up to 16 cameras (4 cameras x 10 M: 0.092 vs 0.099 ms direct, level at 8 and 16, 4 % slower at 32).
  const bool pairs = knob(K_I16_X2, 1, 0, 1) != 0 && n_sys >= 2 && (cam_stride & 1) == 0 &&
                     cam_stride >= n_sys + (n_sys & 1) && (reinterpret_cast<uintptr_t>(d_pts16) & 7) == 0;
  constexpr bool kStaged = NCAM <= 16;
#define TDR_MK16XK(NT, BLK)                                                                                    \
  hipLaunchKernelGGL((matrix_i16xk_kernel<NCAM, WANT_ERR, NT, BLK, 2, kStaged>),                                \
                     dim3((unsigned)(((n_sys + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, stream, dc.d_P, n_cam,   \
                     reinterpret_cast<const int*>(d_pts16), cam_stride / 2, n_sys, d_xyz, d_err, d_first_bad)
  if (pairs) {
    if (small) TDR_MK16XK(false, 256);
    else TDR_MK16XK(true, 256);
    return hipGetLastError();
  }
#undef TDR_MK16XK
#define TDR_MK16(NT, BLK)                                                                              \
  hipLaunchKernelGGL((matrix_kernel<NCAM, WANT_ERR, 0, NT, NT, BLK, true>),                              \
                     dim3((unsigned)((n_sys + BLK - 1) / BLK)), dim3(BLK), 0, stream, dc.d_P, n_cam, d_pts16, \
                     cam_stride, n_sys, d_xyz, d_err, d_first_bad, (int64_t)0)
  if (small) TDR_MK16(false, 256);
  else TDR_MK16(true, 256);
#undef TDR_MK16
  return hipGetLastError();
}

template <bool WANT_ERR>
hipError_t launch_matrix_i16_e(const DeviceCams& dc, const void* d_pts16, int n_cam, int64_t cam_stride,
                               int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* fb,
                               hipStream_t st) {
#define TDR_CASE(N)                                                                                   \
  if (n_cam <= N)                                                                                     \
    return launch_matrix_i16_t<N, WANT_ERR>(dc, d_pts16, n_cam, cam_stride, n_sys, d_xyz, d_err, fb, st);
  TDR_CASE(2) TDR_CASE(4) TDR_CASE(6) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16) TDR_CASE(24) TDR_CASE(32)
#undef TDR_CASE
  return hipErrorInvalidValue;
}

// double -> int16 observation format; counts values that are not integers in [-32768, 32767]
__global__ void __launch_bounds__(kBlock)
    pack_i16_kernel(const double2* __restrict__ pts, int n_cam, int64_t cam_stride, int64_t n_sys,
                    int* __restrict__ out, int64_t out_stride, unsigned long long* __restrict__ n_bad) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_sys) return;
  int bad = 0;
  for (int c = 0; c < n_cam; ++c) {
    const double2 uv = pts[(int64_t)c * cam_stride + s];
    const bool okx = uv.x >= -32768.0 && uv.x <= 32767.0 && uv.x == (double)(int)uv.x;
    const bool oky = uv.y >= -32768.0 && uv.y <= 32767.0 && uv.y == (double)(int)uv.y;
    bad += (okx && oky) ? 0 : 1;
    const int xi = okx ? (int)uv.x : -1, yi = oky ? (int)uv.y : -1;
    out[(int64_t)c * out_stride + s] = (xi & 0xffff) | (yi << 16);
  }
  if (bad) atomicAdd(n_bad, (unsigned long long)bad);
}

// int16 observation format -> double rows (the inverse of pack_i16_kernel; exact: the values are integers).
// Used by the host-buffer pipeline when detections travel over the link as int16 pairs but the kernel that
// consumes them (the LM path) reads doubles.
__global__ void __launch_bounds__(kBlock)
    unpack_i16_kernel(const int* __restrict__ in, int n_cam, int64_t in_stride, int64_t n_sys,
                      double2* __restrict__ pts, int64_t cam_stride) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_sys) return;
  for (int c = 0; c < n_cam; ++c) {
    const int v = __builtin_nontemporal_load(&in[(int64_t)c * in_stride + s]);
    double2 uv;
    uv.x = (double)(short)(v & 0xffff);
    uv.y = (double)(v >> 16);
    pts[(int64_t)c * cam_stride + s] = uv;
  }
}

template <bool WANT_ERR>
hipError_t launch_matrix_e(const DeviceCams& dc, int variant, const double* d_pts, int n_cam,
                           int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                           unsigned long long* fb, hipStream_t st) {
#define TDR_CASE(N)                                                                              \
  if (n_cam <= N)                                                                                \
    return launch_matrix_t<N, WANT_ERR>(dc, variant, d_pts, n_cam, cam_stride, n_sys, d_xyz,     \
                                        d_err, fb, st);
  TDR_CASE(2) TDR_CASE(3) TDR_CASE(4) TDR_CASE(5) TDR_CASE(6) TDR_CASE(7) TDR_CASE(8)
  TDR_CASE(10) TDR_CASE(12) TDR_CASE(16) TDR_CASE(24) TDR_CASE(32)
#undef TDR_CASE
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Closed-form ray-closest-point (round 5; SURVEY.md §7.1-6: "an analytic-J / closed-form variant as cross-check").
// What RayTriangulator::triangulatePoint minimises (src/RayTriangulator.cpp:83-107 with the rays of
// src/Triangulator.cpp:15-55) is an exact quadratic, sum_i |d_i x (x - o_i)|^2, so its minimiser solves
//     H x = g,   H = sum_i (|d_i|^2 I - d_i d_i^T),   g = sum_i (|d_i|^2 I - d_i d_i^T) o_i
// — one streaming pass, n_cam x 16 B read + 24 B written per system: the HBM-bound form of the ray path.  It is NOT
// what the LM entry points return on runs that stop early (those return the LM iterate), so it is a separate entry
// point (tdr_ray_closest_point_points[_device]) and a separate bench row, never a replacement.
//
// One system per lane, all detection loads in flight first (nontemporal: every byte is touched once), like
// matrix_kernel.  Ray set-up folded per camera: with the raw quaternion q the reference's rotated unit vector is
// d = M v / |v| (M = |q|^2 R(q), v = (x, y, dpix) affine in the pixel), so
//     e = M v = u A + v B + C   (A, B, C: three constant 3-vectors per camera, CfCam)
//     |d|^2 = |q|^4 = D,        |d|^2 I - d d^T = D I - (D / |e|^2) e e^T
// — 9 multiply-adds and one reciprocal per ray instead of the two Hamilton products, the normalisation and the
// divisions of make_ray() (47 instructions).  Directions differ from make_ray()'s by a few ulp.
// The camera constants are read through a uniform pointer (scalar loads into SGPRs, 128 B per camera): every
// multiply-add below is written with at most ONE constant operand, gfx950's limit per VALU instruction, so no
// SGPR -> VGPR copies are issued (the first matrix kernel's mistake, see matrix_kernel) and the LDS pipe — 8 broadcast
// ds_read_b128 per camera per wave would be 60 % of the kernel's HBM time — stays out of the picture.
// err (optional): mean distance of the solution from the rays, sum_i |d_i x (x - o_i)| / N — RayClosestPoint's
// `error` at the minimiser (src/RayTriangulator.cpp:21-26) — in a second pass that re-derives e from the
// detections still in registers.
// Rank-deficient H (all rays parallel to rounding): the minimiser is a line; the point returned is the one an LM run
// with the reference's DECOMP_EIG pseudo-inverse steps converges to from its start point x0 = mean of the ray
// origins (src/RayTriangulator.cpp:90-98): x0 - pinv(H) (H x0 - g), pinv by rank through lm_cascade.  Cold path.
// ------------------------------------------------------------------------------------------
template <typename GetUV>
__device__ __forceinline__ void cf_rank_deficient(const CfCam* __restrict__ cams, int n_cam, const GetUV& get_uv,
                                                  double h00, double h01, double h02, double h11, double h12,
                                                  double h22, double g0, double g1, double g2, int n_valid,
                                                  double& x0, double& x1, double& x2) {
  double sx = 0, sy = 0, sz = 0;
#pragma unroll 1
  for (int c = 0; c < n_cam; ++c) {
    double u, v;
    if (!get_uv(c, u, v)) continue;
    sx += cams[c].ox; sy += cams[c].oy; sz += cams[c].oz;
  }
  const double m0 = sx / n_valid, m1 = sy / n_valid, m2 = sz / n_valid;
  const double r0 = fma(h00, m0, fma(h01, m1, h02 * m2)) - g0;
  const double r1 = fma(h01, m0, fma(h11, m1, h12 * m2)) - g1;
  const double r2 = fma(h02, m0, fma(h12, m1, h22 * m2)) - g2;
  double d0, d1, d2;
  lm_cascade<false>(h00, h01, h02, h11, h12, h22, r0, r1, r2, n_valid, d0, d1, d2);
  x0 = m0 - d0; x1 = m1 - d1; x2 = m2 - d2;
}

// e = u A + v B + C with one constant operand per instruction (see above)
__device__ __forceinline__ void cf_dir(const CfCam& k, double u, double v, double& ex, double& ey, double& ez) {
  ex = fma(u, k.ax, v * k.bx) + k.cx;
  ey = fma(u, k.ay, v * k.by) + k.cy;
  ez = fma(u, k.az, v * k.bz) + k.cz;
}

// 1 / b: v_rcp_f64 + two Newton steps (<= 1 ulp); b = |e|^2 > 0 for every pixel (M is invertible, v has z = d > 0)
__device__ __forceinline__ double cf_rcp(double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = fma(fma(-b, y, 1.0), y, y);
  return fma(fma(-b, y, 1.0), y, y);
}

// No control flow per camera.  A missing observation — the reference's (-1) sentinel, or a camera beyond n_cam, whose
// registers are set to the sentinel instead of being loaded — runs the same arithmetic on its (finite) pixel with the
// weight D = |d|^2 multiplied by 0: every term it adds to H and g is then an exact 0 (records of unused cameras hold
// C = (0, 0, 1), so |e|^2 = 1 there; tdr_create).  With an exec-masked block per camera (the first version) the
// constants' scalar loads sat inside the block, each followed by its own s_waitcnt: 16 serial round trips per wave,
// and no two cameras' dependent chains could overlap.
template <int NCAM, bool WANT_ERR, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    ray_closest_point_kernel(const CfCam* __restrict__ cams, int n_cam, const double2* __restrict__ pts,
                             int64_t cam_stride, int64_t n_sys, double* __restrict__ xyz, double* __restrict__ err,
                             unsigned long long* __restrict__ first_bad, int64_t sys_base) {
  const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool active = s < n_sys;
  const int64_t sl = active ? s : n_sys - 1;  // clamp: every lane loads mapped memory
  // as in matrix_kernel: up to 16 cameras every detection load is in flight before anything else; the 24- and
  // 32-camera kernels keep a window of 16 and re-request into a register pair as soon as it has been read out
  constexpr bool kRotate = NCAM > 16 && !WANT_ERR;
  constexpr int KB = kRotate ? 16 : NCAM;
  auto load_uv = [&](int c) -> double2 {
    if (c >= n_cam) return make_double2(-1.0, -1.0);  // wave-uniform
    const double2* src = pts + ((int64_t)c * cam_stride + sl);
    if (NT) {
      dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(src));
      return make_double2(t.x, t.y);
    }
    return *src;
  };
  double2 uv[KB];
#pragma unroll
  for (int c = 0; c < KB; ++c) uv[c] = load_uv(c);
  double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0, dsum = 0;
  int n = 0;
  // kGroup cameras per basic block: the always-true test on an opaque operand only fences instruction selection (see
  // lm_normal_full).  As ONE block the 13 x NCAM constants were all requested up front: 340 SGPRs spilled to VGPR lanes
  // at 16 cameras, twice the instructions.  Per block: the records of its cameras (2 x 26 SGPRs) are requested at the
  // top, one wait, then two interleaved dependent chains.
  int nc = NCAM;
  constexpr int kGroup = 2;
#pragma unroll
  for (int c0 = 0; c0 < NCAM; c0 += kGroup) {
    asm volatile("" : "+s"(nc));
    if (c0 < nc) {
#pragma unroll
      for (int c = c0; c < c0 + kGroup && c < NCAM; ++c) {
        const double u = uv[c % KB].x, v = uv[c % KB].y;
        if (kRotate && c + KB < NCAM) uv[c % KB] = load_uv(c + KB);
        const bool ok = !is_missing(u, v);
        const CfCam& k = cams[c];
        double ex, ey, ez;
        cf_dir(k, u, v, ex, ey, ez);
        const double n2 = fma(ez, ez, fma(ey, ey, ex * ex));
        const double okf = ok ? 1.0 : 0.0;  // one v_cndmask on the high word
        const double D = okf * k.D;
        const double sc = cf_rcp(n2) * D;  // |d|^2 / |e|^2
        const double tx = sc * ex, ty = sc * ey, tz = sc * ez;
        h00 = fma(-tx, ex, h00); h01 = fma(-tx, ey, h01); h02 = fma(-tx, ez, h02);
        h11 = fma(-ty, ey, h11); h12 = fma(-ty, ez, h12); h22 = fma(-tz, ez, h22);
        const double eo = fma(ex, k.ox, fma(ey, k.oy, ez * k.oz));
        g0 = fma(D, k.ox, fma(-tx, eo, g0)); g1 = fma(D, k.oy, fma(-ty, eo, g1)); g2 = fma(D, k.oz, fma(-tz, eo, g2));
        dsum += D;
        n += ok ? 1 : 0;
      }
    }
  }
  h00 += dsum; h11 += dsum; h22 += dsum;
  double x0, x1, x2;
  const bool full_rank = solve_spd3(h00, h01, h02, h11, h12, h22, g0, g1, g2, x0, x1, x2);
  if (n < 2) {  // the reference throws for this frame (src/RayTriangulator.cpp:72-74)
    x0 = x1 = x2 = __builtin_nan("");
    if (active) atomicMin(first_bad, (unsigned long long)(s + sys_base));
  }
  const bool slow = n >= 2 && !full_rank;
  if (__ballot(slow) != 0ull) {
    if (slow)
      cf_rank_deficient(cams, n_cam,
                        [&](int c, double& u, double& v) {
                          const double2 t = pts[(int64_t)c * cam_stride + sl];
                          u = t.x;
                          v = t.y;
                          return !is_missing(u, v);
                        },
                        h00, h01, h02, h11, h12, h22, g0, g1, g2, n, x0, x1, x2);
  }
  if (WANT_ERR) {
    double sum = 0;
#pragma unroll
    for (int c0 = 0; c0 < NCAM; c0 += kGroup) {
      asm volatile("" : "+s"(nc));
      if (c0 < nc) {
#pragma unroll
        for (int c = c0; c < c0 + kGroup && c < NCAM; ++c) {
          const double u = uv[c % KB].x, v = uv[c % KB].y;  // WANT_ERR: KB == NCAM
          const bool ok = !is_missing(u, v);
          const CfCam& k = cams[c];
          double ex, ey, ez;
          cf_dir(k, u, v, ex, ey, ez);
          const double n2 = fma(ez, ez, fma(ey, ey, ex * ex));
          const double wx = x0 - k.ox, wy = x1 - k.oy, wz = x2 - k.oz;
          const double cx = fma(ey, wz, -(ez * wy)), cy = fma(ez, wx, -(ex * wz)), cz = fma(ex, wy, -(ey * wx));
          const double c2 = fma(cz, cz, fma(cy, cy, cx * cx));
          // |d x w|^2 = (D / |e|^2) |e x w|^2
          const double r = sqrt_pos((c2 * cf_rcp(n2)) * k.D);
          sum += ok ? r : 0.0;
        }
      }
    }
    const double e = sum / (double)n;
    if (active) {
      if (NT) __builtin_nontemporal_store(e, err + s); else err[s] = e;
    }
  }
  if (active) {
    if (NT) {
      __builtin_nontemporal_store(x0, xyz + s * 3 + 0);
      __builtin_nontemporal_store(x1, xyz + s * 3 + 1);
      __builtin_nontemporal_store(x2, xyz + s * 3 + 2);
    } else {
      xyz[s * 3 + 0] = x0;
      xyz[s * 3 + 1] = x1;
      xyz[s * 3 + 2] = x2;
    }
  }
}

template <int NCAM, bool WANT_ERR>
hipError_t launch_ray_cf_t(const DeviceCams& dc, const double* d_pts, int n_cam, int64_t cam_stride, int64_t n_sys,
                           double* d_xyz, double* d_err, unsigned long long* d_first_bad, hipStream_t stream) {
  (void)hipGetLastError();
  // cache policy and block size by the same rule as the matrix path (launch_matrix_t): a batch that fits the
  // Infinity Cache keeps the default policy, beyond that nontemporal; large batches go out in slices
  const bool small = (uint64_t)n_sys * (uint64_t)(n_cam * 16 + 24) <= (192ull << 20);
  constexpr int BLK = NCAM <= 5 ? 256 : 64;
  for (int64_t off = 0; off < n_sys; off += kMatrixSlice) {
    const int64_t m = n_sys - off < kMatrixSlice ? n_sys - off : kMatrixSlice;
    const double2* pts = reinterpret_cast<const double2*>(d_pts + 2 * off);
    double* e = d_err ? d_err + off : nullptr;
    if (small)
      hipLaunchKernelGGL((ray_closest_point_kernel<NCAM, WANT_ERR, false, 256>), dim3((unsigned)((m + 255) / 256)),
                         dim3(256), 0, stream, dc.d_cf, n_cam, pts, cam_stride, m, d_xyz + 3 * off, e, d_first_bad, off);
    else
      hipLaunchKernelGGL((ray_closest_point_kernel<NCAM, WANT_ERR, true, BLK>), dim3((unsigned)((m + BLK - 1) / BLK)),
                         dim3(BLK), 0, stream, dc.d_cf, n_cam, pts, cam_stride, m, d_xyz + 3 * off, e, d_first_bad, off);
    hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) return rc;
  }
  return hipSuccess;
}

template <bool WANT_ERR>
hipError_t launch_ray_cf_e(const DeviceCams& dc, const double* d_pts, int n_cam, int64_t cam_stride, int64_t n_sys,
                           double* d_xyz, double* d_err, unsigned long long* fb, hipStream_t st) {
#define TDR_CASE(N) \
  if (n_cam <= N) return launch_ray_cf_t<N, WANT_ERR>(dc, d_pts, n_cam, cam_stride, n_sys, d_xyz, d_err, fb, st);
  TDR_CASE(2) TDR_CASE(3) TDR_CASE(4) TDR_CASE(6) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16) TDR_CASE(24) TDR_CASE(32)
#undef TDR_CASE
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Single-system kernels: per-observation constants in device memory (CamPointPair semantics,
// src/Triangulator.h:11-15).  One lane does the work; these exist so triangulatePoint() runs
// the same device arithmetic as the batch path.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void matrix_point_solve(const double* __restrict__ Pobs, const double* __restrict__ uv,
                                                   int n_obs, double* __restrict__ out4) {
  NormalEq q;
  for (int i = 0; i < n_obs; ++i) accumulate_rows(q, Pobs + 12 * i, uv[2 * i], uv[2 * i + 1]);
  double x0, x1, x2;
  if (!solve_spd3(q.m00, q.m01, q.m02, q.m11, q.m12, q.m22, q.v0, q.v1, q.v2, x0, x1, x2))
    pinv_minimum_norm(Pobs, n_obs,
                      [&](int c, double& u, double& v) {
                        u = uv[2 * c];
                        v = uv[2 * c + 1];
                        return true;
                      },
                      x0, x1, x2);
  double ss = 0;
  for (int i = 0; i < n_obs; ++i)
    ss += residual_rows(Pobs + 12 * i, uv[2 * i], uv[2 * i + 1], x0, x1, x2);
  out4[0] = x0;
  out4[1] = x1;
  out4[2] = x2;
  out4[3] = sqrt(ss / (double)(2 * n_obs));
}

__global__ void matrix_point_kernel(const double* __restrict__ Pobs, const double* __restrict__ uv,
                                    int n_obs, double* __restrict__ out4) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  matrix_point_solve(Pobs, uv, n_obs, out4);
}

// The same solve with the observation's constants and pixels IN THE KERNEL ARGUMENT (3.5 KB at 32 observations, under
// the 4 KB segment) and the four result doubles written straight into a host-coherent pinned slot, followed by a
// sequence word: the call is one launch and one wait, with no copy in either direction (round 5; round 4's call was
// copy up, launch, copy down, synchronise — the reference's caller issues one such call per candidate combination,
// src/DroneClassifier.cpp:180-181).  out4 / seq_word are device addresses of pinned host memory.
struct PointArgs {  // == tdr_amd::PointArgs (tdr_kernels.h)
  double P[kMaxCam][12];
  double uv[kMaxCam][2];
};
__device__ __forceinline__ void publish_result(unsigned long long* seq_word, unsigned long long seq) {
  __threadfence_system();  // the result doubles are visible to the host before the sequence word is
  __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(kWave)
    matrix_point_arg_kernel(const PointArgs a, int n_obs, double* __restrict__ out4,
                            unsigned long long* __restrict__ seq_word, unsigned long long seq) {
  // The wave fetches the whole argument in ONE round trip (each lane a few doubles) and parks it in LDS; lane 0 then
  // solves from there.  Read by lane 0 alone, observation by observation, every camera cost its own trip to the
  // kernarg segment: 0.47 us per observation (9.7 us per call at two observations, 12.5 at eight).
  __shared__ double sP[kMaxCam * 12];
  __shared__ double sUV[kMaxCam * 2];
  const double* gp = &a.P[0][0];
  const double* gu = &a.uv[0][0];
  for (int i = threadIdx.x; i < n_obs * 12; i += kWave) sP[i] = gp[i];
  if ((int)threadIdx.x < n_obs * 2) sUV[threadIdx.x] = gu[threadIdx.x];
  __syncthreads();
  if (threadIdx.x != 0) return;
  matrix_point_solve(sP, sUV, n_obs, out4);
  publish_result(seq_word, seq);
}

// ------------------------------------------------------------------------------------------
// Batched combination scoring — the hot loop of DroneClassifier::fillCombinationQueue
// (src/DroneClassifier.cpp:156-198), all frames in one launch.
//
// The reference walks a prefix tree over cameras (digit 0 = "no detection", k = detection k-1),
// triangulates every node with >= 2 chosen detections and cuts the subtree of a node whose error
// exceeds the threshold.  Whether a LEAF (full combination) is pushed therefore depends only on its
// own digits: every prefix of it with >= 2 detections must have err <= error_max.  That makes the
// leaves independent: one lane per leaf, normal equations grown camera by camera, the prefix tests
// evaluated on the way (a dead lane stops early), survivors appended through one atomic counter and
// sorted by leaf index on the host (leaf index order == the reference's push order).
// ------------------------------------------------------------------------------------------
struct Candidate {  // == tdr_candidate (include/tdr_amd.h)
  int frame;
  short n_used;
  short flags;  // TDR_CANDIDATE_*
  long long leaf;  // index of the combination inside its frame's tree, camera 0 most significant
  double x, y, z, err;
};

template <int NCAM>
__global__ void __launch_bounds__(kBlock)
    score_leaves_kernel(const double* __restrict__ Pg, int n_cam, int n_frames,
                        const long long* __restrict__ frame_leaf_offset,
                        const int* __restrict__ det_count, const long long* __restrict__ det_offset,
                        const double2* __restrict__ dets, double error_max, int min_cameras,
                        long long n_leaves, Candidate* __restrict__ out,
                        unsigned long long* __restrict__ keys, long long cap,
                        unsigned long long* __restrict__ n_out) {
  __shared__ __attribute__((aligned(16))) double ldsP[NCAM * 12];
  for (int i = threadIdx.x; i < n_cam * 12; i += kBlock) ldsP[i] = Pg[i];
  __syncthreads();
  const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (g >= n_leaves) return;
  // frame of this leaf: last f with frame_leaf_offset[f] <= g
  int lo = 0, hi = n_frames;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (frame_leaf_offset[mid] <= g) lo = mid; else hi = mid;
  }
  const int f = lo;
  long long rem = g - frame_leaf_offset[f];
  const long long leaf = rem;
  int digit[NCAM];
#pragma unroll
  for (int c = NCAM - 1; c >= 0; --c) {
    if (c < n_cam) {
      const int size = det_count[(long long)f * n_cam + c] + 1;
      digit[c] = (int)(rem % size);
      rem /= size;
    } else {
      digit[c] = 0;
    }
  }
  double uu[NCAM], vv[NCAM];
  NormalEq q;
  double x0 = 0, x1 = 0, x2 = 0, err = 0;
  bool alive = true;
#pragma unroll
  for (int c = 0; c < NCAM; ++c) {
    uu[c] = vv[c] = 0;
    if (c < n_cam && alive && digit[c] > 0) {
      const double2 d = dets[det_offset[(long long)f * n_cam + c] + digit[c] - 1];
      uu[c] = d.x;
      vv[c] = d.y;
      double p[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) p[k] = ldsP[12 * c + k];
      q.n += 1;
      accumulate_rows(q, p, d.x, d.y);
      if (q.n >= 2) {  // the reference triangulates this prefix node (src/DroneClassifier.cpp:168-181)
        if (!solve_spd3(q.m00, q.m01, q.m02, q.m11, q.m12, q.m22, q.v0, q.v1, q.v2, x0, x1, x2)) {
          // rank-deficient prefix (duplicate camera, two cameras on one line of sight): the reference's pseudo-inverse
          // answer, as matrix_kernel gives it — the tree sweep solves its nodes through that kernel, and both routes
          // must keep the same survivors.  Cold: the detections are found again from the leaf number
          // instead of indexing the register arrays by a run-time camera.
          const int last = c;
          pinv_minimum_norm(ldsP, last + 1, [&](int k, double& u, double& v) {
            long long r = leaf;
            for (int j = n_cam - 1; j > k; --j) r /= det_count[(long long)f * n_cam + j] + 1;
            const int dk = (int)(r % (det_count[(long long)f * n_cam + k] + 1));
            if (dk == 0) return false;
            const double2 d = dets[det_offset[(long long)f * n_cam + k] + dk - 1];
            u = d.x; v = d.y;
            return true;
          }, x0, x1, x2);
        }
        double ss = 0;
#pragma unroll
        for (int k = 0; k <= c; ++k) {
          if (digit[k] > 0) {
            double pk[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) pk[j] = ldsP[12 * k + j];
            ss += residual_rows(pk, uu[k], vv[k], x0, x1, x2);
          }
        }
        err = sqrt(ss / (double)(2 * q.n));
        if (err > error_max) alive = false;  // cut (src/DroneClassifier.cpp:185-187)
      }
    }
  }
  if (alive && q.n >= 2 && q.n >= min_cameras) {  // push (src/DroneClassifier.cpp:190-197)
    const unsigned long long slot = atomicAdd(n_out, 1ull);
    if ((long long)slot < cap) {
      Candidate cnd;
      cnd.frame = f;
      cnd.n_used = (short)q.n;
      cnd.flags = 0;
      cnd.leaf = leaf;
      cnd.x = x0; cnd.y = x1; cnd.z = x2; cnd.err = err;
      out[slot] = cnd;
      keys[slot] = (unsigned long long)g;  // global leaf index: ascending == (frame, leaf) order
    }
  }
}

// ------------------------------------------------------------------------------------------
// Breadth-first sweep of the reference's prefix tree ON THE DEVICE (DroneClassifier::fillCombinationQueue with
// its Iterator / cut(), src/DroneClassifier.cpp:43-85,156-198) — the scoring route of the ray triangulator and of
// trees too large to enumerate leaf by leaf.  The node list of one camera depth lives in HBM, sorted by
// (frame, prefix); per depth:
//   sizes    each node's number of children (detections of camera L in its frame, + 1 for "no detection")
//   scan     exclusive sum -> where each node's children go (hipCUB)
//   expand   children written in (parent, digit) order, which keeps the list sorted; a child that adds a
//            detection and then holds >= 2 of them is marked: the reference triangulates exactly these nodes
//   scan     marked children -> slots of ONE batch
//   gather   the batch's detections as pts[camera][slot], unchosen cameras = (-1,-1)
//   solve    the batch kernel of the triangulator (matrix_kernel / the LM kernel), unchanged
//   prune    err > error_max cuts the node with its whole subtree (:185-187); survivors keep point and error
//   scan + compact   the next depth's node list
// The host only reads two counters per depth (the sizes of the next launches / buffers); no node, detection or
// result crosses PCIe until the final list of full combinations comes back.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    sweep_sizes_kernel(const Candidate* __restrict__ nodes, long long n, const int* __restrict__ det_count, int n_cam,
                       int L, long long* __restrict__ sizes) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;
  sizes[i] = i < n ? (long long)det_count[(long long)nodes[i].frame * n_cam + L] + 1 : 0;  // sizes[n] = 0: scan total
}

__global__ void __launch_bounds__(kBlock)
    sweep_expand_kernel(const Candidate* __restrict__ parents, long long n, const long long* __restrict__ offsets,
                        const int* __restrict__ det_count, int n_cam, int L, Candidate* __restrict__ children,
                        long long* __restrict__ marks, int* __restrict__ overflow) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) {
    if (i == n) marks[offsets[n]] = 0;  // terminator of the marks array (scan total)
    return;
  }
  const Candidate nd = parents[i];
  const long long size = (long long)det_count[(long long)nd.frame * n_cam + L] + 1;
  if (nd.leaf > (0x7fffffffffffffffll - size) / size) {
    *overflow = 1;  // the prefix index no longer fits 63 bits
    return;
  }
  const long long base = offsets[i];
  for (long long d = 0; d < size; ++d) {
    Candidate ch = nd;
    ch.leaf = nd.leaf * size + d;
    long long mark = 0;
    if (d > 0) {
      ch.n_used = (short)(nd.n_used + 1);
      mark = ch.n_used >= 2 ? 1 : 0;  // the reference triangulates this node (src/DroneClassifier.cpp:168-181)
    }
    children[base + d] = ch;
    marks[base + d] = mark;
  }
}

__global__ void __launch_bounds__(kBlock)
    sweep_gather_kernel(const Candidate* __restrict__ children, long long n_children,
                        const long long* __restrict__ slot, const int* __restrict__ det_count,
                        const long long* __restrict__ det_offset, const double2* __restrict__ dets, int n_cam, int L,
                        long long stride, double2* __restrict__ pts) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_children) return;
  const long long e = slot[i];
  if (slot[i + 1] == e) return;  // not marked
  const Candidate ch = children[i];
  long long p = ch.leaf;
  for (int c = n_cam - 1; c > L; --c) pts[(long long)c * stride + e] = make_double2(-1.0, -1.0);
  for (int c = L; c >= 0; --c) {  // digits of cameras 0..L, camera L least significant
    const long long fc = (long long)ch.frame * n_cam + c;
    const long long size = (long long)det_count[fc] + 1;
    const long long digit = p % size;
    p /= size;
    pts[(long long)c * stride + e] = digit > 0 ? dets[det_offset[fc] + digit - 1] : make_double2(-1.0, -1.0);
  }
}

// lm_status (ray triangulator; may be null): the TDR_LM_* word of every solved node.  A node solved by a run that
// did not end at the minimiser carries an `err` no other implementation reproduces, so its prune
// decision is not a property of the reference: a survivor inherits kCandFlagged (and hands it to its subtree), a
// cut node is reported in `cuts` (appended through `n_cuts`) with its depth.
constexpr short kCandFlagged = 1, kCandFlaggedCut = 2;  // TDR_CANDIDATE_* (include/tdr_amd.h)
__global__ void __launch_bounds__(kBlock)
    sweep_prune_kernel(Candidate* __restrict__ children, long long n_children, const long long* __restrict__ slot,
                       const double* __restrict__ xyz, const double* __restrict__ err, double error_max,
                       const int* __restrict__ lm_status, int depth, Candidate* __restrict__ cuts, long long cuts_cap,
                       unsigned long long* __restrict__ n_cuts, long long* __restrict__ alive) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i > n_children) return;
  if (i == n_children) {
    alive[i] = 0;
    return;
  }
  const long long e = slot[i];
  long long keep = 1;
  if (slot[i + 1] != e) {
    const double er = err[e];
    const bool flagged = lm_status && (lm_status[e] & (kLmCap | kLmNotConverged)) != 0;
    if (er > error_max) {  // cut (src/DroneClassifier.cpp:185-187): the reference's predicate, a NaN is not cut
      keep = 0;
      if (flagged && n_cuts) {
        const unsigned long long at = atomicAdd(n_cuts, 1ull);
        if ((long long)at < cuts_cap) {
          Candidate c = children[i];
          c.flags = (short)(kCandFlaggedCut | (depth << 8));
          c.x = xyz[3 * e]; c.y = xyz[3 * e + 1]; c.z = xyz[3 * e + 2]; c.err = er;
          cuts[at] = c;
        }
      }
    } else {
      children[i].x = xyz[3 * e];
      children[i].y = xyz[3 * e + 1];
      children[i].z = xyz[3 * e + 2];
      children[i].err = er;
      if (flagged) children[i].flags |= kCandFlagged;
    }
  }
  alive[i] = keep;
}

// final push rule (src/DroneClassifier.cpp:190-197) as a keep flag
__global__ void __launch_bounds__(kBlock)
    sweep_final_kernel(const Candidate* __restrict__ nodes, long long n, int min_cameras, long long* __restrict__ keep) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;
  keep[i] = (i < n && nodes[i].n_used >= 2 && nodes[i].n_used >= min_cameras) ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock)
    sweep_compact_kernel(const Candidate* __restrict__ in, long long n, const long long* __restrict__ pos,
                         Candidate* __restrict__ out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] != pos[i]) out[pos[i]] = in[i];
}

__global__ void __launch_bounds__(kBlock)
    sweep_roots_kernel(Candidate* __restrict__ nodes, int n_frames) {
  const int f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= n_frames) return;
  Candidate c;
  c.frame = f;
  c.n_used = 0;
  c.flags = 0;
  c.leaf = 0;
  c.x = c.y = c.z = c.err = 0.0;
  nodes[f] = c;
}

// ------------------------------------------------------------------------------------------
// Calibration streams (not part of the path): what this box's HBM delivers for the matrix kernel's
// access shape, so the roofline fraction can be read against a known-good reference measured on
// the same device (guide §5.4 rule 10).  mode 0: read-only, `streams` concurrent 16-B/lane
// nontemporal streams per lane (like the n_cam detection rows) + one 8-B store per lane;
// mode 1: 16-B/lane nontemporal copy.
// ------------------------------------------------------------------------------------------
template <int STREAMS>
__global__ void __launch_bounds__(kBlock)
    stream_read_kernel(const dbl2* __restrict__ src, int64_t stride, int64_t n, double* __restrict__ sink) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  dbl2 v[STREAMS];
#pragma unroll
  for (int k = 0; k < STREAMS; ++k) v[k] = __builtin_nontemporal_load(src + k * stride + i);
  double acc = 0;
#pragma unroll
  for (int k = 0; k < STREAMS; ++k) acc += v[k].x + v[k].y;
  __builtin_nontemporal_store(acc, sink + i);
}

// mode 2 / 3 (round 5): the matrix kernel's own access pattern at `STREAMS` cameras in 64-thread blocks — STREAMS
// nontemporal 16-B rows per lane — with nothing written (mode 2: what the HBM delivers for the read side alone) or
// with three nontemporal 8-B stores per lane at a 24-B stride (mode 3: the kernel's memory traffic, byte for byte,
// without its arithmetic — its "traffic twin").  n must be a multiple of 64.
template <int STREAMS, bool STORE>
__global__ void __launch_bounds__(kWave)
    stream_twin_kernel(const dbl2* __restrict__ src, int64_t stride, int64_t n, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (i >= n) return;
  dbl2 v[STREAMS];
#pragma unroll
  for (int k = 0; k < STREAMS; ++k) v[k] = __builtin_nontemporal_load(src + k * stride + i);
  double a = 0, b = 0, c = 0;
#pragma unroll
  for (int k = 0; k < STREAMS; ++k) { a += v[k].x; b += v[k].y; c = fma(v[k].x, v[k].y, c); }
  if (STORE) {
    __builtin_nontemporal_store(a, out + 3 * i);
    __builtin_nontemporal_store(b, out + 3 * i + 1);
    __builtin_nontemporal_store(c, out + 3 * i + 2);
  } else if (a + b + c == 1.2345e301) {  // never true: keeps the loads alive
    out[0] = a;
  }
}

__global__ void __launch_bounds__(kBlock)
    stream_copy_kernel(const dbl2* __restrict__ src, dbl2* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// ------------------------------------------------------------------------------------------
// Triangulator::getDistFromRay (src/Triangulator.cpp:57-61), batched
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    dist_from_ray_kernel(const RayCam* __restrict__ cams, int n_cam,
                         const int32_t* __restrict__ cam_idx,
                         const double2* __restrict__ uv, const double* __restrict__ point, int64_t n,
                         double* __restrict__ dist) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int c = cam_idx[i];
  if (c < 0 || c >= n_cam) {
    dist[i] = __builtin_nan("");
    return;
  }
  double2 p = uv[i];
  Ray r = make_ray(cams[c], p.x, p.y);
  dist[i] = dist_to_ray(r, point[3 * i], point[3 * i + 1], point[3 * i + 2]);
}

// one distance with explicit camera constants (the static Triangulator::getDistFromRay has no context)
// seq_word != nullptr: `out` is a host-coherent pinned slot, the sequence word is published behind it (see
// matrix_point_arg_kernel)
__global__ void dist_from_ray_point_kernel(const RayCam cam, double u, double v, double px, double py,
                                           double pz, double* __restrict__ out,
                                           unsigned long long* __restrict__ seq_word, unsigned long long seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Ray r = make_ray(cam, u, v);
  out[0] = dist_to_ray(r, px, py, pz);
  if (seq_word) publish_result(seq_word, seq);
}

// ------------------------------------------------------------------------------------------
// Synthetic detection stream.  Counter-based RNG: every value is a pure
// function of (seed, global system index, camera, draw), so shards of one stream agree.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) {  // (0,1]
  return ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}

__global__ void __launch_bounds__(kBlock)
    synth_kernel(const double* __restrict__ Pd, const RayCam* __restrict__ Rd, int n_cam,
                 uint64_t seed,
                 int n_drones, int64_t sys_offset, int64_t n_sys, int64_t cam_stride,
                 double noise_px, double dropout_p, int round_to_int, int ray_model,
                 double2* __restrict__ pts, double* __restrict__ truth) {
  const int64_t sl = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (sl >= n_sys) return;
  const int64_t g = sys_offset + sl;
  const int64_t frame = g / n_drones;
  const int drone = (int)(g % n_drones);
  // smooth bounded path: per-drone phases/frequencies, step <= ~45 mm/frame
  const uint64_t hd = mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(drone + 1)));
  const double ph0 = 6.283185307179586 * u01(mix64(hd + 1));
  const double ph1 = 6.283185307179586 * u01(mix64(hd + 2));
  const double ph2 = 6.283185307179586 * u01(mix64(hd + 3));
  const double f = (double)frame;
  const double X = 1500.0 * sin(0.0170 * f + ph0) + 300.0 * sin(0.0071 * f + ph1);
  const double Y = 1500.0 * sin(0.0130 * f + ph1) + 300.0 * sin(0.0093 * f + ph2);
  const double Z = 1250.0 + 800.0 * sin(0.0110 * f + ph2);
  if (truth) {
    truth[3 * sl] = X;
    truth[3 * sl + 1] = Y;
    truth[3 * sl + 2] = Z;
  }
  const uint64_t hs = mix64(seed + 0x632BE59BD9B4E019ull * (uint64_t)(g + 1));
  // two cameras that always stay valid
  int keep0 = (int)(mix64(hs + 11) % (uint64_t)n_cam);
  int keep1 = (int)(mix64(hs + 12) % (uint64_t)(n_cam - 1));
  if (keep1 >= keep0) ++keep1;
  for (int c = 0; c < n_cam; ++c) {
    double u, v;
    if (ray_model) {
      const RayCam rc = Rd[c];
      // inverse of make_ray: world -> camera by conj(q) . (0,w) . q, then pinhole with focal d
      double wx = X - rc.ox, wy = Y - rc.oy, wz = Z - rc.oz;
      double qw = rc.qw, qx = -rc.qx, qy = -rc.qy, qz = -rc.qz;
      double tw = -qx * wx - qy * wy - qz * wz;
      double tx = qw * wx + qy * wz - qz * wy;
      double ty = qw * wy - qx * wz + qz * wx;
      double tz = qw * wz + qx * wy - qy * wx;
      double cx = -tw * qx + tx * qw - ty * qz + tz * qy;
      double cy = -tw * qy + tx * qz + ty * qw - tz * qx;
      double cz = -tw * qz - tx * qy + ty * qx + tz * qw;
      double xn = cx / cz * rc.d, yn = cy / cz * rc.d;
      u = (xn / rc.aspect + 1) * rc.width * 0.5 - 0.5;
      v = (yn + 1) * rc.height * 0.5 - 0.5;
    } else {
      const double* p = Pd + 12 * c;
      double w = p[8] * X + p[9] * Y + p[10] * Z + p[11];
      u = (p[0] * X + p[1] * Y + p[2] * Z + p[3]) / w;
      v = (p[4] * X + p[5] * Y + p[6] * Z + p[7]) / w;
    }
    const uint64_t hc = mix64(hs ^ (0xA0761D6478BD642Full * (uint64_t)(c + 1)));
    if (noise_px > 0) {  // Box-Muller
      double r = sqrt(-2.0 * log(u01(mix64(hc + 1)))) * noise_px;
      double a = 6.283185307179586 * u01(mix64(hc + 2));
      u += r * cos(a);
      v += r * sin(a);
    }
    if (round_to_int) {
      u = rint(u);
      v = rint(v);
    }
    bool drop = (c != keep0) && (c != keep1) && (u01(mix64(hc + 3)) <= dropout_p);
    if (drop) u = v = -1.0;
    pts[(int64_t)c * cam_stride + sl] = make_double2(u, v);
  }
}

}  // namespace

// ============================================================================== launchers
RayCam make_raycam(const double* a) {
  RayCam c;
  c.qw = a[0]; c.qx = a[1]; c.qy = a[2]; c.qz = a[3];
  c.ox = a[4]; c.oy = a[5]; c.oz = a[6];
  c.d = 1 / tan(a[7] * 0.0174533 / 2);  // src/Triangulator.cpp:34 (truncated constant kept)
  c.width = a[8];
  c.height = a[9];
  c.aspect = a[8] / a[9];
  return c;
}

// make_ray() (tdr_math.h; src/Triangulator.cpp:15-55) as an affine map of the pixel: with the raw quaternion q,
// q (0, v) conj(q) = M v, M = |q|^2 R(q); x = aspect (2 (u + .5) / width - 1), y = 2 (v + .5) / height - 1, z = d.
CfCam make_cfcam(const RayCam& rc) {
  const double w = rc.qw, x = rc.qx, y = rc.qy, z = rc.qz;
  const double M[3][3] = {{w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)},
                          {2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)},
                          {2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z}};
  const double ax = 2 * rc.aspect / rc.width, bx = rc.aspect * (1 / rc.width - 1);
  const double ay = 2 / rc.height, by = 1 / rc.height - 1;
  CfCam k;
  k.ax = ax * M[0][0]; k.ay = ax * M[1][0]; k.az = ax * M[2][0];
  k.bx = ay * M[0][1]; k.by = ay * M[1][1]; k.bz = ay * M[2][1];
  k.cx = bx * M[0][0] + by * M[0][1] + rc.d * M[0][2];
  k.cy = bx * M[1][0] + by * M[1][1] + rc.d * M[1][2];
  k.cz = bx * M[2][0] + by * M[2][1] + rc.d * M[2][2];
  k.ox = rc.ox; k.oy = rc.oy; k.oz = rc.oz;
  const double nq = w * w + x * x + y * y + z * z;
  k.D = nq * nq;
  k.pad[0] = k.pad[1] = k.pad[2] = 0.0;
  return k;
}

hipError_t launch_ray_closest_point(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride,
                                    int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* d_first_bad,
                                    hipStream_t stream) {
  if (n_sys == 0) return hipSuccess;
  if (d_err) return launch_ray_cf_e<true>(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_first_bad, stream);
  return launch_ray_cf_e<false>(cams, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_first_bad, stream);
}

hipError_t launch_matrix(const DeviceCams& cams, int variant, const double* d_pts, int n_cam_used,
                         int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                         unsigned long long* d_first_bad, hipStream_t stream) {
  if (n_sys == 0) return hipSuccess;
  if (d_err)
    return launch_matrix_e<true>(cams, variant, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err,
                                 d_first_bad, stream);
  return launch_matrix_e<false>(cams, variant, d_pts, n_cam_used, cam_stride, n_sys, d_xyz, d_err,
                                d_first_bad, stream);
}

hipError_t launch_matrix_i16(const DeviceCams& cams, const void* d_pts16, int n_cam_used, int64_t cam_stride,
                             int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* d_first_bad,
                             hipStream_t stream) {
  if (n_sys == 0) return hipSuccess;
  if (d_err)
    return launch_matrix_i16_e<true>(cams, d_pts16, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_first_bad,
                                     stream);
  return launch_matrix_i16_e<false>(cams, d_pts16, n_cam_used, cam_stride, n_sys, d_xyz, d_err, d_first_bad,
                                    stream);
}

hipError_t launch_pack_i16(const double* d_pts, int n_cam, int64_t cam_stride, int64_t n_sys, void* d_out,
                           int64_t out_stride, unsigned long long* d_n_bad, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_sys == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_i16_kernel, dim3((unsigned)((n_sys + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const double2*>(d_pts), n_cam, cam_stride, n_sys,
                     reinterpret_cast<int*>(d_out), out_stride, d_n_bad);
  return hipGetLastError();
}

hipError_t launch_unpack_i16(const void* d_in, int n_cam, int64_t in_stride, int64_t n_sys, double* d_pts,
                             int64_t cam_stride, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_sys == 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_i16_kernel, dim3((unsigned)((n_sys + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const int*>(d_in), n_cam, in_stride, n_sys,
                     reinterpret_cast<double2*>(d_pts), cam_stride);
  return hipGetLastError();
}

hipError_t launch_matrix_point(const double* d_Pobs, const double* d_uv, int n_obs, double* d_out4,
                               hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  hipLaunchKernelGGL(matrix_point_kernel, dim3(1), dim3(kWave), 0, stream, d_Pobs, d_uv, n_obs,
                     d_out4);
  return hipGetLastError();
}

// Calibration, not part of the path: the sustained fp64 FMA issue rate.  Every lane runs 16 independent FMA
// chains for `iters` rounds (16 * iters v_fma_f64 per lane, nothing else in the loop but the counter), so
// time / (waves * 16 * iters) is the cost of one wave-level fp64 instruction at the clock the chip actually
// holds under vector-fp64 load — the number the LM kernel's "VALU issue slots" have to be priced with.
__global__ void __launch_bounds__(kBlock)
    fma_peak_kernel(double* __restrict__ out, int iters, double a, double b) {
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = (double)(threadIdx.x + k);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = fma(x[k], a, b);
  }
  double sum = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += x[k];
  if (sum == 12345.678) out[0] = sum;  // keeps the chains alive; practically never true
}

// Test hook, not part of the path: the LM loop's DECOMP_EIG restatement (sym3_eig_pinv) on n symmetric matrices, one
// per lane.  in[i] = {m00, m01, m02, m11, m12, m22, v0, v1, v2}; out[i] = {x0, x1, x2 (= pinv(M) v), max_k |pinv(M)_kk|}.
__global__ void __launch_bounds__(kBlock)
    eig3_debug_kernel(const double* __restrict__ in, int64_t n, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double* m = in + 9 * i;
  double x0, x1, x2;
  sym3_eig_solve(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], x0, x1, x2);
  out[4 * i + 0] = x0; out[4 * i + 1] = x1; out[4 * i + 2] = x2;
  out[4 * i + 3] = sym3_eig_max_inv_diag(m[0], m[1], m[2], m[3], m[4], m[5]);
}

hipError_t launch_eig3_debug(const double* d_in, int64_t n, double* d_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(eig3_debug_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_in, n,
                     d_out);
  return hipGetLastError();
}

// Test hook (tdr_debug_lm_cascade_device): the LM loop's eigen-free restatement of cv::solve / cv::invert(DECOMP_EIG)
// (lm_cascade), one matrix per lane, as a system of n_rays rays meets it.
// out[i] = {x0, x1, x2 (= pinv(M) v), max_k |pinv(M)_kk|, the rank it decided on}.
__global__ void __launch_bounds__(kBlock)
    cascade_debug_kernel(const double* __restrict__ in, int64_t n, int n_rays, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double* m = in + 9 * i;
  double x0 = 0, x1 = 0, x2 = 0, mx = 0, u1, u2;
  const int rank = lm_cascade<false>(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], n_rays, x0, x1, x2);
  lm_cascade<true>(m[0], m[1], m[2], m[3], m[4], m[5], 0.0, 0.0, 0.0, n_rays, mx, u1, u2);
  out[5 * i + 0] = x0; out[5 * i + 1] = x1; out[5 * i + 2] = x2; out[5 * i + 3] = mx;
  out[5 * i + 4] = (double)rank;
}

hipError_t launch_cascade_debug(const double* d_in, int64_t n, int n_rays, double* d_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cascade_debug_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_in, n,
                     n_rays, d_out);
  return hipGetLastError();
}

hipError_t launch_fma_peak(double* d_out, int blocks, int iters, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(fma_peak_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, d_out, iters, 0.999999, 1e-6);
  return hipGetLastError();
}

hipError_t launch_stream(int mode, int streams, const double* d_src, double* d_dst, int64_t n_lanes,
                         hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  if (n_lanes <= 0) return hipSuccess;
  const dim3 grid((unsigned)((n_lanes + kBlock - 1) / kBlock)), block(kBlock);
  const dbl2* src = reinterpret_cast<const dbl2*>(d_src);
  if (mode == 2 || mode == 3) {
    const dim3 g64((unsigned)((n_lanes + kWave - 1) / kWave)), b64(kWave);
#define TDR_TWIN(S)                                                                                                  \
  if (mode == 2) hipLaunchKernelGGL((stream_twin_kernel<S, false>), g64, b64, 0, stream, src, n_lanes, n_lanes, d_dst); \
  else hipLaunchKernelGGL((stream_twin_kernel<S, true>), g64, b64, 0, stream, src, n_lanes, n_lanes, d_dst)
    if (streams == 4) { TDR_TWIN(4); } else if (streams == 16) { TDR_TWIN(16); } else { TDR_TWIN(8); }
#undef TDR_TWIN
  } else if (mode == 1) {
    hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, stream, src, reinterpret_cast<dbl2*>(d_dst), n_lanes);
  } else if (streams == 4) {
    hipLaunchKernelGGL((stream_read_kernel<4>), grid, block, 0, stream, src, n_lanes, n_lanes, d_dst);
  } else if (streams == 16) {
    hipLaunchKernelGGL((stream_read_kernel<16>), grid, block, 0, stream, src, n_lanes, n_lanes, d_dst);
  } else {
    hipLaunchKernelGGL((stream_read_kernel<8>), grid, block, 0, stream, src, n_lanes, n_lanes, d_dst);
  }
  return hipGetLastError();
}

// Sort the appended survivors by their global leaf index on the device (rocPRIM radix sort through
// hipCUB): the atomic append order is arbitrary, the reference's push order is ascending leaf.
// Two-call protocol like hipCUB's: d_temp == nullptr returns the scratch size.
hipError_t sort_candidates(void* d_temp, size_t& temp_bytes, const unsigned long long* d_keys_in,
                           unsigned long long* d_keys_out, const void* d_in, void* d_out, long long n,
                           hipStream_t stream) {
  return hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys_in, d_keys_out,
                                            reinterpret_cast<const Candidate*>(d_in),
                                            reinterpret_cast<Candidate*>(d_out), (int)n, 0, 64, stream);
}

// ---- device tree sweep (launchers; node records are Candidate = tdr_candidate, 48 B)
static inline dim3 grid_for(long long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

hipError_t sweep_scan(void* d_temp, size_t& temp_bytes, const long long* d_in, long long* d_out, long long n,
                      hipStream_t stream) {
  return hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_in, d_out, (int)n, stream);
}
hipError_t launch_sweep_roots(void* d_nodes, int n_frames, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_roots_kernel, grid_for(n_frames), dim3(kBlock), 0, st, reinterpret_cast<Candidate*>(d_nodes),
                     n_frames);
  return hipGetLastError();
}
hipError_t launch_sweep_sizes(const void* d_nodes, long long n, const int* d_det_count, int n_cam, int L,
                              long long* d_sizes, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_sizes_kernel, grid_for(n + 1), dim3(kBlock), 0, st,
                     reinterpret_cast<const Candidate*>(d_nodes), n, d_det_count, n_cam, L, d_sizes);
  return hipGetLastError();
}
hipError_t launch_sweep_expand(const void* d_parents, long long n, const long long* d_offsets, const int* d_det_count,
                               int n_cam, int L, void* d_children, long long* d_marks, int* d_overflow,
                               hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_expand_kernel, grid_for(n + 1), dim3(kBlock), 0, st,
                     reinterpret_cast<const Candidate*>(d_parents), n, d_offsets, d_det_count, n_cam, L,
                     reinterpret_cast<Candidate*>(d_children), d_marks, d_overflow);
  return hipGetLastError();
}
hipError_t launch_sweep_gather(const void* d_children, long long n_children, const long long* d_slot,
                               const int* d_det_count, const long long* d_det_offset, const double* d_dets, int n_cam,
                               int L, long long stride, double* d_pts, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_gather_kernel, grid_for(n_children), dim3(kBlock), 0, st,
                     reinterpret_cast<const Candidate*>(d_children), n_children, d_slot, d_det_count, d_det_offset,
                     reinterpret_cast<const double2*>(d_dets), n_cam, L, stride, reinterpret_cast<double2*>(d_pts));
  return hipGetLastError();
}
hipError_t launch_sweep_prune(void* d_children, long long n_children, const long long* d_slot, const double* d_xyz,
                              const double* d_err, double error_max, const int* d_lm_status, int depth, void* d_cuts,
                              long long cuts_cap, unsigned long long* d_n_cuts, long long* d_alive, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_prune_kernel, grid_for(n_children + 1), dim3(kBlock), 0, st,
                     reinterpret_cast<Candidate*>(d_children), n_children, d_slot, d_xyz, d_err, error_max,
                     d_lm_status, depth, reinterpret_cast<Candidate*>(d_cuts), cuts_cap, d_n_cuts, d_alive);
  return hipGetLastError();
}
hipError_t launch_sweep_final(const void* d_nodes, long long n, int min_cameras, long long* d_keep, hipStream_t st) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(sweep_final_kernel, grid_for(n + 1), dim3(kBlock), 0, st,
                     reinterpret_cast<const Candidate*>(d_nodes), n, min_cameras, d_keep);
  return hipGetLastError();
}
hipError_t launch_sweep_compact(const void* d_in, long long n, const long long* d_pos, void* d_out, hipStream_t st) {
  (void)hipGetLastError();
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sweep_compact_kernel, grid_for(n), dim3(kBlock), 0, st, reinterpret_cast<const Candidate*>(d_in),
                     n, d_pos, reinterpret_cast<Candidate*>(d_out));
  return hipGetLastError();
}

hipError_t launch_score_leaves(const DeviceCams& cams, int n_frames, const long long* d_frame_leaf_offset,
                               const int* d_det_count, const long long* d_det_offset,
                               const double* d_dets, double error_max, int min_cameras,
                               long long n_leaves, void* d_out, unsigned long long* d_keys,
                               long long cap, unsigned long long* d_n_out, hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  if (n_leaves == 0) return hipSuccess;
  const int n_cam = cams.n_cam;
  const dim3 grid((unsigned)((n_leaves + kBlock - 1) / kBlock)), block(kBlock);
#define TDR_CASE(N)                                                                              \
  if (n_cam <= N) {                                                                              \
    hipLaunchKernelGGL((score_leaves_kernel<N>), grid, block, 0, stream, cams.d_P, n_cam,        \
                       n_frames, d_frame_leaf_offset, d_det_count, d_det_offset,                 \
                       reinterpret_cast<const double2*>(d_dets), error_max, min_cameras,        \
                       n_leaves, reinterpret_cast<Candidate*>(d_out), d_keys, cap, d_n_out);     \
    return hipGetLastError();                                                                    \
  }
  TDR_CASE(4) TDR_CASE(8) TDR_CASE(12) TDR_CASE(16) TDR_CASE(32)
#undef TDR_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_dist_from_ray(const DeviceCams& cams, const int32_t* d_cam_idx,
                                const double* d_uv, const double* d_point, int64_t n,
                                double* d_dist, hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(dist_from_ray_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, cams.d_ray, cams.n_cam, d_cam_idx,
                     reinterpret_cast<const double2*>(d_uv), d_point, n, d_dist);
  return hipGetLastError();
}

hipError_t launch_dist_from_ray_point(const RayCam& cam, double u, double v, const double* point, double* d_out,
                                      unsigned long long* seq_word, unsigned long long seq, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(dist_from_ray_point_kernel, dim3(1), dim3(kWave), 0, stream, cam, u, v, point[0], point[1],
                     point[2], d_out, seq_word, seq);
  return hipGetLastError();
}

hipError_t launch_matrix_point_args(const PointArgsHost& args, int n_obs, double* out4, unsigned long long* seq_word,
                                    unsigned long long seq, hipStream_t stream) {
  (void)hipGetLastError();
  static_assert(sizeof(PointArgsHost) == sizeof(PointArgs), "one layout");
  hipLaunchKernelGGL(matrix_point_arg_kernel, dim3(1), dim3(kWave), 0, stream,
                     *reinterpret_cast<const PointArgs*>(&args), n_obs, out4, seq_word, seq);
  return hipGetLastError();
}

hipError_t launch_synth(const DeviceCams& cams, uint64_t seed, int n_drones, int64_t sys_offset,
                        int64_t n_sys, int64_t cam_stride, double noise_px, double dropout_p,
                        int round_to_int, int ray_model, double* d_pts, double* d_truth,
                        hipStream_t stream) {
  (void)hipGetLastError();  // drop any stale error another library left on this thread
  if (n_sys == 0) return hipSuccess;
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n_sys + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, stream, cams.d_P, cams.d_ray, cams.n_cam, seed, n_drones, sys_offset,
                     n_sys, cam_stride, noise_px, dropout_p, round_to_int, ray_model,
                     reinterpret_cast<double2*>(d_pts), d_truth);
  return hipGetLastError();
}

int knob_set(const char* name, int value, int unset) {
  if (!name) return -1;
  for (int k = 0; k < K_COUNT; ++k)
    if (std::strcmp(name, kKnobNames[k]) == 0) {
      knob_table().v[k].store(unset ? kKnobUnset : value, std::memory_order_relaxed);
      return 0;
    }
  return -1;
}

bool knob_sweep_on_host() { return knob(K_SWEEP_HOST, 0, 0, 1) != 0; }

}  // namespace tdr_amd
