/*
 * tdr_oracle.c — TEST INFRASTRUCTURE ONLY.  Not shipped, not linked by the product.
 *
 * Plain-C, fp64, OpenCV-free CPU restatement of the reference hot path
 *   Triangulator::triangulatePoints  (Grzetan/3D-Reconstruction-Triangulation)
 * used ONLY as the checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product path (HIP kernels behind
 * include/tdr_amd.h) never calls into this file.
 *
 * PARITY UNPINNED: the reference cannot be built here (it hard-requires OpenCV,
 * CMakeLists.txt:4, which is absent from this image and there is no network), and
 * the reference ships no unit tests, golden vectors or fixtures for this path
 * (SURVEY.md §4, §8c).  The oracle is therefore pinned only by
 *   (1) known-answer-by-construction scenes (exact projections -> exact 3D point),
 *   (2) an independent LAPACK solve (numpy.linalg.pinv / lstsq) of the same A,b,
 *   (3) the closed-form ray-closest-point minimiser for the LM path.
 * Third-party arithmetic restated from memory of OpenCV 4.x (version unpinned in the
 * reference): cv::invert(DECOMP_SVD) = one-sided Jacobi SVD + back-substitution with
 * the 2*DBL_EPSILON*sum(w) cut; cv::Mat::inv() on 3x3 = adjugate/determinant;
 * cv::LMSolver::run = levmarq.cpp loop; cv::solve/invert(DECOMP_EIG) = Jacobi
 * eigen-decomposition + back-substitution.  These are marked [OpenCV-memory].
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_E_SHAPE (-1)
#define ORACLE_E_TOO_FEW (-2)
#define ORACLE_E_CAMERA (-3)

#define ORACLE_MAX_CAM 64

/* src/Camera.h:23-26 */
static const double RAD_TO_DEG = 57.29577951308232087679;
static const double DEG_TO_RAD = 0.01745329251994329576;

/* src/RayTriangulator.h:9-11 */
#define RAY_THRESHOLD 1e-4
#define RAY_MAX_ITERATIONS 1000

typedef struct {
  int width, height, cx, cy;
  double fx, fy, fovx, fovy;
  double rquat[4]; /* (a,b,c,d) = w-first, as stored; never normalised */
  double tvec[3];  /* XCP POSITION */
  double campos[3];
  double K[9];  /* row-major 3x3 */
  double E[12]; /* row-major 3x4 */
  double P[12]; /* row-major 3x4 = cameraPerspectiveMatrix */
} oracle_camera;

/* ---------------------------------------------------------------------------
 * 3x3 helpers
 * ------------------------------------------------------------------------- */

/* [OpenCV-memory] cv::invert fast path for 3x3 (DECOMP_LU, the default of Mat::inv()):
 * determinant by first-row cofactors, adjugate scaled by 1/det; singular -> zeros. */
static int inv3x3(const double* S, double* t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
             S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.0) {
    memset(t, 0, 9 * sizeof(double));
    return 0;
  }
  d = 1.0 / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return 1;
}

/* src/Camera.h:274-287 toRotMatrix: normalised copy is computed and IGNORED; the
 * stored quaternion (a,b,c,d) is used as-is with the w-first formula. */
static void quat_to_rotmat(const double* q, double* R) {
  double a = q[0], b = q[1], c = q[2], d = q[3];
  R[0] = 1 - 2 * (c * c + d * d);
  R[1] = 2 * (b * c - a * d);
  R[2] = 2 * (b * d + a * c);
  R[3] = 2 * (b * c + a * d);
  R[4] = 1 - 2 * (b * b + d * d);
  R[5] = 2 * (c * d - a * b);
  R[6] = 2 * (b * d - a * c);
  R[7] = 2 * (c * d + a * b);
  R[8] = 1 - 2 * (b * b + c * c);
}

/* ---------------------------------------------------------------------------
 * tdr::Camera::compCamParams chain  (src/Camera.h:177-187) as driven by
 * createCamera (src/utils.cpp:94-107).
 * ------------------------------------------------------------------------- */
int oracle_camera_derive(int width, int height, double focal_length, const double* tvec,
                         const double* rquat, oracle_camera* cam) {
  memset(cam, 0, sizeof(*cam));
  cam->width = width;
  cam->height = height;
  cam->fx = focal_length;
  memcpy(cam->tvec, tvec, 3 * sizeof(double));
  memcpy(cam->rquat, rquat, 4 * sizeof(double));

  /* compCxCy, Camera.h:78-83 */
  if (width == 0 || height == 0) return ORACLE_E_CAMERA;
  cam->cx = (int)round(width / 2.0);
  cam->cy = (int)round(height / 2.0);
  /* compFovx, Camera.h:89-92 — truncated constant 57.2958 */
  if (cam->fx == 0) return ORACLE_E_CAMERA;
  cam->fovx = 2 * atan(width / (2 * cam->fx)) * 57.2958;
  /* compFovy, Camera.h:98-104 */
  if (cam->fovx == 0) return ORACLE_E_CAMERA;
  cam->fovy =
      2.0 * atan(tan(cam->fovx * 0.5 * DEG_TO_RAD) / ((double)width / (double)height)) * RAD_TO_DEG;
  /* compFxFy, Camera.h:111-116 — fx and fy are RE-derived from the fovs */
  if (cam->fovy == 0) return ORACLE_E_CAMERA;
  cam->fx = (width / 2.0) / (tan((cam->fovx / 2.0) * DEG_TO_RAD));
  cam->fy = (height / 2.0) / (tan((cam->fovy / 2.0) * DEG_TO_RAD));
  /* compCamPos, Camera.h:167-170: R = toRotMatrix(q).inv(); camPos = -R * tvec */
  double R[9], Rinv[9];
  quat_to_rotmat(cam->rquat, R);
  inv3x3(R, Rinv);
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (-Rinv[i * 3 + k]) * cam->tvec[k];
    cam->campos[i] = s;
  }
  /* createCamMat, Camera.h:122-127 */
  if (cam->fx == 0 || cam->fy == 0 || cam->cx == 0 || cam->cy == 0) return ORACLE_E_CAMERA;
  double* K = cam->K;
  K[0] = cam->fx; K[1] = 0;       K[2] = cam->cx;
  K[3] = 0;       K[4] = cam->fy; K[5] = cam->cy;
  K[6] = 0;       K[7] = 0;       K[8] = 1;
  /* createExtricsicMat, Camera.h:133-153: [Rinv | camPos] */
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) cam->E[i * 4 + j] = Rinv[i * 3 + j];
    cam->E[i * 4 + 3] = cam->campos[i];
  }
  /* createPerspectiveMat, Camera.h:159-161: P = K * E */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * cam->E[k * 4 + j];
      cam->P[i * 4 + j] = s;
    }
  return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * [OpenCV-memory] cv::invert(A, inv, DECOMP_SVD) for an m x 3 matrix (m = 2N):
 * one-sided (Hestenes) Jacobi SVD on the columns, then pinv = V diag(1/w) U^T with
 * singular values w_i <= 2*DBL_EPSILON*sum(w) dropped (SVD::backSubst).
 * Used at src/MatrixTriangulator.cpp:52-53.
 *   A  : m x 3 row-major      pinv : 3 x m row-major
 * ------------------------------------------------------------------------- */
static void pinv_jacobi_mx3(const double* A, int m, double* pinv) {
  enum { n = 3 };
  /* columns of A as rows of At (n x m); V accumulates the right rotations */
  double At[n * 2 * ORACLE_MAX_CAM]; /* m <= 2*ORACLE_MAX_CAM */
  double V[n][n] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  double W[n];
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) {
      double t = A[k * n + i];
      At[i * m + k] = t;
      sd += t * t;
    }
    W[i] = sd;
  }
  const double eps = DBL_EPSILON * 10;
  int max_iter = m > 30 ? m : 30;
  for (int iter = 0; iter < max_iter; ++iter) {
    int changed = 0;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        double* Ai = At + i * m;
        double* Aj = At + j * m;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = hypot(p, beta), c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; ++k) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = 1;
        for (int k = 0; k < n; ++k) {
          double t0 = c * V[i][k] + s * V[j][k];
          double t1 = -s * V[i][k] + c * V[j][k];
          V[i][k] = t0;
          V[j][k] = t1;
        }
      }
    if (!changed) break;
  }
  double wsum = 0;
  for (int i = 0; i < n; ++i) {
    double sd = 0;
    for (int k = 0; k < m; ++k) sd += At[i * m + k] * At[i * m + k];
    W[i] = sqrt(sd);
    wsum += W[i];
  }
  /* At row i = w_i * u_i^T ;  V row i = v_i^T.  pinv = sum_i v_i (1/w_i) u_i^T */
  double thr = 2 * DBL_EPSILON * wsum;
  memset(pinv, 0, sizeof(double) * n * m);
  for (int i = 0; i < n; ++i) {
    if (W[i] <= thr) continue;
    double iw = 1.0 / W[i];
    for (int j = 0; j < n; ++j) {
      double vj = V[i][j] * iw;
      for (int k = 0; k < m; ++k) pinv[j * m + k] += vj * (At[i * m + k] * iw);
    }
  }
}

/* the same arithmetic for oracle/ref_cost_model.cpp (the reference-shaped cost model of the CPU baseline) */
void oracle_pinv_jacobi_mx3(const double* A, int m, double* pinv) { pinv_jacobi_mx3(A, m, pinv); }

/* ---------------------------------------------------------------------------
 * MatrixTriangulator::triangulatePoint  (src/MatrixTriangulator.cpp:3-62)
 *   P   : n x 12 (row-major 3x4 per observation's camera)
 *   uv  : n x 2
 * A (-1) point is rejected here (the reference leaves its rows uninitialised,
 * :12-14 — unreachable through triangulatePoints; not reproduced).
 * ------------------------------------------------------------------------- */
int oracle_matrix_triangulate_point(const double* P, const double* uv, int n, double* X,
                                    double* err) {
  if (n < 1 || n > ORACLE_MAX_CAM) return ORACLE_E_SHAPE;
  double A[2 * ORACLE_MAX_CAM * 3], b[2 * ORACLE_MAX_CAM], pinv[3 * 2 * ORACLE_MAX_CAM];
  for (int i = 0; i < n; ++i) {
    const double* p = P + 12 * i;
    double u = uv[2 * i], v = uv[2 * i + 1];
    if (u == -1 || v == -1) return ORACLE_E_SHAPE;
    /* :16-40 */
    A[(2 * i) * 3 + 0] = p[0] - u * p[8];
    A[(2 * i) * 3 + 1] = p[1] - u * p[9];
    A[(2 * i) * 3 + 2] = p[2] - u * p[10];
    A[(2 * i + 1) * 3 + 0] = p[4] - v * p[8];
    A[(2 * i + 1) * 3 + 1] = p[5] - v * p[9];
    A[(2 * i + 1) * 3 + 2] = p[6] - v * p[10];
    /* :42-49 */
    b[2 * i] = u * p[11] - p[3];
    b[2 * i + 1] = v * p[11] - p[7];
  }
  int m = 2 * n;
  pinv_jacobi_mx3(A, m, pinv); /* :52-53 */
  for (int j = 0; j < 3; ++j) { /* :54  X = inv_L * bias */
    double s = 0;
    for (int k = 0; k < m; ++k) s += pinv[j * m + k] * b[k];
    X[j] = s;
  }
  if (err) { /* :55-59 */
    double ss = 0;
    for (int k = 0; k < m; ++k) {
      double r = (A[k * 3] * X[0] + A[k * 3 + 1] * X[1] + A[k * 3 + 2] * X[2]) - b[k];
      ss += r * r;
    }
    *err = sqrt(ss / (double)(2 * n));
  }
  return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * MatrixTriangulator::triangulatePoints  (src/MatrixTriangulator.cpp:70-100)
 *   P        : n_cam x 12  (the triangulator's cameras)
 *   pts      : [n_cam_pts][n_sys][2], (-1,-1)-style sentinel: x==-1 || y==-1  (:86)
 *   cameras used: n_cam_pts and n_cam are combined with min()            (:84)
 *   <2 valid -> whole batch fails, *first_bad = index of that system     (:92-94)
 *   out_err is optional (the reference discards it, :96)
 * ------------------------------------------------------------------------- */
static int matrix_range(const double* P, int n_cam, const double* pts, int n_cam_pts,
                        int64_t n_sys, int64_t s0, int64_t s1, double* out_xyz, double* out_err,
                        int64_t* first_bad) {
  int nc = n_cam_pts < n_cam ? n_cam_pts : n_cam;
  double Pf[ORACLE_MAX_CAM * 12], uv[ORACLE_MAX_CAM * 2];
  for (int64_t s = s0; s < s1; ++s) {
    int n = 0;
    for (int c = 0; c < nc; ++c) {
      double x = pts[((int64_t)c * n_sys + s) * 2], y = pts[((int64_t)c * n_sys + s) * 2 + 1];
      if (x == -1 || y == -1) continue;
      memcpy(Pf + 12 * n, P + 12 * c, 12 * sizeof(double));
      uv[2 * n] = x;
      uv[2 * n + 1] = y;
      ++n;
    }
    if (n < 2) {
      *first_bad = s;
      return ORACLE_E_TOO_FEW;
    }
    double e;
    oracle_matrix_triangulate_point(Pf, uv, n, out_xyz + 3 * s, &e);
    if (out_err) out_err[s] = e;
  }
  return ORACLE_OK;
}

int oracle_matrix_triangulate_points(const double* P, int n_cam, const double* pts, int n_cam_pts,
                                     int64_t n_sys, double* out_xyz, double* out_err,
                                     int64_t* first_bad) {
  int64_t fb = -1;
  if (first_bad) *first_bad = -1;
  /* empty `points` is UB in the reference (:72,:80); rejected here */
  if (n_cam < 1 || n_cam_pts < 1 || n_cam > ORACLE_MAX_CAM || n_sys < 0) return ORACLE_E_SHAPE;
  int rc = matrix_range(P, n_cam, pts, n_cam_pts, n_sys, 0, n_sys, out_xyz, out_err, &fb);
  if (first_bad) *first_bad = fb;
  return rc;
}

/* Same maths, frames block-partitioned over n_threads pthreads (for the CPU baseline;
 * the reference itself is single-threaded). */
typedef struct {
  const double *P, *pts;
  int n_cam, n_cam_pts;
  int64_t n_sys, s0, s1;
  double *xyz, *err;
  int64_t first_bad;
  int rc;
  int fast;
} mt_job;

static int matrix_range_fast(const double* P, int n_cam, const double* pts, int n_cam_pts,
                             int64_t n_sys, int64_t s0, int64_t s1, double* out_xyz,
                             double* out_err, int64_t* first_bad);

static void* mt_worker(void* arg) {
  mt_job* j = (mt_job*)arg;
  j->first_bad = -1;
  j->rc = (j->fast ? matrix_range_fast : matrix_range)(j->P, j->n_cam, j->pts, j->n_cam_pts,
                                                       j->n_sys, j->s0, j->s1, j->xyz, j->err,
                                                       &j->first_bad);
  return NULL;
}

static int matrix_mt(const double* P, int n_cam, const double* pts, int n_cam_pts, int64_t n_sys,
                     double* out_xyz, double* out_err, int64_t* first_bad, int n_threads,
                     int fast) {
  if (first_bad) *first_bad = -1;
  if (n_cam < 1 || n_cam_pts < 1 || n_cam > ORACLE_MAX_CAM || n_sys < 0) return ORACLE_E_SHAPE;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  mt_job jobs[256];
  for (int t = 0; t < n_threads; ++t) {
    mt_job* j = &jobs[t];
    j->P = P; j->pts = pts; j->n_cam = n_cam; j->n_cam_pts = n_cam_pts; j->n_sys = n_sys;
    j->s0 = n_sys * t / n_threads;
    j->s1 = n_sys * (t + 1) / n_threads;
    j->xyz = out_xyz; j->err = out_err; j->fast = fast;
    pthread_create(&th[t], NULL, mt_worker, j);
  }
  int rc = ORACLE_OK;
  int64_t fb = -1;
  for (int t = 0; t < n_threads; ++t) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc != ORACLE_OK && rc == ORACLE_OK) { /* lowest block first = first failing index */
      rc = jobs[t].rc;
      fb = jobs[t].first_bad;
    }
  }
  if (first_bad) *first_bad = fb;
  return rc;
}

int oracle_matrix_triangulate_points_mt(const double* P, int n_cam, const double* pts,
                                        int n_cam_pts, int64_t n_sys, double* out_xyz,
                                        double* out_err, int64_t* first_bad, int n_threads) {
  return matrix_mt(P, n_cam, pts, n_cam_pts, n_sys, out_xyz, out_err, first_bad, n_threads, 0);
}

/* "Best-effort CPU" flavour: flat arrays, normal equations + Cholesky,
 * no per-system heap traffic.  Same inputs/outputs/semantics as the faithful path; the
 * solution differs from the pinv only by round-off for full-rank systems. */
static int matrix_range_fast(const double* P, int n_cam, const double* pts, int n_cam_pts,
                             int64_t n_sys, int64_t s0, int64_t s1, double* out_xyz,
                             double* out_err, int64_t* first_bad) {
  int nc = n_cam_pts < n_cam ? n_cam_pts : n_cam;
  for (int64_t s = s0; s < s1; ++s) {
    double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0, v0 = 0, v1 = 0, v2 = 0;
    int n = 0;
    for (int c = 0; c < nc; ++c) {
      const double* q = pts + ((int64_t)c * n_sys + s) * 2;
      double x = q[0], y = q[1];
      if (x == -1 || y == -1) continue;
      const double* p = P + 12 * c;
      double a0 = p[0] - x * p[8], a1 = p[1] - x * p[9], a2 = p[2] - x * p[10];
      double c0 = p[4] - y * p[8], c1 = p[5] - y * p[9], c2 = p[6] - y * p[10];
      double b0 = x * p[11] - p[3], b1 = y * p[11] - p[7];
      m00 += a0 * a0 + c0 * c0; m01 += a0 * a1 + c0 * c1; m02 += a0 * a2 + c0 * c2;
      m11 += a1 * a1 + c1 * c1; m12 += a1 * a2 + c1 * c2; m22 += a2 * a2 + c2 * c2;
      v0 += a0 * b0 + c0 * b1; v1 += a1 * b0 + c1 * b1; v2 += a2 * b0 + c2 * b1;
      ++n;
    }
    if (n < 2) {
      *first_bad = s;
      return ORACLE_E_TOO_FEW;
    }
    /* Cholesky M = L L^T */
    double l00 = sqrt(m00), l10 = m01 / l00, l20 = m02 / l00;
    double l11 = sqrt(m11 - l10 * l10), l21 = (m12 - l20 * l10) / l11;
    double l22 = sqrt(m22 - l20 * l20 - l21 * l21);
    double y0 = v0 / l00, y1 = (v1 - l10 * y0) / l11, y2 = (v2 - l20 * y0 - l21 * y1) / l22;
    double X2 = y2 / l22, X1 = (y1 - l21 * X2) / l11, X0 = (y0 - l10 * X1 - l20 * X2) / l00;
    out_xyz[3 * s] = X0; out_xyz[3 * s + 1] = X1; out_xyz[3 * s + 2] = X2;
    if (out_err) {
      double ss = 0;
      for (int c = 0; c < nc; ++c) {
        const double* q = pts + ((int64_t)c * n_sys + s) * 2;
        double x = q[0], y = q[1];
        if (x == -1 || y == -1) continue;
        const double* p = P + 12 * c;
        double r0 = (p[0] - x * p[8]) * X0 + (p[1] - x * p[9]) * X1 + (p[2] - x * p[10]) * X2 -
                    (x * p[11] - p[3]);
        double r1 = (p[4] - y * p[8]) * X0 + (p[5] - y * p[9]) * X1 + (p[6] - y * p[10]) * X2 -
                    (y * p[11] - p[7]);
        ss += r0 * r0 + r1 * r1;
      }
      out_err[s] = sqrt(ss / (double)(2 * n));
    }
  }
  return ORACLE_OK;
}

int oracle_matrix_triangulate_points_fast(const double* P, int n_cam, const double* pts,
                                          int n_cam_pts, int64_t n_sys, double* out_xyz,
                                          double* out_err, int64_t* first_bad, int n_threads) {
  return matrix_mt(P, n_cam, pts, n_cam_pts, n_sys, out_xyz, out_err, first_bad, n_threads, 1);
}

/* ===========================================================================
 * Ray path
 * ========================================================================= */

typedef struct { double o[3], d[3]; } oracle_ray;

/* Triangulator::calculateRayDirectionForPixel (src/Triangulator.cpp:27-44),
 * rotatePointByQuaternion (:15-25), createRayForPoint (:46-55).
 * cam_aux = [rquat(4) w-first, tvec(3), fovy, width, height]  (10 doubles) */
void oracle_create_ray(const double* cam_aux, double u, double v, double* origin, double* dir) {
  const double* q = cam_aux;
  double fovy = cam_aux[7], width = cam_aux[8], height = cam_aux[9];
  double px = u + 0.5, py = v + 0.5; /* :30-32 pixel centre */
  double d = 1 / tan(fovy * 0.0174533 / 2); /* :34 — second truncated constant */
  double x = (width / height) * ((2 * px / width) - 1); /* :37-38 */
  double y = (2 * py / height) - 1;                     /* :39 */
  /* cv::normalize(Vec3d): v * (1/norm)  [OpenCV-memory] */
  double nrm = sqrt(x * x + y * y + d * d);
  double inv = 1.0 / nrm;
  double vx = x * inv, vy = y * inv, vz = d * inv;
  /* q * (0,v) * conj(q), Hamilton product, Vec4d = (w,x,y,z)  [OpenCV-memory, Vec4 operator*] */
  double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  /* t = q * (0,v) */
  double tw = qw * 0 - qx * vx - qy * vy - qz * vz;
  double tx = qw * vx + qx * 0 + qy * vz - qz * vy;
  double ty = qw * vy - qx * vz + qy * 0 + qz * vx;
  double tz = qw * vz + qx * vy - qy * vx + qz * 0;
  /* r = t * conj(q), conj = (w,-x,-y,-z) */
  double cx = -qx, cy = -qy, cz = -qz;
  dir[0] = tw * cx + tx * qw + ty * cz - tz * cy;
  dir[1] = tw * cy - tx * cz + ty * qw + tz * cx;
  dir[2] = tw * cz + tx * cy - ty * cx + tz * qw;
  origin[0] = cam_aux[4]; /* :49-53 origin = tvec */
  origin[1] = cam_aux[5];
  origin[2] = cam_aux[6];
}

/* Triangulator::distToRay (src/Triangulator.cpp:3-9), non-squared */
static double dist_to_ray(const oracle_ray* r, double px, double py, double pz) {
  double wx = px - r->o[0], wy = py - r->o[1], wz = pz - r->o[2];
  double cx = r->d[1] * wz - r->d[2] * wy;
  double cy = r->d[2] * wx - r->d[0] * wz;
  double cz = r->d[0] * wy - r->d[1] * wx;
  return sqrt(cx * cx + cy * cy + cz * cz);
}

/* Triangulator::getDistFromRay (src/Triangulator.cpp:57-61) */
double oracle_get_dist_from_ray(const double* cam_aux, double u, double v, const double* point) {
  oracle_ray r;
  oracle_create_ray(cam_aux, u, v, r.o, r.d);
  return dist_to_ray(&r, point[0], point[1], point[2]);
}

int oracle_lm_analytic_jacobian = 0; /* 0 = the reference's central difference (always, except in the experiment) */

/* RayClosestPoint::compute (src/RayTriangulator.cpp:8-47).  Returns the mean distance
 * (the `error` member, :26) — summed sequentially (the reference's forEach race, SURVEY §5,
 * is not reproduced). */
static double ray_compute(const oracle_ray* rays, int n, const double* x, double* r, double* J) {
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    r[i] = dist_to_ray(&rays[i], x[0], x[1], x[2]);
    sum += r[i];
  }
  if (J && oracle_lm_analytic_jacobian) {
    /* experiment switch (tests/tools/lm_jacobian_form.py): the exact gradient of the distance,
     * grad r = ((d x w) x d) / r, instead of the reference's central difference — they differ by the
     * truncation term eps^2 / (2 r^2) (|d x e_k|^2 - g_k^2) g_k */
    for (int i = 0; i < n; ++i) {
      const double* dv = rays[i].d;
      double w[3] = {x[0] - rays[i].o[0], x[1] - rays[i].o[1], x[2] - rays[i].o[2]};
      double c[3] = {dv[1] * w[2] - dv[2] * w[1], dv[2] * w[0] - dv[0] * w[2], dv[0] * w[1] - dv[1] * w[0]};
      double p[3] = {c[1] * dv[2] - c[2] * dv[1], c[2] * dv[0] - c[0] * dv[2], c[0] * dv[1] - c[1] * dv[0]};
      for (int k = 0; k < 3; ++k) J[i * 3 + k] = r[i] > 0 ? p[k] / r[i] : 0.0;
      if (oracle_lm_analytic_jacobian >= 2) {
        /* 2: the kernels' rule — the exact gradient only for rays farther than 0.1 mm, the literal central difference
         *    below;  3: the same with the central difference's second-order truncation term carried along:
         *    j_k = g_k (1 - eps^2 / (2 r^2) (|d x e_k|^2 - g_k^2))  (round 1's kernel form) */
        const double e = RAY_THRESHOLD;
        if (r[i] > 0 && r[i] * r[i] < 1e-2) {
          J[i * 3 + 0] = (dist_to_ray(&rays[i], x[0] + e, x[1], x[2]) - dist_to_ray(&rays[i], x[0] - e, x[1], x[2])) / (2 * e);
          J[i * 3 + 1] = (dist_to_ray(&rays[i], x[0], x[1] + e, x[2]) - dist_to_ray(&rays[i], x[0], x[1] - e, x[2])) / (2 * e);
          J[i * 3 + 2] = (dist_to_ray(&rays[i], x[0], x[1], x[2] + e) - dist_to_ray(&rays[i], x[0], x[1], x[2] - e)) / (2 * e);
        } else if (oracle_lm_analytic_jacobian == 3 && r[i] > 0) {
          const double D = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
          const double f = e * e / (2 * r[i] * r[i]);
          for (int k = 0; k < 3; ++k) {
            const double g = J[i * 3 + k];
            J[i * 3 + k] = g * (1 - f * ((D - dv[k] * dv[k]) - g * g));
          }
        }
      }
    }
  } else if (J) {
    const double e = RAY_THRESHOLD;
    for (int i = 0; i < n; ++i) {
      J[i * 3 + 0] = (dist_to_ray(&rays[i], x[0] + e, x[1], x[2]) -
                      dist_to_ray(&rays[i], x[0] - e, x[1], x[2])) / (2 * e);
      J[i * 3 + 1] = (dist_to_ray(&rays[i], x[0], x[1] + e, x[2]) -
                      dist_to_ray(&rays[i], x[0], x[1] - e, x[2])) / (2 * e);
      J[i * 3 + 2] = (dist_to_ray(&rays[i], x[0], x[1], x[2] + e) -
                      dist_to_ray(&rays[i], x[0], x[1], x[2] - e)) / (2 * e);
    }
  }
  return sum / (double)n;
}

/* [OpenCV-memory] symmetric 3x3 Jacobi eigen-decomposition + SVBkSb-style
 * back-substitution = cv::solve / cv::invert with DECOMP_EIG.
 *   inv (optional): 3x3 pseudo-inverse;   x (optional): inv * b */
int oracle_jacobi_stop_rule = 0; /* experiment / test switch, see the loop below; 0 always otherwise */
static void sym3_eig_solve(const double* A, const double* b, double* x, double* inv) {
  double a[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = A[i * 3 + j];
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    /* [OpenCV-memory] JacobiImpl_ stops at |pivot| <= eps.  PARITY UNPINNED: OpenCV's loop is a max-pivot Jacobi with
     * an absolute eps, this one a cyclic Jacobi with a relative threshold — cv2 is not importable here, so how
     * DECOMP_EIG behaves near its 2 eps sum|w| cut cannot be checked against OpenCV itself.  What CAN be checked is
     * that the threshold does not matter: oracle_jacobi_stop_rule = 1 continues to off <= eps^2 diag (round 2's
     * rule), and tests/test_oracle_cpu.py::test_jacobi_stop_rule_does_not_matter runs both on matrices whose
     * w_min / trace spans 1e-17 .. 1e-5 and on whole LM runs. */
    if (off <= (oracle_jacobi_stop_rule ? DBL_EPSILON * DBL_EPSILON : DBL_EPSILON) * diag || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A G */
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- G^T A */
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  double w[3] = {a[0][0], a[1][1], a[2][2]};
  double thr = 2 * DBL_EPSILON * (fabs(w[0]) + fabs(w[1]) + fabs(w[2]));
  double M[9] = {0};
  for (int i = 0; i < 3; ++i) {
    if (fabs(w[i]) <= thr) continue;
    double iw = 1.0 / w[i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[r * 3 + c] += V[r][i] * iw * V[c][i];
  }
  if (inv) memcpy(inv, M, sizeof(M));
  if (x)
    for (int r = 0; r < 3; ++r) x[r] = M[r * 3] * b[0] + M[r * 3 + 1] * b[1] + M[r * 3 + 2] * b[2];
}

/* Experiment switch (tests/tools/lm_solve_form.py; 0 always, except in that experiment): S = sum r_i^2 taken from the
 * squared cross products directly (what the HIP kernel sums: no sqrt-then-square round trip per ray, two roundings
 * fewer per term) instead of norm(r, NORM_L2SQR) over the square-rooted residuals as cv::LMSolver does. */
int oracle_lm_sum_from_squares = 0; /* 2: additionally with the kernel's fused multiply-adds in the cross product */
static double dist2_to_ray(const oracle_ray* r, double px, double py, double pz) {
  double wx = px - r->o[0], wy = py - r->o[1], wz = pz - r->o[2];
  if (oracle_lm_sum_from_squares == 2) {
    double cx = fma(r->d[1], wz, -(r->d[2] * wy));
    double cy = fma(r->d[2], wx, -(r->d[0] * wz));
    double cz = fma(r->d[0], wy, -(r->d[1] * wx));
    return fma(cz, cz, fma(cy, cy, cx * cx));
  }
  double cx = r->d[1] * wz - r->d[2] * wy;
  double cy = r->d[2] * wx - r->d[0] * wz;
  double cz = r->d[0] * wy - r->d[1] * wx;
  return cx * cx + cy * cy + cz * cz;
}

/* Experiment switch (tests/tools/lm_solve_form.py; 0 always, except in that experiment): solve the trial step and
 * the lambda restart the way the ROUND-2 HIP kernel did — LDL^T with reciprocal pivots, |det| / max|cof_kk| — instead
 * of the reference's DECOMP_EIG pseudo-inverse.  The two agree on full-rank matrices; on a rank-deficient J^T J
 * (two rays: always) LDL^T yields a NaN / huge step where DECOMP_EIG drops the null direction and takes the
 * minimum-norm step.  VERDICT r2 item 1 asked which of the two deviations (this one, or the exact-gradient
 * Jacobian above) biases the status words. */
int oracle_lm_ldlt_solve = 0;

static void sym3_ldlt_solve(const double* M, const double* b, double* x) {
  double i0 = 1.0 / M[0];
  double l10 = M[1] * i0, l20 = M[2] * i0;
  double d1 = M[4] - l10 * M[1];
  double t12 = M[5] - l20 * M[1];
  double i1 = 1.0 / d1;
  double l21 = t12 * i1;
  double d2 = (M[8] - l20 * M[2]) - l21 * t12;
  double i2 = 1.0 / d2;
  double y1 = b[1] - l10 * b[0];
  double y2 = (b[2] - l20 * b[0]) - l21 * y1;
  x[2] = y2 * i2;
  x[1] = y1 * i1 - l21 * x[2];
  x[0] = b[0] * i0 - (l10 * x[1] + l20 * x[2]);
}

static double sym3_ldlt_restart(const double* A) {
  double c00 = A[4] * A[8] - A[5] * A[5], c11 = A[0] * A[8] - A[2] * A[2], c22 = A[0] * A[4] - A[1] * A[1];
  double c01 = A[2] * A[5] - A[1] * A[8], c02 = A[1] * A[5] - A[2] * A[4];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double q = fmax(fabs(c00), fmax(fabs(c11), fabs(c22)));
  if (q == 0.0) return 1.0 / DBL_EPSILON;
  return fmin(1.0 / DBL_EPSILON, fabs(det) / q);
}

extern int oracle_lm_hybrid_rule;   /* defined with its description further down */
extern double oracle_lm_hybrid_tau;
extern long oracle_lm_hybrid_eig_systems;
/* the kernels' lambda restart on the hot path: max_k |(A^-1)_kk| = max|cof_kk| / |det| — unless the active rule sends
 * the matrix to the eigen form (returns 0) */
static int hybrid_restart_ldlt(const double* A, double* max_inv_diag) {
  double c00 = A[4] * A[8] - A[5] * A[5], c11 = A[0] * A[8] - A[2] * A[2], c22 = A[0] * A[4] - A[1] * A[1];
  double c01 = A[2] * A[5] - A[1] * A[8], c02 = A[1] * A[5] - A[2] * A[4];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double q = fmax(fabs(c00), fmax(fabs(c11), fabs(c22)));
  double tr = A[0] + A[4] + A[8];
  if (oracle_lm_hybrid_rule == 1) {
    if (!(fabs(det) > oracle_lm_hybrid_tau * tr * q)) return 0;
  } else {
    if (!(det > oracle_lm_hybrid_tau * tr * ((c00 + c11) + c22))) return 0;
  }
  *max_inv_diag = q / fabs(det);
  return 1;
}

/* Experiment switch (tests/tools/lm_solve_form.py; 0 always, except in that experiment): for a system of exactly TWO
 * rays and lambda == 0 — J^T J = g1 g1^T + g2 g2^T has rank 2 by construction and the reference's DECOMP_EIG drops
 * the null direction g1 x g2 — take the minimum-norm step and the pseudo-inverse's diagonal in CLOSED FORM instead of
 * through the Jacobi eigen-decomposition.  With adj = the cofactor matrix (= e2 n n^T for the unit null vector n),
 * e2 = tr adj = w1 w2 and tr = w1 + w2:
 *     pinv(A) v = (tr v - A v) / e2  for v in the range of A (v = J^T r is, by construction),
 *     pinv(A)_kk = (tr (1 - adj_kk / e2) - A_kk) / e2
 * (on an eigenvector of w1: (tr - w1) / (w1 w2) = 1 / w1).  About 35 flops in a dependent chain of ten against
 * ~1000 for the rotations: what the HIP kernels evaluate for such systems since round 4 (lm_rank2_pinv,
 * csrc/tdr_kernels.hip — the same operations in the same order).  Other roundings than the reference's solve, so the
 * experiment asks whether the status words of two-ray systems keep a fair split against the Jacobi form.
 * Returns 0 when w2 / w1 is too small for the form (then the caller takes the Jacobi solve). */
int oracle_lm_rank2_closed_form = 0;
double oracle_lm_rank2_min_e2 = 1e-10; /* the form is refused below this w2 / w1 (experiment: how low may it go) */

static int sym3_rank2_pinv(const double* A, const double* v, double* d, double* max_inv_diag) {
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
  const double c00 = fma(a11, a22, -(a12 * a12)), c11 = fma(a00, a22, -(a02 * a02)), c22 = fma(a00, a11, -(a01 * a01));
  const double e2 = c00 + c11 + c22, tr = a00 + a11 + a22;
  if (!(e2 > oracle_lm_rank2_min_e2 * (tr * tr))) return 0; /* (nearly) rank 1: two parallel gradients */
  const double ie2 = 1.0 / e2;
  if (d) {
    const double w0 = fma(a00, v[0], fma(a01, v[1], a02 * v[2]));
    const double w1 = fma(a01, v[0], fma(a11, v[1], a12 * v[2]));
    const double w2 = fma(a02, v[0], fma(a12, v[1], a22 * v[2]));
    d[0] = fma(tr, v[0], -w0) * ie2;
    d[1] = fma(tr, v[1], -w1) * ie2;
    d[2] = fma(tr, v[2], -w2) * ie2;
  }
  if (max_inv_diag) {
    const double p0 = fma(tr, fma(-c00, ie2, 1.0), -a00) * ie2;
    const double p1 = fma(tr, fma(-c11, ie2, 1.0), -a11) * ie2;
    const double p2 = fma(tr, fma(-c22, ie2, 1.0), -a22) * ie2;
    *max_inv_diag = fmax(fabs(p0), fmax(fabs(p1), fabs(p2)));
  }
  return 1;
}

/* Experiment switch (tests/tools/lm_hybrid_rule.py; 0 always, except in that experiment): WHEN the HIP kernels leave
 * their LDL^T hot path for the reference's DECOMP_EIG form.  The two solves agree to cond x eps on a full-rank matrix
 * and differ in kind only where DECOMP_EIG drops an eigenvalue (|w| <= 2 eps sum|w|), so the kernels need the eigen
 * form only for matrices that may be within that cut; every other matrix is a question of how much conditioning error
 * the LM loop tolerates before its status words shift.
 *   1: round 3's rule — smallest unpivoted LDL^T pivot <= tau x trace (tau = 1e-5), restart |det| <= tau x trace x
 *      max|cof_kk|.  Safe ((2 eps)^(1/3) = 7.6e-6) but crude: a small FIRST pivot or a lambda D far below the trace
 *      fires it on matrices whose smallest eigenvalue is ten orders above the cut — 4-5 % of all two- and three-ray
 *      systems went to the slow second launch for it.
 *   2: round 4's rule — w_min >= det / e2 with det = the product of the pivots and e2 = the sum of the principal 2x2
 *      minors (= w0 w1 + w0 w2 + w1 w2 >= w1 w2): eigen form iff a pivot is <= 0 or det <= tau x trace x e2. */
int oracle_lm_hybrid_rule = 0;
double oracle_lm_hybrid_tau = 1e-5;
long oracle_lm_hybrid_eig_systems = 0; /* runs that met the eigen form under the active rule (= parked by the kernels) */

static void sym3_pivots(const double* M, double* dpiv) {
  double l10 = M[1] / M[0], l20 = M[2] / M[0];
  double d1 = M[4] - l10 * M[1];
  double t12 = M[5] - l20 * M[1];
  double d2 = (M[8] - l20 * M[2]) - (t12 / d1) * t12;
  dpiv[0] = M[0]; dpiv[1] = d1; dpiv[2] = d2;
}

static double sym3_e2(const double* A) {
  return (A[4] * A[8] - A[5] * A[5]) + (A[0] * A[8] - A[2] * A[2]) + (A[0] * A[4] - A[1] * A[1]);
}

/* 1 = this matrix takes the eigen form under the active rule */
static int hybrid_needs_eig(const double* M) {
  double dp[3];
  sym3_pivots(M, dp);
  double tr = M[0] + M[4] + M[8];
  if (oracle_lm_hybrid_rule == 1) return !(fmin(fmin(dp[0], dp[1]), dp[2]) > oracle_lm_hybrid_tau * tr);
  if (!(dp[0] > 0 && dp[1] > 0 && dp[2] > 0)) return 1;
  return !(dp[0] * dp[1] * dp[2] > oracle_lm_hybrid_tau * tr * sym3_e2(M));
}

/* Rule 3 (round 4): NO eigen-decomposition at all — the DECOMP_EIG answer by rank, decided on the matrix's own
 * invariants tr = sum w, e2 = sum of the principal 2x2 minors (~ w1 w2 when w0 is small), det:
 *   tr <= 0                      zero matrix: pinv = 0
 *   e2 <= 2 eps tr^2             w1 within the reference's cut (2 eps sum|w|): rank 1, pinv = M / tr^2
 *   det <= 2 eps tr e2           w0 = det / e2 within the cut: rank 2, y = (tr v - M v) / e2 (cf. sym3_rank2_pinv); with
 *                                more than two rays y is then cleared of its component along the null vector (the
 *                                largest column of the cofactor matrix)
 *   otherwise                    full rank: LDL^T / the cofactor form of the inverse's diagonal
 * The decisions are noisy within a few eps of the cuts (e2 and det are computed from cancelling products) where the
 * Jacobi form decides on eigenvalues of high relative accuracy; the experiment asks whether that shifts anything. */
static int sym3_cascade(const double* M, const double* v, int n_rays, double* d, double* max_inv_diag) {
  const double a00 = M[0], a01 = M[1], a02 = M[2], a11 = M[4], a12 = M[5], a22 = M[8];
  const double tr = a00 + a11 + a22;
  const double c00 = fma(a11, a22, -(a12 * a12)), c11 = fma(a00, a22, -(a02 * a02)), c22 = fma(a00, a11, -(a01 * a01));
  const double e2 = c00 + c11 + c22;
  const double cut = 2 * DBL_EPSILON;
  if (!(tr > 0)) {
    if (d) d[0] = d[1] = d[2] = 0.0;
    if (max_inv_diag) *max_inv_diag = 0.0;
    return 0;
  }
  if (!(e2 > cut * (tr * tr))) { /* rank 1 */
    const double it2 = 1.0 / (tr * tr);
    if (d) {
      d[0] = fma(a00, v[0], fma(a01, v[1], a02 * v[2])) * it2;
      d[1] = fma(a01, v[0], fma(a11, v[1], a12 * v[2])) * it2;
      d[2] = fma(a02, v[0], fma(a12, v[1], a22 * v[2])) * it2;
    }
    if (max_inv_diag) *max_inv_diag = fmax(fabs(a00), fmax(fabs(a11), fabs(a22))) * it2;
    return 1;
  }
  double dp[3];
  sym3_pivots(M, dp);
  const double det = (dp[0] > 0 && dp[1] > 0 && dp[2] > 0) ? dp[0] * dp[1] * dp[2] : 0.0;
  if (!(det > cut * tr * e2)) { /* rank 2 */
    const double ie2 = 1.0 / e2;
    if (d) {
      const double w0 = fma(a00, v[0], fma(a01, v[1], a02 * v[2]));
      const double w1 = fma(a01, v[0], fma(a11, v[1], a12 * v[2]));
      const double w2 = fma(a02, v[0], fma(a12, v[1], a22 * v[2]));
      const double y0 = fma(tr, v[0], -w0) * ie2, y1 = fma(tr, v[1], -w1) * ie2, y2 = fma(tr, v[2], -w2) * ie2;
      d[0] = y0; d[1] = y1; d[2] = y2;
      if (n_rays > 2) {
        /* (tr I - M) / e2 inverts on the range but sends the null vector to (tr / e2) n instead of 0.  With two rays
         * v = J^T r has no null component by construction and nothing is done — near convergence J^T J tends to rank 1,
         * n becomes noise, and a projection along it biases the runs (measured: 62 652 : 494).  With more rays v has a
         * small genuine null component (<= |r| sqrt(w0)) which the reference drops: the RESULT is cleared of it
         * (n = the largest column of the cofactor matrix; clearing v before the product costs eps x cond^2). */
        const double c01 = fma(a02, a12, -(a01 * a22)), c02 = fma(a01, a12, -(a02 * a11)), c12 = fma(a01, a02, -(a00 * a12));
        double n0, n1, n2;
        if (c00 >= c11 && c00 >= c22) { n0 = c00; n1 = c01; n2 = c02; }
        else if (c11 >= c22) { n0 = c01; n1 = c11; n2 = c12; }
        else { n0 = c02; n1 = c12; n2 = c22; }
        const double nn = fma(n0, n0, fma(n1, n1, n2 * n2));
        const double t = fma(n0, y0, fma(n1, y1, n2 * y2)) / nn;
        d[0] = fma(-t, n0, y0); d[1] = fma(-t, n1, y1); d[2] = fma(-t, n2, y2);
      }
    }
    if (max_inv_diag) {
      const double p0 = fma(tr, fma(-c00, ie2, 1.0), -a00) * ie2;
      const double p1 = fma(tr, fma(-c11, ie2, 1.0), -a11) * ie2;
      const double p2 = fma(tr, fma(-c22, ie2, 1.0), -a22) * ie2;
      *max_inv_diag = fmax(fabs(p0), fmax(fabs(p1), fabs(p2)));
    }
    return 2;
  }
  if (d) sym3_ldlt_solve(M, v, d);
  if (max_inv_diag) *max_inv_diag = fmax(fabs(c00), fmax(fabs(c11), fabs(c22))) / det;
  return 3;
}
int oracle_sym3_cascade(const double* M, const double* v, int n_rays, double* d, double* max_inv_diag) {
  return sym3_cascade(M, v, n_rays, d, max_inv_diag);
}

/* Experiment switch (tests/tools/lm_solve_form.py; 0 always, except in that experiment): J^T J and J^T r of the exact
 * gradient evaluated WITHOUT the gradient — with c = d x w, D = |d|^2 the unit vectors d/|d|, c/|c| and g/|g| are an
 * orthonormal basis, so
 *   g g^T = (D I - d d^T) - (D / |c|^2) c c^T,      sum_i g_i r_i = sum_i (c_i x d_i) = H x - b,
 *   H = sum_i (D_i I - d_i d_i^T),   b = sum_i (D_i o_i - d_i (d_i . o_i))          (per system, not per iterate)
 * — the form a cheaper HIP camera term would use (no p = c x d, no J^T r accumulation).  Same mathematics, other
 * rounding: J^T J comes out of a subtraction from O(n) entries.  The experiment asks whether that changes which
 * runs get flagged. */
int oracle_lm_h_form = 0;

static void h_form_normal(const oracle_ray* rays, int n, const double* x, double* A, double* v) {
  double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}; /* 00 01 02 11 12 22 */
  for (int i = 0; i < n; ++i) {
    const double* d = rays[i].d;
    const double* o = rays[i].o;
    double D = fma(d[2], d[2], fma(d[1], d[1], d[0] * d[0]));
    double dot = fma(d[2], o[2], fma(d[1], o[1], d[0] * o[0]));
    H[0] += fma(-d[0], d[0], D); H[1] -= d[0] * d[1]; H[2] -= d[0] * d[2];
    H[3] += fma(-d[1], d[1], D); H[4] -= d[1] * d[2]; H[5] += fma(-d[2], d[2], D);
    b[0] += fma(D, o[0], -(d[0] * dot)); b[1] += fma(D, o[1], -(d[1] * dot)); b[2] += fma(D, o[2], -(d[2] * dot));
  }
  double a00 = H[0], a01 = H[1], a02 = H[2], a11 = H[3], a12 = H[4], a22 = H[5];
  for (int i = 0; i < n; ++i) {
    const double* d = rays[i].d;
    double wx = x[0] - rays[i].o[0], wy = x[1] - rays[i].o[1], wz = x[2] - rays[i].o[2];
    double cx = d[1] * wz - d[2] * wy, cy = d[2] * wx - d[0] * wz, cz = d[0] * wy - d[1] * wx;
    double r2 = fma(cz, cz, fma(cy, cy, cx * cx));
    double D = fma(d[2], d[2], fma(d[1], d[1], d[0] * d[0]));
    double wa = r2 > 0 ? D / r2 : 0.0;
    double h0 = cx * wa, h1 = cy * wa, h2 = cz * wa;
    a00 = fma(-h0, cx, a00); a01 = fma(-h0, cy, a01); a02 = fma(-h0, cz, a02);
    a11 = fma(-h1, cy, a11); a12 = fma(-h1, cz, a12); a22 = fma(-h2, cz, a22);
  }
  A[0] = a00; A[1] = A[3] = a01; A[2] = A[6] = a02; A[4] = a11; A[5] = A[7] = a12; A[8] = a22;
  v[0] = fma(H[0], x[0], fma(H[1], x[1], H[2] * x[2])) - b[0];
  v[1] = fma(H[1], x[0], fma(H[3], x[1], H[4] * x[2])) - b[1];
  v[2] = fma(H[2], x[0], fma(H[4], x[1], H[5] * x[2])) - b[2];
}

/* exported for tests/test_oracle_cpu.py (the eigen solve against LAPACK) */
void oracle_sym3_eig_solve(const double* A, const double* b, double* x, double* inv) { sym3_eig_solve(A, b, x, inv); }
/* the experiment's closed form (see oracle_lm_rank2_closed_form) for tests/test_gpu_parity.py: x = pinv(A) b,
 * *max_inv_diag = max_k |pinv(A)_kk|; returns 1 where the form applies */
int oracle_sym3_rank2_pinv(const double* A, const double* b, double* x, double* max_inv_diag) {
  return sym3_rank2_pinv(A, b, x, max_inv_diag);
}

static void jtj_jtr(const double* J, const double* r, int n, double* A, double* v) {
  memset(A, 0, 9 * sizeof(double));
  memset(v, 0, 3 * sizeof(double));
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) A[a * 3 + b] += J[i * 3 + a] * J[i * 3 + b];
      v[a] += J[i * 3 + a] * r[i];
    }
}

/* ---------------------------------------------------------------------------
 * RayTriangulator::triangulatePoint (src/RayTriangulator.cpp:83-107) with
 * [OpenCV-memory] cv::LMSolver::run (calib3d levmarq.cpp, OpenCV 4.x), created by
 * LMSolver::create(cb, MAX_ITERATIONS) => epsx = epsf = FLT_EPSILON.
 *   cam_aux : n x 10 (per observation's camera),  uv : n x 2
 *   returns X, err = mean distance at the LAST compute() call (which may be a
 *   rejected trial point, :26/:106), iters (negative if it hit the cap).
 * ------------------------------------------------------------------------- */

/* LM run status (the same two bits are produced by the HIP path: include/tdr_amd.h TDR_LM_*).
 *
 * The reference's solver has two ways of stopping away from the minimiser of its own objective
 * sum_i |d_i x (x - o_i)|^2: the MAX_ITERATIONS cap (src/RayTriangulator.h:10), and the step test
 * |d|_inf < epsx firing on a heavily damped step.  Where it stops then depends on the last bits of hundreds of
 * branchy iterations (tests/tools/lm_sensitivity.py: a 1-ulp change of one ray direction moves such results by
 * up to metres), so no two implementations — not even two builds of one — agree on them.  The objective is an
 * exact quadratic, so "did the run reach the minimiser" has an exact, implementation-independent answer:
 *   x - x* = H^-1 g,   H = sum_i (|d_i|^2 I - d_i d_i^T),   g = sum_i (|d_i|^2 I - d_i d_i^T)(x - o_i).
 *   ORACLE_LM_CAP            the loop ended because iter reached MAX_ITERATIONS
 *   ORACLE_LM_NOT_CONVERGED  |x - x*|_inf > ORACLE_LM_CONV_REL * max(|x*|_inf, 1) at the returned x
 * A run with status 0 ended at the unique minimiser to half the parity tolerance of the tests (1e-7 relative).
 *
 * Oracle-only diagnostics (bits 4, 5): DESIGN.md §2 of round 1 blamed trial steps solved with lambda == 0 on a
 * numerically singular J^T J; they are recorded so that tests/tools/lm_sensitivity.py can show how little of
 * the instability they explain.  rho = det A / (max_k cof_kk * tr A) = 1 / (max_k (A^-1)_kk * tr A) lies in
 * [lmin / tr, 3 lmin / tr] (tr A = number of rays: every gradient of a point-ray distance is a unit vector). */
#define ORACLE_LM_CAP 1
#define ORACLE_LM_NOT_CONVERGED 2
#define ORACLE_LM_CONV_REL 5e-8
#define ORACLE_LM_SINGULAR_TRIAL 16  /* a trial step was solved with lambda == 0 and rho <= oracle_lm_singular_rho */
#define ORACLE_LM_SINGULAR_ACCEPT 32 /* ... and such a trial was accepted (Sd < S) */
double oracle_lm_singular_rho = 1e-6;

typedef struct {
  int status;
  int n_lambda0;   /* trial steps solved with lambda == 0 */
  double min_rho;  /* min of rho over those trials (1 if none) */
  double dist_to_min; /* |x - x*|_inf at the returned x */
} oracle_lm_diag;

static double sym3_rho(const double* A) {
  double c00 = A[4] * A[8] - A[5] * A[5], c11 = A[0] * A[8] - A[2] * A[2], c22 = A[0] * A[4] - A[1] * A[1];
  double c01 = A[2] * A[5] - A[1] * A[8], c02 = A[1] * A[5] - A[2] * A[4];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double tr = A[0] + A[4] + A[8];
  double cmax = fmax(c00, fmax(c11, c22));
  double rho = det / (cmax * tr);
  return rho == rho ? rho : 0.0; /* 0/0 (A == 0) counts as singular */
}

int oracle_ray_triangulate_point_ex(const double* cam_aux, const double* uv, int n, double* X,
                                    double* err, int* iters, oracle_lm_diag* diag, int perturb_ulps);

int oracle_ray_triangulate_point(const double* cam_aux, const double* uv, int n, double* X,
                                 double* err, int* iters) {
  return oracle_ray_triangulate_point_ex(cam_aux, uv, n, X, err, iters, NULL, 0);
}

/* perturb_ulps != 0: the x component of the first ray's direction is moved by that many units in the last
 * place before the solve — the sensitivity probe of tests/test_oracle_cpu.py (a run whose result changes by
 * more than the parity gate under a last-bit change of its input has no reproducible answer). */
static int lm_run(const double* cam_aux, const double* uv, int n, double* X, double* err, int* iters,
                  oracle_lm_diag* diag, int perturb_ulps, int use_h);

int oracle_ray_triangulate_point_ex(const double* cam_aux, const double* uv, int n, double* X,
                                    double* err, int* iters, oracle_lm_diag* diag, int perturb_ulps) {
  /* routing of the HIP kernels (use_h == 2): fewer than 4 rays -> the reference's form from the start */
  return lm_run(cam_aux, uv, n, X, err, iters, diag, perturb_ulps,
                (oracle_lm_h_form == 2 && n < 4) ? 0 : oracle_lm_h_form);
}

/* use_h (experiments only, see oracle_lm_h_form): 1 = the H form throughout; 2 = the H form until the matrix of a
 * trial solve or of a lambda restart is within 1e-5 of singular (smallest LDL^T pivot / |det| test) or a ray comes
 * closer than 0.1 mm, then the whole run again in the reference's form — the routing of the HIP kernels. */
static double sym3_min_pivot(const double* M) {
  double l10 = M[1] / M[0], l20 = M[2] / M[0];
  double d1 = M[4] - l10 * M[1];
  double t12 = M[5] - l20 * M[1];
  double d2 = (M[8] - l20 * M[2]) - (t12 / d1) * t12;
  return fmin(fmin(M[0], d1), d2);
}

static int lm_run(const double* cam_aux, const double* uv, int n, double* X, double* err, int* iters,
                  oracle_lm_diag* diag, int perturb_ulps, int use_h) {
  if (n < 1 || n > ORACLE_MAX_CAM) return ORACLE_E_SHAPE;
  oracle_lm_diag dg = {0, 0, 1.0, 0.0};
  int met_eig = 0; /* experiments: this run went through the eigen form at least once */
  oracle_ray rays[ORACLE_MAX_CAM];
  double r[ORACLE_MAX_CAM], rd[ORACLE_MAX_CAM], J[ORACLE_MAX_CAM * 3];
  double x[3] = {0, 0, 0}, xd[3], A[9], Ap[9], v[3], d[3], D[3];
  for (int i = 0; i < n; ++i) {
    oracle_create_ray(cam_aux + 10 * i, uv[2 * i], uv[2 * i + 1], rays[i].o, rays[i].d);
    x[0] += rays[i].o[0]; /* :90-98 initial guess = mean of origins */
    x[1] += rays[i].o[1];
    x[2] += rays[i].o[2];
  }
  x[0] /= n; x[1] /= n; x[2] /= n;
  for (int k = perturb_ulps; k != 0; k += k > 0 ? -1 : 1)
    rays[0].d[0] = nextafter(rays[0].d[0], k > 0 ? INFINITY : -INFINITY);

  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  double last_err = ray_compute(rays, n, x, r, J);
  double S = 0;
  for (int i = 0; i < n; ++i) S += oracle_lm_sum_from_squares ? dist2_to_ray(&rays[i], x[0], x[1], x[2]) : r[i] * r[i];
  jtj_jtr(J, r, n, A, v);
  if (use_h) h_form_normal(rays, n, x, A, v);
  D[0] = A[0]; D[1] = A[4]; D[2] = A[8];
  const double Rlo = 0.25, Rhi = 0.75;
  double lambda = 1, lc = 0.75;
  int iter = 0;
  for (;;) {
    int singular_trial = 0;
    if (lambda == 0) {
      double rho = sym3_rho(A);
      dg.n_lambda0++;
      if (rho < dg.min_rho) dg.min_rho = rho;
      if (!(rho > oracle_lm_singular_rho)) {
        singular_trial = 1;
        dg.status |= ORACLE_LM_SINGULAR_TRIAL;
      }
    }
    memcpy(Ap, A, sizeof(A));
    for (int i = 0; i < 3; ++i) Ap[i * 3 + i] += lambda * D[i];
    if (use_h == 2) {
      double pm = sym3_min_pivot(Ap);
      if (!(pm > 1e-5 * (Ap[0] + Ap[4] + Ap[8])))
        return lm_run(cam_aux, uv, n, X, err, iters, diag, perturb_ulps, 0);
    }
    if (oracle_lm_ldlt_solve) sym3_ldlt_solve(Ap, v, d);
    else if (oracle_lm_hybrid_rule == 3) sym3_cascade(Ap, v, n, d, NULL);
    else if (oracle_lm_rank2_closed_form && n == 2 && lambda == 0 && sym3_rank2_pinv(Ap, v, d, NULL)) {}
    else if (oracle_lm_hybrid_rule && !hybrid_needs_eig(Ap)) sym3_ldlt_solve(Ap, v, d);
    else {
      met_eig = 1;
      sym3_eig_solve(Ap, v, d, NULL);
    }
    for (int i = 0; i < 3; ++i) xd[i] = x[i] - d[i];
    last_err = ray_compute(rays, n, xd, rd, NULL);
    double Sd = 0;
    for (int i = 0; i < n; ++i)
      Sd += oracle_lm_sum_from_squares ? dist2_to_ray(&rays[i], xd[0], xd[1], xd[2]) : rd[i] * rd[i];
    if (use_h == 2)
      for (int i = 0; i < n; ++i)
        if (rd[i] > 0 && rd[i] * rd[i] < 1e-2) return lm_run(cam_aux, uv, n, X, err, iters, diag, perturb_ulps, 0);
    double td[3]; /* temp_d = 2 v - A d */
    for (int i = 0; i < 3; ++i)
      td[i] = 2 * v[i] - (A[i * 3] * d[0] + A[i * 3 + 1] * d[1] + A[i * 3 + 2] * d[2]);
    double dS = d[0] * td[0] + d[1] * td[1] + d[2] * td[2];
    double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
    if (R > Rhi) {
      lambda *= 0.5;
      if (lambda < lc) lambda = 0;
    } else if (R < Rlo) {
      double t = d[0] * v[0] + d[1] * v[1] + d[2] * v[2];
      double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
      nu = fmin(fmax(nu, 2.), 10.);
      if (lambda == 0 && oracle_lm_ldlt_solve) {
        lambda = lc = sym3_ldlt_restart(A);
        nu *= 0.5;
      } else if (lambda == 0) {
        if (use_h == 2 && !(sym3_ldlt_restart(A) > 1e-5 * (A[0] + A[4] + A[8])))
          return lm_run(cam_aux, uv, n, X, err, iters, diag, perturb_ulps, 0);
        double Ai[9], mid;
        double maxval = DBL_EPSILON;
        if (oracle_lm_hybrid_rule == 3) {
          sym3_cascade(A, NULL, n, NULL, &mid);
          Ai[0] = mid; Ai[4] = Ai[8] = 0.0;
        } else if (oracle_lm_rank2_closed_form && n == 2 && sym3_rank2_pinv(A, NULL, NULL, &mid)) {
          Ai[0] = mid; Ai[4] = Ai[8] = 0.0;
        } else if (oracle_lm_hybrid_rule && hybrid_restart_ldlt(A, &mid)) {
          Ai[0] = mid; Ai[4] = Ai[8] = 0.0;
        } else {
          met_eig = 1;
          sym3_eig_solve(A, NULL, NULL, Ai);
        }
        for (int i = 0; i < 3; ++i) maxval = fmax(maxval, fabs(Ai[i * 3 + i]));
        lambda = lc = 1. / maxval;
        nu *= 0.5;
      }
      lambda *= nu;
    }
    if (Sd < S) {
      if (singular_trial) dg.status |= ORACLE_LM_SINGULAR_ACCEPT;
      S = Sd;
      memcpy(x, xd, sizeof(x));
      last_err = ray_compute(rays, n, x, r, J);
      jtj_jtr(J, r, n, A, v);
      if (use_h) h_form_normal(rays, n, x, A, v);
    }
    iter++;
    double dinf = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2]))), rinf = 0;
    for (int i = 0; i < n; ++i) rinf = fmax(rinf, fabs(r[i]));
    if (!(iter < RAY_MAX_ITERATIONS && dinf >= epsx && rinf >= epsf)) break;
  }
  if (met_eig && oracle_lm_hybrid_rule) __sync_fetch_and_add(&oracle_lm_hybrid_eig_systems, 1);
  X[0] = x[0]; X[1] = x[1]; X[2] = x[2];
  if (err) *err = last_err;
  if (iters) *iters = (iter == RAY_MAX_ITERATIONS) ? -iter : iter;
  if (iter == RAY_MAX_ITERATIONS) dg.status |= ORACLE_LM_CAP;
  if (diag) {
    /* x - x* = H^-1 g (see ORACLE_LM_NOT_CONVERGED above) */
    double H[9] = {0}, g[3] = {0}, Hi[9];
    for (int i = 0; i < n; ++i) {
      const double* dv = rays[i].d;
      double dd = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
      double w[3] = {x[0] - rays[i].o[0], x[1] - rays[i].o[1], x[2] - rays[i].o[2]};
      double dw = dv[0] * w[0] + dv[1] * w[1] + dv[2] * w[2];
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) H[a * 3 + b] += (a == b ? dd : 0.0) - dv[a] * dv[b];
        g[a] += dd * w[a] - dv[a] * dw;
      }
    }
    double dist = INFINITY, xs_inf = 0;
    if (inv3x3(H, Hi)) {
      dist = 0;
      for (int a = 0; a < 3; ++a) {
        double da = Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2];
        dist = fmax(dist, fabs(da));
        xs_inf = fmax(xs_inf, fabs(x[a] - da));
      }
    }
    dg.dist_to_min = dist;
    if (!(dist <= ORACLE_LM_CONV_REL * fmax(xs_inf, 1.0))) dg.status |= ORACLE_LM_NOT_CONVERGED;
    *diag = dg;
  }
  return ORACLE_OK;
}

/* Closed-form minimiser of sum_i |d_i x (x - o_i)|^2 (unit d_i):
 *   (sum_i (I - d_i d_i^T)) x = sum_i (I - d_i d_i^T) o_i
 * Independent pin for the LM path. Uses |d|^2 I - d d^T so it is exact for
 * the reference's objective even when the (un-normalised) quaternion leaves |d| != 1. */
int oracle_ray_closed_form(const double* cam_aux, const double* uv, int n, double* X) {
  double M[9] = {0}, rhs[3] = {0};
  for (int i = 0; i < n; ++i) {
    oracle_ray ry;
    oracle_create_ray(cam_aux + 10 * i, uv[2 * i], uv[2 * i + 1], ry.o, ry.d);
    double dd = ry.d[0] * ry.d[0] + ry.d[1] * ry.d[1] + ry.d[2] * ry.d[2];
    for (int a = 0; a < 3; ++a) {
      double row = 0;
      for (int b = 0; b < 3; ++b) {
        double m = (a == b ? dd : 0.0) - ry.d[a] * ry.d[b];
        M[a * 3 + b] += m;
        row += m * ry.o[b];
      }
      rhs[a] += row;
    }
  }
  double Mi[9];
  if (!inv3x3(M, Mi)) return ORACLE_E_SHAPE;
  for (int a = 0; a < 3; ++a) X[a] = Mi[a * 3] * rhs[0] + Mi[a * 3 + 1] * rhs[1] + Mi[a * 3 + 2] * rhs[2];
  return ORACLE_OK;
}

/* The checker of tdr_ray_closest_point_points (include/tdr_amd.h): the minimiser of the objective
 * RayTriangulator::triangulatePoint hands to cv::LMSolver (src/RayTriangulator.cpp:83-107) in closed form, with
 * the reference's own ray set-up (oracle_create_ray = src/Triangulator.cpp:15-55) and, as `err`, the mean of the
 * reference's distances at that point (distToRay, src/Triangulator.cpp:3-9, summed as RayClosestPoint::compute
 * does, src/RayTriangulator.cpp:21-26).  ONE formula for every rank, written from the start point of the
 * reference's run, x0 = mean of the ray origins (src/RayTriangulator.cpp:90-98):
 *     x = x0 - pinv(H) (H x0 - g),   pinv = the DECOMP_EIG pseudo-inverse the reference's solver steps with
 * (sym3_eig_solve: eigenvalues <= 2 eps sum|w| dropped).  For full-rank H that is H^-1 g; when all rays are parallel
 * it is the point of the line of minimisers that pseudo-inverse steps reach from x0.  Deliberately not the device's
 * route (LDL^T, closed-form rank cascade, folded ray set-up): the two only share the mathematics. */
int oracle_ray_closest_point(const double* cam_aux, const double* uv, int n, double* X, double* err) {
  if (n < 2) return ORACLE_E_TOO_FEW;
  if (n > ORACLE_MAX_CAM) return ORACLE_E_SHAPE;
  oracle_ray rays[ORACLE_MAX_CAM];
  double H[9] = {0}, g[3] = {0}, x0[3] = {0};
  for (int i = 0; i < n; ++i) {
    oracle_create_ray(cam_aux + 10 * i, uv[2 * i], uv[2 * i + 1], rays[i].o, rays[i].d);
    const double* d = rays[i].d;
    double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int a = 0; a < 3; ++a) {
      x0[a] += rays[i].o[a];
      double row = 0;
      for (int b = 0; b < 3; ++b) {
        double m = (a == b ? dd : 0.0) - d[a] * d[b];
        H[a * 3 + b] += m;
        row += m * rays[i].o[b];
      }
      g[a] += row;
    }
  }
  double r[3], step[3];
  for (int a = 0; a < 3; ++a) x0[a] /= n;
  for (int a = 0; a < 3; ++a) r[a] = H[a * 3] * x0[0] + H[a * 3 + 1] * x0[1] + H[a * 3 + 2] * x0[2] - g[a];
  sym3_eig_solve(H, r, step, NULL);
  for (int a = 0; a < 3; ++a) X[a] = x0[a] - step[a];
  if (err) {
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += dist_to_ray(&rays[i], X[0], X[1], X[2]);
    *err = sum / (double)n;
  }
  return ORACLE_OK;
}

/* Batch form with the shapes and the too-few rule of RayTriangulator::triangulatePoints
 * (src/RayTriangulator.cpp:51-81; see oracle_ray_triangulate_points below). */
int oracle_ray_closest_point_points(const double* cam_aux, int n_cam, const double* pts, int n_cam_pts,
                                    int64_t n_sys, double* xyz, double* err, int64_t* first_bad) {
  if (first_bad) *first_bad = -1;
  if (n_cam_pts < 1 || n_cam_pts > n_cam || n_cam_pts > ORACLE_MAX_CAM || n_sys < 0) return ORACLE_E_SHAPE;
  double af[ORACLE_MAX_CAM * 10], uv[ORACLE_MAX_CAM * 2];
  for (int64_t s = 0; s < n_sys; ++s) {
    int n = 0;
    for (int c = 0; c < n_cam_pts; ++c) {
      const double* q = pts + ((int64_t)c * n_sys + s) * 2;
      if (q[0] == -1 || q[1] == -1) continue; /* :66 */
      memcpy(af + 10 * n, cam_aux + 10 * c, 10 * sizeof(double));
      uv[2 * n] = q[0];
      uv[2 * n + 1] = q[1];
      ++n;
    }
    if (n < 2) { /* :72-74 "Too few detections are found" */
      if (first_bad) *first_bad = s;
      return ORACLE_E_TOO_FEW;
    }
    int rc = oracle_ray_closest_point(af, uv, n, xyz + 3 * s, err ? err + s : NULL);
    if (rc != ORACLE_OK) return rc;
  }
  return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * RayTriangulator::triangulatePoints (src/RayTriangulator.cpp:51-81)
 *   cameras indexed by n_cam_pts only (:65) — n_cam_pts > n_cam is UB in the reference
 *   (cameras[n_cam] out of bounds, :69); rejected here.
 * ------------------------------------------------------------------------- */
typedef struct {
  const double *aux, *pts;
  int n_cam_pts;
  int64_t n_sys, s0, s1;
  double *xyz, *err;
  int* iters;
  int* status;  /* optional: ORACLE_LM_* bits per system */
  double* rho;  /* optional: min rho over the lambda == 0 trials per system */
  double* dist; /* optional: |x - x*|_inf per system */
  int perturb_ulps;
  int64_t first_bad;
  int rc;
} ray_job;

static void* ray_worker(void* arg) {
  ray_job* j = (ray_job*)arg;
  double af[ORACLE_MAX_CAM * 10], uv[ORACLE_MAX_CAM * 2];
  j->first_bad = -1;
  j->rc = ORACLE_OK;
  for (int64_t s = j->s0; s < j->s1; ++s) {
    int n = 0;
    for (int c = 0; c < j->n_cam_pts; ++c) {
      const double* q = j->pts + ((int64_t)c * j->n_sys + s) * 2;
      if (q[0] == -1 || q[1] == -1) continue;
      memcpy(af + 10 * n, j->aux + 10 * c, 10 * sizeof(double));
      uv[2 * n] = q[0];
      uv[2 * n + 1] = q[1];
      ++n;
    }
    if (n < 2) {
      j->first_bad = s;
      j->rc = ORACLE_E_TOO_FEW;
      return NULL;
    }
    double e;
    int it;
    oracle_lm_diag dg;
    oracle_ray_triangulate_point_ex(af, uv, n, j->xyz + 3 * s, &e, &it, &dg, j->perturb_ulps);
    if (j->err) j->err[s] = e;
    if (j->iters) j->iters[s] = it;
    if (j->status) j->status[s] = dg.status;
    if (j->rho) j->rho[s] = dg.min_rho;
    if (j->dist) j->dist[s] = dg.dist_to_min;
  }
  return NULL;
}

int oracle_ray_triangulate_points_ex(const double* cam_aux, int n_cam, const double* pts,
                                     int n_cam_pts, int64_t n_sys, double* out_xyz,
                                     double* out_err, int* out_iters, int* out_status, double* out_rho,
                                     double* out_dist, int perturb_ulps, int64_t* first_bad, int n_threads) {
  if (first_bad) *first_bad = -1;
  if (n_cam < 1 || n_cam_pts < 1 || n_cam_pts > n_cam || n_cam > ORACLE_MAX_CAM || n_sys < 0)
    return ORACLE_E_SHAPE;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  ray_job jobs[256];
  for (int t = 0; t < n_threads; ++t) {
    ray_job* j = &jobs[t];
    j->aux = cam_aux; j->pts = pts; j->n_cam_pts = n_cam_pts; j->n_sys = n_sys;
    j->s0 = n_sys * t / n_threads;
    j->s1 = n_sys * (t + 1) / n_threads;
    j->xyz = out_xyz; j->err = out_err; j->iters = out_iters;
    j->status = out_status; j->rho = out_rho; j->dist = out_dist; j->perturb_ulps = perturb_ulps;
    pthread_create(&th[t], NULL, ray_worker, j);
  }
  int rc = ORACLE_OK;
  int64_t fb = -1;
  for (int t = 0; t < n_threads; ++t) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc != ORACLE_OK && rc == ORACLE_OK) {
      rc = jobs[t].rc;
      fb = jobs[t].first_bad;
    }
  }
  if (first_bad) *first_bad = fb;
  return rc;
}

int oracle_ray_triangulate_points_mt(const double* cam_aux, int n_cam, const double* pts,
                                     int n_cam_pts, int64_t n_sys, double* out_xyz,
                                     double* out_err, int* out_iters, int64_t* first_bad,
                                     int n_threads) {
  return oracle_ray_triangulate_points_ex(cam_aux, n_cam, pts, n_cam_pts, n_sys, out_xyz, out_err, out_iters,
                                          NULL, NULL, NULL, 0, first_bad, n_threads);
}

int oracle_ray_triangulate_points(const double* cam_aux, int n_cam, const double* pts,
                                  int n_cam_pts, int64_t n_sys, double* out_xyz, double* out_err,
                                  int* out_iters, int64_t* first_bad) {
  return oracle_ray_triangulate_points_mt(cam_aux, n_cam, pts, n_cam_pts, n_sys, out_xyz, out_err,
                                          out_iters, first_bad, 1);
}

int oracle_sizeof_camera(void) { return (int)sizeof(oracle_camera); }

/* ===========================================================================
 * "Next" row 1: DroneClassifier::fillCombinationQueue
 * (src/DroneClassifier.cpp:156-198) with its prefix-tree Iterator (:43-85), restated literally.
 *   sizes[c] = det_count[c] + 1  ("plus one for no detection", src/DetectionsContainer.cpp:107-114)
 *   combination digit: -1 = camera not reached yet, 0 = no detection, k > 0 = detection k-1
 *   mode 0: matrix triangulator (cam_const = P [n_cam][12]); mode 1: ray (cam_const = aux [n_cam][10])
 * Every visited node with >= 2 chosen detections is triangulated (:180-181); err > error_max cuts the
 * node's whole subtree (:185-187); a full combination (no -1) that is not cut is "pushed" (:190-197).
 * Output in push order: out_comb [cap][n_cam], out_xyz [cap][3], out_err [cap].
 * Returns the number of pushes (may exceed cap; only the first cap are stored); *n_evals = number of
 * triangulatePoint calls.
 * ========================================================================= */
typedef struct {
  int n;
  int comb[ORACLE_MAX_CAM];
  int sizes[ORACLE_MAX_CAM];
  int skip_next;
} oracle_iter;

static int iter_increment(oracle_iter* it) { /* :47-72 */
  if (it->skip_next) {
    it->skip_next = 0;
    return 1;
  }
  for (int i = 0; i < it->n; ++i)
    if (it->comb[i] == -1) {
      it->comb[i]++;
      return 1;
    }
  for (int i = it->n - 1; i >= 0; --i) {
    if (it->comb[i] < it->sizes[i] - 1) {
      it->comb[i]++;
      return 1;
    } else {
      it->comb[i] = -1;
    }
  }
  return 0;
}

static int iter_cut(oracle_iter* it) { /* :74-85 */
  for (int i = it->n - 1; i >= 0; --i) {
    if (it->comb[i] != -1 && it->comb[i] < it->sizes[i] - 1) {
      it->comb[i]++;
      it->skip_next = 1;
      return 1;
    } else {
      it->comb[i] = -1;
    }
  }
  return 0;
}

int64_t oracle_fill_combination_queue_ex(const double* cam_const, int mode, int n_cam, const int32_t* det_count,
                                         const int64_t* det_offset, const double* dets, double error_max,
                                         int min_cameras, int64_t cap, int32_t* out_comb, double* out_xyz,
                                         double* out_err, int64_t* n_evals, int32_t* out_flagged,
                                         int64_t cuts_cap, int32_t* cuts_comb, int64_t* n_cuts);

int64_t oracle_fill_combination_queue(const double* cam_const, int mode, int n_cam,
                                      const int32_t* det_count, const int64_t* det_offset,
                                      const double* dets, double error_max, int min_cameras,
                                      int64_t cap, int32_t* out_comb, double* out_xyz,
                                      double* out_err, int64_t* n_evals) {
  return oracle_fill_combination_queue_ex(cam_const, mode, n_cam, det_count, det_offset, dets, error_max,
                                          min_cameras, cap, out_comb, out_xyz, out_err, n_evals, NULL, 0, NULL, NULL);
}

/* The same walk with the LM status of every solved node tracked (mode 1): out_flagged[i] = 1 when
 * the solve of pushed combination i or of one of its prefixes had ORACLE_LM_CAP / ORACLE_LM_NOT_CONVERGED set
 * (its prune decisions are not reproducible); cuts_comb receives the prefix nodes that were CUT on such a solve
 * (digits, -1 beyond the node's depth). */
int64_t oracle_fill_combination_queue_ex(const double* cam_const, int mode, int n_cam, const int32_t* det_count,
                                         const int64_t* det_offset, const double* dets, double error_max,
                                         int min_cameras, int64_t cap, int32_t* out_comb, double* out_xyz,
                                         double* out_err, int64_t* n_evals, int32_t* out_flagged,
                                         int64_t cuts_cap, int32_t* cuts_comb, int64_t* n_cuts) {
  int status_at[ORACLE_MAX_CAM]; /* LM status of the node last visited at each depth = the current path */
  int64_t cuts = 0;
  memset(status_at, 0, sizeof(status_at));
  oracle_iter it;
  it.n = n_cam;
  it.skip_next = 0;
  for (int c = 0; c < n_cam; ++c) {
    it.comb[c] = -1;
    it.sizes[c] = det_count[c] + 1;
  }
  int64_t pushed = 0, evals = 0;
  const int stride = mode == 0 ? 12 : 10;
  double cf[ORACLE_MAX_CAM * 12], uv[ORACLE_MAX_CAM * 2];
  while (iter_increment(&it)) {
    int count = 0, depth = -1;
    for (int c = 0; c < n_cam; ++c) {
      count += it.comb[c] > 0;
      if (it.comb[c] != -1) depth = c;
    }
    if (depth >= 0) status_at[depth] = 0;
    if (count < 2) continue; /* :168-170 */
    int n = 0;
    for (int c = 0; c < n_cam; ++c) {
      if (it.comb[c] <= 0) continue;
      memcpy(cf + stride * n, cam_const + stride * c, stride * sizeof(double));
      const double* d = dets + 2 * (det_offset[c] + it.comb[c] - 1);
      uv[2 * n] = d[0];
      uv[2 * n + 1] = d[1];
      ++n;
    }
    double X[3], err;
    int flagged = 0;
    if (mode == 0) {
      oracle_matrix_triangulate_point(cf, uv, n, X, &err);
    } else {
      oracle_lm_diag dg;
      oracle_ray_triangulate_point_ex(cf, uv, n, X, &err, NULL, &dg, 0);
      flagged = (dg.status & (ORACLE_LM_CAP | ORACLE_LM_NOT_CONVERGED)) != 0;
    }
    status_at[depth] = flagged;
    ++evals;
    if (err > error_max) { /* :185-187 */
      if (flagged) {
        if (cuts_comb && cuts < cuts_cap)
          for (int c = 0; c < n_cam; ++c) cuts_comb[cuts * n_cam + c] = it.comb[c];
        ++cuts;
      }
      if (!iter_cut(&it)) break;
    } else {
      int full = 1;
      for (int c = 0; c < n_cam; ++c) full &= it.comb[c] != -1;
      if (full && count >= min_cameras) { /* :190-197 */
        if (pushed < cap) {
          for (int c = 0; c < n_cam; ++c) out_comb[pushed * n_cam + c] = it.comb[c];
          memcpy(out_xyz + 3 * pushed, X, sizeof(X));
          out_err[pushed] = err;
          if (out_flagged) {
            int any = 0;
            for (int c = 0; c < n_cam; ++c) any |= status_at[c];
            out_flagged[pushed] = any;
          }
        }
        ++pushed;
      }
    }
  }
  if (n_evals) *n_evals = evals;
  if (n_cuts) *n_cuts = cuts;
  return pushed;
}
