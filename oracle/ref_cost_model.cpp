// ref_cost_model.cpp — a COST MODEL of the reference's batch loop on the CPU.  TEST INFRASTRUCTURE ONLY (like everything
// under oracle/): only bench.py's cpu_baseline leg and tests/ may load it; nothing under the package does.
//
// Why it exists (VERDICT r4 "missing" #6, SURVEY.md §8d flavour (i)): the timed CPU baseline of bench.py is the flat-array
// port of oracle/tdr_oracle.c, which keeps the reference's arithmetic (one-sided Jacobi SVD pseudo-inverse per system)
// but none of its data movement, and so flatters the reference.  The reference itself cannot be built here (OpenCV is
// absent, oracle/Makefile), so this file restates the SHAPE of its loop — what is allocated, copied and reference-counted
// per point — around the very same arithmetic (oracle_pinv_jacobi_mx3 of libtdr_oracle.so, same summation order
// everywhere: results are bit-identical to the port, tests/test_oracle_cpu.py):
//
//   MatrixTriangulator::triangulatePoints  src/MatrixTriangulator.cpp:70-100
//     - `points` arrives BY VALUE: a deep copy of vector<vector<Point2d>>                        (:70-71)
//     - per point a fresh std::vector<CamPointPair>, grown by push_back (no reserve)              (:82-89)
//     - handed BY VALUE to triangulatePoint: a second copy                                        (:96, :3-4)
//     - result.push_back without reserve                                                          (:78, :96)
//   MatrixTriangulator::triangulatePoint   src/MatrixTriangulator.cpp:3-62
//     - two default-constructed matrices, then two heap matrices assigned over them               (:5-8)
//     - CamPointPair::projectionMatrix() returns the camera's matrix header BY VALUE              (src/Triangulator.h:14):
//       14 calls per camera (:16-49), each a header copy + atomic reference-count increment and decrement
//     - invert -> a heap matrix; inv_L * bias, A * X - bias, transpose, errorT * error -> one heap matrix each (:52-58)
//
// `Mat` below is a minimal reference-counted dense matrix header with the fields a cv::Mat header copy moves (flags,
// dims, rows, cols, four data pointers, allocator, the shared block, size / step arrays) — written for this model, not
// OpenCV code.  What the model does NOT carry: OpenCV's allocator (fastMalloc's alignment bookkeeping), MatExpr
// construction, its gemm dispatch and the AutoBuffer logic inside cv::invert / SVD::compute (stack buffers at these
// sizes) — all of which cost the real program more, never less.  So the figure is an UPPER bound on the reference's
// throughput that is tighter than the flat port's, still labelled kind "port" in bench.py; it is a baseline, never a
// target.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <utility>
#include <vector>

extern "C" void oracle_pinv_jacobi_mx3(const double* A, int m, double* pinv);  // oracle/tdr_oracle.c

namespace {

struct Shared {  // the block a matrix header points at: reference count + payload
  std::atomic<int> refs;
  double* data;
};

class Mat {
 public:
  Mat() = default;
  Mat(int r, int c) { create(r, c); }
  Mat(const Mat& m)
      : flags(m.flags), dims(m.dims), rows(m.rows), cols(m.cols), data(m.data), datastart(m.datastart),
        dataend(m.dataend), datalimit(m.datalimit), allocator(m.allocator), u(m.u) {
    if (u) u->refs.fetch_add(1, std::memory_order_acq_rel);
    size[0] = m.size[0]; size[1] = m.size[1];
    step[0] = m.step[0]; step[1] = m.step[1];
  }
  Mat& operator=(const Mat& m) {
    if (this != &m) {
      if (m.u) m.u->refs.fetch_add(1, std::memory_order_acq_rel);
      release();
      flags = m.flags; dims = m.dims; rows = m.rows; cols = m.cols;
      data = m.data; datastart = m.datastart; dataend = m.dataend; datalimit = m.datalimit;
      allocator = m.allocator; u = m.u;
      size[0] = m.size[0]; size[1] = m.size[1];
      step[0] = m.step[0]; step[1] = m.step[1];
    }
    return *this;
  }
  ~Mat() { release(); }
  void create(int r, int c) {
    release();
    u = static_cast<Shared*>(std::malloc(sizeof(Shared)));
    if (!u) throw std::bad_alloc();
    new (&u->refs) std::atomic<int>(1);
    u->data = static_cast<double*>(std::malloc(sizeof(double) * (size_t)r * c + 16));
    if (!u->data) throw std::bad_alloc();
    rows = size[0] = r; cols = size[1] = c; dims = 2; flags = 0x42ff4006;
    step[0] = sizeof(double) * c; step[1] = sizeof(double);
    data = datastart = u->data; dataend = datalimit = u->data + (size_t)r * c;
  }
  double& at(int i, int j) { return *reinterpret_cast<double*>(reinterpret_cast<char*>(data) + step[0] * i + step[1] * j); }
  double at(int i, int j) const { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(data) + step[0] * i + step[1] * j); }
  double& at(int i) { return data[i]; }
  int flags = 0, dims = 0, rows = 0, cols = 0;
  double *data = nullptr, *datastart = nullptr, *dataend = nullptr, *datalimit = nullptr;
  void* allocator = nullptr;
  Shared* u = nullptr;
  int size[2] = {0, 0};
  size_t step[2] = {0, 0};

 private:
  void release() {
    if (u && u->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      std::free(u->data);
      std::free(u);
    }
    u = nullptr; data = datastart = dataend = datalimit = nullptr; rows = cols = 0;
  }
};

struct Point2d { double x, y; };
struct Point3d { double x, y, z; };

struct Camera {  // the one member of tdr::Camera the path reads (src/Camera.h: cameraPerspectiveMatrix, a cv::Mat)
  Mat cameraPerspectiveMatrix;
};

struct CamPointPair {  // src/Triangulator.h:11-15
  const Camera* camera;
  Point2d point;
  Mat projectionMatrix() { return camera->cameraPerspectiveMatrix; }
};

// dst = a * b (+ sign_c * c): a fresh heap matrix per product, as every MatExpr evaluation makes one
Mat gemm(const Mat& a, const Mat& b, const Mat* c, double sign_c) {
  Mat d(a.rows, b.cols);
  for (int i = 0; i < a.rows; ++i)
    for (int j = 0; j < b.cols; ++j) {
      double s = 0;
      for (int k = 0; k < a.cols; ++k) s += a.at(i, k) * b.at(k, j);
      d.at(i, j) = c ? s + sign_c * c->at(i, j) : s;
    }
  return d;
}

Mat transposed(const Mat& a) {
  Mat t(a.cols, a.rows);
  for (int i = 0; i < a.rows; ++i)
    for (int j = 0; j < a.cols; ++j) t.at(j, i) = a.at(i, j);
  return t;
}

void invert_svd(const Mat& a, Mat& inv) {  // cv::invert(A, inv, DECOMP_SVD): the arithmetic is the oracle's
  inv.create(a.cols, a.rows);
  oracle_pinv_jacobi_mx3(a.data, a.rows, inv.data);
}

class MatrixTriangulatorModel {
 public:
  explicit MatrixTriangulatorModel(std::vector<const Camera*> cameras_) { cameras = cameras_; }

  std::pair<Point3d, double> triangulatePoint(std::vector<CamPointPair> images) {  // :3-62
    Mat coefficients, bias;
    coefficients = Mat(2 * (int)images.size(), 3);
    bias = Mat(2 * (int)images.size(), 1);
    for (int n = 0; n < (int)images.size(); n++) {
      if (images[n].point.x == -1 || images[n].point.y == -1) continue;
      // every matrix element read goes through a by-value header, as in the reference (:16-49)
      coefficients.at(0 + n * 2, 0) = images[n].projectionMatrix().at(0, 0) - images[n].point.x * images[n].projectionMatrix().at(2, 0);
      coefficients.at(0 + n * 2, 1) = images[n].projectionMatrix().at(0, 1) - images[n].point.x * images[n].projectionMatrix().at(2, 1);
      coefficients.at(0 + n * 2, 2) = images[n].projectionMatrix().at(0, 2) - images[n].point.x * images[n].projectionMatrix().at(2, 2);
      coefficients.at(1 + n * 2, 0) = images[n].projectionMatrix().at(1, 0) - images[n].point.y * images[n].projectionMatrix().at(2, 0);
      coefficients.at(1 + n * 2, 1) = images[n].projectionMatrix().at(1, 1) - images[n].point.y * images[n].projectionMatrix().at(2, 1);
      coefficients.at(1 + n * 2, 2) = images[n].projectionMatrix().at(1, 2) - images[n].point.y * images[n].projectionMatrix().at(2, 2);
      bias.at(0 + n * 2, 0) = images[n].point.x * images[n].projectionMatrix().at(2, 3) - images[n].projectionMatrix().at(0, 3);
      bias.at(1 + n * 2, 0) = images[n].point.y * images[n].projectionMatrix().at(2, 3) - images[n].projectionMatrix().at(1, 3);
    }
    Mat inv_L;
    invert_svd(coefficients, inv_L);                 // :52-53
    Mat X = gemm(inv_L, bias, nullptr, 0);           // :54
    Mat error = gemm(coefficients, X, &bias, -1.0);  // :55
    Mat errorT = transposed(error);                  // :56-57
    error = gemm(errorT, error, nullptr, 0);         // :58
    const double err = std::sqrt(error.at(0) / (2 * images.size()));
    return {Point3d{X.at(0), X.at(1), X.at(2)}, err};
  }

  std::vector<Point3d> triangulatePoints(std::vector<std::vector<Point2d>> points) {  // :70-100
    for (size_t i = 0; i + 1 < points.size(); i++)
      if (points[i].size() != points[i + 1].size())
        throw std::runtime_error("Every camera should have the same number of points");
    std::vector<Point3d> result;
    for (size_t n_point = 0; n_point < points[0].size(); n_point++) {
      std::vector<CamPointPair> frame{};
      for (size_t n_cam = 0; n_cam < std::min(points.size(), cameras.size()); n_cam++) {
        if (points[n_cam][n_point].x == -1 || points[n_cam][n_point].y == -1) continue;
        frame.push_back(CamPointPair{cameras[n_cam], points[n_cam][n_point]});
      }
      if (frame.size() < 2) throw std::runtime_error("Too few rays are found");
      result.push_back(triangulatePoint(frame).first);
    }
    return result;
  }

 private:
  std::vector<const Camera*> cameras;
};

}  // namespace

// P [n_cam][12], pts [n_cam_pts][n_sys][2] -> out_xyz [n_sys][3].  0 ok, -1 shape, -2 too few rays (as the oracle's codes).
// The caller's flat arrays are turned into the reference's containers OUTSIDE what a caller of the reference would time
// only in so far as the reference's caller owns such containers already: the nested vector is built here first, and
// the by-value call below then copies it, as MatrixTriangulator::triangulatePoints does.
extern "C" int oracle_refcost_matrix_triangulate_points(const double* P, int n_cam, const double* pts, int n_cam_pts,
                                                        int64_t n_sys, double* out_xyz, double* seconds_in_call) {
  if (n_cam < 1 || n_cam_pts < 1 || n_sys < 0) return -1;
  std::vector<Camera> cams(n_cam);
  std::vector<const Camera*> cam_ptrs;
  for (int c = 0; c < n_cam; ++c) {
    cams[c].cameraPerspectiveMatrix.create(3, 4);
    std::memcpy(cams[c].cameraPerspectiveMatrix.data, P + 12 * c, 12 * sizeof(double));
    cam_ptrs.push_back(&cams[c]);
  }
  std::vector<std::vector<Point2d>> points(n_cam_pts);
  for (int c = 0; c < n_cam_pts; ++c) {
    points[c].resize(n_sys);
    std::memcpy(points[c].data(), pts + (int64_t)c * n_sys * 2, sizeof(Point2d) * n_sys);
  }
  MatrixTriangulatorModel tri(cam_ptrs);
  try {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::vector<Point3d> r = tri.triangulatePoints(points);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_in_call) *seconds_in_call = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    std::memcpy(out_xyz, r.data(), sizeof(Point3d) * r.size());
  } catch (const std::runtime_error& e) {
    return std::strstr(e.what(), "Too few") ? -2 : -1;
  }
  return 0;
}
