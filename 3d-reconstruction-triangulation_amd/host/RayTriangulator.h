// RayTriangulator.h — reference src/RayTriangulator.h:13-41 over the MI355X engine.
#pragma once
#include <cstring>

#include "Triangulator.h"

#define THRESHOLD 1e-4       // finite-difference step of the reference Jacobian (src/RayTriangulator.h:9)
#define MAX_ITERATIONS 1000  // src/RayTriangulator.h:10
#define SQUARED false        // src/RayTriangulator.h:11

class RayTriangulator : public Triangulator {
 public:
  explicit RayTriangulator(std::vector<const tdr::Camera*> cameras_, int device = 0) {
    init(std::move(cameras_), "ray", device);
  }

  // src/RayTriangulator.cpp:51-81
  std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) override {
    return runBatch(true, points, "Too few detections are found");
  }
  std::vector<cv::Point3d> triangulatePoints(DetectionsContainer& container, bool narrow = false) override {
    return runBatchFlat(true, container, narrow, "Too few detections are found");
  }

  // triangulatePoints plus what cv::LMSolver::run knows and the reference throws away (engine-side addition):
  // per frame the mean ray distance at the solver's last compute(), the iteration count, and the TDR_LM_* status
  // word — 0 where the run ended at the minimiser of its objective; TDR_LM_CAP / TDR_LM_NOT_CONVERGED where the
  // reference's answer is not reproducible by any implementation (include/tdr_amd.h).
  struct LmReport {
    std::vector<cv::Point3d> points;
    std::vector<double> err;
    std::vector<std::int32_t> iters, status;
  };
  LmReport triangulatePointsWithStatus(const std::vector<std::vector<cv::Point2d>>& points) {
    if (points.empty()) throw std::runtime_error("Every camera should have the same number of points");
    const std::size_t n = points[0].size();
    std::vector<double> flat(points.size() * n * 2);
    for (std::size_t c = 0; c < points.size(); ++c) {
      if (points[c].size() != n) throw std::runtime_error("Every camera should have the same number of points");
      if (n) std::memcpy(flat.data() + c * n * 2, points[c].data(), n * 16);
    }
    LmReport r;
    r.points.resize(n);
    r.err.resize(n);
    r.iters.resize(n);
    r.status.resize(n);
    std::int64_t first_bad = -1;
    check(tdr_ray_triangulate_points_ex(ctx_, flat.data(), static_cast<int>(points.size()),
                                        static_cast<std::int64_t>(n), reinterpret_cast<double*>(r.points.data()),
                                        r.err.data(), r.iters.data(), r.status.data(), &first_bad),
          "Too few detections are found");
    return r;
  }

  // Engine-side addition: literal = true makes triangulatePoints / triangulatePointsWithStatus differentiate exactly as
  // RayClosestPoint::compute does (six shifted distance evaluations with the reference's roundings,
  // src/RayTriangulator.cpp:28-44) instead of through the exact gradient: 2-4 x the time, the reference's status-word
  // statistics on two-ray systems (tdr_set_ray_jacobian, include/tdr_amd.h).  Converged results are the same either way.
  void setLiteralJacobian(bool literal) {
    const int rc = tdr_set_ray_jacobian(ctx_, literal ? 1 : 0);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  }

  // Engine-side addition: the minimiser of the objective triangulatePoint hands to cv::LMSolver
  // (sum_i |d_i x (x - o_i)|^2, src/RayTriangulator.cpp:83-107) in closed form, one HBM-bound pass
  // (tdr_ray_closest_point_points, include/tdr_amd.h).  NOT what triangulatePoints returns where the reference's LM
  // run stops early — there it returns the LM iterate; where triangulatePointsWithStatus reports status 0 the two
  // agree to 1e-7.  Shapes and exceptions as triangulatePoints.  err (optional): mean ray distance per frame.
  std::vector<cv::Point3d> closestPoints(const std::vector<std::vector<cv::Point2d>>& points,
                                         std::vector<double>* err = nullptr) {
    if (points.empty()) throw std::runtime_error("Every camera should have the same number of points");
    const std::size_t n = points[0].size();
    std::vector<const double*> rows(points.size());
    for (std::size_t c = 0; c < points.size(); ++c) {
      if (points[c].size() != n) throw std::runtime_error("Every camera should have the same number of points");
      rows[c] = reinterpret_cast<const double*>(points[c].data());
    }
    std::vector<cv::Point3d> result(n);
    if (err) err->resize(n);
    std::int64_t first_bad = -1;
    check(tdr_ray_closest_point_points_rows(ctx_, rows.data(), static_cast<int>(points.size()),
                                            static_cast<std::int64_t>(n), reinterpret_cast<double*>(result.data()),
                                            err ? err->data() : nullptr, &first_bad),
          "Too few detections are found");
    return result;
  }

  // src/RayTriangulator.cpp:83-107
  std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) override {
    std::vector<double> aux(images.size() * TDR_CAM_AUX_DOUBLES), uv(images.size() * 2);
    for (std::size_t i = 0; i < images.size(); ++i) {
      tdr_amd_host::ray_aux(*images[i].camera, aux.data() + TDR_CAM_AUX_DOUBLES * i);
      uv[2 * i] = images[i].point.x;
      uv[2 * i + 1] = images[i].point.y;
    }
    double X[3], err = 0;
    int rc = tdr_ray_triangulate_point(ctx_, aux.data(), uv.data(), static_cast<int>(images.size()), X, &err);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
    return {cv::Point3d(X[0], X[1], X[2]), err};
  }
};
