// Triangulator.h — the reference's plugin surface (src/Triangulator.h:9-59) over the MI355X
// engine.  Same public names, argument meaning and exceptions; every solve is a HIP kernel
// launch behind include/tdr_amd.h.  No CPU path: construction throws when no gfx950 device or
// libtdr_amd.so is available.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "Camera.h"
#include "tdr_amd.h"
#include "tdr_cv_types.h"

class Triangulator {
 public:
  struct CamPointPair {  // src/Triangulator.h:11-15
    const tdr::Camera* camera;
    cv::Point2d point;
    const std::array<double, 12>& projectionMatrix() const { return camera->cameraPerspectiveMatrix; }
  };

  struct Ray {  // src/Triangulator.h:17-20
    cv::Point3d origin;
    cv::Point3d dir;
  };

 protected:
  std::string type_;
  std::vector<const tdr::Camera*> cameras;
  tdr_ctx* ctx_ = nullptr;

  // Cameras are non-owning and const (src/Triangulator.h:26); their constants are snapshotted
  // onto the GPU once, here.
  void init(std::vector<const tdr::Camera*> cameras_, const char* type, int device) {
    cameras = std::move(cameras_);
    type_ = type;
    std::vector<double> P(cameras.size() * 12), aux(cameras.size() * TDR_CAM_AUX_DOUBLES);
    for (std::size_t c = 0; c < cameras.size(); ++c) {
      for (int k = 0; k < 12; ++k) P[12 * c + k] = cameras[c]->cameraPerspectiveMatrix[k];
      auto a = cameras[c]->rayAux();
      for (int k = 0; k < TDR_CAM_AUX_DOUBLES; ++k) aux[TDR_CAM_AUX_DOUBLES * c + k] = a[k];
    }
    int rc = tdr_create(P.data(), aux.data(), static_cast<int>(cameras.size()), device, &ctx_);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  }

  // The rows of vector<vector<Point2d>> are handed to the engine where they lie (one pointer per camera
  // row, no flattening copy); the equal-length check is the reference's
  // (src/MatrixTriangulator.cpp:72-76, src/RayTriangulator.cpp:54-58).
  std::vector<cv::Point3d> runBatch(bool ray, const std::vector<std::vector<cv::Point2d>>& points,
                                    const char* too_few_message) {
    if (points.empty())  // undefined behaviour in the reference (size()-1 underflow); rejected
      throw std::runtime_error("Every camera should have the same number of points");
    const std::size_t n = points[0].size();
    std::vector<const double*> rows(points.size());
    for (std::size_t c = 0; c < points.size(); ++c) {
      if (points[c].size() != n) throw std::runtime_error("Every camera should have the same number of points");
      rows[c] = reinterpret_cast<const double*>(points[c].data());
    }
    std::vector<cv::Point3d> result(n);
    std::int64_t first_bad = -1;
    double* out = reinterpret_cast<double*>(result.data());
    int rc = ray ? tdr_ray_triangulate_points_rows(ctx_, rows.data(), static_cast<int>(points.size()),
                                                   static_cast<std::int64_t>(n), out, nullptr, &first_bad)
                 : tdr_matrix_triangulate_points_rows(ctx_, rows.data(), static_cast<int>(points.size()),
                                                      static_cast<std::int64_t>(n), out, nullptr, &first_bad);
    if (rc == TDR_E_TOO_FEW) throw std::runtime_error(too_few_message);
    if (rc == TDR_E_SHAPE) throw std::runtime_error(tdr_strerror(rc));
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
    return result;
  }

 public:
  Triangulator() = default;
  Triangulator(const Triangulator&) = delete;
  Triangulator& operator=(const Triangulator&) = delete;
  virtual ~Triangulator() { tdr_destroy(ctx_); }

  std::vector<const tdr::Camera*> getCameras() { return cameras; }

  virtual std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) = 0;

  virtual std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) = 0;

  const tdr::Camera* getCamera(int camera) { return cameras[camera]; }

  std::string getType() { return type_; }

  tdr_ctx* context() { return ctx_; }  // for device-resident / multi-GPU callers

  // src/Triangulator.cpp:57-61.  Static in the reference, so it has no context to launch on: it
  // runs the batched device helper on a process-wide context of the pair's camera (n = 1).
  static double getDistFromRay(CamPointPair pair, cv::Point3d point);
};

inline double Triangulator::getDistFromRay(CamPointPair pair, cv::Point3d point) {
  // one process-wide helper context on GPU 0 (the function is static: it has no triangulator); the camera
  // constants travel with the call, so any camera works
  static tdr_ctx* helper = [] {
    const double P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    tdr_ctx* ctx = nullptr;
    int rc = tdr_create(P, nullptr, 1, 0, &ctx);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
    return ctx;
  }();
  const auto aux = pair.camera->rayAux();
  const double uv[2] = {pair.point.x, pair.point.y}, pt[3] = {point.x, point.y, point.z};
  double dist = 0;
  int rc = tdr_dist_from_ray(helper, aux.data(), uv, pt, &dist);
  if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  return dist;
}
