// MatrixTriangulator.h — reference src/MatrixTriangulator.h:6-25 over the MI355X engine.
#pragma once
#include "Triangulator.h"

class MatrixTriangulator : public Triangulator {
 public:
  std::string type = "matrix";  // the reference's unused public member (src/MatrixTriangulator.h:9)

  explicit MatrixTriangulator(std::vector<const tdr::Camera*> cameras_, int device = 0) {
    init(std::move(cameras_), "matrix", device);
  }

  // src/MatrixTriangulator.cpp:3-62: every pair brings its own camera (projectionMatrix()).
  std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) override {
    std::vector<double> P(images.size() * 12), uv(images.size() * 2);
    for (std::size_t i = 0; i < images.size(); ++i) {
      tdr_amd_host::projection_matrix_of(images[i].projectionMatrix(), P.data() + 12 * i);
      uv[2 * i] = images[i].point.x;
      uv[2 * i + 1] = images[i].point.y;
    }
    double X[3], err = 0;
    int rc = tdr_matrix_triangulate_point(ctx_, P.data(), uv.data(), static_cast<int>(images.size()), X, &err);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
    return {cv::Point3d(X[0], X[1], X[2]), err};
  }

  // src/MatrixTriangulator.cpp:70-100
  std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) override {
    return runBatch(false, points, "Too few rays are found");
  }
  std::vector<cv::Point3d> triangulatePoints(DetectionsContainer& container, bool narrow = false) override {
    return runBatchFlat(false, container, narrow, "Too few rays are found");
  }
};
