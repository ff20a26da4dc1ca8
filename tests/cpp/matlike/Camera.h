// Test double, CPU tests only: a tdr::Camera whose matrix members are cv::Mat-LIKE objects (rows, cols,
// at<double>(r, c)) under the member names the triangulation path reads, so that the adapter headers can be compiled
// the way they are inside a tree that keeps its OpenCV-based camera class — without OpenCV
// (tests/test_adapter_cpu.py::test_adapter_reads_a_mat_like_camera).  Found before host/Camera.h on the include path.
#pragma once
#include <cstddef>
#include <vector>

namespace testmat {
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> d;
  Mat() = default;
  Mat(int r, int c) : rows(r), cols(c), d(static_cast<std::size_t>(r * c), 0.0) {}
  template <class T>
  T& at(int r, int c) {
    return d[static_cast<std::size_t>(r * cols + c)];
  }
  template <class T>
  const T& at(int r, int c) const {
    return d[static_cast<std::size_t>(r * cols + c)];
  }
};
}  // namespace testmat

namespace tdr {
class Camera {
  int id;

 public:
  testmat::Mat tvec, rquat, cameraPerspectiveMatrix;
  int width = 0, height = 0;
  double fovy = 0;
  explicit Camera(int id_) : id(id_) {}
  int getCamId() const { return id; }
};
}  // namespace tdr
