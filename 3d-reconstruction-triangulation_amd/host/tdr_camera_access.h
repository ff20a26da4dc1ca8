// tdr_camera_access.h — how the adapter reads a tdr::Camera.  The triangulation path needs four things from a
// camera: the 3x4 perspective matrix (src/Triangulator.h:14, src/MatrixTriangulator.cpp:9-11) and, for the ray
// model, rquat, tvec, fovy, width, height (src/Triangulator.cpp:15-55).  They are read through the two functions
// below, which accept BOTH camera classes a tree may hold under the name tdr::Camera:
//   * this repo's host/Camera.h — matrix members are row-major std::array<double, N>;
//   * the reference's own src/Camera.h:33-58 kept as it is (build with -DTDR_WITH_OPENCV and do not copy
//     host/Camera.h) — matrix members are cv::Mat of CV_64F, read with .at<double>() at exactly the indices the
//     reference reads: cameraPerspectiveMatrix (r, c), rquat (i, 0) (src/Triangulator.cpp:17-18), tvec (i)
//     (src/Triangulator.cpp:52-54).
// Nothing here names cv::Mat: any type with rows, cols and at<double>(r, c) is read the same way
// (tests/cpp/camera_access_check.cpp compiles the adapter against such a camera without OpenCV).
#pragma once
#include <array>
#include <cstddef>

#include "tdr_amd.h"

namespace tdr_amd_host {

// element (r, c) of a rows x cols matrix held row-major in a std::array
template <std::size_t N>
inline double mat_at(const std::array<double, N>& m, int r, int c, int cols) {
  return m[static_cast<std::size_t>(r * cols + c)];
}
// ... or in a cv::Mat-like object
template <class M>
inline auto mat_at(const M& m, int r, int c, int /*cols*/) -> decltype(static_cast<double>(m.template at<double>(r, c))) {
  return m.template at<double>(r, c);
}

// element i of a vector: std::array, or a cv::Mat-like row or column (the reference indexes tvec with one index)
template <std::size_t N>
inline double vec_at(const std::array<double, N>& v, int i) {
  return v[static_cast<std::size_t>(i)];
}
template <class M>
inline auto vec_at(const M& v, int i) -> decltype(static_cast<double>(v.template at<double>(i, 0))) {
  return v.rows == 1 ? v.template at<double>(0, i) : v.template at<double>(i, 0);
}

// P[12], row-major 3x4
template <class Cam>
inline void projection_matrix(const Cam& cam, double* P) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) P[4 * r + c] = mat_at(cam.cameraPerspectiveMatrix, r, c, 4);
}
template <class Mat>
inline void projection_matrix_of(const Mat& m, double* P) {  // CamPointPair::projectionMatrix()
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) P[4 * r + c] = mat_at(m, r, c, 4);
}

// aux[TDR_CAM_AUX_DOUBLES] = [rquat (4, as given), tvec (3), fovy, width, height]: the ray-model constants of
// include/tdr_amd.h
template <class Cam>
inline void ray_aux(const Cam& cam, double* aux) {
  for (int i = 0; i < 4; ++i) aux[i] = vec_at(cam.rquat, i);
  for (int i = 0; i < 3; ++i) aux[4 + i] = vec_at(cam.tvec, i);
  aux[7] = cam.fovy;
  aux[8] = static_cast<double>(cam.width);
  aux[9] = static_cast<double>(cam.height);
}
static_assert(TDR_CAM_AUX_DOUBLES == 10, "ray_aux fills ten doubles");

}  // namespace tdr_amd_host
