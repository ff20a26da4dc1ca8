// Camera.h — host mirror of tdr::Camera (reference src/Camera.h:33-298) for the fields the
// triangulation path reads.  The derivation itself (compCamParams chain, src/Camera.h:177-187)
// is the library's tdr_camera_derive; this class only carries its result under the reference's
// member names so code written against tdr::Camera keeps compiling.  A tree that keeps the reference's own Camera.h
// (cv::Mat members) does not need this file: the adapter reads either class through tdr_camera_access.h.
#pragma once
#include <array>
#include <cstddef>
#include <stdexcept>

#include "tdr_amd.h"

namespace tdr {

class Camera {
  int id;

 public:
  std::array<double, 3> rvec{};                      // rotation vector (unused by the path)
  std::array<double, 3> tvec{};                      // XCP POSITION
  std::array<double, 4> rquat{};                     // w-first, as given
  std::array<double, 3> camPos{};                    // -R^-1 tvec
  std::array<double, 9> cameraMatrix{};              // K, row-major
  std::array<double, 12> cameraExtrinsicMatrix{};    // [R^-1 | camPos], row-major 3x4
  std::array<double, 12> cameraPerspectiveMatrix{};  // P = K E, row-major 3x4

  int width = 0, height = 0, diagonal = 0, cx = 0, cy = 0;
  double fx = 0, fy = 0, fovx = 0, fovy = 0;
  int fps = 0;

  explicit Camera(int id_) : id(id_) {}
  int getCamId() const { return id; }

  // width, height, fx (= FOCAL_LENGTH), tvec and rquat must be set first (src/utils.cpp:97-104)
  void compCamParams() {
    tdr_camera c;
    int rc = tdr_camera_derive(id, width, height, fx, tvec.data(), rquat.data(), &c);
    if (rc == TDR_E_CAMERA) throw std::runtime_error(tdr_strerror(rc));
    if (rc != TDR_OK) throw std::runtime_error(tdr_strerror(rc));
    cx = c.cx; cy = c.cy; diagonal = c.diagonal;
    fx = c.fx; fy = c.fy; fovx = c.fovx; fovy = c.fovy;
    for (int i = 0; i < 3; ++i) { rvec[i] = c.rvec[i]; camPos[i] = c.campos[i]; }
    for (int i = 0; i < 9; ++i) cameraMatrix[i] = c.K[i];
    for (int i = 0; i < 12; ++i) { cameraExtrinsicMatrix[i] = c.E[i]; cameraPerspectiveMatrix[i] = c.P[i]; }
  }

};

// createCamera (reference src/utils.cpp:94-107)
inline const Camera* createCamera(int id, std::size_t width, std::size_t height, double focalLength,
                                  const std::array<double, 3>& translation,
                                  const std::array<double, 4>& rotation) {
  Camera* cam = new Camera(id);
  cam->width = static_cast<int>(width);
  cam->height = static_cast<int>(height);
  cam->fx = focalLength;
  cam->tvec = translation;
  cam->rquat = rotation;
  cam->compCamParams();
  return cam;
}

}  // namespace tdr
