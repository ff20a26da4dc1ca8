// Camera.h — host mirror of tdr::Camera (reference src/Camera.h:33-298) for the fields the
// triangulation path reads.  The derivation itself (compCamParams chain, src/Camera.h:177-187)
// is the library's tdr_camera_derive; this class only carries its result under the reference's
// member names so code written against tdr::Camera keeps compiling.
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

  // [rquat, tvec, fovy, width, height]: the ray-model constants of include/tdr_amd.h
  std::array<double, TDR_CAM_AUX_DOUBLES> rayAux() const {
    return {rquat[0], rquat[1], rquat[2], rquat[3], tvec[0], tvec[1], tvec[2], fovy,
            static_cast<double>(width), static_cast<double>(height)};
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
