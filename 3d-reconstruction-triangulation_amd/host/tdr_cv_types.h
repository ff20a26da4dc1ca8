// tdr_cv_types.h — the OpenCV value types the Triangulator surface exposes (public: Point2d, Point3d; protected
// static helpers: Vec3d).
// With -DTDR_WITH_OPENCV the real headers are used (drop-in inside the reference tree);
// otherwise layout-compatible stand-ins: cv::Point2d = {double x, y} (16 B),
// cv::Point3d = {double x, y, z} (24 B) — exactly the memory the C ABI consumes/produces.
#pragma once
#ifdef TDR_WITH_OPENCV
#include <opencv2/core.hpp>
#else
namespace cv {
struct Point2d {
  double x, y;
  Point2d() : x(0), y(0) {}
  Point2d(double x_, double y_) : x(x_), y(y_) {}
};
// cv::Vec3d as the protected static helpers use it (src/Triangulator.h:27-42): three doubles, operator[],
// brace-initialisable from three values, convertible from / to Point3d like OpenCV's Vec<double, 3>
struct Vec3d {
  double val[3];
  Vec3d() : val{0, 0, 0} {}
  Vec3d(double v0, double v1, double v2) : val{v0, v1, v2} {}
  double& operator[](int i) { return val[i]; }
  const double& operator[](int i) const { return val[i]; }
};
struct Point3d {
  double x, y, z;
  Point3d() : x(0), y(0), z(0) {}
  Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  Point3d(const Vec3d& v) : x(v[0]), y(v[1]), z(v[2]) {}  // cv::Point3_(const Vec<_Tp, 3>&)
  operator Vec3d() const { return Vec3d(x, y, z); }       // cv::Point3_::operator Vec<_Tp, 3>()
};
}  // namespace cv
#endif
static_assert(sizeof(cv::Point2d) == 16, "Point2d must be two packed doubles");
static_assert(sizeof(cv::Point3d) == 24, "Point3d must be three packed doubles");
static_assert(sizeof(cv::Vec3d) == 24, "Vec3d must be three packed doubles");
