// tdr_cv_types.h — the two OpenCV value types the Triangulator surface exposes.
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
struct Point3d {
  double x, y, z;
  Point3d() : x(0), y(0), z(0) {}
  Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};
}  // namespace cv
#endif
static_assert(sizeof(cv::Point2d) == 16, "Point2d must be two packed doubles");
static_assert(sizeof(cv::Point3d) == 24, "Point3d must be three packed doubles");
