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

#include <array>
#include <atomic>
#include <mutex>

#include "Camera.h"
#include "DetectionsContainer.h"
#include "tdr_amd.h"
#include "tdr_camera_access.h"
#include "tdr_cv_types.h"

class Triangulator {
 public:
  struct CamPointPair {  // src/Triangulator.h:11-15
    const tdr::Camera* camera;
    cv::Point2d point;
    // std::array<double, 12> with host/Camera.h, cv::Mat with the reference's own Camera.h (tdr_camera_access.h)
    const decltype(tdr::Camera::cameraPerspectiveMatrix)& projectionMatrix() const {
      return camera->cameraPerspectiveMatrix;
    }
  };

  struct Ray {  // src/Triangulator.h:17-20
    cv::Point3d origin;
    cv::Point3d dir;
  };

 protected:
  std::string type_;
  // "vector of projection matrices in mocap system" (src/Triangulator.h:24-25): declared by the reference, filled by
  // nobody (its triangulators read the matrix through CamPointPair::projectionMatrix()); kept so that a subclass
  // naming it compiles.  Element type = the camera class's own matrix type (cv::Mat with the reference's Camera.h).
  std::vector<decltype(tdr::Camera::cameraPerspectiveMatrix)> projection_matrices;
  std::vector<const tdr::Camera*> cameras;
  tdr_ctx* ctx_ = nullptr;

  // ---- the reference's protected static helpers (src/Triangulator.h:27-42, src/Triangulator.cpp:3-55), same names and
  // signatures: what a subclass written like the reference's own RayTriangulator::RayClosestPoint
  // (src/RayTriangulator.cpp:21-44) calls per ray.  Scalar host functions of the engine library (tdr_create_ray,
  // tdr_dist_to_ray, ...: the very expressions the kernels evaluate, csrc/tdr_math.h) — not the oracle, and not a
  // compute path: a subclass that wants throughput batches through triangulatePoints.
  static double distToRay(const Ray& r, const cv::Vec3d& p, bool squared = false) {
    const double o[3] = {r.origin.x, r.origin.y, r.origin.z}, d[3] = {r.dir.x, r.dir.y, r.dir.z};
    const double q[3] = {p[0], p[1], p[2]};
    return tdr_dist_to_ray(o, d, q, squared ? 1 : 0);
  }
  static double distToRay_(const Ray& r, const cv::Vec3d p, bool squared = false) { return distToRay(r, p, squared); }
  // `quat` has the camera class's own type (std::array<double, 4> here, a 4 x 1 cv::Mat in the reference)
  static cv::Vec3d rotatePointByQuaternion(cv::Vec3d point, const decltype(tdr::Camera::rquat) quat) {
    double q[4], v[3] = {point[0], point[1], point[2]}, out[3];
    for (int i = 0; i < 4; ++i) q[i] = tdr_amd_host::vec_at(quat, i);
    tdr_rotate_point_by_quaternion(q, v, out);
    return cv::Vec3d(out[0], out[1], out[2]);
  }
  static cv::Vec3d calculateRayDirectionForPixel(const tdr::Camera* cam, const cv::Point2d& point) {
    std::array<double, TDR_CAM_AUX_DOUBLES> aux;
    tdr_amd_host::ray_aux(*cam, aux.data());
    const double uv[2] = {point.x, point.y};
    double d[3];
    tdr_ray_direction_for_pixel(aux.data(), uv, d);
    return cv::Vec3d(d[0], d[1], d[2]);
  }
  static Ray createRayForPoint(const tdr::Camera* cam, const cv::Point2d& point) {
    std::array<double, TDR_CAM_AUX_DOUBLES> aux;
    tdr_amd_host::ray_aux(*cam, aux.data());
    const double uv[2] = {point.x, point.y};
    double o[3], d[3];
    tdr_create_ray(aux.data(), uv, o, d);
    return {cv::Point3d(o[0], o[1], o[2]), cv::Point3d(d[0], d[1], d[2])};
  }

  // Cameras are non-owning and const (src/Triangulator.h:26); their constants are snapshotted
  // onto the GPU once, here.
  void init(std::vector<const tdr::Camera*> cameras_, const char* type, int device) {
    cameras = std::move(cameras_);
    type_ = type;
    std::vector<double> P(cameras.size() * 12), aux(cameras.size() * TDR_CAM_AUX_DOUBLES);
    for (std::size_t c = 0; c < cameras.size(); ++c) {
      tdr_amd_host::projection_matrix(*cameras[c], P.data() + 12 * c);
      tdr_amd_host::ray_aux(*cameras[c], aux.data() + TDR_CAM_AUX_DOUBLES * c);
    }
    int rc = tdr_create(P.data(), aux.data(), static_cast<int>(cameras.size()), device, &ctx_);
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
    std::lock_guard<std::mutex> lock(registry_mutex());
    live_contexts().push_back({ctx_, device});
  }

  // The triangulators alive in this process and the GPU each was created on.  Bookkeeping only: the registry lock is
  // never held across a call into the engine.  Its one reader is the static getDistFromRay, which looks up WHERE the
  // process computes (so that its helper context does not open another GPU) the first time it is called.
  struct LiveContext {
    tdr_ctx* ctx;
    int device;
  };
  static std::mutex& registry_mutex() {
    static std::mutex m;
    return m;
  }
  static std::vector<LiveContext>& live_contexts() {
    static std::vector<LiveContext> v;
    return v;
  }

  // The context behind the static getDistFromRay: created by its first call, on the GPU of a live triangulator if
  // there is one (else GPU 0), owned by the adapter for the rest of the process and destroyed at exit.  The camera
  // constants travel with every call, so the context needs none of its own.  Round 3 borrowed a live triangulator's
  // context under the registry lock — every static call in the process then queued on ONE lock for a device round
  // trip, behind whatever that triangulator's stream was doing, and constructors / destructors queued behind the
  // calls.  Now the static touches no triangulator: calls serialise only on the helper context's own mutex (inside
  // tdr_dist_from_ray), and a triangulator can be created or destroyed at any time.  A caller with many points uses
  // tdr_dist_from_ray_batch on its own context instead (one launch for all of them).
  // Device: the GPU of the first triangulator alive at the moment of the first successful creation, else GPU 0 (also
  // for a triangulator spread over every GPU, device -1).  A creation that FAILS (device busy, out of memory) is not
  // remembered: the next static call tries again — only the created context is cached, for good.
  struct HelperContext {
    std::mutex mu;                      // creation only; calls go through the atomic pointer
    std::atomic<tdr_ctx*> ctx{nullptr};
    HelperContext() = default;
    HelperContext(const HelperContext&) = delete;
    ~HelperContext() { tdr_destroy(ctx.load()); }
  };
  static HelperContext& helper_context() {
    static HelperContext h;
    return h;
  }

  static void check(int rc, const char* too_few_message) {
    if (rc == TDR_E_TOO_FEW) throw std::runtime_error(too_few_message);
    if (rc == TDR_E_SHAPE) throw std::runtime_error(tdr_strerror(rc));
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  }

  // One-drone recording straight from the container's flat rows[camera][frame] block (pinned when the
  // container was given the pinned allocator): no nested vectors are built, the engine DMAs the block as it
  // lies.  narrow = true uses the int16 rows when every coordinate fits (matrix path; bit-identical results).
  std::vector<cv::Point3d> runBatchFlat(bool ray, DetectionsContainer& container, bool narrow,
                                        const char* too_few_message) {
    const double* rows = container.flatRows();  // throws the reference's one-drone exception
    const int n_cam = container.getCamCount();
    const std::int64_t n = container.getFrameCount();
    std::vector<cv::Point3d> result(static_cast<std::size_t>(n));
    double* out = reinterpret_cast<double*>(result.data());
    std::int64_t first_bad = -1;
    const std::int16_t* rows16 = (!ray && narrow) ? container.flatRowsI16() : nullptr;
    int rc;
    if (rows16)
      rc = tdr_matrix_triangulate_points_i16(ctx_, rows16, n_cam, n, out, nullptr, &first_bad);
    else if (ray)
      rc = tdr_ray_triangulate_points(ctx_, rows, n_cam, n, out, nullptr, &first_bad);
    else
      rc = tdr_matrix_triangulate_points(ctx_, rows, n_cam, n, out, nullptr, &first_bad);
    check(rc, too_few_message);
    return result;
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
    check(rc, too_few_message);
    return result;
  }

 public:
  Triangulator() = default;
  Triangulator(const Triangulator&) = delete;
  Triangulator& operator=(const Triangulator&) = delete;
  virtual ~Triangulator() {
    {
      std::lock_guard<std::mutex> lock(registry_mutex());
      auto& v = live_contexts();
      for (std::size_t i = 0; i < v.size(); ++i)
        if (v[i].ctx == ctx_) {
          v.erase(v.begin() + static_cast<std::ptrdiff_t>(i));
          break;
        }
    }
    tdr_destroy(ctx_);
  }

  std::vector<const tdr::Camera*> getCameras() { return cameras; }

  virtual std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) = 0;

  virtual std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) = 0;

  // Engine-side overload of the reference's triangulatePoints(container.getDataForTriangulation())
  // (README.md:82-85): same result, same exceptions, without the nested vectors in between.  NOT pure: the
  // reference's surface has exactly two pure virtuals (src/Triangulator.h:46-52), and a third-party subclass written
  // against it must keep compiling — the default is the reference's own composition; MatrixTriangulator and
  // RayTriangulator override it with the flat-row route.
  virtual std::vector<cv::Point3d> triangulatePoints(DetectionsContainer& container, bool narrow = false) {
    (void)narrow;
    return triangulatePoints(container.getDataForTriangulation());
  }

  const tdr::Camera* getCamera(int camera) { return cameras[camera]; }

  std::string getType() { return type_; }

  tdr_ctx* context() { return ctx_; }  // for device-resident / multi-GPU callers

  // src/Triangulator.cpp:57-61.  Static in the reference, so it belongs to no triangulator: it runs on the adapter's
  // own helper context (helper_context() above) and holds no adapter lock while the device works.
  // tests/cpp/adapter_registry_check.cpp calls it from several threads while triangulators are created and deleted,
  // under ThreadSanitizer / AddressSanitizer.
  static double getDistFromRay(CamPointPair pair, cv::Point3d point);
};

inline double Triangulator::getDistFromRay(CamPointPair pair, cv::Point3d point) {
  std::array<double, TDR_CAM_AUX_DOUBLES> aux;
  tdr_amd_host::ray_aux(*pair.camera, aux.data());
  const double uv[2] = {pair.point.x, pair.point.y}, pt[3] = {point.x, point.y, point.z};
  HelperContext& h = helper_context();
  tdr_ctx* ctx = h.ctx.load(std::memory_order_acquire);
  if (!ctx) {
    std::lock_guard<std::mutex> create_lock(h.mu);
    ctx = h.ctx.load(std::memory_order_acquire);
    if (!ctx) {
      int device = 0;
      {
        std::lock_guard<std::mutex> lock(registry_mutex());
        if (!live_contexts().empty() && live_contexts().front().device >= 0) device = live_contexts().front().device;
      }
      const double P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      const int rc = tdr_create(P, nullptr, 1, device, &ctx);
      if (rc != TDR_OK)  // no gfx950 device (there is no CPU path) or a transient failure: thrown, not cached
        throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
      h.ctx.store(ctx, std::memory_order_release);
    }
  }
  double dist = 0;
  const int rc = tdr_dist_from_ray(ctx, aux.data(), uv, pt, &dist);
  if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  return dist;
}
