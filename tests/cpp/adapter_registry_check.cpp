// The C++ adapter's static Triangulator::getDistFromRay and its context registry under ThreadSanitizer /
// AddressSanitizer (tests/test_adapter_cpu.py), against the stub C ABI of tdr_amd_stub.cpp: several threads call the
// static while other threads create and delete triangulators.  History: round 2 copied a live triangulator's context
// pointer under the registry lock and used it after releasing the lock (a triangulator deleted meanwhile freed it
// mid-call); round 3 held the lock for the whole borrowed call (correct, but every static call in the process queued
// on one lock for a device round trip); round 4 gives the static a process-lifetime helper context of its own — it
// touches no triangulator and holds no adapter lock while the device works.  Checked here besides the sanitizers'
// silence: exactly ONE helper context is created however many threads race for the first call, it is destroyed at
// exit (leak check), and construction / destruction of triangulators is never blocked for the length of a static call.
// Also here: a reference-style third-party subclass (only the reference's two pure virtuals, src/Triangulator.h:46-52)
// compiles, constructs without an engine context, and the container overload falls back to its own
// triangulatePoints(getDataForTriangulation()).
#include <atomic>
#include <cstdio>
#include <thread>

#include "MatrixTriangulator.h"
#include "RayTriangulator.h"

extern std::atomic<long> tdr_stub_created;  // tdr_amd_stub.cpp: contexts handed out by tdr_create

class ThirdPartyTriangulator : public Triangulator {  // written against the reference's header
 public:
  explicit ThirdPartyTriangulator(std::vector<const tdr::Camera*> cameras_) {
    type_ = "matrix";
    cameras = cameras_;
  }
  std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) override {
    return {cv::Point3d(images[0].point.x, images[0].point.y, 0), 0.0};
  }
  std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) override {
    std::vector<cv::Point3d> out;
    for (const auto& p : points[0]) out.emplace_back(p.x, p.y, 7.0);
    return out;
  }
};

// A subclass written the way the reference's own RayTriangulator is (src/RayTriangulator.cpp:8-47,83-107): it builds
// its rays with the PROTECTED static createRayForPoint, evaluates residuals with distToRay and the central-difference
// Jacobian with distToRay_ on brace-initialised cv::Vec3d points, and may name every protected member of
// src/Triangulator.h:22-42 (against the round-4 header this class did not compile).
class RayStyleTriangulator : public Triangulator {
  class RayClosestPoint {  // the reference's callback, minus cv::LMSolver::Callback
   public:
    explicit RayClosestPoint(const std::vector<Ray>& rays, const double epsilon = 1e-4, const bool squared = false)
        : rays_(rays), epsilon_(epsilon), squared_(squared) {}
    double compute(double x, double y, double z, std::vector<double>* J) const {
      const cv::Vec3d currentPoint(x, y, z);
      double summedError = 0;
      for (const Ray& r : rays_) summedError += distToRay(r, currentPoint);
      if (J)
        for (const Ray& r : rays_) {
          J->push_back((distToRay_(r, {x + epsilon_, y, z}, squared_) - distToRay_(r, {x - epsilon_, y, z}, squared_)) /
                       (2 * epsilon_));
          J->push_back((distToRay_(r, {x, y + epsilon_, z}, squared_) - distToRay_(r, {x, y - epsilon_, z}, squared_)) /
                       (2 * epsilon_));
          J->push_back((distToRay_(r, {x, y, z + epsilon_}, squared_) - distToRay_(r, {x, y, z - epsilon_}, squared_)) /
                       (2 * epsilon_));
        }
      return summedError / (double)rays_.size();
    }

   private:
    std::vector<Ray> rays_;
    const double epsilon_;
    const bool squared_;
  };

 public:
  explicit RayStyleTriangulator(std::vector<const tdr::Camera*> cameras_) {
    cameras = cameras_;
    type_ = "ray";
    projection_matrices.clear();  // declared by the reference, used by nobody: it must at least exist
  }
  std::pair<cv::Point3d, double> triangulatePoint(std::vector<CamPointPair> images) override {
    std::vector<Ray> rays;
    cv::Point3d initGuess(0, 0, 0);
    for (auto& im : images) {
      rays.push_back(createRayForPoint(im.camera, im.point));
      initGuess.x += rays.back().origin.x;
    }
    std::vector<double> J;
    const double err = RayClosestPoint(rays).compute(initGuess.x, 0, 0, &J);
    const cv::Vec3d pix = calculateRayDirectionForPixel(images[0].camera, images[0].point);
    const cv::Vec3d rot = rotatePointByQuaternion(pix, images[0].camera->rquat);
    return {cv::Point3d(rot[0], rot[1], (double)J.size()), err};
  }
  std::vector<cv::Point3d> triangulatePoints(std::vector<std::vector<cv::Point2d>> points) override {
    return std::vector<cv::Point3d>(points[0].size());
  }
};

int main() {
  std::vector<const tdr::Camera*> cams;
  for (int i = 0; i < 2; ++i) cams.push_back(tdr::createCamera(i, 1924, 1082, 1081.0, {100.0 * i, 0, 0}, {1, 0, 0, 0}));
  std::atomic<bool> stop{false};
  std::atomic<long> calls{0};
  // callers of the static helper: sometimes a triangulator is alive, sometimes none — it makes no difference any more
  std::vector<std::thread> callers;
  for (int t = 0; t < 6; ++t)
    callers.emplace_back([&] {
      while (!stop.load()) {
        const double d = Triangulator::getDistFromRay({cams[0], cv::Point2d(3, 4)}, cv::Point3d(5, 6, 7));
        if (d != 5.0) std::abort();
        calls.fetch_add(1);
      }
    });
  // churn: triangulators come and go on two more threads
  std::vector<std::thread> churn;
  for (int t = 0; t < 2; ++t)
    churn.emplace_back([&, t] {
      for (int i = 0; i < 400; ++i) {
        if ((i + t) % 2) {
          MatrixTriangulator m(cams);
          (void)m.triangulatePoint({{cams[0], cv::Point2d(1, 2)}, {cams[1], cv::Point2d(3, 4)}});
        } else {
          RayTriangulator r(cams);
          (void)r.getType();
        }
        if (i % 50 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));  // windows with no live context
      }
    });
  for (auto& th : churn) th.join();
  stop.store(true);
  for (auto& th : callers) th.join();
  if (calls.load() == 0) return 3;
  // one helper context for the whole process: 800 triangulators + 1 helper were created, not one per call
  if (tdr_stub_created.load() != 2 * 400 + 1) {
    std::printf("contexts created: %ld\n", tdr_stub_created.load());
    return 6;
  }

  // a reference-style subclass
  {
    ThirdPartyTriangulator tp(cams);
    Triangulator* base = &tp;
    if (base->getType() != "matrix" || base->getCameras().size() != 2) return 4;
    auto out = base->triangulatePoints({{cv::Point2d(1, 2), cv::Point2d(3, 4)}, {cv::Point2d(5, 6), cv::Point2d(7, 8)}});
    if (out.size() != 2 || out[1].z != 7.0) return 5;
  }
  // ... and one that uses the protected static helpers the way the reference's RayTriangulator does
  {
    RayStyleTriangulator rs(cams);
    Triangulator* base = &rs;
    // stub arithmetic: origin = tvec, dir = (u, v, 1), dist = (p.x - o.x) + 10 dir.x; pixel direction = (u + .5, v + .5,
    // fovy); rotation = q0 p + q_{1..3}
    auto out = base->triangulatePoint({{cams[0], cv::Point2d(3, 4)}, {cams[1], cv::Point2d(5, 6)}});
    // initGuess.x = 0 + 100; residuals (100 - 0) + 30 and (100 - 100) + 50 -> mean 90; J has 3 entries per ray
    if (out.second != 90.0 || out.first.z != 6.0) {
      std::printf("ray-style subclass: err %g, J entries %g\n", out.second, out.first.z);
      return 7;
    }
    if (out.first.x != 3.5 || out.first.y != 4.5) return 8;  // identity quaternion (1, 0, 0, 0): q0 p + 0
  }
  for (const auto* c : cams) delete c;
  std::printf("ok\n");
  return 0;
}
