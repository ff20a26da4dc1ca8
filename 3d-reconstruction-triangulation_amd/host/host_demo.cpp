// host_demo.cpp — test driver for the C++ adapter: uses ONLY the reference's class surface
// (tdr::createCamera, MatrixTriangulator / RayTriangulator, triangulatePoints, triangulatePoint,
// getType, getDistFromRay) the way the reference's main.cpp / README does, reads a scene written by
// tests/test_gpu_host_adapter.py and writes the results back for comparison with the CPU oracle.
//
//   host_demo <in.bin> <out.bin>
// modes 2 (matrix) / 3 (ray): DroneClassifier::classifyDrones, see run_classifier() below
// in : i32 mode(0 matrix,1 ray) | i32 n_cam | i32 n_rows | i64 n_sys | i32 ragged_row(-1 none)
//      n_cam x { i32 width, i32 height, f64 focal, f64 tvec[3], f64 quat[4] }
//      f64 pts[n_rows][n_sys][2]
// out: i32 status(0 ok, 1 exception) | i32 len | char msg[len] | type string (i32 len + chars)
//      i64 n | f64 xyz[n][3] | i32 n_single | n_single x f64[4] (x,y,z,err) | f64 dist
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "DroneClassifier.h"
#include "MatrixTriangulator.h"
#include "RayTriangulator.h"
#include "utils.h"

template <class T>
static T rd(std::ifstream& f) {
  T v;
  f.read(reinterpret_cast<char*>(&v), sizeof(T));
  return v;
}
template <class T>
static void wr(std::ofstream& f, const T& v) {
  f.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
static void wr_str(std::ofstream& f, const std::string& s) {
  wr<std::int32_t>(f, static_cast<std::int32_t>(s.size()));
  f.write(s.data(), static_cast<std::streamsize>(s.size()));
}

// in (after the cameras): i32 n_frames | i32 n_drones | i32 det_count[n_frames][n_cam] | f64 dets[total][2]
// out: i32 status | msg | i32 n_drones | per drone: i64 n | f64 xyz[n][3]
static int run_classifier(std::ifstream& in, std::ofstream& out, int mode, int n_cam,
                          const std::vector<const tdr::Camera*>& cameras) {
  const auto n_frames = rd<std::int32_t>(in);
  const auto n_drones = rd<std::int32_t>(in);
  std::vector<std::int32_t> counts(static_cast<std::size_t>(n_frames) * n_cam);
  in.read(reinterpret_cast<char*>(counts.data()), static_cast<std::streamsize>(counts.size() * 4));
  DetectionsContainer container(n_cam);
  for (int f = 0; f < n_frames; ++f) {
    container.addEmptyFrame();
    for (int c = 0; c < n_cam; ++c)
      for (int d = 0; d < counts[static_cast<std::size_t>(f) * n_cam + c]; ++d) {
        const double x = rd<double>(in), y = rd<double>(in);
        container.addDetectionToCamera(cv::Point2d(x, y), c);
      }
  }
  std::int32_t status = 0;
  std::string msg;
  std::vector<std::vector<cv::Point3d>> paths;
  try {
    // object lifetimes exactly as in the reference's main (src/main.cpp:54-88): raw pointer, stack classifier,
    // and the triangulator deleted while the classifier is still in scope
    Triangulator* triangulator;
    if (mode == 2)
      triangulator = new MatrixTriangulator(cameras);
    else
      triangulator = new RayTriangulator(cameras);
    DroneClassifier classifier(triangulator, static_cast<std::size_t>(n_drones));
    classifier.classifyDrones(container, paths);
    delete triangulator;
  } catch (const std::runtime_error& e) {
    status = 1;
    msg = e.what();
  }
  wr<std::int32_t>(out, status);
  wr_str(out, msg);
  wr<std::int32_t>(out, static_cast<std::int32_t>(paths.size()));
  for (const auto& p : paths) {
    wr<std::int64_t>(out, static_cast<std::int64_t>(p.size()));
    out.write(reinterpret_cast<const char*>(p.data()), static_cast<std::streamsize>(p.size() * 24));
  }
  return 0;
}

// host_demo --one-drone <cameras.xcp> <dl_data dir> <matrix|ray> <out.bin> <vicon order 0|1> <form>
// The reference's library use for a one-drone recording (README.md:82-85):
//     loadCamerasXML -> DetectionsContainer(path, 0, 7) -> triangulator.triangulatePoints(...)
//   form 0: triangulatePoints(container.getDataForTriangulation())         — the reference call, nested vectors
//   form 1: triangulatePoints(container)                                   — flat rows in pinned memory
//   form 2: triangulatePoints(container, narrow = true)                    — flat int16 rows (matrix path)
// out: i32 status | msg | i64 n | f64 xyz[n][3] | f64 parse_seconds | f64 solve_seconds
static int run_one_drone(int argc, char** argv) {
  if (argc != 8) return 2;
  const std::string kind = argv[4];
  const int form = std::atoi(argv[7]);
  std::ofstream out(argv[5], std::ios::binary);
  std::int32_t status = 0;
  std::string msg;
  std::vector<cv::Point3d> path;
  double t_parse = 0, t_solve = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<const tdr::Camera*> cameras;
  try {
    cameras = loadCamerasXML(argv[2], std::atoi(argv[6]) != 0);
    std::unique_ptr<Triangulator> triangulator;
    if (kind == "matrix")
      triangulator.reset(new MatrixTriangulator(cameras));
    else
      triangulator.reset(new RayTriangulator(cameras));
    tdr_io::RowAllocator pinned;
    pinned.alloc = [](std::size_t n) {
      void* p = nullptr;
      return tdr_host_alloc(static_cast<std::int64_t>(n), &p) == TDR_OK ? p : nullptr;
    };
    pinned.free = [](void* p) { (void)tdr_host_free(p); };
    const double t0 = now();
    DetectionsContainer container(argv[3], 0, 7, form == 0 ? tdr_io::RowAllocator() : pinned);
    const double t1 = now();
    if (form == 0)
      path = triangulator->triangulatePoints(container.getDataForTriangulation());
    else
      path = triangulator->triangulatePoints(container, form == 2);
    t_solve = now() - t1;
    t_parse = t1 - t0;
  } catch (const std::runtime_error& e) {
    status = 1;
    msg = e.what();
  }
  wr<std::int32_t>(out, status);
  wr_str(out, msg);
  wr<std::int64_t>(out, static_cast<std::int64_t>(path.size()));
  out.write(reinterpret_cast<const char*>(path.data()), static_cast<std::streamsize>(path.size() * 24));
  wr<double>(out, t_parse);
  wr<double>(out, t_solve);
  for (auto* c : cameras) delete c;
  return 0;
}

// --static-stress <threads> <calls per thread>: the static Triangulator::getDistFromRay on the real engine from several
// threads at once while the main thread creates and destroys triangulators (round 4: the static runs on the adapter's
// own helper context and holds no adapter lock across the device round trip).  Every call must return the value a
// single-threaded call returns, bit for bit.  Prints "ok <calls> <triangulators churned> <reference distance>".
static int run_static_stress(int argc, char** argv) {
  const int n_threads = argc > 2 ? std::atoi(argv[2]) : 6;
  const int n_calls = argc > 3 ? std::atoi(argv[3]) : 200;
  std::vector<const tdr::Camera*> cameras;
  for (int i = 0; i < 4; ++i) {
    const double a = 0.7 * i;
    cameras.push_back(tdr::createCamera(i, 1924, 1082, 1081.0810546875, {5000.0 * std::cos(a), 5000.0 * std::sin(a), 1500.0},
                                        {std::cos(a / 2), 0.0, 0.0, std::sin(a / 2)}));
  }
  const Triangulator::CamPointPair pair{cameras[1], cv::Point2d(873.0, 455.0)};
  const cv::Point3d point(120.0, -340.0, 910.0);
  const double expect = Triangulator::getDistFromRay(pair, point);  // also creates the helper context
  if (!(expect > 0.0) || !std::isfinite(expect)) return 3;
  std::atomic<long> bad{0}, calls{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> threads;
  for (int t = 0; t < n_threads; ++t)
    threads.emplace_back([&] {
      while (!go.load()) std::this_thread::yield();
      for (int i = 0; i < n_calls; ++i) {
        const double d = Triangulator::getDistFromRay(pair, point);
        if (std::memcmp(&d, &expect, sizeof(d)) != 0) bad.fetch_add(1);
        calls.fetch_add(1);
      }
    });
  go.store(true);
  long churned = 0;
  while (calls.load() < static_cast<long>(n_threads) * n_calls) {  // triangulators come and go meanwhile
    MatrixTriangulator m(cameras);
    RayTriangulator r(cameras);
    (void)m.triangulatePoint({{cameras[0], cv::Point2d(900, 500)}, {cameras[2], cv::Point2d(1000, 520)}});
    churned += 2;
  }
  for (auto& th : threads) th.join();
  for (auto* c : cameras) delete c;
  if (bad.load() != 0) {
    std::printf("mismatch in %ld of %ld calls\n", bad.load(), calls.load());
    return 4;
  }
  std::printf("ok %ld %ld %.17g\n", calls.load(), churned, expect);
  return 0;
}

// --single-call-latency <calls>: what ONE call through the plugin surface costs — the reference's own, unchanged caller
// issues exactly these (triangulatePoint per candidate combination, getDistFromRay per detection,
// src/DroneClassifier.cpp:180-181,231).  Times triangulatePoint of both triangulators for 2..8 observations and the
// static getDistFromRay, `calls` calls each after a warm-up, and prints one JSON object (median / min / p90 in
// microseconds).  Run it twice for the A/B: TDR_POINT_ARGS=0 is round 4's copy-up / launch / copy-down / synchronise.
static int run_single_call_latency(int argc, char** argv) {
  const int n_calls = argc > 2 ? std::atoi(argv[2]) : 2000;
  std::vector<const tdr::Camera*> cameras;
  for (int i = 0; i < 8; ++i) {
    const double a = 0.7 * i;
    cameras.push_back(tdr::createCamera(i, 1924, 1082, 1081.0810546875, {5000.0 * std::cos(a), 5000.0 * std::sin(a), 1500.0},
                                        {std::cos(a / 2), 0.0, 0.0, std::sin(a / 2)}));
  }
  MatrixTriangulator m(cameras);
  RayTriangulator r(cameras);
  auto stats = [&](std::vector<double>& us) {
    std::sort(us.begin(), us.end());
    char buf[160];
    std::snprintf(buf, sizeof buf, "{\"median_us\": %.2f, \"min_us\": %.2f, \"p90_us\": %.2f}", us[us.size() / 2], us[0],
                  us[us.size() * 9 / 10]);
    return std::string(buf);
  };
  auto time_calls = [&](auto&& call) {
    for (int i = 0; i < 200; ++i) call();
    std::vector<double> us(static_cast<std::size_t>(n_calls));
    for (int i = 0; i < n_calls; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      call();
      us[static_cast<std::size_t>(i)] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return stats(us);
  };
  const char* mode = std::getenv("TDR_POINT_ARGS");
  std::printf("{\"TDR_POINT_ARGS\": \"%s\", \"calls\": %d", mode ? mode : "default", n_calls);
  double sink = 0;
  for (int n_obs : {2, 4, 8}) {
    std::vector<Triangulator::CamPointPair> images;
    for (int c = 0; c < n_obs; ++c) images.push_back({cameras[static_cast<std::size_t>(c)], cv::Point2d(900.0 + 7 * c, 500.0 + 3 * c)});
    std::printf(", \"matrix_triangulatePoint_%d\": %s", n_obs, time_calls([&] { sink += m.triangulatePoint(images).second; }).c_str());
    std::printf(", \"ray_triangulatePoint_%d\": %s", n_obs, time_calls([&] { sink += r.triangulatePoint(images).second; }).c_str());
  }
  const Triangulator::CamPointPair pair{cameras[1], cv::Point2d(873.0, 455.0)};
  const cv::Point3d point(120.0, -340.0, 910.0);
  std::printf(", \"getDistFromRay\": %s", time_calls([&] { sink += Triangulator::getDistFromRay(pair, point); }).c_str());
  std::printf(", \"checksum\": %.17g}\n", sink);
  for (auto* c : cameras) delete c;
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "--single-call-latency") return run_single_call_latency(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "--one-drone") return run_one_drone(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "--static-stress") return run_static_stress(argc, argv);
  if (argc != 3) {
    std::fprintf(stderr, "usage: host_demo <in.bin> <out.bin>\n");
    return 2;
  }
  std::ifstream in(argv[1], std::ios::binary);
  std::ofstream out(argv[2], std::ios::binary);
  const auto mode = rd<std::int32_t>(in);
  const auto n_cam = rd<std::int32_t>(in);
  const auto n_rows = rd<std::int32_t>(in);
  const auto n_sys = rd<std::int64_t>(in);
  const auto ragged_row = rd<std::int32_t>(in);

  std::vector<const tdr::Camera*> cameras;
  for (int c = 0; c < n_cam; ++c) {
    const auto w = rd<std::int32_t>(in), h = rd<std::int32_t>(in);
    const auto focal = rd<double>(in);
    std::array<double, 3> t;
    std::array<double, 4> q;
    for (auto& v : t) v = rd<double>(in);
    for (auto& v : q) v = rd<double>(in);
    cameras.push_back(tdr::createCamera(c, w, h, focal, t, q));
  }
  if (mode >= 2) {
    int rc = run_classifier(in, out, mode, n_cam, cameras);
    for (auto* c : cameras) delete c;
    return rc;
  }
  std::vector<std::vector<cv::Point2d>> points(n_rows);
  for (int r = 0; r < n_rows; ++r) {
    points[r].resize(static_cast<std::size_t>(n_sys));
    in.read(reinterpret_cast<char*>(points[r].data()), static_cast<std::streamsize>(n_sys * 16));
  }
  if (ragged_row >= 0 && !points[ragged_row].empty()) points[ragged_row].pop_back();

  std::int32_t status = 0;
  std::string msg, type;
  std::vector<cv::Point3d> path;
  std::vector<std::pair<cv::Point3d, double>> singles;
  double dist = 0;
  try {
    std::unique_ptr<Triangulator> triangulator;
    if (mode == 0)
      triangulator.reset(new MatrixTriangulator(cameras));
    else
      triangulator.reset(new RayTriangulator(cameras));
    type = triangulator->getType();
    path = triangulator->triangulatePoints(points);
    // a few single-system solves with an explicit camera subset, as DroneClassifier issues them
    const int n_used = n_rows < n_cam ? n_rows : n_cam;
    for (std::int64_t s = 0; s < n_sys && s < 4; ++s) {
      std::vector<Triangulator::CamPointPair> images;
      for (int c = n_used - 1; c >= 0; --c) {  // reversed order on purpose
        const cv::Point2d& p = points[c][static_cast<std::size_t>(s)];
        if (p.x == -1 || p.y == -1) continue;
        images.push_back({triangulator->getCamera(c), p});
      }
      if (images.size() >= 2) singles.push_back(triangulator->triangulatePoint(images));
    }
    if (n_sys > 0 && !path.empty())
      dist = Triangulator::getDistFromRay({cameras[0], points[0][0]}, path[0]);
  } catch (const std::runtime_error& e) {
    status = 1;
    msg = e.what();
  }
  wr<std::int32_t>(out, status);
  wr_str(out, msg);
  wr_str(out, type);
  wr<std::int64_t>(out, static_cast<std::int64_t>(path.size()));
  out.write(reinterpret_cast<const char*>(path.data()), static_cast<std::streamsize>(path.size() * 24));
  wr<std::int32_t>(out, static_cast<std::int32_t>(singles.size()));
  for (const auto& s : singles) {
    wr<double>(out, s.first.x);
    wr<double>(out, s.first.y);
    wr<double>(out, s.first.z);
    wr<double>(out, s.second);
  }
  wr<double>(out, dist);
  // ray mode: the status-bearing form of the batch call on the same input, appended
  //   i32 status2 | i64 n | i32 lm_status[n] | i32 iters[n] | f64 xyz[n][3]
  if (mode == 1 && status == 0 && ragged_row < 0) {
    std::int32_t status2 = 0;
    RayTriangulator::LmReport rep;
    try {
      RayTriangulator ray(cameras);
      rep = ray.triangulatePointsWithStatus(points);
    } catch (const std::runtime_error&) {
      status2 = 1;
    }
    wr<std::int32_t>(out, status2);
    wr<std::int64_t>(out, static_cast<std::int64_t>(rep.points.size()));
    out.write(reinterpret_cast<const char*>(rep.status.data()), static_cast<std::streamsize>(rep.status.size() * 4));
    out.write(reinterpret_cast<const char*>(rep.iters.data()), static_cast<std::streamsize>(rep.iters.size() * 4));
    out.write(reinterpret_cast<const char*>(rep.points.data()), static_cast<std::streamsize>(rep.points.size() * 24));
  }
  for (auto* c : cameras) delete c;
  return 0;
}
