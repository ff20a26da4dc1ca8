// host_io_check.cpp — CPU-only driver for the host-side I/O code (no GPU call is made): XCP camera
// ingest, CSV detection ingest, PLY output and the error statistics, dumped as text for
// tests/test_host_io_cpu.py to compare with the Python restatements in oracle/io_oracle.py.
//   host_io_check cameras <file.xcp> <0|1 vicon order>
//   host_io_check csv <dir>
//   host_io_check ply <out.ply>            (points read from stdin: "x y z" per line)
//   host_io_check stats                    (stdin: "lx ly lz px py pz" per line)
//   host_io_check oneframe <dir>           (getDataForTriangulation of a one-drone recording)
#include <cstdio>
#include <iostream>

#include "DetectionsContainer.h"
#include "utils.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  std::printf("%s", "");
  try {
    if (mode == "cameras" && argc == 4) {
      auto cams = loadCamerasXML(argv[2], std::atoi(argv[3]) != 0);
      for (const auto* c : cams) {
        std::printf("%d %d %d %d %d %.17g %.17g %.17g %.17g", c->getCamId(), c->width, c->height, c->cx, c->cy,
                    c->fx, c->fy, c->fovx, c->fovy);
        for (double v : c->cameraPerspectiveMatrix) std::printf(" %.17g", v);
        std::printf("\n");
        delete c;
      }
    } else if (mode == "csv" && argc == 3) {
      DetectionsContainer dc(argv[2], 0, 7);
      std::printf("%d %d\n", dc.getFrameCount(), dc.getCamCount());
      for (int f = 0; f < dc.getFrameCount(); ++f)
        for (int c = 0; c < dc.getCamCount(); ++c) {
          std::printf("%d", dc.detCountForCam(c, f));
          for (int d = 0; d < dc.detCountForCam(c, f); ++d)
            std::printf(" %.17g %.17g", dc.getRecord(c, f, d).x, dc.getRecord(c, f, d).y);
          std::printf("\n");
        }
    } else if (mode == "oneframe" && argc == 3) {
      DetectionsContainer dc(argv[2], 0, 7);
      auto pts = dc.getDataForTriangulation();
      for (const auto& row : pts) {
        for (const auto& p : row) std::printf("%.17g %.17g ", p.x, p.y);
        std::printf("\n");
      }
    } else if (mode == "ply" && argc == 3) {
      std::vector<cv::Point3d> pts;
      double x, y, z;
      while (std::cin >> x >> y >> z) pts.emplace_back(x, y, z);
      writeOutputFile(argv[2], pts);
    } else if (mode == "stats") {
      Path label, pred;
      double a, b, c, d, e, f;
      while (std::cin >> a >> b >> c >> d >> e >> f) {
        label.emplace_back(a, b, c);
        pred.emplace_back(d, e, f);
      }
      const double mean = calculateError(label, pred);
      std::printf("%.17g %.17g %.17g %.17g\n", mean, calculateMedian(label, pred), calculateStd(label, pred, mean),
                  calcualteQuarterDeviation(label, pred));
    } else {
      return 2;
    }
  } catch (const std::runtime_error& e) {
    std::printf("EXCEPTION %s\n", e.what());
  }
  return 0;
}
