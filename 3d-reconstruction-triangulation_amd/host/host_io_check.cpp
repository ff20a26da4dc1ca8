// host_io_check.cpp — CPU-only driver for the host-side I/O code (no GPU call is made): XCP camera
// ingest, CSV detection ingest, PLY output and the error statistics, dumped as text for
// tests/test_host_io_cpu.py to compare with the Python restatements in oracle/io_oracle.py.
//   host_io_check cameras <file.xcp> <0|1 vicon order>
//   host_io_check csv <dir>
//   host_io_check csvwin <dir> <offset> <startFrame> <endFrame> <threads>   (the same dump with a frame window)
//   host_io_check ply <out.ply>            (points read from stdin: "x y z" per line)
//   host_io_check stats                    (stdin: "lx ly lz px py pz" per line)
//   host_io_check oneframe <dir>           (getDataForTriangulation of a one-drone recording)
//   host_io_check csvseq <dir>             (the same dump through a reference-style sequential reader)
//   host_io_check gencsv <dir> <n_cam> <n_frames> <seed>   (synthetic one-drone recording in the on-disk format)
//   host_io_check parsebench <dir> <threads> <reps>        (MB/s: parallel parser vs the sequential reader; equality)
//   host_io_check labels <labels.csv> <frequency> <n_drones> <markers>
//   host_io_check evaluate <pred.ply> <labels.csv> <frequency> <n_drones> <markers>
#include <chrono>
#include <cstdio>
#include <iostream>
#include <filesystem>
#include <random>

#include "DetectionsContainer.h"
#include "utils.h"

namespace {

// Reference-style reader: getline / istringstream / stoi into nested vectors, one thread
// (the shape of src/DetectionsContainer.cpp:19-76).  Lives only in this check tool: it is the baseline the
// parallel parser is timed against and compared with.
typedef std::vector<std::vector<std::vector<cv::Point2d>>> Nested;  // [camera][frame][detection]
Nested read_files_sequential(const std::vector<std::string>& files, int offset, int recordSize) {
  Nested data;
  for (const auto& name : files) {
    std::ifstream file(name);
    data.emplace_back();
    auto& frames = data.back();
    std::string line, token;
    int n_line = 0, frame = -1;
    while (std::getline(file, line)) {
      if (offset > n_line++) continue;
      std::vector<int> fields;
      std::istringstream iss(line);
      while (std::getline(iss, token, ','))
        if (!token.empty()) fields.push_back(std::stoi(token));
      if (fields.empty()) throw std::runtime_error("Invalid CSV file!");
      for (int i = 0; i < fields[0] - frame - 1; ++i) frames.emplace_back();
      frame = fields[0];
      if ((fields.size() - 1) % static_cast<std::size_t>(recordSize) != 0) throw std::runtime_error("Invalid CSV file!");
      frames.emplace_back();
      for (std::size_t j = 0; j < fields.size() / static_cast<std::size_t>(recordSize); ++j)
        frames.back().emplace_back(fields[j * recordSize + 5], fields[j * recordSize + 6]);
    }
  }
  if (data.size() < 2) throw std::runtime_error("There must be at least 2 cameras");
  for (std::size_t i = 1; i < data.size(); ++i)
    if (data[i - 1].size() != data[i].size()) throw std::runtime_error("Number of frames on all cameras must be the same");
  return data;
}

double seconds_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

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
    } else if (mode == "csvwin" && argc == 7) {  // csvwin <dir> <offset> <startFrame> <endFrame> <threads>
      DetectionsContainer dc(argv[2], std::atoi(argv[3]), 7, tdr_io::RowAllocator(), std::atoi(argv[6]),
                             std::atoi(argv[4]), std::atoi(argv[5]));
      std::printf("%d %d\n", dc.getFrameCount(), dc.getCamCount());
      for (int f = 0; f < dc.getFrameCount(); ++f)
        for (int c = 0; c < dc.getCamCount(); ++c) {
          std::printf("%d", dc.detCountForCam(c, f));
          for (int d = 0; d < dc.detCountForCam(c, f); ++d)
            std::printf(" %.17g %.17g", dc.getRecord(c, f, d).x, dc.getRecord(c, f, d).y);
          std::printf("\n");
        }
    } else if (mode == "csvseq" && argc == 3) {
      const Nested data = read_files_sequential(DetectionsContainer::getFiles(argv[2]), 0, 7);
      std::printf("%zu %zu\n", data[0].size(), data.size());
      for (std::size_t f = 0; f < data[0].size(); ++f)
        for (std::size_t c = 0; c < data.size(); ++c) {
          std::printf("%zu", data[c][f].size());
          for (const auto& p : data[c][f]) std::printf(" %.17g %.17g", p.x, p.y);
          std::printf("\n");
        }
    } else if (mode == "gencsv" && argc == 6) {
      // frame,(x,y,w,h,cx,cy,conf): float-formatted fields like the S-series recordings (std::stoi truncates),
      // a missing frame number now and then (gap filling) and an occasional line without a detection
      const int n_cam = std::atoi(argv[3]);
      const long n_frames = std::atol(argv[4]);
      std::filesystem::create_directories(argv[2]);
      for (int c = 0; c < n_cam; ++c) {
        std::mt19937_64 rng(static_cast<unsigned long long>(std::atoll(argv[5])) * 1000 + c);
        std::uniform_real_distribution<double> ux(0, 1923), uy(0, 1081), u01(0, 1);
        char name[64];
        std::snprintf(name, sizeof name, "/cam%02d.csv", c);
        FILE* f = std::fopen((std::string(argv[2]) + name).c_str(), "w");
        if (!f) throw std::runtime_error("cannot write");
        for (long fr = 0; fr < n_frames; ++fr) {
          const double r = u01(rng);
          if (r < 0.001 && fr > 0 && fr + 1 < n_frames) continue;  // frame number missing from this file
          if (r < 0.011) {
            std::fprintf(f, "%ld\n", fr);  // a frame without detections
            continue;
          }
          const double cx = ux(rng), cy = uy(rng);
          std::fprintf(f, "%ld,%.2f,%.2f,%.1f,%.1f,%.3f,%.3f,%.2f\n", fr, cx - 20, cy - 10, 40.0, 20.0, cx, cy, 0.5 + 0.5 * u01(rng));
        }
        std::fclose(f);
      }
      std::printf("ok\n");
    } else if (mode == "parsebench" && argc == 5) {
      const auto files = DetectionsContainer::getFiles(argv[2]);
      const int threads = std::atoi(argv[3]), reps = std::atoi(argv[4]);
      double best_par = 1e30, best_seq = 1e30;
      std::size_t bytes = 0;
      for (int r = 0; r < reps; ++r) {
        const double t0 = seconds_now();
        DetectionsContainer dc(argv[2], 0, 7, tdr_io::RowAllocator(), threads);
        best_par = std::min(best_par, seconds_now() - t0);
        bytes = dc.bytesParsed();
      }
      Nested ref;
      for (int r = 0; r < (reps > 2 ? 2 : reps); ++r) {
        const double t0 = seconds_now();
        ref = read_files_sequential(files, 0, 7);
        best_seq = std::min(best_seq, seconds_now() - t0);
      }
      DetectionsContainer dc(argv[2], 0, 7, tdr_io::RowAllocator(), threads);
      bool equal = static_cast<std::size_t>(dc.getCamCount()) == ref.size() &&
                   static_cast<std::size_t>(dc.getFrameCount()) == ref[0].size();
      const double* rows = dc.isOneDrone() ? dc.flatRows() : nullptr;
      for (std::size_t c = 0; equal && c < ref.size(); ++c)
        for (std::size_t f = 0; equal && f < ref[c].size(); ++f) {
          equal = static_cast<std::size_t>(dc.detCountForCam(static_cast<int>(c), static_cast<int>(f))) == ref[c][f].size();
          for (std::size_t k = 0; equal && k < ref[c][f].size(); ++k) {
            const cv::Point2d p = dc.getRecord(static_cast<int>(c), static_cast<int>(f), static_cast<int>(k));
            equal = p.x == ref[c][f][k].x && p.y == ref[c][f][k].y;
          }
          if (equal && rows) {
            const double* r = rows + 2 * (c * ref[c].size() + f);
            equal = ref[c][f].empty() ? (r[0] == -1 && r[1] == -1) : (r[0] == ref[c][f][0].x && r[1] == ref[c][f][0].y);
          }
        }
      std::printf("{\"bytes\": %zu, \"frames\": %d, \"cameras\": %d, \"threads\": %d, \"parallel_s\": %.6f, "
                  "\"parallel_MBps\": %.1f, \"sequential_s\": %.6f, \"sequential_MBps\": %.1f, \"equal\": %s, "
                  "\"one_drone\": %s}\n",
                  bytes, dc.getFrameCount(), dc.getCamCount(), threads, best_par, bytes / best_par / 1e6, best_seq,
                  bytes / best_seq / 1e6, equal ? "true" : "false", dc.isOneDrone() ? "true" : "false");
    } else if (mode == "labels" && argc == 6) {
      std::vector<Path> paths;
      readInputCSV(argv[2], paths, std::atoi(argv[3]), 0, 0, static_cast<std::size_t>(std::atoi(argv[4])),
                   static_cast<std::size_t>(std::atoi(argv[5])));
      for (const auto& p : paths) {
        std::printf("path %zu\n", p.size());
        for (const auto& q : p) std::printf("%.17g %.17g %.17g\n", q.x, q.y, q.z);
      }
    } else if (mode == "evaluate" && argc == 7) {
      std::vector<Path> labels;
      readInputCSV(argv[3], labels, std::atoi(argv[4]), 0, 0, static_cast<std::size_t>(std::atoi(argv[5])),
                   static_cast<std::size_t>(std::atoi(argv[6])));
      Path pred = readOutputFile(argv[2]);
      const PathEvaluation e = evaluatePath(labels, pred);
      std::printf("%d %zu %.17g %.17g %.17g %.17g\n", e.label, pred.size(), e.average, e.median, e.std, e.qstd);
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
  } catch (const std::invalid_argument& e) {
    std::printf("EXCEPTION invalid_argument %s\n", e.what());
  } catch (const std::out_of_range& e) {
    std::printf("EXCEPTION out_of_range %s\n", e.what());
  }
  return 0;
}
