// tdr_main.cpp — the reference's `main` (src/main.cpp:15-89) on the MI355X engine:
//   tdr_main cameras.xcp data_dir [--n_drones N] [--triangulator matrix|ray]
//            [--xcp-vicon-order] [--output_dir DIR]
// Loads the Vicon calibration and the per-camera detection CSVs, associates detections to drones,
// triangulates one 3D point per drone per frame and writes results/droneK.ply.
#include <chrono>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <memory>

#include "DetectionsContainer.h"
#include "DroneClassifier.h"
#include "MatrixTriangulator.h"
#include "RayTriangulator.h"
#include "utils.h"

int main(int argc, const char** argv) {
  std::string cameras_path, data_path, triangulatorType = "matrix", outputDir = "./results/";
  int n_drones = 1;
  bool viconOrder = false;
  int positional = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&]() -> std::string {
      if (i + 1 >= argc) throw std::runtime_error("missing value for " + a);
      return argv[++i];
    };
    try {
      if (a == "--n_drones") n_drones = std::stoi(value());
      else if (a == "--triangulator") triangulatorType = value();
      else if (a == "--output_dir") outputDir = value();
      else if (a == "--xcp-vicon-order") viconOrder = true;
      else if (positional == 0) { cameras_path = a; ++positional; }
      else if (positional == 1) { data_path = a; ++positional; }
      else throw std::runtime_error("unexpected argument " + a);
    } catch (const std::exception& e) {
      std::cerr << e.what() << std::endl;
      return 1;
    }
  }
  if (positional != 2) {
    std::cerr << "usage: tdr_main cameras_path data_path [--n_drones N] [--triangulator matrix|ray] "
                 "[--xcp-vicon-order] [--output_dir DIR]" << std::endl;
    return 1;
  }
  if (!outputDir.empty() && outputDir.back() != '/') outputDir += '/';
  if (std::filesystem::exists(outputDir)) std::filesystem::remove_all(outputDir);
  std::filesystem::create_directories(outputDir);

  std::vector<const tdr::Camera*> cameras = loadCamerasXML(cameras_path.c_str(), viconOrder);
  std::unique_ptr<Triangulator> triangulator;
  if (triangulatorType == "matrix")
    triangulator.reset(new MatrixTriangulator(cameras));
  else if (triangulatorType == "ray")
    triangulator.reset(new RayTriangulator(cameras));
  else
    throw std::runtime_error("Invalid --triangulator argument. Allowed options are 'matrix' and 'ray'");

  DroneClassifier classifier(triangulator.get(), static_cast<std::size_t>(n_drones));
  DetectionsContainer container(data_path.c_str(), 0, 7);

  const auto start = std::chrono::high_resolution_clock::now();
  std::vector<std::vector<cv::Point3d>> triangulatedPoints;
  classifier.classifyDrones(container, triangulatedPoints);
  const auto stop = std::chrono::high_resolution_clock::now();
  std::cout << "Execution time: "
            << std::chrono::duration_cast<std::chrono::microseconds>(stop - start).count() * 1e-6 << "s"
            << std::endl;

  for (int i = 1; i <= n_drones; ++i)
    writeOutputFile((outputDir + "drone" + std::to_string(i) + ".ply").c_str(), triangulatedPoints[i - 1]);
  triangulator.reset();
  for (const auto* cam : cameras) delete cam;
  return 0;
}
