// DetectionsContainer.h — host mirror of the reference's 2D-detection store
// (src/DetectionsContainer.{h,cpp}): data[camera][frame][detection] of cv::Point2d, the CSV reader
// ("next" row 3 of SURVEY.md §8f) and the accessors Triangulator / DroneClassifier use.
// Plain host code: it only moves bytes into the layout the engine consumes.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "tdr_cv_types.h"

typedef std::vector<cv::Point2d> Detections;
typedef std::vector<Detections> Cameras;  // (sic) one entry per frame, as in the reference

class DetectionsContainer {
 public:
  int n_frames = 0, n_cameras = 0;
  std::vector<Cameras> data;  // [camera][frame][detection]
  // optional map tmp-detection-index -> original index + 1 (0 = no detection), per camera per frame
  std::vector<std::vector<std::map<int, int>>> offsetVector;

  // CSV files under `path`, sorted by full path; anything containing ".csv" but not ".avi"
  // (src/DetectionsContainer.cpp:78-92)
  std::vector<std::string> getFiles(const char* path) {
    std::vector<std::string> files;
    for (const auto& entry : std::filesystem::recursive_directory_iterator(path)) {
      const std::string p = entry.path().u8string();
      if (p.find(".csv") != std::string::npos && p.find(".avi") == std::string::npos) files.push_back(p);
    }
    std::sort(files.begin(), files.end());
    return files;
  }

  // One line per frame: frame,(x,y,w,h,cx,cy,conf) x k.  Every field goes through std::stoi (floats are
  // truncated); the detection is the bbox centre at columns j*recordSize+5, +6; frame numbers missing
  // from a file become empty frames; [startFrame, endFrame] filtering as in the reference
  // (src/DetectionsContainer.cpp:19-76).
  void readFiles(const std::vector<std::string>& files, int offset, int recordSize, int startFrame,
                 int endFrame) {
    for (const auto& name : files) {
      std::ifstream file(name);
      data.emplace_back();
      Cameras& frames = data.back();
      std::string line, token;
      int n_line = 0, frame = -1;
      while (std::getline(file, line)) {
        if (offset > n_line++) continue;
        std::vector<int> fields;
        std::istringstream iss(line);
        while (std::getline(iss, token, ','))
          if (!token.empty()) fields.push_back(std::stoi(token));
        if (fields.empty()) continue;  // the reference would index fields[0] of an empty vector (UB)
        if ((fields[0] <= startFrame || fields[0] > endFrame) && startFrame != endFrame) {
          frame = fields[0];
          continue;
        }
        for (int i = 0; i < fields[0] - frame - 1; ++i) frames.emplace_back();
        frame = fields[0];
        if ((fields.size() - 1) % static_cast<std::size_t>(recordSize) != 0)
          throw std::runtime_error("Invalid CSV file!");
        frames.emplace_back();
        for (std::size_t j = 0; j < fields.size() / static_cast<std::size_t>(recordSize); ++j)
          frames.back().emplace_back(fields[j * recordSize + 5], fields[j * recordSize + 6]);
      }
    }
    if (data.size() < 2) throw std::runtime_error("There must be at least 2 cameras");
    for (std::size_t i = 1; i < data.size(); ++i)
      if (data[i - 1].size() != data[i].size())
        throw std::runtime_error("Number of frames on all cameras must be the same");
    n_frames = static_cast<int>(data[0].size());
    n_cameras = static_cast<int>(data.size());
  }

  DetectionsContainer(const char* path, int offset, int recordSize, int startFrame = 0, int endFrame = 0) {
    readFiles(getFiles(path), offset, recordSize, startFrame, endFrame);
  }

  explicit DetectionsContainer(int camCount, bool useOffset = false) : n_cameras(camCount) {
    data.resize(camCount);
    if (useOffset) offsetVector.resize(camCount);
  }

  std::vector<Detections> getFrame(int i) const {
    std::vector<Detections> frame;
    for (const auto& cam : data) frame.push_back(cam[i]);
    return frame;
  }
  int getFrameCount() const { return n_frames; }
  int getCamCount() const { return n_cameras; }

  std::vector<int> getDetectionsCount(int frame) const {  // +1: "no detection" (:107-114)
    std::vector<int> n(n_cameras);
    for (int i = 0; i < n_cameras; ++i) n[i] = static_cast<int>(data[i][frame].size()) + 1;
    return n;
  }
  int detCountForCam(int cam, int frame) const { return static_cast<int>(data[cam][frame].size()); }
  cv::Point2d getRecord(int camera, int frame, int detection) const { return data[camera][frame][detection]; }

  void addEmptyFrame() {
    for (int i = 0; i < n_cameras; ++i) {
      data[i].emplace_back();
      if (!offsetVector.empty()) {
        offsetVector[i].emplace_back();
        offsetVector[i].back()[0] = 0;
      }
    }
    n_frames = static_cast<int>(data.empty() ? 0 : data[0].size());
  }

  void addDetectionToCamera(cv::Point2d det, int cam, int originalIndex = -1) {
    data[cam].back().push_back(det);
    if (originalIndex > -1 && !offsetVector.empty())
      offsetVector[cam].back()[static_cast<int>(data[cam].back().size())] = originalIndex + 1;
  }

  // [camera][frame] with the (-1,-1) sentinel; more than one detection per frame is an error (:145-171)
  std::vector<std::vector<cv::Point2d>> getDataForTriangulation() {
    std::vector<std::vector<cv::Point2d>> result(n_cameras);
    for (int frame = 0; frame < n_frames; ++frame)
      for (int cam = 0; cam < n_cameras; ++cam) {
        const std::size_t size = data[cam][frame].size();
        if (size > 1)
          throw std::runtime_error(
              "Function 'getDataForTriangulation' can be used only for one drone. Each frame can have max "
              "one detection");
        result[cam].push_back(size == 1 ? data[cam][frame][0] : cv::Point2d(-1, -1));
      }
    return result;
  }

  std::vector<int> getOriginalCombination(const std::vector<int>& combination, int frame) const {
    if (offsetVector.empty()) return combination;
    std::vector<int> original;
    for (std::size_t i = 0; i < combination.size(); ++i)
      original.push_back(offsetVector[i][frame].at(combination[i]));
    return original;
  }
};
