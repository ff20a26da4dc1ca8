// DetectionsContainer.h — host side of the reference's 2D-detection store (src/DetectionsContainer.{h,cpp})
// re-designed for the engine ("next" row 3 of SURVEY.md §8f): same constructors, accessors and exceptions, but
//   * storage is flat: per camera one CSR track (count[frame], first[frame], points[]) instead of
//     vector<vector<vector<Point2d>>> — no per-frame allocations, and exactly the det_count / dets arrays the
//     batched scoring entry points take (DroneClassifier.h);
//   * the CSV files are parsed by a team of threads straight out of the page cache (DetectionsParser.h);
//   * a one-drone recording additionally keeps ONE flat block rows[camera][frame] of (x, y) with (-1,-1) for
//     "no detection" — byte for byte the rows getDataForTriangulation() returns — in pinned host memory when a
//     pinned allocator is supplied (tdr_host_alloc), plus the same rows as int16 pairs.  flatRows() /
//     flatRowsI16() hand them to Triangulator::triangulatePoints(const DetectionsContainer&) without ever
//     building the nested vectors.
// getDataForTriangulation() itself is kept for callers of the reference API.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <filesystem>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "DetectionsParser.h"
#include "tdr_cv_types.h"

typedef std::vector<cv::Point2d> Detections;

class DetectionsContainer {
 public:
  int n_frames = 0, n_cameras = 0;
  // optional map tmp-detection-index -> original index + 1 (0 = no detection), per camera per frame
  std::vector<std::vector<std::map<int, int>>> offsetVector;

  // CSV files under `path`, sorted by full path; anything containing ".csv" but not ".avi"
  // (src/DetectionsContainer.cpp:78-92)
  static std::vector<std::string> getFiles(const char* path) {
    std::vector<std::string> files;
    for (const auto& entry : std::filesystem::recursive_directory_iterator(path)) {
      const std::string p = entry.path().u8string();
      if (p.find(".csv") != std::string::npos && p.find(".avi") == std::string::npos) files.push_back(p);
    }
    std::sort(files.begin(), files.end());
    return files;
  }

  // One line per frame: frame,(x,y,w,h,cx,cy,conf) x k — semantics of src/DetectionsContainer.cpp:19-76, see
  // DetectionsParser.h.  `threads` <= 0: one per core, at most 16.
  void readFiles(const std::vector<std::string>& files, int offset, int recordSize, int startFrame, int endFrame,
                 int threads = 0) {
    release_rows();
    tdr_io::ParsedDetections p =
        tdr_io::parse_detection_files(files, offset, recordSize, startFrame, endFrame, threads, allocator_);
    cams_ = std::move(p.cams);
    n_cameras = p.n_cameras;
    n_frames = static_cast<int>(p.n_frames);
    rows_ = p.rows;
    rows16_ = p.rows16;
    one_drone_ = p.one_drone;
    fits_i16_ = p.fits_i16;
    bytes_parsed_ = p.bytes_parsed;
  }

  DetectionsContainer(const char* path, int offset, int recordSize, int startFrame = 0, int endFrame = 0) {
    readFiles(getFiles(path), offset, recordSize, startFrame, endFrame);
  }
  // the same with the memory of the flat rows taken from `allocator` (pinned host memory: tdr_host_alloc) and an
  // explicit thread count
  DetectionsContainer(const char* path, int offset, int recordSize, const tdr_io::RowAllocator& allocator,
                      int threads = 0, int startFrame = 0, int endFrame = 0)
      : allocator_(allocator) {
    readFiles(getFiles(path), offset, recordSize, startFrame, endFrame, threads);
  }

  explicit DetectionsContainer(int camCount, bool useOffset = false) : n_cameras(camCount) {
    cams_.resize(static_cast<std::size_t>(camCount));
    if (useOffset) offsetVector.resize(static_cast<std::size_t>(camCount));
  }

  ~DetectionsContainer() { release_rows(); }
  DetectionsContainer(const DetectionsContainer&) = delete;
  DetectionsContainer& operator=(const DetectionsContainer&) = delete;
  DetectionsContainer(DetectionsContainer&& o) noexcept { *this = std::move(o); }
  DetectionsContainer& operator=(DetectionsContainer&& o) noexcept {
    if (this != &o) {
      release_rows();
      n_frames = o.n_frames;
      n_cameras = o.n_cameras;
      offsetVector = std::move(o.offsetVector);
      cams_ = std::move(o.cams_);
      allocator_ = o.allocator_;
      rows_ = o.rows_;
      rows16_ = o.rows16_;
      one_drone_ = o.one_drone_;
      fits_i16_ = o.fits_i16_;
      bytes_parsed_ = o.bytes_parsed_;
      o.rows_ = nullptr;
      o.rows16_ = nullptr;
    }
    return *this;
  }

  std::vector<Detections> getFrame(int i) const {
    std::vector<Detections> frame(static_cast<std::size_t>(n_cameras));
    for (int c = 0; c < n_cameras; ++c)
      for (int d = 0; d < detCountForCam(c, i); ++d) frame[static_cast<std::size_t>(c)].push_back(getRecord(c, i, d));
    return frame;
  }
  int getFrameCount() const { return n_frames; }
  int getCamCount() const { return n_cameras; }

  std::vector<int> getDetectionsCount(int frame) const {  // +1: "no detection" (:107-114)
    std::vector<int> n(static_cast<std::size_t>(n_cameras));
    for (int i = 0; i < n_cameras; ++i) n[static_cast<std::size_t>(i)] = detCountForCam(i, frame) + 1;
    return n;
  }
  int detCountForCam(int cam, int frame) const {
    return cams_[static_cast<std::size_t>(cam)].count[static_cast<std::size_t>(frame)];
  }
  cv::Point2d getRecord(int camera, int frame, int detection) const {
    const tdr_io::CameraTrack& t = cams_[static_cast<std::size_t>(camera)];
    const double* p = t.pts.data() + 2 * (t.first[static_cast<std::size_t>(frame)] + detection);
    return cv::Point2d(p[0], p[1]);
  }

  void addEmptyFrame() {
    release_rows();
    for (int i = 0; i < n_cameras; ++i) {
      tdr_io::CameraTrack& t = cams_[static_cast<std::size_t>(i)];
      t.count.push_back(0);
      t.first.push_back(static_cast<int64_t>(t.pts.size() / 2));
      if (!offsetVector.empty()) {
        offsetVector[static_cast<std::size_t>(i)].emplace_back();
        offsetVector[static_cast<std::size_t>(i)].back()[0] = 0;
      }
    }
    n_frames = static_cast<int>(cams_.empty() ? 0 : cams_[0].count.size());
  }

  // appends to the camera's LAST frame, like the reference (data[cam].back().push_back)
  void addDetectionToCamera(cv::Point2d det, int cam, int originalIndex = -1) {
    release_rows();
    tdr_io::CameraTrack& t = cams_[static_cast<std::size_t>(cam)];
    t.pts.push_back(det.x);
    t.pts.push_back(det.y);
    const int k = ++t.count.back();
    if (k > 1) one_drone_ = false;
    if (originalIndex > -1 && !offsetVector.empty())
      offsetVector[static_cast<std::size_t>(cam)].back()[k] = originalIndex + 1;
  }

  // [camera][frame] with the (-1,-1) sentinel; more than one detection per frame is an error (:145-171)
  std::vector<std::vector<cv::Point2d>> getDataForTriangulation() {
    std::vector<std::vector<cv::Point2d>> result(static_cast<std::size_t>(n_cameras));
    for (auto& r : result) r.reserve(static_cast<std::size_t>(n_frames));
    for (int frame = 0; frame < n_frames; ++frame)
      for (int cam = 0; cam < n_cameras; ++cam) {
        const int size = detCountForCam(cam, frame);
        if (size > 1) throw std::runtime_error(kOneDroneOnly);
        result[static_cast<std::size_t>(cam)].push_back(size == 1 ? getRecord(cam, frame, 0) : cv::Point2d(-1, -1));
      }
    return result;
  }

  std::vector<int> getOriginalCombination(const std::vector<int>& combination, int frame) const {
    if (offsetVector.empty()) return combination;
    std::vector<int> original;
    for (std::size_t i = 0; i < combination.size(); ++i)
      original.push_back(offsetVector[i][static_cast<std::size_t>(frame)].at(combination[i]));
    return original;
  }

  // ---- flat views (engine side) ------------------------------------------------------------------------
  // rows[camera][frame] x (x, y): the memory of getDataForTriangulation()'s rows laid end to end.  Throws the
  // reference's one-drone exception when some frame holds more than one detection.
  const double* flatRows() {
    if (!rows_) build_rows();
    return rows_;
  }
  // the same rows as int16 pairs, or nullptr when a coordinate does not fit
  const int16_t* flatRowsI16() {
    if (!rows_) build_rows();
    return fits_i16_ ? rows16_ : nullptr;
  }
  bool isOneDrone() const { return one_drone_; }
  std::size_t bytesParsed() const { return bytes_parsed_; }
  const tdr_io::CameraTrack& track(int cam) const { return cams_[static_cast<std::size_t>(cam)]; }

 private:
  static constexpr const char* kOneDroneOnly =
      "Function 'getDataForTriangulation' can be used only for one drone. Each frame can have max one detection";

  void build_rows() {
    if (!one_drone_) throw std::runtime_error(kOneDroneOnly);
    const std::size_t n = static_cast<std::size_t>(n_cameras) * static_cast<std::size_t>(n_frames);
    rows_ = static_cast<double*>(allocator_.alloc(n * 16));
    rows16_ = static_cast<int16_t*>(allocator_.alloc(n * 4));
    if (!rows_ || !rows16_) {  // the destructor frees what is set: leave no half-built pair behind
      if (rows_) allocator_.free(rows_);
      if (rows16_) allocator_.free(rows16_);
      rows_ = nullptr;
      rows16_ = nullptr;
      throw std::bad_alloc();
    }
    fits_i16_ = true;
    for (int c = 0; c < n_cameras; ++c)
      for (int f = 0; f < n_frames; ++f) {
        double x = -1, y = -1;
        if (detCountForCam(c, f) == 1) {
          const cv::Point2d p = getRecord(c, f, 0);
          x = p.x;
          y = p.y;
        }
        const std::size_t i = 2 * (static_cast<std::size_t>(c) * static_cast<std::size_t>(n_frames) + static_cast<std::size_t>(f));
        rows_[i] = x;
        rows_[i + 1] = y;
        const bool in16 = x >= -32768 && x <= 32767 && y >= -32768 && y <= 32767 && x == static_cast<double>(static_cast<int>(x)) &&
                          y == static_cast<double>(static_cast<int>(y));
        fits_i16_ = fits_i16_ && in16;
        rows16_[i] = in16 ? static_cast<int16_t>(x) : static_cast<int16_t>(-1);
        rows16_[i + 1] = in16 ? static_cast<int16_t>(y) : static_cast<int16_t>(-1);
      }
  }
  void release_rows() {
    if (rows_) allocator_.free(rows_);
    if (rows16_) allocator_.free(rows16_);
    rows_ = nullptr;
    rows16_ = nullptr;
  }

  std::vector<tdr_io::CameraTrack> cams_;
  tdr_io::RowAllocator allocator_;
  double* rows_ = nullptr;
  int16_t* rows16_ = nullptr;
  bool one_drone_ = true, fits_i16_ = true;
  std::size_t bytes_parsed_ = 0;
};
