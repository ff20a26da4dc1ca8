// utils.h — host I/O around the path ("next" rows 2 and 4 of SURVEY.md §8f): Vicon .xcp camera
// ingest (reference src/utils.cpp:46-107, without pugixml), ASCII PLY output (:109-131) and the
// error statistics of the evaluation tool (:133-213).  Plain host code; no solve happens here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Camera.h"
#include "tdr_cv_types.h"

typedef std::vector<cv::Point3d> Path;

namespace tdr_xcp {

// value of attribute `name` inside one tag's text (`<Tag a="1" b="2"/>`); empty string if absent
inline std::string attribute(const std::string& tag, const std::string& name) {
  std::size_t pos = 0;
  const std::string key = name + "=\"";
  while ((pos = tag.find(key, pos)) != std::string::npos) {
    // must start an attribute: preceded by whitespace
    if (pos > 0 && (tag[pos - 1] == ' ' || tag[pos - 1] == '\t' || tag[pos - 1] == '\n' || tag[pos - 1] == '\r')) {
      const std::size_t b = pos + key.size(), e = tag.find('"', b);
      return e == std::string::npos ? std::string() : tag.substr(b, e - b);
    }
    pos += key.size();
  }
  return {};
}

// text of the first `<name ...>` start tag at or after `from` inside [from, limit)
inline bool find_tag(const std::string& xml, const std::string& name, std::size_t from, std::size_t limit,
                     std::size_t& tag_begin, std::size_t& tag_end) {
  std::size_t pos = from;
  const std::string open = "<" + name;
  while ((pos = xml.find(open, pos)) != std::string::npos && pos < limit) {
    const char next = pos + open.size() < xml.size() ? xml[pos + open.size()] : '\0';
    if (next == ' ' || next == '>' || next == '/' || next == '\t' || next == '\n' || next == '\r') {
      tag_begin = pos;
      tag_end = xml.find('>', pos);
      return tag_end != std::string::npos && tag_end < limit;
    }
    pos += open.size();
  }
  return false;
}

}  // namespace tdr_xcp

// loadCamerasXML (src/utils.cpp:46-92): one tdr::Camera per <Camera> that has a <ControlFrame>
// (the 28 Vicon IR cameras of R02_D1.xcp have an empty <ControlFrames/> and are skipped, :54-57);
// width/height = 2 x PRINCIPAL_POINT read as ints (:66-71); ORIENTATION is read as "w i j k" (:80-87).
//
// viconOrder = true applies the documented fix-up (README.md:37): the shipped raw
// Vicon files store ORIENTATION as "x y z w", so the quaternion handed to the camera becomes
// (w, -x, -y, -z).  false reproduces the reference loader literally.
inline std::vector<const tdr::Camera*> loadCamerasXML(const char* path, bool viconOrder = false) {
  std::ifstream file(path, std::ios::binary);
  if (!file) throw std::runtime_error("Cannot open camera XML file");
  std::stringstream buffer;
  buffer << file.rdbuf();
  const std::string xml = buffer.str();
  std::size_t root_b = 0, root_e = 0;
  if (!tdr_xcp::find_tag(xml, "Cameras", 0, xml.size(), root_b, root_e))
    throw std::runtime_error("Cannot open camera XML file");

  std::vector<const tdr::Camera*> cameras;
  std::size_t pos = root_e;
  std::size_t cam_b = 0, cam_e = 0;
  while (tdr_xcp::find_tag(xml, "Camera", pos, xml.size(), cam_b, cam_e)) {
    std::size_t close = xml.find("</Camera>", cam_e);
    if (xml[cam_e - 1] == '/') close = cam_e;  // <Camera .../> has no children
    if (close == std::string::npos) close = xml.size();
    pos = close + 1;
    std::size_t cfs_b = 0, cfs_e = 0, cf_b = 0, cf_e = 0;
    if (!tdr_xcp::find_tag(xml, "ControlFrames", cam_e, close, cfs_b, cfs_e)) continue;
    if (xml[cfs_e - 1] == '/') continue;  // <ControlFrames/>
    const std::size_t cfs_close = std::min(xml.find("</ControlFrames>", cfs_e), close);
    if (!tdr_xcp::find_tag(xml, "ControlFrame", cfs_e, cfs_close, cf_b, cf_e)) continue;
    const std::string camTag = xml.substr(cam_b, cam_e - cam_b + 1);
    const std::string cf = xml.substr(cf_b, cf_e - cf_b + 1);

    const int id = static_cast<int>(std::strtol(tdr_xcp::attribute(camTag, "DEVICEID").c_str(), nullptr, 10));
    const double focalLength = std::strtod(tdr_xcp::attribute(cf, "FOCAL_LENGTH").c_str(), nullptr);
    int width = 0, height = 0;
    std::stringstream principal(tdr_xcp::attribute(cf, "PRINCIPAL_POINT"));
    principal >> width >> height;
    width *= 2;
    height *= 2;
    double x = 0, y = 0, z = 0;
    std::stringstream position(tdr_xcp::attribute(cf, "POSITION"));
    position >> x >> y >> z;
    double w = 0, i = 0, j = 0, k = 0;
    std::stringstream orientation(tdr_xcp::attribute(cf, "ORIENTATION"));
    orientation >> w >> i >> j >> k;
    std::array<double, 4> quat = {w, i, j, k};
    if (viconOrder) quat = {k, -w, -i, -j};  // file holds x y z w -> (w, -x, -y, -z)
    cameras.push_back(tdr::createCamera(id, static_cast<std::size_t>(width), static_cast<std::size_t>(height),
                                        focalLength, {x, y, z}, quat));
  }
  return cameras;
}

// writeOutputFile (src/utils.cpp:109-131): ASCII PLY, vertices only, default ostream formatting
// (6 significant digits).
inline void writeOutputFile(const char* path, const std::vector<cv::Point3d>& triangulatedPoints) {
  std::ofstream out(path);
  out << "ply\n"
      << "format ascii 1.0\n"
      << "element vertex " << static_cast<int>(triangulatedPoints.size()) << "\n"
      << "property float x\nproperty float y\nproperty float z\n"
      << "element face " << 0 << "\n"
      << "property list uchar int vertex_index\n"
      << "end_header\n";
  for (const auto& p : triangulatedPoints) out << p.x << " " << p.y << " " << p.z << "\n";
}

namespace tdr_stats {
// per-frame errors the reference's statistics share (:137-146): frames whose prediction is exactly
// (0,0,0) are skipped, NaN errors dropped
inline std::vector<double> errors(const Path& labelPath, const Path& predPath) {
  std::vector<double> e;
  const std::size_t size = std::min(labelPath.size(), predPath.size());
  for (std::size_t i = 0; i < size; ++i) {
    if (predPath[i].x == 0 && predPath[i].y == 0 && predPath[i].z == 0) continue;
    const double dx = predPath[i].x - labelPath[i].x, dy = predPath[i].y - labelPath[i].y,
                 dz = predPath[i].z - labelPath[i].z;
    const double err = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (!std::isnan(err)) e.push_back(err);
  }
  return e;
}
}  // namespace tdr_stats

// The reference divides by min(|label|, |pred|), not by the number of frames actually summed (:147).
inline double calculateError(Path& labelPath, Path& predPath) {
  double sum = 0;
  for (double e : tdr_stats::errors(labelPath, predPath)) sum += e;
  return sum / static_cast<double>(std::min(labelPath.size(), predPath.size()));
}

// Returns the (biased, same denominator as above) VARIANCE despite its name (:150-166).
inline double calculateStd(Path& labelPath, Path& predPath, double mean) {
  double sum = 0;
  for (double e : tdr_stats::errors(labelPath, predPath)) sum += (e - mean) * (e - mean);
  return sum / static_cast<double>(std::min(labelPath.size(), predPath.size()));
}

// (sic) quartile deviation with quartile indices taken from the UNfiltered size (:168-190)
inline double calcualteQuarterDeviation(Path& labelPath, Path& predPath) {
  std::vector<double> e = tdr_stats::errors(labelPath, predPath);
  std::sort(e.begin(), e.end());
  const std::size_t size = std::min(labelPath.size(), predPath.size());
  if (e.empty() || 3 * (size / 4) >= e.size())  // the reference indexes out of bounds here
    throw std::runtime_error("There are no errors to calculate quartiles from");
  return (e[3 * (size / 4)] - e[size / 4]) / 2;
}

// Median of the errors in FRAME order — the reference never sorts them (:192-213).
inline double calculateMedian(Path& labelPath, Path& predPath) {
  const std::vector<double> e = tdr_stats::errors(labelPath, predPath);
  if (e.empty()) throw std::runtime_error("There are no errors to calculate median from");
  if (e.size() % 2 == 1) return e[(e.size() - 1) / 2];
  return (e[e.size() / 2] + e[e.size() / 2 - 1]) / 2;
}

// ---------------------------------------------------------------------------------------------------------
// Evaluation against Vicon marker labels ("next" row 4 of SURVEY.md §8f): src/utils.cpp:215-340 and the
// comparison loop of src/test.cpp:57-88.
// ---------------------------------------------------------------------------------------------------------

// cross3d (src/utils.cpp:275-317): the crossing point of a drone's two marker arms.  The first pairing of the
// four points whose two arm lengths match `arms` within `err` decides which markers are opposite; the crossing
// point is the foot of marker (pairing[1]) on the line marker0 -> marker(pairing[0]).  Returns the five points
// [arm x a, arm x b, arm y a, arm y b, crossing] or an empty vector when no pairing matches.
inline std::vector<cv::Point3d> cross3d(std::vector<cv::Point3d> four_points,
                                        std::vector<double> arms = std::vector<double>{9, 6}, float err = 1000) {
  const int pairings[3][3] = {{1, 2, 3}, {2, 1, 3}, {3, 1, 2}};
  const int* order = nullptr;
  auto length = [](const cv::Point3d& a, const cv::Point3d& b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz);
  };
  for (const auto& j : pairings) {
    const double l1 = length(four_points[0], four_points[j[0]]), l2 = length(four_points[j[1]], four_points[j[2]]);
    if ((std::abs(l1 - arms[0]) < err && std::abs(l2 - arms[1]) < err) ||
        (std::abs(l1 - arms[1]) < err && std::abs(l2 - arms[0]) < err)) {
      order = j;
      break;
    }
  }
  if (!order) return {};
  const cv::Point3d& p0 = four_points[0];
  const cv::Point3d dc(four_points[order[0]].x - p0.x, four_points[order[0]].y - p0.y, four_points[order[0]].z - p0.z);
  const cv::Point3d dn(four_points[order[1]].x - p0.x, four_points[order[1]].y - p0.y, four_points[order[1]].z - p0.z);
  const double t = (dc.x * dn.x + dc.y * dn.y + dc.z * dn.z) / (dc.x * dc.x + dc.y * dc.y + dc.z * dc.z);
  const cv::Point3d crossing(p0.x + dc.x * t, p0.y + dc.y * t, p0.z + dc.z * t);
  return {p0, four_points[order[0]], four_points[order[1]], four_points[order[2]], crossing};
}

// convert2global (src/utils.cpp:319-344): a point given in the drone's arm frame -> world frame.  The z axis
// is the un-normalised cross product of the two unit arm directions, as in the reference.
inline cv::Point3d convert2global(std::vector<cv::Point3d> cross, cv::Point3d localPoint) {
  auto unit = [](const cv::Point3d& a, const cv::Point3d& b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, n = std::sqrt(dx * dx + dy * dy + dz * dz);
    return cv::Point3d(dx / n, dy / n, dz / n);
  };
  const cv::Point3d vx = unit(cross[1], cross[0]), vy = unit(cross[3], cross[2]);
  const cv::Point3d vz(vx.y * vy.z - vx.z * vy.y, vx.z * vy.x - vx.x * vy.z, vx.x * vy.y - vx.y * vy.x);
  return cv::Point3d(cross[4].x + localPoint.x * vx.x + localPoint.y * vy.x + localPoint.z * vz.x,
                     cross[4].y + localPoint.x * vx.y + localPoint.y * vy.y + localPoint.z * vz.y,
                     cross[4].z + localPoint.x * vx.z + localPoint.y * vy.z + localPoint.z * vz.z);
}

// readInputCSV (src/utils.cpp:215-273): Vicon label file -> one path of arm-crossing points per drone.  The
// reference hard-codes 4 drones x 5 markers (61 columns); the shipped single-drone recordings hold 4 markers
// (dataset/R02_D1/R02_D1.csv: 13 columns), hence the two trailing parameters — their defaults are the
// reference's constants, (1, 4) reads the shipped files.  Lines containing "Drone" are headers; every
// `frequency`-th data line is one frame; a line with another column count throws "Error at CSV file".
inline void readInputCSV(const char* dir, std::vector<Path>& paths, int frequency, int startFrame = 0,
                         int endFrame = 0, std::size_t nDrones = 4, std::size_t markersPerDrone = 5) {
  std::ifstream file(dir);
  std::string line, token;
  int i = 0;
  for (std::size_t k = 0; k < nDrones; ++k) paths.push_back({});
  while (std::getline(file, line)) {
    if (line.find("Drone") != std::string::npos) continue;
    if (i % frequency != 0 || ((i / frequency < startFrame || i / frequency > endFrame) && startFrame < endFrame)) {
      ++i;
      continue;
    }
    ++i;
    std::istringstream iss(line);
    std::vector<double> fields;
    while (std::getline(iss, token, ',')) fields.push_back(std::stod(token));
    if (fields.size() != 1 + nDrones * markersPerDrone * 3) throw std::runtime_error("Error at CSV file");
    for (std::size_t drone = 0; drone < nDrones; ++drone) {
      std::vector<cv::Point3d> markers;
      for (std::size_t j = 0; j < markersPerDrone; ++j) {
        const std::size_t b = markersPerDrone * drone * 3 + j * 3;
        markers.emplace_back(fields[b + 1], fields[b + 2], fields[b + 3]);
      }
      const std::vector<cv::Point3d> cross = cross3d(markers);
      if (cross.size() != 5) continue;
      paths[drone].push_back(cross[4]);
    }
  }
}

// vertices of an ASCII PLY as writeOutputFile produces it (the reference's test.cpp reads them through happly)
inline Path readOutputFile(const char* path) {
  std::ifstream in(path);
  std::string line;
  std::size_t n = 0;
  bool header_done = false;
  while (std::getline(in, line)) {
    if (line.rfind("element vertex", 0) == 0) n = static_cast<std::size_t>(std::stoull(line.substr(14)));
    if (line == "end_header") {
      header_done = true;
      break;
    }
  }
  if (!header_done) throw std::runtime_error("not a PLY file");
  Path pts;
  double x, y, z;
  while (pts.size() < n && (in >> x >> y >> z)) pts.emplace_back(x, y, z);
  return pts;
}

// The comparison of src/test.cpp:57-88 for one predicted path: the label path with the smallest average error
// wins; its statistics are reported (in the units of the inputs).
struct PathEvaluation {
  int label = -1;  // index of the winning label path
  double average = 1e+8, median = 1e+8, std = 1e+8, qstd = 1e+8;
};
inline PathEvaluation evaluatePath(std::vector<Path>& labelPaths, Path& predPath) {
  PathEvaluation best;
  for (std::size_t i = 0; i < labelPaths.size(); ++i) {
    const double error = calculateError(labelPaths[i], predPath);
    const double median = calculateMedian(labelPaths[i], predPath);
    const double sd = calculateStd(labelPaths[i], predPath, error);
    const double qstd = calcualteQuarterDeviation(labelPaths[i], predPath);
    if (error < best.average) best = PathEvaluation{static_cast<int>(i), error, median, sd, qstd};
  }
  return best;
}
