// DroneClassifier.h — the reference's multi-drone association (src/DroneClassifier.{h,cpp}) over
// the MI355X engine, same public surface:
//     DroneClassifier(Triangulator*, size_t n_drones);
//     void classifyDrones(const DetectionsContainer&, std::vector<std::vector<cv::Point3d>>&);
//
// What changed is where the work happens (SURVEY.md §8f "next" row 1).  The reference walks, per
// frame and per path, a prefix tree of detection combinations and calls triangulatePoint once per
// node (src/DroneClassifier.cpp:156-198).  Whether a full combination survives that walk depends
// only on its own detections (all of its camera prefixes with >= 2 detections must have
// err <= error_), so ALL combinations of ALL frames are scored up front — matrix triangulator: one
// launch of the leaf kernel (tdr_matrix_score_combinations); ray triangulator: a breadth-first sweep
// with one LM batch per camera depth (tdr_ray_score_combinations) — and the per-frame logic then only
// filters and orders those survivors:
//   - fillCombinationQueue(container, frame)            = the frame's survivors,
//   - fillCombinationQueue(tmpContainer near lastPos)   = the survivors whose detections all lie
//     within MAX_STEP of the path's last position (distances: one batched getDistFromRay launch per
//     frame, tdr_dist_from_ray_batch).
// The sequential part (priority order, uniqueness, path assignment) is plain host control flow.
// If the container's camera count differs from the triangulator's, the generic walk with one
// triangulatePoint launch per node is used (correct, slow).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

#include "DetectionsContainer.h"
#include "Triangulator.h"

#define MAX_ERROR_MATRIX 1e+5  // src/DroneClassifier.h:11
#define MAX_ERROR_RAY 120      // :12
#define MAX_STEP 200           // :13-15
#define MIN_CAMERAS 2          // :16
#define PATH_TAIL 3            // :17

class DroneClassifier {
 public:
  struct Combination {
    std::vector<int> combination_;
    cv::Point3d point;
    double error = 0;

    int zeros() const { return static_cast<int>(std::count(combination_.begin(), combination_.end(), 0)); }
    // priority order of the reference's operator< / operator> (src/DroneClassifier.cpp:12-30):
    // fewer "no detection" entries first, then the smaller error
    bool before(const Combination& c) const {
      const int a = zeros(), b = c.zeros();
      return a != b ? a < b : error < c.error;
    }
    bool isCombinationUnique(const std::vector<Combination>& others) const {  // :32-41
      for (const auto& o : others)
        for (std::size_t i = 0; i < combination_.size(); ++i)
          if (combination_[i] == o.combination_[i] && combination_[i] != 0) return false;
      return true;
    }
  };

  // isCombinationUnique against a growing set, as one bit per (camera, detection): a combination clashes with the
  // set iff one of its chosen detections is already taken on the same camera
  struct TakenMask {
    std::vector<std::uint64_t> bits;  // [n_cam][2]: digits are < 128
    int n_cam = 0;
    void reset(int cams) {
      n_cam = cams;
      bits.assign(static_cast<std::size_t>(cams) * 2, 0);
    }
    template <typename D>
    void add(const D* digits) {
      for (int c = 0; c < n_cam; ++c)
        if (digits[c] != 0) bits[2 * c + (digits[c] >> 6)] |= std::uint64_t(1) << (digits[c] & 63);
    }
    template <typename D>
    bool clashes(const D* digits) const {
      std::uint64_t hit = 0;
      for (int c = 0; c < n_cam; ++c)
        hit |= digits[c] != 0 ? bits[2 * c + (digits[c] >> 6)] >> (digits[c] & 63) & 1u : 0u;
      return hit != 0;
    }
  };

  DroneClassifier(Triangulator* triangulator, std::size_t n_drones)
      : triangulator_(triangulator), n_drones_(n_drones) {
    if (triangulator->getType() == "matrix")
      error_ = MAX_ERROR_MATRIX;
    else if (triangulator->getType() == "ray")
      error_ = MAX_ERROR_RAY;
    else  // the reference leaves error_ uninitialised here (src/DroneClassifier.cpp:5-9)
      throw std::runtime_error("DroneClassifier: unknown triangulator type " + triangulator->getType());
  }

  // No destructor work and no device memory of its own: the reference's main deletes the triangulator BEFORE
  // the (stack) classifier goes out of scope (src/main.cpp:85), so nothing here may touch triangulator_ after
  // classifyDrones returns.

  void classifyDrones(const DetectionsContainer& container,
                      std::vector<std::vector<cv::Point3d>>& triangulatedPoints) {
    const int n_frames = container.getFrameCount();
    const bool trace = std::getenv("TDR_CLASSIFIER_TRACE") != nullptr;  // where the time goes, on stderr
    const auto t_start = std::chrono::steady_clock::now();
    dist_batch_s_ = near_s_ = scan_s_ = pick_s_ = 0;
    pick_frames_ = 0;
    prepare(container);
    const auto t_prepared = std::chrono::steady_clock::now();
    std::vector<std::vector<int>> emptyFrames(n_drones_);
    triangulatedPoints.assign(n_drones_, {});

    for (int frame = 0; frame < n_frames; ++frame) {
      std::vector<int> processedPaths;
      std::vector<Combination> usedCombinations;
      // paths that already have a (non-zero) last position: look near it first (:116-131)
      std::vector<int> active;
      for (std::size_t p = 0; p < triangulatedPoints.size(); ++p) {
        const auto& path = triangulatedPoints[p];
        if (!path.empty() && !(path.back().x == 0 && path.back().y == 0 && path.back().z == 0))
          active.push_back(static_cast<int>(p));
      }
      const auto t_near = std::chrono::steady_clock::now();
      std::vector<std::vector<char>> near = nearLastPositions(container, frame, active, triangulatedPoints);
      near_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_near).count();
      const auto t_scan = std::chrono::steady_clock::now();
      for (std::size_t a = 0; a < active.size(); ++a) {
        const int p = active[a];
        Combination best = triangulateWithLastPos(triangulatedPoints[p].back(), container, near[a],
                                                  usedCombinations, frame);
        if (!best.combination_.empty()) {
          processedPaths.push_back(p);
          usedCombinations.push_back(best);
          triangulatedPoints[p].push_back(best.point);
        }
      }
      const auto t_scanned = std::chrono::steady_clock::now();
      scan_s_ += std::chrono::duration<double>(t_scanned - t_scan).count();
      if (processedPaths.size() == triangulatedPoints.size()) continue;
      std::vector<Combination> finalCombinations = pickBestCombinations(container, frame, usedCombinations);
      classifyPaths(finalCombinations, triangulatedPoints, processedPaths, emptyFrames, frame);
      pick_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scanned).count();
      ++pick_frames_;
    }
    if (trace) {
      const auto t_end = std::chrono::steady_clock::now();
      std::fprintf(stderr, "classifyDrones: %d frames, %zu surviving combinations; scoring %.1f ms (of which decoding + "
                   "ordering the survivors %.1f ms), frame loop %.1f ms (of which distance batches %.1f ms, their set-up "
                   "%.1f ms, continuing the paths %.1f ms, new assignments %.1f ms in %d frames)\n", n_frames, survivors_.size(),
                   std::chrono::duration<double, std::milli>(t_prepared - t_start).count(), load_frame_s_ * 1e3,
                   std::chrono::duration<double, std::milli>(t_end - t_prepared).count(), dist_batch_s_ * 1e3,
                   (near_s_ - dist_batch_s_) * 1e3, scan_s_ * 1e3, pick_s_ * 1e3, pick_frames_);
    }
    for (std::size_t i = 0; i < emptyFrames.size(); ++i)  // keep the frame count (:147-153)
      for (int f : emptyFrames[i])
        triangulatedPoints[i].insert(triangulatedPoints[i].begin() + f, cv::Point3d(0, 0, 0));
  }

 private:
  Triangulator* triangulator_;
  std::size_t n_drones_;
  double error_ = 0;

  // ---- batched state (matrix triangulator)
  bool batched_ = false;
  int n_cam_ = 0;
  std::vector<std::int32_t> det_count_;     // [frame][cam]
  std::vector<std::int64_t> det_offset_;    // [frame][cam] into the flat detection list
  std::vector<double> dets_;                // [total][2]
  // the scoring call's output: up to 100+ MB that the call overwrites — no value-initialisation
  struct CandidateArray {
    std::unique_ptr<tdr_candidate[]> p;
    std::size_t n = 0, cap = 0;
    void reserve_uninitialised(std::size_t c) {  // contents are lost
      if (c > cap) {
        p.reset(new tdr_candidate[c]);
        cap = c;
      }
      n = 0;
    }
    void set_size(std::size_t k) { n = k; }
    void clear() { n = 0; }
    std::size_t size() const { return n; }
    tdr_candidate* data() { return p.get(); }
    const tdr_candidate* data() const { return p.get(); }
    tdr_candidate& operator[](std::size_t i) { return p[i]; }
    const tdr_candidate& operator[](std::size_t i) const { return p[i]; }
  };
  CandidateArray survivors_;                // sorted by (frame, leaf)
  double dist_batch_s_ = 0;                 // time spent in tdr_dist_from_ray_batch (TDR_CLASSIFIER_TRACE)
  double load_frame_s_ = 0, near_s_ = 0;    // ... in orderSurvivors / nearLastPositions
  double scan_s_ = 0, pick_s_ = 0;          // ... continuing paths near their last position / assigning the rest
  int pick_frames_ = 0;
  // the current frame's survivors, decoded once: digits [n][n_cam], and their indices in the reference's priority
  // order (fewer "no detection" entries first, then the smaller error; ties keep the survivors' own order, which
  // is what a stable sort of any filtered subset of them gives as well)
  std::vector<std::int8_t> all_digits_;     // [survivor][n_cam] (a camera holds far fewer than 127 detections)
  std::vector<std::int32_t> all_order_;     // per frame: survivor indices (relative to the frame) in priority order
  std::vector<char> usable_;                // scratch of triangulateWithLastPos
  TakenMask taken_;                         // scratch of the two scans
  const std::int8_t* frame_digits_ = nullptr;
  const std::int32_t* frame_order_ = nullptr;
  std::size_t frame_n_ = 0;
  std::vector<std::int64_t> frame_begin_;   // [n_frames + 1] into survivors_

  static void check(int rc) {
    if (rc != TDR_OK) throw std::runtime_error(std::string(tdr_strerror(rc)) + ": " + tdr_last_error());
  }

  // flatten the container and score every combination of every frame in one launch
  void prepare(const DetectionsContainer& container) {
    n_cam_ = container.getCamCount();
    const int n_frames = container.getFrameCount();
    const bool ray = triangulator_->getType() == "ray";
    batched_ = (ray || triangulator_->getType() == "matrix") &&
               n_cam_ == static_cast<int>(triangulator_->getCameras().size());
    det_count_.assign(static_cast<std::size_t>(n_frames) * n_cam_, 0);
    det_offset_.assign(static_cast<std::size_t>(n_frames) * n_cam_, 0);
    dets_.clear();
    for (int f = 0; f < n_frames; ++f)
      for (int c = 0; c < n_cam_; ++c) {
        const int k = container.detCountForCam(c, f);
        if (k > 126) batched_ = false;  // the per-frame tables below hold a camera's choice in an int8
        det_count_[static_cast<std::size_t>(f) * n_cam_ + c] = k;
        det_offset_[static_cast<std::size_t>(f) * n_cam_ + c] = static_cast<std::int64_t>(dets_.size() / 2);
        for (int d = 0; d < k; ++d) {
          const cv::Point2d p = container.getRecord(c, f, d);
          dets_.push_back(p.x);
          dets_.push_back(p.y);
        }
      }
    survivors_.clear();
    frame_begin_.assign(static_cast<std::size_t>(n_frames) + 1, 0);
    if (!batched_ || n_frames == 0) return;
    const bool trace = std::getenv("TDR_CLASSIFIER_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_flat = now();
    std::int64_t cap = std::max<std::int64_t>(1024, 8 * static_cast<std::int64_t>(n_frames)), n = 0;
    int calls = 0;
    for (;;) {
      ++calls;
      survivors_.reserve_uninitialised(static_cast<std::size_t>(cap));
      // matrix: brute-force leaf kernel; ray: breadth-first sweep, one LM batch per camera depth
      int rc = ray ? tdr_ray_score_combinations(triangulator_->context(), n_frames, det_count_.data(),
                                                dets_.data(), error_, MIN_CAMERAS, survivors_.data(), cap, &n)
                   : tdr_matrix_score_combinations(triangulator_->context(), n_frames, det_count_.data(),
                                                   dets_.data(), error_, MIN_CAMERAS, survivors_.data(), cap, &n);
      if (rc == TDR_E_ARG && n > cap) {  // buffer too small: the needed size came back
        cap = n;
        survivors_.reserve_uninitialised(static_cast<std::size_t>(cap));
        // a sweep keeps its complete result on the device: fetch it instead of sweeping again
        std::int64_t m = 0;
        if (tdr_fetch_scored_combinations(triangulator_->context(), survivors_.data(), cap, &m) == TDR_OK && m == n) break;
        continue;
      }
      check(rc);
      break;
    }
    const auto t_scored = now();
    if (trace) std::fprintf(stderr, "  prepare: scoring calls %d, %.1f ms (capacity %lld, survivors %lld)\n", calls, ms(t_flat, t_scored), (long long)cap, (long long)n);
    survivors_.set_size(static_cast<std::size_t>(n));
    std::size_t s = 0;
    for (int f = 0; f <= n_frames; ++f) {
      while (s < survivors_.size() && survivors_[s].frame < f) ++s;
      frame_begin_[f] = static_cast<std::int64_t>(s);
    }
    const auto t_o = std::chrono::steady_clock::now();
    orderSurvivors();
    load_frame_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_o).count();
  }

  Combination decode(const tdr_candidate& c) const {
    Combination out;
    out.combination_.resize(n_cam_);
    check(tdr_decode_combination(n_cam_, &det_count_[static_cast<std::size_t>(c.frame) * n_cam_], c.leaf,
                                 out.combination_.data()));
    out.point = cv::Point3d(c.x, c.y, c.z);
    out.error = c.err;
    return out;
  }

  // ---- generic walk: the reference's Iterator (src/DroneClassifier.cpp:43-85) + one
  // triangulatePoint per node; `allowed` (optional) restricts the detections like the tmpContainer
  // of triangulateWithLastPos does.
  std::vector<Combination> walkTree(const DetectionsContainer& container, int frame,
                                    const std::vector<char>* allowed) {
    std::vector<std::vector<int>> choice(n_cam_);  // per camera: usable detection indices
    for (int c = 0; c < n_cam_; ++c)
      for (int d = 0; d < container.detCountForCam(c, frame); ++d)
        if (!allowed || (*allowed)[static_cast<std::size_t>(flatIndex(frame, c, d))]) choice[c].push_back(d);
    std::vector<Combination> pushed;
    std::vector<int> digit(n_cam_, -1);
    int depth = 0;  // number of cameras decided; digits beyond are -1
    // pre-order DFS over prefixes: descend with 0, advance the last decided digit, climb when exhausted
    auto advance = [&]() {  // move to the next sibling/uncle; false when the tree is exhausted
      while (depth > 0) {
        if (digit[depth - 1] < static_cast<int>(choice[depth - 1].size())) {
          ++digit[depth - 1];
          return true;
        }
        digit[depth - 1] = -1;
        --depth;
      }
      return false;
    };
    bool more = true;
    digit[0] = 0;
    depth = 1;
    while (more) {
      int count = 0;
      for (int c = 0; c < depth; ++c) count += digit[c] > 0;
      bool cut = false;
      if (count >= 2) {
        std::vector<Triangulator::CamPointPair> images;
        for (int c = 0; c < depth; ++c)
          if (digit[c] > 0)
            images.push_back({triangulator_->getCameras()[c], container.getRecord(c, frame, choice[c][digit[c] - 1])});
        auto res = triangulator_->triangulatePoint(images);
        if (res.second > error_) {
          cut = true;
        } else if (depth == n_cam_ && count >= MIN_CAMERAS) {
          Combination cmb;
          cmb.combination_.resize(n_cam_);
          for (int c = 0; c < n_cam_; ++c) cmb.combination_[c] = digit[c] > 0 ? choice[c][digit[c] - 1] + 1 : 0;
          cmb.point = res.first;
          cmb.error = res.second;
          pushed.push_back(cmb);
        }
      }
      if (!cut && depth < n_cam_) {  // descend
        digit[depth] = 0;
        ++depth;
      } else {
        more = advance();
      }
    }
    return pushed;
  }

  std::int64_t flatIndex(int frame, int cam, int det) const {
    return det_offset_[static_cast<std::size_t>(frame) * n_cam_ + cam] - frameDetBase(frame) + det;
  }
  std::int64_t frameDetBase(int frame) const { return det_offset_[static_cast<std::size_t>(frame) * n_cam_]; }
  std::int64_t frameDetCount(int frame) const {
    std::int64_t n = 0;
    for (int c = 0; c < n_cam_; ++c) n += det_count_[static_cast<std::size_t>(frame) * n_cam_ + c];
    return n;
  }

  // ---- batched route: the frame's survivors decoded and ordered once, then scanned in priority order.  The
  // reference builds a fresh priority queue per path and per frame (fillCombinationQueue) and takes its first
  // acceptable entry; scanning one ordered list for the first acceptable entry gives the same answer without
  // materialising a vector of combinations per path.
  // all frames at once, after the scoring call: a team of threads, each taking a contiguous range of frames
  void orderSurvivors() {
    const int n_frames = static_cast<int>(frame_begin_.size()) - 1;
    all_digits_.resize(survivors_.size() * static_cast<std::size_t>(n_cam_));
    all_order_.resize(survivors_.size());
    auto work = [&](int f0, int f1) {
      std::vector<int> zeros;
      for (int frame = f0; frame < f1; ++frame) {
        const std::int64_t b = frame_begin_[frame], e = frame_begin_[frame + 1];
        const std::size_t n = static_cast<std::size_t>(e - b);
        zeros.resize(n);
        const std::int32_t* sizes = &det_count_[static_cast<std::size_t>(frame) * n_cam_];
        std::int8_t* digits = all_digits_.data() + static_cast<std::size_t>(b) * n_cam_;
        std::int32_t* order = all_order_.data() + b;
        for (std::size_t i = 0; i < n; ++i) {
          std::int64_t leaf = survivors_[static_cast<std::size_t>(b) + i].leaf;  // tdr_decode_combination, inline
          int z = 0;
          for (int c = n_cam_ - 1; c >= 0; --c) {
            const std::int64_t size = static_cast<std::int64_t>(sizes[c]) + 1;
            const int d = static_cast<int>(leaf % size);
            leaf /= size;
            digits[i * n_cam_ + c] = static_cast<std::int8_t>(d);
            z += d == 0;
          }
          zeros[i] = z;
          order[i] = static_cast<std::int32_t>(i);
        }
        const tdr_candidate* sv = survivors_.data() + b;
        std::stable_sort(order, order + n, [&](std::int32_t x, std::int32_t y) {
          return zeros[x] != zeros[y] ? zeros[x] < zeros[y] : sv[x].err < sv[y].err;
        });
      }
    };
    const int T = std::max(1, std::min<int>({8, static_cast<int>(std::thread::hardware_concurrency()), n_frames / 64 + 1}));
    std::vector<std::thread> team;
    struct Join {
      std::vector<std::thread>& t;
      ~Join() {
        for (auto& th : t)
          if (th.joinable()) th.join();
      }
    } join{team};
    // equal shares of the survivors rather than of the frames
    int f0 = 0;
    for (int t = 0; t < T; ++t) {
      const std::int64_t target = static_cast<std::int64_t>(survivors_.size()) * (t + 1) / T;
      int f1 = t + 1 == T ? n_frames : f0;
      while (t + 1 < T && f1 < n_frames && frame_begin_[f1] < target) ++f1;
      if (t + 1 == T) work(f0, f1);
      else team.emplace_back(work, f0, f1);
      f0 = f1;
    }
  }
  void loadFrame(int frame) {  // the frame's slice of the arrays orderSurvivors() filled
    frame_digits_ = all_digits_.data() + static_cast<std::size_t>(frame_begin_[frame]) * n_cam_;
    frame_order_ = all_order_.data() + frame_begin_[frame];
    frame_n_ = static_cast<std::size_t>(frame_begin_[frame + 1] - frame_begin_[frame]);
  }
  const std::int8_t* digitsOf(std::size_t i) const { return frame_digits_ + i * static_cast<std::size_t>(n_cam_); }
  Combination materialise(int frame, std::size_t i) const {
    const tdr_candidate& sv = survivors_[static_cast<std::size_t>(frame_begin_[frame]) + i];
    Combination c;
    c.combination_.assign(digitsOf(i), digitsOf(i) + n_cam_);
    c.point = cv::Point3d(sv.x, sv.y, sv.z);
    c.error = sv.err;
    return c;
  }

  // pushed combinations of fillCombinationQueue(container, frame) in priority order
  std::vector<Combination> candidates(const DetectionsContainer& container, int frame,
                                      const std::vector<char>* allowed) {
    std::vector<Combination> out;
    if (batched_) {
      for (std::int64_t s = frame_begin_[frame]; s < frame_begin_[frame + 1]; ++s) {
        Combination c = decode(survivors_[static_cast<std::size_t>(s)]);
        bool ok = true;
        if (allowed)
          for (int cam = 0; cam < n_cam_ && ok; ++cam)
            if (c.combination_[cam] > 0)
              ok = (*allowed)[static_cast<std::size_t>(flatIndex(frame, cam, c.combination_[cam] - 1))] != 0;
        if (ok) out.push_back(std::move(c));
      }
    } else {
      out = walkTree(container, frame, allowed);
    }
    std::stable_sort(out.begin(), out.end(), [](const Combination& a, const Combination& b) { return a.before(b); });
    return out;
  }

  // For every active path: which detections of the frame lie within MAX_STEP of its last position
  // (the tmpContainer of triangulateWithLastPos, src/DroneClassifier.cpp:224-236).  One batched
  // getDistFromRay launch for all paths of the frame.
  std::vector<std::vector<char>> nearLastPositions(const DetectionsContainer& container, int frame,
                                                   const std::vector<int>& active,
                                                   const std::vector<std::vector<cv::Point3d>>& paths) {
    const std::int64_t n_det = frameDetCount(frame);
    std::vector<std::vector<char>> near(active.size(), std::vector<char>(static_cast<std::size_t>(n_det), 0));
    const std::int64_t n = n_det * static_cast<std::int64_t>(active.size());
    if (n == 0) return near;
    tdr_ctx* ctx = triangulator_->context();
    std::vector<std::int32_t> idx(static_cast<std::size_t>(n));
    std::vector<double> uv(static_cast<std::size_t>(2 * n)), pt(static_cast<std::size_t>(3 * n)),
        dist(static_cast<std::size_t>(n));
    std::size_t k = 0;
    for (std::size_t a = 0; a < active.size(); ++a) {
      const cv::Point3d pos = paths[active[a]].back();
      for (int c = 0; c < n_cam_; ++c)
        for (int d = 0; d < container.detCountForCam(c, frame); ++d, ++k) {
          const cv::Point2d r = container.getRecord(c, frame, d);
          idx[k] = c;
          uv[2 * k] = r.x; uv[2 * k + 1] = r.y;
          pt[3 * k] = pos.x; pt[3 * k + 1] = pos.y; pt[3 * k + 2] = pos.z;
        }
    }
    // one pinned staging block, one launch, one synchronisation per frame (the frames are inherently
    // sequential: a path's last position is the result of the previous frame)
    const auto t0 = std::chrono::steady_clock::now();
    check(tdr_dist_from_ray_batch(ctx, idx.data(), uv.data(), pt.data(), n, dist.data()));
    dist_batch_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (std::size_t a = 0; a < active.size(); ++a)
      for (std::int64_t j = 0; j < n_det; ++j)
        near[a][static_cast<std::size_t>(j)] = dist[a * static_cast<std::size_t>(n_det) + j] < MAX_STEP;
    return near;
  }

  // src/DroneClassifier.cpp:219-250
  Combination triangulateWithLastPos(cv::Point3d pos, const DetectionsContainer& container,
                                     const std::vector<char>& near,
                                     const std::vector<Combination>& usedCombinations, int frame) {
    if (batched_) {
      loadFrame(frame);
      const tdr_candidate* sv = survivors_.data() + frame_begin_[frame];
      // usable[cam][digit]: digit 0 ("no detection") always, digit d > 0 iff detection d - 1 lies near the path's
      // last position (the tmpContainer) — one table per call, so the scan below is n_cam byte loads per candidate
      constexpr int kStride = 128;  // digits are int8
      usable_.assign(static_cast<std::size_t>(n_cam_) * kStride, 0);
      for (int cam = 0; cam < n_cam_; ++cam) {
        char* u = usable_.data() + static_cast<std::size_t>(cam) * kStride;
        u[0] = 1;
        const int k = det_count_[static_cast<std::size_t>(frame) * n_cam_ + cam];
        for (int d = 0; d < k && d + 1 < kStride; ++d) u[d + 1] = near[static_cast<std::size_t>(flatIndex(frame, cam, d))];
      }
      taken_.reset(n_cam_);
      for (const auto& u : usedCombinations) taken_.add(u.combination_.data());
      for (std::size_t oi = 0; oi < frame_n_; ++oi) {
        const std::int32_t i = frame_order_[oi];
        const std::int8_t* digits = digitsOf(static_cast<std::size_t>(i));
        char ok = 1;
        for (int cam = 0; cam < n_cam_; ++cam) ok &= usable_[static_cast<std::size_t>(cam) * kStride + digits[cam]];
        if (!ok) continue;
        const double dx = sv[i].x - pos.x, dy = sv[i].y - pos.y, dz = sv[i].z - pos.z;
        if (!taken_.clashes(digits) && sv[i].err < error_ &&
            std::sqrt(dx * dx + dy * dy + dz * dz) < MAX_STEP)
          return materialise(frame, static_cast<std::size_t>(i));
      }
      return {};
    }
    for (const Combination& c : candidates(container, frame, &near)) {
      const double dx = c.point.x - pos.x, dy = c.point.y - pos.y, dz = c.point.z - pos.z;
      if (c.isCombinationUnique(usedCombinations) && c.error < error_ &&
          std::sqrt(dx * dx + dy * dy + dz * dz) < MAX_STEP)
        return c;
    }
    return {};
  }

  // src/DroneClassifier.cpp:200-217
  std::vector<Combination> pickBestCombinations(const DetectionsContainer& container, int frame,
                                                const std::vector<Combination>& usedCombinations) {
    std::vector<Combination> finalCombinations;
    if (batched_) {
      loadFrame(frame);
      const tdr_candidate* sv = survivors_.data() + frame_begin_[frame];
      taken_.reset(n_cam_);  // usedCombinations and, as they are picked, finalCombinations
      for (const auto& u : usedCombinations) taken_.add(u.combination_.data());
      for (std::size_t oi = 0; oi < frame_n_; ++oi) {
        const std::int32_t i = frame_order_[oi];
        const std::int8_t* digits = digitsOf(static_cast<std::size_t>(i));
        if (!taken_.clashes(digits) && sv[i].err < error_) {
          finalCombinations.push_back(materialise(frame, static_cast<std::size_t>(i)));
          taken_.add(digits);
        }
      }
      return finalCombinations;
    }
    for (const Combination& c : candidates(container, frame, nullptr))
      if (c.isCombinationUnique(finalCombinations) && c.isCombinationUnique(usedCombinations) && c.error < error_)
        finalCombinations.push_back(c);
    return finalCombinations;
  }

  // src/DroneClassifier.cpp:263-331
  void classifyPaths(const std::vector<Combination>& finalCombinations,
                     std::vector<std::vector<cv::Point3d>>& paths, std::vector<int>& processedPaths,
                     std::vector<std::vector<int>>& emptyFrames, int frame) {
    struct Assign {
      std::size_t combination, path;
      double error;
    };
    auto taken = [&](std::size_t p) {
      return std::find(processedPaths.begin(), processedPaths.end(), static_cast<int>(p)) != processedPaths.end();
    };
    std::vector<Assign> assign;
    for (std::size_t i = 0; i < finalCombinations.size(); ++i) {
      std::size_t bestPath = 0;
      double bestDist = -1;
      for (std::size_t j = 0; j < n_drones_; ++j) {
        if (taken(j)) continue;
        const std::size_t tail = std::min<std::size_t>(paths[j].size(), PATH_TAIL);
        if (tail == 0) continue;
        double dist = 0;
        for (std::size_t t = paths[j].size() - tail; t < paths[j].size(); ++t) {
          const double dx = paths[j][t].x - finalCombinations[i].point.x,
                       dy = paths[j][t].y - finalCombinations[i].point.y,
                       dz = paths[j][t].z - finalCombinations[i].point.z;
          dist += std::sqrt(dx * dx + dy * dy + dz * dz);
        }
        dist /= static_cast<double>(tail);
        if (dist < bestDist || bestDist == -1) {
          bestDist = dist;
          bestPath = j;
        }
      }
      assign.push_back({i, bestPath, bestDist});
    }
    std::stable_sort(assign.begin(), assign.end(), [](const Assign& a, const Assign& b) { return a.error < b.error; });
    for (const Assign& a : assign) {
      if (taken(a.path)) {  // closest path already taken (or no path has points yet): open an empty path
        int emptyPath = -1;
        for (std::size_t i = 0; i < paths.size(); ++i)
          if (paths[i].empty()) {
            emptyPath = static_cast<int>(i);
            break;
          }
        if (emptyPath != -1) {
          paths[emptyPath].push_back(finalCombinations[a.combination].point);
          processedPaths.push_back(emptyPath);
        }
      } else {
        paths[a.path].push_back(finalCombinations[a.combination].point);
        processedPaths.push_back(static_cast<int>(a.path));
      }
    }
    for (std::size_t i = 0; i < n_drones_; ++i)
      if (!taken(i)) emptyFrames[i].push_back(frame);
  }
};
