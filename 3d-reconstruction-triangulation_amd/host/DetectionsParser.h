// DetectionsParser.h — parallel ingest of the reference's per-camera detection CSVs ("next" row 3 of
// SURVEY.md §8f) straight into the flat layouts the engine consumes.
//
// The reference reads every file line by line through std::getline / std::istringstream / std::stoi into
// vector<vector<vector<Point2d>>> (src/DetectionsContainer.cpp:19-76).  Here a file is mapped into memory
// and parsed by a team of threads in two passes:
//   pass 1 (parallel)   every thread takes a byte range cut at line starts and records, per line, where it
//                       begins and its frame number (first field);
//   index   (serial)    one cheap walk over the frame numbers applies the reference's sequential rules —
//                       `offset` lines skipped, [startFrame, endFrame] filter, "frame numbers missing from a
//                       file become empty frames" (:46-49) — and gives every kept line its output frame;
//   pass 2 (parallel)   the lines are parsed field by field and the bbox centres (columns j*recordSize+5, +6)
//                       go into the camera's flat CSR arrays (count / first / points) — what the container and
//                       the classifier read; no nested vectors anywhere;
//   rows    (parallel)  while no frame of any camera holds more than one detection, ONE flat block
//                       rows[camera][frame] of (x, y) doubles with (-1,-1) for "no detection" is filled: byte for
//                       byte what getDataForTriangulation() returns row by row
//                       (src/DetectionsContainer.cpp:145-171) and what tdr_*_triangulate_points take.  The block
//                       comes from a caller-supplied allocator, so the adapter hands out pinned host memory
//                       (tdr_host_alloc) and the upload is a plain DMA; a second block holds the same rows as
//                       int16 pairs (the narrow observation format, include/tdr_amd.h) when every value fits.
// Kept from the reference: std::stoi semantics for EVERY field (leading white space, optional sign, digits,
// the rest of the token ignored: "10.7" -> 10; no digits -> std::invalid_argument; overflow ->
// std::out_of_range), empty tokens skipped, "Invalid CSV file!" when (fields - 1) % recordSize != 0, files in
// sorted path order, and the container-level exceptions ("There must be at least 2 cameras", "Number of frames
// on all cameras must be the same").  A line without any field indexes an empty vector in the reference
// (undefined behaviour); here it is "Invalid CSV file!".
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace tdr_io {

struct CameraTrack {  // one camera's detections, frame-major CSR
  std::vector<int32_t> count;  // [n_frames] detections in the frame
  std::vector<int64_t> first;  // [n_frames] index of the frame's first detection in `pts`
  std::vector<double> pts;     // [total][2]
};

// memory for the flat rows: alloc(bytes) / free(ptr).  Defaults to malloc / free.
struct RowAllocator {
  std::function<void*(std::size_t)> alloc = [](std::size_t n) { return std::malloc(n ? n : 1); };
  std::function<void(void*)> free = [](void* p) { std::free(p); };
};

struct ParsedDetections {
  int n_cameras = 0;
  int64_t n_frames = 0;
  std::vector<CameraTrack> cams;
  bool one_drone = true;  // no frame of any camera holds more than one detection
  bool fits_i16 = true;   // every coordinate of `rows` is an int16 value
  double* rows = nullptr;     // [n_cameras][n_frames][2], valid while one_drone
  int16_t* rows16 = nullptr;  // the same as int16 pairs, valid while one_drone && fits_i16
  std::size_t bytes_parsed = 0;
};

namespace detail {

struct MappedFile {
  const char* data = nullptr;
  std::size_t size = 0;
  int fd = -1;
  explicit MappedFile(const std::string& path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return;  // the reference's ifstream silently reads nothing from a file it cannot open
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size == 0) return;
    void* p = ::mmap(nullptr, (std::size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return;
    data = static_cast<const char*>(p);
    size = (std::size_t)st.st_size;
    (void)::madvise(p, size, MADV_SEQUENTIAL);
  }
  ~MappedFile() {
    if (data) ::munmap(const_cast<char*>(data), size);
    if (fd >= 0) ::close(fd);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
};

enum ParseError { kNone = 0, kInvalidArgument = 1, kOutOfRange = 2, kInvalidCsv = 3 };

// std::stoi on the token [p, end): 0 = ok
inline int stoi_token(const char* p, const char* end, int& out) {
  while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;  // std::isspace in the "C" locale
  bool neg = false;
  if (p < end && (*p == '+' || *p == '-')) neg = *p++ == '-';
  if (p >= end || *p < '0' || *p > '9') return kInvalidArgument;
  long long v = 0;
  bool overflow = false;
  for (; p < end && *p >= '0' && *p <= '9'; ++p) {
    v = v * 10 + (*p - '0');
    if (v > (long long)INT_MAX + 1) overflow = true, v = (long long)INT_MAX + 1;  // keep consuming digits
  }
  if (neg) v = -v;
  if (overflow || v > INT_MAX || v < INT_MIN) return kOutOfRange;
  out = (int)v;
  return kNone;
}

struct LineRef {
  std::size_t begin, end;  // [begin, end) without the '\n'
  int frame;               // first field
  int err;                 // error met while reading the first field — or, on a line the frame window drops, any field
};

template <class F>
void parallel_for(int threads, F&& body) {
  if (threads <= 1) {
    body(0);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 1; t < threads; ++t) th.emplace_back([&body, t] { body(t); });
  body(0);
  for (auto& x : th) x.join();
}

[[noreturn]] inline void throw_parse_error(int kind) {
  if (kind == kInvalidArgument) throw std::invalid_argument("stoi");
  if (kind == kOutOfRange) throw std::out_of_range("stoi");
  throw std::runtime_error("Invalid CSV file!");
}

}  // namespace detail

// One file -> one CameraTrack.
struct ParsedFile {
  CameraTrack track;
  int64_t n_frames = 0;
  bool one_drone = true;
  std::size_t bytes = 0;
};

inline ParsedFile parse_detection_file(const std::string& path, int offset, int recordSize, int startFrame,
                                       int endFrame, int threads) {
  using namespace detail;
  ParsedFile out;
  MappedFile file(path);
  out.bytes = file.size;
  const char* d = file.data;
  const std::size_t size = file.size;
  if (threads < 1) threads = 1;
  if (size < ((std::size_t)1 << 16)) threads = 1;
  // ---- pass 1: line starts and frame numbers
  std::vector<std::size_t> cut((std::size_t)threads + 1, size);
  cut[0] = 0;
  for (int t = 1; t < threads; ++t) {
    std::size_t p = size * (std::size_t)t / (std::size_t)threads;
    while (p < size && d[p - 1] != '\n') ++p;  // next line start
    cut[(std::size_t)t] = p;
  }
  std::vector<std::vector<LineRef>> found((std::size_t)threads);
  parallel_for(threads, [&](int t) {
    std::size_t p = cut[(std::size_t)t];
    const std::size_t stop = cut[(std::size_t)t + 1];
    auto& v = found[(std::size_t)t];
    v.reserve((stop - p) / 24 + 4);
    while (p < stop) {
      const char* nl = static_cast<const char*>(std::memchr(d + p, '\n', size - p));
      const std::size_t e = nl ? (std::size_t)(nl - d) : size;
      LineRef ln{p, e, 0, kNone};
      // first non-empty token
      std::size_t q = p;
      bool have = false;
      while (q <= e && !have) {
        std::size_t c = q;
        while (c < e && d[c] != ',') ++c;
        if (c > q) {
          ln.err = stoi_token(d + q, d + c, ln.frame);
          have = true;
        }
        if (c >= e) break;
        q = c + 1;
      }
      if (!have) ln.err = kInvalidCsv;  // no field at all: undefined behaviour in the reference
      // A line the frame window drops is never seen by pass 2, but the reference has already run std::stoi over
      // ALL of its fields by then (src/DetectionsContainer.cpp:33-38 precede the filter at :40-44): a malformed
      // field there throws.  Whether the window drops a line depends on its own frame number only, so the rest of
      // such a line is validated right here; kept lines are validated where they are parsed.
      if (have && ln.err == kNone && startFrame != endFrame && (ln.frame <= startFrame || ln.frame > endFrame)) {
        bool first = true;  // the first non-empty token is the frame number, parsed above
        q = p;
        while (q <= e && ln.err == kNone) {
          std::size_t c = q;
          while (c < e && d[c] != ',') ++c;
          if (c > q) {
            int unused = 0;
            if (!first) ln.err = stoi_token(d + q, d + c, unused);
            first = false;
          }
          if (c >= e) break;
          q = c + 1;
        }
      }
      v.push_back(ln);
      p = e + 1;
    }
  });
  // ---- index: the reference's sequential rules (src/DetectionsContainer.cpp:29-49)
  std::size_t n_lines = 0;
  for (auto& v : found) n_lines += v.size();
  std::vector<LineRef> kept;
  std::vector<int64_t> kept_frame;
  int index_err = kNone;  // a line with a malformed first field (or any field, if the window drops it); every kept line precedes it
  kept.reserve(n_lines);
  kept_frame.reserve(n_lines);
  {
    int n_line = 0, frame = -1;
    int64_t idx = 0;
    for (auto& v : found) {
      if (index_err != kNone) break;
      for (const LineRef& ln : v) {
        if (offset > n_line++) continue;
        if (ln.err != kNone) {  // the reference stops at the first bad line in file order: nothing after it counts
          index_err = ln.err;
          break;
        }
        if ((ln.frame <= startFrame || ln.frame > endFrame) && startFrame != endFrame) {
          frame = ln.frame;
          continue;
        }
        const int gap = ln.frame - frame - 1;
        if (gap > 0) idx += gap;
        frame = ln.frame;
        kept.push_back(ln);
        kept_frame.push_back(idx++);
      }
    }
    out.n_frames = idx;
  }
  // ---- pass 2: fields of the kept lines.  Detections are buffered per thread (file order inside a thread,
  // threads own consecutive line ranges), then laid out by one prefix sum.
  const std::size_t n_kept = kept.size();
  std::vector<int32_t> line_count(n_kept, 0);
  std::vector<std::vector<double>> dets((std::size_t)threads);
  std::vector<std::size_t> first_bad((std::size_t)threads, SIZE_MAX);
  std::vector<int> bad_kind((std::size_t)threads, kNone);
  parallel_for(threads, [&](int t) {
    const std::size_t a = n_kept * (std::size_t)t / (std::size_t)threads, b = n_kept * (std::size_t)(t + 1) / (std::size_t)threads;
    auto& buf = dets[(std::size_t)t];
    buf.reserve((b - a) * 2 + 16);
    std::vector<int> fields;
    for (std::size_t i = a; i < b; ++i) {
      const LineRef& ln = kept[i];
      fields.clear();
      int err = kNone;
      std::size_t q = ln.begin;
      while (q <= ln.end && err == kNone) {
        std::size_t c = q;
        while (c < ln.end && d[c] != ',') ++c;
        if (c > q) {
          int v = 0;
          err = stoi_token(d + q, d + c, v);
          fields.push_back(v);
        }
        if (c >= ln.end) break;
        q = c + 1;
      }
      if (err == kNone && (fields.size() - 1) % (std::size_t)recordSize != 0) err = kInvalidCsv;
      if (err != kNone) {
        first_bad[(std::size_t)t] = i;
        bad_kind[(std::size_t)t] = err;
        return;  // later lines of this thread come after the failing one: never reached by the reference
      }
      const std::size_t k = fields.size() / (std::size_t)recordSize;
      line_count[i] = (int32_t)k;
      for (std::size_t j = 0; j < k; ++j) {
        buf.push_back((double)fields[j * (std::size_t)recordSize + 5]);
        buf.push_back((double)fields[j * (std::size_t)recordSize + 6]);
      }
    }
  });
  {
    std::size_t worst = SIZE_MAX;
    int kind = kNone;
    for (int t = 0; t < threads; ++t)
      if (first_bad[(std::size_t)t] < worst) worst = first_bad[(std::size_t)t], kind = bad_kind[(std::size_t)t];
    if (kind != kNone) throw_parse_error(kind);
    if (index_err != kNone) throw_parse_error(index_err);
  }
  CameraTrack& tr = out.track;
  tr.count.assign((std::size_t)out.n_frames, 0);
  tr.first.assign((std::size_t)out.n_frames, 0);
  std::size_t total = 0;
  for (auto& b : dets) total += b.size() / 2;
  tr.pts.resize(total * 2);
  {
    std::size_t at = 0;
    for (auto& b : dets) {
      if (!b.empty()) std::memcpy(tr.pts.data() + at, b.data(), b.size() * sizeof(double));
      at += b.size();
    }
  }
  // count per frame, then first[] as the running sum over ALL frames (gap frames included)
  for (std::size_t i = 0; i < n_kept; ++i) {
    tr.count[(std::size_t)kept_frame[i]] = line_count[i];
    if (line_count[i] > 1) out.one_drone = false;
  }
  int64_t run = 0;
  for (int64_t f = 0; f < out.n_frames; ++f) {
    tr.first[(std::size_t)f] = run;
    run += tr.count[(std::size_t)f];
  }
  return out;
}

// All files of a recording -> ParsedDetections.  Throws the reference's exceptions.
inline ParsedDetections parse_detection_files(const std::vector<std::string>& files, int offset, int recordSize,
                                              int startFrame, int endFrame, int threads,
                                              const RowAllocator& allocator = RowAllocator()) {
  ParsedDetections out;
  if (threads <= 0) {
    threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
  }
  // files are independent: min(files, threads) of them are parsed at the same time, each by its share of the
  // threads.  An exception is re-thrown for the FIRST failing file in sorted order, as a sequential read would.
  std::vector<int64_t> frames;
  {
    const int n_files = (int)files.size();
    const int outer = n_files < threads ? n_files : threads;
    const int inner = outer > 0 ? (threads / outer > 0 ? threads / outer : 1) : 1;
    std::vector<ParsedFile> parsed((std::size_t)n_files);
    std::vector<std::exception_ptr> failed((std::size_t)n_files);
    detail::parallel_for(outer, [&](int w) {
      for (int i = w; i < n_files; i += outer) {
        try {
          parsed[(std::size_t)i] = parse_detection_file(files[(std::size_t)i], offset, recordSize, startFrame, endFrame, inner);
        } catch (...) {
          failed[(std::size_t)i] = std::current_exception();
        }
      }
    });
    for (int i = 0; i < n_files; ++i)
      if (failed[(std::size_t)i]) std::rethrow_exception(failed[(std::size_t)i]);
    for (auto& pf : parsed) {
      out.bytes_parsed += pf.bytes;
      out.one_drone = out.one_drone && pf.one_drone;
      frames.push_back(pf.n_frames);
      out.cams.push_back(std::move(pf.track));
    }
  }
  if (out.cams.size() < 2) throw std::runtime_error("There must be at least 2 cameras");
  for (std::size_t i = 1; i < frames.size(); ++i)
    if (frames[i - 1] != frames[i]) throw std::runtime_error("Number of frames on all cameras must be the same");
  out.n_cameras = (int)out.cams.size();
  out.n_frames = frames[0];
  if (out.one_drone) {
    const std::size_t n = (std::size_t)out.n_cameras * (std::size_t)out.n_frames;
    out.rows = static_cast<double*>(allocator.alloc(n * 16));
    out.rows16 = static_cast<int16_t*>(allocator.alloc(n * 4));
    if (!out.rows || !out.rows16) {  // nobody owns the blocks yet: release what was obtained
      if (out.rows) allocator.free(out.rows);
      if (out.rows16) allocator.free(out.rows16);
      out.rows = nullptr;
      out.rows16 = nullptr;
      throw std::bad_alloc();
    }
    std::vector<char> fits((std::size_t)threads, 1);
    detail::parallel_for(threads, [&](int t) {
      const int64_t a = out.n_frames * t / threads, b = out.n_frames * (t + 1) / threads;
      for (int c = 0; c < out.n_cameras; ++c) {
        const CameraTrack& tr = out.cams[(std::size_t)c];
        double* r = out.rows + 2 * ((std::size_t)c * (std::size_t)out.n_frames);
        int16_t* r16 = out.rows16 + 2 * ((std::size_t)c * (std::size_t)out.n_frames);
        for (int64_t f = a; f < b; ++f) {
          double x = -1, y = -1;
          if (tr.count[(std::size_t)f] == 1) {
            x = tr.pts[2 * (std::size_t)tr.first[(std::size_t)f]];
            y = tr.pts[2 * (std::size_t)tr.first[(std::size_t)f] + 1];
          }
          r[2 * f] = x;
          r[2 * f + 1] = y;
          const bool in16 = x >= -32768 && x <= 32767 && y >= -32768 && y <= 32767;
          if (!in16) fits[(std::size_t)t] = 0;
          r16[2 * f] = in16 ? (int16_t)x : (int16_t)-1;
          r16[2 * f + 1] = in16 ? (int16_t)y : (int16_t)-1;
        }
      }
    });
    for (char ok : fits) out.fits_i16 = out.fits_i16 && ok;
  }
  return out;
}

}  // namespace tdr_io
