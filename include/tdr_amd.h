/*
 * tdr_amd.h — C ABI of the MI355X-native multi-view triangulation engine.
 *
 * Drop-in boundary for the reference's `Triangulator::triangulatePoints` hot path
 * (Grzetan/3D-Reconstruction-Triangulation).  The reference has no FFI of its own: its
 * "plugin" surface is the C++ abstract class `Triangulator` (src/Triangulator.h:9-59).
 * This header is what a C++/cgo/ctypes binding of that surface binds to; the C++ adapter
 * with the reference's exact class names and signatures lives in
 * 3d-reconstruction-triangulation_amd/host/ and is a thin layer over these entry points.
 * INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; nothing throws across this boundary.  Every function
 *     returns a tdr_status (0 = ok, negative = error); tdr_strerror() gives the reference's
 *     exception text for the codes that map to one.
 *   - all arithmetic is IEEE fp64, as in the reference.
 *   - detections:  pts[cam][sys] of (double x, double y), 16 B per observation, camera-major —
 *     exactly the memory of `std::vector<std::vector<cv::Point2d>>` rows as produced by
 *     DetectionsContainer::getDataForTriangulation (src/DetectionsContainer.cpp:145-171).
 *     A missing observation is the reference's sentinel: x == -1 || y == -1
 *     (src/MatrixTriangulator.cpp:86, src/RayTriangulator.cpp:66).
 *   - results: xyz[sys] of (double x, y, z), 24 B per system = the memory of
 *     `std::vector<cv::Point3d>`.
 *   - a "system" is one independent solve (one frame of one drone).
 *   - `_device` entry points take device pointers, enqueue on `stream` (a hipStream_t passed
 *     as void*, NULL = the context's own stream) and do not synchronise.
 *   - there is NO CPU fallback: without a usable MI355X/HIP device tdr_create fails with
 *     TDR_E_NO_DEVICE and nothing else can be called.
 */
#ifndef TDR_AMD_H
#define TDR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDR_MAX_CAMERAS 32
#define TDR_CAM_AUX_DOUBLES 10 /* rquat[4] (w-first), tvec[3], fovy, width, height */

typedef enum tdr_status {
  TDR_OK = 0,
  /* "Every camera should have the same number of points" / malformed shapes
   * (src/MatrixTriangulator.cpp:72-76, src/RayTriangulator.cpp:54-58); also the inputs that are
   * undefined behaviour in the reference (empty `points`; ray path with more point rows than
   * cameras, src/RayTriangulator.cpp:65,69) */
  TDR_E_SHAPE = -1,
  /* "Too few rays are found" (src/MatrixTriangulator.cpp:92-94) /
   * "Too few detections are found" (src/RayTriangulator.cpp:72-74): a system has < 2 valid
   * observations; the whole batch fails and *first_bad is that system's index */
  TDR_E_TOO_FEW = -2,
  /* tdr::Camera's "Set ... first." throws (src/Camera.h:79,90,99,112,123) */
  TDR_E_CAMERA = -3,
  TDR_E_HIP = -4,       /* a HIP runtime call failed; tdr_last_error() has the text */
  TDR_E_NO_DEVICE = -5, /* no HIP device / kernels not loadable: there is no CPU fallback */
  TDR_E_RCCL = -6,
  TDR_E_ARG = -7 /* NULL pointer / out-of-range argument */
} tdr_status;

const char* tdr_strerror(int status);
/* Text of the last HIP/RCCL failure on the calling thread ("" if none). */
const char* tdr_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Camera model — replaces tdr::Camera::compCamParams (src/Camera.h:177-187) as driven by
 * createCamera (src/utils.cpp:94-107).  Host-side, once per camera.
 * ---------------------------------------------------------------------------------------- */
typedef struct tdr_camera {
  int32_t id, width, height, cx, cy, diagonal;
  double fx, fy, fovx, fovy;
  double rquat[4];  /* as given, w-first, never normalised (src/Camera.h:274-279) */
  double tvec[3];   /* XCP POSITION */
  double rvec[3];   /* toRotVec(rquat) (src/utils.cpp:102); unused by the path */
  double campos[3]; /* -R^-1 tvec */
  double K[9];      /* cameraMatrix, row-major */
  double E[12];     /* cameraExtrinsicMatrix [R^-1 | campos], row-major 3x4 */
  double P[12];     /* cameraPerspectiveMatrix = K E, row-major 3x4 */
} tdr_camera;

int tdr_camera_derive(int id, int width, int height, double focal_length, const double tvec[3],
                      const double rquat[4], tdr_camera* out);
/* Pack the per-camera constants the kernels consume. P_out [n][12], aux_out [n][10]. */
int tdr_camera_pack(const tdr_camera* cams, int n_cam, double* P_out, double* aux_out);

/* ------------------------------------------------------------------------------------------
 * Context = one triangulator's camera set resident on one GPU.  Replaces the constructors
 * MatrixTriangulator(std::vector<const tdr::Camera*>) (src/MatrixTriangulator.h:19) and
 * RayTriangulator(...) (src/RayTriangulator.h:34): P / aux are snapshotted at creation.
 *   P        [n_cam][12]  row-major 3x4 cameraPerspectiveMatrix   (matrix path)
 *   cam_aux  [n_cam][10]  rquat, tvec, fovy, width, height         (ray path; may be NULL if
 *                          only the matrix path is used)
 *   device   HIP device ordinal
 * ---------------------------------------------------------------------------------------- */
typedef struct tdr_ctx tdr_ctx;

int tdr_create(const double* P, const double* cam_aux, int n_cam, int device, tdr_ctx** out);
/* Single-process multi-GPU context for hosts that keep their data in host memory (the C++ adapter):
 * one sub-context per entry of `devices` (NULL = ordinals 0..n_shards-1; an ordinal may repeat).
 * tdr_create(..., device = -1, ...) is shorthand for "every visible GPU".  The host-buffer batch
 * entry points block-partition the systems over the sub-contexts with one host thread each (every GPU
 * stages its own slice over its own PCIe link; results land in the caller's array, so no gather is
 * needed) and report the globally first failing system.  Single-system, scoring and helper entry
 * points run on the first sub-context; the `_device` batch entry points need a per-device context. */
int tdr_create_multi(const double* P, const double* cam_aux, int n_cam, const int* devices, int n_shards,
                     tdr_ctx** out);
void tdr_destroy(tdr_ctx* ctx);
/* Device-resident use of a multi-GPU context (C++ hosts that keep the stream on the GPUs): the sub-context of
 * shard g (for tdr_device_alloc / tdr_synth_detections_device / ... on that GPU), and the sharded batch with
 * the single gather of the path: every shard g triangulates n_sys[g] systems from d_pts[g] (camera rows
 * cam_stride[g] systems apart) into d_xyz[g] on its own GPU — no collective on the data path — then the
 * [n_sys[g]][3] blocks are gathered in shard order into d_xyz_root on shard 0's device (NULL: stay sharded).
 * transport: 1 = RCCL send/recv group over xGMI (one shard per distinct GPU; librccl is dlopen'ed on first
 * use), 2 = hipMemcpyPeerAsync, 0 = RCCL when usable, else peer copies.  Synchronous; *first_bad is the
 * globally first system with < 2 observations (then TDR_E_TOO_FEW), else -1. */
int tdr_multi_shard_count(const tdr_ctx* ctx);
int tdr_multi_shard(tdr_ctx* ctx, int g, tdr_ctx** shard);
int tdr_multi_matrix_triangulate_gather(tdr_ctx* ctx, const double* const* d_pts, int n_cam_pts,
                                        const int64_t* cam_stride, const int64_t* n_sys,
                                        double* const* d_xyz, double* d_xyz_root, int transport,
                                        int64_t* first_bad);
int tdr_n_cameras(const tdr_ctx* ctx);
int tdr_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Batch entry points on HOST buffers — replace
 *   MatrixTriangulator::triangulatePoints (src/MatrixTriangulator.cpp:70-100)
 *   RayTriangulator::triangulatePoints    (src/RayTriangulator.cpp:51-81)
 *   pts        [n_cam_pts][n_sys][2]
 *   out_xyz    [n_sys][3]
 *   out_err    [n_sys] or NULL.  matrix: sqrt(|AX-b|^2 / 2N) (src/MatrixTriangulator.cpp:55-59);
 *              ray: mean ray distance at the solver's last compute() (src/RayTriangulator.cpp:26,106)
 *   first_bad  (nullable) index of the first system with < 2 valid observations, else -1
 * Cameras used: matrix path min(n_cam_pts, n_cam) (src/MatrixTriangulator.cpp:84); ray path
 * n_cam_pts, which must be <= n_cam (src/RayTriangulator.cpp:65).
 * Synchronous.  On TDR_E_TOO_FEW the outputs are unspecified (the reference throws and
 * returns nothing).
 * Matrix path, rank-deficient or ill-conditioned systems (duplicate camera, cameras on one line of sight,
 * collinear centres through the point): the result is the reference's Moore-Penrose answer
 * (cv::invert(A, DECOMP_SVD), src/MatrixTriangulator.cpp:52-54: singular values <= 2 DBL_EPSILON sum(w)
 * dropped -> minimum-norm point), computed by a one-sided Jacobi SVD of A on the flagged lanes.
 * Large batches are cut into chunks that are uploaded, solved and downloaded concurrently (both directions
 * of the PCIe link busy at once); TDR_HOST_PIPE / TDR_HOST_CHUNK tune it.
 * ---------------------------------------------------------------------------------------- */
int tdr_matrix_triangulate_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys,
                                  double* out_xyz, double* out_err, int64_t* first_bad);
int tdr_ray_triangulate_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys,
                               double* out_xyz, double* out_err, int64_t* first_bad);

/* ------------------------------------------------------------------------------------------
 * LM run status of the ray path (tdr_ray_triangulate_points_ex / _device_ex).
 *
 * cv::LMSolver::run as the reference uses it (src/RayTriangulator.cpp:100-104, MAX_ITERATIONS 1000,
 * src/RayTriangulator.h:10) can stop away from the minimiser of its own objective
 * sum_i |d_i x (x - o_i)|^2 in two ways: at the iteration cap, and when the step test |d|_inf < FLT_EPSILON
 * fires on a heavily damped step.  The point it returns then depends on the last bits of hundreds of branchy
 * iterations — a one-ulp change of one ray direction moves it by up to metres
 * (tests/tools/lm_sensitivity.py, profiles/r02_lm_sensitivity.json) — so the reference has no reproducible
 * answer there, in any implementation.  Because the objective is an exact quadratic, whether a run reached
 * the minimiser x* has an exact answer (x - x* = H^-1 g, H = sum (|d|^2 I - d d^T),
 * g = sum (|d|^2 I - d d^T)(x - o)); the status word reports it:
 *   TDR_LM_CAP            the loop ended because iter reached MAX_ITERATIONS
 *   TDR_LM_NOT_CONVERGED  |x - x*|_inf > TDR_LM_CONVERGED_REL * max(|x*|_inf, 1) at the returned x
 *   TDR_LM_TOO_FEW        the system has < 2 valid observations (its xyz is NaN; the batch call fails)
 * status == 0: the run ended at the unique minimiser, and every faithful implementation of the reference
 * returns the same point to 2 * TDR_LM_CONVERGED_REL.  out_iters is the iteration count of the reference's
 * loop at its exit (1000 at the cap).
 * ---------------------------------------------------------------------------------------- */
#define TDR_LM_CAP 1
#define TDR_LM_NOT_CONVERGED 2
#define TDR_LM_TOO_FEW 4
#define TDR_LM_CONVERGED_REL 5e-8
/* tdr_ray_triangulate_points with the two optional per-system outputs out_iters [n_sys], out_status [n_sys]
 * (int32; either may be NULL).  Asking for the status costs one extra streaming pass over the detections on
 * the device; without it nothing extra runs. */
int tdr_ray_triangulate_points_ex(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys,
                                  double* out_xyz, double* out_err, int32_t* out_iters, int32_t* out_status,
                                  int64_t* first_bad);

/* Same calls with the input given as one pointer per camera row: rows[c] -> n_sys x (x, y), i.e. the
 * storage of each inner std::vector<cv::Point2d> of the reference's argument, staged to the GPU straight
 * from where it lies (no flattening copy on the host). */
int tdr_matrix_triangulate_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts,
                                       int64_t n_sys, double* out_xyz, double* out_err,
                                       int64_t* first_bad);
int tdr_ray_triangulate_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts, int64_t n_sys,
                                    double* out_xyz, double* out_err, int64_t* first_bad);

/* ------------------------------------------------------------------------------------------
 * Single-system entry points — replace the pure virtual
 *   triangulatePoint(std::vector<CamPointPair>) (src/Triangulator.h:46-47;
 *   src/MatrixTriangulator.cpp:3-62, src/RayTriangulator.cpp:83-107).
 * Each observation carries ITS OWN camera constants, as CamPointPair does
 * (src/Triangulator.h:11-15), so arbitrary camera subsets/orderings are allowed.
 *   P_obs   [n_obs][12]   (matrix)     aux_obs [n_obs][10]  (ray)      uv [n_obs][2]
 * A sentinel (-1) observation is rejected with TDR_E_SHAPE (the reference leaves its rows
 * uninitialised, src/MatrixTriangulator.cpp:12-14).  Runs on the GPU, one launch per call: the matrix call carries
 * its constants and pixels in the kernel argument and reads the result from a host-coherent pinned slot the kernel
 * writes (one launch, one wait, no copy); per-call cost and what it means for the reference's unchanged
 * DroneClassifier: INTEGRATION.md §5, profiles/r05_single_call.json.
 * ---------------------------------------------------------------------------------------- */
int tdr_matrix_triangulate_point(tdr_ctx* ctx, const double* P_obs, const double* uv, int n_obs,
                                 double out_xyz[3], double* out_err);
int tdr_ray_triangulate_point(tdr_ctx* ctx, const double* aux_obs, const double* uv, int n_obs,
                              double out_xyz[3], double* out_err);

/* ------------------------------------------------------------------------------------------
 * Device-resident entry points (pipelines, multi-GPU shards, roofline runs).
 *   d_pts      device, camera row c starts at d_pts + 2*c*cam_stride doubles; cam_stride is in
 *              systems (>= n_sys) so a GPU can work on a slice of a larger [cam][frame] array
 *   d_xyz      device [n_sys][3];   d_err device [n_sys] or NULL
 * Asynchronous on `stream`.  Failures of individual systems (<2 valid observations) are
 * recorded in the context (their xyz is NaN) and read back with tdr_first_bad(), which
 * synchronises the stream, returns the lowest failing index seen since the last call
 * (or -1) and clears it.
 * ---------------------------------------------------------------------------------------- */
int tdr_matrix_triangulate_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts,
                                         int64_t cam_stride, int64_t n_sys, double* d_xyz,
                                         double* d_err, void* stream);
int tdr_ray_triangulate_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts,
                                      int64_t cam_stride, int64_t n_sys, double* d_xyz,
                                      double* d_err, void* stream);
/* tdr_ray_triangulate_points_device with the per-system outputs d_iters / d_status (device int32 [n_sys],
 * either may be NULL) described at TDR_LM_* above. */
int tdr_ray_triangulate_points_device_ex(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                         int64_t n_sys, double* d_xyz, double* d_err, int32_t* d_iters,
                                         int32_t* d_status, void* stream);
int tdr_first_bad(tdr_ctx* ctx, void* stream, int64_t* first_bad);

/* ------------------------------------------------------------------------------------------
 * Ray path in CLOSED FORM — the cross-check SURVEY.md §7.1-6 asks for, and the HBM-bound form of the ray path.
 * What RayTriangulator::triangulatePoint hands to cv::LMSolver (src/RayTriangulator.cpp:83-107; rays from
 * createRayForPoint, src/Triangulator.cpp:15-55; residual distToRay, :3-9) is sum_i |d_i x (x - o_i)|^2, an exact
 * quadratic: its minimiser solves  sum_i (|d_i|^2 I - d_i d_i^T) x = sum_i (|d_i|^2 I - d_i d_i^T) o_i.  These entry
 * points return that minimiser in one streaming pass (n_cam x 16 B read + 24 B written per system).
 * They do NOT replace tdr_ray_triangulate_points*: the reference returns the LM iterate, which on a run that stops
 * early (TDR_LM_CAP / TDR_LM_NOT_CONVERGED) is not this point.  Where the LM status is 0 the two agree to
 * 2 * TDR_LM_CONVERGED_REL.
 *   out_err  (nullable) mean distance of the returned point from the rays, sum_i |d_i x (x - o_i)| / N — what
 *            RayClosestPoint::getError() holds after a compute() at the minimiser (src/RayTriangulator.cpp:21-26)
 * Shapes, the (-1) sentinel, the camera rule (n_cam_pts <= n_cam, src/RayTriangulator.cpp:65) and the too-few rule
 * ("Too few detections are found", :72-74; TDR_E_TOO_FEW with *first_bad) are those of tdr_ray_triangulate_points.
 * All rays parallel (the minimiser is a line): the point of that line an LM run with pseudo-inverse steps reaches
 * from the reference's start point, the mean of the ray origins (:90-98).
 * ---------------------------------------------------------------------------------------- */
int tdr_ray_closest_point_points(tdr_ctx* ctx, const double* pts, int n_cam_pts, int64_t n_sys, double* out_xyz,
                                 double* out_err, int64_t* first_bad);
int tdr_ray_closest_point_points_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                        int64_t n_sys, double* d_xyz, double* d_err, void* stream);
/* ... with the input as one pointer per camera row (see tdr_matrix_triangulate_points_rows) */
int tdr_ray_closest_point_points_rows(tdr_ctx* ctx, const double* const* rows, int n_cam_pts, int64_t n_sys,
                                      double* out_xyz, double* out_err, int64_t* first_bad);

/* How the ray path differentiates (per context; batch entry points tdr_ray_triangulate_points*):
 *   0 (default)  the exact gradient of the point-ray distance (the reference's central difference only for rays closer
 *                than 0.1 mm): the accurate evaluation of the reference's Jacobian, at full speed;
 *   1            RayClosestPoint::compute LITERALLY (src/RayTriangulator.cpp:19-44): six distance evaluations at the
 *                shifted points with the reference's roundings, S from the square-rooted residuals, Jacobi eigen form
 *                on rank-deficient solves — 2.1-4.3 x the time.  The reference's finite difference carries ~2.5e-9 of
 *                rounding noise per Jacobian entry, and that noise decides where its creeping two-ray runs stop: with
 *                this setting the status words (TDR_LM_*) have the reference's statistics (fair against the CPU oracle
 *                on 38.4 M two-ray systems; the default misses by 6.8e-5 per two-ray system).
 * Either way every system whose status is 0 ends at the same point to 2 * TDR_LM_CONVERGED_REL.  The single-system
 * entry point always uses 0.  (The process-wide knob TDR_LM_LITERAL_FD of tdr_debug_set_knob overrides this.) */
int tdr_set_ray_jacobian(tdr_ctx* ctx, int literal);

/* Kernel variant selection for the matrix path (A/B and profiling; 0 = default). */
int tdr_set_matrix_variant(tdr_ctx* ctx, int variant);
/* How the host-buffer batch entry points move a large batch (one that does not fit the small-batch pinned
 * block): mode 0 = serial per 16 M-system chunk (upload, kernel, download, synchronise), 1 = chunks pipelined
 * with copies straight from / to the caller's memory (uploads and downloads issued by two host threads),
 * 2 = pipelined through pinned staging slots filled / emptied by worker threads, true DMA in both directions,
 * 3 = mode 2 with narrow transport: the staging workers convert the detections — integer pixel coordinates in
 * the reference's data (std::stoi, src/DetectionsContainer.cpp:36,57-60) — to int16 pairs on the way, 4 B instead
 * of 16 B per observation cross PCIe; a matrix chunk is solved by the int16 kernel, an LM chunk is unpacked to
 * doubles on the device in front of the kernel, results bit-identical either way; a chunk holding a value that
 * is not an integer in [-32768, 32767] travels as doubles.  LM chunks alternate between two compute streams (one
 * chunk's drain beside the next chunk's start).  -1 = default (TDR_HOST_PIPE, else 3).  chunk_systems = systems
 * per pipeline chunk (0 = default: TDR_HOST_CHUNK, else as many as make a chunk's detection rows 64 MiB — 2^19 at
 * 8 cameras — and at most 131072 on the LM path).
 * Results are bit-identical in every mode. */
int tdr_set_host_pipeline(tdr_ctx* ctx, int mode, int64_t chunk_systems);

/* ------------------------------------------------------------------------------------------
 * Batched candidate scoring — replaces the per-candidate triangulatePoint calls inside
 * DroneClassifier::fillCombinationQueue (src/DroneClassifier.cpp:156-198) for ALL frames at once
 * (SURVEY.md §8f "next" row 1).
 *
 * A combination assigns every camera c a digit: 0 = no detection, k = detection k-1 of that camera
 * in that frame.  The reference enumerates combinations as a prefix tree over cameras, triangulates
 * every node with >= 2 chosen detections and cuts the subtree of a node whose error exceeds
 * error_max (MAX_ERROR_MATRIX = 1e5, src/DroneClassifier.h:11); a full combination that survives is
 * pushed into the priority queue.  A full combination survives iff each of its camera prefixes
 * with >= 2 detections has err <= error_max — this entry point evaluates exactly that for every
 * full combination of every frame and returns the survivors, per frame in the reference's push
 * order (ascending `leaf`).
 *
 *   det_count [n_frames][n_cam]   detections per camera per frame (DetectionsContainer::detCountForCam)
 *   dets      [total][2]          all detections, frame-major, camera-major inside a frame, in
 *                                 detection order (DetectionsContainer::getRecord(cam, frame, det))
 *   out       [cap]               survivors; *n_out = their total number.  If *n_out > cap only the
 *                                 first cap (in (frame, leaf) order of the ones that fitted) are valid and
 *                                 TDR_E_ARG is returned: call again with cap >= *n_out.
 * tdr_candidate.leaf is the combination's index in its frame's tree with camera 0 most significant:
 *   leaf = sum_c digit_c * prod_{c' > c} (det_count[c'] + 1);  tdr_decode_combination() inverts it.
 * Uses min(n_cam of the context) cameras = n_cam.  Matrix triangulator only (the reference's default).
 * ---------------------------------------------------------------------------------------- */
typedef struct tdr_candidate {
  int32_t frame;
  int16_t n_used; /* cameras with a detection in this combination */
  int16_t flags;  /* TDR_CANDIDATE_* (ray triangulator only; 0 for the matrix triangulator) */
  int64_t leaf;
  double x, y, z; /* triangulated point of the full combination */
  double err;     /* its residual error (src/MatrixTriangulator.cpp:55-59) */
} tdr_candidate;

/* Ray triangulator: the walk prunes on the LM solver's `err`, which is a reproducible quantity only where the
 * run ended at the minimiser (TDR_LM_* status 0, see above).  The sweep therefore tracks the status of every node
 * it solves:
 *   TDR_CANDIDATE_FLAGGED      (survivors) the solve of this combination or of one of its camera prefixes had a
 *                              non-zero status: another implementation of the reference may have cut it
 *   TDR_CANDIDATE_FLAGGED_CUT  (records of tdr_ray_score_combinations_ex's `cuts` array) a prefix node that was cut
 *                              on such a solve: another implementation may have kept its subtree.  The record's
 *                              flags >> 8 is the node's depth d (cameras 0..d are decided), `leaf` its index over
 *                              those cameras (camera d least significant), n_used / x / y / z / err the node's own.
 * Below no flagged node the survivors of any two faithful implementations are the same set
 * (tests/test_gpu_score_combinations.py::test_ray_sweep_parity_is_exact_outside_flagged_subtrees). */
#define TDR_CANDIDATE_FLAGGED 1
#define TDR_CANDIDATE_FLAGGED_CUT 2

int tdr_matrix_score_combinations(tdr_ctx* ctx, int n_frames, const int32_t* det_count,
                                  const double* dets, double error_max, int min_cameras,
                                  tdr_candidate* out, int64_t cap, int64_t* n_out);
/* Same contract for the ray triangulator (error_max = MAX_ERROR_RAY = 120, src/DroneClassifier.h:12).  An
 * LM solve per tree node is too expensive for the brute-force leaf kernel, so this entry point sweeps the
 * tree breadth-first: per camera depth, every node the reference would triangulate goes into ONE batch of
 * the LM kernel and dead nodes are dropped before the next depth.  tdr_score_combinations_levelwise runs
 * the same sweep for either triangulator (path 0 = matrix, 1 = ray). */
int tdr_ray_score_combinations(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                               double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                               int64_t* n_out);
int tdr_score_combinations_levelwise(tdr_ctx* ctx, int path, int n_frames, const int32_t* det_count,
                                     const double* dets, double error_max, int min_cameras,
                                     tdr_candidate* out, int64_t cap, int64_t* n_out);
/* tdr_ray_score_combinations plus the list of prefix nodes cut on a flagged LM solve (unordered; *n_cuts may
 * exceed cuts_cap, then only the first cuts_cap are stored — no error). */
int tdr_ray_score_combinations_ex(tdr_ctx* ctx, int n_frames, const int32_t* det_count, const double* dets,
                                  double error_max, int min_cameras, tdr_candidate* out, int64_t cap,
                                  int64_t* n_out, tdr_candidate* cuts, int64_t cuts_cap, int64_t* n_cuts);
/* A scoring call whose survivors did not fit `cap` returns TDR_E_ARG with *n_out = the number needed.  When that
 * call went through the breadth-first sweep (every ray call, matrix calls above the leaf-kernel limit), the complete
 * result is still on the device: this fetches it into a buffer of the right size instead of repeating the sweep.
 * Valid until the next scoring call on the context; TDR_E_ARG when there is nothing to fetch (the caller then
 * repeats the scoring call, as before) or when `cap` is again too small (*n_out = the number needed). */
int tdr_fetch_scored_combinations(tdr_ctx* ctx, tdr_candidate* out, int64_t cap, int64_t* n_out);
/* digits[c] for c in [0, n_cam): 0 = no detection, k = detection k-1 */
int tdr_decode_combination(int n_cam, const int32_t* det_count_frame, int64_t leaf, int32_t* digits);

/* ------------------------------------------------------------------------------------------
 * Static ray helper — replaces Triangulator::getDistFromRay (src/Triangulator.cpp:57-61),
 * batched: dist[i] = | dir(cam, uv[i]) x (point[i] - origin(cam)) |.
 *   d_cam_idx [n] int32 camera ordinal in the context, d_uv [n][2], d_point [n][3], d_dist [n]
 * ---------------------------------------------------------------------------------------- */
int tdr_dist_from_ray_device(tdr_ctx* ctx, const int32_t* d_cam_idx, const double* d_uv,
                             const double* d_point, int64_t n, double* d_dist, void* stream);
/* The same on host buffers (the per-frame "which detections lie within MAX_STEP of the path's last position"
 * test of DroneClassifier::triangulateWithLastPos, src/DroneClassifier.cpp:224-236): inputs and result go
 * through the context's pinned staging block — one H2D, one launch, one D2H, one synchronisation. */
int tdr_dist_from_ray_batch(tdr_ctx* ctx, const int32_t* cam_idx, const double* uv, const double* point,
                            int64_t n, double* out_dist);
/* The same for ONE observation with explicit camera constants (cam_aux[10], any context): what the static
 * Triangulator::getDistFromRay(CamPointPair, Point3d) binds to.  Synchronous, one tiny launch. */
int tdr_dist_from_ray(tdr_ctx* ctx, const double* cam_aux, const double uv[2], const double point[3],
                      double* out_dist);

/* ------------------------------------------------------------------------------------------
 * The PROTECTED static helpers of the plugin surface (src/Triangulator.h:27-42, src/Triangulator.cpp:3-55) — what a
 * subclass written the way the reference's own RayTriangulator::RayClosestPoint is (src/RayTriangulator.cpp:21-44)
 * calls per ray on the host: createRayForPoint, calculateRayDirectionForPixel, rotatePointByQuaternion, distToRay /
 * distToRay_.  Host-side scalar functions — no context, no GPU, nothing batched: they are the reference's scalar
 * helpers, evaluated with the very functions the kernels use (csrc/tdr_math.h: make_ray, dist_to_ray), not a compute
 * path (every triangulatePoint(s) entry point above runs on the device and only there).
 *   cam_aux  [10] rquat (w first, as stored), tvec, fovy, width, height
 *   tdr_create_ray                  origin = tvec, dir = rquat (0, pixel direction) conj(rquat)     (:46-55)
 *   tdr_ray_direction_for_pixel     normalised camera-frame direction of the pixel CENTRE (+0.5)   (:27-44)
 *   tdr_rotate_point_by_quaternion  Hamilton product with the raw quaternion                       (:15-25)
 *   tdr_dist_to_ray                 |dir x (point - origin)|, or its square                        (:3-13)
 * Return TDR_E_ARG on a NULL pointer (tdr_dist_to_ray: NaN).
 * ---------------------------------------------------------------------------------------- */
int tdr_create_ray(const double* cam_aux, const double uv[2], double origin[3], double dir[3]);
int tdr_ray_direction_for_pixel(const double* cam_aux, const double uv[2], double dir_cam[3]);
int tdr_rotate_point_by_quaternion(const double quat[4], const double point[3], double out[3]);
double tdr_dist_to_ray(const double origin[3], const double dir[3], const double point[3], int squared);

/* ------------------------------------------------------------------------------------------
 * Synthetic detection streams generated ON DEVICE: a smooth bounded path per
 * drone projected through the context's P matrices, Gaussian pixel noise, rounded to
 * integer-valued doubles like the real CSV data, observation drop-out with every system
 * keeping >= 2 valid observations.
 *   d_pts   device [n_cam][cam_stride][2] (written for systems [0,n_sys))
 *   d_truth device [n_sys][3] or NULL: the noise-free 3D point
 *   sys_offset  global index of local system 0 (so shards of one stream are consistent)
 *   ray_model != 0: project with the ray path's pinhole model (pixel-centre offset, fovy focal,
 *   src/Triangulator.cpp:27-44) instead of P — the two models are not mutually exact.
 * ---------------------------------------------------------------------------------------- */
int tdr_synth_detections_device(tdr_ctx* ctx, uint64_t seed, int n_drones, int64_t sys_offset,
                                int64_t n_sys, int64_t cam_stride, double noise_px,
                                double dropout_p, int round_to_int, int ray_model, double* d_pts,
                                double* d_truth, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device memory helpers so hosts without a HIP binding (ctypes, cgo) can drive the device
 * entry points.  Thin wrappers over hipMalloc/hipFree/hipMemcpy/hipStreamSynchronize on the
 * context's device.
 * ---------------------------------------------------------------------------------------- */
int tdr_device_alloc(tdr_ctx* ctx, int64_t bytes, void** d_ptr);
int tdr_device_free(tdr_ctx* ctx, void* d_ptr);
int tdr_memcpy_h2d(tdr_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes);
int tdr_memcpy_d2h(tdr_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes);
int tdr_stream_synchronize(tdr_ctx* ctx, void* stream);
/* Page-locked host memory for callers that assemble their detections themselves (the C++ adapter's
 * DetectionsContainer parses the reference's CSV files straight into such a block,
 * src/DetectionsContainer.cpp:19-92): the host-buffer entry points then upload by plain DMA, with no staging
 * copy and no first-touch pinning.  Thin wrappers over hipHostMalloc / hipHostFree; no context needed. */
int tdr_host_alloc(int64_t bytes, void** h_ptr);
int tdr_host_free(void* h_ptr);

/* ------------------------------------------------------------------------------------------
 * Narrow observation format (the separate format row SURVEY.md §8d allows).  The reference's detections are
 * integer pixel coordinates (std::stoi, src/DetectionsContainer.cpp:36,57-60), so an observation fits two
 * int16: pts16[cam][sys] x (int16 x, int16 y), 4 B instead of 16 B, (-1,-1) (either coordinate -1) still the
 * missing sentinel.  Arithmetic stays fp64 — the kernel converts in registers and then runs the same code —
 * so results are bit-identical to the double entry points on the same values.  Algorithmic bytes per system:
 * n_cam x 4 + 24 (56 B at 8 cameras).
 *   tdr_pack_detections_i16_device: double [cam][cam_stride] -> int16 [cam][out_stride]; TDR_E_ARG (and
 *     tdr_last_error) if any coordinate is not an integer in [-32768, 32767].  Synchronises `stream`.
 * ---------------------------------------------------------------------------------------- */
/* Host-side packer (no context, no GPU): n_obs (x, y) double pairs -> n_obs int16 pairs, vectorised (AVX2 where
 * the CPU has it, SSE2 otherwise); *all_exact = 1 iff every coordinate is an integer in [-32768, 32767] (only then
 * is `out` a faithful copy).  This is the conversion the pipelined host-buffer path applies on the fly
 * (tdr_set_host_pipeline mode 3). */
int tdr_pack_detections_i16_host(const double* pts, int64_t n_obs, int16_t* out, int* all_exact);
int tdr_pack_detections_i16_device(tdr_ctx* ctx, const double* d_pts, int n_cam_pts, int64_t cam_stride,
                                   int64_t n_sys, int16_t* d_out, int64_t out_stride, void* stream);
int tdr_matrix_triangulate_points_i16_device(tdr_ctx* ctx, const int16_t* d_pts16, int n_cam_pts,
                                             int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                                             void* stream);
int tdr_matrix_triangulate_points_i16(tdr_ctx* ctx, const int16_t* pts16, int n_cam_pts, int64_t n_sys,
                                      double* out_xyz, double* out_err, int64_t* first_bad);

/* Time `iters` back-to-back launches of a device entry point between two HIP events on the
 * launch stream (after `warmup` untimed launches).  path: 0 = matrix, 1 = ray (LM), 3 = ray closed form, 2 = matrix on the int16
 * format (d_pts then points to int16 pairs).
 * *ms_per_launch = elapsed / iters. */
int tdr_time_launches(tdr_ctx* ctx, int path, const double* d_pts, int n_cam_pts,
                      int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err, int warmup,
                      int iters, void* stream, double* ms_per_launch);

/* Calibration, not part of the path: sustained fp64 FMA rate with `waves_per_simd` resident waves per SIMD
 * (16 independent chains per lane, `iters` rounds).  *tflops = achieved vector fp64 TFLOP/s,
 * *ns_per_wave_inst = time one SIMD needs per wave-level v_fma_f64 — the unit the LM kernel's instruction
 * counts are priced with. */
int tdr_time_fp64_fma(tdr_ctx* ctx, int waves_per_simd, int iters, double* tflops, double* ns_per_wave_inst);

/* Calibration, not part of the path: times a plain HBM stream of the matrix kernel's access shape on
 * this device so the roofline fraction can be read against a reference measured on the same box.
 *   mode 0: `streams` (4 | 8 | 16) concurrent read rows of n_lanes x 16 B, nontemporal, + 8 B written per
 *           lane (d_src needs streams*n_lanes*16 B, d_dst n_lanes*8 B);  bytes = n_lanes*(16*streams+8)
 *   mode 1: nontemporal 16-B/lane copy (d_src, d_dst: n_lanes*16 B);    bytes = n_lanes*32
 *   mode 2: the matrix kernel's READ side alone: `streams` rows of n_lanes x 16 B in 64-thread blocks, nothing
 *           written (d_dst: 8 B);                                        bytes = n_lanes*16*streams
 *   mode 3: the matrix kernel's memory traffic without its arithmetic ("traffic twin"): mode 2's reads + three
 *           nontemporal 8-B stores per lane (d_dst: n_lanes*24 B);       bytes = n_lanes*(16*streams+24) */
int tdr_time_stream(tdr_ctx* ctx, int mode, int streams, const double* d_src, double* d_dst,
                    int64_t n_lanes, int warmup, int iters, double* ms_per_launch);

/* Test / A-B switch, not part of the path: the execution-form and tuning knobs (TDR_LM_COOP, TDR_LM_COOP_MAX,
 * TDR_LM_FULL, TDR_LM_CHUNK, TDR_LM_REFILL, TDR_LM_NLDS, TDR_I16_X2, TDR_SWEEP_HOST, TDR_POINT_ARGS) take their start value from the
 * environment variable of the same name, read ONCE per process; this call is the only way to change one afterwards
 * (the launchers never call getenv).  unset != 0 restores the built-in default.  Every setting of these computes the
 * same results (tests/test_gpu_parity.py::test_ray_kernel_forms_agree_bitwise, tests/test_gpu_lm_builds.py).
 * One knob is a different BUILD of the LM kernels, not another execution form: TDR_LM_LITERAL_FD = 1 takes
 * RayClosestPoint::compute literally for every ray (src/RayTriangulator.cpp:19-44: the central-difference Jacobian from
 * six distance evaluations at the shifted points with the reference's non-FMA roundings, S from the square-rooted
 * residuals, Jacobi eigen form on rank-deficient solves) where the default kernels evaluate the exact gradient — a
 * diagnostic (2.1-4.3 x the time) that reproduces the reference's status-word statistics on two-ray systems, which the
 * default kernels miss by 6.8e-5 per system; results agree with the default build's where the LM status
 * is 0, not bit for bit.
 * Returns TDR_OK or TDR_E_ARG for an unknown name.  Not to be called while launches are in flight on other threads. */
int tdr_debug_set_knob(const char* name, int value, int unset);

/* Test hook, not part of the path: the device restatement of the Jacobi sweeps of cv::solve / cv::invert with
 * DECOMP_EIG, which the LM loop used in round 3 on (nearly) rank-deficient J^T J and which remains as the device-side
 * yardstick of tdr_debug_lm_cascade_device (src/RayTriangulator.cpp:100-104 -> cv::LMSolver::run) on n symmetric 3x3
 * matrices, one per lane.  d_in [n][9] = {m00, m01, m02, m11, m12, m22, v0, v1, v2};
 * d_out [n][4] = {pinv(M) v (3), max_k |pinv(M)_kk|}.  Enqueued on `stream` (NULL: the context's stream). */
int tdr_debug_sym3_eig_device(tdr_ctx* ctx, const double* d_in, int64_t n, double* d_out, void* stream);

/* Test hook, not part of the path: what the LM loop evaluates INSTEAD of that eigen solve since round 4 — the same
 * pseudo-inverse by rank, the rank decided on the matrix's invariants at the reference's 2 eps sum|w| cut, each rank in
 * closed form (lm_cascade in csrc/tdr_kernels.hip; measured against the Jacobi form on the CPU oracle before it was
 * built: profiles/r04_lm_hybrid_rule.json).  d_in as above, as a system of n_rays (>= 2) rays meets it (the rank-2
 * solution is cleared of its null component only when there are more than two rays);
 * d_out [n][5] = {pinv(M) v (3), max_k |pinv(M)_kk|, the rank decided (0..3)}. */
int tdr_debug_lm_cascade_device(tdr_ctx* ctx, const double* d_in, int64_t n, int n_rays, double* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDR_AMD_H */
