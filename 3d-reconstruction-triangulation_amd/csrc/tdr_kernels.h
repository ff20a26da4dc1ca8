// tdr_kernels.h — launchers of the gfx950 kernels (internal; the public surface is
// include/tdr_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <climits>

#include "tdr_math.h"

namespace tdr_amd {

constexpr int kMaxCam = 32;
constexpr int kLmCounters = 64;
// words per LM launch in the counter ring: word 0 is the head of the launch's work queue, the rest pads the record to
// 32 B (round 3 kept the count of parked systems and the second launch's queue head there)
constexpr int kLmCounterWords = 4;

// Camera constants travel in the kernarg segment (scalar loads, no per-lane traffic):
// 32 x 96 B = 3 KiB for P, 32 x 88 B for the ray model.
struct CamSetP {
  double p[kMaxCam][12];
};
struct CamSetRay {
  RayCam c[kMaxCam];
};

// Per-camera constants of the closed-form ray kernel (ray_closest_point_kernel): the ray of pixel (u, v) before
// normalisation is e = u A + v B + C (make_ray()'s pinhole + quaternion rotation folded into one affine map),
// |direction|^2 = D = |q|^4 for every pixel, origin o.  128 B, read through scalar loads.
struct CfCam {
  double ax, ay, az, bx, by, bz, cx, cy, cz;
  double ox, oy, oz;
  double D, pad[3];
};
static_assert(sizeof(CfCam) == 128, "CfCam is one 128-B record");

struct DeviceCams {  // host-side copy held by the context
  int n_cam;
  CamSetP P;
  CamSetRay ray;
  bool has_ray;
  const double* d_P;    // device copy [kMaxCam][12]
  const RayCam* d_ray;  // device copy [kMaxCam]
  const CfCam* d_cf;    // device copy [kMaxCam] (closed-form ray kernel; valid when has_ray)
  // LM kernel's work-queue heads: a ring of kLmCounters records of kLmCounterWords words, one per launch in flight
  // (zeroed on the launch stream right before the kernel), so batches enqueued on different streams never share one
  unsigned long long* d_work_counter;
  mutable std::atomic<unsigned> lm_launches{0};
  unsigned long long* d_first_bad_scratch; // sink for the single-system LM launch
  // tdr_set_ray_jacobian: 1 = this context's LM batches run the literal build (tdr_lm_literal.hip); the process-wide
  // knob TDR_LM_LITERAL_FD, when set, overrides it either way
  std::atomic<int> lm_literal{0};
};

// variant: 0 = default (nontemporal loads/stores, 8-B result stores, block size by camera count),
//          1 = default cache policy, 2 = LDS-staged 16-B result stores, 3/4 = 64/128-thread blocks
hipError_t launch_matrix(const DeviceCams& cams, int variant, const double* d_pts, int n_cam_used,
                         int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                         unsigned long long* d_first_bad, hipStream_t stream);

hipError_t launch_matrix_i16(const DeviceCams& cams, const void* d_pts16, int n_cam_used, int64_t cam_stride,
                             int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* d_first_bad,
                             hipStream_t stream);
hipError_t launch_pack_i16(const double* d_pts, int n_cam, int64_t cam_stride, int64_t n_sys, void* d_out,
                           int64_t out_stride, unsigned long long* d_n_bad, hipStream_t stream);
// the inverse: int16 pairs [n_cam][in_stride] -> double rows [n_cam][cam_stride] (exact)
hipError_t launch_unpack_i16(const void* d_in, int n_cam, int64_t in_stride, int64_t n_sys, double* d_pts,
                             int64_t cam_stride, hipStream_t stream);
hipError_t launch_fma_peak(double* d_out, int blocks, int iters, hipStream_t stream);
// test hook: round 3's Jacobi restatement of DECOMP_EIG on n matrices (in [n][9], out [n][4]; tdr_debug_sym3_eig_device)
hipError_t launch_eig3_debug(const double* d_in, int64_t n, double* d_out, hipStream_t stream);
// test hook: the LM loop's eigen-free DECOMP_EIG restatement (in [n][9], out [n][5]; tdr_debug_lm_cascade_device)
hipError_t launch_cascade_debug(const double* d_in, int64_t n, int n_rays, double* d_out, hipStream_t stream);
// d_iters / d_status: optional per-system outputs (iteration count of the reference's loop; TDR_LM_* bits — the
// convergence certificate is a second small kernel on the same stream, launched only when d_status != nullptr)
hipError_t launch_ray(const DeviceCams& cams, const double* d_pts, int n_cam_used,
                      int64_t cam_stride, int64_t n_sys, double* d_xyz, double* d_err,
                      int* d_iters, int* d_status, unsigned long long* d_first_bad, hipStream_t stream);

// closed-form minimiser of the ray path's objective (tdr_ray_closest_point_points_device), one streaming pass
hipError_t launch_ray_closest_point(const DeviceCams& cams, const double* d_pts, int n_cam_used, int64_t cam_stride,
                                    int64_t n_sys, double* d_xyz, double* d_err, unsigned long long* d_first_bad,
                                    hipStream_t stream);

// one system, per-observation constants in device memory; out4 = {x, y, z, err}
hipError_t launch_matrix_point(const double* d_Pobs, const double* d_uv, int n_obs, double* d_out4,
                               hipStream_t stream);
hipError_t launch_ray_point(const DeviceCams& cams, const double* d_auxobs, const double* d_uv,
                            int n_obs, double* d_out4, hipStream_t stream);
// <= 16 observations: the camera-per-lane kernel with the constants in the kernel argument (tdr_lm_kernels.hip)
hipError_t launch_ray_point_coop(const DeviceCams& cams, const RayCam* rc_host, const double* d_uv, int n_obs,
                                 double* d_out4, hipStream_t stream);

// batched DroneClassifier::fillCombinationQueue scoring; d_out = Candidate[cap] (48 B records)
hipError_t launch_score_leaves(const DeviceCams& cams, int n_frames, const long long* d_frame_leaf_offset,
                               const int* d_det_count, const long long* d_det_offset,
                               const double* d_dets, double error_max, int min_cameras,
                               long long n_leaves, void* d_out, unsigned long long* d_keys,
                               long long cap, unsigned long long* d_n_out, hipStream_t stream);
// device radix sort of the appended candidates by key (hipCUB two-call protocol)
hipError_t sort_candidates(void* d_temp, size_t& temp_bytes, const unsigned long long* d_keys_in,
                           unsigned long long* d_keys_out, const void* d_in, void* d_out, long long n,
                           hipStream_t stream);

// device tree sweep of the combination tree (tdr_kernels.hip); node records are tdr_candidate (48 B); every
// flag / position array has one element more than the list it describes (the scan total lands there)
hipError_t sweep_scan(void* d_temp, size_t& temp_bytes, const long long* d_in, long long* d_out, long long n,
                      hipStream_t stream);
hipError_t launch_sweep_roots(void* d_nodes, int n_frames, hipStream_t st);
hipError_t launch_sweep_sizes(const void* d_nodes, long long n, const int* d_det_count, int n_cam, int L,
                              long long* d_sizes, hipStream_t st);
hipError_t launch_sweep_expand(const void* d_parents, long long n, const long long* d_offsets, const int* d_det_count,
                               int n_cam, int L, void* d_children, long long* d_marks, int* d_overflow,
                               hipStream_t st);
hipError_t launch_sweep_gather(const void* d_children, long long n_children, const long long* d_slot,
                               const int* d_det_count, const long long* d_det_offset, const double* d_dets, int n_cam,
                               int L, long long stride, double* d_pts, hipStream_t st);
hipError_t launch_sweep_prune(void* d_children, long long n_children, const long long* d_slot, const double* d_xyz,
                              const double* d_err, double error_max, const int* d_lm_status, int depth, void* d_cuts,
                              long long cuts_cap, unsigned long long* d_n_cuts, long long* d_alive, hipStream_t st);
hipError_t launch_sweep_final(const void* d_nodes, long long n, int min_cameras, long long* d_keep, hipStream_t st);
hipError_t launch_sweep_compact(const void* d_in, long long n, const long long* d_pos, void* d_out, hipStream_t st);

hipError_t launch_dist_from_ray(const DeviceCams& cams, const int32_t* d_cam_idx,
                                const double* d_uv, const double* d_point, int64_t n,
                                double* d_dist, hipStream_t stream);

// seq_word != nullptr: d_out is a host-coherent pinned slot and `seq` is stored behind the result (system scope)
hipError_t launch_dist_from_ray_point(const RayCam& cam, double u, double v, const double* point, double* d_out,
                                      unsigned long long* seq_word, unsigned long long seq, hipStream_t stream);

// one matrix system with its constants and pixels in the kernel argument, result + sequence word into pinned host
// memory (tdr_matrix_triangulate_point; tdr_kernels.hip: matrix_point_arg_kernel)
struct PointArgsHost {
  double P[kMaxCam][12];
  double uv[kMaxCam][2];
};
hipError_t launch_matrix_point_args(const PointArgsHost& args, int n_obs, double* out4, unsigned long long* seq_word,
                                    unsigned long long seq, hipStream_t stream);

hipError_t launch_synth(const DeviceCams& cams, uint64_t seed, int n_drones, int64_t sys_offset,
                        int64_t n_sys, int64_t cam_stride, double noise_px, double dropout_p,
                        int round_to_int, int ray_model, double* d_pts, double* d_truth,
                        hipStream_t stream);

// calibration streams: mode 0 read-only (`streams` 16-B rows of n_lanes + 8 B out per lane), mode 1 copy
hipError_t launch_stream(int mode, int streams, const double* d_src, double* d_dst, int64_t n_lanes,
                         hipStream_t stream);

RayCam make_raycam(const double* aux10);
CfCam make_cfcam(const RayCam& rc);

// ---- tuning knobs (A/B switches and the execution-form switches of the tests).  Read from the environment ONCE, when
// the first launcher needs one; afterwards they change only through knob_set() (C ABI: tdr_debug_set_knob).  Launchers
// run concurrently from several host threads (one per shard / per pipeline stage): a getenv() per launch raced with
// the Python side's os.environ writes and cost host time on the 0.2 ms chunk hand-over path.
enum Knob { K_LM_COOP, K_LM_COOP_MAX, K_LM_FULL, K_LM_CHUNK, K_LM_REFILL, K_I16_X2, K_SWEEP_HOST, K_LM_LITERAL_FD,
            K_LM_NLDS, K_POINT_ARGS, K_MATRIX_PAD_LDS, K_COUNT };
constexpr int kKnobUnset = INT_MIN;
int knob_raw(int k);  // the one table of the process (tdr_kernels.hip); kKnobUnset = not set
inline int knob(Knob k, int dflt, int lo, int hi) {
  int t = knob_raw(k);
  if (t == kKnobUnset) t = dflt;
  return t < lo ? lo : (t > hi ? hi : t);
}

// unset != 0 restores the built-in default.  -1: unknown name.
int knob_set(const char* name, int value, int unset);
bool knob_sweep_on_host();

}  // namespace tdr_amd
