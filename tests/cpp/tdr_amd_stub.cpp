// A stand-in for libtdr_amd.so's C ABI, for the CPU sanitizer tests of the C++ adapter headers only
// (tests/test_adapter_cpu.py).  No device, no arithmetic: a context is a heap block whose `alive` word every call
// reads, a call takes a little while, tdr_destroy scribbles over the block and frees it — so a context destroyed
// under a call in flight is a heap-use-after-free for AddressSanitizer and a data race for ThreadSanitizer.
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

#include "tdr_amd.h"

struct tdr_ctx {
  std::mutex mu;  // the real context serialises its calls the same way
  int alive = 1;
  int calls = 0;
};

std::atomic<long> tdr_stub_created{0};  // contexts handed out by tdr_create (adapter_registry_check.cpp)

extern "C" {
// what the last tdr_create / point call / tdr_dist_from_ray was handed (tests/cpp/camera_access_check.cpp)
double tdr_stub_P[64 * 12];
double tdr_stub_aux[64 * TDR_CAM_AUX_DOUBLES];
int tdr_stub_n = 0;
static std::mutex g_capture;
static void capture(const double* P, const double* aux, int n) {
  std::lock_guard<std::mutex> lock(g_capture);
  tdr_stub_n = n < 64 ? n : 64;
  if (P) std::memcpy(tdr_stub_P, P, sizeof(double) * 12 * tdr_stub_n);
  if (aux) std::memcpy(tdr_stub_aux, aux, sizeof(double) * TDR_CAM_AUX_DOUBLES * tdr_stub_n);
}
const char* tdr_strerror(int) { return "stub"; }
const char* tdr_last_error(void) { return ""; }
int tdr_camera_derive(int id, int width, int height, double focal_length, const double tvec[3],
                      const double rquat[4], tdr_camera* out) {
  std::memset(out, 0, sizeof(*out));
  out->id = id; out->width = width; out->height = height; out->fx = out->fy = focal_length;
  out->cx = width / 2; out->cy = height / 2; out->fovx = out->fovy = 50.0;
  for (int i = 0; i < 3; ++i) out->tvec[i] = tvec[i];
  for (int i = 0; i < 4; ++i) out->rquat[i] = rquat[i];
  out->P[0] = out->P[5] = out->P[10] = 1.0;
  return TDR_OK;
}
int tdr_create(const double* P, const double* aux, int n_cam, int, tdr_ctx** out) {
  capture(P, aux, n_cam);
  *out = new tdr_ctx;
  tdr_stub_created.fetch_add(1);
  return TDR_OK;
}
void tdr_destroy(tdr_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->alive = 0;
  }
  delete ctx;
}
static int touch(tdr_ctx* ctx) {
  if (!ctx) return TDR_E_ARG;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (ctx->alive != 1) return TDR_E_ARG;
  ctx->calls++;
  std::this_thread::sleep_for(std::chrono::microseconds(50));
  return ctx->alive == 1 ? TDR_OK : TDR_E_ARG;
}
int tdr_dist_from_ray(tdr_ctx* ctx, const double* aux, const double*, const double* point, double* dist) {
  capture(nullptr, aux, 1);
  *dist = point[0];
  return touch(ctx);
}
int tdr_matrix_triangulate_point(tdr_ctx* ctx, const double* P, const double* uv, int n, double* X, double* err) {
  capture(P, nullptr, n);
  X[0] = uv[0]; X[1] = uv[1]; X[2] = 0; *err = 0;
  return touch(ctx);
}
int tdr_ray_triangulate_point(tdr_ctx* ctx, const double* aux, const double* uv, int n, double* X, double* err) {
  capture(nullptr, aux, n);
  X[0] = uv[0]; X[1] = uv[1]; X[2] = 0; *err = 0;
  return touch(ctx);
}
static int fill(tdr_ctx* ctx, int64_t n, double* out, int64_t* first_bad) {
  for (int64_t i = 0; i < 3 * n; ++i) out[i] = (double)i;
  if (first_bad) *first_bad = -1;
  return touch(ctx);
}
int tdr_matrix_triangulate_points(tdr_ctx* c, const double*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_ray_triangulate_points(tdr_ctx* c, const double*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_ray_triangulate_points_ex(tdr_ctx* c, const double*, int, int64_t n, double* o, double*, int32_t*, int32_t*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_matrix_triangulate_points_rows(tdr_ctx* c, const double* const*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_ray_triangulate_points_rows(tdr_ctx* c, const double* const*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_matrix_triangulate_points_i16(tdr_ctx* c, const int16_t*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
int tdr_set_ray_jacobian(tdr_ctx* c, int) { return touch(c); }
int tdr_ray_closest_point_points_rows(tdr_ctx* c, const double* const*, int, int64_t n, double* o, double*, int64_t* fb) { return fill(c, n, o, fb); }
// the protected static helpers' exports: host-side in the real library too; here recognisable numbers, no arithmetic
int tdr_create_ray(const double* aux, const double uv[2], double origin[3], double dir[3]) {
  capture(nullptr, aux, 1);
  for (int i = 0; i < 3; ++i) origin[i] = aux[4 + i];
  dir[0] = uv[0]; dir[1] = uv[1]; dir[2] = 1.0;
  return TDR_OK;
}
int tdr_ray_direction_for_pixel(const double* aux, const double uv[2], double d[3]) {
  capture(nullptr, aux, 1);
  d[0] = uv[0] + 0.5; d[1] = uv[1] + 0.5; d[2] = aux[7];
  return TDR_OK;
}
int tdr_rotate_point_by_quaternion(const double q[4], const double p[3], double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = q[0] * p[i] + q[i + 1];
  return TDR_OK;
}
double tdr_dist_to_ray(const double o[3], const double d[3], const double p[3], int squared) {
  const double v = (p[0] - o[0]) + 10 * d[0];
  return squared ? v * v : v;
}
}
