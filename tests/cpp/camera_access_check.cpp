// Which numbers does the adapter hand to the C ABI for a camera?  Compiled twice by tests/test_adapter_cpu.py — against
// host/Camera.h (std::array members) and, with -DTDR_TEST_MATLIKE and tests/cpp/matlike first on the include path,
// against a camera with cv::Mat-like members (the reference's own src/Camera.h:38-46 kept in place) — and linked with
// the stub ABI, which records what tdr_create / the point entry points / tdr_dist_from_ray received.  Both builds
// must print the same lines.
#include <cstdio>
#include <vector>

#include "MatrixTriangulator.h"
#include "RayTriangulator.h"

extern "C" {
extern double tdr_stub_P[64 * 12];
extern double tdr_stub_aux[64 * TDR_CAM_AUX_DOUBLES];
extern int tdr_stub_n;
}

static tdr::Camera* make(int id) {
  tdr::Camera* cam = new tdr::Camera(id);
  cam->width = 1924 + id;
  cam->height = 1082 - id;
  cam->fovy = 53.25 + id;
#ifdef TDR_TEST_MATLIKE
  cam->cameraPerspectiveMatrix = testmat::Mat(3, 4);
  cam->rquat = testmat::Mat(4, 1);
  cam->tvec = id % 2 ? testmat::Mat(1, 3) : testmat::Mat(3, 1);  // the reference indexes tvec with one index
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) cam->cameraPerspectiveMatrix.at<double>(r, c) = 100.0 * id + 10.0 * r + c + 0.5;
  for (int i = 0; i < 4; ++i) cam->rquat.at<double>(i, 0) = 0.1 * (i + 1) + id;
  for (int i = 0; i < 3; ++i) {
    if (id % 2) cam->tvec.at<double>(0, i) = 1000.0 * (i + 1) + id;
    else cam->tvec.at<double>(i, 0) = 1000.0 * (i + 1) + id;
  }
#else
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) cam->cameraPerspectiveMatrix[4 * r + c] = 100.0 * id + 10.0 * r + c + 0.5;
  for (int i = 0; i < 4; ++i) cam->rquat[i] = 0.1 * (i + 1) + id;
  for (int i = 0; i < 3; ++i) cam->tvec[i] = 1000.0 * (i + 1) + id;
#endif
  return cam;
}

static void dump(const char* what, bool with_P, bool with_aux) {
  std::printf("%s n=%d\n", what, tdr_stub_n);
  for (int i = 0; i < tdr_stub_n; ++i) {
    if (with_P) {
      std::printf(" P");
      for (int k = 0; k < 12; ++k) std::printf(" %.17g", tdr_stub_P[12 * i + k]);
      std::printf("\n");
    }
    if (with_aux) {
      std::printf(" aux");
      for (int k = 0; k < TDR_CAM_AUX_DOUBLES; ++k) std::printf(" %.17g", tdr_stub_aux[TDR_CAM_AUX_DOUBLES * i + k]);
      std::printf("\n");
    }
  }
}

int main() {
  std::vector<const tdr::Camera*> cams = {make(0), make(1), make(2)};
  {
    MatrixTriangulator m(cams);
    dump("create", true, true);
    m.triangulatePoint({{cams[2], {1.0, 2.0}}, {cams[0], {3.0, 4.0}}});
    dump("matrix point", true, false);
    RayTriangulator r(cams);
    r.triangulatePoint({{cams[1], {1.0, 2.0}}, {cams[2], {3.0, 4.0}}});
    dump("ray point", false, true);
    Triangulator::getDistFromRay({cams[1], {5.0, 6.0}}, {1.0, 2.0, 3.0});
    dump("dist", false, true);
  }
  for (const tdr::Camera* c : cams) delete c;
  return 0;
}
