// tdr_device_common.h — what every device translation unit of the library shares: wave / block constants, the knob
// table's accessor, the camera-constant carrier of the LM kernels and the reference's sentinel test.  Internal.
//
// The kernels are split over several .hip files only so that they compile in parallel (the LM kernels are 26 large
// instantiations; one file took three minutes): tdr_kernels.hip (matrix path, closed-form ray path, combination
// scoring, streams, test hooks), tdr_lm_kernels.hip (the LM kernels), tdr_lm_literal.hip (their literal
// central-difference build, knob TDR_LM_LITERAL_FD).  Device code lives in anonymous namespaces: no kernel calls
// across files.
#pragma once
#include "tdr_kernels.h"

namespace tdr_amd {

namespace {

typedef double dbl2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr double kDblEps = 2.220446049250313e-16;

template <int N>
struct CamR {
  RayCam c[N];
};

__device__ __forceinline__ bool is_missing(double x, double y) {
  // the reference's sentinel test (src/MatrixTriangulator.cpp:86, src/RayTriangulator.cpp:66)
  return (x == -1.0) || (y == -1.0);
}

}  // namespace
}  // namespace tdr_amd
