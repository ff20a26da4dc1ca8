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

#include <climits>

namespace tdr_amd {

// ---- tuning knobs (A/B switches and the execution-form switches of the tests).  Read from the environment ONCE, when
// the first launcher needs one; afterwards they change only through knob_set() (C ABI: tdr_debug_set_knob).  Launchers
// run concurrently from several host threads (one per shard / per pipeline stage): a getenv() per launch raced with
// the Python side's os.environ writes and cost host time on the 0.2 ms chunk hand-over path.
enum Knob { K_LM_COOP, K_LM_COOP_MAX, K_LM_FULL, K_LM_CHUNK, K_LM_REFILL, K_I16_X2, K_SWEEP_HOST, K_LM_LITERAL_FD,
            K_LM_NLDS, K_COUNT };
constexpr int kKnobUnset = INT_MIN;
int knob_raw(int k);  // the one table of the process (tdr_kernels.hip); kKnobUnset = not set
inline int knob(Knob k, int dflt, int lo, int hi) {
  int t = knob_raw(k);
  if (t == kKnobUnset) t = dflt;
  return t < lo ? lo : (t > hi ? hi : t);
}

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
