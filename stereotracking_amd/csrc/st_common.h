// Shared helpers for libstereotrack_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/stereotrack.h"

namespace st {

std::string& last_error();
int set_error(int code, const char* fmt, ...);

// Raw uint8 input of the fused stem (stem_focus_conv.hip): N separate [3][h][w] frames, padded on the fly to H x W.
struct StemRawInput {
  const unsigned char* const* frames;   // HOST array of N device pointers
  int h, w;
  float pad_value;
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// floats per prior in the head buffer: num_classes class logits, x, y, w, h, obj (+ padding to 16-byte rows); 8 for the
// shipped 1..3-class heads, nc + 5 rounded up to a multiple of 4 beyond (st_head_row_floats)
inline int head_row_floats(int nc) { return nc <= 3 ? 8 : (nc + 5 + 3) / 4 * 4; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace st

#define ST_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return st::set_error(ST_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                           hipGetErrorString(e_), __FILE__, __LINE__);                  \
  } while (0)

#define ST_REQUIRE(cond, ...)                                        \
  do {                                                               \
    if (!(cond)) return st::set_error(ST_ERR_INVALID, __VA_ARGS__);  \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call: raise the limit of a kernel only when a
// launch needs more dynamic LDS than any earlier launch of it did (`cached` = a static int next to the call site).
#define ST_ENSURE_DYNAMIC_LDS(kern, bytes, cached)                                                           \
  do {                                                                                                       \
    if ((int)(bytes) > (cached)) {                                                                           \
      ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));          \
      (cached) = (int)(bytes);                                                                               \
    }                                                                                                        \
  } while (0)

#define ST_CHECK(expr)        \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != ST_OK) return rc_; \
  } while (0)
