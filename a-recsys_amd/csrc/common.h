// common.h -- shared helpers for libarx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "arx.h"

namespace arx {

void set_error(const char* fmt, ...);

constexpr int kWave = 64;  // CDNA wavefront

#define ARX_CHECK_ARG(cond, ...)                  \
  do {                                            \
    if (!(cond)) {                                \
      ::arx::set_error(__VA_ARGS__);              \
      return ARX_EINVAL;                          \
    }                                             \
  } while (0)

#define ARX_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::arx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                       __FILE__, __LINE__);                                        \
      return ARX_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define ARX_CHECK_LAUNCH()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      ::arx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                        \
      return ARX_EHIP;                                                             \
    }                                                                              \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// lanes of a wave that cooperate on one table row of d floats (float4 per lane)
static inline int lanes_per_row(int d) {
  int q = d / 4;
  int l = 1;
  while (l < q && l < 64) l <<= 1;
  return l;
}

// number of CUs of the current device (cached)
int cu_count();

}  // namespace arx
