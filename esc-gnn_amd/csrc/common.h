// common.h — shared helpers for the gfx950 kernels of libescgnn_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/escgnn_hip.h"

namespace esc {

constexpr int WAVE = 64;  // CDNA4 wavefront

void set_error(const char* fmt, ...);

#define ESC_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      esc::set_error(__VA_ARGS__);                \
      return ESC_EINVAL;                          \
    }                                             \
  } while (0)

#define ESC_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t err__ = hipGetLastError();                                   \
    if (err__ != hipSuccess) {                                              \
      esc::set_error("%s: launch failed: %s", name, hipGetErrorString(err__)); \
      return ESC_ELAUNCH;                                                   \
    }                                                                       \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// per-launch kernel timing (include/escgnn_hip.h "profiling hook").  When a kernel family is being
// profiled its launches go through hipExtLaunchKernelGGL with a start/stop event pair, which stamps
// the dispatch packet itself — the same begin/end a rocprofv3 kernel trace reports — instead of
// bracketing the launch with stream events (that adds ~3 us of queue latency to a 7 us kernel).
bool prof_slot(int kind, hipEvent_t* start, hipEvent_t* stop);

template <typename... KArgs, typename... Args>
inline void launch(int kind, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s,
                   Args... args) {
  hipEvent_t a = nullptr, b = nullptr;
  if (prof_slot(kind, &a, &b))
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, a, b, 0, static_cast<KArgs>(args)...);
  else
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, static_cast<KArgs>(args)...);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-uniform value into an SGPR so that dependent loads become scalar loads
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace esc
