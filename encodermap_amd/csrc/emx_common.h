// Shared host/device helpers of libemx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "emx.h"

namespace emx {

// thread-local text of the last rejected call (emx_last_error)
char* error_buffer();

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EMX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EMX_OK;
}

#define EMX_REQUIRE(cond, ...)                             \
  do {                                                     \
    if (!(cond)) return emx::fail(EMX_EINVAL, __VA_ARGS__); \
  } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blocks of NT threads (multiple of 64); result valid in thread 0.
// Fixed reduction order -> deterministic.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* smem /* >= NT/64 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}


}  // namespace emx
