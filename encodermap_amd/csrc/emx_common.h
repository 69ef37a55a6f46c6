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

// sin and cos of an angle of moderate size (|x| up to a few thousand: data angles are within a few periods): three-term
// Cody-Waite reduction by pi/2 and the classic degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4], ~1.5 ulp.  The
// library sincosf carries a Payne-Hanek path for huge arguments that costs ~3 KB of scratch per lane in every kernel that
// merely contains it.
__device__ __forceinline__ void sincos_bounded(float x, float* s, float* c) {
  const float q = rintf(x * 0.63661977236758134f);
  float r = fmaf(q, -1.57079601287841796875f, x);
  r = fmaf(q, -3.1391647326017846353e-7f, r);
  r = fmaf(q, -5.3903025299577647655e-15f, r);
  const int n = (int)q;
  const float r2 = r * r;
  const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.95152959e-4f, 8.33216087e-3f), -1.66666546e-1f), r);
  const float cp = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.44331571e-5f, -1.38873163e-3f), 4.16666457e-2f), -0.5f), 1.f);
  const float a = (n & 1) ? cp : sp, b = (n & 1) ? sp : cp;
  *s = (n & 2) ? -a : a;
  *c = ((n + 1) & 2) ? -b : b;
}

}  // namespace emx
