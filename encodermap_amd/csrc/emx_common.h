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
// launches issued so far (emx_launch_count)
long long& launch_counter();

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  ++launch_counter();
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EMX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EMX_OK;
}

// Counter words through which workgroups of ONE launch hand work to one another (tickets of a split reduction, arrival counts, the
// sticky give-up flag): they must be zero before the first launch that uses them, and every launch leaves them at zero again.  The
// library does not trust its caller with the first part: the first time an entry point is handed a pointer -- per process -- it
// zeroes the words on the caller's stream in front of its launch (hipMemsetAsync; later calls cost a hash lookup).  Seen for the
// first time INSIDE a stream capture the memset would become a node of the graph and clear the give-up flag on every replay:
// that is refused (EMX_EINVAL: call once outside the capture first -- a warm-up step does).  A caller that recycles such memory for
// something else and hands it back later must zero it again (emx_forget_counters drops the library's note).
int ensure_counters_zeroed(void* words, size_t bytes, hipStream_t stream, const char* who);
// Ticket words that belong to a caller's SCRATCH buffer (the split reduction of the all-pairs pass): kept in memory the LIBRARY
// owns, one zeroed block per scratch address it has seen -- scratch buffers come and go through caching allocators, and a
// recycled address must not bring stale or foreign bits back as ticket counts; a block of the library's is only ever written by
// launches that leave it at zero.  Allocated at first sight (not inside a stream capture: EMX_EINVAL, as above).
unsigned* owned_tickets(const void* scratch, size_t bytes, hipStream_t stream, const char* who, int* rc);

#define EMX_REQUIRE(cond, ...)                             \
  do {                                                     \
    if (!(cond)) return emx::fail(EMX_EINVAL, __VA_ARGS__); \
  } while (0)

constexpr int WAVE = 64;

// The value of lane ^ 16 / lane ^ 32 beside the lane's own, as the unordered pair (a, b): v_permlane16_swap_b32 /
// v_permlane32_swap_b32 (gfx950) exchange the odd rows (the upper half) of one register with the even rows (the lower half)
// of another -- applied to two copies of x, every lane then holds its own value in one and its partner's in the other.  A VALU
// instruction (no LDS crossbar, no wait) where __shfl_xor is a ds_bpermute_b32; for commutative combinations (sum, max) the
// order inside the pair does not matter.  All lanes must be active.
__device__ __forceinline__ void lane_xor16_pair(unsigned x, unsigned& a, unsigned& b) {
  a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_xor32_pair(unsigned x, unsigned& a, unsigned& b) {
  a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned max_xor16(unsigned x) { unsigned a, b; lane_xor16_pair(x, a, b); return a > b ? a : b; }
__device__ __forceinline__ unsigned max_xor32(unsigned x) { unsigned a, b; lane_xor32_pair(x, a, b); return a > b ? a : b; }
__device__ __forceinline__ float sum_xor16(float x) {
  unsigned a, b;
  lane_xor16_pair(__builtin_bit_cast(unsigned, x), a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float sum_xor32(float x) {
  unsigned a, b;
  lane_xor32_pair(__builtin_bit_cast(unsigned, x), a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// Sum over the 64 lanes of a wave, in every lane; all lanes must be active.  The association is the xor butterfly's
// (1, 2, 4, 8, 16, 32) -- bit for bit what six __shfl_xor steps give -- but none of the steps goes through the LDS crossbar
// (ds_bpermute_b32: ~100 cycles each, and each waits for the one before): within a row of 16 lanes the partner's value comes
// by DPP (quad permutations; then mirrors, which exchange the right groups because all lanes of a group already agree), and
// the four row sums are read into scalar registers.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // lane ^ 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror: quad ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror: group of 8 ^ 1
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);   // (rows ^ 1, then halves ^ 1: every lane of the butterfly ends with this sum, up to commutations)
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

// Two float32 values as the two f16 planes of the split operands, packed (element 0 in the low half): hi = f16(v) (round to nearest
// even), lo = f16(v - hi).  THREE instructions for the pair where the compiler's form of `(_Float16)(v - (float)h)` is six
// (v_cvt_f32_f16 + v_sub_f32 per element and a second v_cvt_pk_f16_f32): v_fma_mixlo / mixhi_f16 read an f16 half as a float32
// source and round the float32 result to f16 once; v - hi is exact in float32 (hi is v rounded to eleven bits), so these are the
// same two roundings, bit for bit (f16 denormals are kept by both forms).  EMX_SPLIT_PLAIN builds the compiler's form (A/B).
__device__ __forceinline__ void split_f16_pair(const float v0, const float v1, unsigned& hi, unsigned& lo) {
#ifdef EMX_SPLIT_PLAIN
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const h2 vh = __builtin_convertvector(f2{v0, v1}, h2);
  const h2 vl = __builtin_convertvector(f2{v0 - (float)vh[0], v1 - (float)vh[1]}, h2);
  hi = __builtin_bit_cast(unsigned, vh), lo = __builtin_bit_cast(unsigned, vl);
#else
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
#endif
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
