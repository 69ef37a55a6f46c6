// Two-plane f16 split of the (shifted, row-scaled) high-dimensional rows: the operand format of the all-pairs kernel's
// Gram product.
//
// The float32 matrix instructions of gfx950 run at the VALU's rate AND hold the SIMD's vector issue, so a tile's f32 Gram
// product (64 v_mfma_f32_32x32x2_f32 per wave, 4096 cycles) serialises with its sigmoid epilogue (~2500 cycles).  The
// 16-bit matrix cores are 16x faster and co-issue with the VALU, and a float32 value is the sum of two f16 values to
// float32 accuracy:
//     xs = s x  (s a power of two that brings the row's largest element to ~2^14: no f16 overflow whatever the data's scale)
//     h = f16(xs), l = f16(xs - h):   |xs - h - l| <= 2^-24 |xs|   (11 + 11 mantissa bits, both roundings to nearest)
//     xs . ys = hh + hl + lh + O(2^-24 |xs||ys|)                    (the dropped ll product)
// three v_mfma_f32_32x32x16_f16 per 16 reduction indices, every partial product exact in the f32 accumulator (22-bit
// products) -- the accuracy class of the f32 instruction at 0.19x its cycles -- and the result is scaled back by the two
// rows' 1/s in the epilogue.  The split is made once per batch row (beside the gather) from x - x_ref (row 0 of the
// batch: the shift that keeps |x_i|^2 + |x_j|^2 - 2 x_i.x_j well conditioned), together with the row's squared norm.
//
// Workspace layout: planes[p][k / 32][row][k % 32] as f16 (k padded with zeros to a multiple of 32; p = 0 (h), 1 (l)) --
// chunk-major, because the tile kernel stages 32 reduction indices of 64 consecutive rows at a time: one contiguous 4-KB
// block here, 64 separate half-used cache lines in a row-major plane -- then norms[row], then rscale[row] (= 1/s) as float.
#pragma once
#include <cstdint>

#include "emx_common.h"

namespace emx {

constexpr int SPLIT_KC = 32;
__host__ __device__ inline int split_ld(int D) { return (D + SPLIT_KC - 1) / SPLIT_KC * SPLIT_KC; }
inline size_t split_bytes(int B, int D) { return (size_t)2 * B * split_ld(D) * sizeof(uint16_t) + (size_t)2 * B * sizeof(float); }

struct SplitRows {
  uint16_t* planes;    // f16 bit patterns
  long plane_stride;   // elements between planes
  int ld;              // padded row length (multiple of SPLIT_KC)
  int B;               // rows
  float* norms;        // |x - x_ref|^2
  float* rscale;       // 1 / s
};
inline SplitRows split_view(void* ws, int B, int D) {
  SplitRows s;
  s.planes = static_cast<uint16_t*>(ws);
  s.ld = split_ld(D);
  s.B = B;
  s.plane_stride = (long)B * s.ld;
  s.norms = reinterpret_cast<float*>(s.planes + 2 * s.plane_stride);
  s.rscale = s.norms + B;
  return s;
}
// element offset of (row, k) inside a plane
__host__ __device__ inline long split_index(const SplitRows& s, long row, int k) {
  return ((long)(k / SPLIT_KC) * s.B + row) * SPLIT_KC + (k % SPLIT_KC);
}

__device__ __forceinline__ uint16_t f16_bits(_Float16 v) { return __builtin_bit_cast(uint16_t, v); }

// one wave: row `src` minus `ref` (D columns) -> the two planes of row b, its squared norm and its 1/s
__device__ __forceinline__ void split_row(const SplitRows& s, const float* __restrict__ src, const float* __restrict__ ref, int D, long b,
                                          int lane) {
  float nrm = 0.f, mx = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = src[k] - ref[k];
    nrm = fmaf(v, v, nrm);
    mx = fmaxf(mx, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  nrm = wave_sum(nrm);
  // s = 2^(14 - E) with mx = f 2^E, f in [0.5, 1): s mx <= 2^14.  (mx = 0, denormal or huge: exponent clamped, s stays finite)
  int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;
  E = mx > 0.f ? min(max(E, -100), 100) : 14;
  const float sc = __uint_as_float((uint32_t)(127 + 14 - E) << 23), rs = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
  for (int k = lane; k < s.ld; k += 64) {
    const float v = k < D ? (src[k] - ref[k]) * sc : 0.f;   // exact scaling
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    uint16_t* o = s.planes + split_index(s, b, k);
    o[0] = f16_bits(h);
    o[s.plane_stride] = f16_bits(l);
  }
  if (lane == 0) {
    s.norms[b] = nrm;
    s.rscale[b] = rs;
  }
}

}  // namespace emx
