// Weight operand store of the dense layers ("mirror" in the C ABI, emx_mlp_mirror): the layout the row-block kernels
// (mlp.hip) stream, shared with the kernels that keep it current (the optimizer, elementwise.hip).
//
// A float32 product on gfx950's f32 matrix instruction costs 16x the cycles of a 16-bit one (MI355X_MICROARCH.md), and a
// float32 value is the sum of two f16 values to 22 significant bits.  The kernel [K = in, N = out] of every layer is
// therefore kept as BLOCK-SCALED f16 PAIRS: per 32 x 32 block (kb, nb) of W one power-of-two scale s (the block's
// largest |w| goes to ~2^14, so any magnitude is representable) and two f16 planes
//     h = f16(s w),  l = f16(s w - h)            |s w - h - l| <= max(2^-22 |s w|, 2^-25)
// and a product x.w runs as the three 16-bit matrix instructions hh + hl + lh (every partial product exact in the f32
// accumulator; the dropped l.l term is 2^-22 relative), scaled back by 1/s per K-step in the consuming kernel.
//
// Per layer the store holds, in this order (byte offsets from the layer's begin[]):
//   forward operand   B[k = in][col = out]  : pieces [nb][kb][tile t < 2][plane < 2], 1 KB each
//   backward operand  B[k = out][col = in]  : pieces [kb][nb][tile t < 2][plane < 2], 1 KB each
//   1/s               float [kb][nb]
// A piece is the B fragment of ONE v_mfma_f32_16x16x32_f16: lane (n = lane & 15, g = lane >> 4) holds the 8 halves
//   B[32 ks + kk(g, e)][32 cg + 16 t + n],  kk(g, e) = 4 g + e (e < 4), 16 + 4 g + (e - 4) (e >= 4)
// (the k order inside a K-step is free as long as both operands agree; this one makes the activations' transposed LDS
// read conflict-free, mlp.hip) -- 16 bytes per lane, so one global_load_dwordx4 per wave fetches eight whole lines.
// Zero padded: K and N are rounded up to multiples of 32.
#pragma once
#include <cstdint>

#include "emx_common.h"

namespace emx {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4_builtin __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct WopArgs {
  int n_layers;
  int K[EMX_MAX_LAYERS], N[EMX_MAX_LAYERS];
  long w_off[EMX_MAX_LAYERS];          // float offset of the [K, N] kernel inside the flat parameter buffer
  long begin[EMX_MAX_LAYERS + 1];      // byte offset of the layer's region inside the store
  int item_begin[EMX_MAX_LAYERS + 1];  // prefix sum of 32 x 32 blocks (work items of the kernels that write the store)
};

__host__ __device__ inline int groups32(int x) { return (x + 31) >> 5; }
__host__ __device__ inline long wop_operand_bytes(int K, int N) { return 4096L * groups32(K) * groups32(N); }
__host__ __device__ inline long wop_layer_bytes(int K, int N) {
  return 2 * wop_operand_bytes(K, N) + (((long)groups32(K) * groups32(N) * 4 + 1023) / 1024) * 1024;   // a multiple of the piece size
}
// byte offsets inside a layer's region
__host__ __device__ inline long wop_fwd_piece(int K, int N, int kb, int nb, int t, int plane) {
  return ((((long)nb * groups32(K) + kb) * 2 + t) * 2 + plane) * 1024;
}
__host__ __device__ inline long wop_bwd_piece(int K, int N, int kb, int nb, int t, int plane) {
  return wop_operand_bytes(K, N) + ((((long)kb * groups32(N) + nb) * 2 + t) * 2 + plane) * 1024;
}
__host__ __device__ inline long wop_scale(int K, int N, int kb, int nb) { return 2 * wop_operand_bytes(K, N) + ((long)kb * groups32(N) + nb) * 4; }

inline void fill_wop_args(const emx_mlp_desc* desc, WopArgs& wa) {
  wa.n_layers = desc->n_layers;
  long o = 0;
  int items = 0;
  for (int l = 0; l <= desc->n_layers; ++l) {
    wa.begin[l] = o;
    wa.item_begin[l] = items;
    if (l < desc->n_layers) {
      wa.K[l] = desc->width[l];
      wa.N[l] = desc->width[l + 1];
      wa.w_off[l] = desc->w_off[l];
      o += wop_layer_bytes(wa.K[l], wa.N[l]);
      items += groups32(wa.K[l]) * groups32(wa.N[l]);
    }
  }
}

// power-of-two scale that brings mx (>= 0) to [2^13, 2^14), and its reciprocal; mx = 0, denormal or huge: exponent clamped
__device__ __forceinline__ void pow2_scale(float mx, float& s, float& rs) {
  int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;   // mx = f 2^E, f in [0.5, 1)
  E = mx > 0.f ? min(max(E, -100), 100) : 14;
  s = __uint_as_float((uint32_t)(127 + 14 - E) << 23);
  rs = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ half4 lds_read_tr16(const _Float16* p) {
  return (half4)__builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_builtin*)(p));
}

// One wave writes the two operand pieces-sets and the scale of block (kb, nb) of layer l from the block's weights:
// lane (kr = lane >> 3, c4 = lane & 7) passes wv[4 j + i] = W[32 kb + kr + 8 j][32 nb + 4 c4 + i] (zero outside the layer).
// img: 4 KB of LDS private to the wave (two [32 k][32 n] f16 planes); every lane of the wave must call (the transposed
// LDS read needs EXEC all ones).
__device__ __forceinline__ void wop_write_block(const WopArgs& wa, char* __restrict__ store, int l, int kb, int nb, const float (&wv)[16],
                                                int lane, _Float16* img) {
  const int K = wa.K[l], N = wa.N[l];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fabsf(wv[i]));
  mx = wave_max(mx);
  float s, rs;
  pow2_scale(mx, s, rs);
  const int kr = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    half4 h, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = wv[4 * j + i] * s;   // exact
      h[i] = (_Float16)t;
      lo[i] = (_Float16)(t - (float)h[i]);
    }
    *reinterpret_cast<half4*>(img + (kr + 8 * j) * 32 + 4 * c4) = h;
    *reinterpret_cast<half4*>(img + 1024 + (kr + 8 * j) * 32 + 4 * c4) = lo;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (no other wave touches img)
  __builtin_amdgcn_wave_barrier();
  char* base = store + wa.begin[l];
  const int n = lane & 15, g = lane >> 4;
#pragma unroll
  for (int plane = 0; plane < 2; ++plane) {
    const _Float16* pl = img + 1024 * plane;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // forward piece: B[k = in][col = out]: transposed reads of the [k][n] image -- lane 4 q + p of a 16-lane group addresses
      // row k0 + q, columns 16 t + 4 p ..; lane i of the group receives column 16 t + i of the four rows
      const int q = (lane & 15) >> 2, p = lane & 3;
      const half4 f0 = lds_read_tr16(pl + (4 * g + q) * 32 + 16 * t + 4 * p);
      const half4 f1 = lds_read_tr16(pl + (16 + 4 * g + q) * 32 + 16 * t + 4 * p);
      half8 fw = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
      *reinterpret_cast<half8*>(base + wop_fwd_piece(K, N, kb, nb, t, plane) + lane * 16) = fw;
      // backward piece: B[k = out][col = in]: row 16 t + n of the image, columns kk(g, e)
      const half4 b0 = *reinterpret_cast<const half4*>(pl + (16 * t + n) * 32 + 4 * g);
      const half4 b1 = *reinterpret_cast<const half4*>(pl + (16 * t + n) * 32 + 16 + 4 * g);
      half8 bw = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      *reinterpret_cast<half8*>(base + wop_bwd_piece(K, N, kb, nb, t, plane) + lane * 16) = bw;
    }
  }
  if (lane == 0) *reinterpret_cast<float*>(base + wop_scale(K, N, kb, nb)) = rs;
  __builtin_amdgcn_wave_barrier();   // img may be rewritten by the caller's next block only behind these reads
}

// loads the block's weights from the flat parameter buffer in wop_write_block's lane order (zero outside the layer)
__device__ __forceinline__ void wop_load_block(const WopArgs& wa, const float* __restrict__ params, int l, int kb, int nb, int lane,
                                               float (&wv)[16]) {
  const int K = wa.K[l], N = wa.N[l];
  const int kr = lane >> 3, c4 = lane & 7;
  const float* W = params + wa.w_off[l];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 32 * kb + kr + 8 * j, c = 32 * nb + 4 * c4;
    if ((N & 3) == 0 && (wa.w_off[l] & 3) == 0 && k < K && c + 3 < N) {
      const float4 v = *reinterpret_cast<const float4*>(W + (long)k * N + c);
      wv[4 * j] = v.x, wv[4 * j + 1] = v.y, wv[4 * j + 2] = v.z, wv[4 * j + 3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[4 * j + i] = (k < K && c + i < N) ? W[(long)k * N + c + i] : 0.f;
    }
  }
}

// work item -> (layer, kb, nb)
__device__ __forceinline__ void wop_item(const WopArgs& wa, int item, int& l, int& kb, int& nb) {
  l = 0;
  while (l + 1 < wa.n_layers && item >= wa.item_begin[l + 1]) ++l;
  const int rel = item - wa.item_begin[l], NB = groups32(wa.N[l]);
  kb = rel / NB;
  nb = rel - kb * NB;
}

}  // namespace emx
