// Weight operand store of the dense layers ("mirror" in the C ABI, emx_mlp_mirror): the layout the row-block kernels
// (mlp.hip) stream, shared with the kernels that keep it current (the optimizer, elementwise.hip).
//
// A float32 product on gfx950's f32 matrix instruction costs 16x the cycles of a 16-bit one (MI355X_MICROARCH.md), and a
// float32 value is the sum of two f16 values to 22 significant bits.  The kernel [K = in, N = out] of every layer is
// therefore kept as a SCALED f16 PAIR: one power-of-two scale s per layer and two f16 planes
//     h = f16(s w),  l = f16(s w - h)            |s w - h - l| <= max(2^-22 |s w|, 2^-25)
// and a product x.w runs as the three 16-bit matrix instructions hh + hl + lh (every partial product exact in the f32
// accumulator; the dropped l.l term is 2^-22 relative), scaled back by 1/s once per output element.
//
// The scale.  One scale per LAYER lets a whole reduction accumulate inside the matrix instruction (a scale per K-step
// costs a multiply-add per accumulator element and K-step, which is what bounded the first version of these kernels).
// It brings the layer's largest |w| to [2^12, 2^13): three binades of headroom below the f16 maximum, and -- f16 being a
// floating-point format -- no loss of precision for being 8x smaller than it could be.  The optimizer rewrites the store in
// the launch that updates the weights, where the new maximum is not known before every workgroup has finished; it
// therefore uses the maximum RECORDED BY THE PREVIOUS UPDATE (weights move by ~lr per step) and records its own for the
// next one: slot[t & 1] of the layer's header receives (t << 32 | bits of max|w|) by a 64-bit atomic maximum, so a new
// step's first record supersedes whatever an older step left, without a reset.  A weight that outgrows the headroom within
// ONE update becomes an f16 infinity and the losses NaN: loud, not silent.  emx_mlp_mirror (fresh build) uses the exact
// maximum and seeds both slots.
//
// Layout.  Every layer is padded with zeros to multiples of 128 in both dimensions, so that the row-block kernels run
// REGULAR UNITS -- 128 output columns x 128 reduction indices: 8 tiles of 16 columns (one per wave) x 4 K-steps of 32 x 2
// planes = 64 pieces of 1 KB -- and the units are stored in the order the kernels execute them, so that a launch walks
// the store front to back and the address of a unit's package is the previous one + 64 KB:
//   headers           64 bytes per layer (WopHeader), region padded to 1 KB
//   forward units     for l = 0 .. n-1: for pass p (128 columns of out): for chunk c (128 k of in): [wave][K-step][plane] x 1 KB
//   backward units    for l = n-1 .. 0: for pass p (128 columns of in):  for chunk c (128 k of out): likewise
// A piece is the B fragment of ONE v_mfma_f32_16x16x32_f16: lane (n = lane & 15, g = lane >> 4) holds the 8 halves
//   B[32 ks + kk(g, e)][16 t + n],  kk(g, e) = 4 g + e (e < 4), 16 + 4 g + (e - 4) (e >= 4)
// (the k order inside a K-step is free as long as both operands agree; this one makes the activations' transposed LDS
// read conflict-free, mlp.hip) -- 16 bytes per lane, so one global_load_dwordx4 per wave fetches eight whole lines.
#pragma once
#include <cstdint>

#include "emx_common.h"

namespace emx {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4_builtin __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr long WOP_UNIT = 65536;   // bytes of one unit's package

struct WopArgs {
  int n_layers;
  int K[EMX_MAX_LAYERS], N[EMX_MAX_LAYERS];
  long w_off[EMX_MAX_LAYERS];          // float offset of the [K, N] kernel inside the flat parameter buffer
  long fwd_unit[EMX_MAX_LAYERS + 1];   // index of the layer's first forward unit (units of WOP_UNIT behind units_begin)
  long bwd_unit[EMX_MAX_LAYERS + 1];   // index of the layer's first backward unit; [n_layers]: total number of units
  long units_begin;                    // byte offset of unit 0
  int item_begin[EMX_MAX_LAYERS + 1];  // prefix sum of 8 x 32 slabs (work items of the kernels that write the store)
};

struct WopHeader {                 // 64 bytes per layer at the head of the store
  float rs;                        // 1/s of the layer's pieces as they stand
  float pad;
  unsigned long long slot[2];      // (update index << 32) | bits of the largest |w| that update wrote
  unsigned long long reserved[5];
};

__host__ __device__ inline int groups32(int x) { return (x + 31) >> 5; }
__host__ __device__ inline int groups128(int x) { return (x + 127) >> 7; }

inline void fill_wop_args(const emx_mlp_desc* desc, WopArgs& wa) {
  const int n = desc->n_layers;
  wa.n_layers = n;
  wa.units_begin = ((long)n * 64 + 1023) / 1024 * 1024;
  long u = 0;
  int items = 0;
  for (int l = 0; l < n; ++l) {
    wa.K[l] = desc->width[l];
    wa.N[l] = desc->width[l + 1];
    wa.w_off[l] = desc->w_off[l];
    wa.fwd_unit[l] = u;
    wa.item_begin[l] = items;
    u += (long)groups128(wa.K[l]) * groups128(wa.N[l]);
    items += ((wa.K[l] + 7) >> 3) * groups32(wa.N[l]);
  }
  wa.fwd_unit[n] = u;
  wa.item_begin[n] = items;
  for (int l = n - 1; l >= 0; --l) {
    wa.bwd_unit[l] = u;
    u += (long)groups128(wa.K[l]) * groups128(wa.N[l]);
  }
  wa.bwd_unit[n] = u;   // total
}
inline long wop_total_bytes(const WopArgs& wa) { return wa.units_begin + wa.bwd_unit[wa.n_layers] * WOP_UNIT; }

// byte offset of the piece (plane) of block row kb (32 k of in) / block column nb (32 columns of out), tile t of the pair
__host__ __device__ inline long wop_fwd_piece(const WopArgs& wa, int l, int kb, int nb, int t, int plane) {
  const long unit = wa.fwd_unit[l] + (long)(nb >> 2) * groups128(wa.K[l]) + (kb >> 2);
  return wa.units_begin + unit * WOP_UNIT + (((2 * (nb & 3) + t) * 4 + (kb & 3)) * 2 + plane) * 1024;
}
__host__ __device__ inline long wop_bwd_piece(const WopArgs& wa, int l, int kb, int nb, int t, int plane) {
  const long unit = wa.bwd_unit[l] + (long)(kb >> 2) * groups128(wa.N[l]) + (nb >> 2);
  return wa.units_begin + unit * WOP_UNIT + (((2 * (kb & 3) + t) * 4 + (nb & 3)) * 2 + plane) * 1024;
}

// Power-of-two scale from the bit pattern of a maximum |x| = f 2^E (f in [0.5, 1)): s = 2^(TOP - E) brings it to
// [2^(TOP-1), 2^TOP), rs = 1/s.  The exponent field is clamped to E in [-100, 100]; an all-zero group gets the largest
// scale (0 * s = 0).  Unsigned integer arithmetic only -- maxima of |x| are taken on the bit patterns (unsigned order =
// float order there), where fmaxf would cost a canonicalising extra instruction per operand.
template <int TOP>
__device__ __forceinline__ void pow2_scale_bits(unsigned ab, float& s, float& rs) {
  const unsigned ef = min(max(ab & 0x7f800000u, 26u << 23), 226u << 23);
  s = __uint_as_float(((unsigned)(253 + TOP) << 23) - ef);
  rs = __uint_as_float(ef - ((unsigned)(TOP - 1) << 23));
}
__device__ __forceinline__ unsigned abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
constexpr int WOP_TOP = 13;    // weights: the layer's maximum goes to [2^12, 2^13)
constexpr int ACT_TOP = 14;    // activations and gradients: a group's maximum goes to [2^13, 2^14)

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {   // (all lanes active; DPP inside the rows, lane swaps across them: no ds_bpermute)
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
  return max_xor32(max_xor16(v));
}

__device__ __forceinline__ half4 lds_read_tr16(const _Float16* p) {
  return (half4)__builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_builtin*)(p));
}

// One wave writes its part of the pieces of layer l from a SLAB of 8 x 32 weights -- rows k = 8 k8 .. 8 k8 + 7 of the
// kernel, columns 32 nb .. 32 nb + 31: lane (kr = lane >> 3, c4 = lane & 7) passes w[i] = W[8 k8 + kr][32 nb + 4 c4 + i]
// (zero outside the layer).  s: the layer's scale.  A slab is a quarter of the 32 x 32 block behind two forward and two
// backward pieces per plane: every lane ends up with 8 of the 16 bytes of one fragment lane, in each direction, and the
// stores of the four slabs of a block complete each other's cache lines (one float4 per lane and tensor, as in the flat
// optimizer pass: the optimizer is a latency-bound stream, a whole block per wave made its critical path four times longer).
// img: 1 KB of LDS private to the wave (two [8 k][32 n] f16 planes); every lane must call (transposed LDS read: EXEC all
// ones).  Returns the largest |w| of the slab (bit pattern, wave-uniform).
__device__ __forceinline__ unsigned wop_write_slab(const WopArgs& wa, char* __restrict__ store, int l, int k8, int nb, const float (&w)[4], float s,
                                                   int lane, _Float16* img) {
  unsigned mx = max(max(abs_bits(w[0]), abs_bits(w[1])), max(abs_bits(w[2]), abs_bits(w[3])));
  mx = wave_max_u32(mx);
  const int kr = lane >> 3, c4 = lane & 7;
  half4 h, lo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = w[i] * s;   // exact
    h[i] = (_Float16)t;
    lo[i] = (_Float16)(t - (float)h[i]);
  }
  *reinterpret_cast<half4*>(img + kr * 32 + 4 * c4) = h;
  *reinterpret_cast<half4*>(img + 256 + kr * 32 + 4 * c4) = lo;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (no other wave touches img)
  __builtin_amdgcn_wave_barrier();
  const int kb = k8 >> 2, j = k8 & 3;   // the slab is rows 8 j .. 8 j + 7 of block row kb
#pragma unroll
  for (int plane = 0; plane < 2; ++plane) {
    const _Float16* pl = img + 256 * plane;
    {
      // forward pieces B[k = in][col = out]: one transposed read serves both tiles -- the 16-lane group G = lane >> 4 takes
      // the 4 x 16 block (k block G & 1, tile G >> 1); lane 4 q + p of the group addresses row 4 (G & 1) + q, columns
      // 16 (G >> 1) + 4 p ..; lane i receives column 16 t + i of the four rows = 4 consecutive k of one fragment lane
      const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, i = lane & 15;
      const half4 f = lds_read_tr16(pl + (4 * (G & 1) + q) * 32 + 16 * (G >> 1) + 4 * p);
      const int r = 8 * j + 4 * (G & 1);             // first of the four k (0 .. 31) inside the K-step
      const int fg = (r & 15) >> 2, second = r >> 4;  // fragment lane group, element half (kk(g, e) of the piece)
      *reinterpret_cast<half4*>(store + wop_fwd_piece(wa, l, kb, nb, G >> 1, plane) + (16 * fg + i) * 16 + 8 * second) = f;
    }
    {
      // backward pieces B[k = out][col = in]: row kr of the slab is in-feature 8 j + kr of block row kb: tile t = j >> 1,
      // fragment column n = 8 (j & 1) + kr; lane (kr, g, half) takes out = 4 g + e / 16 + 4 g + e from the slab's row
      const int g = (lane >> 1) & 3, second = lane & 1;
      const half4 b = *reinterpret_cast<const half4*>(pl + kr * 32 + 16 * second + 4 * g);
      *reinterpret_cast<half4*>(store + wop_bwd_piece(wa, l, kb, nb, j >> 1, plane) + (16 * g + 8 * (j & 1) + kr) * 16 + 8 * second) = b;
    }
  }
  __builtin_amdgcn_wave_barrier();   // img may be rewritten by the caller's next slab only behind these reads
  return mx;
}

// work item -> (layer, k8, nb): slabs of 8 rows x 32 columns
__device__ __forceinline__ void wop_item(const WopArgs& wa, int item, int& l, int& k8, int& nb) {
  l = 0;
  while (l + 1 < wa.n_layers && item >= wa.item_begin[l + 1]) ++l;
  const int rel = item - wa.item_begin[l], NB = groups32(wa.N[l]);
  k8 = rel / NB;
  nb = rel - k8 * NB;
}

__device__ __forceinline__ WopHeader* wop_header(char* store, int l) { return reinterpret_cast<WopHeader*>(store) + l; }

}  // namespace emx
