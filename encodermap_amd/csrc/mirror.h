// Operand store ("mirror") of the dense layers: layout shared by the kernels that read it (mlp.hip) and the ones that
// keep it current (Adam, elementwise.hip).  See include/emx.h, emx_mlp_mirror.
//
// Per layer two blocks of 256 ceil(K/16) ceil(N/16) floats: the forward operand (output column j = n, reduction index
// k = input feature) and the backward operand (j = input feature, k = output feature), each stored as
// [tile t][K-step q][lane = 16 g + m][i]: value (j = 16 t + m, k = 16 q + 4 g + i), zero outside the layer.
#pragma once
#include "emx_common.h"

namespace emx {

struct MirrorArgs {
  int n_layers;
  int width[EMX_MAX_LAYERS + 1];
  long w_off[EMX_MAX_LAYERS];
  long begin[EMX_MAX_LAYERS + 1];   // float offset of layer l's pair of blocks
};

__host__ __device__ inline long operand_block(int K, int N) { return 256L * ((K + 15) / 16) * ((N + 15) / 16); }

inline void fill_mirror_args(const emx_mlp_desc* desc, MirrorArgs& ma) {
  ma.n_layers = desc->n_layers;
  long o = 0;
  for (int l = 0; l <= desc->n_layers; ++l) {
    ma.width[l] = desc->width[l];
    ma.begin[l] = o;
    if (l < desc->n_layers) {
      ma.w_off[l] = desc->w_off[l];
      o += 2 * operand_block(desc->width[l], desc->width[l + 1]);
    }
  }
}

// offset of element (j, k) inside an operand block with Q K-steps per tile
__device__ __forceinline__ long operand_index(int j, int k, int Q) {
  return (((long)(j >> 4) * Q + (k >> 4)) << 8) + ((((k & 15) >> 2) << 4 | (j & 15)) << 2) + (k & 3);
}

// writes parameter i (value w) to its two places in the mirror; biases have none
__device__ __forceinline__ void mirror_scatter(const MirrorArgs& ma, float* __restrict__ mirror, long i, float w) {
#pragma unroll
  for (int l = 0; l < EMX_MAX_LAYERS; ++l) {
    if (l >= ma.n_layers) break;
    const int K = ma.width[l], N = ma.width[l + 1];
    const long rel = i - ma.w_off[l];
    if (rel >= 0 && rel < (long)K * N) {
      const int k = (int)(rel / N), n = (int)(rel - (long)k * N);
      float* fwd = mirror + ma.begin[l];
      fwd[operand_index(n, k, (K + 15) >> 4)] = w;
      fwd[operand_block(K, N) + operand_index(k, n, (N + 15) >> 4)] = w;
      return;
    }
  }
}

// four consecutive parameters i0 .. i0+3 (i0 % 4 == 0): when they are four consecutive output columns of one kernel row
// whose width is a multiple of 4 (the usual case) the backward operand takes them as ONE 16-byte store and the row / layer
// search runs once; otherwise element by element
__device__ __forceinline__ void mirror_scatter4(const MirrorArgs& ma, float* __restrict__ mirror, long i0, float w0, float w1, float w2,
                                                float w3) {
#pragma unroll
  for (int l = 0; l < EMX_MAX_LAYERS; ++l) {
    if (l >= ma.n_layers) break;
    const int K = ma.width[l], N = ma.width[l + 1];
    const long rel = i0 - ma.w_off[l];
    if ((N & 3) == 0 && (ma.w_off[l] & 3) == 0 && rel >= 0 && rel < (long)K * N) {
      const int k = (int)(rel / N), n = (int)(rel - (long)k * N);   // n % 4 == 0
      float* fwd = mirror + ma.begin[l];
      const long f0 = operand_index(n, k, (K + 15) >> 4);           // j = n .. n+3 -> lanes m .. m+3: 4 floats apart
      fwd[f0] = w0;
      fwd[f0 + 4] = w1;
      fwd[f0 + 8] = w2;
      fwd[f0 + 12] = w3;
      float4* bwd = reinterpret_cast<float4*>(fwd + operand_block(K, N) + operand_index(k, n, (N + 15) >> 4));   // k index n .. n+3 = i
      *bwd = make_float4(w0, w1, w2, w3);
      return;
    }
  }
  mirror_scatter(ma, mirror, i0, w0);
  mirror_scatter(ma, mirror, i0 + 1, w1);
  mirror_scatter(ma, mirror, i0 + 2, w2);
  mirror_scatter(ma, mirror, i0 + 3, w3);
}

}  // namespace emx
