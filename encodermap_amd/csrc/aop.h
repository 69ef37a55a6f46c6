// Activation operand store of the weight-gradient kernel (mlp.hip): what the row-block kernels leave for
// dW_l = X_l^T delta_l, written in their epilogues while the values are in registers.
//
// Every tensor T [rows, n] that takes part in a weight gradient (the network input, the activations h_l, the
// pre-activation gradients delta_l) is stored per ROW BLOCK of 16 rows and COLUMN GROUP of 32 columns as one block-scaled
// f16 pair (wop.h): a power-of-two scale s for the 16 x 32 block and two 1-KB fragments (h, l planes), each the operand
// of ONE v_mfma_f32_32x32x16_f16 with the reduction index on the ROWS:
//     lane (r = lane & 31, hh = lane >> 5) holds the 8 halves  T[16 rb + 8 hh + e][32 cg + r],  e < 8
// (A[i = r][k = 8 hh + e] of X^T, and equally B[k][j = r] of delta).  Layout per tensor:
//     fragments [rb][cg][plane] x 1 KB, then 1/s as float [rb][cg]
// Tensor ids: 0 = the input of layer 0, 1 + l = h_l, 1 + n_layers + l = delta_l.
#pragma once
#include "emx_common.h"
#include "wop.h"

namespace emx {

struct AopLayout {
  int n_layers, RB;                         // row blocks
  long off[2 * EMX_MAX_LAYERS + 2];         // byte offset of tensor id
  int CG[2 * EMX_MAX_LAYERS + 2];           // column groups of tensor id
  long total;
};

__host__ __device__ inline long aop_tensor_bytes(int RB, int CG) { return (long)RB * CG * 2048 + (((long)RB * CG * 4 + 255) / 256) * 256; }

inline void aop_layout(const emx_mlp_desc* desc, int n_rows, AopLayout& L) {
  const int n = desc->n_layers;
  L.n_layers = n;
  L.RB = (n_rows + 15) / 16;
  long o = 0;
  auto put = [&](int id, int width) {
    L.off[id] = o;
    L.CG[id] = groups32(width);
    o += aop_tensor_bytes(L.RB, L.CG[id]);
  };
  put(0, desc->width[0]);
  for (int l = 0; l < n; ++l) put(1 + l, desc->width[l + 1]);
  for (int l = 0; l < n; ++l) put(1 + n + l, desc->width[l + 1]);
  L.total = o;
}

struct AopTensor {   // device view of one tensor
  char* frag;        // fragments; null: the tensor is not kept
  float* rscale;
  int CG;
};
inline AopTensor aop_tensor(void* aop, const AopLayout& L, int id) {
  AopTensor t{nullptr, nullptr, 0};
  if (!aop) return t;
  t.frag = static_cast<char*>(aop) + L.off[id];
  t.rscale = reinterpret_cast<float*>(t.frag + (long)L.RB * L.CG[id] * 2048);
  t.CG = L.CG[id];
  return t;
}

}  // namespace emx
