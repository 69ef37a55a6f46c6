// Activations of the dense layers and their derivatives: shared by the training kernels (mlp.hip) and the inference kernel (infer.hip).
#pragma once
#include "emx_common.h"

namespace emx {

constexpr float TANH_RS = 1.f / 16384.f;   // reciprocal of the fixed scale of tanh outputs

// tanh(x) = 1 - 2 / (e^{2x} + 1): two transcendental + three plain VALU instructions, absolute error <= ~1.5e-7 over the
// whole range (v_exp_f32 and v_rcp_f32 are 1-ulp; e = inf gives rcp = 0 -> 1, e = 0 gives -1), which is the float32
// resolution of the activations.
__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);        // e^{2x}
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// Activations of a layer (emx_mlp_desc::act; models.py:3179-3220 hands any Keras activation name to Dense): 1 = tanh is the fast
// path of every kernel here (fixed-scale images, its own switch cases); 2 .. 8 are the Keras activations whose derivative is a
// function of the OUTPUT -- so that, like tanh, the backward pass needs h and not the pre-activation -- evaluated with the
// hardware exp2 / log2 / rcp (absolute error ~1e-7, the float32 resolution of O(1) activations).
enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_ELU = 4, ACT_SOFTPLUS = 5, ACT_SELU = 6, ACT_SOFTSIGN = 7, ACT_EXP = 8, ACT_LAST = 8 };
constexpr float SELU_ALPHA = 1.6732632423543772f, SELU_SCALE = 1.0507009873554805f, LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float act_apply(const int kind, const float x) {   // kind: wave-uniform
  switch (kind) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-LOG2E * x));
    case ACT_ELU: return x > 0.f ? x : __builtin_amdgcn_exp2f(LOG2E * x) - 1.f;
    case ACT_SOFTPLUS: return fmaxf(x, 0.f) + LN2 * __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-LOG2E * fabsf(x)));
    case ACT_SELU: return SELU_SCALE * (x > 0.f ? x : SELU_ALPHA * (__builtin_amdgcn_exp2f(LOG2E * x) - 1.f));
    case ACT_SOFTSIGN: return x * __builtin_amdgcn_rcpf(1.f + fabsf(x));
    case ACT_EXP: return __builtin_amdgcn_exp2f(LOG2E * x);
    default: return x;
  }
}
__device__ __forceinline__ float act_deriv_from_output(const int kind, const float h) {   // d act / d x as a function of h = act(x)
  switch (kind) {
    case ACT_RELU: return h > 0.f ? 1.f : 0.f;
    case ACT_SIGMOID: return h * (1.f - h);
    case ACT_ELU: return h > 0.f ? 1.f : h + 1.f;
    case ACT_SOFTPLUS: return 1.f - __builtin_amdgcn_exp2f(-LOG2E * h);
    case ACT_SELU: return h > 0.f ? SELU_SCALE : h + SELU_SCALE * SELU_ALPHA;
    case ACT_SOFTSIGN: { const float t = 1.f - fabsf(h); return t * t; }
    case ACT_EXP: return h;
    default: return 1.f;
  }
}
__host__ __device__ inline bool act_bounded(int kind) { return kind == ACT_TANH || kind == ACT_SIGMOID || kind == ACT_SOFTSIGN; }

}  // namespace emx
