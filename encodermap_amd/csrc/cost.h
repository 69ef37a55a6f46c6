// K5 element functions of the row-mean cost on (periodic) differences, fused with PeriodicOutput (atan2): shared by the
// stand-alone cost kernel (elementwise.hip) and the cost phase of the forward chain kernel (mlp.hip).
//   replaces: loss_functions.py:708-769 (auto_loss), misc/distances.py:70-86 (periodic_distance), models/layers.py:181-190
#pragma once
#include "emx_common.h"

namespace emx {

constexpr float COST_TWO_PI = 6.283185307179586f;

struct CostArgs {
  const float* y;
  const float* target;
  const float* dpred_add;
  const float* coef_scale;
  float* dY;
  float* ws;
  float* loss;
  int ldy, ldt, off_sin, off_cos, periodic_output;
  int n, row_begin, row_end, variant, wrap, prescale;
  int wide;   // stand-alone kernel: S >= 1 workgroups per row (long rows, or few rows) instead of one wave per row (0)
  int vec4;   // row-block kernels (mlp.hip): a non-periodic output with mean_abs / mean_square -- the column-wise cost phase (cost_cols)
  float period, out_scale, gnorm;  // gnorm = coef / (B*n) (abs, square) or coef / B (norm)
};

// host: the scalar part of the description (pointers and row range are the caller's)
inline void fill_cost_scalars(CostArgs& a, int B, int n, int ldy, int ldt, int off_sin, int off_cos, int periodic_output, float periodicity,
                              int variant, float coef) {
  a.ldy = ldy;
  a.ldt = ldt;
  a.off_sin = off_sin;
  a.off_cos = off_cos;
  a.periodic_output = periodic_output;
  a.n = n;
  a.variant = variant;
  // periodicity < 0 (finite): the OUTPUT's period is |periodicity| (the atan2 of a periodic output is scaled to it) but the
  // differences are plain, not wrapped -- reconstruction_loss / "mse" behind PeriodicOutput (loss_functions.py:541-547)
  const float per = std::fabs(periodicity);
  a.wrap = std::isfinite(periodicity) && periodicity > 0.f;
  a.period = per;
  a.out_scale = (std::isfinite(per) && per != COST_TWO_PI) ? per / COST_TWO_PI : 1.f;
  const double denom = (variant == EMX_MEAN_NORM) ? (double)B : (double)B * n;
  a.gnorm = (float)(coef / denom);
}

// prediction (pred_s, pred_c: the sin / cos columns when the output is periodic), |difference| (wrapped) and its sign.
// PO: 1 / 0 = the output is / is not periodic at compile time (left to a run-time test the compiler if-converts the branch:
// atan2f and its ~60 instructions run for every element of a non-periodic output too), -1 = run-time test.
template <int PO = -1>
__device__ __forceinline__ void cost_elem(const CostArgs& a, const float* yrow, const float* trow, int k, float& pred_s,
                                          float& pred_c, float& delta, float& ddelta) {
#pragma clang fp contract(off)   // the stand-alone kernel and the cost phase of the chain kernel must round alike, whatever surrounds the call
  float pred;
  if (PO == 1 || (PO < 0 && a.periodic_output)) {
    pred_s = yrow[a.off_sin + k];
    pred_c = yrow[a.off_cos + k];
    pred = atan2f(pred_s, pred_c) * a.out_scale;
  } else {
    pred_s = pred_c = 0.f;
    pred = yrow[a.off_sin + k];
  }
  float diff = pred - trow[k];
  float d = fabsf(diff);
  float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
  if (a.wrap) {
    float w = a.period - d;
    if (d <= w) {
      delta = d;
      ddelta = sgn;
    } else {
      delta = w;
      ddelta = -sgn;
    }
  } else {
    delta = d;
    ddelta = sgn;
  }
}

template <int PO = -1>
__device__ __forceinline__ void cost_store(const CostArgs& a, float* drow, const float* addrow, int k, float g, float s, float c) {
#pragma clang fp contract(off)
  if (!drow) return;
  if (addrow) g += addrow[k];
  if (PO == 1 || (PO < 0 && a.periodic_output)) {
    float inv = a.out_scale / (s * s + c * c);
    drow[a.off_sin + k] = g * c * inv;
    drow[a.off_cos + k] = -g * s * inv;
  } else {
    drow[a.off_sin + k] = g;
  }
}

}  // namespace emx
