// The head of an ADC training step (the batch's feature rows and Cartesian rows by index, the sin/cos encoder input of the angular
// blocks, the soft-start scale of the step, the batch means of the feature columns): the work of adc_head_kernel (backmap.hip) as a
// function of a flat thread index, so that the optimizer launch of the step in front can run it on workgroups of its own
// (elementwise.hip: emx_next_batch::adc_head), as it runs the row gather of the EncoderMap step.
#pragma once
#include <cstdint>

#include "emx_common.h"

namespace emx {

constexpr int ADC_CM_PH = 16;   // row phases of the column means (emx_colmean's partition)

struct AdcHeadArgs {
  const float* feat_data;
  const float* cart_data;
  const int64_t* idx;
  const int64_t* cursor;
  float* feat;
  float* cart;
  float* xin;
  int B, F, C3, ld_xin, n_parts;
  int off[4], n[4], col[4];
  float ang_scale;
  const int64_t* step;
  float soft_a, soft_b, scale_c;
  float* cart_scale;
  float* feat_mean;   // [F] batch means of the feature columns (null: not wanted)
  int row_blocks;     // virtual workgroups of 256 threads that copy rows
  int batch_step;     // the batch gathered is *cursor + batch_step ...
  int n_batches;      // ... unless that lies behind the index block (0: no check)
};

// rows: virtual workgroup vb of nvb (256 threads each), thread t of it.  step_ahead: how far the step this head belongs to lies
// in front of the counter -- 0 for the head launch of a step and for a head prefetched behind the step's bookkeeping (the
// counter has advanced already), 1 when the optimizer launch of step t prepares step t + 1 BEFORE the counter moves
// (EMX_FOLD_FINISH=0): the soft-start scale is that of the step the batch is for (callbacks.py:584-606).
__device__ __forceinline__ void adc_head_rows(const AdcHeadArgs& a, const int64_t* __restrict__ idx, int vb, int nvb, int t, int step_ahead = 0) {
  if (a.cart_scale && vb == 0 && t == 0) {
    float sc = a.scale_c;
    if (a.soft_a >= 0.f) {
      const float s = (float)(*a.step + step_ahead);
      if (s < a.soft_a) sc = 0.f;
      else if (s <= a.soft_b) sc = a.scale_c / (a.soft_b - a.soft_a) * (s - a.soft_a);
    }
    *a.cart_scale = sc;
  }
  for (int b = vb; b < a.B; b += nvb) {
    const long row = idx[b];
    const float* __restrict__ fs = a.feat_data + row * a.F;
    const float* __restrict__ cs = a.cart_data + row * a.C3;
    float* fd = a.feat + (long)b * a.F;
    float* cd = a.cart + (long)b * a.C3;
    for (int k = t; k < a.F; k += 256) fd[k] = fs[k];
    for (int k = t; k < a.C3; k += 256) cd[k] = cs[k];
    for (int p = 0; p < a.n_parts; ++p) {
      float* xo = a.xin + (long)b * a.ld_xin + a.col[p];
      const int n = a.n[p];
      for (int k = t; k < n; k += 256) {
        float sn, cn;
        sincos_bounded(fs[a.off[p] + k] * a.ang_scale, &sn, &cn);
        xo[k] = sn;
        xo[n + k] = cn;
      }
    }
  }
}

// column means (emx_colmean's partition and order of additions): a workgroup of CW x 16 threads takes CW columns, thread
// (cl, ph) = (t % CW, t / CW) the rows ph, ph + 16, ... of column c0 + cl; part: [16][CW] floats of LDS
template <int CW>
__device__ __forceinline__ void adc_head_means(const AdcHeadArgs& a, const int64_t* __restrict__ idx, int c0, int t, float (*part)[CW]) {
  const int cl = t % CW, ph = t / CW, c = c0 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < a.F) {
    int b = ph;
    for (; b + 3 * ADC_CM_PH < a.B; b += 4 * ADC_CM_PH) {
      const long r0 = idx[b], r1 = idx[b + ADC_CM_PH], r2 = idx[b + 2 * ADC_CM_PH], r3 = idx[b + 3 * ADC_CM_PH];
      s0 += a.feat_data[r0 * a.F + c];
      s1 += a.feat_data[r1 * a.F + c];
      s2 += a.feat_data[r2 * a.F + c];
      s3 += a.feat_data[r3 * a.F + c];
    }
    for (; b < a.B; b += ADC_CM_PH) s0 += a.feat_data[idx[b] * a.F + c];
  }
  part[ph][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < a.F) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < ADC_CM_PH; q += 4) s += (part[q][cl] + part[q + 1][cl]) + (part[q + 2][cl] + part[q + 3][cl]);
    a.feat_mean[c] = s / (float)a.B;
  }
}

}  // namespace emx
