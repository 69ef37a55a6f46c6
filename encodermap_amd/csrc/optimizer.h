// The optimizer's device code, shared by the optimizer launch (elementwise.hip, adam_kernel) and by the weight-gradient launch
// that runs the optimizer as its tail (mlp.hip, mlp_wgrad_adam_kernel): the next batch's row gather, L2 + clip + Keras Adam on
// slice partials, and the rewrite of the weight operand store (wop.h) with the new weights.
//   autoencoder.py:741-743 (Adam(lr, clipvalue=1).apply_gradients), loss_functions.py:488-506 (regularization_loss),
//   autoencoder.py:845-856 (tf.data batches)
#pragma once
#include "emx_common.h"
#include "split.h"
#include "wop.h"
#include <climits>
#include <cmath>

#ifndef ASTAMP
#define ASTAMP(k, blk)
#endif

namespace emx {

struct AdcHeadArgs;

struct GatherArgs {
  const float* data;
  int D;
  const int64_t* idx;
  const int64_t* cursor;
  int B;
  int batch_step;     // gather batch *cursor + batch_step ...
  int n_batches;      // ... unless that is >= n_batches (0: no bound)
  float ang_scale;
  float* xb;
  float* xin;
  SplitRows split;    // optional (planes != null): f16 planes + norms of (row - row 0 of the batch) for the all-pairs kernel
  const emx::AdcHeadArgs* adc_head;   // optional (optimizer launch only): the head of the NEXT ADC step instead of the row gather
};

// rows b = wave, wave + nwaves, ... of the batch
__device__ __forceinline__ void gather_rows_body(const GatherArgs& g, int wave, int nwaves, int lane) {
  const int64_t* __restrict__ idx = g.idx;
  if (g.cursor) {
    const long batch = *g.cursor + g.batch_step;
    if (g.n_batches > 0 && batch >= g.n_batches) return;
    idx += batch * (long)g.B;
  }
  const int D = g.D;
  const float* __restrict__ ref = g.split.planes ? g.data + idx[0] * (long)D : nullptr;
  for (int b = wave; b < g.B; b += nwaves) {
    const float* __restrict__ src = g.data + idx[b] * (long)D;
    if (g.split.planes) split_row(g.split, src, ref, D, b, lane);
    for (int k = lane; k < D; k += 64) {
      const float v = src[k];
      if (g.xb) g.xb[(long)b * D + k] = v;
      if (g.xin) {
        float s, c;
        sincos_bounded(v * g.ang_scale, &s, &c);
        g.xin[(long)b * 2 * D + k] = s;
        g.xin[(long)b * 2 * D + D + k] = c;
      }
    }
  }
}

// The same for the optimizer launch, where the gather is on the critical path of a short kernel: TWO rows per wave, one per
// half-wave, 16 bytes per lane (a 128-column row is ONE load per lane, and the wave's dependent trips to memory -- cursor, row
// indices, rows -- serve two rows).  Half as many waves as rows: with one wave per row the launch was more waves than the
// chip holds at once, and the last rows started when the first were done.  Needs D % 4 == 0 and 16-byte aligned rows.
__device__ __forceinline__ void gather_rows2_body(const GatherArgs& g, int wave, int nwaves, int lane) {
  const int64_t* __restrict__ idx = g.idx;
  if (g.cursor) {
    const long batch = *g.cursor + g.batch_step;
    if (g.n_batches > 0 && batch >= g.n_batches) return;
    idx += batch * (long)g.B;
  }
  const int D = g.D, half = lane >> 5, l32 = lane & 31;
  const bool split = g.split.planes != nullptr;
  for (int b0 = 2 * wave; b0 < g.B; b0 += 2 * nwaves) {
    const int b = b0 + half;
    const bool live = b < g.B;
    const long irow = idx[min(b, g.B - 1)], iref = idx[0];
    const float4* __restrict__ src = reinterpret_cast<const float4*>(g.data + irow * (long)D);
    const float4* __restrict__ ref = reinterpret_cast<const float4*>(g.data + iref * (long)D);
    const int D4 = D >> 2;
    float sc = 1.f;
    if (split) {
      float nrm = 0.f, mx = 0.f;
#pragma unroll 2
      for (int k4 = l32; k4 < D4; k4 += 32) {
        const float4 x = src[k4], r = ref[k4];
        const float v0 = x.x - r.x, v1 = x.y - r.y, v2 = x.z - r.z, v3 = x.w - r.w;
        nrm = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, nrm))));
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
      }
      // (within the half-wave: lane ^ 16 by a lane swap, the rest by DPP inside the rows of 16 -- no ds_bpermute)
      {
        unsigned ma, mb;
        lane_xor16_pair(__float_as_uint(mx), ma, mb);
        mx = fmaxf(__uint_as_float(ma), __uint_as_float(mb));
        nrm = sum_xor16(nrm);
#define GR_STEP(ctl_)                                                                                                             \
  mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), ctl_, 0xF, 0xF, true)));  \
  nrm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, nrm), ctl_, 0xF, 0xF, true));
        GR_STEP(0x140) GR_STEP(0x141) GR_STEP(0x4E) GR_STEP(0xB1)   // groups of 8, quads, lane ^ 2, lane ^ 1
#undef GR_STEP
      }
      int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;   // as split_row
      E = mx > 0.f ? min(max(E, -100), 100) : 14;
      sc = __uint_as_float((uint32_t)(127 + 14 - E) << 23);
      if (l32 == 0 && live) {
        g.split.norms[b] = nrm;
        g.split.rscale[b] = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
      }
    }
    const int k4end = split ? max(g.split.ld >> 2, D4) : D4;
    for (int k4 = l32; k4 < k4end; k4 += 32) {
      if (!live) continue;
      const bool in = k4 < D4;
      const float4 x = in ? src[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int k = 4 * k4;
      if (split && k < g.split.ld) {
        const float4 r = in ? ref[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v[4] = {(x.x - r.x) * sc, (x.y - r.y) * sc, (x.z - r.z) * sc, (x.w - r.w) * sc};   // exact scaling
        uint16_t hb[4], lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const _Float16 h = (_Float16)v[i];
          hb[i] = f16_bits(h);
          lb[i] = f16_bits((_Float16)(v[i] - (float)h));
        }
        uint16_t* o = g.split.planes + split_index(g.split, b, k);   // (k % 4 == 0: the four halves are adjacent, 8-byte aligned)
        *reinterpret_cast<uint2*>(o) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
        *reinterpret_cast<uint2*>(o + g.split.plane_stride) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
      }
      if (in) {
        if (g.xb) reinterpret_cast<float4*>(g.xb + (long)b * D)[k4] = x;
        if (g.xin) {
          float4 sn, cs;
          sincos_bounded(x.x * g.ang_scale, &sn.x, &cs.x);
          sincos_bounded(x.y * g.ang_scale, &sn.y, &cs.y);
          sincos_bounded(x.z * g.ang_scale, &sn.z, &cs.z);
          sincos_bounded(x.w * g.ang_scale, &sn.w, &cs.w);
          reinterpret_cast<float4*>(g.xin + (long)b * 2 * D)[k4] = sn;
          reinterpret_cast<float4*>(g.xin + (long)b * 2 * D + D)[k4] = cs;
        }
      }
    }
  }
}

// The same with U row PAIRS of a wave in flight at once, for launches that can spare only a few workgroups for the gather (the
// optimizer as the tail of the weight-gradient launch, mlp.hip): the wave's dependent trips -- cursor, row indices, rows -- serve
// 2 U rows.  Rows of up to 128 features and a split leading dimension of up to 128 (one 16-byte load per lane and row); longer
// rows take gather_rows2_body.  Same arithmetic in the same order: the same bits.
template <int U>
__device__ __forceinline__ void gather_rows2_multi(const GatherArgs& g, int wave, int nwaves, int lane) {
  const int D = g.D, D4 = D >> 2;
  const bool split = g.split.planes != nullptr;
  if (D4 > 32 || (split && (g.split.ld >> 2) > 32)) {
    gather_rows2_body(g, wave, nwaves, lane);
    return;
  }
  const int64_t* __restrict__ idx = g.idx;
  if (g.cursor) {
    const long batch = *g.cursor + g.batch_step;
    if (g.n_batches > 0 && batch >= g.n_batches) return;
    idx += batch * (long)g.B;
  }
  const int half = lane >> 5, l32 = lane & 31;
  const bool in = l32 < D4;
  const int k4end = split ? max(g.split.ld >> 2, D4) : D4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = wave * U; 2 * p0 < g.B; p0 += nwaves * U) {
    long irow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) irow[u] = idx[min(2 * (p0 + u) + half, g.B - 1)];
    const long iref = idx[0];
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = in ? reinterpret_cast<const float4*>(g.data + irow[u] * (long)D)[l32] : zero4;
    const float4 r = (split && in) ? reinterpret_cast<const float4*>(g.data + iref * (long)D)[l32] : zero4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = 2 * (p0 + u) + half;
      const bool live = b < g.B;
      float sc = 1.f;
      if (split) {
        float nrm = 0.f, mx = 0.f;
        if (in) {
          const float v0 = x[u].x - r.x, v1 = x[u].y - r.y, v2 = x[u].z - r.z, v3 = x[u].w - r.w;
          nrm = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, nrm))));
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
        }
        {
          unsigned ma, mb;
          lane_xor16_pair(__float_as_uint(mx), ma, mb);
          mx = fmaxf(__uint_as_float(ma), __uint_as_float(mb));
          nrm = sum_xor16(nrm);
#define GR_STEP(ctl_)                                                                                                             \
  mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), ctl_, 0xF, 0xF, true)));  \
  nrm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, nrm), ctl_, 0xF, 0xF, true));
          GR_STEP(0x140) GR_STEP(0x141) GR_STEP(0x4E) GR_STEP(0xB1)
#undef GR_STEP
        }
        int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;   // as split_row
        E = mx > 0.f ? min(max(E, -100), 100) : 14;
        sc = __uint_as_float((uint32_t)(127 + 14 - E) << 23);
        if (l32 == 0 && live) {
          g.split.norms[b] = nrm;
          g.split.rscale[b] = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
        }
      }
      if (!live || l32 >= k4end) continue;
      const int k = 4 * l32;
      if (split && k < g.split.ld) {
        const float v[4] = {(x[u].x - r.x) * sc, (x[u].y - r.y) * sc, (x[u].z - r.z) * sc, (x[u].w - r.w) * sc};   // exact scaling
        uint16_t hb[4], lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const _Float16 hh = (_Float16)v[i];
          hb[i] = f16_bits(hh);
          lb[i] = f16_bits((_Float16)(v[i] - (float)hh));
        }
        uint16_t* o = g.split.planes + split_index(g.split, b, k);
        *reinterpret_cast<uint2*>(o) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
        *reinterpret_cast<uint2*>(o + g.split.plane_stride) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
      }
      if (in) {
        if (g.xb) reinterpret_cast<float4*>(g.xb + (long)b * D)[l32] = x[u];
        if (g.xin) {
          float4 sn, cs;
          sincos_bounded(x[u].x * g.ang_scale, &sn.x, &cs.x);
          sincos_bounded(x[u].y * g.ang_scale, &sn.y, &cs.y);
          sincos_bounded(x[u].z * g.ang_scale, &sn.z, &cs.z);
          sincos_bounded(x[u].w * g.ang_scale, &sn.w, &cs.w);
          reinterpret_cast<float4*>(g.xin + (long)b * 2 * D)[l32] = sn;
          reinterpret_cast<float4*>(g.xin + (long)b * 2 * D + D)[l32] = cs;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K11 + K10: fused (L2 gradient) -> clip -> Keras Adam on the flat buffer.  float4 lanes.
// ---------------------------------------------------------------------------------------------------
struct AdamArgs {
  float* w;
  const float* g;
  const float* gpart;   // optional: gradient = sum over n_slices slices of gpart[s * gstride + i] (fixed order)
  long gstride;
  int n_slices;
  float* m;
  float* v;
  const float* l2coef;
  const int64_t* step;
  float* ws;
  long n;
  float lr, b1, b2, eps, clip;
  float log2_b1, log2_b2;
  int step_ahead;       // added to the update's index: the launch itself advances *step at its end (emx_mlp_wgrad_adam)
  int late;             // the step's bookkeeping already ran (emx_finish_desc of emx_mlp_wgrad): t = *step, the regulariser shares are
                        // those of the NEW weights (the next step's cost), the next batch is the one under *cursor

  char* mirror;         // optional: weight operand store of the dense layers, rewritten with the new weights (wop.h)
  WopArgs mir;
  int bias_item_begin[EMX_MAX_LAYERS + 1];   // with a store: work items behind the 32 x 32 kernel blocks, 64 bias elements each
  long b_off[EMX_MAX_LAYERS];
  struct Layer {        // what a slab or bias item needs of its layer: one 64-byte scalar load, indexed by the layer found from
    int K, N;           // the two tables above with static indices (a chain of dependent trips to the kernel arguments is
    int item_begin;     // what made this kernel long: it has ~250 instructions of work per wave)
    int bias_item_begin;
    long w_off, b_off, fwd_unit, bwd_unit;
    int layer, slab_begin, slab_end, kind;   // planes: the layer's index, the plane's range of slabs, 0 slabs / 1 bias items
  } lay[EMX_MAX_LAYERS];
  // With a store the optimizer workgroups form PLANES of 64: workgroup b belongs to plane b >> 6, and the plane's record (a
  // copy of its layer's, plus the plane's range of the layer's slabs) sits at an address that depends on the workgroup index
  // alone -- it arrives with the first trip to the kernel arguments, where a table lookup made it a second, dependent one.
  // Layers are given planes in proportion to their slabs; the last plane holds the bias items.
  static constexpr int PLANES = 16;
  static_assert(64 * PLANES <= EMX_REDUCE_WS && PLANES > EMX_MAX_LAYERS, "one regulariser partial per optimizer workgroup; a plane per layer + the bias plane");
  Layer plane[PLANES];
  int n_planes;
  int n_items;          // with a store: slab + bias items
  int adam_blocks;      // workgroups beyond these gather the next batch (next.data != null)
  int next_vec4;        // the rows, xb and xin allow 16-byte lanes: two rows per wave (gather_rows2_body)
  GatherArgs next;

  // optional (emx_adam_step_history): the LAST workgroup of the launch writes the history row of the step whose counters are
  // already advanced -- history[*hstep - 1] = sum(terms), term_history row -- from terms that were final when the launch began
  const float* row_terms;
  int row_n;
  float* history;
  float* term_history;
  long capacity;
  const int64_t* hstep;
};

__device__ __forceinline__ void adam_one(float& w, float g, float& m, float& v, float l2c, const AdamArgs& a, float alpha,
                                         float& reg) {
  if (!a.late) reg = fmaf(0.5f * l2c * w, w, reg);
  g = fmaf(l2c, w, g);
  if (a.clip > 0.f) g = fminf(fmaxf(g, -a.clip), a.clip);
  m = fmaf(g - m, 1.f - a.b1, m);
  v = fmaf(fmaf(g, g, -v), 1.f - a.b2, v);
  w = fmaf(-alpha, m / (sqrtf(v) + a.eps), w);   // (explicit: the flat and the block-wise path of adam_kernel must round alike)
  if (a.late) reg = fmaf(0.5f * l2c * w, w, reg);
}

// SC1: the partials were written by other workgroups of the SAME launch (emx_mlp_wgrad_adam): every load is an sc1 buffer load
// (MI355X_MICROARCH.md, inter-workgroup visibility); same association, same bits
template <bool SC1>
__device__ __forceinline__ float4 slice4(const AdamArgs& a, const __amdgpu_buffer_rsrc_t& rG, int s, long i4) {
  if constexpr (SC1) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ q = __builtin_amdgcn_raw_buffer_load_b128(rG, (int)(i4 * 16), (int)((long)s * a.gstride * 4), 16);
    return make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
  } else {
    return reinterpret_cast<const float4*>(a.gpart + (long)s * a.gstride)[i4];
  }
}
template <bool SC1>
__device__ __forceinline__ float slice1(const AdamArgs& a, const __amdgpu_buffer_rsrc_t& rG, int s, long i) {
  if constexpr (SC1) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, (int)(i * 4), (int)((long)s * a.gstride * 4), 16));
  else return a.gpart[(long)s * a.gstride + i];
}

template <bool SC1 = false>
__device__ __forceinline__ float4 grad4(const AdamArgs& a, long i4) {
  if (!a.gpart) return reinterpret_cast<const float4*>(a.g)[i4];
  const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gpart), 0, 0x7fffffff, 0x00020000);   // (unused without SC1)
  // fixed association (slice s goes to partial s % 4); 32 loads in flight -- the optimizer is a latency-bound stream (a
  // round trip costs ~2.7 us behind the weight-gradient kernel's fresh lines); with 4 in flight the 32 slices of a
  // 4096-row batch were eight dependent round trips
  float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
  int s = 0;
  for (; s + 32 <= a.n_slices; s += 32) {
    float4 q[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) q[u] = slice4<SC1>(a, rG, s + u, i4);
#pragma unroll
    for (int u = 0; u < 32; u += 4) {
      p0.x += q[u].x; p0.y += q[u].y; p0.z += q[u].z; p0.w += q[u].w;
      p1.x += q[u + 1].x; p1.y += q[u + 1].y; p1.z += q[u + 1].z; p1.w += q[u + 1].w;
      p2.x += q[u + 2].x; p2.y += q[u + 2].y; p2.z += q[u + 2].z; p2.w += q[u + 2].w;
      p3.x += q[u + 3].x; p3.y += q[u + 3].y; p3.z += q[u + 3].z; p3.w += q[u + 3].w;
    }
  }
  for (; s + 16 <= a.n_slices; s += 16) {
    float4 q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = slice4<SC1>(a, rG, s + u, i4);
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      p0.x += q[u].x; p0.y += q[u].y; p0.z += q[u].z; p0.w += q[u].w;
      p1.x += q[u + 1].x; p1.y += q[u + 1].y; p1.z += q[u + 1].z; p1.w += q[u + 1].w;
      p2.x += q[u + 2].x; p2.y += q[u + 2].y; p2.z += q[u + 2].z; p2.w += q[u + 2].w;
      p3.x += q[u + 3].x; p3.y += q[u + 3].y; p3.z += q[u + 3].z; p3.w += q[u + 3].w;
    }
  }
  for (; s + 4 <= a.n_slices; s += 4) {
    const float4 q0 = slice4<SC1>(a, rG, s, i4);
    const float4 q1 = slice4<SC1>(a, rG, s + 1, i4);
    const float4 q2 = slice4<SC1>(a, rG, s + 2, i4);
    const float4 q3 = slice4<SC1>(a, rG, s + 3, i4);
    p0.x += q0.x; p0.y += q0.y; p0.z += q0.z; p0.w += q0.w;
    p1.x += q1.x; p1.y += q1.y; p1.z += q1.z; p1.w += q1.w;
    p2.x += q2.x; p2.y += q2.y; p2.z += q2.z; p2.w += q2.w;
    p3.x += q3.x; p3.y += q3.y; p3.z += q3.z; p3.w += q3.w;
  }
  for (; s < a.n_slices; ++s) {
    const float4 q0 = slice4<SC1>(a, rG, s, i4);
    p0.x += q0.x; p0.y += q0.y; p0.z += q0.z; p0.w += q0.w;
  }
  return make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                     (p0.w + p1.w) + (p2.w + p3.w));
}

// one element, with grad4's association (slice s goes to partial s % 4): the same bits whichever path updates an element
template <bool SC1 = false>
__device__ __forceinline__ float grad1(const AdamArgs& a, long i) {
  if (!a.gpart) return a.g[i];
  const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gpart), 0, 0x7fffffff, 0x00020000);   // (unused without SC1)
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= a.n_slices; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] += slice1<SC1>(a, rG, s + u, i);
  }
  for (; s < a.n_slices; ++s) p[0] += slice1<SC1>(a, rG, s, i);
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// One virtual optimizer workgroup (ONE wave) of the store-walking optimizer: workgroup vb belongs to plane vb >> 6 and takes the
// plane's slabs / bias items x64, x64 + 64, ...  Returns the lane's share of the regulariser cost (the caller sums the wave and
// writes ws[vb]).  kargs: the AdamArgs inside the kernel-argument segment (records are fetched with one scalar load each).
template <bool SC1>
__device__ __forceinline__ float adam_store_block(const AdamArgs& a, const void __attribute__((address_space(4))) * kargs, const char* p_mirror,
                                                  const int vb, const int lane, const long step_now, _Float16* img) {
  float reg = 0.f;
  // alpha = lr * sqrt(1 - b2^t) / (1 - b1^t)   (Keras 2.13-2.15 Adam.update_step); b^t = 2^(t log2 b) on the hardware
  // exp2 (the library powf is several hundred instructions in front of the first load of a latency-bound kernel)
  auto alpha_of = [&](long st) {
    const float t = (float)(st + (a.late ? 0 : 1) + a.step_ahead);
    return a.lr * sqrtf(1.f - __builtin_amdgcn_exp2f(t * a.log2_b2)) / (1.f - __builtin_amdgcn_exp2f(t * a.log2_b1));
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(a.w) | reinterpret_cast<uintptr_t>(a.g) | reinterpret_cast<uintptr_t>(a.m) |
                     reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.l2coef) | reinterpret_cast<uintptr_t>(a.gpart) |
                     (uintptr_t)(a.gstride * 4)) & 15) == 0;
  // With a weight operand store the kernels are walked by SLAB (8 rows x 32 columns, one wave each, one float4 per lane
  // and tensor as in the flat pass): the slab's new weights and its part of the f16 pair fragments for both directions
  // leave in the same pass (wop.h).  Lane (kr = lane >> 3, c4 = lane & 7) owns W[8 k8 + kr][32 nb + 4 c4 + i].  Biases
  // follow as items of 64.
  const long tcur = step_now + (a.late ? 0 : 1) + a.step_ahead;   // index of this update
  typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
  typedef const u32x16 __attribute__((address_space(4))) * KVec;
  typedef const char __attribute__((address_space(4))) * KBytes;
  const KBytes kbase = (KBytes)kargs;
  static_assert(sizeof(AdamArgs::Layer) == 64, "one s_load_dwordx16 per record");
  auto record = [&](size_t off, int i) __attribute__((always_inline)) {
    const u32x16 v = *(KVec)(kbase + off + 64 * i);
    AdamArgs::Layer L;
    __builtin_memcpy(&L, &v, 64);
    return L;
  };
  const int n_layers = a.mir.n_layers;
  const int x64 = vb & 63;
  const AdamArgs::Layer PL = record(offsetof(AdamArgs, plane), min(vb >> 6, AdamArgs::PLANES - 1));
  const int n_bias_items = a.n_items - a.mir.item_begin[n_layers];
  for (int it = (PL.kind ? 0 : PL.slab_begin) + x64; it < (PL.kind ? n_bias_items : PL.slab_end); it += 64) {
    if (!PL.kind) {
      const AdamArgs::Layer& L = PL;
      const int l = L.layer;
      const int K = L.K, N = L.N;
      const long woff = L.w_off;
      const int rel = it, NB = groups32(N);
      const int k8 = rel / NB, nb = rel - k8 * NB;
      const int kr = lane >> 3, c4 = lane & 7;
      const int k = 8 * k8 + kr, c = 32 * nb + 4 * c4;
      // the layer's scale: from the largest |w| the PREVIOUS update recorded (wop.h); both slots are requested with the
      // operands (which one is the previous update's depends on the step counter, itself still on its way)
      WopHeader* hdr = wop_header(const_cast<char*>(p_mirror), l);
      if (K > 0) ASTAMP(38, 0);
      const unsigned long long slot0 = hdr->slot[0], slot1 = hdr->slot[1];
      WopArgs wl;          // the layer as layer 0 of a one-layer view: static indices, registers
      wl.n_layers = 1;
      wl.K[0] = K, wl.N[0] = N, wl.w_off[0] = woff, wl.fwd_unit[0] = L.fwd_unit, wl.bwd_unit[0] = L.bwd_unit;
      wl.units_begin = a.mir.units_begin;
      float wv[4];
      if (vec && (N & 3) == 0 && (woff & 3) == 0 && k < K && c + 3 < N) {
        const long i4 = (woff + (long)k * N + c) >> 2;
        float4 w4 = reinterpret_cast<float4*>(a.w)[i4];
        float4 m4 = reinterpret_cast<float4*>(a.m)[i4];
        float4 v4 = reinterpret_cast<float4*>(a.v)[i4];
        const float4 l4 = a.l2coef ? reinterpret_cast<const float4*>(a.l2coef)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = grad4<SC1>(a, i4);
        const float alpha = alpha_of(step_now);
        ASTAMP(33, 0);
        adam_one(w4.x, g4.x, m4.x, v4.x, l4.x, a, alpha, reg);
        adam_one(w4.y, g4.y, m4.y, v4.y, l4.y, a, alpha, reg);
        adam_one(w4.z, g4.z, m4.z, v4.z, l4.z, a, alpha, reg);
        adam_one(w4.w, g4.w, m4.w, v4.w, l4.w, a, alpha, reg);
        reinterpret_cast<float4*>(a.w)[i4] = w4;
        reinterpret_cast<float4*>(a.m)[i4] = m4;
        reinterpret_cast<float4*>(a.v)[i4] = v4;
        wv[0] = w4.x, wv[1] = w4.y, wv[2] = w4.z, wv[3] = w4.w;
      } else {
        const float alpha = alpha_of(step_now);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float wn = 0.f;
          if (k < K && c + i < N) {
            const long e = woff + (long)k * N + c + i;
            float mm = a.m[e], vv = a.v[e];
            wn = a.w[e];
            adam_one(wn, grad1<SC1>(a, e), mm, vv, a.l2coef ? a.l2coef[e] : 0.f, a, alpha, reg);
            a.w[e] = wn;
            a.m[e] = mm;
            a.v[e] = vv;
          }
          wv[i] = wn;
        }
      }
      const unsigned long long prev = (tcur & 1) ? slot0 : slot1, mine_so_far = (tcur & 1) ? slot1 : slot0;
      float ws_, wrs_;
      pow2_scale_bits<WOP_TOP>((unsigned)prev, ws_, wrs_);
      ASTAMP(34, 0);
      const unsigned bmax = wop_write_slab(wl, const_cast<char*>(p_mirror), 0, k8, nb, wv, ws_, lane, img);
      ASTAMP(35, 0);
      if (lane == 0) {
        // record this update's maximum for the next one; most slabs find a larger record of this update already there
        const unsigned long long rec = ((unsigned long long)tcur << 32) | bmax;
        if (rec > mine_so_far) atomicMax(&hdr->slot[tcur & 1], rec);
        if (k8 == 0 && nb == 0) hdr->rs = wrs_;
      }
    } else {
      const int item = a.mir.item_begin[n_layers] + it;
      int l = 0;
#pragma unroll
      for (int i = 1; i < EMX_MAX_LAYERS; ++i) l += (item >= a.bias_item_begin[i]) ? 1 : 0;
      l = min(l, n_layers - 1);
      const AdamArgs::Layer L = record(offsetof(AdamArgs, lay), l);
      const int j = (item - L.bias_item_begin) * 64 + lane;
      if (j < L.N) {
        const long e = L.b_off + j;
        float wn = a.w[e], mm = a.m[e], vv = a.v[e];
        adam_one(wn, grad1<SC1>(a, e), mm, vv, a.l2coef ? a.l2coef[e] : 0.f, a, alpha_of(step_now), reg);
        a.w[e] = wn;
        a.m[e] = mm;
        a.v[e] = vv;
      }
    }
  }
  return reg;
}

// One 8 x 32 slab of a kernel with the gradient in REGISTERS (the weight-gradient launch that sums a block over all rows inside one
// workgroup, mlp.hip mlp_wgrad_local_kernel): the slab's parameters and moments are requested up front (adam_slab_prefetch, at the
// head of the launch) and consumed when the gradient is there (adam_slab_finish: L2 + clip + Adam, the operand store's pair
// fragments, the record of the largest |w| -- the arithmetic of adam_store_block, element for element).
struct AdamSlab {
  float4 w4, m4, v4, l4;
  unsigned long long slot0, slot1;
  long i4;
  bool vec;      // the lane's four elements are one aligned float4 inside the kernel
};
__device__ __forceinline__ AdamSlab adam_slab_prefetch(const AdamArgs& a, const AdamArgs::Layer& L, const char* p_mirror, int k8, int nb, int lane) {
  AdamSlab s;
  const int kr = lane >> 3, c4 = lane & 7;
  const int k = 8 * k8 + kr, c = 32 * nb + 4 * c4;
  const bool al = ((reinterpret_cast<uintptr_t>(a.w) | reinterpret_cast<uintptr_t>(a.m) | reinterpret_cast<uintptr_t>(a.v) |
                    reinterpret_cast<uintptr_t>(a.l2coef)) & 15) == 0;
  s.vec = al && (L.N & 3) == 0 && (L.w_off & 3) == 0 && k < L.K && c + 3 < L.N;
  s.i4 = (L.w_off + (long)k * L.N + c) >> 2;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  s.w4 = s.m4 = s.v4 = s.l4 = z;
  if (s.vec) {
    s.w4 = reinterpret_cast<const float4*>(a.w)[s.i4];
    s.m4 = reinterpret_cast<const float4*>(a.m)[s.i4];
    s.v4 = reinterpret_cast<const float4*>(a.v)[s.i4];
    if (a.l2coef) s.l4 = reinterpret_cast<const float4*>(a.l2coef)[s.i4];
  }
  const WopHeader* hdr = wop_header(const_cast<char*>(p_mirror), L.layer);
  s.slot0 = hdr->slot[0];
  s.slot1 = hdr->slot[1];
  return s;
}
__device__ __forceinline__ float adam_slab_finish(const AdamArgs& a, const AdamArgs::Layer& L, const char* p_mirror, AdamSlab& s, const float (&g)[4],
                                                  int k8, int nb, int lane, long step_now, _Float16* img) {
  float reg = 0.f;
  const float t = (float)(step_now + (a.late ? 0 : 1) + a.step_ahead);
  const float alpha = a.lr * sqrtf(1.f - __builtin_amdgcn_exp2f(t * a.log2_b2)) / (1.f - __builtin_amdgcn_exp2f(t * a.log2_b1));
  const long tcur = step_now + (a.late ? 0 : 1) + a.step_ahead;
  const int K = L.K, N = L.N;
  const int kr = lane >> 3, c4 = lane & 7;
  const int k = 8 * k8 + kr, c = 32 * nb + 4 * c4;
  float wv[4];
  if (s.vec) {
    adam_one(s.w4.x, g[0], s.m4.x, s.v4.x, s.l4.x, a, alpha, reg);
    adam_one(s.w4.y, g[1], s.m4.y, s.v4.y, s.l4.y, a, alpha, reg);
    adam_one(s.w4.z, g[2], s.m4.z, s.v4.z, s.l4.z, a, alpha, reg);
    adam_one(s.w4.w, g[3], s.m4.w, s.v4.w, s.l4.w, a, alpha, reg);
    reinterpret_cast<float4*>(a.w)[s.i4] = s.w4;
    reinterpret_cast<float4*>(a.m)[s.i4] = s.m4;
    reinterpret_cast<float4*>(a.v)[s.i4] = s.v4;
    wv[0] = s.w4.x, wv[1] = s.w4.y, wv[2] = s.w4.z, wv[3] = s.w4.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float wn = 0.f;
      if (k < K && c + i < N) {
        const long e = L.w_off + (long)k * N + c + i;
        float mm = a.m[e], vv = a.v[e];
        wn = a.w[e];
        adam_one(wn, g[i], mm, vv, a.l2coef ? a.l2coef[e] : 0.f, a, alpha, reg);
        a.w[e] = wn;
        a.m[e] = mm;
        a.v[e] = vv;
      }
      wv[i] = wn;
    }
  }
  WopHeader* hdr = wop_header(const_cast<char*>(p_mirror), L.layer);
  WopArgs wl;          // the layer as layer 0 of a one-layer view: static indices, registers
  wl.n_layers = 1;
  wl.K[0] = K, wl.N[0] = N, wl.w_off[0] = L.w_off, wl.fwd_unit[0] = L.fwd_unit, wl.bwd_unit[0] = L.bwd_unit;
  wl.units_begin = a.mir.units_begin;
  const unsigned long long prev = (tcur & 1) ? s.slot0 : s.slot1, mine_so_far = (tcur & 1) ? s.slot1 : s.slot0;
  float ws_, wrs_;
  pow2_scale_bits<WOP_TOP>((unsigned)prev, ws_, wrs_);
  const unsigned bmax = wop_write_slab(wl, const_cast<char*>(p_mirror), 0, k8, nb, wv, ws_, lane, img);
  if (lane == 0) {
    const unsigned long long rec = ((unsigned long long)tcur << 32) | bmax;
    if (rec > mine_so_far) atomicMax(&hdr->slot[tcur & 1], rec);
    if (k8 == 0 && nb == 0) hdr->rs = wrs_;
  }
  return reg;
}

// ---- host side ------------------------------------------------------------------------------------------------
inline bool gather_vec4(const emx_next_batch* next) {
  return next && next->D % 4 == 0 && ((reinterpret_cast<uintptr_t>(next->data) | reinterpret_cast<uintptr_t>(next->xb) |
                                       reinterpret_cast<uintptr_t>(next->xin)) & 15) == 0;
}

inline int check_next(const emx_next_batch* next, const char* who) {
  if (!next) return EMX_OK;
  if (next->adc_head) {
    EMX_REQUIRE(next->idx && next->cursor && next->B > 0 && next->n_batches > 0 && next->D >= 0, "%s: incomplete next-batch description (ADC head)", who);
    return EMX_OK;
  }
  EMX_REQUIRE(next->data && next->idx && next->cursor && (next->xb || next->xin), "%s: incomplete next-batch description", who);
  EMX_REQUIRE(next->n_rows > 0 && next->D > 0 && next->B > 0 && next->n_batches > 0, "%s: bad next-batch shape", who);
  EMX_REQUIRE(!next->xin || (std::isfinite(next->periodicity) && next->periodicity > 0.f), "%s: sin/cos rows need a finite periodicity", who);
  return EMX_OK;
}


inline AdamArgs make_adam(float* w, const float* g, const float* gpart, int n_slices, int64_t gstride, float* m, float* v, const float* l2coef,
                   int64_t n, float lr, float beta1, float beta2, float eps, float clip, const int64_t* step, float* ws,
                   const emx_mlp_desc* mirror_desc, float* mirror, const emx_next_batch* next  = nullptr) {
  AdamArgs a{};
  if (next && next->adc_head) {
    a.next = GatherArgs{};
    a.next.idx = next->idx;
    a.next.cursor = next->cursor;
    a.next.B = next->B;
    a.next.batch_step = 1;
    a.next.n_batches = next->n_batches;
    a.next.adc_head = static_cast<const emx::AdcHeadArgs*>(next->adc_head);
  } else if (next) {
    const bool periodic = next->xin != nullptr;
    a.next = GatherArgs{next->data, next->D, next->idx, next->cursor, next->B, 1, next->n_batches,
                        periodic ? (next->periodicity == 6.283185307179586f ? 1.f : 6.283185307179586f / next->periodicity) : 1.f, next->xb, next->xin, SplitRows{}};
    if (next->split_ws) a.next.split = split_view(next->split_ws, next->B, next->D);
    a.next_vec4 = gather_vec4(next) ? 1 : 0;
  }
  a.mirror = (mirror_desc && mirror) ? reinterpret_cast<char*>(mirror) : nullptr;
  if (a.mirror) {
    fill_wop_args(mirror_desc, a.mir);
    int items = a.mir.item_begin[mirror_desc->n_layers];
    for (int l = 0; l <= EMX_MAX_LAYERS; ++l) {
      a.bias_item_begin[l] = items;
      if (l < mirror_desc->n_layers) {
        a.b_off[l] = mirror_desc->b_off[l];
        a.lay[l] = AdamArgs::Layer{a.mir.K[l], a.mir.N[l], a.mir.item_begin[l], items, a.mir.w_off[l], (long)mirror_desc->b_off[l],
                                   a.mir.fwd_unit[l], a.mir.bwd_unit[l], l, 0, 0, 0};
        items += (mirror_desc->width[l + 1] + 63) / 64;
      }
    }
    a.n_items = items;
    for (int l = mirror_desc->n_layers + 1; l <= EMX_MAX_LAYERS; ++l) a.mir.item_begin[l] = INT_MAX;
    {
      // planes: one per layer first, the rest of the PLANES - 1 slab planes to the layers with the most slabs per plane
      const int nl = mirror_desc->n_layers;
      int cnt[EMX_MAX_LAYERS], slabs[EMX_MAX_LAYERS];
      for (int l = 0; l < nl; ++l) cnt[l] = 1, slabs[l] = a.mir.item_begin[l + 1] == INT_MAX ? 0 : 0;
      for (int l = 0; l < nl; ++l) slabs[l] = ((a.mir.K[l] + 7) >> 3) * groups32(a.mir.N[l]);
      for (int extra = AdamArgs::PLANES - 1 - nl; extra > 0; --extra) {
        int best = -1;
        for (int l = 0; l < nl; ++l)
          if (slabs[l] > 64 * cnt[l] && (best < 0 || (long)slabs[l] * cnt[best] > (long)slabs[best] * cnt[l])) best = l;
        if (best < 0) break;
        ++cnt[best];
      }
      int np = 0;
      for (int l = 0; l < nl; ++l)
        for (int c = 0; c < cnt[l]; ++c) {
          AdamArgs::Layer& P = a.plane[np++];
          P = a.lay[l];
          P.slab_begin = (int)((long)slabs[l] * c / cnt[l]);
          P.slab_end = (int)((long)slabs[l] * (c + 1) / cnt[l]);
          P.kind = 0;
        }
      AdamArgs::Layer& Pb = a.plane[np++];
      Pb = a.lay[0];
      Pb.kind = 1;
      a.n_planes = np;
    }
  }
  a.w = w;
  a.g = g;
  a.gpart = gpart;
  a.gstride = gstride;
  a.n_slices = n_slices;
  a.m = m;
  a.v = v;
  a.l2coef = l2coef;
  a.step = step;
  a.ws = ws;
  a.n = n;
  a.lr = lr;
  a.b1 = beta1;
  a.b2 = beta2;
  a.log2_b1 = (float)std::log2((double)beta1);
  a.log2_b2 = (float)std::log2((double)beta2);
  a.eps = eps;
  a.clip = clip;
  return a;
}

}  // namespace emx
