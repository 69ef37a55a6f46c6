// Inference: encode() / decode() / generate() as ONE streaming launch (gfx950).
//   replaces: Autoencoder.encode / generate / decode   encodermap/autoencoder/autoencoder.py:1110-1195
//             SequentialModel.encoder / decoder         encodermap/models/models.py:3331-3359  (sin / cos in, atan2 out)
//             tf.keras.layers.Dense forward             encodermap/models/models.py:3179-3220
//
// What the pass is bound by.  Whole trajectories go through it (1e6 .. 1e7 rows): per row 4 D bytes come in and 4 L go out
// (encode; the decoder mirrored), against ~2 (D 128 + 128 128 + 128 L) multiply-adds -- at D = 128 the HBM floor of 1e6 rows is
// 64 us, the floor of the split-f16 matrix products (three per float32 product) 80 us: the two roofs meet, and nothing but the
// input rows and the result may touch HBM.  The training kernels are built for 16-row blocks whose activations and gradients leave
// as fragments for the weight gradients; run forward-only on 1e6 rows they stream the 132-KB operand store from L2 once per row
// block (0.98 ms per 1e6 rows).
//
// The kernel.  One 1024-thread workgroup per compute unit.  (1) The workgroup builds the network's operand pieces IN ITS LDS, from
// the float32 parameters themselves (no operand store, no workspace): per layer one power-of-two scale (the layer's largest |w| to
// [2^12, 2^13), as csrc/wop.h), two f16 planes, in 1-KB pieces of v_mfma_f32_16x16x32_f16 -- piece (tile t of 16 out-features,
// K-step s of 32 inputs, plane): lane (m, g) holds W[32 s + kk(g, e)][16 t + m], e < 8 -- 136 KB for 128 -> 128 -> 128 -> 2.
// (2) Every WAVE then walks 16-row blocks on its own, transposed (H^T = W^T X^T, the picture of csrc/tchain.h): the weights are the
// A operand, the 16 rows the N dimension, lane (row n, g) of a D tile holds out-features 16 t + 4 g + e -- with kk(g, e) = 4 g + e,
// 16 + 4 g + (e - 4) the accumulators of tiles 2 s, 2 s + 1 ARE the lane's eight k of K-step s of the next layer.  Products run
// K-step-outer / tile-inner into all of a layer's accumulators (32 registers), so that the layer's operand is dead when the epilogue
// starts and the next one -- bias, activation, f16 split -- is written IN PLACE, in registers: one operand set, ~105 registers a
// wave, FOUR waves per SIMD; a block's rows never touch LDS or HBM between the input load and the result store, and no wave ever
// waits for another (no barrier after (1)).  A periodic input (PeriodicInput, layers.py:204-215) is formed on chip -- the k order of
// layer 0 is then the interleaved one (sin x_0, cos x_0, sin x_1, ...), so that a lane's eight k are four raw values; a periodic
// output (PeriodicOutput, layers.py:1118-1125) likewise: the out-features of the last layer are interleaved, atan2 runs on
// register pairs.
//
// Why four waves per SIMD and not hand-laid streams.  The first form of this kernel (profiles/r06_infer_notes.txt; kept as
// tools/experiments/infer_form1_32rows.patch) ran 32-row blocks on v_mfma_f32_32x32x16_f16 with two operand sets and the next
// block's rows in flight: 350-500 registers, ONE wave per SIMD, its instruction stream laid out by hand (operand pieces two
// K-steps ahead, the epilogue software-pipelined between the products).  One wave issues a vector instruction every ~8.8 cycles
// whatever the stream looks like; it ended at 0.38 ms per 1e6 rows.  This form: 0.30 ms (0.25 ms of kernel; counters:
// profiles/r06_infer_counters.json) -- VALUBusy 59 % + MfmaUtil 40 %: the two pipes of a SIMD add up rather than overlap, so what is
// left is their sum: ~50 vector instructions (16 of them exp2 / rcp) and 12.75 matrix instructions per row.
//
// Arithmetic: the dense layers' precision class (split-f16 x 3 products, float32 accumulation; one operand scale per row), another
// order of summation than the training kernels: results agree with them to float32 rounding, not bit for bit.
// Scope: every layer of the range at most 128 wide (2 x 64 raw features for a periodic input / output) and the pieces of the range
// within LDS -- or ONE wide end (a first layer of any number of inputs, or a last layer of any number of out-features) on the
// standard shape [wide, tanh 65..128, tanh 65..128, last]: infer16_wide_kernel below stages the wide layer's pieces chunk by chunk.
// Anything else answers EMX_EUNSUPPORTED and the caller runs emx_mlp_forward.
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>

#include "act.h"
#include "emx_common.h"
#include "wop.h"

namespace {

using namespace emx;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IN_MAXL = EMX_MAX_LAYERS;
constexpr int IN_LDS_MAX = 160 * 1024;
constexpr float IN_TWO_PI = 6.283185307179586f;
constexpr float IN_TANH_ARG = 2.8853900817779268f;   // 2 log2(e): tanh(x) = 1 - 2 / (2^(x 2 log2 e) + 1)

struct InLayer {
  long w_off, b_off;   // float offsets of the [K, N] kernel and the bias inside the parameter buffer
  int K, N;            // the layer as the network has it
  int S, T;            // K-steps of 16 and tiles of 32 out-features the kernel runs (zero-padded)
  int pieces;          // byte offset of the layer's operand pieces in LDS: [tile][K-step][plane] x 1 KB
  int act;
  int kperm, nperm;    // the k / out-feature order is the interleaved one of a periodic input / output
};
struct InArgs {
  InLayer L[IN_MAXL];
  const float* params;
  const float* in;
  float* out;
  long ld_in, ld_out, n_rows, n_blocks;
  int n_layers, in_cols, out_cols;   // in_cols / out_cols: columns of `in` / `out` as the caller has them (raw features where periodic)
  float in_scale, out_scale;         // periodic input: 2 pi / period (0: none); periodic output: period / 2 pi (0: none)
  int bias_tab, scale_tab;           // LDS byte offsets: [layer][128] floats; [layer] floats, then 8 words of scratch
  int vec_in, vec_out;               // floats per memory instruction the input / output rows allow (4, 2 or 1)
  int wide_in, wide_out;             // the first layer has more than 128 inputs / the last more than 128 out-features: their pieces are STAGED
  int stage;                         // LDS byte offset of the two stage buffers (infer16_wide_kernel)
};
// (the record is indexed at run time -- layer tables -- so it is read through the kernel-argument segment itself: indexing the
//  by-value struct makes hipcc copy all of it to scratch)
typedef const InArgs __attribute__((address_space(4))) * InK;

// interleaved order of a periodic tensor of 2 n columns [sin | cos]: virtual index v -> column
__device__ __forceinline__ int in_unperm(int v, int n) { return (v & 1) ? n + (v >> 1) : (v >> 1); }

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#ifdef IN_STAMP
// diagnostics build (tools/build_infer_variant.sh ... "-DIN_STAMP"; tools/infer_stamps.py): s_memtime stamps of workgroup 0's waves,
// [wave][block iteration][slot]; slot 0 loop top, 1 operand made, 2 next rows requested, 3 + l layer l done, 7 rows stored; iteration 15: kernel
// entry (slot 0) and pieces built (slot 1).  The stamp values go to a buffer of their own; no output is computed from them.
__device__ unsigned long long g_in_stamps[8][16][8];
#define INS(it, k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && lane == 0 && (it) < 16) g_in_stamps[wave][it][k] = t_; } while (0)
#else
#define INS(it, k) do {} while (0)
#endif
constexpr int IN2_NT = 1024, IN2_NW = IN2_NT / 64, IN2_ROWS = 16;
__host__ __device__ inline int in2_kk(int g, int e) { return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4); }
__host__ __device__ inline int in2_class_S(int K) { return K <= 32 ? 1 : (K <= 64 ? 2 : 4); }           // K-steps of 32
__host__ __device__ inline int in2_class_T(int N) { return N <= 16 ? 1 : (N <= 32 ? 2 : (N <= 64 ? 4 : 8)); }   // tiles of 16

typedef float in2_f2 __attribute__((ext_vector_type(2)));

// two float32 values -> the operand's two f16 planes (elements i, i + 1 of a K-step)
__device__ __forceinline__ void in2_split2(const float v0, const float v1, half8& oh, half8& ol, const int i) {
  unsigned ph, pl;
  split_f16_pair(v0, v1, ph, pl);   // (three instructions a pair: emx_common.h)
  const half2v vh = __builtin_bit_cast(half2v, ph), vl = __builtin_bit_cast(half2v, pl);
  oh[i] = vh[0], oh[i + 1] = vh[1], ol[i] = vl[0], ol[i + 1] = vl[1];
}

// One layer of class (S, T) on the wave's 16 rows, products AND epilogue with every bound a compile-time constant: run-time tile
// guards put each of the 64 operand / accumulator registers behind joins of their own, and the joins became ~45 v_mov per row (94
// vector instructions per row where the arithmetic is 45: profiles/r06_infer_counters.json).
//   kind 0: a hidden tanh layer -- the next operand in place, at the fixed scale 2^14 (bias table and cm carry 2 log2 e)
//   kind 1: a hidden layer of another activation -- float32 rows, activation, the row's power-of-two scale, the split
//   kind 2: the last layer -- the activated float32 rows are left in `res` for the caller's stores
template <int S, int T>
__device__ __forceinline__ void in2_products(const char* __restrict__ pieces /* + lane * 16 */, const half8 (&bh)[4], const half8 (&bl)[4], f32x4 (&acc)[T]) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const char* p = pieces + (t * S + s) * 2048;
      const half8 ah = *reinterpret_cast<const half8*>(p), al = *reinterpret_cast<const half8*>(p + 1024);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], s == 0 ? zero : acc[t], 0, 0, 0);   // small products first
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc[t], 0, 0, 0);
      // (a fence behind every second tile: left alone the scheduler hoists a K-step's sixteen operand reads -- 64 registers -- in front
      //  of its products, and the kernel needs 229 registers where four waves per SIMD have 128; the other waves cover the reads)
      if (t & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// bias, activation and what follows by the layer's kind, from the raw accumulators of its T tiles
template <int T, int KIND = -1>
__device__ __forceinline__ void in2_epilogue(f32x4 (&acc)[T], const float* __restrict__ bt /* bias + 4 g */, const float cm, const int kind_rt,
                                             const int act, half8 (&bh)[4], half8 (&bl)[4], float& rs_b, f32x4 (&res)[8]) {
  const int kind = KIND >= 0 ? KIND : kind_rt;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + 16 * t);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(acc[t][e], cm, b4[e]);
  }
  constexpr int SN = (T + 1) / 2;          // K-steps of the next layer these tiles make
  if (kind == 0) {
    // tanh(x) 2^14 = 2^14 - 2^15 / (2^(x') + 1); tiles 2 s, 2 s + 1 are K-step s of the next layer
    rs_b = TANH_RS;
#pragma unroll
    for (int s2 = 0; s2 < SN; ++s2) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const int t = 2 * s2 + (i >> 2);
        float v0 = 0.f, v1 = 0.f;          // (an odd tile count: the second half of the last K-step does not exist)
        if (t < T) {
          v0 = fmaf(-32768.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[t][i & 3]) + 1.f), 16384.f);
          v1 = fmaf(-32768.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[t][(i & 3) + 1]) + 1.f), 16384.f);
        }
        in2_split2(v0, v1, bh[s2], bl[s2], i);
      }
    }
    return;
  }
  if (act != ACT_NONE) {
    // (the switch outside the element loops)
#define IN2_ACT(K_, EXPR_)                                                                       \
  case K_:                                                                                       \
    _Pragma("unroll") for (int t = 0; t < T; ++t)                                                \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float xv = acc[t][e]; acc[t][e] = EXPR_; } \
    break;
    switch (act) {
      IN2_ACT(ACT_TANH, tanh_f32(xv))
      IN2_ACT(ACT_RELU, act_apply(ACT_RELU, xv))
      IN2_ACT(ACT_SIGMOID, act_apply(ACT_SIGMOID, xv))
      IN2_ACT(ACT_ELU, act_apply(ACT_ELU, xv))
      IN2_ACT(ACT_SOFTPLUS, act_apply(ACT_SOFTPLUS, xv))
      IN2_ACT(ACT_SELU, act_apply(ACT_SELU, xv))
      IN2_ACT(ACT_SOFTSIGN, act_apply(ACT_SOFTSIGN, xv))
      IN2_ACT(ACT_EXP, act_apply(ACT_EXP, xv))
      default: break;
    }
#undef IN2_ACT
  }
  if (kind == 2) {
#pragma unroll
    for (int t = 0; t < 8; ++t) res[t] = t < T ? acc[t] : zero;
    return;
  }
  // the next layer's operand from the float32 rows: the row's power-of-two scale first
  float sc = 16384.f;
  rs_b = TANH_RS;
  if (!act_bounded(act)) {
    float mf = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) mf = __builtin_fmaxf(mf, __builtin_fabsf(acc[t][e]));
    const unsigned m = max_xor32(max_xor16(abs_bits(mf)));
    pow2_scale_bits<ACT_TOP>(m, sc, rs_b);
  }
#pragma unroll
  for (int s2 = 0; s2 < SN; ++s2) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int t = 2 * s2 + (i >> 2);
      in2_split2(t < T ? acc[t][i & 3] * sc : 0.f, t < T ? acc[t][(i & 3) + 1] * sc : 0.f, bh[s2], bl[s2], i);
    }
  }
}
template <int S, int T, int KIND = -1>   // KIND >= 0: the layer's kind at compile time (the run-time argument is ignored)
__device__ __forceinline__ void in2_layer(const char* __restrict__ pieces /* + lane * 16 */, const float* __restrict__ bt /* bias + 4 g */,
                                          const float cm, const int kind_rt, const int act, half8 (&bh)[4], half8 (&bl)[4], float& rs_b,
                                          f32x4 (&res)[8]) {
  f32x4 acc[T];
  in2_products<S, T>(pieces, bh, bl, acc);
  in2_epilogue<T, KIND>(acc, bt, cm, kind_rt, act, bh, bl, rs_b, res);
}

// the block's rows as the first layer's operand: lane (n, g) takes the eight k of each of the S0 K-steps (compile time: see in2_layer)
template <bool PER_IN, int S0>
__device__ __forceinline__ void in2_input(const float* __restrict__ rp, const int g, const int in_cols, const int vec_in, const float in_scale,
                                          half8 (&bh)[4], half8 (&bl)[4], float& rs_b) {
  float x[S0][8];
  // (ONE test of the access width around all the requests: a test per request puts every value behind a join of its own, and the
  //  register allocator answers with a wait and a scratch store behind each load)
  if (vec_in == (PER_IN ? 2 : 4)) {
#pragma unroll
    for (int s = 0; s < S0; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (PER_IN) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(rp + min(16 * s + 8 * j + 2 * g, in_cols - 2));   // raw columns c, c + 1
          x[s][4 * j] = v[0], x[s][4 * j + 1] = v[1], x[s][4 * j + 2] = 0.f, x[s][4 * j + 3] = 0.f;
        } else {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rp + min(32 * s + 16 * j + 4 * g, in_cols - 4));   // columns c .. c + 3
#pragma unroll
          for (int e = 0; e < 4; ++e) x[s][4 * j + e] = v[e];
        }
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < S0; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (PER_IN) {
          const int c = 16 * s + 8 * j + 2 * g;
          x[s][4 * j] = rp[min(c, in_cols - 1)], x[s][4 * j + 1] = rp[min(c + 1, in_cols - 1)], x[s][4 * j + 2] = 0.f, x[s][4 * j + 3] = 0.f;
        } else {
          const int c = 32 * s + 16 * j + 4 * g;
#pragma unroll
          for (int e = 0; e < 4; ++e) x[s][4 * j + e] = rp[min(c + e, in_cols - 1)];
        }
      }
    }
  }
  if constexpr (PER_IN) {
    rs_b = TANH_RS;                                   // sines and cosines: the fixed scale 2^14
#pragma unroll
    for (int s = 0; s < S0; ++s) {
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {                                // raw value pq of the step: half pq >> 1, element pq & 1
        const int c = 16 * s + 8 * (pq >> 1) + 2 * g + (pq & 1);
        float sn, cs;
        sincos_bounded(x[s][4 * (pq >> 1) + (pq & 1)] * in_scale, &sn, &cs);
        sn = c < in_cols ? sn * 16384.f : 0.f;
        cs = c < in_cols ? cs * 16384.f : 0.f;
        in2_split2(sn, cs, bh[s], bl[s], 4 * (pq >> 1) + 2 * (pq & 1));
      }
    }
  } else {
    float mf = 0.f;
    if (in_cols < 32 * S0) {                            // (wave-uniform: rows that fill their K-steps need no masks)
#pragma unroll
      for (int s = 0; s < S0; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[s][i] = (32 * s + in2_kk(g, i)) < in_cols ? x[s][i] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < S0; ++s)
#pragma unroll
      for (int i = 0; i < 8; i += 2) mf = __builtin_fmaxf(mf, __builtin_fmaxf(__builtin_fabsf(x[s][i]), __builtin_fabsf(x[s][i + 1])));
    const unsigned m = max_xor32(max_xor16(abs_bits(mf)));          // the row's four lanes
    float sc;
    pow2_scale_bits<ACT_TOP>(m, sc, rs_b);
#pragma unroll
    for (int s = 0; s < S0; ++s)
#pragma unroll
      for (int i = 0; i < 8; i += 2) in2_split2(x[s][i] * sc, x[s][i + 1] * sc, bh[s], bl[s], i);
  }
}

// atan2 for the periodic outputs: a = min / max of the magnitudes in [0, 1] (v_rcp_f32: 1 ulp), atan(a) = a P(a^2) with a degree-8
// polynomial fitted at Chebyshev nodes (error 1.2e-8 in exact arithmetic), octant and sign by selects -- 24 instructions where the
// library's atan2f is ~45; the largest error against float64 over the plane and along the unit circle is 3e-7 rad (one ulp of pi),
// atan2(0, 0) = 0 as numpy has it.
__device__ __forceinline__ float in_atan2(const float y, const float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
  const float a = mn * __builtin_amdgcn_rcpf(__builtin_fmaxf(mx, 1e-37f));
  const float z = a * a;
  float p = 2.8340642988e-03f;
  p = fmaf(p, z, -1.6005030503e-02f);
  p = fmaf(p, z, 4.2587607465e-02f);
  p = fmaf(p, z, -7.4954454434e-02f);
  p = fmaf(p, z, 1.0636754098e-01f);
  p = fmaf(p, z, -1.4202570512e-01f);
  p = fmaf(p, z, 1.9992483579e-01f);
  p = fmaf(p, z, -3.3333066781e-01f);
  p = fmaf(p, z, 9.9999998424e-01f);
  float r = a * p;
  r = ay > ax ? 1.5707963267948966f - r : r;
  r = x < 0.f ? 3.14159265358979f - r : r;
  return __builtin_copysignf(r, y);
}

// the result rows: lane (n, g) holds out-features 16 t + 4 g + e of row n; tiles the last layer does not have hold zeros and fall to the
// per-lane column tests.  orow: the lane's output row (nullptr: the row does not exist)
__device__ __forceinline__ void in2_store(const f32x4 (&res)[8], float* const orow, const int g, const int oc, const int vec_out, const float out_scale) {
  if (orow == nullptr) return;
  if (out_scale != 0.f) {
    // PeriodicOutput: virtual out-features (2 k, 2 k + 1) = (sin, cos) of angle k = 8 t + 2 g (+ 1)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (8 * t < oc) {                                  // (wave-uniform: no atan2 for tiles beyond the row)
        const int k = 8 * t + 2 * g;
        const float a0 = in_atan2(res[t][0], res[t][1]) * out_scale;
        const float a1 = in_atan2(res[t][2], res[t][3]) * out_scale;
        if (vec_out >= 2) {
          if (k < oc) *reinterpret_cast<f32x2*>(orow + k) = f32x2{a0, a1};
        } else {
          if (k < oc) orow[k] = a0;
          if (k + 1 < oc) orow[k + 1] = a1;
        }
      }
    }
  } else if (vec_out == 4) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (16 * t + 4 * g < oc) *reinterpret_cast<f32x4*>(orow + 16 * t + 4 * g) = res[t];
  } else if (vec_out == 2) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = 16 * t + 4 * g;
      if (c < oc) *reinterpret_cast<f32x2*>(orow + c) = f32x2{res[t][0], res[t][1]};
      if (c + 2 < oc) *reinterpret_cast<f32x2*>(orow + c + 2) = f32x2{res[t][2], res[t][3]};
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (16 * t + 4 * g + e < oc) orow[16 * t + 4 * g + e] = res[t][e];
  }
}

// the operand pieces, biases and weight scales of every layer of the range in LDS, from the float32 parameters (all threads of the workgroup)
__device__ __forceinline__ void in2_build(const InK ka, char* lds) {
  const auto& a = *ka;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = a.n_layers;
  float* const bias_tab = reinterpret_cast<float*>(lds + a.bias_tab);
  float* const scale_tab = reinterpret_cast<float*>(lds + a.scale_tab);
  unsigned* const red = reinterpret_cast<unsigned*>(scale_tab + IN_MAXL);
  for (int l = 0; l < nl; ++l) {
    const int K = a.L[l].K, N = a.L[l].N, S = a.L[l].S, T = a.L[l].T;
    const int kperm = a.L[l].kperm, nperm = a.L[l].nperm;
    const float* __restrict__ W = a.params + a.L[l].w_off;
    const float* __restrict__ bias = a.params + a.L[l].b_off;
    unsigned mx = 0;
    for (int e = tid; e < K * N; e += IN2_NT) mx = max(mx, abs_bits(W[e]));
    mx = wave_max_u32(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < IN2_NW; ++w) mx = max(mx, red[w]);
    float sc, rs;
    pow2_scale_bits<WOP_TOP>(mx, sc, rs);
    if (tid == 0) scale_tab[l] = rs;
    char* const pieces = lds + a.L[l].pieces;
    const bool staged = (l == 0 && a.wide_in) || (l == nl - 1 && a.wide_out);   // (chunk by chunk, inside the row loop)
    for (int task = tid; task < (staged ? 0 : T * S * 64); task += IN2_NT) {
      const int ln = task & 63, ts = task >> 6;
      const int t = ts / S, s = ts - t * S;
      const int pm = ln & 15, pg = ln >> 4;
      const int vf = 16 * t + pm;
      const int f = nperm ? in_unperm(vf, N >> 1) : vf;
      half8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int vk = 32 * s + in2_kk(pg, i);
        const int k = kperm ? in_unperm(vk, K >> 1) : vk;
        const bool ok = (vf < N) & (vk < K);
        const float w = ok ? W[(long)k * N + f] : 0.f;
        const float v = w * sc;
        hi[i] = (_Float16)v;
        lo[i] = (_Float16)(v - (float)hi[i]);
      }
      char* p = pieces + (long)ts * 2048 + ln * 16;
      *reinterpret_cast<half8*>(p) = hi;
      *reinterpret_cast<half8*>(p + 1024) = lo;
    }
    const float bscale = (l + 1 < nl && a.L[l].act == ACT_TANH) ? IN_TANH_ARG : 1.f;
    for (int vf = tid; vf < 128; vf += IN2_NT) {   // (a wide last layer reads its biases with the staged chunks)
      const int f = nperm ? in_unperm(vf, N >> 1) : vf;
      bias_tab[l * 128 + vf] = vf < N ? bias[min(f, N - 1)] * bscale : 0.f;
    }
    __syncthreads();
  }

}

template <bool PER_IN>
__global__ __launch_bounds__(IN2_NT) void infer16_kernel(const InArgs) {
  extern __shared__ __align__(16) char lds[];
  const InK ka = (InK)__builtin_amdgcn_kernarg_segment_ptr();
  const auto& a = *ka;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g0 = lane >> 4;
  const int nl = a.n_layers;
  float* const bias_tab = reinterpret_cast<float*>(lds + a.bias_tab);
  float* const scale_tab = reinterpret_cast<float*>(lds + a.scale_tab);
  unsigned* const red = reinterpret_cast<unsigned*>(scale_tab + IN_MAXL);

  in2_build(ka, lds);

  // ---- (2) every wave for itself: 16-row blocks
  const long n_rows = a.n_rows, n_blocks = a.n_blocks;
  const long stride = (long)gridDim.x * IN2_NW;
  int in_cols = a.in_cols, vec_in = a.vec_in, out_cols = a.out_cols, vec_out = a.vec_out;
  const long ld_in = a.ld_in, ld_out = a.ld_out;
  const float* const in = a.in;
  float* const out = a.out;
  const float in_scale = a.in_scale, out_scale = a.out_scale;
  const int S0 = a.L[0].S;

  int it = 0;
  for (long blk = (long)blockIdx.x * IN2_NW + wave; blk < n_blocks; blk += stride, ++it) {
    if (wave < 8) INS(it, 0);
    // (the column limits -- and the lane's column group -- as values the compiler cannot prove loop-invariant: see infer_kernel; here the
    //  hoisted per-lane column numbers of the masks alone were 32 registers, spilled)
    asm volatile("" : "+s"(in_cols), "+s"(out_cols), "+s"(vec_in), "+s"(vec_out));
    int g = g0;
    asm volatile("" : "+v"(g));
    const long row = blk * IN2_ROWS + n;
    const float* rp = in + min(row, n_rows - 1) * ld_in;
    half8 bh[4], bl[4];
    float rs_b;
    if (S0 == 4) in2_input<PER_IN, 4>(rp, g, in_cols, vec_in, in_scale, bh, bl, rs_b);
    else if (S0 == 2) in2_input<PER_IN, 2>(rp, g, in_cols, vec_in, in_scale, bh, bl, rs_b);
    else in2_input<PER_IN, 1>(rp, g, in_cols, vec_in, in_scale, bh, bl, rs_b);
    if (wave < 8) INS(it, 1);
    // ---- the hidden layers
    {
      f32x4 unused[8];
      for (int l = 0; l + 1 < nl; ++l) {
        const int S = a.L[l].S, T = a.L[l].T, act = a.L[l].act;
        const int kind = act == ACT_TANH ? 0 : 1;
        const float cm = scale_tab[l] * rs_b * (kind == 0 ? IN_TANH_ARG : 1.f);
        switch (S * 16 + T) {
#define IN2_CASE(S_, T_) case S_ * 16 + T_: in2_layer<S_, T_>(lds + a.L[l].pieces + lane * 16, bias_tab + l * 128 + 4 * g, cm, kind, act, bh, bl, rs_b, unused); break;
          IN2_CASE(1, 1) IN2_CASE(1, 2) IN2_CASE(1, 4) IN2_CASE(1, 8)
          IN2_CASE(2, 1) IN2_CASE(2, 2) IN2_CASE(2, 4) IN2_CASE(2, 8)
          IN2_CASE(4, 1) IN2_CASE(4, 2) IN2_CASE(4, 4) IN2_CASE(4, 8)
#undef IN2_CASE
          default: break;
        }
        if (wave < 8) INS(it, 2 + (l < 4 ? l : 4));
      }
    }
    // ---- the last layer (its float32 rows live here only: carried through the layer loop they cost every hidden layer 32 registers)
    f32x4 res[8];
    {
      const int l = nl - 1;
      const int S = a.L[l].S, T = a.L[l].T, act = a.L[l].act;
      const float cm = scale_tab[l] * rs_b;
      switch (S * 16 + T) {
#define IN2_CASE(S_, T_) case S_ * 16 + T_: in2_layer<S_, T_>(lds + a.L[l].pieces + lane * 16, bias_tab + l * 128 + 4 * g, cm, 2, act, bh, bl, rs_b, res); break;
        IN2_CASE(1, 1) IN2_CASE(1, 2) IN2_CASE(1, 4) IN2_CASE(1, 8)
        IN2_CASE(2, 1) IN2_CASE(2, 2) IN2_CASE(2, 4) IN2_CASE(2, 8)
        IN2_CASE(4, 1) IN2_CASE(4, 2) IN2_CASE(4, 4) IN2_CASE(4, 8)
#undef IN2_CASE
        default:
#pragma unroll
          for (int t = 0; t < 8; ++t) res[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          break;
      }
      if (wave < 8) INS(it, 6);
    }
    in2_store(res, row < n_rows ? out + row * ld_out : nullptr, g, out_cols, vec_out, out_scale);
    if (wave < 8) INS(it, 7);
  }
}

// The STANDARD shape -- two hidden tanh layers of 65 .. 128 units and a last layer: n_neurons [128, 128, L], the default of
// parameters.py:611-639, as encoder (S0 K-steps of raw input, a last layer of TL tiles = the latent) and as decoder (one K-step of
// latent, a last layer of up to eight tiles) -- with every class a template parameter: no layer loop, no class switch.  In the generic
// kernel above the operand registers travel through the switch's joins and the loop's back edge: ~56 v_mov per layer and wave.
template <bool PER_IN, int S0, int TL>
__global__ __launch_bounds__(IN2_NT) void infer16_std_kernel(const InArgs) {
  extern __shared__ __align__(16) char lds[];
  const InK ka = (InK)__builtin_amdgcn_kernarg_segment_ptr();
  const auto& a = *ka;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g0 = lane >> 4;
  float* const bias_tab = reinterpret_cast<float*>(lds + a.bias_tab);
  float* const scale_tab = reinterpret_cast<float*>(lds + a.scale_tab);
  in2_build(ka, lds);
  const long n_rows = a.n_rows, n_blocks = a.n_blocks;
  const long stride = (long)gridDim.x * IN2_NW;
  int in_cols = a.in_cols, vec_in = a.vec_in, out_cols = a.out_cols, vec_out = a.vec_out;
  const long ld_in = a.ld_in, ld_out = a.ld_out;
  const float* const in = a.in;
  float* const out = a.out;
  const float in_scale = a.in_scale, out_scale = a.out_scale;
  const int act_last = a.L[2].act;
  const char* const p0 = lds + a.L[0].pieces + lane * 16;
  const char* const p1 = lds + a.L[1].pieces + lane * 16;
  const char* const p2 = lds + a.L[2].pieces + lane * 16;
  const float rsw0 = scale_tab[0] * IN_TANH_ARG, rsw1 = scale_tab[1] * IN_TANH_ARG * TANH_RS, rsw2 = scale_tab[2] * TANH_RS;
  for (long blk = (long)blockIdx.x * IN2_NW + wave; blk < n_blocks; blk += stride) {
    asm volatile("" : "+s"(in_cols), "+s"(out_cols), "+s"(vec_in), "+s"(vec_out));
    int g = g0;
    asm volatile("" : "+v"(g));
    const long row = blk * IN2_ROWS + n;
    const float* rp = in + min(row, n_rows - 1) * ld_in;
    half8 bh[4], bl[4];
    float rs_b;
    f32x4 res[8];
    in2_input<PER_IN, S0>(rp, g, in_cols, vec_in, in_scale, bh, bl, rs_b);
    in2_layer<S0, 8, 0>(p0, bias_tab + 4 * g, rsw0 * rs_b, 0, ACT_TANH, bh, bl, rs_b, res);
    in2_layer<4, 8, 0>(p1, bias_tab + 128 + 4 * g, rsw1, 0, ACT_TANH, bh, bl, rs_b, res);
    in2_layer<4, TL, 2>(p2, bias_tab + 256 + 4 * g, rsw2, 2, act_last, bh, bl, rs_b, res);
    in2_store(res, row < n_rows ? out + row * ld_out : nullptr, g, out_cols, vec_out, out_scale);
  }
}

// ---- WIDE ENDS: a first layer of more than 128 inputs and / or a last layer of more than 128 out-features (C4: 2 x 512 sines and
// cosines in, C5: 256 columns; the decoders mirrored).  Their pieces do not fit beside the others in LDS, and streamed per 16-row
// block from L2 they are what the training kernels run forward-only are bound by (512 KB per block at C4: 64 GB per pass).  Here
// the WORKGROUP stages them chunk by chunk -- two K-steps of the first layer, four tiles of the last: 32 KB, built from the float32
// parameters by all 1024 threads, double-buffered, one barrier per chunk -- and its sixteen waves run the chunk on their own 16-row
// blocks: a staged chunk serves 256 rows (32 KB of L2 reads per 256 rows and chunk).  Everything between the ends is the resident
// form above.  The rows of a wide input arrive chunk by chunk as well (the next chunk's 16 values per lane and the thread's eight
// weights requested at the head of the current chunk: its operand arithmetic and its products to arrive under); a non-periodic row is scaled by the running maximum of the chunks seen so far -- a power of two that only
// ever falls, so the accumulators are rescaled exactly (x ratio <= 1) when it does.
constexpr int INW_KC = 2;                    // K-steps of a wide first layer per staged chunk (8 tiles x 2 steps x 2 planes = 32 pieces)
constexpr int INW_NC = 4;                    // tiles of a wide last layer per staged chunk (4 tiles x 4 steps x 2 planes = 32 pieces)
constexpr int INW_STAGE = 32 * 1024 + 256;   // a stage buffer: the pieces, then the chunk's 64 biases (wide last layer)

// 2^14 (sin, cos) of an angle given in QUARTER TURNS (u = x 4 / period): the reduction is exact -- q = rint(u), f = u - q in [-1/2, 1/2]
// -- where sincos_bounded (emx_common.h) spends three fused steps on pi / 2 in three pieces, and the fixed scale 2^14 of the operand is
// folded into the polynomials' constants: 33 instructions a pair with the split against 41.  The same polynomials and quadrant logic;
// results differ from sincos_bounded's in the last bit or two (the wide first layer only: the resident form keeps sincos_bounded).
__device__ __forceinline__ void inw_sincos16(const float u, float& s16, float& c16) {
  const float q = __builtin_rintf(u);
  const float f = u - q;
  const float r = f * 1.5707963267948966f, r16 = f * (1.5707963267948966f * 16384.f);
  const int n = (int)q;
  const float r2 = r * r;
  const float sp = fmaf(r16 * r2, fmaf(r2, fmaf(r2, -1.95152959e-4f, 8.33216087e-3f), -1.66666546e-1f), r16);
  const float cp = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.44331571e-5f * 16384.f, -1.38873163e-3f * 16384.f), 4.16666457e-2f * 16384.f), -8192.f), 16384.f);
  const float sa = (n & 1) ? cp : sp, cb = (n & 1) ? sp : cp;
  s16 = (n & 2) ? -sa : sa;
  c16 = ((n + 1) & 2) ? -cb : cb;
}

// the 16 values of a lane's row in chunk c of a wide first layer (two K-steps); periodic: 8 raw angles in x[sl][4 j], x[sl][4 j + 1]
template <bool PER_IN>
__device__ __forceinline__ void inw_request(const float* __restrict__ rp, const int c, const int g, const int in_cols, const int vec_in,
                                            float (&x)[INW_KC][8]) {
  if (vec_in == (PER_IN ? 2 : 4)) {
#pragma unroll
    for (int sl = 0; sl < INW_KC; ++sl) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (PER_IN) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(rp + min(16 * (INW_KC * c + sl) + 8 * j + 2 * g, in_cols - 2));
          x[sl][4 * j] = v[0], x[sl][4 * j + 1] = v[1], x[sl][4 * j + 2] = 0.f, x[sl][4 * j + 3] = 0.f;
        } else {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rp + min(32 * (INW_KC * c + sl) + 16 * j + 4 * g, in_cols - 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) x[sl][4 * j + e] = v[e];
        }
      }
    }
  } else {
#pragma unroll
    for (int sl = 0; sl < INW_KC; ++sl) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (PER_IN) {
          const int col = 16 * (INW_KC * c + sl) + 8 * j + 2 * g;
          x[sl][4 * j] = rp[min(col, in_cols - 1)], x[sl][4 * j + 1] = rp[min(col + 1, in_cols - 1)], x[sl][4 * j + 2] = 0.f, x[sl][4 * j + 3] = 0.f;
        } else {
          const int col = 32 * (INW_KC * c + sl) + 16 * j + 4 * g;
#pragma unroll
          for (int e = 0; e < 4; ++e) x[sl][4 * j + e] = rp[min(col + e, in_cols - 1)];
        }
      }
    }
  }
}

// The shape is the standard one (infer16_std_kernel) with ONE end wide, every class a template parameter:
//   WIDE_IN:  [K0 > 128 -> 65 .. 128, tanh] [-> 65 .. 128, tanh] [-> TL tiles]              (an encoder; SX unused)
//   WIDE_OUT: [SX K-steps -> 65 .. 128, tanh] [-> 65 .. 128, tanh] [-> N > 128]              (a decoder; TL unused)
template <bool PER_IN, bool WIDE_IN, bool WIDE_OUT, int SX, int TL>
__global__ __launch_bounds__(IN2_NT) void infer16_wide_kernel(const InArgs) {
  static_assert(WIDE_IN != WIDE_OUT, "one staged end per launch: two stage pairs and the resident layers do not fit in LDS");
  extern __shared__ __align__(16) char lds[];
  const InK ka = (InK)__builtin_amdgcn_kernarg_segment_ptr();
  const auto& a = *ka;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g0 = lane >> 4;
  constexpr int nl = 3;
  float* const bias_tab = reinterpret_cast<float*>(lds + a.bias_tab);
  float* const scale_tab = reinterpret_cast<float*>(lds + a.scale_tab);
  char* const stage = lds + a.stage;
  in2_build(ka, lds);

  const long n_rows = a.n_rows, n_iters = (a.n_blocks + IN2_NW - 1) / IN2_NW;
  int in_cols = a.in_cols, vec_in = a.vec_in, out_cols = a.out_cols, vec_out = a.vec_out;
  const long ld_in = a.ld_in, ld_out = a.ld_out;
  const float* const in = a.in;
  float* const out = a.out;
  const float in_scale = a.in_scale, out_scale = a.out_scale;
  const float in_quarter = in_scale * 0.63661977236758134f;   // angle -> quarter turns (4 / period)

  // the staged layer and this thread's part of a chunk: piece lane (ln = tid & 63) of piece pair ts = tid >> 6
  const int sl_ = WIDE_IN ? 0 : nl - 1;
  const int SK = a.L[sl_].K, SNn = a.L[sl_].N;
  const float* const SW = a.params + a.L[sl_].w_off;
  const float* const Sbias = a.params + a.L[sl_].b_off;
  const int s_kperm = a.L[sl_].kperm, s_nperm = a.L[sl_].nperm;
  const int n_chunks = WIDE_IN ? (a.L[sl_].S + INW_KC - 1) / INW_KC : (a.L[sl_].T + INW_NC - 1) / INW_NC;
  const int ts = tid >> 6, pm = lane & 15, pg = lane >> 4;
  float s_sc, s_rs;
  {
    // (the layer's scale as in2_build left it: rs in the table; sc = 1 / rs, both powers of two)
    s_rs = scale_tab[sl_];
    s_sc = __uint_as_float((254u << 23) - __float_as_uint(s_rs));
  }
  float wv[8];
  // The thread's eight weights of a chunk: rows k_i of the layer's kernel, ONE column f.  With kk(pg, i) = 4 pg + {0, 1, 2, 3, 16, 17, 18,
  // 19} the rows are a per-thread base plus eight wave-uniform steps (interleaved (sin, cos) inputs: {0, K/2, 1, K/2 + 1, 8, K/2 + 8,
  // 9, K/2 + 9}), so the requests are eight scalar base pointers and one 32-bit offset register -- formed per thread and chunk they
  // were 150 of a chunk's 430 vector instructions, kept per thread eight registers too many.
  const int t0 = WIDE_IN ? (ts >> 1) : (ts >> 2), st0 = WIDE_IN ? (ts & 1) : (ts & 3);
  const int vf0 = 16 * t0 + pm;                         // virtual out-feature (of chunk 0 where the chunks walk the out-features)
  const int vkb = 32 * st0 + 4 * pg;                    // virtual k of element 0 (of chunk 0 where the chunks walk the inputs)
  const int w_off0 = (s_kperm ? (vkb >> 1) : vkb) * SNn + (s_nperm ? in_unperm(vf0, SNn >> 1) : vf0);
  const int w_step = WIDE_IN ? (s_kperm ? 32 : 64) * SNn : (s_nperm ? 32 : 64);   // a chunk further, in floats
  // (rows of element i beyond element 0, branch-free: ci = 2 (ci >> 1) + (ci & 1) for plain inputs, (ci >> 1) + (ci & 1) K / 2 for
  //  interleaved ones -- a test of s_kperm inside the request loop becomes a branch and a full wait around every load)
  const int row_pair = s_kperm ? 1 : 2, row_odd = s_kperm ? (SK >> 1) : 1;
  auto stage_request = [&](const int c) __attribute__((always_inline)) {
    const float* const Wc = SW + (long)c * w_step;
    const int klim = WIDE_IN ? SK - 64 * c : SK;        // virtual k of the chunk below this are inside the layer
    const bool f_ok = (WIDE_IN ? vf0 : vf0 + 64 * c) < SNn;
    int vkb_ = vkb, off_ = w_off0;
    asm volatile("" : "+v"(vkb_), "+v"(off_));          // (loop-variant to the compiler: hoisted, the eight addresses are sixteen registers -- spilled)
    const unsigned off = f_ok ? (unsigned)off_ : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                        // all eight requests first, unconditionally (inside the kernel: offset 0 where invalid)
      const int ci = i < 4 ? i : 12 + i;                 // kk(pg, i) - 4 pg
      const long di = (long)((ci >> 1) * row_pair + (ci & 1) * row_odd) * SNn;   // (wave-uniform)
      const float* const q = (vkb_ + ci < klim) ? Wc + di + off : SW;   // (a select of addresses, then ONE load: no branch)
      wv[i] = *q;
    }
    // (what was requested out of bounds is zeroed in stage_store, behind the chunk's products: a select right here would wait for
    //  the eight requests in front of them)
  };
  auto stage_store = [&](const int c, char* const buf) __attribute__((always_inline)) {
    half8 hi, lo;
    const int klim = WIDE_IN ? SK - 64 * c : SK;
    const bool f_ok = (WIDE_IN ? vf0 : vf0 + 64 * c) < SNn;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int c0 = i < 4 ? i : 12 + i;
      const float w0 = (f_ok & (vkb + c0 < klim)) ? wv[i] * s_sc : 0.f, w1 = (f_ok & (vkb + c0 + 1 < klim)) ? wv[i + 1] * s_sc : 0.f;
      unsigned ph, pl;
      split_f16_pair(w0, w1, ph, pl);
      const half2v vh = __builtin_bit_cast(half2v, ph), vl = __builtin_bit_cast(half2v, pl);
      hi[i] = vh[0], hi[i + 1] = vh[1], lo[i] = vl[0], lo[i + 1] = vl[1];
    }
    char* const p = buf + ts * 2048 + lane * 16;
    *reinterpret_cast<half8*>(p) = hi;
    *reinterpret_cast<half8*>(p + 1024) = lo;
    if constexpr (WIDE_OUT) {
      if (tid < 16 * INW_NC) {
        const int vf = 16 * INW_NC * c + tid;
        const int f = s_nperm ? in_unperm(vf, SNn >> 1) : vf;
        reinterpret_cast<float*>(buf + 32 * 1024)[tid] = vf < SNn ? Sbias[f] : 0.f;
      }
    }
  };

  for (long wi = blockIdx.x; wi < n_iters; wi += gridDim.x) {
    asm volatile("" : "+s"(in_cols), "+s"(out_cols), "+s"(vec_in), "+s"(vec_out));
    int g = g0;
    asm volatile("" : "+v"(g));
    const long row = (wi * IN2_NW + wave) * IN2_ROWS + n;
    const float* rp = in + min(row, n_rows - 1) * ld_in;
    float* const orow = row < n_rows ? out + row * ld_out : nullptr;
    half8 bh[4], bl[4];
    float rs_b;
    if constexpr (WIDE_IN) {
      // ---- the wide first layer, chunk by chunk into the eight tiles' accumulators
      float xn[INW_KC][8];
      inw_request<PER_IN>(rp, 0, g, in_cols, vec_in, xn);
      stage_request(0);
      stage_store(0, stage);
      __syncthreads();
      f32x4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned m_run = 0;
      float rs_prev = 0.f;
      rs_b = TANH_RS;
      for (int c = 0; c < n_chunks; ++c) {
        asm volatile("" : "+s"(in_cols), "+s"(vec_in));
        asm volatile("" : "+v"(g));
        // the next chunk's rows and weights first: they have this chunk's operand arithmetic AND its products to arrive under (requested
        // behind the operand -- one set of row registers -- they had the products alone, and every wave of the workgroup met the
        // barrier waiting for them)
        float x[INW_KC][8];
#pragma unroll
        for (int sl = 0; sl < INW_KC; ++sl)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[sl][i] = xn[sl][i];
        if (c + 1 < n_chunks) {
          inw_request<PER_IN>(rp, c + 1, g, in_cols, vec_in, xn);
          stage_request(c + 1);
        }
        // the chunk's operand
        half8 ch[INW_KC], cl[INW_KC];
        if constexpr (PER_IN) {
#pragma unroll
          for (int sl = 0; sl < INW_KC; ++sl) {
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
              // (no column masks: columns beyond the row re-read its last ones -- finite -- and meet zero pieces)
              float sn, cs;
              inw_sincos16(x[sl][4 * (pq >> 1) + (pq & 1)] * in_quarter, sn, cs);
              in2_split2(sn, cs, ch[sl], cl[sl], 4 * (pq >> 1) + 2 * (pq & 1));
            }
          }
        } else {
          // (no column masks: columns beyond the row re-read its last ones -- no news for the maximum -- and meet zero pieces)
          float mf = 0.f;
#pragma unroll
          for (int sl = 0; sl < INW_KC; ++sl)
#pragma unroll
            for (int i = 0; i < 8; i += 2) mf = __builtin_fmaxf(mf, __builtin_fmaxf(__builtin_fabsf(x[sl][i]), __builtin_fabsf(x[sl][i + 1])));
          m_run = max(m_run, max_xor32(max_xor16(abs_bits(mf))));   // the row's four lanes; the running maximum over the chunks
          float sc;
          pow2_scale_bits<ACT_TOP>(m_run, sc, rs_b);
          const float ratio = c == 0 ? 1.f : sc * rs_prev;          // <= 1, a power of two
          rs_prev = rs_b;
          if (__builtin_amdgcn_ballot_w64(ratio != 1.f) != 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[t][e] *= ratio;
          }
#pragma unroll
          for (int sl = 0; sl < INW_KC; ++sl)
#pragma unroll
            for (int i = 0; i < 8; i += 2) in2_split2(x[sl][i] * sc, x[sl][i + 1] * sc, ch[sl], cl[sl], i);
        }
        // the chunk's products
        const char* const pb = stage + (c & 1) * INW_STAGE + lane * 16;
#pragma unroll
        for (int sl = 0; sl < INW_KC; ++sl) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const char* p = pb + (t * INW_KC + sl) * 2048;
            const half8 ah = *reinterpret_cast<const half8*>(p), al = *reinterpret_cast<const half8*>(p + 1024);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, cl[sl], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ch[sl], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ch[sl], acc[t], 0, 0, 0);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (c + 1 < n_chunks) stage_store(c + 1, stage + ((c + 1) & 1) * INW_STAGE);
        __syncthreads();
      }
      f32x4 unused[8];
      in2_epilogue<8, 0>(acc, bias_tab + 4 * g, scale_tab[0] * rs_b * IN_TANH_ARG, 0, ACT_TANH, bh, bl, rs_b, unused);
    } else {
      f32x4 unused[8];
      in2_input<PER_IN, SX>(rp, g, in_cols, vec_in, in_scale, bh, bl, rs_b);
      in2_layer<SX, 8, 0>(lds + a.L[0].pieces + lane * 16, bias_tab + 4 * g, scale_tab[0] * IN_TANH_ARG * rs_b, 0, ACT_TANH, bh, bl, rs_b, unused);
    }
    {
      f32x4 unused[8];
      in2_layer<4, 8, 0>(lds + a.L[1].pieces + lane * 16, bias_tab + 128 + 4 * g, scale_tab[1] * IN_TANH_ARG * TANH_RS, 0, ACT_TANH, bh, bl, rs_b, unused);
    }
    if constexpr (!WIDE_OUT) {
      f32x4 res[8];
      in2_layer<4, TL, 2>(lds + a.L[2].pieces + lane * 16, bias_tab + 256 + 4 * g, scale_tab[2] * TANH_RS, 2, a.L[2].act, bh, bl, rs_b, res);
      in2_store(res, orow, g, out_cols, vec_out, out_scale);
    } else {
      // ---- the wide last layer, four tiles at a time: products, bias, activation, (atan2,) stores
      const int l = nl - 1;
      const int act = a.L[l].act;
      const float cm = scale_tab[l] * rs_b;          // (the operand: the four K-steps the tanh layer in front of it wrote)
      stage_request(0);
      stage_store(0, stage);
      __syncthreads();
      for (int c = 0; c < n_chunks; ++c) {
        asm volatile("" : "+s"(out_cols), "+s"(vec_out));
        asm volatile("" : "+v"(g));
        if (c + 1 < n_chunks) stage_request(c + 1);
        const char* const pb = stage + (c & 1) * INW_STAGE;
        f32x4 acc[INW_NC];
#pragma unroll
        for (int t = 0; t < INW_NC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int t = 0; t < INW_NC; ++t) {
            const char* p = pb + (t * 4 + s) * 2048 + lane * 16;
            const half8 ah = *reinterpret_cast<const half8*>(p), al = *reinterpret_cast<const half8*>(p + 1024);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc[t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const float* const bt = reinterpret_cast<const float*>(pb + 32 * 1024) + 4 * g;
#pragma unroll
        for (int t = 0; t < INW_NC; ++t) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + 16 * t);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(acc[t][e], cm, b4[e]);
        }
        if (act != ACT_NONE) {
#define INW_ACT(K_, EXPR_)                                                                       \
  case K_:                                                                                       \
    _Pragma("unroll") for (int t = 0; t < INW_NC; ++t)                                           \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float xv = acc[t][e]; acc[t][e] = EXPR_; } \
    break;
          switch (act) {
            INW_ACT(ACT_TANH, tanh_f32(xv))
            INW_ACT(ACT_RELU, act_apply(ACT_RELU, xv))
            INW_ACT(ACT_SIGMOID, act_apply(ACT_SIGMOID, xv))
            INW_ACT(ACT_ELU, act_apply(ACT_ELU, xv))
            INW_ACT(ACT_SOFTPLUS, act_apply(ACT_SOFTPLUS, xv))
            INW_ACT(ACT_SELU, act_apply(ACT_SELU, xv))
            INW_ACT(ACT_SOFTSIGN, act_apply(ACT_SOFTSIGN, xv))
            INW_ACT(ACT_EXP, act_apply(ACT_EXP, xv))
            default: break;
          }
#undef INW_ACT
        }
        if (orow != nullptr) {
          if (out_scale != 0.f) {
            // PeriodicOutput: virtual out-features (2 k, 2 k + 1) = (sin, cos) of angle k = 32 c + 8 t + 2 g (+ 1)
#pragma unroll
            for (int t = 0; t < INW_NC; ++t) {
              const int k = 8 * (INW_NC * c + t) + 2 * g;
              if (8 * (INW_NC * c + t) < out_cols) {     // (wave-uniform)
                const float a0 = in_atan2(acc[t][0], acc[t][1]) * out_scale;
                const float a1 = in_atan2(acc[t][2], acc[t][3]) * out_scale;
                if (vec_out >= 2) {
                  if (k < out_cols) *reinterpret_cast<f32x2*>(orow + k) = f32x2{a0, a1};
                } else {
                  if (k < out_cols) orow[k] = a0;
                  if (k + 1 < out_cols) orow[k + 1] = a1;
                }
              }
            }
          } else if (vec_out == 4) {
#pragma unroll
            for (int t = 0; t < INW_NC; ++t) {
              const int col = 16 * (INW_NC * c + t) + 4 * g;
              if (col < out_cols) *reinterpret_cast<f32x4*>(orow + col) = acc[t];
            }
          } else if (vec_out == 2) {
#pragma unroll
            for (int t = 0; t < INW_NC; ++t) {
              const int col = 16 * (INW_NC * c + t) + 4 * g;
              if (col < out_cols) *reinterpret_cast<f32x2*>(orow + col) = f32x2{acc[t][0], acc[t][1]};
              if (col + 2 < out_cols) *reinterpret_cast<f32x2*>(orow + col + 2) = f32x2{acc[t][2], acc[t][3]};
            }
          } else {
#pragma unroll
            for (int t = 0; t < INW_NC; ++t)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int col = 16 * (INW_NC * c + t) + 4 * g + e;
                if (col < out_cols) orow[col] = acc[t][e];
              }
          }
        }
        if (c + 1 < n_chunks) stage_store(c + 1, stage + ((c + 1) & 1) * INW_STAGE);
        __syncthreads();
      }
    }
  }
}

int cu_count() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev] = v;
  }
  return cached[dev];
}

int vec_width(const void* base, long ld, int cols, int want) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  if (want >= 4 && (p & 15) == 0 && (ld & 3) == 0 && (cols & 3) == 0 && cols >= 4) return 4;
  if (want >= 2 && (p & 7) == 0 && (ld & 1) == 0 && (cols & 1) == 0 && cols >= 2) return 2;
  return 1;
}

}  // namespace

extern "C" {

int emx_mlp_infer(const emx_mlp_desc* desc, const float* params, int layer_begin, int layer_end, const float* in, int64_t ld_in,
                  float periodicity_in, int64_t n_rows, float* out, int64_t ld_out, float periodicity_out, emx_stream_t stream) {
  const char* who = "emx_mlp_infer";
  EMX_REQUIRE(desc && params, "%s: null argument", who);
  EMX_REQUIRE(n_rows == 0 || (in && out), "%s: null rows", who);
  EMX_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= EMX_MAX_LAYERS, "%s: n_layers=%d", who, desc->n_layers);
  EMX_REQUIRE(layer_begin >= 0 && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: layer range [%d, %d)", who, layer_begin, layer_end);
  EMX_REQUIRE(n_rows >= 0, "%s: n_rows=%lld", who, (long long)n_rows);
  const bool per_in = periodicity_in > 0.f && std::isfinite(periodicity_in);
  const bool per_out = periodicity_out > 0.f && std::isfinite(periodicity_out);
  const int K0 = desc->width[layer_begin], NL = desc->width[layer_end];
  EMX_REQUIRE(!per_in || (K0 & 1) == 0, "%s: a periodic input needs an even first width (sin | cos), got %d", who, K0);
  EMX_REQUIRE(!per_out || (NL & 1) == 0, "%s: a periodic output needs an even last width (sin | cos), got %d", who, NL);
  const int in_cols = per_in ? K0 / 2 : K0, out_cols = per_out ? NL / 2 : NL;
  EMX_REQUIRE(ld_in >= in_cols && ld_out >= out_cols, "%s: row strides %lld / %lld below the row lengths %d / %d", who, (long long)ld_in,
              (long long)ld_out, in_cols, out_cols);
  if (n_rows == 0) return EMX_OK;
  InArgs a{};
  const int nl = layer_end - layer_begin;
  int lds = 0;
  for (int i = 0; i < nl; ++i) {
    const int l = layer_begin + i;
    const int K = desc->width[l], N = desc->width[l + 1];
    EMX_REQUIRE(K >= 1 && N >= 1, "%s: layer %d is %d x %d", who, l, K, N);
    const bool w_in = i == 0 && K > 128 && N <= 128, w_out = i == nl - 1 && N > 128 && K <= 128;
    if ((K > 128 || N > 128) && !(w_in || w_out))
      return fail(EMX_EUNSUPPORTED, "%s: layer %d is %d x %d: the streaming pass takes layers of up to 128 x 128 between a wide first and a wide last layer", who, l, K, N);
    if (w_in) a.wide_in = 1;
    if (w_out) a.wide_out = 1;
    EMX_REQUIRE(desc->act[l] >= 0 && desc->act[l] <= ACT_LAST, "%s: layer %d: activation %d", who, l, desc->act[l]);
    InLayer& L = a.L[i];
    L.w_off = desc->w_off[l];
    L.b_off = desc->b_off[l];
    L.K = K;
    L.N = N;
    L.S = w_in ? (K + 31) / 32 : in2_class_S(K);     // (a staged layer: the true counts; INW_KC K-steps / INW_NC tiles per chunk)
    L.T = w_out ? (N + 15) / 16 : (w_in ? 8 : in2_class_T(N));
    L.act = desc->act[l];
    L.kperm = (i == 0 && per_in) ? 1 : 0;
    L.nperm = (i == nl - 1 && per_out) ? 1 : 0;
    L.pieces = lds;
    if (!(w_in || w_out)) lds += L.S * L.T * 2048;
  }
  if (a.wide_in && a.wide_out)
    return fail(EMX_EUNSUPPORTED, "%s: layers [%d, %d) have a wide first AND a wide last layer: one staged end per launch", who, layer_begin, layer_end);
  a.bias_tab = lds;
  lds += nl * 128 * 4;
  a.scale_tab = lds;
  lds += (IN_MAXL + 16) * 4;   // (16 >= waves of a workgroup)
  if (a.wide_in || a.wide_out) {
    a.stage = lds;             // (a multiple of 16: pieces are 2 KB, the tables multiples of 64 bytes)
    lds += 2 * INW_STAGE;
  }
  if (lds > IN_LDS_MAX) return fail(EMX_EUNSUPPORTED, "%s: the operand pieces of layers [%d, %d) take %d bytes of LDS (%d at most)", who, layer_begin,
                                    layer_end, lds, IN_LDS_MAX);
  a.params = params;
  a.in = in;
  a.out = out;
  a.ld_in = ld_in;
  a.ld_out = ld_out;
  a.n_rows = n_rows;
  a.n_blocks = (n_rows + IN2_ROWS - 1) / IN2_ROWS;
  a.n_layers = nl;
  a.in_cols = in_cols;
  a.out_cols = out_cols;
  a.in_scale = per_in ? IN_TWO_PI / periodicity_in : 0.f;
  a.out_scale = per_out ? periodicity_out / IN_TWO_PI : 0.f;
  a.vec_in = vec_width(in, ld_in, in_cols, per_in ? 2 : 4);
  a.vec_out = vec_width(out, ld_out, out_cols, per_out ? 2 : 4);
  const long wb = (a.n_blocks + IN2_NW - 1) / IN2_NW;
  const int grid = (int)std::min<long>(cu_count(), wb);
  typedef void (*Fn)(const InArgs);
  Fn fn = per_in ? &infer16_kernel<true> : &infer16_kernel<false>;
  if (a.wide_in || a.wide_out) {
    // one end wide: the standard shape only (two tanh layers of 65 .. 128 units in front of the last layer)
    const bool std_shape = nl == 3 && a.L[0].act == ACT_TANH && a.L[1].act == ACT_TANH && a.L[0].N > 64 && a.L[0].N <= 128 && a.L[1].N > 64 &&
                           a.L[1].N <= 128;
    if (!std_shape)
      return fail(EMX_EUNSUPPORTED, "%s: layers [%d, %d) with a wide end: the streaming pass takes [wide, tanh 65..128, tanh 65..128, last] shapes only", who,
                  layer_begin, layer_end);
#define INW_IN(P_, T_) (P_ ? (Fn)&infer16_wide_kernel<true, true, false, 4, T_> : (Fn)&infer16_wide_kernel<false, true, false, 4, T_>)
#define INW_OUT(P_, S_) (P_ ? (Fn)&infer16_wide_kernel<true, false, true, S_, 8> : (Fn)&infer16_wide_kernel<false, false, true, S_, 8>)
    if (a.wide_in) fn = a.L[2].T == 1 ? INW_IN(per_in, 1) : a.L[2].T == 2 ? INW_IN(per_in, 2) : a.L[2].T == 4 ? INW_IN(per_in, 4) : INW_IN(per_in, 8);
    else fn = a.L[0].S == 1 ? INW_OUT(per_in, 1) : a.L[0].S == 2 ? INW_OUT(per_in, 2) : INW_OUT(per_in, 4);
#undef INW_IN
#undef INW_OUT
  }
  // the standard shape (two hidden tanh layers of 65 .. 128 units, then the last layer) has a kernel of its own per (S0, TL)
  if (!a.wide_in && !a.wide_out && nl == 3 && a.L[0].T == 8 && a.L[1].S == 4 && a.L[1].T == 8 && a.L[2].S == 4 && a.L[0].act == ACT_TANH && a.L[1].act == ACT_TANH) {
#define IN2_STD(P_, S_, T_) (P_ ? (Fn)&infer16_std_kernel<true, S_, T_> : (Fn)&infer16_std_kernel<false, S_, T_>)
#define IN2_STD_T(P_, S_) (a.L[2].T == 1 ? IN2_STD(P_, S_, 1) : a.L[2].T == 2 ? IN2_STD(P_, S_, 2) : a.L[2].T == 4 ? IN2_STD(P_, S_, 4) : IN2_STD(P_, S_, 8))
    fn = a.L[0].S == 1 ? IN2_STD_T(per_in, 1) : a.L[0].S == 2 ? IN2_STD_T(per_in, 2) : IN2_STD_T(per_in, 4);
#undef IN2_STD_T
#undef IN2_STD
  }
  {
    static std::mutex mu;                                   // (40 kernels, each raised to the full LDS once)
    static std::set<const void*> attr_done;
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_done.count(reinterpret_cast<const void*>(fn))) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, IN_LDS_MAX) != hipSuccess)
        return fail(EMX_ELAUNCH, "%s: hipFuncSetAttribute(%d bytes of LDS)", who, IN_LDS_MAX);
      attr_done.insert(reinterpret_cast<const void*>(fn));
    }
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(IN2_NT), (size_t)lds, (hipStream_t)stream, a);
  return check_launch(who);
}

#ifdef IN_STAMP
int emx_debug_infer_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_in_stamps), sizeof(unsigned long long) * 8 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
