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
// block and write every layer's rows.
//
// The kernel.  One 512-thread workgroup per compute unit.  (1) The workgroup builds the network's operand pieces IN ITS LDS, from
// the float32 parameters themselves (no operand store, no workspace): per layer one power-of-two scale (the layer's largest |w| to
// [2^12, 2^13), as csrc/wop.h), two f16 planes, in 1-KB pieces of v_mfma_f32_32x32x16_f16 -- piece (tile t of 32 out-features,
// K-step s of 16 inputs, plane): lane (lr, h) holds W[k(s, h, i)][32 t + lr], i < 8 -- 144 KB for 128 -> 128 -> 128 -> 2.
// (2) Every WAVE then walks 32-row blocks on its own, transposed (H^T = W^T X^T, the picture of csrc/tchain.h with the wider
// instruction): the weights are the A operand, the 32 rows the N dimension, and the accumulators of tile t in lane (row lr, h) are
// out-features 32 t + 8 q + 4 h + e -- with k(s, h, i) = 16 s + 8 (i >> 2) + 4 h + (i & 3) exactly the eight k that lane supplies
// for K-steps 2 t and 2 t + 1 of the NEXT layer: bias, activation and the f16 split happen in place, in registers; a block's rows
// never touch LDS or HBM between the input load and the result store, and no wave ever waits for another (no barrier after (1)).
// The next block's input rows are requested before the current block's products (64 registers in flight per lane).
// A periodic input (PeriodicInput, layers.py:204-215) is formed on chip -- the k order of layer 0 is then the interleaved one
// (sin x_0, cos x_0, sin x_1, ...), so that a lane's eight k are four raw values; a periodic output (PeriodicOutput,
// layers.py:1118-1125) likewise: the out-features of the last layer are interleaved, atan2 runs on register pairs.
//
// Arithmetic: the dense layers' precision class (split-f16 x 3 products, float32 accumulation; one operand scale per row), another
// order of summation than the training kernels: results agree with them to float32 rounding, not bit for bit.
// Scope: every layer of the range at most 128 wide (2 x 64 raw features for a periodic input / output) and the pieces of the range
// within LDS; anything else answers EMX_EUNSUPPORTED and the caller runs emx_mlp_forward.
#include <cstdlib>
#include <type_traits>

#include "act.h"
#include "emx_common.h"
#include "wop.h"

namespace {

using namespace emx;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IN_ROWS = 32;
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
};
// (the record is indexed at run time -- layer tables -- so it is read through the kernel-argument segment itself: indexing the
//  by-value struct makes hipcc copy all of it to scratch)
typedef const InArgs __attribute__((address_space(4))) * InK;

__device__ __forceinline__ int in_vk(int s, int h, int i) { return 16 * s + 8 * (i >> 2) + 4 * h + (i & 3); }
// interleaved order of a periodic tensor of 2 n columns [sin | cos]: virtual index v -> column
__device__ __forceinline__ int in_unperm(int v, int n) { return (v & 1) ? n + (v >> 1) : (v >> 1); }

// One layer on the wave's 32 rows.  S K-steps, T tiles (compile time: straight-line code, LDS offsets as immediates).
//
// The instruction stream is laid out BY HAND and fenced (__builtin_amdgcn_sched_barrier): one wave per SIMD runs it, so whatever is
// to overlap must stand interleaved in program order.  Left to the scheduler the stream was [two LDS reads, wait, three products]
// per K-step with every tile's bias / tanh / split as one block behind its products -- 38 000 cycles per 32-row block where the
// products alone are 6 900.  Here: the operand pieces travel two K-steps ahead through a register ring of three (the (tile, K-step)
// sequence is one contiguous run of 2-KB packages in LDS), and between the three products of a K-step stand the epilogue
// instructions of the PREVIOUS tile's element pairs -- a pair's fifteen instructions in three groups of about five, the issue
// slots a 32-cycle matrix instruction leaves (MI355X_MICROARCH.md: an MFMA holds the vector issue for 8 of its 32 cycles).
//
// ACT 1, a hidden tanh layer (the fast path): a tile's result becomes K-steps 2 t, 2 t + 1 of the next layer's operand at once, bias,
// tanh and the f16 split in registers -- tiles 0 .. T - 2 into a second register set (nh, nl), which moves into (bh, bl) beside the
// LAST tile's products (K-step s of the old operand is dead once that tile's products of step s are issued), the last tile's
// directly.  The bias table of such a layer holds b 2 log2(e), cm carries the same factor: tanh(x) 2^14 = 2^14 - 2^15 / (2^(x') + 1).
// ACT 0 / 2, linear / any other Keras activation (kind: wave-uniform): the tiles' float32 results are left in `keep` -- the caller
// stores them (last layer) or makes the next operand from them (the row's scale needs every tile first).
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
#define IN_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int S, int T, int ACT>
__device__ __forceinline__ void in_layer(const char* __restrict__ pieces /* + lane * 16 */, const float* __restrict__ bt /* bias + 4 h */,
                                         const float cm, half8 (&bh)[8], half8 (&bl)[8], f32x16 (&keep)[4]) {
  constexpr int J = S * T;          // packages of the layer, in execution order
  constexpr int PPS = 8 / S;        // element pairs of a finished tile that START their epilogue per K-step of the next tile
  half8 nh[ACT == 1 ? 8 : 1], nl[ACT == 1 ? 8 : 1];
  half8 ah[3], al[3];
  auto fetch = [&](const int j) __attribute__((always_inline)) {
    const char* p = pieces + j * 2048;
    ah[j % 3] = *reinterpret_cast<const half8*>(p);
    al[j % 3] = *reinterpret_cast<const half8*>(p + 1024);
  };
  fetch(0);
  if (J > 1) fetch(1);
  f32x16 acc[2];
  // a tile's biases, by tile parity: requested in front of the tile's LAST K-step -- a request right in front of its use would make
  // the epilogue's first instruction wait for every LDS read in flight (s_waitcnt lgkmcnt(0): the operand pieces two steps ahead too)
  f32x4 bias[2][4];
  // The epilogue of an element pair (elements 2 p, 2 p + 1 of a tile's sixteen) is NINE dependent steps; one wave alone on its SIMD
  // pays the full latency of every dependent vector instruction (a pair's fifteen instructions back to back took ~150 cycles, not
  // 60).  So the pairs are software-pipelined over three K-steps: a pair that starts at step j runs stage 0 (read, scale + bias,
  // exp2) in the three product slots of step j, stage 1 (+ 1, rcp, 2^14 - 2^15 r) in those of step j + 1, stage 2 (f16 high plane,
  // remainder, low plane) in those of step j + 2 -- every slot holds one instruction of each of three pairs, all independent, each
  // fed by what the slot before it left.  pv: the pairs' values between slots, by start step modulo 3.
  float pv[3][PPS][2];
  auto part = [&](const int tp, const int p, const int stage, const int m, float (&v)[2]) __attribute__((always_inline)) {
    f32x16& A = acc[tp & 1];
    const int r = 2 * p;
    if constexpr (ACT == 1) {
      const int s2 = 2 * tp + (p >> 2), i2 = 4 * ((p >> 1) & 1) + 2 * (p & 1);   // out-feature 32 t + 8 q + 4 h + e is k (K-step 2 t + (q >> 1), element 4 (q & 1) + e) of the next layer
      half8& oh = tp == T - 1 ? bh[s2] : nh[s2];
      half8& ol = tp == T - 1 ? bl[s2] : nl[s2];
      typedef float f2 __attribute__((ext_vector_type(2)));
      if (stage == 0) {
        if (m == 0) v[0] = A[r], v[1] = A[r + 1];
        else if (m == 1) v[0] = fmaf(v[0], cm, bias[tp & 1][r >> 2][r & 3]), v[1] = fmaf(v[1], cm, bias[tp & 1][r >> 2][(r & 3) + 1]);
        else v[0] = __builtin_amdgcn_exp2f(v[0]), v[1] = __builtin_amdgcn_exp2f(v[1]);
      } else if (stage == 1) {
        if (m == 0) v[0] += 1.f, v[1] += 1.f;
        else if (m == 1) v[0] = __builtin_amdgcn_rcpf(v[0]), v[1] = __builtin_amdgcn_rcpf(v[1]);
        else v[0] = fmaf(-32768.f, v[0], 16384.f), v[1] = fmaf(-32768.f, v[1], 16384.f);
      } else {
        if (m == 0) {
          const half2v vh = __builtin_convertvector(f2{v[0], v[1]}, half2v);
          oh[i2] = vh[0], oh[i2 + 1] = vh[1];
        } else if (m == 1) {
          v[0] -= (float)oh[i2], v[1] -= (float)oh[i2 + 1];
        } else {
          const half2v vl = __builtin_convertvector(f2{v[0], v[1]}, half2v);
          ol[i2] = vl[0], ol[i2 + 1] = vl[1];
        }
      }
    } else {
      if (stage == 0 && m == 1) {
        keep[tp][r] = fmaf(A[r], cm, bias[tp & 1][r >> 2][r & 3]);
        keep[tp][r + 1] = fmaf(A[r + 1], cm, bias[tp & 1][r >> 2][(r & 3) + 1]);
      }
    }
  };
  // slot m of (possibly virtual) step j: one part of every pair in flight.  Pairs that START at step j: those of tile j / S - 1,
  // (j % S) PPS .. + PPS - 1 -- for j < J + S (the steps S beyond the loop start the last tile's pairs).
  auto slot = [&](const int j, const int m) __attribute__((always_inline)) {
#pragma unroll
    for (int stage = 0; stage < (ACT == 1 ? 3 : 1); ++stage) {
      const int js = j - stage;                       // the step these pairs started at
      if (js >= S && js < J + S) {
#pragma unroll
        for (int pp = 0; pp < PPS; ++pp) part(js / S - 1, (js % S) * PPS + pp, stage, m, pv[js % 3][pp]);
      }
    }
  };
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t & 1][r] = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int j = t * S + s;
      if (s == S - 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[t & 1][q] = *reinterpret_cast<const f32x4*>(bt + 32 * t + 8 * q);
      }
      if (j + 2 < J) fetch(j + 2);
      IN_FENCE();
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const half8& A = m == 1 ? al[j % 3] : ah[j % 3];
        const half8& B = m == 0 ? bl[s] : bh[s];            // small products first: hi lo, lo hi, hi hi
        acc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[t & 1], 0, 0, 0);
        IN_FENCE();
        slot(j, m);
        IN_FENCE();
      }
      // (eight K-steps: the operand steps of tiles 0 .. T - 3 are complete two steps into the last tile, and step s of the old operand
      //  is dead behind the last tile's products of step s -- they move beside those products; everything else behind the drain)
      if (ACT == 1 && S == 8 && t == T - 1 && s >= 2 && s < 2 * (T - 2)) bh[s] = nh[s], bl[s] = nl[s];
    }
  }
  // the drain: the last tile's pairs start here (virtual steps J .. J + S - 1), two more steps finish the pipeline
#pragma unroll
  for (int j = J; j < J + S + 2; ++j)
#pragma unroll
    for (int m = 0; m < 3; ++m) slot(j, m);
  if constexpr (ACT == 1) {
#pragma unroll
    for (int s = 0; s < 2 * (T - 1); ++s)
      if (!(S == 8 && s >= 2 && s < 2 * (T - 2))) bh[s] = nh[s], bl[s] = nl[s];
  }
}

// any Keras activation but tanh-in-a-hidden-layer on the kept rows (kind: wave-uniform; the switch outside the element loops)
template <int T>
__device__ __forceinline__ void in_activate(const int kind, f32x16 (&keep)[4]) {
#define IN_ACT_CASE(K_)                                                     \
  case K_:                                                                  \
    _Pragma("unroll") for (int t = 0; t < T; ++t)                           \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) keep[t][r] = act_apply(K_, keep[t][r]); \
    break;
  switch (kind) {
    case ACT_TANH:                                    // (the last layer's tanh: act_apply leaves tanh to the kernels' fast paths)
      _Pragma("unroll") for (int t = 0; t < T; ++t)
        _Pragma("unroll") for (int r = 0; r < 16; ++r) keep[t][r] = tanh_f32(keep[t][r]);
      break;
    IN_ACT_CASE(ACT_RELU)
    IN_ACT_CASE(ACT_SIGMOID)
    IN_ACT_CASE(ACT_ELU)
    IN_ACT_CASE(ACT_SOFTPLUS)
    IN_ACT_CASE(ACT_SELU)
    IN_ACT_CASE(ACT_SOFTSIGN)
    IN_ACT_CASE(ACT_EXP)
    default: break;
  }
#undef IN_ACT_CASE
}

// (S, T) classes: K-steps padded to 1, 2, 4, 8; tiles to 1, 4
__host__ __device__ inline int in_class_S(int K) { const int s = (K + 15) / 16; return s <= 1 ? 1 : (s <= 2 ? 2 : (s <= 4 ? 4 : 8)); }
__host__ __device__ inline int in_class_T(int N) { return N <= 32 ? 1 : 4; }

// a layer of class (S, T): TANH: a hidden tanh layer, finished in place (keep unused); else the float32 rows are left in keep
template <bool TANH>
__device__ __forceinline__ void in_layer_st(const int S, const int T, const char* pieces, const float* bt, const float cm, half8 (&bh)[8],
                                            half8 (&bl)[8], f32x16 (&keep)[4]) {
  constexpr int A = TANH ? 1 : 0;
  if (T == 4) {
    if (S == 8) in_layer<8, 4, A>(pieces, bt, cm, bh, bl, keep);
    else if (S == 4) in_layer<4, 4, A>(pieces, bt, cm, bh, bl, keep);
    else if (S == 2) in_layer<2, 4, A>(pieces, bt, cm, bh, bl, keep);
    else in_layer<1, 4, A>(pieces, bt, cm, bh, bl, keep);
  } else {
    if (S == 8) in_layer<8, 1, A>(pieces, bt, cm, bh, bl, keep);
    else if (S == 4) in_layer<4, 1, A>(pieces, bt, cm, bh, bl, keep);
    else if (S == 2) in_layer<2, 1, A>(pieces, bt, cm, bh, bl, keep);
    else in_layer<1, 1, A>(pieces, bt, cm, bh, bl, keep);
  }
}

template <bool PER_IN, int NT, bool PREFETCH, bool GEN>   // GEN: hidden layers of other kinds than tanh exist
__global__ __launch_bounds__(NT) void infer_kernel(const InArgs) {
  extern __shared__ __align__(16) char lds[];
  constexpr int NW = NT / 64;
  const InK ka = (InK)__builtin_amdgcn_kernarg_segment_ptr();
  const auto& a = *ka;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const int nl = a.n_layers;
  float* const bias_tab = reinterpret_cast<float*>(lds + a.bias_tab);
  float* const scale_tab = reinterpret_cast<float*>(lds + a.scale_tab);
  unsigned* const red = reinterpret_cast<unsigned*>(scale_tab + IN_MAXL);

  INS(15, 0);
  // ---- (1) the operand pieces of every layer, from the float32 parameters
  for (int l = 0; l < nl; ++l) {
    const int K = a.L[l].K, N = a.L[l].N, S = a.L[l].S, T = a.L[l].T;
    const int kperm = a.L[l].kperm, nperm = a.L[l].nperm;
    const float* __restrict__ W = a.params + a.L[l].w_off;
    const float* __restrict__ bias = a.params + a.L[l].b_off;
    unsigned mx = 0;
    for (int e = tid; e < K * N; e += NT) mx = max(mx, abs_bits(W[e]));
    mx = wave_max_u32(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = max(mx, red[w]);
    float sc, rs;
    pow2_scale_bits<WOP_TOP>(mx, sc, rs);
    if (tid == 0) scale_tab[l] = rs;
    char* const pieces = lds + a.L[l].pieces;
    for (int task = tid; task < T * S * 64; task += NT) {
      const int ln = task & 63, ts = task >> 6;
      const int t = ts / S, s = ts - t * S;
      const int plr = ln & 31, ph = ln >> 5;
      const int vf = 32 * t + plr;
      const int f = nperm ? in_unperm(vf, N >> 1) : vf;
      half8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int vk = in_vk(s, ph, i);
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
    // (a hidden tanh layer's epilogue works on x 2 log2 e: in_layer)
    const float bscale = (l + 1 < nl && a.L[l].act == ACT_TANH) ? IN_TANH_ARG : 1.f;
    for (int vf = tid; vf < 128; vf += NT) {
      const int f = nperm ? in_unperm(vf, N >> 1) : vf;
      bias_tab[l * 128 + vf] = vf < N ? bias[f] * bscale : 0.f;
    }
    __syncthreads();   // (the scratch words are reused by the next layer; the last one publishes the image)
  }

  INS(15, 1);
  // ---- (2) every wave for itself: 32-row blocks wave, wave + (waves of the launch), ...
  const long n_rows = a.n_rows, n_blocks = a.n_blocks;
  const long stride = (long)gridDim.x * NW;
  int in_cols = a.in_cols, vec_in = a.vec_in;
  const long ld_in = a.ld_in, ld_out = a.ld_out;
  const float* const in = a.in;
  float* const out = a.out;
  const float in_scale = a.in_scale;
  const int S0 = a.L[0].S;
  int out_cols = a.out_cols, vec_out = a.vec_out;
  const float out_scale = a.out_scale;

  // the block's input as it arrives: PER_IN: four raw values per K-step (two per half); else eight
  constexpr int XE = PER_IN ? 4 : 8;
  float x[8][XE];
  auto load_steps = [&](const float* rp, auto S_) __attribute__((always_inline)) {
    constexpr int SS = decltype(S_)::value;
#pragma unroll
    for (int s = 0; s < SS; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (PER_IN) {
          const int c = 8 * s + 4 * j + 2 * h;                    // raw columns c, c + 1
          if (vec_in >= 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(rp + min(c, in_cols - 2));
            x[s][2 * j] = v[0], x[s][2 * j + 1] = v[1];
          } else {
            x[s][2 * j] = rp[min(c, in_cols - 1)], x[s][2 * j + 1] = rp[min(c + 1, in_cols - 1)];
          }
        } else {
          const int c = 16 * s + 8 * j + 4 * h;                   // columns c .. c + 3
          if (vec_in == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + min(c, in_cols - 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) x[s][4 * j + e] = v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[s][4 * j + e] = rp[min(c + e, in_cols - 1)];
          }
        }
      }
    }
  };
  auto load_block = [&](const long b) __attribute__((always_inline)) {
    const long row = min(b * IN_ROWS + lr, n_rows - 1);
    const float* rp = in + row * ld_in;
    if (S0 == 8) load_steps(rp, std::integral_constant<int, 8>{});
    else if (S0 == 4) load_steps(rp, std::integral_constant<int, 4>{});
    else if (S0 == 2) load_steps(rp, std::integral_constant<int, 2>{});
    else load_steps(rp, std::integral_constant<int, 1>{});
  };
  // the arrived rows as the first layer's operand: masked, scaled by the row's power of two, split
  auto make_operand = [&](half8 (&bh)[8], half8 (&bl)[8], float& rs_b, auto S_) __attribute__((always_inline)) {
    constexpr int SS = decltype(S_)::value;
    if constexpr (PER_IN) {
      rs_b = TANH_RS;                                   // sines and cosines: the fixed scale 2^14
#pragma unroll
      for (int s = 0; s < SS; ++s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int c = 8 * s + 4 * (p >> 1) + 2 * h + (p & 1);
          float sn, cs;
          sincos_bounded(x[s][p] * in_scale, &sn, &cs);
          sn = c < in_cols ? sn * 16384.f : 0.f;
          cs = c < in_cols ? cs * 16384.f : 0.f;
          bh[s][2 * p] = (_Float16)sn;
          bl[s][2 * p] = (_Float16)(sn - (float)bh[s][2 * p]);
          bh[s][2 * p + 1] = (_Float16)cs;
          bl[s][2 * p + 1] = (_Float16)(cs - (float)bh[s][2 * p + 1]);
        }
      }
    } else {
      if (in_cols < 16 * SS) {                            // (workgroup-uniform: rows that fill their K-steps need no masks)
#pragma unroll
        for (int s = 0; s < SS; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) x[s][i] = in_vk(s, h, i) < in_cols ? x[s][i] : 0.f;
      }
      float mf = 0.f;
#pragma unroll
      for (int s = 0; s < SS; ++s)
#pragma unroll
        for (int i = 0; i < 8; i += 2) mf = __builtin_fmaxf(mf, __builtin_fmaxf(__builtin_fabsf(x[s][i]), __builtin_fabsf(x[s][i + 1])));   // v_max3_f32 |.|
      unsigned m = max_xor32(abs_bits(mf));              // the row's other half (a NaN in the row: the maximum of the rest; the NaN itself travels on)
      float sc;
      pow2_scale_bits<ACT_TOP>(m, sc, rs_b);
      typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int s = 0; s < SS; ++s) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const f2 v{x[s][i] * sc, x[s][i + 1] * sc};
          const half2v vh = __builtin_convertvector(v, half2v);
          const f2 lo{v[0] - (float)vh[0], v[1] - (float)vh[1]};
          const half2v vl = __builtin_convertvector(lo, half2v);
          bh[s][i] = vh[0], bh[s][i + 1] = vh[1], bl[s][i] = vl[0], bl[s][i + 1] = vl[1];
        }
      }
    }
  };

  long blk = (long)blockIdx.x * NW + wave;
  if (PREFETCH && blk < n_blocks) load_block(blk);
  for (int it = 0; blk < n_blocks; blk += stride, ++it) {
    half8 bh[8], bl[8];
    float rs_b;
    INS(it, 0);
    // (the column limits as values the compiler cannot prove loop-invariant: left invariant, every per-lane comparison against them
    //  -- a hundred 64-bit lane masks -- is hoisted out of this loop and kept in scalar registers, 400 of them spilled)
    asm volatile("" : "+s"(in_cols), "+s"(out_cols), "+s"(vec_in), "+s"(vec_out));
    if (!PREFETCH) load_block(blk);
    if (S0 == 8) make_operand(bh, bl, rs_b, std::integral_constant<int, 8>{});
    else if (S0 == 4) make_operand(bh, bl, rs_b, std::integral_constant<int, 4>{});
    else if (S0 == 2) make_operand(bh, bl, rs_b, std::integral_constant<int, 2>{});
    else make_operand(bh, bl, rs_b, std::integral_constant<int, 1>{});
    INS(it, 1);
    // ---- the next block's rows go on their way
    if (PREFETCH && blk + stride < n_blocks) load_block(blk + stride);
    INS(it, 2);

    // ---- the hidden layers
    for (int l = 0; l + 1 < nl; ++l) {
      const int S = a.L[l].S, T = a.L[l].T, act = a.L[l].act;
      const char* const pieces = lds + a.L[l].pieces + lane * 16;
      const float* const bt = bias_tab + l * 128 + 4 * h;
      if (!GEN || act == ACT_TANH) {
        f32x16 unused[4];
        in_layer_st<true>(S, T, pieces, bt, scale_tab[l] * rs_b * IN_TANH_ARG, bh, bl, unused);
        rs_b = TANH_RS;
      } else if constexpr (GEN) {
        // a hidden layer of another kind: float32 rows, activation, then the row's power-of-two scale and the split
        f32x16 keep[4];
        in_layer_st<false>(S, T, pieces, bt, scale_tab[l] * rs_b, bh, bl, keep);
        if (act != ACT_NONE) {
          if (T == 4) in_activate<4>(act, keep);
          else in_activate<1>(act, keep);
        }
        float sc = 16384.f;
        rs_b = TANH_RS;
        if (!act_bounded(act)) {
          unsigned m = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < T) {
#pragma unroll
              for (int r = 0; r < 16; ++r) m = max(m, abs_bits(keep[t][r]));
            }
          m = max_xor32(m);                                // the row's other half
          pow2_scale_bits<ACT_TOP>(m, sc, rs_b);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (t < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = keep[t][r] * sc;
              const int s2 = 2 * t + (r >> 3), i2 = r & 7;    // q = r >> 2, e = r & 3: K-step 2 t + (q >> 1), element 4 (q & 1) + e
              bh[s2][i2] = (_Float16)v;
              bl[s2][i2] = (_Float16)(v - (float)bh[s2][i2]);
            }
          }
      }
      INS(it, 3 + (l < 3 ? l : 3));
    }
    // ---- the last layer
    {
      const int l = nl - 1;
      const int S = a.L[l].S, T = a.L[l].T, act = a.L[l].act;
      f32x16 keep[4];
      in_layer_st<false>(S, T, lds + a.L[l].pieces + lane * 16, bias_tab + l * 128 + 4 * h, scale_tab[l] * rs_b, bh, bl, keep);
      if (act != ACT_NONE) {
        if (T == 4) in_activate<4>(act, keep);
        else in_activate<1>(act, keep);
      }
      INS(it, 6);
      const long row = blk * IN_ROWS + lr;
      if (row < n_rows) {
        float* const orow = out + row * ld_out;
        const int oc = out_cols;
        if (out_scale != 0.f) {
          // PeriodicOutput: virtual out-features (2 k, 2 k + 1) = (sin, cos) of angle k
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < T && 16 * t < oc) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int k = 16 * t + 4 * q + 2 * h;
                const float a0 = atan2f(keep[t][4 * q], keep[t][4 * q + 1]) * out_scale;
                const float a1 = atan2f(keep[t][4 * q + 2], keep[t][4 * q + 3]) * out_scale;
                if (vec_out >= 2) {
                  if (k < oc) *reinterpret_cast<f32x2*>(orow + k) = f32x2{a0, a1};
                } else {
                  if (k < oc) orow[k] = a0;
                  if (k + 1 < oc) orow[k + 1] = a1;
                }
              }
            }
        } else if (vec_out == 4) {                           // (the row length is a multiple of the access width: one test per access)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < T && 32 * t < oc) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q + 4 * h;
                if (c < oc) *reinterpret_cast<f32x4*>(orow + c) = f32x4{keep[t][4 * q], keep[t][4 * q + 1], keep[t][4 * q + 2], keep[t][4 * q + 3]};
              }
            }
        } else if (vec_out == 2) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < T && 32 * t < oc) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q + 4 * h;
                if (c < oc) *reinterpret_cast<f32x2*>(orow + c) = f32x2{keep[t][4 * q], keep[t][4 * q + 1]};
                if (c + 2 < oc) *reinterpret_cast<f32x2*>(orow + c + 2) = f32x2{keep[t][4 * q + 2], keep[t][4 * q + 3]};
              }
            }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < T && 32 * t < oc) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (c + e < oc) orow[c + e] = keep[t][4 * q + e];
              }
            }
        }
      }
    }
    INS(it, 7);
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
    if (K > 128 || N > 128) return fail(EMX_EUNSUPPORTED, "%s: layer %d is %d x %d: the streaming pass takes layers of up to 128 x 128", who, l, K, N);
    EMX_REQUIRE(desc->act[l] >= 0 && desc->act[l] <= ACT_LAST, "%s: layer %d: activation %d", who, l, desc->act[l]);
    InLayer& L = a.L[i];
    L.w_off = desc->w_off[l];
    L.b_off = desc->b_off[l];
    L.K = K;
    L.N = N;
    L.S = in_class_S(K);
    L.T = in_class_T(N);
    L.act = desc->act[l];
    L.kperm = (i == 0 && per_in) ? 1 : 0;
    L.nperm = (i == nl - 1 && per_out) ? 1 : 0;
    L.pieces = lds;
    lds += L.S * L.T * 2048;
  }
  a.bias_tab = lds;
  lds += nl * 128 * 4;
  a.scale_tab = lds;
  lds += (IN_MAXL + 8) * 4;   // (8 >= waves of a workgroup)
  if (lds > IN_LDS_MAX) return fail(EMX_EUNSUPPORTED, "%s: the operand pieces of layers [%d, %d) take %d bytes of LDS (%d at most)", who, layer_begin,
                                    layer_end, lds, IN_LDS_MAX);
  a.params = params;
  a.in = in;
  a.out = out;
  a.ld_in = ld_in;
  a.ld_out = ld_out;
  a.n_rows = n_rows;
  a.n_blocks = (n_rows + IN_ROWS - 1) / IN_ROWS;
  a.n_layers = nl;
  a.in_cols = in_cols;
  a.out_cols = out_cols;
  a.in_scale = per_in ? IN_TWO_PI / periodicity_in : 0.f;
  a.out_scale = per_out ? periodicity_out / IN_TWO_PI : 0.f;
  a.vec_in = vec_width(in, ld_in, in_cols, per_in ? 2 : 4);
  a.vec_out = vec_width(out, ld_out, out_cols, per_out ? 2 : 4);
  // one wave per SIMD (512 registers a wave: two operand sets, the next block's rows in flight).  A build with two waves per SIMD at
  // 256 registers spills (profiles/r06_infer_notes.txt).
  constexpr int waves = 4;
  const long wave_blocks = (a.n_blocks + waves - 1) / waves;
  const int grid = (int)std::min<long>(cu_count(), wave_blocks);
  bool gen = false;                                   // a hidden layer of another kind than tanh: the kernel that carries that code
  for (int i = 0; i + 1 < nl; ++i) gen = gen || a.L[i].act != ACT_TANH;
  typedef void (*Fn)(const InArgs);
  const Fn fn = gen ? (per_in ? &infer_kernel<true, 256, false, true> : &infer_kernel<false, 256, false, true>)
                    : (per_in ? &infer_kernel<true, 256, true, false> : &infer_kernel<false, 256, true, false>);
  static bool attr_set[2][2] = {{false, false}, {false, false}};
  if (!attr_set[gen][per_in]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, IN_LDS_MAX) != hipSuccess)
      return fail(EMX_ELAUNCH, "%s: hipFuncSetAttribute(%d bytes of LDS)", who, IN_LDS_MAX);
    attr_set[gen][per_in] = true;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), (size_t)lds, (hipStream_t)stream, a);
  return check_launch(who);
}

#ifdef IN_STAMP
int emx_debug_infer_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_in_stamps), sizeof(unsigned long long) * 8 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
