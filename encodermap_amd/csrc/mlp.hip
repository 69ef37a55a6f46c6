// K12: the dense encoder/decoder as fused row-block kernels on the 16-bit matrix cores with float32-accurate operands (gfx950).
//
// Replaces the per-layer Keras Dense ops of SequentialModel (encodermap/models/models.py:3179-3220: y = act(x W + b),
// W is [in, out]) and their autodiff (tape.gradient, models.py:3392-3393).  The layers are tiny (<= 3e5 parameters)
// and B is a few thousand rows, so per-layer GEMM launches are launch- and tail-bound (hipBLASLt: ~36 launches,
// 330 us per step at B=4096, D=128).  Here:
//
//   chain_kernel      one workgroup = 16 batch rows pushed through a RANGE of layers (forward, backward-data, or forward +
//                     auto cost + backward in one launch); the row block never leaves the CU -- its activations live in
//                     LDS images, one per layer.
//   mlp_wgrad_kernel  dW_l = X_l^T delta_l and db_l for ALL layers in one launch, split over row slices; the slice
//                     partials are summed in a fixed order by the optimizer (or mlp_wgrad_reduce_kernel): deterministic.
//
// Arithmetic.  The f32 matrix instruction of gfx950 runs at the vector rate (1/16 of the 16-bit matrix rate,
// MI355X_MICROARCH.md), and round 1 measured these kernels at 0.2 of even that roof.  Every float32 operand is therefore
// carried as a SCALED f16 PAIR (wop.h): x = (h + l) / s with s a power of two, h = f16(s x), l = f16(s x - h), 22
// significant bits, and a product runs as the three instructions hh + hl + lh of v_mfma_f32_16x16x32_f16 (row-block
// kernels) / v_mfma_f32_32x32x16_f16 (weight gradients) with exact partial products in the f32 accumulator.  Scales:
//   weights      one per 32 x 32 block of W (wop.h), kept current by the optimizer;
//   activations  one per (row, 32 columns) in the LDS images -- a K-step of the matrix instruction covers exactly 32
//                columns produced by ONE wave, so the scale needs no cross-wave reduction, and every row's numbers are
//                independent of which rows share its block (row ranges and batch sizes do not change a row's result);
//   wgrad        one per (16 rows, 32 columns) block (aop.h): the row-scaled pair with its exponents shifted, exact.
// The unscaling is one fused multiply-add per accumulator element and K-step (acc += d * (1/s_a)(1/s_w)).
//
// Lane maps (cdna_hip_programming.md section 3; checked with integer data by tests/test_mlp_gpu.py):
//   16x16x32: A[m = lane & 15][k = 8 (lane >> 4) + e], B[k][n = lane & 15], D[row = 4 (lane >> 4) + reg][col = lane & 15];
//   the k labels inside a K-step are free as long as A and B agree -- here kk(g, e) = 4 g + e (e < 4), 16 + 4 g + e - 4.
#include "aop.h"
#include "cost.h"
#include "emx_common.h"
#include "finish.h"
#include "wop.h"

namespace {

using namespace emx;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;            // weight-gradient kernels
constexpr int NTC = 256;           // row-block chain kernels: 4 waves, one per SIMD, two 16-column MFMA tiles each
constexpr int R = 16;              // batch rows per workgroup
constexpr int MAXL = EMX_MAX_LAYERS;
constexpr int MAXG = 2 * MAXL + 1;   // GEMMs per launch
constexpr int MAXU = 120;            // units (128 output columns x 128 k) per launch

// tanh(x) = 1 - 2 / (e^{2x} + 1): two transcendental + three plain VALU instructions, absolute error <= ~1.5e-7 over the
// whole range (v_exp_f32 and v_rcp_f32 are 1-ulp; e = inf gives rcp = 0 -> 1, e = 0 gives -1), which is the float32
// resolution of the activations.
__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);        // e^{2x}
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// maximum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane of the row
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp_f32<0xB1>(x));    // quad_perm [1, 0, 3, 2]
  x = fmaxf(x, dpp_f32<0x4E>(x));    // quad_perm [2, 3, 0, 1]
  x = fmaxf(x, dpp_f32<0x124>(x));   // row_ror:4
  x = fmaxf(x, dpp_f32<0x128>(x));   // row_ror:8
  return x;
}

// ---------------------------------------------------------------------------------------------------------
// Row-block chain kernel (forward, backward-data and the fused forward + cost + backward launch share it).
//
// LDS images.  The activations (forward) or pre-activation gradients (backward) of the row block, one image per layer
// boundary: two f16 planes stored TRANSPOSED, [k = column][16 rows] (32 bytes per column; the lo plane behind the hi plane),
// and the per-(32-column group, row) reciprocal scales as floats.  A K-step's A fragment is then two ds_read_b64_tr_b16 per
// plane at  plane + 1024 ks + 8 lane  (+ 512): lane (m, g) receives row m, columns 32 ks + 4 g + q and 32 ks + 16 + 4 g + q --
// the k order kk(g, e) of wop.h, conflict-free (a 32-lane half reads 256 contiguous bytes); the epilogue's D tile (lane =
// column, registers = 4 consecutive rows) is one 8-byte LDS write per tile and plane.
//
// Weight stream.  Every workgroup needs every weight (64 KB per 128 x 128 layer), so the kernel is built around that
// stream as in round 1: the host flattens the layer range into UNITS (layer, pass of 128 columns, chunk of 128 k; one unit
// = 24 matrix instructions per wave), everything a unit needs from memory -- 16 operand pieces, the weight scales, the
// epilogue operands -- travels in one package NB - 1 units ahead in a register ring, results are stored one unit late,
// and every iteration issues the same vector-memory sequence (no conditional load or exec-masked store), so that every
// s_waitcnt the compiler inserts is for the OLDEST package only (vmcnt is one in-order counter for loads and stores).
// ---------------------------------------------------------------------------------------------------------
enum { U_FIRST = 1, U_LAST = 2, U_LAYER_END = 4 };
enum { S_ACT = 1, S_JOIN = 2, S_BWD = 4 };

struct GemmRec {          // one layer of the range in one direction
  int op_kb;              // first 1-KB piece of the operand block inside the weight store
  int Q, T;               // K-steps (32 k) and column groups (32 columns) of the operand
  int ws_off, ws_ss, ws_sw;   // weight scales: float index of block (K-step 0, group 0), strides per K-step / column group
  int n;                  // output columns
  int sflags;             // S_*
  const float* aux;       // epilogue operand: the bias (forward), rows of h for the activation derivative (backward)
  int aux_rs;             // its row stride (0: one broadcast row)
  int src_pl, src_sc;     // LDS byte offsets of the source image: planes, scales
  int dst_pl, dst_sc;     // destination image (-1: none)
  int dst_f32, dst_f32_pitch;   // float32 copy of the output rows in LDS, [16][pitch] (-1: none): cost input, tanh' of the fused backward
  int h_f32, h_f32_pitch;       // fused backward units: float32 copy of h for the activation derivative (-1: through aux)
  float* out;             // [*, n] float32 rows in HBM (null: not kept)
  AopTensor aop;          // fragments of the output for the weight gradients (frag null: not kept)
};
struct UnitRec {
  unsigned char gemm, pass, chunk, flags;
};

enum { CHAIN_FWD = 0, CHAIN_BWD = 1, CHAIN_FUSED = 2 };

struct ChainArgs {
  GemmRec g[MAXG];
  UnitRec u[MAXU];
  int nu;
  const char* wop;        // weight operand store
  const float* in;        // forward: input rows; backward: d loss / d h_top
  int ld_in, in_cols, in_Q;
  int in_pl, in_sc;       // image that receives the row block
  AopTensor in_aop;       // its fragments for the weight gradients (forward from layer 0: the network input; backward: delta_top)
  int row0, row_end;
  // backward only: the top of the range becomes delta_top = (in [+ join]) * act'(h_top), stored to top_out (may be null)
  const float* top_h;
  float* top_out;
  int top_flags;
  // join: rows added to dh of one layer (S_JOIN): dadd and / or the latent gradient of the all-pairs cost
  const float* dadd;
  emx_latent_grad lg;
  int join_n;             // width of the joined layer (0: no join)
  int lat_off;            // LDS byte offset of the joined rows [16][join_n]
  // forward, optional auto cost on the rows of the last layer (replaces emx_periodic_cost_fwd_bwd behind it)
  CostArgs cost;
  int y_f32, y_pitch;     // float32 copy of the last forward layer's rows: the cost's input, receives d cost / d output in place
  int tgt_off;            // LDS byte offset of the target rows [16][cost.n]
  // fused mode: the cost runs behind unit cost_at; d cost / d output is re-staged into the image (top_pl, top_sc), the
  // source of the first backward unit, and into top_aop
  int cost_at, top_pl, top_sc, top_Q;
  AopTensor top_aop;
};

struct PendStore {        // results of a unit's epilogue, stored at the head of the next iteration through buffer
  float v[8];             // resources whose bounds check drops what does not exist (no exec-masked vector-memory instruction)
  half4 fr[4];
  float rs;
  const float* out_base;
  int out_bytes, out_n;
  unsigned out_voff[2];   // per tile: byte offset of (row 4 g, the tile's column), out of bounds when the column does not exist
  const char* fr_base;
  int fr_bytes;
  const float* rs_base;
};

template <int V>
struct IC_ {
  static constexpr int value = V;
};

template <int MODE, int NB>
__global__ __launch_bounds__(NTC) void chain_kernel(const ChainArgs a) {
  constexpr bool BWD = MODE == CHAIN_BWD;
  constexpr int NE = BWD ? 8 : 2;   // epilogue operands per lane: h of 4 rows x 2 tiles (backward) or the bias of 2 columns
  extern __shared__ __align__(16) char lds[];
  __shared__ float cred[NTC / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int r0 = a.row0 + blockIdx.x * R;
  const int rb = r0 >> 4;   // row block of the weight-gradient fragments (row0 is a multiple of 16 whenever they are kept)
  const int rows_valid = min(R, a.row_end - r0);
  const int nu = a.nu;
  const float* wsf = reinterpret_cast<const float*>(a.wop);
  float* lat = reinterpret_cast<float*>(lds + a.lat_off);
  float* tgt = reinterpret_cast<float*>(lds + a.tgt_off);

  if (BWD && a.join_n > 0) {
    // rows joined into dh of one layer: dadd and / or the latent gradient of the all-pairs cost (one wave per element,
    // lanes over the per-tile slots, fixed order)
    const int N = a.join_n;
    if (a.lg.partials != nullptr) {
      for (int el = wave; el < R * N; el += NTC / 64) {
        const int r = el / N, c = el - r * N;
        const bool rok = r0 + r < a.row_end;
        float v = 0.f;
        if (rok) {
          const float* __restrict__ P = a.lg.partials + (long)(r0 + r - a.lg.row_base) * N + c;
          for (int sidx = lane; sidx < a.lg.n_slots; sidx += 64) v += P[(long)sidx * a.lg.slot_stride];
        }
        v = wave_sum(v);
        if (lane == 0) {
          float t = 0.f;
          if (rok) {
            const long gi = (long)(r0 + r) * N + c;
            t = fmaf(a.lg.center_grad, a.lg.Z[gi], a.lg.gscale * v);
            if (a.dadd != nullptr) t += a.dadd[gi];
          }
          lat[el] = t;
        }
      }
    } else {
      for (int el = tid; el < R * N; el += NTC) {
        const int r = el / N, c = el - r * N;
        lat[el] = r0 + r < a.row_end ? a.dadd[(long)(r0 + r) * N + c] : 0.f;
      }
    }
    __syncthreads();
  }

  // ---- staging a row block [16][ncols] of float32 values into an image: wave w takes the column groups w, w + 4, ...;
  //      lane (c = lane & 31, hf = lane >> 5) the column 32 group + c, rows 8 hf .. 8 hf + 7 -- its 8 halves are 16
  //      contiguous bytes of the transposed plane AND, with lane as the fragment's lane, the weight-gradient fragment (aop.h).
  const int sc_c = lane & 31, sc_hf = lane >> 5;
  auto stage_finish = [&](const float (&v)[8], int grp, int Q, int pl, int sc, const AopTensor& ao) {
    // per-row scale over the group's 32 columns (lanes of one half)
    float s[8], rs[8];
    float bm = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float m = row16_max(fabsf(v[e]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      pow2_scale(m, s[e], rs[e]);
      bm = fmaxf(bm, m);
    }
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e] * s[e];
      hi[e] = (_Float16)t;
      lo[e] = (_Float16)(t - (float)hi[e]);
    }
    const int k = 32 * grp + sc_c;
    *reinterpret_cast<half8*>(lds + pl + k * 32 + sc_hf * 16) = hi;
    *reinterpret_cast<half8*>(lds + pl + Q * 1024 + k * 32 + sc_hf * 16) = lo;
    if (sc_c == 0) {
      float* dst = reinterpret_cast<float*>(lds + sc) + grp * 16 + 8 * sc_hf;
      *reinterpret_cast<f32x4*>(dst) = f32x4{rs[0], rs[1], rs[2], rs[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{rs[4], rs[5], rs[6], rs[7]};
    }
    if (ao.frag != nullptr) {
      // block scale (16 rows x 32 columns) for the weight gradients: the row-scaled pair with its exponents shifted
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      float sb, rsb;
      pow2_scale(bm, sb, rsb);
      half8 fh, fl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 ratio = (_Float16)(sb * rs[e]);   // a power of two <= 1 (0 beyond 2^-24: such a row is below the block's resolution)
        fh[e] = hi[e] * ratio;
        fl[e] = lo[e] * ratio;
      }
      char* f = ao.frag + ((long)rb * ao.CG + grp) * 2048 + lane * 16;
      *reinterpret_cast<half8*>(f) = fh;
      *reinterpret_cast<half8*>(f + 1024) = fl;
      if (lane == 0) ao.rscale[(long)rb * ao.CG + grp] = rsb;
    }
  };
  // the row block from HBM (forward: input rows; backward: delta_top = (dtop [+ join]) * act'(h_top))
  const int sK = a.in_cols;
  auto stage_load = [&](int grp, float (&v)[8]) {
    const int col = 32 * grp + sc_c, cc = min(col, sK - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 8 * sc_hf + e;
      const long grow = (long)min(r0 + row, a.row_end - 1);
      float x = a.in[grow * a.ld_in + cc];
      if (BWD) {
        if (a.top_flags & S_JOIN) x += lat[row * sK + cc];
        if (a.top_flags & S_ACT) {
          const float hv = a.top_h[grow * sK + cc];
          x *= fmaf(-hv, hv, 1.f);
        }
      }
      v[e] = (col < sK && row < rows_valid) ? x : 0.f;
    }
  };
  auto stage_top_out = [&](int grp, const float (&v)[8]) {
    if (!BWD || a.top_out == nullptr) return;
    const int col = 32 * grp + sc_c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 8 * sc_hf + e;
      if (col < sK && row < rows_valid) a.top_out[(long)(r0 + row) * sK + col] = v[e];
    }
  };

  // ---- the register ring
  half8 wp[NB][4][2][2];   // [slot][K-step][tile][plane]
  float ep[NB][NE];
  float wsc[NB];

  struct Req {             // what a request burst needs (wave-uniform)
    const char* pbase;     // first piece of this wave's column group in this chunk
    int kq;                // K-steps in this chunk, 1..4
    const float* aux0;     // epilogue operand row base
    int aux_rs, aux_n, col0;
    const float* ws;       // weight scale of K-step 0
    int ws_ss;
  };
  auto make_req = [&](int uu) {
    const UnitRec U = a.u[max(0, min(uu, nu - 1))];
    const GemmRec& G = a.g[U.gemm];
    const int Q = G.Q, cg = min(4 * U.pass + wave, G.T - 1), ks0 = 4 * U.chunk;
    Req q;
    q.kq = min(4, Q - ks0);
    q.pbase = a.wop + ((long)G.op_kb + ((long)cg * Q + ks0) * 4) * 1024;
    q.aux0 = G.aux;
    q.aux_rs = G.aux_rs;
    q.aux_n = G.n;
    q.col0 = 128 * U.pass + 32 * wave;
    q.ws = wsf + G.ws_off + (long)ks0 * G.ws_ss + (long)cg * G.ws_sw;
    q.ws_ss = G.ws_ss;
    return q;
  };
  auto request_head = [&](const Req& q, float (&ee)[NE], float& wq) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ccol = min(q.col0 + 16 * t + n, q.aux_n - 1);
      if (BWD) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ee[4 * t + reg] = q.aux0[(long)min(r0 + 4 * g + reg, a.row_end - 1) * q.aux_rs + ccol];
      } else {
        ee[t] = q.aux0[ccol];
      }
    }
    wq = q.ws[(long)min(lane & 3, q.kq - 1) * q.ws_ss];
  };
  auto request_pieces = [&](const Req& q, int s, half8 (&w)[2][2]) {
    const char* p = q.pbase + (long)min(s, q.kq - 1) * 4096 + lane * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) w[t][pl] = *reinterpret_cast<const half8*>(p + (t * 2 + pl) * 1024);
  };

  PendStore pend;
#pragma unroll
  for (int i = 0; i < 8; ++i) pend.v[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) pend.fr[i] = half4{0, 0, 0, 0};
  pend.rs = 0.f;
  pend.out_base = wsf;
  pend.out_bytes = 0;
  pend.out_n = 0;
  pend.out_voff[0] = pend.out_voff[1] = 0x80000000u;
  pend.fr_base = a.wop;
  pend.fr_bytes = 0;
  pend.rs_base = wsf;
  auto flush_pend = [&]() {
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend.out_base), 0, pend.out_bytes, 0x00020000);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pend.v[4 * t + reg]), ro, pend.out_voff[t] + reg * pend.out_n * 4, 0, 0);
    __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pend.fr_base), 0, pend.fr_bytes, 0x00020000);
    const unsigned fvoff = (unsigned)((32 * (g >> 1) + n) * 16 + (g & 1) * 8);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pend.fr[2 * t + pl]), rf, fvoff + pl * 1024 + t * 256, 0, 0);
      }
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend.rs_base), 0, pend.fr_bytes ? 4 : 0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pend.rs), rr, lane == 0 ? 0u : 0x80000000u, 0, 0);
    pend.out_bytes = 0;
    pend.fr_bytes = 0;
  };

  // auto cost on the last forward layer's rows (float32 copy in LDS): 16 threads per row; d cost / d output goes to HBM
  // and, in place, into the copy
  auto cost_phase = [&]() {
    const CostArgs& cst = a.cost;
    const int row = tid >> 4, l16 = tid & 15;
    const bool rok = row < rows_valid;
    const long grow = (long)min(r0 + row, a.row_end - 1);
    float* yrow = reinterpret_cast<float*>(lds + a.y_f32) + row * a.y_pitch;
    const float* trow = tgt + row * cst.n;
    float* drow = (rok && cst.dY != nullptr) ? cst.dY + grow * cst.ldy : nullptr;
    float lsum = 0.f;
    if (cst.variant == EMX_MEAN_NORM) {
      float ss = 0.f;
      for (int k = l16; k < cst.n; k += 16) {
        float sn, cs, delta, dd;
        cost_elem(cst, yrow, trow, k, sn, cs, delta, dd);
        ss = fmaf(delta, delta, ss);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);   // stays inside the row's 16 lanes
      const float nrm = sqrtf(ss);
      const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
      if (l16 == 0 && rok) lsum = nrm;
      for (int k = l16; k < cst.n; k += 16) {
        float sn, cs, delta, dd;
        cost_elem(cst, yrow, trow, k, sn, cs, delta, dd);
        const float gq = rok ? cst.gnorm * delta * inv * dd : 0.f;
        cost_store(cst, drow, nullptr, k, gq, sn, cs);
        cost_store(cst, yrow, nullptr, k, gq, sn, cs);
      }
    } else {
      for (int k = l16; k < cst.n; k += 16) {
        float sn, cs, delta, dd;
        cost_elem(cst, yrow, trow, k, sn, cs, delta, dd);
        float gq;
        if (cst.variant == EMX_MEAN_ABS) {
          if (rok) lsum += fabsf(delta);
          gq = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
        } else {
          if (rok) lsum = fmaf(delta, delta, lsum);
          gq = 2.f * delta * dd;
        }
        gq = rok ? cst.gnorm * gq : 0.f;
        cost_store(cst, drow, nullptr, k, gq, sn, cs);
        cost_store(cst, yrow, nullptr, k, gq, sn, cs);
      }
    }
    const float tsum = block_sum<NTC>(lsum, cred);
    if (tid == 0) cst.ws[blockIdx.x] = tsum * cst.gnorm;
  };

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float sv[8];
  float tv[4] = {0.f, 0.f, 0.f, 0.f};
  const bool with_cost = !BWD && a.cost.ws != nullptr;
  Req Lq = make_req(0);
  UnitRec XU = a.u[0];

  // Iteration ui requests the package of unit ui + NB - 1 into ring slot r and executes unit ui from slot r + 1; the first
  // NB - 1 iterations (ui < 0) only request.
  auto unit = [&](auto RC, const int ui) {
    constexpr int r = decltype(RC)::value;
    constexpr int rx = (r + 1) % NB;
    const bool run = ui >= 0 && ui < nu;
    const GemmRec& X = a.g[XU.gemm];
    const int Q = X.Q;
    const int ks0 = 4 * XU.chunk;
    const int kq = run ? min(4, Q - ks0) : 0;
    const int cgw = 4 * XU.pass + wave;
    const bool mine = cgw < X.T;

    flush_pend();
    if (ui == -(NB - 1)) {
      // the row block itself travels ahead of the first package and reaches LDS behind it
      if (wave < a.in_Q) stage_load(wave, sv);
      if (with_cost) {
        const int c = 4 * (tid & 15), row = tid >> 4;
        const long grow = (long)min(r0 + row, a.row_end - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = a.cost.target[grow * a.cost.ldt + min(c + i, a.cost.n - 1)];
      }
    }
    request_head(Lq, ep[r], wsc[r]);

    if (run && (XU.flags & U_FIRST)) acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* apl = lds + X.src_pl + ks0 * 1024 + lane * 8;
    const float* asc = reinterpret_cast<const float*>(lds + X.src_sc) + ks0 * 16 + 4 * g;
    const float wq = wsc[rx];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (mine && s < kq) {
        const half4 h0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + s * 1024));
        const half4 h1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + s * 1024 + 512));
        const half4 l0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + Q * 1024 + s * 1024));
        const half4 l1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + Q * 1024 + s * 1024 + 512));
        const f32x4 rsa = *reinterpret_cast<const f32x4*>(asc + s * 16);
        const half8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const half8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        const float rsw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wq), s));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wp[rx][s][t][1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wp[rx][s][t][0], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wp[rx][s][t][0], d, 0, 0, 0);
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[t][reg] = fmaf(d[reg], rsa[reg] * rsw, acc[t][reg]);
        }
      }
      request_pieces(Lq, s, wp[r][s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next iteration's records: scalar loads issued here land behind the epilogue
    const Req Ln = make_req(ui + NB);
    const UnitRec XUn = a.u[max(0, min(ui + 1, nu - 1))];
    Lq = Ln;

    if (ui == -(NB - 1)) {
      if (wave < a.in_Q) {
        stage_top_out(wave, sv);
        stage_finish(sv, wave, a.in_Q, a.in_pl, a.in_sc, a.in_aop);
      }
      for (int grp = wave + 4; grp < a.in_Q; grp += 4) {
        float v[8];
        stage_load(grp, v);
        stage_top_out(grp, v);
        stage_finish(v, grp, a.in_Q, a.in_pl, a.in_sc, a.in_aop);
      }
      if (with_cost) {
        const int c = 4 * (tid & 15), row = tid >> 4;
        const long grow = (long)min(r0 + row, a.row_end - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c + i < a.cost.n) tgt[row * a.cost.n + c + i] = tv[i];
        for (int cc = c + 64; cc < a.cost.n; cc += 64)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cc + i < a.cost.n) tgt[row * a.cost.n + cc + i] = a.cost.target[grow * a.cost.ldt + cc + i];
      }
      __syncthreads();
    }
    if (!run) {
      XU = XUn;
      return;
    }

    // ---- epilogue
    if (XU.flags & U_LAST) {
      const int sflags = X.sflags;
      const int cols = X.n - 128 * XU.pass;           // columns of this pass that exist (may exceed 128)
      const int colw = 32 * wave;                     // this wave's first column inside the pass
      const bool ubwd = (sflags & S_BWD) != 0;
      float v[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int cp = colw + 16 * t + n;             // column inside the pass
        const int col = 128 * XU.pass + cp;
        const bool cok = mine && cp < cols;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = 4 * g + reg;
          float x = acc[t][reg];
          if (!ubwd) {
            x += ep[rx][BWD ? 0 : t];
            if (sflags & S_ACT) x = tanh_f32(x);
          } else {
            if (sflags & S_JOIN) x += lat[row * X.n + min(col, X.n - 1)];
            if (sflags & S_ACT) {
              float hv;
              if (MODE == CHAIN_FUSED)
                hv = reinterpret_cast<const float*>(lds + X.h_f32)[row * X.h_f32_pitch + min(col, X.n - 1)];
              else
                hv = ep[rx][BWD ? 4 * t + reg : 0];
              x *= fmaf(-hv, hv, 1.f);
            }
          }
          v[t][reg] = (cok && row < rows_valid) ? x : 0.f;
        }
      }
      if (mine) {
        if (X.dst_f32 >= 0) {
          float* f = reinterpret_cast<float*>(lds + X.dst_f32);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int col = 128 * XU.pass + colw + 16 * t + n;
            if (col < X.n) {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) f[(4 * g + reg) * X.dst_f32_pitch + col] = v[t][reg];
            }
          }
        }
        if (X.dst_pl >= 0) {
          // per-row scale over the wave's 32 columns, split, transposed image write (lane = column, 4 rows = 8 bytes)
          float s[4], rs[4];
          half4 hi[2], lo[2];
          float bm = 0.f;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float m = row16_max(fmaxf(fabsf(v[0][reg]), fabsf(v[1][reg])));
            pow2_scale(m, s[reg], rs[reg]);
            bm = fmaxf(bm, m);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const float x = v[t][reg] * s[reg];
              hi[t][reg] = (_Float16)x;
              lo[t][reg] = (_Float16)(x - (float)hi[t][reg]);
            }
          const int Qd = (X.n + 31) >> 5;
          char* dp = lds + X.dst_pl + (128 * XU.pass + colw + n) * 32 + 8 * g;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            *reinterpret_cast<half4*>(dp + t * 512) = hi[t];
            *reinterpret_cast<half4*>(dp + Qd * 1024 + t * 512) = lo[t];
          }
          if (n == 0) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(lds + X.dst_sc) + cgw * 16 + 4 * g) = f32x4{rs[0], rs[1], rs[2], rs[3]};
          if (X.aop.frag != nullptr) {
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            float sb, rsb;
            pow2_scale(bm, sb, rsb);
            half4 ratio;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ratio[reg] = (_Float16)(sb * rs[reg]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              pend.fr[2 * t] = hi[t] * ratio;
              pend.fr[2 * t + 1] = lo[t] * ratio;
            }
            pend.rs = rsb;
            pend.fr_base = X.aop.frag + ((long)rb * X.aop.CG + cgw) * 2048;
            pend.fr_bytes = 2048;
            pend.rs_base = X.aop.rscale + (long)rb * X.aop.CG + cgw;
          }
        }
        if (X.out != nullptr) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pend.v[4 * t + reg] = v[t][reg];
          const int col = 128 * XU.pass + colw + n;
          pend.out_base = X.out + (long)r0 * X.n;
          pend.out_n = X.n;
          pend.out_bytes = max(0, rows_valid) * X.n * 4;
          // a column beyond n would alias the next row inside the buffer's bounds: such a lane's offset is pushed out of them
#pragma unroll
          for (int t = 0; t < 2; ++t) pend.out_voff[t] = col + 16 * t < X.n ? (unsigned)((4 * g * X.n + col + 16 * t) * 4) : 0x80000000u;
        }
      }
    }
    if (XU.flags & U_LAYER_END) __syncthreads();
    if (MODE == CHAIN_FUSED && ui == a.cost_at) {
      // the forward part is complete: cost; d cost / d output -> HBM and, re-staged, the first backward unit's image
      cost_phase();
      __syncthreads();
      const CostArgs& cst = a.cost;
      const float* yb = reinterpret_cast<const float*>(lds + a.y_f32);
      const int wout = cst.ldy;
      for (int grp = wave; grp < a.top_Q; grp += 4) {
        const int col = 32 * grp + sc_c;
        const bool covered = (col >= cst.off_sin && col < cst.off_sin + cst.n) ||
                             (cst.periodic_output && col >= cst.off_cos && col < cst.off_cos + cst.n);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (covered && col < wout) ? yb[(8 * sc_hf + e) * a.y_pitch + col] : 0.f;
        stage_finish(v, grp, a.top_Q, a.top_pl, a.top_sc, a.top_aop);
      }
      __syncthreads();
    }
    XU = XUn;
  };

  for (int ui = -(NB - 1);; ui += NB) {
    unit(IC_<0>{}, ui);
    if (ui + 1 >= nu) break;
    unit(IC_<1>{}, ui + 1);
    if (ui + 2 >= nu) break;
    if constexpr (NB >= 3) {
      unit(IC_<2>{}, ui + 2);
      if (ui + 3 >= nu) break;
    }
  }
  flush_pend();
  if (MODE == CHAIN_FWD && with_cost) cost_phase();
}

// the operand store from the flat parameter buffer: one wave per 32 x 32 block (the optimizer keeps it current afterwards)
__global__ __launch_bounds__(64) void mlp_mirror_kernel(const WopArgs wa, const float* __restrict__ params, char* __restrict__ store) {
  __shared__ __align__(16) _Float16 img[2048];
  const int lane = threadIdx.x;
  int l, kb, nb;
  wop_item(wa, blockIdx.x, l, kb, nb);
  float wv[16];
  wop_load_block(wa, params, l, kb, nb, lane, wv);
  wop_write_block(wa, store, l, kb, nb, wv, lane, img);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients: partial[slice][l] = X_l^T delta_l over the slice's rows, from the fragments the row-block kernels left
// (aop.h).  grid = (128 x 128 output blocks of all layers) x slices; a wave owns a 64 x 64 quadrant = 2 x 2 tiles of
// v_mfma_f32_32x32x16_f16 whose reduction index runs over the 16 rows of ONE row block, so both operands of an instruction
// carry one scale each and the tile is unscaled by a wave-uniform factor when it is added to the accumulator.  Operands go
// straight from HBM/L2 to the matrix instruction: one 16-byte load per lane and fragment, no LDS, no barrier.
// db comes from the same fragments through a ones operand (two instructions per column group, row 0 of the result).
// ---------------------------------------------------------------------------------------------------------
struct WgradDev {
  int l0, l1;
  int K[MAXL], N[MAXL];
  long w_off[MAXL], b_off[MAXL];
  AopTensor X[MAXL], D[MAXL];
  int tile_begin[MAXL + 1];  // prefix sum of 128x128 output blocks per layer (relative to l0)
  float* partial;            // [n_slices][n_params]
  long n_params;
  int rb0, rb_end, rbs_per_slice;
  int has_fin;               // the last blockIdx.x column is not a dW block: its y = 0 workgroup runs the step's scalar epilogue
  FinishArgs fin;
};

__global__ __launch_bounds__(NT) void mlp_wgrad_kernel(const WgradDev a) {
  __shared__ float fin_red[EMX_MAX_SUM_JOBS][FIN_NT / 64];
  __shared__ float fin_tl[64];
  if (a.has_fin && blockIdx.x == gridDim.x - 1) {
    // The step's scalar epilogue (cost sums, history, counters) rides along as one extra workgroup: its inputs are complete
    // before this launch starts, and a launch of its own costs ~6.5 us of the step.
    if (blockIdx.y == 0) finish_step_body<emx::FQ_EMBEDDED>(a.fin, fin_red, fin_tl);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int l = a.l0;
  while (l + 1 < a.l1 && (int)blockIdx.x >= a.tile_begin[l + 1 - a.l0]) ++l;
  const int tb = blockIdx.x - a.tile_begin[l - a.l0];
  const int Kin = a.K[l], N = a.N[l];
  const int nbn = (N + 127) >> 7;
  const int bi = tb / nbn, bj = tb - bi * nbn;
  const int ig0 = 4 * bi + 2 * (wave >> 1), jg0 = 4 * bj + 2 * (wave & 1);   // first column groups of the wave's quadrant
  const AopTensor X = a.X[l], D = a.D[l];
  const int CGx = X.CG, CGd = D.CG;
  const int rbs = a.rb0 + blockIdx.y * a.rbs_per_slice;
  const int rbe = min(rbs + a.rbs_per_slice, a.rb_end);
  float* __restrict__ out = a.partial + (long)blockIdx.y * a.n_params;
  const bool iact[2] = {ig0 < CGx, ig0 + 1 < CGx}, jact[2] = {jg0 < CGd, jg0 + 1 < CGd};
  const bool do_bias = bi == 0 && (wave >> 1) == 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][v][i] = 0.f;
  float bacc[2] = {0.f, 0.f};
  half8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (_Float16)1.f;

  if (iact[0] && jact[0]) {
    struct Frags {
      half8 x[2][2], d[2][2];   // [group][plane]
      float sx[2], sd[2];
    };
    auto load = [&](int rbk, Frags& f) {
      const int rbc = min(rbk, a.rb_end - 1);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ig = min(ig0 + u, CGx - 1), jg = min(jg0 + u, CGd - 1);
        const char* px = X.frag + ((long)rbc * CGx + ig) * 2048 + lane * 16;
        const char* pd = D.frag + ((long)rbc * CGd + jg) * 2048 + lane * 16;
        f.x[u][0] = *reinterpret_cast<const half8*>(px);
        f.x[u][1] = *reinterpret_cast<const half8*>(px + 1024);
        f.d[u][0] = *reinterpret_cast<const half8*>(pd);
        f.d[u][1] = *reinterpret_cast<const half8*>(pd + 1024);
        f.sx[u] = X.rscale[(long)rbc * CGx + ig];
        f.sd[u] = D.rscale[(long)rbc * CGd + jg];
      }
    };
    auto mma = [&](const Frags& f) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          if (iact[u] && jact[v]) {
            f32x16 d;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = 0.f;
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.x[u][0], f.d[v][1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.x[u][1], f.d[v][0], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.x[u][0], f.d[v][0], d, 0, 0, 0);
            const float sc = f.sx[u] * f.sd[v];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][v][i] = fmaf(d[i], sc, acc[u][v][i]);
          }
        }
      if (do_bias) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          if (jact[v]) {
            f32x16 d;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = 0.f;
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, f.d[v][1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, f.d[v][0], d, 0, 0, 0);
            bacc[v] = fmaf(d[0], f.sd[v], bacc[v]);
          }
        }
      }
    };
    // two fragment sets: set A is loop-carried and reloaded right behind the instructions that consumed it
    Frags fa, fb;
    load(rbs, fa);
    for (int rbk = rbs; rbk < rbe; rbk += 2) {
      load(rbk + 1, fb);
      mma(fa);
      load(rbk + 2, fa);
      if (rbk + 1 < rbe) mma(fb);
    }
    // store: tile (u, v), D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        if (iact[u] && jact[v]) {
          const int jj = 32 * (jg0 + v) + (lane & 31);
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int ii = 32 * (ig0 + u) + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (ii < Kin && jj < N) out[a.w_off[l] + (long)ii * N + jj] = acc[u][v][reg];
          }
        }
      }
    if (do_bias && lane < 32) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int jj = 32 * (jg0 + v) + lane;
        if (jact[v] && jj < N) out[a.b_off[l] + jj] = bacc[v];
      }
    }
  }
}

// g[e] = sum over slices (fixed order) of partial[s][e] for the parameter ranges of layers [l0, l1)
__global__ __launch_bounds__(NT) void mlp_wgrad_reduce_kernel(const float* __restrict__ partial, long n_params, int n_slices, long lo,
                                                              long hi, float* __restrict__ g) {
  long e = lo + (long)blockIdx.x * NT + threadIdx.x;
  if (e >= hi) return;
  float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // fixed association -> deterministic
  int k = 0;
  for (; k + 8 <= n_slices; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] += partial[(long)(k + u) * n_params + e];
  }
  for (int u = 0; k < n_slices; ++k, ++u) part[u] += partial[(long)k * n_params + e];
  g[e] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int check_desc(const emx_mlp_desc* desc, const char* who) {
  EMX_REQUIRE(desc, "%s: null descriptor", who);
  EMX_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= MAXL, "%s: %d layers not supported (1..%d)", who, desc->n_layers, MAXL);
  for (int l = 0; l <= desc->n_layers; ++l) EMX_REQUIRE(desc->width[l] >= 1, "%s: layer width must be >= 1", who);
  return EMX_OK;
}

int set_lds(const void* fn, size_t bytes, const char* who) {
  if (bytes > 160 * 1024) return emx::fail(EMX_EUNSUPPORTED, "%s: layer width too large for the LDS row block (%zu bytes)", who, bytes);
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return emx::fail(EMX_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, bytes, hipGetErrorString(e));
  }
  return EMX_OK;
}

// Builds the launch description: LDS allocation (two ping-pong images sized per parity, float32 copies, side buffers), the
// GEMM records and the flattened unit list.
struct ChainBuilder {
  ChainArgs a{};
  WopArgs wa{};
  const emx_mlp_desc* desc;
  const char* who;
  int ng = 0;
  int Qp[2] = {0, 0};
  int extra[2] = {0, 0};   // bytes another tenant of the parity's region needs (a float32 copy that lives there once the image is dead)
  int pl[2] = {0, 0}, sc[2] = {0, 0};
  int lds = 0;
  ChainBuilder(const emx_mlp_desc* d, const char* w) : desc(d), who(w) {
    fill_wop_args(d, wa);
    a.lat_off = a.tgt_off = 0;
    a.y_f32 = -1;
    a.cost_at = -1;
  }
  void need(int parity, int width) { Qp[parity] = std::max(Qp[parity], groups32(width)); }
  int alloc(int bytes) {
    const int o = lds;
    lds += (bytes + 15) & ~15;
    return o;
  }
  void layout_images() {
    for (int p = 0; p < 2; ++p) {
      const int q = std::max(Qp[p], 1);
      pl[p] = alloc(std::max(q * (2048 + 64), extra[p]));
      sc[p] = pl[p] + q * 2048;
    }
  }
  static int f32_pitch(int n) { return ((n + 7) / 8) * 8 + 4; }
  // one layer in one direction; returns the GEMM's index or a negative error
  int add_gemm(int l, bool bwd, int src_par, int dst_par, const float* aux, int aux_rs, float* out, int sflags, const AopTensor& aop) {
    if (ng >= MAXG) return emx::fail(EMX_EUNSUPPORTED, "%s: more than %d GEMMs in one launch", who, MAXG);
    const int K = desc->width[l], N = desc->width[l + 1];
    const int kred = bwd ? N : K, nout = bwd ? K : N;
    const long base = wa.begin[l];
    const long opb = bwd ? base + wop_operand_bytes(K, N) : base;
    if ((opb + wop_operand_bytes(K, N)) / 1024 > 0x7fffffffL) return emx::fail(EMX_EUNSUPPORTED, "%s: operand store too large", who);
    GemmRec& G = a.g[ng];
    G.op_kb = (int)(opb / 1024);
    G.Q = groups32(kred);
    G.T = groups32(nout);
    G.ws_off = (int)((base + 2 * wop_operand_bytes(K, N)) / 4);
    G.ws_ss = bwd ? 1 : groups32(N);
    G.ws_sw = bwd ? groups32(N) : 1;
    G.n = nout;
    G.sflags = sflags | (bwd ? S_BWD : 0);
    G.aux = aux;
    G.aux_rs = aux_rs;
    G.src_pl = pl[src_par];
    G.src_sc = sc[src_par];
    G.dst_pl = dst_par >= 0 ? pl[dst_par] : -1;
    G.dst_sc = dst_par >= 0 ? sc[dst_par] : -1;
    G.dst_f32 = G.h_f32 = -1;
    G.dst_f32_pitch = G.h_f32_pitch = 0;
    G.out = out;
    G.aop = aop;
    // the source image was laid out for its own width: the lo plane sits Q K-steps behind the hi plane
    if (G.Q > 255 * 4 || G.T > 255 * 4) return emx::fail(EMX_EUNSUPPORTED, "%s: layer too wide", who);
    const int P = (nout + 127) / 128, C = (kred + 127) / 128;
    for (int p = 0; p < P; ++p)
      for (int c = 0; c < C; ++c) {
        if (a.nu >= MAXU) return emx::fail(EMX_EUNSUPPORTED, "%s: layer range needs more than %d 128x128 units", who, MAXU);
        a.u[a.nu++] = UnitRec{(unsigned char)ng, (unsigned char)p, (unsigned char)c,
                              (unsigned char)((c == 0 ? U_FIRST : 0) | (c == C - 1 ? U_LAST : 0) | (c == C - 1 && p == P - 1 ? U_LAYER_END : 0))};
      }
    return ng++;
  }
};

// The source image's lo plane offset is Q_image * 1024 where Q_image is the K-step count of the image as laid out for the
// tensor it holds: ping-pong images are shared between tensors of different widths, so every tensor uses its OWN Q for the
// plane offset (planes of a narrower tensor simply occupy the head of the allocation); producer and consumer agree because
// both derive it from the tensor's width.

int fill_cost(ChainArgs& a, const emx_auto_cost* cost, int wout, float* dY, const char* who) {
  EMX_REQUIRE(cost->target && cost->ws && cost->n > 0 && cost->ldt >= cost->n && cost->B > 0, "%s: incomplete cost description", who);
  EMX_REQUIRE(cost->variant >= EMX_MEAN_ABS && cost->variant <= EMX_MEAN_NORM, "%s: cost variant %d not available", who, cost->variant);
  EMX_REQUIRE(cost->off_sin >= 0 && cost->off_sin + cost->n <= wout &&
                  (!cost->periodic_output || (cost->off_cos >= 0 && cost->off_cos + cost->n <= wout && std::isfinite(cost->periodicity))),
              "%s: cost column window outside the output row", who);
  a.cost.target = cost->target;
  a.cost.dY = dY;
  a.cost.ws = cost->ws;
  fill_cost_scalars(a.cost, cost->B, cost->n, wout, cost->ldt, cost->off_sin, cost->off_cos, cost->periodic_output, cost->periodicity,
                    cost->variant, cost->coef);
  return EMX_OK;
}

int check_aop(const void* aop, int aop_rows, int row_begin, int row_end, const char* who) {
  if (!aop) return EMX_OK;
  EMX_REQUIRE((row_begin & 15) == 0, "%s: with a weight-gradient operand store the row range must start at a multiple of 16 (got %d)", who, row_begin);
  EMX_REQUIRE(aop_rows >= row_end, "%s: the weight-gradient operand store was laid out for %d rows, the range ends at %d", who, aop_rows, row_end);
  return EMX_OK;
}

template <int MODE>
int launch_chain(ChainBuilder& b, int row_begin, int row_end, size_t lds, hipStream_t st, const char* what) {
  const int grid = (row_end - row_begin + R - 1) / R;
  // more than one row block per CU: the two-deep ring (<= 256 registers, two workgroups per CU); else the three-deep one
  static const int force_nb = getenv("EMX_CHAIN_NB") ? atoi(getenv("EMX_CHAIN_NB")) : 0;
  const bool deep = force_nb ? force_nb >= 3 : grid <= 256;
  int rc;
  if (deep) {
    rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<MODE, 3>), lds, b.who);
    if (rc) return rc;
    hipLaunchKernelGGL((chain_kernel<MODE, 3>), dim3(grid), dim3(NTC), lds, st, b.a);
  } else {
    rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<MODE, 2>), lds, b.who);
    if (rc) return rc;
    hipLaunchKernelGGL((chain_kernel<MODE, 2>), dim3(grid), dim3(NTC), lds, st, b.a);
  }
  return emx::check_launch(what);
}

}  // namespace

extern "C" {

size_t emx_mlp_mirror_floats(const emx_mlp_desc* desc) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL) return 0;
  WopArgs wa;
  fill_wop_args(desc, wa);
  return (size_t)(wa.begin[desc->n_layers] / 4);
}

int emx_mlp_mirror(const emx_mlp_desc* desc, const float* params, float* mirror, emx_stream_t stream) {
  int rc = check_desc(desc, "emx_mlp_mirror");
  if (rc) return rc;
  EMX_REQUIRE(params && mirror, "emx_mlp_mirror: null pointer argument");
  WopArgs wa;
  fill_wop_args(desc, wa);
  hipLaunchKernelGGL(mlp_mirror_kernel, dim3((unsigned)wa.item_begin[desc->n_layers]), dim3(64), 0, (hipStream_t)stream, wa, params,
                     reinterpret_cast<char*>(mirror));
  return emx::check_launch("mlp_mirror_kernel");
}

size_t emx_mlp_aop_bytes(const emx_mlp_desc* desc, int n_rows) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL || n_rows < 1) return 0;
  AopLayout L;
  aop_layout(desc, n_rows, L);
  return (size_t)L.total;
}

int emx_mlp_row_blocks(int n_rows) { return std::max((n_rows + R - 1) / R, 1); }

int emx_mlp_forward_cost(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                         int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, void* aop, int aop_rows,
                         emx_stream_t stream) {
  const char* who = "emx_mlp_forward";
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && h && in, "%s: null pointer argument", who);
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: bad layer range", who);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_in >= desc->width[layer_begin], "%s: bad rows / stride", who);
  rc = check_aop(aop, aop_rows, row_begin, row_end, who);
  if (rc) return rc;
  ChainBuilder b(desc, who);
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  const int nf = layer_end - layer_begin;
  const int wout = desc->width[layer_end];
  for (int st = 0; st < nf; ++st) b.need(st & 1, desc->width[layer_begin + st]);
  // the cost's float32 copy of the output rows lives where the image of parity nf & 1 was: the last layer reads the other
  // parity, and every reader of this one is behind the barrier in front of the last layer
  if (cost) b.extra[nf & 1] = R * ChainBuilder::f32_pitch(wout) * 4;
  b.layout_images();
  for (int st = 0; st < nf; ++st) {
    const int l = layer_begin + st;
    const bool last = st == nf - 1;
    // the fragments of h_l are the X operand of layer l + 1's weight gradient: not needed for the network's last layer
    const AopTensor ao = (aop && l + 1 < desc->n_layers) ? aop_tensor(aop, AL, 1 + l) : AopTensor{nullptr, nullptr, 0};
    const int gi = b.add_gemm(l, false, st & 1, last ? -1 : (st + 1) & 1, params + desc->b_off[l], 0, h[l], desc->act[l] ? S_ACT : 0, ao);
    if (gi < 0) return gi;
    if (last && ao.frag) {   // (a sub-range that ends inside the network: the fragments need the split, i.e. a destination image)
      b.need((st + 1) & 1, wout);
      return emx::fail(EMX_EUNSUPPORTED, "%s: a layer range that ends inside the network cannot leave weight-gradient fragments of its last layer", who);
    }
  }
  ChainArgs& a = b.a;
  if (cost) {
    rc = fill_cost(a, cost, wout, cost->dY, who);
    if (rc) return rc;
    EMX_REQUIRE(cost->dY, "%s: incomplete cost description", who);
    GemmRec& G = a.g[b.ng - 1];
    G.dst_f32_pitch = a.y_pitch = ChainBuilder::f32_pitch(wout);
    G.dst_f32 = a.y_f32 = b.pl[nf & 1];
    a.tgt_off = b.alloc(R * cost->n * 4);
  }
  a.wop = reinterpret_cast<const char*>(mirror);
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.in_Q = groups32(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = (aop && layer_begin == 0) ? aop_tensor(aop, AL, 0) : AopTensor{nullptr, nullptr, 0};
  a.row0 = row_begin;
  a.row_end = row_end;
  return launch_chain<CHAIN_FWD>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (forward)");
}

int emx_mlp_forward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                    int ld_in, int row_begin, int row_end, float* const* h, void* aop, int aop_rows, emx_stream_t stream) {
  return emx_mlp_forward_cost(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, nullptr, aop, aop_rows, stream);
}

int emx_mlp_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* dtop,
                     int ld_top, const float* dadd, int layer_add, const emx_latent_grad* latent, float* dbelow, int row_begin,
                     int row_end, float* const* h, float* const* delta, void* aop, int aop_rows, emx_stream_t stream) {
  const char* who = "emx_mlp_backward";
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && dtop && h && delta, "%s: null pointer argument", who);
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: bad layer range", who);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_top >= desc->width[layer_end], "%s: bad rows / stride", who);
  EMX_REQUIRE((!dadd && !latent) || (layer_add >= layer_begin && layer_add < layer_end), "%s: layer_add outside the layer range", who);
  rc = check_aop(aop, aop_rows, row_begin, row_end, who);
  if (rc) return rc;
  ChainBuilder b(desc, who);
  ChainArgs& a = b.a;
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  const int n = desc->n_layers;
  const bool join = dadd || latent;
  if (latent) {
    a.lg = *latent;
    EMX_REQUIRE(a.lg.partials && a.lg.Z && a.lg.n_slots > 0, "%s: incomplete latent gradient descriptor", who);
  }
  if (join) a.join_n = desc->width[layer_add + 1];
  a.dadd = dadd;
  for (int l = layer_begin; l < layer_end; ++l) EMX_REQUIRE(!desc->act[l] || h[l], "%s: activations of layer %d are needed for its derivative", who, l);
  auto sflags = [&](int l) { return (desc->act[l] ? S_ACT : 0) | ((join && l == layer_add) ? S_JOIN : 0); };
  auto dop = [&](int l) { return aop ? aop_tensor(aop, AL, 1 + n + l) : AopTensor{nullptr, nullptr, 0}; };
  const int top = layer_end - 1;
  a.top_h = h[top];
  a.top_out = delta[top];
  a.top_flags = sflags(top);
  // boundary j = 0 is the top (width[layer_end]); GEMM of layer l: delta_l W_l^T -> dh_{l-1} -> delta_{l-1} (or dbelow)
  const int lo = dbelow ? layer_begin : layer_begin + 1;
  b.need(0, desc->width[layer_end]);
  for (int l = top; l >= lo; --l) b.need((top - l + 1) & 1, desc->width[l]);
  b.layout_images();
  for (int l = top; l >= lo; --l) {
    const int Kin = desc->width[l];
    const int st = top - l;
    const bool last = l == lo;
    int gi;
    if (l > layer_begin)
      gi = b.add_gemm(l, true, st & 1, last ? -1 : (st + 1) & 1, desc->act[l - 1] ? h[l - 1] : params, desc->act[l - 1] ? Kin : 0, delta[l - 1],
                      sflags(l - 1), dop(l - 1));
    else
      gi = b.add_gemm(l, true, st & 1, -1, params, 0, dbelow, 0, AopTensor{nullptr, nullptr, 0});
    if (gi < 0) return gi;
    // delta_{l-1} is an operand of the next GEMM and / or of the weight gradients: both need the split, i.e. an image
    GemmRec& G = a.g[gi];
    if (l > layer_begin && G.dst_pl < 0 && G.aop.frag) {
      b.a.g[gi].dst_pl = b.pl[(st + 1) & 1];
      b.a.g[gi].dst_sc = b.sc[(st + 1) & 1];
    }
  }
  if (join) a.lat_off = b.alloc(R * a.join_n * 4);
  a.wop = reinterpret_cast<const char*>(mirror);
  a.in = dtop;
  a.ld_in = ld_top;
  a.in_cols = desc->width[layer_end];
  a.in_Q = groups32(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = dop(top);
  a.row0 = row_begin;
  a.row_end = row_end;
  if (a.nu == 0) {   // a one-layer range without dbelow: only delta_top is formed; the kernel still wants one readable record
    a.g[0] = GemmRec{};
    a.g[0].Q = a.g[0].T = 1;
    a.g[0].n = 1;
    a.g[0].aux = params;
    a.g[0].src_pl = b.pl[0];
    a.g[0].src_sc = b.sc[0];
    a.g[0].dst_pl = a.g[0].dst_sc = a.g[0].dst_f32 = a.g[0].h_f32 = -1;
    a.g[0].sflags = S_BWD;
    a.u[0] = UnitRec{0, 0, 0, 0};
  }
  return launch_chain<CHAIN_BWD>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (backward)");
}

int emx_mlp_forward_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                             int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                             float* const* delta, float* dbelow, void* aop, int aop_rows, emx_stream_t stream) {
  const char* who = "emx_mlp_forward_backward";
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && h && in && cost && delta, "%s: null pointer argument", who);
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: bad layer range", who);
  EMX_REQUIRE(layer_begin <= bwd_begin && bwd_begin < layer_end, "%s: backward range [%d, %d) outside the forward range", who, bwd_begin, layer_end);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_in >= desc->width[layer_begin], "%s: bad rows / stride", who);
  const int n = desc->n_layers, top = layer_end - 1, wout = desc->width[layer_end];
  EMX_REQUIRE(!desc->act[top], "%s: the fused launch needs a linear top layer (d cost / d output is its delta)", who);
  EMX_REQUIRE(cost->dY == delta[top] || cost->dY == nullptr, "%s: d cost / d output is delta[layer_end - 1]; pass that buffer (or NULL) as cost->dY", who);
  rc = check_aop(aop, aop_rows, row_begin, row_end, who);
  if (rc) return rc;
  ChainBuilder b(desc, who);
  ChainArgs& a = b.a;
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  auto dop = [&](int l) { return aop ? aop_tensor(aop, AL, 1 + n + l) : AopTensor{nullptr, nullptr, 0}; };
  const int nf = layer_end - layer_begin;
  for (int st = 0; st < nf; ++st) b.need(st & 1, desc->width[layer_begin + st]);
  const int lo = dbelow ? bwd_begin : bwd_begin + 1;
  b.need(0, wout);
  for (int l = top; l >= lo; --l) b.need((top - l + 1) & 1, desc->width[l]);
  b.layout_images();
  int hcopy[MAXL + 1], hpitch[MAXL + 1];
  for (int l = 0; l <= MAXL; ++l) hcopy[l] = -1, hpitch[l] = 0;
  for (int st = 0; st < nf; ++st) {
    const int l = layer_begin + st;
    const bool last = st == nf - 1;
    const AopTensor ao = (aop && l + 1 < n) ? aop_tensor(aop, AL, 1 + l) : AopTensor{nullptr, nullptr, 0};
    const int gi = b.add_gemm(l, false, st & 1, last ? -1 : (st + 1) & 1, params + desc->b_off[l], 0, h[l], desc->act[l] ? S_ACT : 0, ao);
    if (gi < 0) return gi;
    EMX_REQUIRE(!(last && ao.frag), "%s: the fused launch ends at the network's last layer", who);
    // float32 copy of h_l in LDS: the cost's input (last layer) or the activation derivative of the backward units
    const bool for_bwd = desc->act[l] && l >= bwd_begin && l < top;
    if (last || for_bwd) {
      GemmRec& G = a.g[gi];
      G.dst_f32_pitch = hpitch[l] = ChainBuilder::f32_pitch(desc->width[l + 1]);
      G.dst_f32 = hcopy[l] = b.alloc(R * G.dst_f32_pitch * 4);
    }
  }
  a.cost_at = a.nu - 1;
  a.y_f32 = hcopy[top];
  a.y_pitch = hpitch[top];
  for (int l = top; l >= lo; --l) {
    const int st = top - l;
    const bool last = l == lo;
    int gi;
    if (l > bwd_begin) {
      // delta_{l-1} = (delta_l W_l^T) * act'(h_{l-1}); h_{l-1} is still in its float32 LDS copy
      gi = b.add_gemm(l, true, st & 1, (st + 1) & 1, params, 0, delta[l - 1], desc->act[l - 1] ? S_ACT : 0, dop(l - 1));
      if (gi < 0) return gi;
      a.g[gi].h_f32 = hcopy[l - 1];
      a.g[gi].h_f32_pitch = hpitch[l - 1];
      if (last && !a.g[gi].aop.frag) a.g[gi].dst_pl = a.g[gi].dst_sc = -1;
    } else {
      gi = b.add_gemm(l, true, st & 1, -1, params, 0, dbelow, 0, AopTensor{nullptr, nullptr, 0});
      if (gi < 0) return gi;
    }
  }
  rc = fill_cost(a, cost, wout, delta[top], who);
  if (rc) return rc;
  a.tgt_off = b.alloc(R * cost->n * 4);
  a.top_pl = b.pl[0];
  a.top_sc = b.sc[0];
  a.top_Q = groups32(wout);
  a.top_aop = dop(top);
  a.wop = reinterpret_cast<const char*>(mirror);
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.in_Q = groups32(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = (aop && layer_begin == 0) ? aop_tensor(aop, AL, 0) : AopTensor{nullptr, nullptr, 0};
  a.row0 = row_begin;
  a.row_end = row_end;
  return launch_chain<CHAIN_FUSED>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (forward + cost + backward)");
}

// rows per slice: EMX_WGRAD_SLICE_ROWS, doubled until at most EMX_WGRAD_MAX_SLICES slices remain -- every slice costs a
// write and a read of the whole gradient (partials), so their number must not grow with the batch
static int wgrad_slice_rows(int n_rows) {
  int r = EMX_WGRAD_SLICE_ROWS;
  while ((long)r * EMX_WGRAD_MAX_SLICES < n_rows) r *= 2;
  // small batches: shorter slices, so that a few hundred rows still spread over eight slices' worth of workgroups
  static const int min_rows = getenv("EMX_WGRAD_MIN_SLICE") ? std::max(16, atoi(getenv("EMX_WGRAD_MIN_SLICE")) / 16 * 16) : 32;
  while (r > min_rows && n_rows < 8 * r) r /= 2;
  return r;
}

int emx_mlp_wgrad_slices(int n_rows) {
  const int r = wgrad_slice_rows(n_rows);
  return std::max((n_rows + r - 1) / r, 1);
}

size_t emx_mlp_wgrad_ws_floats(int64_t n_params, int n_rows) {
  return (size_t)emx_mlp_wgrad_slices(n_rows) * (size_t)n_params;
}

int emx_mlp_wgrad(const emx_mlp_desc* desc, float* params, int64_t n_params, int layer_begin, int layer_end, const void* aop, int aop_rows,
                  int row_begin, int row_end, float* grad, float* ws, size_t ws_floats, const emx_finish_desc* finish, emx_stream_t stream) {
  const char* who = "emx_mlp_wgrad";
  WgradDev a{};
  a.has_fin = 0;
  if (finish) {
    int rcf = fill_finish_args(a.fin, finish->jobs, finish->n_jobs, finish->terms, finish->n_terms, finish->history, finish->term_history,
                               finish->capacity, finish->hstep, finish->cursor, finish->counter2, "emx_mlp_wgrad (finish)");
    if (rcf) return rcf;
    a.has_fin = 1;
  }
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && aop && ws, "%s: null pointer argument", who);
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: bad layer range", who);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end, "%s: bad rows", who);
  rc = check_aop(aop, aop_rows, row_begin, row_end, who);
  if (rc) return rc;
  const int nrows = row_end - row_begin;
  if (ws_floats < emx_mlp_wgrad_ws_floats(n_params, nrows))
    return emx::fail(EMX_EWORKSPACE, "%s: workspace %zu < %zu floats", who, ws_floats, emx_mlp_wgrad_ws_floats(n_params, nrows));
  AopLayout AL;
  aop_layout(desc, aop_rows, AL);
  const int n = desc->n_layers;
  a.l0 = layer_begin;
  a.l1 = layer_end;
  int tiles = 0;
  long lo = -1, hi = 0;
  for (int l = layer_begin; l < layer_end; ++l) {
    a.K[l] = desc->width[l];
    a.N[l] = desc->width[l + 1];
    a.w_off[l] = desc->w_off[l];
    a.b_off[l] = desc->b_off[l];
    a.X[l] = aop_tensor(const_cast<void*>(aop), AL, l);          // id 0: the network input, id l = 1 + (l - 1): h_{l-1}
    a.D[l] = aop_tensor(const_cast<void*>(aop), AL, 1 + n + l);
    a.tile_begin[l - layer_begin] = tiles;
    tiles += ((desc->width[l] + 127) / 128) * ((desc->width[l + 1] + 127) / 128);
    const long wl = desc->w_off[l], wh = desc->b_off[l] + desc->width[l + 1];
    lo = (lo < 0) ? wl : std::min(lo, wl);
    hi = std::max(hi, wh);
  }
  a.tile_begin[layer_end - layer_begin] = tiles;
  a.partial = ws;
  a.n_params = n_params;
  const int rps = wgrad_slice_rows(nrows);
  a.rb0 = row_begin / 16;
  a.rb_end = (row_end + 15) / 16;
  a.rbs_per_slice = rps / 16;
  const int slices = emx_mlp_wgrad_slices(nrows);
  hipStream_t st = (hipStream_t)stream;
  static_assert(FIN_NT == NT, "the scalar epilogue shares the weight-gradient kernel's workgroup size");
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(tiles + a.has_fin, slices), dim3(NT), 0, st, a);
  rc = emx::check_launch("mlp_wgrad_kernel");
  if (rc) return rc;
  if (!grad) return EMX_OK;   // the caller sums the slices itself (emx_adam_step_slices)
  const long nn = hi - lo;
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((unsigned)((nn + NT - 1) / NT)), dim3(NT), 0, st, ws, (long)n_params, slices, lo, hi, grad);
  return emx::check_launch("mlp_wgrad_reduce_kernel");
}

}  // extern "C"
