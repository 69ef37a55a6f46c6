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
// MI355X_MICROARCH.md).  Every float32 operand is therefore carried as a SCALED f16 PAIR (wop.h): x = (h + l) / s with s a
// power of two, h = f16(s x), l = f16(s x - h), 22 significant bits, and a product runs as the three instructions
// hh + hl + lh of v_mfma_f32_16x16x32_f16 (row-block kernels) / v_mfma_f32_32x32x16_f16 (weight gradients) with exact
// partial products in the f32 accumulator.  Scales:
//   weights      one per layer (wop.h), kept current by the optimizer;
//   activations  tanh outputs lie in [-1, 1]: the FIXED scale 2^14 -- a whole reduction accumulates inside the matrix
//                instruction and is unscaled once per output element.  Everything else (the network input, linear layers,
//                gradients): one scale per (row, 32 columns) in the LDS image; a K-step of the matrix instruction covers
//                exactly those 32 columns, its result is unscaled by one multiply-add per accumulator element.  A row's
//                numbers do not depend on which rows share its block (row ranges and batch sizes do not change results);
//   wgrad        one per (16 rows, 32 columns) block (aop.h): the row-scaled pair with its exponents shifted, exact.
//
// What bounds the kernels (profiles/r02_*): a wave issues about one instruction of ANY kind per 4-5 cycles, so a
// workgroup with one wave per SIMD is bound by its instruction count, not by the matrix pipe or the weight stream (the
// stream alone: 0.55 us per 64-KB unit and CU, tools/ubench_stream.hip).  Hence eight waves per workgroup (two per SIMD,
// one 16-column tile each) and an instruction-lean hot path: no per-element validity selects (buffer bounds and zero
// scales do that), no per-K-step unscaling behind tanh layers, addresses hoisted out of the unit loop.
//
// Lane maps (cdna_hip_programming.md section 3; checked with integer data by tests/test_mlp_gpu.py):
//   16x16x32: A[m = lane & 15][k = 8 (lane >> 4) + e], B[k][n = lane & 15], D[row = 4 (lane >> 4) + reg][col = lane & 15];
//   the k labels inside a K-step are free as long as A and B agree -- here kk(g, e) = 4 g + e (e < 4), 16 + 4 g + e - 4.
#include <atomic>
#include "act.h"
#include "adc_head.h"
#include "aop.h"
#include "cost.h"
#include "emx_common.h"
#include "finish.h"
#include "optimizer.h"
#include "sk_tile.h"
#include "wop.h"

namespace {

using namespace emx;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;            // weight-gradient kernels
constexpr int NTC = 512;           // row-block chain kernels: 8 waves, two per SIMD, one 16-column MFMA tile each
constexpr int NWC = NTC / 64;
constexpr int R = 16;              // batch rows per workgroup
constexpr int MAXL = EMX_MAX_LAYERS;
constexpr int MAXG = 2 * MAXL + 1;   // GEMMs per launch
constexpr int MAXU = 120;            // units (128 output columns x 128 k) per launch
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true);
}
// maximum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane of the row.  On bit patterns of
// |x| (unsigned order = float order there): v_max_u32 with a DPP operand, one instruction per step.
__device__ __forceinline__ unsigned row16_max(unsigned x) {
  x = max(x, dpp_u32<0xB1>(x));    // quad_perm [1, 0, 3, 2]
  x = max(x, dpp_u32<0x4E>(x));    // quad_perm [2, 3, 0, 1]
  x = max(x, dpp_u32<0x124>(x));   // row_ror:4
  x = max(x, dpp_u32<0x128>(x));   // row_ror:8
  return x;
}

// ---------------------------------------------------------------------------------------------------------
// Row-block chain kernel (forward, backward-data and the fused forward + cost + backward launch share it).
//
// LDS images.  The activations (forward) or pre-activation gradients (backward) of the row block, one image per layer
// boundary: two f16 planes stored TRANSPOSED, [k = column][16 rows] (32 bytes per column; the lo plane behind the hi plane),
// and -- unless the image holds tanh outputs -- the per-(32-column group, row) reciprocal scales as floats.  A K-step's A
// fragment is two ds_read_b64_tr_b16 per plane at  plane + 1024 ks + 8 lane  (+ 512): lane (m, g) receives row m, columns
// 32 ks + 4 g + q and 32 ks + 16 + 4 g + q -- the k order kk(g, e) of wop.h, conflict-free (a 32-lane half reads 256
// contiguous bytes); the epilogue's D tile (lane = column, registers = 4 consecutive rows) is one 8-byte LDS write per plane.
//
// Weight stream.  Every workgroup needs every weight (64 KB per 128 x 128 layer): the host flattens the layer range into
// UNITS (layer, pass of 128 columns, chunk of 128 k; one unit = 12 matrix instructions per wave), everything a unit needs
// from memory -- 8 operand pieces, the layer's scale, the epilogue operands -- travels in one package NB - 1 units ahead
// in a register ring, and every iteration issues the same vector-memory sequence (no conditional load, no exec-masked
// store: results leave through buffer resources whose bounds check drops what does not exist), so that every s_waitcnt
// the compiler inserts is for the OLDEST package only (vmcnt is one in-order counter for loads and stores).
// ---------------------------------------------------------------------------------------------------------
// S_ACT: the layer has an activation (its derivative in backward units); S_ACTX: ... and it is not tanh -- its kind sits in bits
// 8 .. 11; S_BND: a forward layer whose outputs lie in [-1, 1] (tanh, sigmoid, softsign): its image is written at the fixed scale 2^14
enum { S_ACT = 1, S_JOIN = 2, S_BWD = 4, S_PUB = 8, S_WAIT = 16, S_ACTX = 32, S_BND = 64 };
constexpr int S_KIND_SHIFT = 8;
inline int act_flags(int kind) {
  if (kind == ACT_NONE) return 0;
  int f = S_ACT | (kind << S_KIND_SHIFT);
  if (kind != ACT_TANH) f |= S_ACTX;
  if (kind == ACT_TANH || kind == ACT_SIGMOID || kind == ACT_SOFTSIGN) f |= S_BND;
  return f;
}
// S_WAIT (PUB builds): the tile workgroups' partial sums are awaited and joined in front of this epilogue   // S_PUB: the layer's rows are published to the tile workgroups of the launch (PUB builds)

struct GemmRec {          // one layer of the range in one direction: P x C regular units (128 columns x 128 k)
  int P, C;               // passes of 128 output columns, chunks of 128 reduction indices
  int n;                  // output columns that exist
  int sflags;             // S_*
  int aux_rs;             // row stride of aux (0: one broadcast row)
  float src_rs;           // != 0: the source image holds tanh outputs at this fixed reciprocal scale (no scale table)
  const float* ws;        // the layer's reciprocal weight scale (its header in the store)
  const float* aux;       // epilogue operand: the bias (forward), rows of h for the activation derivative (backward)
  int src_pl, src_lo, src_sc;   // LDS byte offsets of the source image: hi plane, distance of the lo plane, scales
  int dst_pl, dst_lo, dst_sc;   // destination image (dst_pl -1: none)
  int dst_f32, dst_f32_pitch;   // float32 copy of the output rows in LDS, [16][pitch] (-1: none): cost input, tanh' of the fused backward
  int h_f32, h_f32_pitch;       // fused backward units: float32 copy of h for the activation derivative (-1: through aux)
  float* out;             // [*, n] float32 rows in HBM (null: not kept)
  AopTensor aop;          // fragments of the output for the weight gradients (frag null: not kept)
};

enum { CHAIN_FWD = 0, CHAIN_BWD = 1, CHAIN_FUSED = 2 };

struct ChainArgs {
  GemmRec g[MAXG];
  int ng;                 // GEMMs of the launch
  int ng_fwd;             // fused mode: the first ng_fwd are the forward layers, the cost runs behind them
  const char* units;      // first unit of the launch in the weight operand store (wop.h: units lie in execution order)
  const char* units_bwd;  // fused mode: first unit of the backward part
  const float* in;        // forward: input rows; backward: d loss / d h_top
  int ld_in, in_cols, in_Q, in_Qp;   // in_Q: column groups of 32 that exist, in_Qp: of the image (padded to whole chunks)
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
  // fused mode: d cost / d output is re-staged into the image (top_pl, top_sc), the source of the first backward GEMM, and
  // into top_aop
  int top_pl, top_sc, top_Q, top_Qp;
  AopTensor top_aop;
};

#ifdef EMX_CHAIN_TIMING   // in-kernel timeline (tools/abl_timing.py); stamps go to LDS so that they do not touch vmcnt
__device__ long long emx_dbg[8192];
#define TSTAMP(slot) do { if (lane == 0 && (slot) < 64) dbg_lds[wave * 64 + (slot)] = clock64(); } while (0)
#else
#define TSTAMP(slot)
#endif

template <int V>
struct IC_ {
  static constexpr int value = V;
};

// where the chain workgroups of a launch that also runs all-pairs tiles (chain_tiles_kernel) publish the latent rows
struct PubArgs {
  float* zpub;        // [B][n] rows of the published layer
  unsigned* ready;    // counter of workgroups that have published
};

// The kernel's body as a function, so that chain_kernel and chain_tiles_kernel (below) share it.  PUB_OFF > 0: the forward GEMM
// flagged S_PUB hands its rows (two columns) to the tile workgroups of the same launch (sk_tile.h); the offset is that of a
// PubArgs record in the kernel's arguments.
template <int MODE, int NB, int PUB_OFF, int KS = 1, bool SMALL = false>
__device__ __forceinline__ void chain_body(const float* p_in, const float* p_top_h, const float* p_target, int p_ld_in, int p_in_cols,
                                           int p_row0, int p_row_end, int p_ldt, int p_cost_n, int p_in_Q, int p_with_cost,
                                           const ChainArgs& a) {
  // KS: waves per 16-column tile.  1: eight waves, a wave runs all four K-steps of a unit.  2: SIXTEEN waves (four per SIMD) --
  // wave (tile tw, kh) runs K-steps 2 kh, 2 kh + 1 and, after an exchange of accumulators through LDS, the epilogue of rows
  // 4 g + 2 kh, 4 g + 2 kh + 1 of the tile: half the matrix instructions, LDS fragments, piece loads and epilogue elements per
  // wave, twice the waves to hide the unit's dependent trips behind.  Built for SMALL launches only.
  // SMALL: every GEMM of the launch is ONE unit (all widths of the range <= 128): no pass / chunk bookkeeping, every unit ends
  // a layer.
  static_assert(KS == 1 || (KS == 2 && SMALL && PUB_OFF == 0), "sixteen-wave row blocks: SMALL launches of the plain kernel only");
  constexpr bool BWD = MODE == CHAIN_BWD;
  constexpr int NW_ = NWC * KS, NT_ = NTC * KS;   // waves / threads of the workgroup
  constexpr int NS = 4 / KS;                      // K-steps of a unit per wave
  constexpr int NR = 4 / KS;                      // rows of its tile column per lane in the epilogue
  constexpr int NE = BWD ? NR : 1;   // epilogue operands per lane: h of its rows (backward) or the bias of its column
  extern __shared__ __align__(16) char lds[];
  __shared__ float cred[NW_];
  __shared__ __align__(16) unsigned mxbuf[2][NWC][R];   // row maxima of the waves' tiles (dynamic-scale epilogues), two copies by unit parity
  __shared__ __align__(16) float xch[KS == 2 ? 2 * NW_ * 64 * 2 : 4];   // KS = 2: the accumulator halves the waves of a tile hand to each other
#ifdef EMX_CHAIN_TIMING
  __shared__ long long dbg_lds[NT_];
  int tcount = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = KS == 2 ? (wave & 7) : wave;   // the wave's 16-column tile
  const int kh = KS == 2 ? (wave >> 3) : 0;     // its half of the reduction and of the rows
  const int n = lane & 15, g = lane >> 4;
  TSTAMP(0);
  const int r0 = p_row0 + blockIdx.x * R;
  const int rb = r0 >> 4;   // row block of the weight-gradient fragments (row0 is a multiple of 16 whenever they are kept)
  const int rows_valid = min(R, p_row_end - r0);
  // the GEMM table is indexed at run time: read it through the kernel-argument segment itself (scalar loads from constant
  // memory; indexing the by-value struct makes hipcc copy all of it to scratch in the larger instantiations)
  typedef const ChainArgs __attribute__((address_space(4))) * KArgs;
  static_assert(sizeof(ChainArgs) % 8 == 0, "ChainArgs follows 56 bytes of scalars");
  const KArgs ka = (KArgs)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + 56);
  float* lat = reinterpret_cast<float*>(lds + a.lat_off);
  float* tgt = reinterpret_cast<float*>(lds + a.tgt_off);

  // (the rows joined into dh of one layer are collected in stage_rest, behind the requests for the row block and for the first
  //  packages: everything this launch waits for first is then in flight together)
  // ---- staging a row block [16][ncols] of float32 values into an image: wave w takes the column groups w, w + 8, ...;
  //      lane (c = lane & 31, hf = lane >> 5) the column 32 group + c, rows 8 hf .. 8 hf + 7 -- its 8 halves are 16
  //      contiguous bytes of the transposed plane AND, with lane as the fragment's lane, the weight-gradient fragment (aop.h).
  //      Rows beyond the batch and columns beyond the tensor enter as zeros; groups beyond the tensor (the image is padded
  //      to whole 128-column chunks) are zero-filled.
  const int sc_c = lane & 31, sc_hf = lane >> 5;
  auto stage_finish = [&](const float (&v)[8], int grp, int lo_off, int pl, int sc, const AopTensor& ao, bool keep_frag) __attribute__((always_inline)) {
    // per-row scale over the group's 32 columns (lanes of one half)
    float s[8], rs[8];
    unsigned bm = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned m = row16_max(abs_bits(v[e]));
      m = max_xor16(m);
      pow2_scale_bits<ACT_TOP>(m, s[e], rs[e]);
      bm = max(bm, m);
    }
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      unsigned ph, pl;
      split_f16_pair(v[e] * s[e], v[e + 1] * s[e + 1], ph, pl);   // (three instructions a pair: emx_common.h)
      const half2v vh = __builtin_bit_cast(half2v, ph), vl = __builtin_bit_cast(half2v, pl);
      hi[e] = vh[0], hi[e + 1] = vh[1], lo[e] = vl[0], lo[e + 1] = vl[1];
    }
    const int k = 32 * grp + sc_c;
    *reinterpret_cast<half8*>(lds + pl + k * 32 + sc_hf * 16) = hi;
    *reinterpret_cast<half8*>(lds + pl + lo_off + k * 32 + sc_hf * 16) = lo;
    {   // (every lane of a half writes the same 32 bytes: no branch)
      float* dst = reinterpret_cast<float*>(lds + sc) + grp * 16 + 8 * sc_hf;
      *reinterpret_cast<f32x4*>(dst) = f32x4{rs[0], rs[1], rs[2], rs[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{rs[4], rs[5], rs[6], rs[7]};
    }
    if (keep_frag && ao.frag != nullptr) {
      // block scale (16 rows x 32 columns) for the weight gradients: the row-scaled pair with its exponents shifted
      bm = max_xor32(bm);
      float sb, rsb;
      pow2_scale_bits<ACT_TOP>(bm, sb, rsb);
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
  // the row block from HBM (forward: input rows; backward: delta_top = (dtop [+ join]) * act'(h_top)).  Addresses are
  // clamped and the values masked by a multiplication: a select on a loaded value makes hipcc emit an exec-masked load
  // followed by s_waitcnt vmcnt(0) per element, eight dependent round trips at the head of the kernel.
  const int sK = p_in_cols;
  auto stage_load = [&](int grp, float (&x)[8], float (&hv)[8]) __attribute__((always_inline)) {
    const int cc = min(32 * grp + sc_c, sK - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long grow = (long)min(r0 + 8 * sc_hf + e, p_row_end - 1);
      x[e] = p_in[grow * p_ld_in + cc];
      if (BWD) hv[e] = p_top_h[grow * sK + cc];
    }
  };
  auto stage_value = [&](int grp, const float (&x)[8], const float (&hv)[8], float (&v)[8]) __attribute__((always_inline)) {
    const int col = 32 * grp + sc_c, cc = min(col, sK - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 8 * sc_hf + e;
      float y = x[e];
      if (BWD) {
        if (a.top_flags & S_JOIN) y += lat[row * sK + cc];
        if (a.top_flags & S_ACT) y *= (a.top_flags & S_ACTX) ? act_deriv_from_output((a.top_flags >> S_KIND_SHIFT) & 15, hv[e]) : fmaf(-hv[e], hv[e], 1.f);
      }
      v[e] = y * ((col < sK && row < rows_valid) ? 1.f : 0.f);
      if (BWD && a.top_out != nullptr && col < sK && row < rows_valid) a.top_out[(long)(r0 + row) * sK + col] = v[e];
    }
  };

  // ---- the register ring
  half8 wp[NB][NS][2];   // [slot][K-step of this wave][plane]
  float ep[NB][NE];
  float wsc[NB];

  // per-GEMM state of a unit stream, wave-uniform (scalar registers); the request stream runs NB - 1 units ahead of the
  // execution stream.  Units are regular (wop.h): the next unit's package is 64 KB behind this one's.  SMALL launches (every
  // layer of the range at most 128 wide in both dimensions): a GEMM is ONE unit, the stream is the GEMM index.
  struct Stream {
    int gi, g_end;          // current GEMM, end of the segment
    int p, c;               // pass, chunk
    int P, C;
  };
  auto stream_next = [&](Stream& st) __attribute__((always_inline)) -> bool {   // returns true when the GEMM changed
    if constexpr (!SMALL) {
      if (++st.c < st.C) return false;
      st.c = 0;
      if (++st.p < st.P) return false;
      st.p = 0;
    }
    ++st.gi;
    return true;
  };

  // request stream
  Stream rq;
  const char* rq_ptr;       // this wave's pieces of the unit to request next
  const float* rq_aux;
  const float* rq_ws;
  int rq_aux_rs, rq_n;
  bool rq_live;
  auto rq_load_gemm = [&]() __attribute__((always_inline)) {
    const auto& G = ka->g[rq.gi];
    if constexpr (!SMALL) {
      rq.P = G.P;
      rq.C = G.C;
    }
    rq_aux = G.aux;
    rq_aux_rs = G.aux_rs;
    rq_n = G.n;
    rq_ws = G.ws;
  };
  auto request = [&](auto RC, const int ln) __attribute__((always_inline)) {   // the package of the request stream's unit into ring slot r, then advance
    constexpr int r = decltype(RC)::value;
    const int n = ln & 15, g = ln >> 4;
    const int ccol = min((SMALL ? 0 : 128 * rq.p) + 16 * tw + n, rq_n - 1);
    if (BWD) {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) ep[r][rr] = rq_aux[(long)min(r0 + 4 * g + kh * NR + rr, a.row_end - 1) * rq_aux_rs + ccol];
    } else {
      ep[r][0] = rq_aux[ccol];
    }
    wsc[r] = *rq_ws;
  };
  auto request_pieces = [&](auto RC, int s, const int ln) __attribute__((always_inline)) {
    constexpr int r = decltype(RC)::value;
    const char* p = rq_ptr + s * 2048 + ln * 16;
    wp[r][s][0] = *reinterpret_cast<const half8*>(p);
    wp[r][s][1] = *reinterpret_cast<const half8*>(p + 1024);
  };
  // Advancing the request stream, in two parts: the scalar loads of the next unit's GEMM record go out EARLY (rq_prefetch, behind the
  // first K-step of the unit in execution: scalar loads and LDS accesses share one counter, so a record requested in front of the
  // A fragments makes the first matrix instruction wait for it, and one requested behind the matrix instructions makes the
  // epilogue's first LDS access wait) -- the stream's state changes LATE (rq_commit, behind the unit's last piece request, which
  // still reads the old pointer).
  Stream rq_nx;
  const float* nq_aux;
  const float* nq_ws;
  int nq_aux_rs, nq_n, nq_P, nq_C;
  int rq_step;   // 0: the stream has ended; 1: next unit of the same GEMM; 2: next GEMM
  auto rq_prefetch = [&]() __attribute__((always_inline)) {
    rq_step = 0;
    if (!rq_live) return;
    rq_nx = rq;
    const bool changed = stream_next(rq_nx);
    if (changed && rq_nx.gi >= rq_nx.g_end) return;   // the segment's last unit: keep requesting it (the same instruction sequence; nothing depends on it)
    rq_step = changed ? 2 : 1;
    if (changed) {
      const auto& G = ka->g[rq_nx.gi];
      if constexpr (!SMALL) {
        nq_P = G.P;
        nq_C = G.C;
      }
      nq_aux = G.aux;
      nq_aux_rs = G.aux_rs;
      nq_n = G.n;
      nq_ws = G.ws;
    }
  };
  auto rq_commit = [&]() __attribute__((always_inline)) {
    if (rq_step == 0) {
      rq_live = false;
      return;
    }
    rq = rq_nx;
#ifndef EMX_EXP_NOSTREAM   // (experiment: every unit re-reads the first package -- times the kernel without its weight stream)
    rq_ptr += WOP_UNIT;
#endif
    if (rq_step == 2) {
      if constexpr (!SMALL) {
        rq.P = nq_P;
        rq.C = nq_C;
      }
      rq_aux = nq_aux;
      rq_aux_rs = nq_aux_rs;
      rq_n = nq_n;
      rq_ws = nq_ws;
    }
  };
  auto rq_advance_plain = [&]() __attribute__((always_inline)) {   // both parts at once, no temporaries
    if (!rq_live) return;
    Stream nx = rq;
    const bool changed = stream_next(nx);
    if (changed && nx.gi >= nx.g_end) {
      rq_live = false;
      return;
    }
    rq = nx;
#ifndef EMX_EXP_NOSTREAM
    rq_ptr += WOP_UNIT;
#endif
    if (changed) rq_load_gemm();
  };
  auto rq_advance = [&]() __attribute__((always_inline)) { rq_advance_plain(); };

  // execution stream: the current GEMM's record in scalar registers
  Stream ex;
  struct GX {
    int n, sflags;
    float src_rs;
    int src_pl, src_lo, src_sc, dst_pl, dst_lo, dst_sc, dst_f32, dst_f32_pitch, h_f32, h_f32_pitch;
    float* out;
    char* frag;
    float* rscale;
    int CG;
  } X;
  auto load_gx = [&](int gi, int g_end, GX& x, int& P, int& C) __attribute__((always_inline)) {
    const auto& G = ka->g[min(gi, g_end - 1)];
    if constexpr (!SMALL) {
      P = G.P;
      C = G.C;
    }
    x.n = G.n;
    x.sflags = G.sflags;
    x.src_rs = G.src_rs;
    x.src_pl = G.src_pl;
    x.src_lo = G.src_lo;
    x.src_sc = G.src_sc;
    x.dst_pl = G.dst_pl;
    x.dst_lo = G.dst_lo;
    x.dst_sc = G.dst_sc;
    x.dst_f32 = G.dst_f32;
    x.dst_f32_pitch = G.dst_f32_pitch;
    x.h_f32 = G.h_f32;
    x.h_f32_pitch = G.h_f32_pitch;
    x.out = G.out;
    x.frag = G.aop.frag;
    x.rscale = G.aop.rscale;
    x.CG = G.aop.CG;
  };
  auto ex_load_gemm = [&]() __attribute__((always_inline)) { load_gx(ex.gi, ex.g_end, X, ex.P, ex.C); };

  // auto cost on the last forward layer's rows (float32 copy in LDS): 32 threads per row; d cost / d output goes to HBM
  // and, in place, into the copy
  auto cost_body = [&](auto PO_) __attribute__((always_inline)) {
    constexpr int PO = decltype(PO_)::value;   // periodic output or not: a wave-uniform branch around the whole body (cost.h)
    const CostArgs& cst = a.cost;
    // (written for 512 threads = 16 rows x 32: the upper waves of a sixteen-wave build run along with nothing to do -- they read row
    //  15, store nothing, add zeros to the sum)
    const bool act = KS == 1 || tid < NTC;
    const int row = KS == 1 ? tid >> 5 : min(tid >> 5, R - 1), l32 = tid & 31;
    const bool rok = act && row < rows_valid;
    const long grow = (long)min(r0 + row, a.row_end - 1);
    float* yrow = reinterpret_cast<float*>(lds + a.y_f32) + row * a.y_pitch;
    float* const ywr = act ? yrow : nullptr;   // (cost_store drops a null row)
    const float* trow = tgt + row * cst.n;
    float* drow = (rok && cst.dY != nullptr) ? cst.dY + grow * cst.ldy : nullptr;
    float lsum = 0.f;
    if (cst.variant == EMX_MEAN_NORM) {
      float ss = 0.f;
      for (int k = l32; k < cst.n; k += 32) {
        float sn, cs, delta, dd;
        cost_elem<PO>(cst, yrow, trow, k, sn, cs, delta, dd);
        ss = fmaf(delta, delta, ss);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);   // stays inside the row's 32 lanes
      const float nrm = sqrtf(ss);
      const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
      if (l32 == 0 && rok) lsum = nrm;
      for (int k = l32; k < cst.n; k += 32) {
        float sn, cs, delta, dd;
        cost_elem<PO>(cst, yrow, trow, k, sn, cs, delta, dd);
        const float gq = rok ? cst.gnorm * delta * inv * dd : 0.f;
        cost_store<PO>(cst, drow, nullptr, k, gq, sn, cs);
        cost_store<PO>(cst, ywr, nullptr, k, gq, sn, cs);
      }
    } else {
      for (int k = l32; k < cst.n; k += 32) {
        float sn, cs, delta, dd;
        cost_elem<PO>(cst, yrow, trow, k, sn, cs, delta, dd);
        float gq;
        if (cst.variant == EMX_MEAN_ABS) {
          if (rok) lsum += fabsf(delta);
          gq = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
        } else {
          if (rok) lsum = fmaf(delta, delta, lsum);
          gq = 2.f * delta * dd;
        }
        gq = rok ? cst.gnorm * gq : 0.f;
        cost_store<PO>(cst, drow, nullptr, k, gq, sn, cs);
        cost_store<PO>(cst, ywr, nullptr, k, gq, sn, cs);
      }
    }
    const float tsum = block_sum<NT_>(lsum, cred);
    if (tid == 0) cst.ws[blockIdx.x] = tsum * cst.gnorm;
  };
  // The same for the common shape -- a non-periodic output with mean_abs or mean_square -- in the STAGING layout: wave w takes the
  // column groups w, w + 8, ..; lane (c, hf) column 32 grp + c, rows 8 hf .. 8 hf + 7.  In the fused launch the lane then holds
  // exactly what stage_finish needs: d cost / d output goes straight from registers into the first backward GEMM's image and the
  // weight-gradient fragments (RESTAGE) -- no write-back of the gradient into the output copy, no barrier, no second pass over
  // LDS (the cost phase and the re-staging were 2 500 + 3 000 cycles of the fused launch, profiles/r04_chain_timeline.txt).
  // Another assignment of elements to threads than cost_body's, hence another rounding of the workgroup's cost share (the forward
  // launch with the cost and the fused launch share this code and agree bit for bit).
  auto cost_cols = [&](auto RESTAGE_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
    constexpr bool RESTAGE = decltype(RESTAGE_)::value != 0;
    const CostArgs& cst = a.cost;
    const float* const yb = reinterpret_cast<const float*>(lds + a.y_f32);
    const int wout = cst.ldy;
    const int ngrp = RESTAGE ? a.top_Qp : groups32(wout);
    float lsum = 0.f;
    // (the first group outside the loop: in front of a LOOP that holds stores the compiler drains the vector-memory counter --
    //  s_waitcnt vmcnt(0) -- and the first backward package, requested a moment ago from lines the optimizer rewrote on other
    //  XCDs, is a trip of ~2.5 us: the cost phase then could not start before it had arrived.  Rows of up to 256 columns are one
    //  group per wave, no loop.)
    auto group = [&](const int grp) __attribute__((always_inline)) {
      const int col = 32 * grp + sc_c;
      const bool covered = col >= cst.off_sin && col < cst.off_sin + cst.n;
      const int cy = min(col, wout - 1), ct = min(max(col - cst.off_sin, 0), cst.n - 1);
      float yv[8], tv8[8], g8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        yv[e] = yb[(8 * sc_hf + e) * a.y_pitch + cy];
        tv8[e] = tgt[(8 * sc_hf + e) * cst.n + ct];
      }
      // (variant and wrap are wave-uniform: ONE test each in front of straight-line code.  Without a wrap |diff| sgn(diff) IS diff:
      //  d cost / d y is 2 gnorm diff resp. gnorm sgn(diff), the same bits as the general form below in 10 instead of 45 instructions
      //  per element -- four waves run this phase alone on their SIMDs, every instruction at its full latency)
      auto elems = [&](auto WRAP_, auto ABS_) __attribute__((always_inline)) {
        constexpr bool WRAP = decltype(WRAP_)::value != 0, ABS = decltype(ABS_)::value != 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = 8 * sc_hf + e;
          const bool live = covered && row < rows_valid;
          const float diff = yv[e] - tv8[e];
          float gq;
          if constexpr (!WRAP) {
            if constexpr (ABS) {
              if (live) lsum += fabsf(diff);
              gq = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
            } else {
              if (live) lsum = fmaf(diff, diff, lsum);
              gq = 2.f * diff;
            }
          } else {
            const float d = fabsf(diff);
            const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
            float delta = d, dd = sgn;
            const float w = cst.period - d;
            if (!(d <= w)) delta = w, dd = -sgn;
            if constexpr (ABS) {
              if (live) lsum += fabsf(delta);
              gq = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
            } else {
              if (live) lsum = fmaf(delta, delta, lsum);
              gq = 2.f * delta * dd;
            }
          }
          g8[e] = live ? cst.gnorm * gq : 0.f;
          if (live && cst.dY != nullptr) cst.dY[(long)(r0 + row) * cst.ldy + col] = g8[e];
        }
      };
      if (cst.wrap) {
        if (cst.variant == EMX_MEAN_ABS) elems(IC_<1>{}, IC_<1>{});
        else elems(IC_<1>{}, IC_<0>{});
      } else {
        if (cst.variant == EMX_MEAN_ABS) elems(IC_<0>{}, IC_<1>{});
        else elems(IC_<0>{}, IC_<0>{});
      }
      if constexpr (RESTAGE) {
        TSTAMP(51);
        stage_finish(g8, grp, a.top_Qp * 1024, a.top_pl, a.top_sc, a.top_aop, grp < a.top_Q);
        TSTAMP(52);
      }
    };
    if (wave < ngrp) group(wave);
    if (ngrp > NW_)   // (an explicit guard: the drain sits in the loop's preheader, in front of its own trip-count test)
      for (int grp = wave + NW_; grp < ngrp; grp += NW_) group(grp);
    const float tsum = block_sum<NT_>(lsum, cred);   // (its barriers also close the re-staging: the image is complete behind it)
    if constexpr (RESTAGE) TSTAMP(53);
    if (tid == 0) cst.ws[blockIdx.x] = tsum * cst.gnorm;
  };
  auto cost_phase = [&]() __attribute__((always_inline)) {
    if (a.cost.periodic_output) cost_body(IC_<1>{});
    else if (a.cost.vec4) cost_cols(IC_<0>{});
    else cost_body(IC_<0>{});
  };

  // The tile's accumulator in units of 1 / (weight scale [x source scale]), in TWO halves of the reduction: K-steps 0, 1 and
  // K-steps 2, 3 of every chunk accumulate separately and are added once, in front of the epilogue.  With sixteen waves (KS = 2)
  // the halves live in the two waves that share a tile -- wave (tile, kh) runs K-steps 2 kh, 2 kh + 1 -- and the waves exchange
  // them through LDS; the eight-wave builds keep both halves and add them in the same order, so every build of this kernel forms
  // the same bits.
  f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
  int uparity = 0;   // units executed so far, mod 2: which copy of the exchange areas this unit uses

  const float* const wsf = reinterpret_cast<const float*>(a.units);

  // Executes the execution stream's unit from ring slot rx while the request stream's unit is requested into slot r.  Every
  // call issues the same vector-memory sequence: NE + 1 epilogue operands, 2 NS pieces (two per K-step of matrix work), the
  // result stores through buffer resources whose bounds check drops what does not exist -- no conditional load, no
  // exec-masked store, so that every wait the compiler inserts is for the oldest package only.
  // PUB builds, the encoder's backward units in the SAME launch (emx_mlp_forward_backward_tiles_join): in front of the epilogue that
  // forms dh of the latent layer the workgroup waits -- one lane polls -- until the LAST tile workgroup has reported (sync[3]; the
  // tile workgroups stored their partial sums with sc1 and drained them), then collects its sixteen rows of the latent gradient
  // as the backward launch's staging does (one wave per element, lanes over the tiles' slots, the same association: the same
  // bits), with sc1 loads.  The last chain workgroup through clears sync[3] / sync[4] for the next launch.
  auto wait_join = [&]() __attribute__((always_inline)) {
    if constexpr (PUB_OFF > 0) {
      int off = PUB_OFF;
      asm volatile("" : "+s"(off));
      typedef const PubArgs __attribute__((address_space(4))) * KPub;
      const KPub kp = (KPub)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + off);
      unsigned* const sync = kp->ready;
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          if (++spins > emx::sk::TILE_ROLE_SPIN_LIMIT) {
            __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();
      // (the join's description is fetched from the kernel arguments HERE, through an offset the optimiser cannot see through: held
      //  in scalar registers over the whole unit loop it costs the loop its registers -- as the two pointers of the publish hook)
      int o2 = 56;
      asm volatile("" : "+s"(o2));
      const KArgs kq = (KArgs)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + o2);
      const int N = kq->join_n, n_slots = kq->lg.n_slots, row_base = kq->lg.row_base, row_end = kq->row_end;
      const long slot_stride = kq->lg.slot_stride;
      const float gscale = kq->lg.gscale, center_grad = kq->lg.center_grad;
      const float* const Zp = kq->lg.Z;
      const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kq->lg.partials), 0, 0x7fffffff, 0x00020000);
      for (int el = wave; el < R * N; el += NWC) {
        const int r = el / N, c = el - r * N;
        const bool rok = r0 + r < row_end;
        float v = 0.f, zz = 0.f;
        if (rok) {
          const int s0 = min(lane, n_slots - 1);
          const long e0 = (long)(r0 + r - row_base) * N + c;
          v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rP, (int)((e0 + (long)s0 * slot_stride) * 4), 0, 16)) * (lane < n_slots ? 1.f : 0.f);
          zz = emx::sk::load_sc1(Zp + (long)(r0 + r) * N + c);     // (this workgroup's own rows of the latent layer, stored earlier in this launch: past L1)
        }
        const float sum = wave_sum(v);
        if (lane == 0) lat[el] = fmaf(center_grad, zz, gscale * sum);
      }
      __syncthreads();
      if (tid == 0) {
        const unsigned nc = (unsigned)((p_row_end - p_row0 + R - 1) / R);
        const unsigned c = __hip_atomic_fetch_add(&sync[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == nc - 1u && __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {   // (after a give-up nothing is reset)
          __hip_atomic_store(&sync[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sync[4], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };
  auto exec = [&](auto RC) __attribute__((always_inline)) {
    constexpr int r = decltype(RC)::value;          // slot being requested into
    constexpr int rx = (r + 1) % NB;                // slot being executed
    // The lane-dependent address parts.  Where the kernel must fit 128 registers (PUB_OFF != 0: chain_tiles_kernel shares its CUs
    // with the tile workgroups, chain_kernel_c128 with a second chain workgroup; KS = 2: sixteen waves are four per SIMD) they are
    // re-derived from the work-item id in every unit instead of living in registers across the unit loop: there the allocator
    // spilled them to scratch, and a scratch reload sits in the same in-order memory counter as the weight packages -- every unit
    // then waited for its NEXT package.
    int ln = lane;
    if constexpr (PUB_OFF != 0 || KS == 2) {
      ln = (int)(threadIdx.x & 63u);
      asm volatile("" : "+v"(ln));
    }
    const int lane = ln, n = ln & 15, g = ln >> 4;
    const int lane8 = ln * 8;
    // this lane's bytes inside a fragment: lane (n, g) of tile t = tw & 1 holds rows 4 g .. 4 g + 3 (KS = 2: two of them) of
    // column 16 t + n = fragment lane 32 (g >> 1) + 16 t + n, elements 4 (g & 1) + ..
    const unsigned fvoff = (unsigned)((32 * (g >> 1) + 16 * (tw & 1) + n) * 16 + (g & 1) * 8 + kh * (2 * NR));
    const bool dyn = X.src_rs == 0.f;
#ifdef EMX_CHAIN_TIMING
    const int ui = tcount++;
#endif
    TSTAMP(3 + 5 * ui);
    request(RC, ln);
    // the next unit of the execution stream; when it belongs to another GEMM its record is fetched now (scalar loads from
    // the kernel arguments), a whole unit ahead of its first use
    Stream exn = ex;
    const bool ex_changed = stream_next(exn);
    GX Xn = X;
    // (the scalar loads of the unit: the next unit's record of the execution stream and of the request stream -- issued behind the
    //  first K-step, see rq_prefetch)
    auto records = [&](const bool behind_fragments) __attribute__((always_inline)) {
      // (every A fragment of the unit has arrived by now -- said explicitly: with scalar loads outstanding every later wait for
      //  an LDS read would have to be a wait for ALL of them, lgkmcnt(0), in the middle of the matrix instructions)
      if (behind_fragments) __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); vmcnt, expcnt untouched
      if (ex_changed) load_gx(exn.gi, exn.g_end, Xn, exn.P, exn.C);
      rq_prefetch();
    };

    const int exc = SMALL ? 0 : ex.c, exp_ = SMALL ? 0 : ex.p;
    if (exc == 0) accA = accB = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments of the wave's K-steps up front (one LDS latency per unit) -- or, where the backward kernel is held to 128
    // registers (chain_kernel_c128: a second workgroup on the CU runs in this one's gaps), two K-steps at a time
    constexpr int AF = (BWD && PUB_OFF != 0) ? 2 : NS;
    // The early records (behind the first K-step) are the SMALL builds': in the general and the 128-register builds the extra live
    // scalars of the split request stream cost more (17 -> 86 spilled scalar registers in chain_kernel_c128<FUSED>, +6 us at C5's
    // shape) than the shorter wait saves -- there the records go out in front of the fragments, as they always did, and the request
    // stream advances behind the matrix instructions.
    constexpr bool EARLY_REC = SMALL && AF == NS;
    if constexpr (!EARLY_REC) {
      if (ex_changed) load_gx(exn.gi, exn.g_end, Xn, exn.P, exn.C);
    }
    half8 ah[AF], al[AF];
    f32x4 rsa[AF];
    const char* apl = lds + X.src_pl + exc * 4096 + kh * (NS * 1024) + lane8;
    const char* asc = lds + X.src_sc + exc * 256 + kh * (NS * 64) + g * 16;
    auto fragments = [&](const int s0) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < AF; ++s) {
        const half4 h0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + (s0 + s) * 1024));
        const half4 h1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + (s0 + s) * 1024 + 512));
        const half4 l0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + X.src_lo + (s0 + s) * 1024));
        const half4 l1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + X.src_lo + (s0 + s) * 1024 + 512));
        ah[s] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        al[s] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      }
      if (dyn) {
#pragma unroll
        for (int s = 0; s < AF; ++s) rsa[s] = *reinterpret_cast<const f32x4*>(asc + (s0 + s) * 64);
      }
    };
    fragments(0);
    // (Round 4 tried the unit's twelve matrix instructions interleaved over independent accumulators -- the two halves of the
    //  reduction; the four K-steps' own accumulators behind dynamic scales -- with the next unit's pieces requested behind them:
    //  same bits, 32 registers fewer, and SLOWER on a same-box A/B: fused launch 21.0 against 19.8 us, step 69.5 against 67.8.  The
    //  piece requests between the K-steps are worth more than the shorter dependency chains.)
    if (dyn) {
      // per-(row, 32 columns) source scales: each K-step's product is unscaled as it is added
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (AF < NS && s == AF) fragments(AF);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s % AF], wp[rx][s][0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][0], d, 0, 0, 0);
        if (KS == 2 || s < 2) accA = d * rsa[s % AF] + accA;
        else accB = d * rsa[s % AF] + accB;
        request_pieces(RC, s, ln);
        if (EARLY_REC && s == 0) {
          __builtin_amdgcn_sched_barrier(0);
          records(true);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // tanh outputs at their fixed scale: a whole half of the reduction accumulates inside the matrix instruction
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (AF < NS && s == AF) fragments(AF);
        if (KS == 2 || s < 2) {
          accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][1], accA, 0, 0, 0);
          accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s % AF], wp[rx][s][0], accA, 0, 0, 0);
          accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][0], accA, 0, 0, 0);
        } else {
          accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][1], accB, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s % AF], wp[rx][s][0], accB, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % AF], wp[rx][s][0], accB, 0, 0, 0);
        }
        request_pieces(RC, s, ln);
        if (EARLY_REC && s == 0) {
          __builtin_amdgcn_sched_barrier(0);
          records(true);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    TSTAMP(4 + 5 * ui);
    if constexpr (EARLY_REC) rq_commit();
    else rq_advance_plain();
    TSTAMP(5 + 5 * ui);

    // ---- epilogue (last chunk of a pass).  Columns beyond the layer and rows beyond the batch are NOT masked: their
    //      values are finite (clamped addresses, zero-padded weights), they never reach HBM (buffer bounds), a later layer
    //      multiplies such columns with zero weights, and such rows enter the weight-gradient fragments with factor 0.
    //      A lane owns NR rows of its column: rows 4 g + kh NR + rr.
    const bool last = SMALL || ex.c == ex.C - 1;
    const int sflags = X.sflags;
    const bool ubwd = (sflags & S_BWD) != 0;
    const bool bounded = !ubwd && (sflags & S_BND);            // outputs in [-1, 1] (tanh, sigmoid, softsign): fixed scale
    const bool has_img = X.dst_pl >= 0;
    const int col = 128 * exp_ + 16 * tw + n;
    const int row0l = 4 * g + kh * NR;                          // first of the lane's rows inside the row block
    float v[NR];
    typedef _Float16 halfR __attribute__((ext_vector_type(NR)));
    halfR hi, lo;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) v[rr] = 0.f, hi[rr] = (_Float16)0.f, lo[rr] = (_Float16)0.f;
    float rsb = TANH_RS;
    if constexpr (PUB_OFF > 0) {
      if (last && (sflags & S_WAIT)) wait_join();
    }
    if (last) {
      // the two halves of the reduction, added once
      float tot[NR];
      if constexpr (KS == 2) {
        // wave (tile, kh) keeps rows kh NR .. of the tile and hands the other wave's rows over: one 8-byte LDS write per lane,
        // a barrier, one 8-byte read of what the other wave left for this one (two copies of the area, by unit parity: a wave may
        // be a whole unit ahead of its partner's read, not two -- the barrier of the unit in between orders them)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2* const xarea = reinterpret_cast<f32x2*>(xch) + uparity * (NW_ * 64);
        const f32x2 give = kh ? f32x2{accA[0], accA[1]} : f32x2{accA[2], accA[3]};
        xarea[wave * 64 + lane] = give;
        __syncthreads();
        const f32x2 got = xarea[(wave ^ 8) * 64 + lane];
        tot[0] = (kh ? accA[2] : accA[0]) + got[0];
        tot[1] = (kh ? accA[3] : accA[1]) + got[1];
      } else {
        const f32x4 t4 = accA + accB;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) tot[rr] = t4[rr];
      }
      const float cmul = wsc[rx] * (dyn ? 1.f : X.src_rs);
      const int colc = min(col, X.n - 1);
      // The unit's kind -- forward / backward, with or without an activation (its derivative), with or without joined rows -- is
      // wave-uniform: ONE scalar switch in front of straight-line code per kind.  (Left as tests inside the loop over the lane's
      // rows the compiler branched per ELEMENT, and waited for each element's LDS operand before it asked for the next one's.)
      auto values = [&](auto UB_, auto AC_, auto JN_) __attribute__((always_inline)) {
        constexpr bool UB = decltype(UB_)::value != 0, JN = decltype(JN_)::value != 0;
        constexpr int AC = decltype(AC_)::value;   // 0: none, 1: tanh, 2: another activation (its kind in the flags)
        const int akind = (sflags >> S_KIND_SHIFT) & 15;
        if constexpr (!UB) {
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) {
            float x = fmaf(tot[rr], cmul, ep[rx][0]);
            if constexpr (AC == 1) x = tanh_f32(x);
            if constexpr (AC == 2) x = act_apply(akind, x);
            v[rr] = x;
          }
        } else {
          float jn[NR], hv[NR];
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) {
            jn[rr] = hv[rr] = 0.f;
            if constexpr (JN) jn[rr] = lat[(row0l + rr) * X.n + colc];
            if constexpr (AC) {
              if (MODE == CHAIN_FUSED) hv[rr] = reinterpret_cast<const float*>(lds + X.h_f32)[(row0l + rr) * X.h_f32_pitch + colc];
              else hv[rr] = ep[rx][BWD ? rr : 0];
            }
          }
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) {
            float x = tot[rr] * cmul;
            if constexpr (JN) x += jn[rr];
            if constexpr (AC == 1) x *= fmaf(-hv[rr], hv[rr], 1.f);
            if constexpr (AC == 2) x *= act_deriv_from_output(akind, hv[rr]);
            v[rr] = x;
          }
        }
      };
      switch (sflags & (S_ACT | S_JOIN | S_BWD | S_ACTX)) {
        case 0: values(IC_<0>{}, IC_<0>{}, IC_<0>{}); break;
        case S_ACT: values(IC_<0>{}, IC_<1>{}, IC_<0>{}); break;
        case S_BWD: values(IC_<1>{}, IC_<0>{}, IC_<0>{}); break;
        case S_BWD | S_ACT: values(IC_<1>{}, IC_<1>{}, IC_<0>{}); break;
        case S_BWD | S_JOIN: values(IC_<1>{}, IC_<0>{}, IC_<1>{}); break;
        case S_BWD | S_JOIN | S_ACT: values(IC_<1>{}, IC_<1>{}, IC_<1>{}); break;
        case S_ACT | S_ACTX: values(IC_<0>{}, IC_<2>{}, IC_<0>{}); break;
        case S_BWD | S_ACT | S_ACTX: values(IC_<1>{}, IC_<2>{}, IC_<0>{}); break;
        case S_BWD | S_JOIN | S_ACT | S_ACTX: values(IC_<1>{}, IC_<2>{}, IC_<1>{}); break;
        default: values(IC_<0>{}, IC_<0>{}, IC_<0>{}); break;   // (forward units carry no join)
      }
      if (X.dst_f32 >= 0) {   // (the copy's rows are padded to whole passes: every lane has a slot)
        float* f = reinterpret_cast<float*>(lds + X.dst_f32) + row0l * X.dst_f32_pitch + col;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) f[rr * X.dst_f32_pitch] = v[rr];
      }
      if (has_img) {
        float s[NR], rs[NR];
        unsigned bm = 0;
        if (bounded) {
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) s[rr] = 16384.f, rs[rr] = TANH_RS;
        } else {
          // per-row scale over the 32 columns of the tile PAIR (the K-step of the consumer): the waves of a pair (and, KS = 2, of
          // both halves of the rows) exchange their tiles' row maxima through LDS (all waves are here: the condition is
          // workgroup-uniform); two copies of the area by unit parity, as above
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          unsigned m[NR];
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) m[rr] = row16_max(abs_bits(v[rr]));
          unsigned (*const mxa)[R] = mxbuf[uparity];
          u32x4 own;
          if constexpr (KS == 2) {
            *reinterpret_cast<u32x2*>(&mxa[tw][row0l]) = u32x2{m[0], m[1]};
            __syncthreads();
            own = *reinterpret_cast<const u32x4*>(&mxa[tw][4 * g]);
          } else {
            own = u32x4{m[0], m[1], m[2], m[3]};
            // (all sixteen lanes of a row group write the same four maxima to the same address: the hardware merges them -- writing
            //  from one lane only changed nothing in SQ_LDS_BANK_CONFLICT and cost an exec-mask sequence: round 4, reverted)
            *reinterpret_cast<u32x4*>(&mxa[tw][4 * g]) = own;
            __syncthreads();
          }
          const u32x4 pm = *reinterpret_cast<const u32x4*>(&mxa[tw ^ 1][4 * g]);
          unsigned mm[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            mm[q] = max(own[q], pm[q]);
            bm = max(bm, mm[q]);
          }
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) {
            const unsigned mr = KS == 2 ? (kh ? mm[2 + rr] : mm[rr]) : mm[rr];
            pow2_scale_bits<ACT_TOP>(mr, s[rr], rs[rr]);
          }
          // (both waves of the pair and the 16 lanes of a row group write the same bytes: no branch)
          float* const scp = reinterpret_cast<float*>(lds + X.dst_sc + (4 * exp_ + (tw >> 1)) * 64 + g * 16) + kh * NR;
          if constexpr (KS == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2*>(scp) = f32x2{rs[0], rs[1]};
          } else {
            *reinterpret_cast<f32x4*>(scp) = f32x4{rs[0], rs[1], rs[2], rs[3]};
          }
        }
#pragma unroll
        for (int rr = 0; rr < NR; rr += 2) {   // (NR is 2 or 4)
          typedef _Float16 half2v __attribute__((ext_vector_type(2)));
          unsigned ph, pl;
          split_f16_pair(v[rr] * s[rr], v[rr + 1] * s[rr + 1], ph, pl);   // (three instructions a pair: emx_common.h)
          const half2v vh = __builtin_bit_cast(half2v, ph), vl = __builtin_bit_cast(half2v, pl);
          hi[rr] = vh[0], hi[rr + 1] = vh[1], lo[rr] = vl[0], lo[rr + 1] = vl[1];
        }
        char* dp = lds + X.dst_pl + col * 32 + 8 * g + kh * (2 * NR);
        *reinterpret_cast<halfR*>(dp) = hi;
        *reinterpret_cast<halfR*>(dp + X.dst_lo) = lo;
        if (X.frag != nullptr) {
          // the weight-gradient fragment: the image's pair under the block's scale (16 rows x 32 columns); rows beyond
          // the batch get factor 0
          halfR ratio;
          if (bounded) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) ratio[rr] = (_Float16)(row0l + rr < rows_valid ? 1.f : 0.f);
          } else {
            bm = max_xor32(max_xor16(bm));   // (lane swaps, not ds_bpermute: no trip through the LDS crossbar)
            float sb;
            pow2_scale_bits<ACT_TOP>(bm, sb, rsb);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) ratio[rr] = (_Float16)(row0l + rr < rows_valid ? sb * rs[rr] : 0.f);
          }
          hi = hi * ratio;
          lo = lo * ratio;
        }
      }
    }
    {
      // float32 rows: column col of rows 4 g + ..; a column beyond n would alias the next row inside the buffer's
      // bounds, so such a lane's offset is pushed out of them
      // (a tensor that is not kept: the same address arithmetic on its null pointer and a buffer of ZERO bytes, which drops every
      //  store -- a select of the size instead of a branch around the 64-bit arithmetic)
      const bool keep = last && X.out != nullptr;
      __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(X.out + (long)r0 * X.n, 0, keep ? max(0, rows_valid) * X.n * 4 : 0, 0x00020000);
      const unsigned vo = col < X.n ? (unsigned)((row0l * X.n + col) * 4) : 0x80000000u;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rr]), ro, vo + rr * X.n * 4, 0, 0);
      // weight-gradient fragment of the tile pair; column groups beyond the tensor do not exist
      const int grp = 4 * exp_ + (tw >> 1);
      const bool keepf = last && X.frag != nullptr && has_img && grp < X.CG;
      const long fblk = (long)rb * X.CG + grp;
      __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(X.frag + fblk * 2048, 0, keepf ? 2048 : 0, 0x00020000);
      if constexpr (KS == 2) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hi), rf, fvoff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lo), rf, fvoff + 1024, 0, 0);
      } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rf, fvoff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rf, fvoff + 1024, 0, 0);
      }
      const bool keeps = keepf && (tw & 1) == 0 && kh == 0;
      __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(X.rscale + fblk, 0, keeps ? 4 : 0, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rsb), rr_, lane == 0 ? 0u : 0x80000000u, 0, 0);
    }
    TSTAMP(6 + 5 * ui);
    // the layer's rows are complete in the destination image when its last unit is done
    const bool layer_end = SMALL || (last && ex.p == ex.P - 1);
    if (layer_end) __syncthreads();
    if constexpr (PUB_OFF > 0) {
      if (layer_end && (sflags & S_PUB) && wave == 0) {
        // the latent rows (two columns) for the tile workgroups of this launch (sk_tile.h): from the layer's float32 copy in LDS, by
        // ONE wave, as write-through stores -- 16 rows x 2 columns are one 128-byte line, one store instruction; the wave
        // drains it, then its lane 0 reports in.  The two pointers are fetched from the kernel arguments HERE (an offset the
        // optimiser cannot see through): held in registers over the whole unit loop they cost the loop its scalar registers.
        int off = PUB_OFF;
        asm volatile("" : "+s"(off));
        typedef const PubArgs __attribute__((address_space(4))) * KPub;
        const KPub kp = (KPub)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + off);
        const float* zc = reinterpret_cast<const float*>(lds + X.dst_f32);
        const int row = lane >> 1, c = lane & 1;
        if (lane < 2 * R && row < rows_valid) emx::sk::store_sc1(kp->zpub + (long)(r0 + row) * 2 + c, zc[row * X.dst_f32_pitch + c]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(kp->ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    TSTAMP(7 + 5 * ui);
    uparity ^= 1;
    ex = exn;
    X = Xn;
  };

  // GEMMs [g0, g1) through the ring, their units starting at `units`: the packages of the first NB - 1 units are requested,
  // `between` runs while they are in flight (staging the row block; the cost phase of the fused launch), then every
  // iteration executes one unit and requests the one NB - 1 ahead.
  // (Round 4 tried ONE stream of units over the forward and the backward GEMMs of the fused launch, the cost phase as a hook between
  //  two units -- the second pipeline's start-up is three dependent trips to the kernel arguments, ~1600 cycles.  With the hook inside
  //  the unit loop everything the cost phase holds is live across the loop: 142 -> 216 registers for the plain fused kernel and a
  //  thousand spilled registers in the 128-register builds.  Two pipelines it stays.)
  auto pipeline = [&](const int g0, const int g1, const char* units, auto between, auto LATE_) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(LATE_)::value != 0;
    rq = Stream{g0, g1, 0, 0, 1, 1};
    rq_live = true;
    rq_ptr = units + tw * 8192 + kh * (NS * 2048);
    rq_load_gemm();
    request(IC_<0>{}, lane);
#pragma unroll
    for (int s = 0; s < NS; ++s) request_pieces(IC_<0>{}, s, lane);
    // LATE (the backward pipeline of the fused launch): `between` -- the cost phase -- runs with the first package in flight and
    // BEFORE the rest of the streams' state exists: the execution stream's GEMM record alone is twenty scalar registers, and with
    // it live the cost code re-read its own scalars from the kernel arguments again and again (a hundred scalar loads, each waited
    // for: ~4 000 cycles).  Elsewhere the state is set up first, so that its scalar loads overlap the staging (the encoder backward
    // loses 1 300 cycles the other way round).
    if constexpr (LATE) between();
    rq_advance();
    if constexpr (NB >= 3) {
      request(IC_<1>{}, lane);
#pragma unroll
      for (int s = 0; s < NS; ++s) request_pieces(IC_<1>{}, s, lane);
      rq_advance();
    }
    ex = Stream{g0, g1, 0, 0, 1, 1};
    ex_load_gemm();
    if constexpr (!LATE) between();
    // iteration k executes the k-th unit from slot k % NB and requests the (k + NB - 1)-th into slot (k + NB - 1) % NB
    for (;;) {
      exec(IC_<NB - 1>{});
      if (ex.gi >= g1) break;
      exec(IC_<0>{});
      if (ex.gi >= g1) break;
      if constexpr (NB >= 3) {
        exec(IC_<1>{});
        if (ex.gi >= g1) break;
      }
    }
  };

  const bool with_cost = !BWD && p_with_cost != 0;
  // the row block travels ahead of the first packages and reaches LDS behind them
  float sx[8], shv[8], tv[4] = {0.f, 0.f, 0.f, 0.f};
  if (wave < p_in_Q) stage_load(wave, sx, shv);
  if (with_cost && (KS == 1 || tid < NTC)) {
    const int c = 4 * (tid & 31), row = tid >> 5;
    const long grow = (long)min(r0 + row, p_row_end - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) tv[i] = p_target[grow * p_ldt + min(c + i, p_cost_n - 1)];
  }
  auto join_rows = [&]() __attribute__((always_inline)) {
  if (BWD && a.join_n > 0) {
    // rows joined into dh of one layer: dadd and / or the latent gradient of the all-pairs cost (one wave per element,
    // lanes over the per-tile slots, fixed order)
    const int N = a.join_n;
    const bool vec_join = a.lg.partials != nullptr && a.lg.n_slots <= 64 && rows_valid == R && (a.lg.slot_stride & 3) == 0 &&
                          (((a.lg.row_base * N) | (r0 * N)) & 3) == 0 &&
                          ((reinterpret_cast<uintptr_t>(a.lg.partials) | reinterpret_cast<uintptr_t>(a.lg.Z) | reinterpret_cast<uintptr_t>(a.dadd)) & 15) == 0;
    if (vec_join) {
      // A slot's 16 x N partial sums of this row block are ONE run of 64 N bytes: lane = slot, a wave takes four consecutive elements
      // as one 16-byte load (the eight waves share the slots' cache lines), then sums each element over the lanes in the fixed
      // butterfly order.  As scalar loads -- one element per wave-load, 64 lanes on 64 different lines, 16 N of them per workgroup --
      // the address unit alone was busy for thousands of cycles (encoder backward at 4096 rows: the row block staged 10 000 cycles
      // after the wave's start).  Same sums in the same order as the element-wise form below: the same bits.
      for (int q = wave; 4 * q < R * N; q += NW_) {
        const int s0 = min(lane, a.lg.n_slots - 1);
        const float msk = lane < a.lg.n_slots ? 1.f : 0.f;
        const f32x4 pv = *reinterpret_cast<const f32x4*>(a.lg.partials + (long)(r0 - a.lg.row_base) * N + 4 * q + (long)s0 * a.lg.slot_stride);
        const f32x4 z4 = *reinterpret_cast<const f32x4*>(a.lg.Z + (long)r0 * N + 4 * q);
        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
        if (a.dadd != nullptr) d4 = *reinterpret_cast<const f32x4*>(a.dadd + (long)r0 * N + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sum = wave_sum(pv[i] * msk);
          if (lane == 0) lat[4 * q + i] = fmaf(a.lg.center_grad, z4[i], a.lg.gscale * sum) + d4[i];
        }
      }
    } else if (a.lg.partials != nullptr) {
      // (four elements per wave and pass, every load of the pass requested before the first sum: the partials were written
      //  by the all-pairs kernel on other XCDs a moment ago, a dependent trip per element was four trips)
      for (int el0 = wave; el0 < R * N; el0 += 4 * NW_) {
        float v[4], zz[4], dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int el = el0 + u * NW_;
          const int r = el / N, c = el - r * N;
          const bool rok = el < R * N && r0 + r < a.row_end;
          v[u] = 0.f, zz[u] = 0.f, dd[u] = 0.f;
          if (rok) {
            const float* __restrict__ P = a.lg.partials + (long)(r0 + r - a.lg.row_base) * N + c;
            // (up to 64 slots -- batches of up to 4096 rows -- are ONE load per lane and element, clamped and masked instead of
            //  looped: behind a loop the compiler waits for each element's load before it requests the next one's)
            const int s0 = min(lane, a.lg.n_slots - 1);
            v[u] = P[(long)s0 * a.lg.slot_stride] * (lane < a.lg.n_slots ? 1.f : 0.f);
            for (int sidx = lane + 64; sidx < a.lg.n_slots; sidx += 64) v[u] += P[(long)sidx * a.lg.slot_stride];
            const long gi = (long)(r0 + r) * N + c;
            zz[u] = a.lg.Z[gi];
            if (a.dadd != nullptr) dd[u] = a.dadd[gi];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int el = el0 + u * NW_;
          const float sum = wave_sum(v[u]);
          if (lane == 0 && el < R * N) lat[el] = fmaf(a.lg.center_grad, zz[u], a.lg.gscale * sum) + dd[u];
        }
      }
    } else {
      for (int el = tid; el < R * N; el += NT_) {
        const int r = el / N, c = el - r * N;
        lat[el] = r0 + r < a.row_end ? a.dadd[(long)(r0 + r) * N + c] : 0.f;
      }
    }
    __syncthreads();
  }
  };
  TSTAMP(1);
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto stage_rest = [&]() __attribute__((always_inline)) {
    join_rows();
    const int lo_off = a.in_Qp * 1024;
    if (wave < a.in_Q) {
      float v[8];
      stage_value(wave, sx, shv, v);
      stage_finish(v, wave, lo_off, a.in_pl, a.in_sc, a.in_aop, true);
    } else if (wave < a.in_Qp) {
      stage_finish(zero8, wave, lo_off, a.in_pl, a.in_sc, a.in_aop, false);
    }
    for (int grp = wave + NW_; grp < a.in_Qp; grp += NW_) {
      if (grp < a.in_Q) {
        float x[8], hv[8], v[8];
        stage_load(grp, x, hv);
        stage_value(grp, x, hv, v);
        stage_finish(v, grp, lo_off, a.in_pl, a.in_sc, a.in_aop, true);
      } else {
        stage_finish(zero8, grp, lo_off, a.in_pl, a.in_sc, a.in_aop, false);
      }
    }
    if (with_cost && (KS == 1 || tid < NTC)) {
      const int c = 4 * (tid & 31), row = tid >> 5;
      const long grow = (long)min(r0 + row, a.row_end - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c + i < a.cost.n) tgt[row * a.cost.n + c + i] = tv[i];
      for (int cc = c + 128; cc < a.cost.n; cc += 128)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (cc + i < a.cost.n) tgt[row * a.cost.n + cc + i] = a.cost.target[grow * a.cost.ldt + cc + i];
    }
    __syncthreads();
    TSTAMP(2);
  };
  if (MODE == CHAIN_FUSED) {
    pipeline(0, a.ng_fwd, a.units, stage_rest, IC_<0>{});
    TSTAMP(49);
    if (a.ng_fwd < a.ng) {
      // the forward part is complete.  The first backward packages are requested, then -- while they are in flight -- the
      // cost runs and d cost / d output goes to HBM and, re-staged, into the first backward GEMM's image
      pipeline(a.ng_fwd, a.ng, a.units_bwd, [&]() __attribute__((always_inline)) {
        TSTAMP(50);
        if (a.cost.vec4 && !a.cost.periodic_output) {   // cost and re-staging in one pass (cost_cols)
          cost_cols(IC_<1>{});
          TSTAMP(54);
          return;
        }
        cost_phase();
        TSTAMP(51);
        __syncthreads();
        TSTAMP(52);
        const CostArgs& cst = a.cost;
        const float* yb = reinterpret_cast<const float*>(lds + a.y_f32);
        const int wout = cst.ldy;
        for (int grp = wave; grp < a.top_Qp; grp += NW_) {
          const int col = 32 * grp + sc_c;
          const bool covered = (col >= cst.off_sin && col < cst.off_sin + cst.n) ||
                               (cst.periodic_output && col >= cst.off_cos && col < cst.off_cos + cst.n);
          float vv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = (covered && col < wout) ? yb[(8 * sc_hf + e) * a.y_pitch + min(col, wout - 1)] : 0.f;
          stage_finish(vv, grp, a.top_Qp * 1024, a.top_pl, a.top_sc, a.top_aop, grp < a.top_Q);
        }
        TSTAMP(53);
        __syncthreads();
        TSTAMP(54);
      }, IC_<1>{});
    } else {
      cost_phase();
    }
  } else {
    pipeline(0, a.ng, a.units, stage_rest, IC_<0>{});
  }
  TSTAMP(60);
  if (MODE == CHAIN_FWD && with_cost) cost_phase();
  TSTAMP(61);
#ifdef EMX_CHAIN_TIMING
  __syncthreads();
  if (blockIdx.x == 0 || blockIdx.x == 100) emx_dbg[(blockIdx.x ? 2048 : 0) + tid] = dbg_lds[tid];
#endif
}

// (the leading scalars duplicate what the first loads of a workgroup need -- the row block's source, the auto cost's targets:
//  preloaded into scalar registers at wave launch (-amdgpu-kernarg-preload-count), those loads go out with the first
//  instructions instead of behind a trip to the argument segment; 56 bytes, the struct follows at offset 56)
template <int MODE, int NB>
__global__ __launch_bounds__(NTC) void chain_kernel(const float* p_in, const float* p_top_h, const float* p_target, int p_ld_in, int p_in_cols,
                                                    int p_row0, int p_row_end, int p_ldt, int p_cost_n, int p_in_Q, int p_with_cost,
                                                    const ChainArgs a) {
  chain_body<MODE, NB, 0>(p_in, p_top_h, p_target, p_ld_in, p_in_cols, p_row0, p_row_end, p_ldt, p_cost_n, p_in_Q, p_with_cost, a);
}

// SMALL launches (every GEMM of the launch one unit: all layer widths of the range <= 128 -- the default network [128, 128, 2] on up
// to 128 input columns): the eight-wave build with the unit bookkeeping of one-unit GEMMs folded away.  (A sixteen-wave build, two
// waves per 16-column tile, was measured slower -- 25.8 against 21.9 us for the fused launch at 4096 rows -- and is gone.)
template <int MODE, int NB>
__global__ __launch_bounds__(NTC) void chain_kernel_small(const float* p_in, const float* p_top_h, const float* p_target, int p_ld_in, int p_in_cols,
                                                          int p_row0, int p_row_end, int p_ldt, int p_cost_n, int p_in_Q, int p_with_cost,
                                                          const ChainArgs a) {
  chain_body<MODE, NB, 0, 1, true>(p_in, p_top_h, p_target, p_ld_in, p_in_cols, p_row0, p_row_end, p_ldt, p_cost_n, p_in_Q, p_with_cost, a);
}

// The same kernels held to 128 registers per lane: TWO workgroups per CU (four waves per SIMD) where the plain build has room
// for one.  For batches of several row blocks per CU (8192 rows and more): a unit of the chain is a serial string of dependent
// trips -- LDS fragments, matrix instructions, epilogue, barrier -- and a second resident row block runs in its gaps.
template <int MODE, int NB>
__global__ __launch_bounds__(NTC) __attribute__((amdgpu_waves_per_eu(4, 4))) void chain_kernel_c128(
    const float* p_in, const float* p_top_h, const float* p_target, int p_ld_in, int p_in_cols, int p_row0, int p_row_end, int p_ldt, int p_cost_n,
    int p_in_Q, int p_with_cost, const ChainArgs a) {
  chain_body<MODE, NB, -1>(p_in, p_top_h, p_target, p_ld_in, p_in_cols, p_row0, p_row_end, p_ldt, p_cost_n, p_in_Q, p_with_cost, a);
}

// The fused launch with the all-pairs tiles aboard (sk_tile.h): workgroups [0, row blocks) are the chain workgroups, the rest run two
// tiles each.  128 registers per lane: a chain workgroup (two waves per SIMD) and a tile workgroup share a CU.  The role is
// told from the preloaded scalars alone (the chain workgroups' first loads must not wait for a trip to the arguments).
// SMALL: every GEMM of the launch one unit (chain_body) -- the default network on up to 128 input columns, i.e. the default
// configuration (B = 256, periodic) and every Euclidean batch of up to 1536 rows and 128 features that takes this launch.
template <int NB, bool FASTSIG, int NCH, int KIND, bool SMALL = false>   // KIND 0: Gram tiles on split rows (NCH chunks); 1 / 2: periodic 32 x 32 VALU tiles, float / phase form
__global__ __launch_bounds__(NTC) __attribute__((amdgpu_waves_per_eu(4, 4))) void chain_tiles_kernel(
    const float* p_in, const float* p_top_h, const float* p_target, int p_ld_in, int p_in_cols, int p_row0, int p_row_end, int p_ldt, int p_cost_n,
    int p_in_Q, int p_with_cost, const ChainArgs a, const emx::sk::TileRole tr) {
  const int n_chain = (p_row_end - p_row0 + R - 1) / R;
  if ((int)blockIdx.x >= n_chain) {
    extern __shared__ __align__(16) char lds[];
    if constexpr (KIND == 0) emx::sk::tile_pair_role<FASTSIG, NCH>(lds, tr, (int)blockIdx.x - n_chain);
    else emx::sk::tile32_pair_role<KIND, FASTSIG>(lds, tr, (int)blockIdx.x - n_chain);
    return;
  }
  static_assert(offsetof(emx::sk::TileRole, sync) == offsetof(emx::sk::TileRole, zpub) + 8 && sizeof(ChainArgs) % 8 == 0, "PubArgs inside TileRole");
  chain_body<CHAIN_FUSED, NB, 56 + (int)sizeof(ChainArgs) + (int)offsetof(emx::sk::TileRole, zpub), 1, SMALL>(p_in, p_top_h, p_target, p_ld_in, p_in_cols,
                                                                                                           p_row0, p_row_end, p_ldt, p_cost_n, p_in_Q,
                                                                                                           p_with_cost, a);
}

// The operand store from the flat parameter buffer (the optimizer keeps it current afterwards).  Two launches: the layers'
// exact maxima -> headers (scale, both slots seeded, wop.h), then one wave per 32 x 32 block writes the pieces.
__global__ __launch_bounds__(256) void mlp_mirror_max_kernel(const WopArgs wa, const float* __restrict__ params, char* __restrict__ store) {
  __shared__ unsigned red[4];
  const int l = blockIdx.x;
  const long nW = (long)wa.K[l] * wa.N[l];
  const float* W = params + wa.w_off[l];
  unsigned mx = 0;
  for (long i = threadIdx.x; i < nW; i += 256) mx = max(mx, abs_bits(W[i]));
  mx = wave_max_u32(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    float s, rs;
    pow2_scale_bits<WOP_TOP>(mx, s, rs);
    WopHeader* h = wop_header(store, l);
    h->rs = rs;
    h->pad = 0.f;
    h->slot[0] = h->slot[1] = mx;   // update index 0: any optimizer update supersedes it
  }
}
__global__ __launch_bounds__(64) void mlp_mirror_kernel(const WopArgs wa, const float* __restrict__ params, char* __restrict__ store) {
  __shared__ __align__(16) _Float16 img[512];
  const int lane = threadIdx.x;
  int l, k8, nb;
  wop_item(wa, blockIdx.x, l, k8, nb);
  const int K = wa.K[l], N = wa.N[l];
  const int k = 8 * k8 + (lane >> 3), c = 32 * nb + 4 * (lane & 7);
  const float* W = params + wa.w_off[l];
  float wv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wv[i] = (k < K && c + i < N) ? W[(long)k * N + c + i] : 0.f;
  float s, rs;
  pow2_scale_bits<WOP_TOP>((unsigned)wop_header(store, l)->slot[0], s, rs);
  wop_write_slab(wa, store, l, k8, nb, wv, s, lane, img);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients: partial[slice][l] = X_l^T delta_l over the slice's rows, from the fragments the row-block kernels left
// (aop.h).  grid = (128 x 128 output blocks of all layers) x slices; a wave owns a 64 x 64 quadrant = 2 x 2 tiles of
// v_mfma_f32_32x32x16_f16 whose reduction index runs over the 16 rows of ONE row block, so both operands of an instruction
// carry one power-of-two scale each; the row blocks of a slice are brought to the slice's common scales (below) and accumulate
// in the matrix unit.  Operands go straight from HBM/L2 to the matrix instruction: one 16-byte load per lane and fragment,
// no LDS, no barrier.
// db comes from the same (down-shifted) fragments on the VALU: v_dot2_f32_f16 with ones over the lane's eight rows.
// ---------------------------------------------------------------------------------------------------------
struct WgLayer {               // everything a workgroup needs to know about its layer: ONE 64-byte scalar load
  const char* xfrag;
  const float* xrs;
  const char* dfrag;
  const float* drs;
  long w_off, b_off;
  int K, N, CGx, CGd;
};
static_assert(sizeof(WgLayer) == 64, "WgLayer is read with one s_load_dwordx16");
struct WgHead {                // the first 128 bytes of the arguments: one s_load_dwordx16 x 2, every workgroup's first trip
  int tile_begin[MAXL + 1];    // prefix sum of 128x128 output blocks per layer (entry i: layer l0 + i); unused entries INT_MAX
  int nl;                      // number of layers in this launch
  int rb0, rb_end, rbs_per_slice;
  int has_fin;                 // the last blockIdx.x column is not a dW block: its y = 0 workgroup runs the step's scalar epilogue
  float* partial;              // [n_slices][n_params]
  long n_params;
  int pad[10];
};
static_assert(sizeof(WgHead) == 128 && MAXL == 12, "WgHead is read with two s_load_dwordx16");
struct WgradDev {
  WgHead h;
  WgLayer lay[MAXL];           // entry i: layer l0 + i
  FinishArgs fin;
};
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

#ifdef EMX_CHAIN_TIMING   // timeline of workgroups (0, 0), (0, last slice) and the scalar epilogue: tools/abl_wgrad.py
#define WSTAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (blockIdx.y == 0 || blockIdx.y == gridDim.y - 1)) \
    emx_dbg[4096 + ((blockIdx.x ? 2 : 0) + (blockIdx.y ? 1 : 0)) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define WSTAMP(k)
#endif

struct WgHd {
  int rb0, rb_end, rbs_per_slice, has_fin;
  float* partial;
  long n_params;
};
typedef const u32x16 __attribute__((address_space(4))) * WgKVec;

// One weight-gradient workgroup: output block bx (over all layers) of row slice by.  SC1: the partials are stored with sc1 (write
// through: workgroups of the SAME launch read them, mlp_wgrad_adam_kernel).
template <bool SC1>
__device__ __forceinline__ void wgrad_block(const unsigned (&tbw)[7], const int bx, const int by, const WgHd& hd, const WgKVec kv) {
  constexpr int ST_AUX = SC1 ? 16 : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int li = 0, tbase = 0;
#pragma unroll
  for (int i = 1; i < MAXL; ++i) {
    const int t = (int)((tbw[i >> 1] >> (16 * (i & 1))) & 0xffffu);   // (unused entries: 0xffff)
    if (bx >= t) {
      li = i;
      tbase = t;
    }
  }
  const int tb = bx - tbase;
  const u32x16 lrec = kv[li];
  WgLayer L;
  __builtin_memcpy(&L, &lrec, 64);
  const int Kin = L.K, N = L.N;
  if (Kin > 0) WSTAMP(6);   // (the layer record is in registers)
  const int nbn = (N + 127) >> 7;
  const int bi = tb / nbn, bj = tb - bi * nbn;
  const int ig0 = 4 * bi + 2 * (wave >> 1), jg0 = 4 * bj + 2 * (wave & 1);   // first column groups of the wave's quadrant
  const int CGx = L.CGx, CGd = L.CGd;
  const int rbs = hd.rb0 + by * hd.rbs_per_slice;
  const int rbe = min(rbs + hd.rbs_per_slice, hd.rb_end);
  float* __restrict__ out = hd.partial + (long)by * hd.n_params;
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
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  const half2v one2 = {(_Float16)1.f, (_Float16)1.f};

  if (iact[0] && jact[0]) {
    struct Frags {
      half8 x[2][2], d[2][2];   // [group][plane]
    };
    // Fragment loads: buffer loads with the lane's 16 bytes as the vector offset and the fragment's position as the SCALAR
    // offset (one s_add per load instead of a 64-bit multiply-add per lane: at one wave per SIMD every instruction of the
    // wave costs ~7 clocks of the slice's time).  The blocks' scales come with ONE vector load per tensor for the whole group
    // (lane 2 q + u: row block q, column group u), read back with v_readlane into scalar registers.
    const int igc[2] = {min(ig0, CGx - 1), min(ig0 + 1, CGx - 1)}, jgc[2] = {min(jg0, CGd - 1), min(jg0 + 1, CGd - 1)};
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(L.xfrag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(L.dfrag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rXs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.xrs), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rDs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.drs), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    auto load = [&](int rbc, Frags& f) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ox = (rbc * CGx + igc[u]) * 2048, od = (rbc * CGd + jgc[u]) * 2048;
        f.x[u][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ox, 0));
        f.x[u][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rX, lane16, ox + 1024, 0));
        f.d[u][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rD, lane16, od, 0));
        f.d[u][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rD, lane16, od + 1024, 0));
      }
    };
    // A group of GB row blocks (a 128-row slice is one group).  All fragments of the group are requested before the first
    // matrix instruction: the slice pays ONE trip to HBM/L2 instead of one per row block (a workgroup is alone on its CU,
    // one wave per SIMD: the 8 x 36 registers are there).  The blocks' power-of-two scales differ, so each fragment is
    // shifted DOWN to the group's common scale first (v_pk_mul_f16 by 2^-d: exact while the low plane stays above 2^-24,
    // i.e. for blocks within 2^10 of the group's largest; beyond that their last bits fall 2^-37 below that largest) --
    // then the products of all row blocks accumulate inside the matrix unit, four independent accumulators in turn (a
    // dependent v_mfma_f32_32x32x16_f16 waits 16 passes for its predecessor: back to back on one accumulator the matrix unit
    // idles half of the time, which is what the per-row-block zero / multiply / scaled add of the first version did).
    constexpr int GB = 8;
    auto ratio = [&](unsigned sblk, unsigned sgrp) __attribute__((always_inline)) {   // sblk / sgrp: bit patterns of powers of two, sblk <= sgrp
      const int d = min((int)(sgrp >> 23) - (int)(sblk >> 23), 30);
      return (_Float16)__uint_as_float((unsigned)(127 - d) << 23);
    };
    for (int g = rbs; g < rbe; g += GB) {
      Frags f[GB];
      // (bit patterns: the scales are positive, unsigned order = float order)
      const int sq = min(g + ((lane >> 1) & (GB - 1)), rbe - 1), su = lane & 1;
      // (buffer loads: a pointer out of the copied record is a generic one, and a flat load makes every later wait a full one)
      const unsigned vsx = __builtin_amdgcn_raw_buffer_load_b32(rXs, (sq * CGx + (su ? igc[1] : igc[0])) * 4, 0, 0);
      const unsigned vsd = __builtin_amdgcn_raw_buffer_load_b32(rDs, (sq * CGd + (su ? jgc[1] : jgc[0])) * 4, 0, 0);
      // (the memory counter of a wave counts to 63: the scales and seven row blocks are requested here, the eighth behind
      //  the first one's products, so that every wait names the loads it needs and the products start with the first arrival;
      //  row blocks beyond the slice repeat its last one -- straight-line requests, their products are skipped)
#pragma unroll
      for (int q = 0; q < GB - 1; ++q) load(min(g + q, rbe - 1), f[q]);
      WSTAMP(1);
      unsigned sx[GB][2], sd[GB][2], Rxb[2] = {0u, 0u}, Rdb[2] = {0u, 0u};
#pragma unroll
      for (int q = 0; q < GB; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // (row blocks beyond the slice repeat its last one: the maxima are the slice's)
          sx[q][u] = (unsigned)__builtin_amdgcn_readlane((int)vsx, 2 * q + u);
          sd[q][u] = (unsigned)__builtin_amdgcn_readlane((int)vsd, 2 * q + u);
          Rxb[u] = max(Rxb[u], sx[q][u]);
          Rdb[u] = max(Rdb[u], sd[q][u]);
        }
      const float Rx[2] = {__uint_as_float(Rxb[0]), __uint_as_float(Rxb[1])}, Rd[2] = {__uint_as_float(Rdb[0]), __uint_as_float(Rdb[1])};
      f32x16 ga[2][2];
      float gb[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; ++i) ga[0][0][i] = ga[0][1][i] = ga[1][0][i] = ga[1][1][i] = 0.f;
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        if (g + q < rbe) {
          Frags& fq = f[q];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const _Float16 rx = ratio(sx[q][u], Rxb[u]), rd = ratio(sd[q][u], Rdb[u]);
            if (rx != (_Float16)1.f) {   // (wave-uniform)
              fq.x[u][0] *= rx;
              fq.x[u][1] *= rx;
            }
            if (rd != (_Float16)1.f) {
              fq.d[u][0] *= rd;
              fq.d[u][1] *= rd;
            }
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {   // small products first: (h, l), (l, h), (h, h)
            const int px = c == 1 ? 1 : 0, pd = c == 0 ? 1 : 0;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int v = 0; v < 2; ++v)
                if (iact[u] && jact[v]) ga[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq.x[u][px], fq.d[v][pd], ga[u][v], 0, 0, 0);
          }
          if (do_bias) {   // db: the lane's eight rows of its column, both planes (v_dot2_f32_f16 with ones: exact products, f32 sums)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
              for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                  half2v pr = {fq.d[v][pl][e], fq.d[v][pl][e + 1]};
                  gb[v] = __builtin_amdgcn_fdot2(pr, one2, gb[v], false);
                }
          }
        }
        if (q == 0) load(min(g + GB - 1, rbe - 1), f[GB - 1]);
      }
      WSTAMP(2);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float sc = Rx[u] * Rd[v];
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[u][v][i] = fmaf(ga[u][v][i], sc, acc[u][v][i]);
        }
#pragma unroll
      for (int v = 0; v < 2; ++v) bacc[v] = fmaf(gb[v], Rd[v], bacc[v]);
    }
    WSTAMP(3);
    // store: tile (u, v), D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31] -- buffer stores: a lane outside
    // the layer's [K, N] gets an offset beyond the buffer and the hardware drops it (no exec masks, no branches)
    {
      float* wbase = out + L.w_off;
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(wbase, 0, Kin * N * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          if (iact[u] && jact[v]) {
            const int jj = 32 * (jg0 + v) + (lane & 31);
            const int i0 = 32 * (ig0 + u) + 4 * (lane >> 5);
            const unsigned vo = jj < N ? (unsigned)((i0 * N + jj) * 4) : 0x80000000u;   // (rows beyond K fall off the end of the buffer)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[u][v][reg]), rw, vo + ((reg & 3) + 8 * (reg >> 2)) * N * 4, 0, ST_AUX);
          }
        }
      if (do_bias) {
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(out + L.b_off, 0, N * 4, 0x00020000);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float bsum = sum_xor32(bacc[v]);   // rows 0..7 and 8..15 of the row blocks
          const int jj = 32 * (jg0 + v) + lane;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rb, (lane < 32 && jact[v]) ? (unsigned)(jj * 4) : 0x80000000u, 0, ST_AUX);
        }
      }
    }
  }
  WSTAMP(4);
}

// The first 16 dwords of the arguments are explicit scalars: with -mllvm -amdgpu-kernarg-preload-count=16 the command processor
// hands them over in scalar registers at wave launch (gfx940+), so the layer's record -- and behind it the fragments -- can be
// requested with the first instructions instead of after a trip for the head (tile table packed as 13 x 16 bits).
__global__ __launch_bounds__(NT) void mlp_wgrad_kernel(unsigned tb01, unsigned tb23, unsigned tb45, unsigned tb67, unsigned tb89, unsigned tbab,
                                                       unsigned tbcd, int rb0, int rb_end, int rbs_per_slice, int has_fin, int grid_x, float* partial,
                                                       long n_params, const WgradDev a) {   // (14 dwords are preloaded: up to `partial`)
  WSTAMP(0);
  __shared__ float fin_red[EMX_MAX_SUM_JOBS][FIN_NT / 64];
  __shared__ float fin_tl[64];
  typedef WgKVec KVec;
  typedef const char __attribute__((address_space(4))) * KBytes;
  const KVec kv = (KVec)((KBytes)__builtin_amdgcn_kernarg_segment_ptr() + 64 + offsetof(WgradDev, lay));   // (64: the explicit scalars)
  const WgHd hd{rb0, rb_end, rbs_per_slice, has_fin, partial, n_params};
  // (the explicit scalars are 64 bytes: 7 + 4 + 1 dwords, a pointer, a long -- the record array's offset above; grid_x = gridDim.x,
  //  which as a hidden argument would be a trip of its own)
  if (hd.has_fin && (int)blockIdx.x == grid_x - 1) {
    // The step's scalar epilogue (cost sums, history, counters) rides along as one extra workgroup: its inputs are complete
    // before this launch starts, and a launch of its own costs ~6.5 us of the step.
    if (blockIdx.y == 0) finish_step_body<emx::FQ_EMBEDDED>(a.fin, fin_red, fin_tl);
    WSTAMP(5);
    return;
  }
  const unsigned tbw[7] = {tb01, tb23, tb45, tb67, tb89, tbab, tbcd};
  wgrad_block<false>(tbw, (int)blockIdx.x, (int)blockIdx.y, hd, kv);
}

#ifdef EMX_CHAIN_TIMING   // timeline of the step's last launch (tools/abl_local.py): rows of 8 stamps at emx_dbg[8 * who]
#define LSTAMP(who, k) do { if (threadIdx.x == 0) emx_dbg[8 * (who) + (k)] = wall_clock64(); } while (0)
#else
#define LSTAMP(who, k)
#endif
constexpr unsigned WG_TAIL_SPIN_LIMIT = 1u << 22;   // x (one sc1 load + s_sleep): seconds -- the waits last microseconds
// (The optimizer as the tail of the SLICE workgroups of mlp_wgrad_kernel -- emx_mlp_wgrad_adam, rounds 3-5: every weight-gradient
//  workgroup resident, waiting inside the launch for all slice partials -- was bit-identical to the two launches and slower, the wait
//  cost more than the launch it saved (profiles/r03_tail_adam_ab.txt); removed in round 6.)

// ---------------------------------------------------------------------------------------------------------
// Small batches: weight gradients WITHOUT slice partials, the optimizer in the same workgroup (emx_mlp_wgrad_adam_local).
//   One workgroup of EIGHT waves owns one 32 x 32 block of one kernel (and, for the blocks of the first 32 in-features, 32 biases)
//   over ALL rows of the batch: each wave takes an eighth of the row blocks (v_mfma_f32_32x32x16_f16 on the fragments of aop.h, two
//   accumulators in turn, groups of four row blocks on a common scale as in mlp_wgrad_kernel, two groups in flight, every request
//   unconditional), the waves add their shares in LDS in a fixed order, and the first four then run L2 + clip + Adam and the
//   operand-store rewrite for one 8 x 32 slab of the block each -- whose parameters and moments they requested at the head of the
//   launch.  No partials of slices x parameters floats (emx_mlp_wgrad writes and re-reads 8.4 MB at B = 4096), no workgroup waits
//   for another, and a step ends with this launch: spare workgroups run the scalar epilogue and gather the next batch's rows (or
//   run the head of the next ADC step).  From 1024 rows on, two workgroups share the rows of a block (what bounds the launch is the
//   bytes the CUs at work have in flight): each leaves its share of the block in `shares` (sc1), takes a ticket, and the LAST one
//   adds the shares in the order of the parts -- whoever it is, the same bits.
//   Counters: every workgroup reads the optimizer step / the batch cursor and then counts in sync[1]; the epilogue workgroup --
//   the only one that waits, at its end -- advances them when all have counted, and clears sync[1].  Regulariser shares: the
//   epilogue reads the half of ws the previous launch wrote (parity of the optimizer step), this launch writes the other.
//   The matrix work of a wave grows with the batch (4096 rows: 192 instructions, ~5 us): the host takes this path for batches
//   of up to EMX_WGRAD_LOCAL_MAX_ROWS (4096) rows and networks of up to 1000 blocks (emx_mlp_wgrad_adam_local_fits;
//   tools/time_local_sweep.py: -0.2 .. -13 us per step against the two launches up to 4096 rows, a tie or a small loss beyond).
// ---------------------------------------------------------------------------------------------------------
struct WlTail {
  unsigned* sync;
  int64_t* cursor;
  int64_t* opt_step;
  int n_counted;       // workgroups that count in sync[1]: blocks + gather
  int gather_wgs;
  unsigned spin_limit;
  int reg_job;         // the epilogue's job that sums the regulariser shares (its source moves with the step's parity); -1: none
  short block_begin[MAXL];   // blocks of the layers before layer l (ws slots are numbered in layer order)
  int splits;          // workgroups that share the rows of a block (1: none of the hand-off below)
  float* shares;       // [splits][n_blocks][WL_SHARE]: the workgroups' shares of their blocks, for the last one of a block to add
};
struct WgradLocalDev {
  WgLayer lay[MAXL];
  FinishArgs fin;
  WlTail t;
  AdamArgs ad;
};
static_assert(sizeof(WgradLocalDev) + 64 <= 4096, "kernel arguments: 4 KB");
constexpr int WL_SHARE = 1024 + 32;   // a block's 32 x 32 sums in slab order (wave, lane, 4) and its 32 bias sums
constexpr int WL_GB = 4;   // row blocks per group (one trip to memory, one common scale)
constexpr int NTL = 512;   // eight waves: the row blocks of a block in eighths -- what bounds the launch is the bytes a CU has in flight
static_assert(NTL == 2 * FIN_NT, "the epilogue runs on the first FIN_NT threads");

__global__ __launch_bounds__(NTL) void mlp_wgrad_local_kernel(unsigned tb01, unsigned tb23, unsigned tb45, unsigned tb67, unsigned tb89, unsigned tbab,
                                                             unsigned tbcd, int rb0, int rb_end, int n_blocks, int per_xcd, int n_slots, const int64_t* p_step,
                                                             const char* p_mirror, const WgradLocalDev a) {
  __shared__ float fin_red[EMX_MAX_SUM_JOBS][FIN_NT / 64];
  __shared__ float fin_tl[64];
  __shared__ __align__(16) _Float16 img[4][512];
  __shared__ float red[NTL / 64][32][33];
  __shared__ float bred[NTL / 64][32];
  __shared__ float rreg[4];
  __shared__ float cm_part[emx::ADC_CM_PH][32];
  static_assert(emx::ADC_CM_PH * 32 == NTL, "the batch means: 32 columns x 16 row phases per workgroup");
  __shared__ unsigned ticket;
  typedef const char __attribute__((address_space(4))) * KBytes;
  const KBytes kseg = (KBytes)__builtin_amdgcn_kernarg_segment_ptr() + 64;   // (64: the explicit scalars)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x;
  unsigned* const sync = a.t.sync;
  const int splits = a.t.splits;
  // (the scalar epilogue and the workgroups that gather the next batch's rows come LAST in the grid.  Round 4 tried them first --
  //  dispatched ahead of the blocks they would run beside the whole fragment phase -- and the step got 5.6 us LONGER at 4096 rows
  //  (70.3 against 64.6 us): their loads then compete with the first, coldest fragment requests of every block workgroup)
  if (bx >= splits * n_slots) {
    const int e = bx - splits * n_slots;   // 0: the scalar epilogue; 1 ..: the next batch's rows
    if (e == 0) {
      if (tid >= FIN_NT) return;   // (the epilogue is written for FIN_NT threads; a wave that has ended takes no part in barriers)
      LSTAMP(1000, 0);
      const long st = *p_step;
      // (its cursor / counter2 are null: advanced below; the regulariser shares: the half of ws the previous launch wrote)
      finish_step_body<emx::FQ_EMBEDDED>(a.fin, fin_red, fin_tl, a.t.reg_job, (st & 1) * EMX_REDUCE_WS);
      LSTAMP(1000, 1);
      if (tid == 0) {
        unsigned spins = 0;
        bool gave_up = false;
        while (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.t.n_counted) {
          if (++spins > a.t.spin_limit) {
            __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        // (after a give-up the count is left alone: stragglers are still adding to it, and a count that is wrong from now on keeps
        //  every later launch of a captured run from looking healthy -- sync[2] is sticky and the host checks it wherever a
        //  result leaves the engine)
        if (!gave_up) __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.t.cursor) *a.t.cursor += 1;
        *a.t.opt_step = st + 1;
      }
      LSTAMP(1000, 2);
    } else {
      LSTAMP(512 + e - 1, 0);
      if (a.ad.next.adc_head != nullptr) {
        // the head of the next ADC step (emx_next_batch::adc_head; adc_head.h): two virtual workgroups of 256 threads for the rows per
        // workgroup, then workgroups of 32 columns x 16 row phases for the batch means
        const emx::AdcHeadArgs& h = *a.ad.next.adc_head;
        const long batch = *a.ad.next.cursor + a.ad.next.batch_step;
        if (!(h.n_batches > 0 && batch >= h.n_batches)) {
          const int64_t* __restrict__ idx = a.ad.next.idx + batch * (long)a.ad.next.B;
          const int row_wgs = (h.row_blocks + 1) >> 1;
          if (e - 1 < row_wgs) {
            const int vb = 2 * (e - 1) + (tid >> 8);
            if (vb < h.row_blocks) emx::adc_head_rows(h, idx, vb, h.row_blocks, tid & 255, a.ad.next.batch_step);
          } else if (h.feat_mean != nullptr) {
            emx::adc_head_means<32>(h, idx, (e - 1 - row_wgs) * 32, tid, cm_part);
          }
        }
      } else if (a.ad.next_vec4) {
        gather_rows2_multi<4>(a.ad.next, (e - 1) * (NTL / 64) + wave, a.t.gather_wgs * (NTL / 64), lane);
      } else {
        gather_rows_body(a.ad.next, (e - 1) * (NTL / 64) + wave, a.t.gather_wgs * (NTL / 64), lane);   // (rows without 16-byte lanes)
      }
      LSTAMP(512 + e - 1, 1);
      __syncthreads();   // (every wave has its cursor: the loads that use it are issued)
      if (tid == 0) __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // ---- one 32 x 32 block ----
  LSTAMP(bx, 0);
  const long step_now = *p_step;
  const unsigned tbw[7] = {tb01, tb23, tb45, tb67, tb89, tbab, tbcd};
  // Workgroup -> block.  Workgroups go to the XCDs in turn (workgroup x, x + 8, .. are XCD x's): by default groups of four
  // consecutive blocks -- which read the same in-feature fragments -- share an XCD (per_xcd = -4); per_xcd > 0 is the experiment
  // that gives every kernel a run of virtual indices inside as few XCDs as possible (half the HBM traffic, a longer launch:
  // profiles/r03_tail_adam_ab.txt); per_xcd = 0: workgroup = block.  Indices outside every kernel's run: nothing to do.
  // (with the rows of a block split over `splits` workgroups: part = bx / n_slots of them takes slot bx % n_slots)
  const int part = bx >= n_slots ? bx / n_slots : 0, bs = bx - part * n_slots;
  // (XCD-aware placement: part p of a kernel's blocks goes 4 p XCDs further, so that the parts do not crowd one XCD)
  // (per_xcd < 0: groups of G = -per_xcd consecutive blocks -- which share a fragment -- on one XCD, the groups dealt evenly:
  //  workgroups x, x + 8, .., x + 8 (G - 1) are one XCD's)
  const int G = -per_xcd;
  const int v = per_xcd > 0 ? (((bs & 7) + 8 - ((4 * part) & 7)) & 7) * per_xcd + (bs >> 3)
              : per_xcd < 0 ? G * ((bs / (8 * G)) * 8 + (bs & 7)) + ((bs >> 3) % G) : bs;
  int li = 0, tbase = (int)(tbw[0] & 0xffffu);
#pragma unroll
  for (int i = 1; i < MAXL; ++i) {
    const int t = (int)((tbw[i >> 1] >> (16 * (i & 1))) & 0xffffu);   // (unused entries: 0xffff)
    if (v >= t && t >= tbase) {
      li = i;
      tbase = t;
    }
  }
  if (v < tbase) return;
  typedef const u32x16 __attribute__((address_space(4))) * KVec;
  const u32x16 lrec = ((KVec)(kseg + offsetof(WgradLocalDev, lay)))[li];
  WgLayer L;
  __builtin_memcpy(&L, &lrec, 64);
  const u32x16 arec = ((KVec)(kseg + offsetof(WgradLocalDev, ad) + offsetof(AdamArgs, lay)))[li];
  AdamArgs::Layer AL;
  __builtin_memcpy(&AL, &arec, 64);
  const int CGx = L.CGx, CGd = L.CGd;
  const int sb = v - tbase;
  if (sb >= CGx * CGd) return;
  const int bi = sb / CGd, bj = sb - bi * CGd;
  const int blk = __builtin_amdgcn_readfirstlane(a.t.block_begin[li] + sb);   // the block's number in layer order (its slot of ws)
  const int k8 = 4 * bi + (wave & 3);                 // the slab of the block that one of the first four waves updates
  const bool slab_live = wave < 4 && 8 * k8 < L.K;
  AdamSlab slab = adam_slab_prefetch(a.ad, AL, p_mirror, slab_live ? k8 : 0, bj, lane);
  // the rows: `splits` workgroups share a block, the eight waves of each share its rows
  const int nrb_all = rb_end - rb0;
  const int per_wg = (nrb_all + splits - 1) / splits;
  const int wg0 = rb0 + part * per_wg, wg1 = min(wg0 + per_wg, rb_end);
  const int per = (max(wg1 - wg0, 0) + NTL / 64 - 1) / (NTL / 64);
  const int rbs = wg0 + wave * per, rbe = min(rbs + per, wg1);
  const bool do_bias = bi == 0;
  // (the 32 biases of the blocks of the first in-features: requested now, like the slabs)
  const bool bias_lane = do_bias && wave == 0 && lane < 32 && 32 * bj + lane < L.N;
  const long be = L.b_off + 32 * bj + (lane & 31);
  float bw = 0.f, bm = 0.f, bv = 0.f, bl2 = 0.f;
  if (bias_lane) {
    bw = a.ad.w[be], bm = a.ad.m[be], bv = a.ad.v[be];
    if (a.ad.l2coef) bl2 = a.ad.l2coef[be];
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bacc = 0.f;
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  const half2v one2 = {(_Float16)1.f, (_Float16)1.f};
  if (rbs < rbe) {
    struct Frags {
      half8 x[2], d[2];   // [plane]
    };
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(L.xfrag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(L.dfrag), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rXs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.xrs), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rDs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.drs), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    constexpr int GB = WL_GB;
    // a group's requests: the scales of its row blocks (lane q: row block g + q) and the four fragments of each
    auto issue = [&](int g, Frags (&f)[GB], unsigned& vsx, unsigned& vsd) __attribute__((always_inline)) {
      // (row blocks beyond the wave's range: an offset outside the buffer -- the request comes back at once with zeros and without
      //  a trip to memory, so nothing is outstanding behind the last products; the requests themselves stay unconditional)
      const int sq = min(g + (lane & (GB - 1)), rbe - 1);
      const int so = g < rbe ? 0 : (int)0x80000000;
      vsx = __builtin_amdgcn_raw_buffer_load_b32(rXs, (sq * CGx + bi) * 4 + so, 0, 0);
      vsd = __builtin_amdgcn_raw_buffer_load_b32(rDs, (sq * CGd + bj) * 4 + so, 0, 0);
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const bool in = g + q < rbe;
        const int rbc = in ? g + q : 0;
        const int vo = in ? lane16 : (int)0x80000000;
        const int ox = (rbc * CGx + bi) * 2048, od = (rbc * CGd + bj) * 2048;
        f[q].x[0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, ox, 0));
        f[q].x[1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, ox + 1024, 0));
        f[q].d[0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rD, vo, od, 0));
        f[q].d[1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rD, vo, od + 1024, 0));
      }
    };
    auto ratio = [&](unsigned sblk, unsigned sgrp) __attribute__((always_inline)) {   // as in wgrad_block
      const int d = min((int)(sgrp >> 23) - (int)(sblk >> 23), 30);
      return (_Float16)__uint_as_float((unsigned)(127 - d) << 23);
    };
    auto products = [&](int g, Frags (&f)[GB], unsigned vsx, unsigned vsd) __attribute__((always_inline)) {
      unsigned sx[GB], sd[GB], Rxb = 0u, Rdb = 0u;
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        sx[q] = (unsigned)__builtin_amdgcn_readlane((int)vsx, q);
        sd[q] = (unsigned)__builtin_amdgcn_readlane((int)vsd, q);
        Rxb = max(Rxb, sx[q]);
        Rdb = max(Rdb, sd[q]);
      }
      f32x16 ga[2];
      float gb = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ga[0][i] = ga[1][i] = 0.f;
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        if (g + q < rbe) {
          Frags& fq = f[q];
          const _Float16 rx = ratio(sx[q], Rxb), rd = ratio(sd[q], Rdb);
          if (rx != (_Float16)1.f) {   // (wave-uniform)
            fq.x[0] *= rx;
            fq.x[1] *= rx;
          }
          if (rd != (_Float16)1.f) {
            fq.d[0] *= rd;
            fq.d[1] *= rd;
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {   // small products first: (h, l), (l, h), (h, h); two accumulators in turn
            const int px = c == 1 ? 1 : 0, pd = c == 0 ? 1 : 0;
            ga[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq.x[px], fq.d[pd], ga[q & 1], 0, 0, 0);
          }
          if (do_bias) {
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                half2v pr = {fq.d[pl][e], fq.d[pl][e + 1]};
                gb = __builtin_amdgcn_fdot2(pr, one2, gb, false);
              }
          }
        }
      }
      const float sc = __uint_as_float(Rxb) * __uint_as_float(Rdb);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(ga[0][i] + ga[1][i], sc, acc[i]);
      bacc = fmaf(gb, __uint_as_float(Rdb), bacc);
    };
    // Two groups in flight, every request unconditional (row blocks beyond the wave's range repeat its last one; their products
    // are skipped): a request under a branch would make every wait behind it a full one.
    Frags fa[GB], fb[GB];
    unsigned vxa, vda, vxb, vdb;
    issue(rbs, fa, vxa, vda);
    for (int g = rbs; g < rbe; g += 2 * GB) {
      issue(g + GB, fb, vxb, vdb);
      products(g, fa, vxa, vda);
      issue(g + 2 * GB, fa, vxa, vda);
      products(g + GB, fb, vxb, vdb);
    }
  }
  LSTAMP(bx, 2);
  static_assert(NTL / 64 == 8, "the fixed-order sums below are written for eight waves");
  // the waves' shares: D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
  if (do_bias) {
    const float bsum = sum_xor32(bacc);   // rows 0..7 and 8..15 of the row blocks
    if (lane < 32) bred[wave][lane] = bsum;
  }
  __syncthreads();
  LSTAMP(bx, 3);
  // Every wave's step counter -- its first request -- arrived long ago: count the workgroup in (the epilogue workgroup advances
  // the counters when all have).  By a wave that has nothing else left to do: the others do not wait for the atomic.
  if (tid == NTL - 64) __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float reg = 0.f;
  const int kr = lane >> 3, c4 = lane & 7;
  float g[4] = {0.f, 0.f, 0.f, 0.f}, db = 0.f;
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * wave + kr, col = 4 * c4 + i;
      g[i] = ((red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col])) +
             ((red[4][row][col] + red[5][row][col]) + (red[6][row][col] + red[7][row][col]));
    }
    if (do_bias && wave == 0 && lane < 32)
      db = ((bred[0][lane] + bred[1][lane]) + (bred[2][lane] + bred[3][lane])) + ((bred[4][lane] + bred[5][lane]) + (bred[6][lane] + bred[7][lane]));
  }
  if (splits > 1) {
    // Rows split over workgroups: each leaves its share of the block (sc1 stores, drained) and takes a ticket; the LAST one
    // adds the shares in the order of the parts -- whoever it is, the same bits -- and updates the parameters.  Nobody waits.
    float* const mine = a.t.shares + ((long)part * n_blocks + blk) * WL_SHARE;
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(mine, 0, WL_SHARE * 4, 0x00020000);
    if (wave < 4) {
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      const u32x4s q = {__float_as_uint(g[0]), __float_as_uint(g[1]), __float_as_uint(g[2]), __float_as_uint(g[3])};
      __builtin_amdgcn_raw_buffer_store_b128(q, rS, (wave * 64 + lane) * 16, 0, 16);
      if (do_bias && wave == 0 && lane < 32) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db), rS, (1024 + lane) * 4, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(&sync[4 + blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)splits - 1u) return;
    if (tid == 0) __hip_atomic_store(&sync[4 + blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave < 4) {
      // every part's share requested at once (a part beyond `splits`: a buffer of zero bytes, which answers at once with zeros),
      // then added in the order of the parts -- as a loop of dependent loads each share was a trip to memory of its own
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      constexpr int MAXP = 4;
      u32x4s xs[MAXP];
      float bs_[MAXP];
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(a.t.shares + ((long)min(q, splits - 1) * n_blocks + blk) * WL_SHARE, 0,
                                                                            q < splits ? WL_SHARE * 4 : 0, 0x00020000);
        xs[q] = __builtin_amdgcn_raw_buffer_load_b128(rQ, (wave * 64 + lane) * 16, 0, 16);
        bs_[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rQ, (do_bias && wave == 0 && lane < 32) ? (1024 + lane) * 4 : (int)0x80000000, 0, 16));
      }
      g[0] = __uint_as_float(xs[0].x), g[1] = __uint_as_float(xs[0].y), g[2] = __uint_as_float(xs[0].z), g[3] = __uint_as_float(xs[0].w);
      db = bs_[0];
#pragma unroll
      for (int q = 1; q < MAXP; ++q) {
        if (q < splits) {   // (wave-uniform)
          g[0] += __uint_as_float(xs[q].x), g[1] += __uint_as_float(xs[q].y), g[2] += __uint_as_float(xs[q].z), g[3] += __uint_as_float(xs[q].w);
          db += bs_[q];
        }
      }
    }
  }
  LSTAMP(bx, 6);
  if (slab_live) {
    reg = adam_slab_finish(a.ad, AL, p_mirror, slab, g, k8, bj, lane, step_now, img[wave & 3]);
  }
  LSTAMP(bx, 7);
  if (bias_lane) {
    const float t = (float)(step_now + (a.ad.late ? 0 : 1) + a.ad.step_ahead);
    const float alpha = a.ad.lr * sqrtf(1.f - __builtin_amdgcn_exp2f(t * a.ad.log2_b2)) / (1.f - __builtin_amdgcn_exp2f(t * a.ad.log2_b1));
    adam_one(bw, db, bm, bv, bl2, a.ad, alpha, reg);
    a.ad.w[be] = bw;
    a.ad.m[be] = bm;
    a.ad.v[be] = bv;
  }
  LSTAMP(bx, 4);
  // the block's share of the regulariser cost of the NEW weights, for the next launch's epilogue (the other half of ws)
  reg = wave_sum(reg);
  if (lane == 0 && wave < 4) rreg[wave] = reg;
  __syncthreads();
  float* const wsn = a.ad.ws + ((step_now + 1) & 1) * EMX_REDUCE_WS;
  if (tid == 0) wsn[blk] = (rreg[0] + rreg[1]) + (rreg[2] + rreg[3]);
  if (blk == 0)
    for (int i = n_blocks + tid; i < EMX_REDUCE_WS; i += NTL) wsn[i] = 0.f;
  LSTAMP(bx, 5);
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
#include "wide.h"
#include "tchain.h"

int check_desc(const emx_mlp_desc* desc, const char* who) {
  EMX_REQUIRE(desc, "%s: null descriptor", who);
  EMX_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= MAXL, "%s: %d layers not supported (1..%d)", who, desc->n_layers, MAXL);
  for (int l = 0; l <= desc->n_layers; ++l) EMX_REQUIRE(desc->width[l] >= 1, "%s: layer width must be >= 1", who);
  for (int l = 0; l < desc->n_layers; ++l)
    EMX_REQUIRE(desc->act[l] >= ACT_NONE && desc->act[l] <= ACT_LAST, "%s: activation code %d of layer %d not available (0 .. %d, include/emx.h)", who,
                desc->act[l], l, (int)ACT_LAST);
  return EMX_OK;
}

int set_lds(const void* fn, size_t bytes, const char* who, size_t static_bytes = 1280) {   // (static: the chain kernels' exchange areas)
  if (bytes + static_bytes > 160 * 1024)
    return emx::fail(EMX_EUNSUPPORTED, "%s: layer width too large for the LDS row block (%zu + %zu bytes)", who, bytes, static_bytes);
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return emx::fail(EMX_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, bytes, hipGetErrorString(e));
  }
  return EMX_OK;
}

template <bool BWD>
int launch_wide(const WideArgs& a, hipStream_t st, const char* who) {
  const int rc = set_lds(reinterpret_cast<const void*>(&wide_layer_kernel<BWD>), WIDE_LDS, who, 0);
  if (rc) return rc;
  const int grid = (a.row_end - a.row0 + R - 1) / R;
  hipLaunchKernelGGL((wide_layer_kernel<BWD>), dim3(grid), dim3(NTC), WIDE_LDS, st, a);
  return emx::check_launch(BWD ? "wide-layer kernel (backward)" : "wide-layer kernel (forward)");
}

// Builds the launch description: LDS allocation (two ping-pong images sized per parity, float32 copies, side buffers) and
// the GEMM records.  Images are padded to whole 128-column chunks (regular units, wop.h).
struct ChainBuilder {
  ChainArgs a{};
  WopArgs wa{};
  const emx_mlp_desc* desc;
  const char* who;
  const char* wop;
  int ng = 0;
  int Qp[2] = {0, 0};
  int extra[2] = {0, 0};   // bytes another tenant of the parity's region needs (a float32 copy that lives there once the image is dead)
  int pl[2] = {0, 0}, sc[2] = {0, 0};
  int lds = 0;
  ChainBuilder(const emx_mlp_desc* d, const float* mirror, const char* w) : desc(d), who(w), wop(reinterpret_cast<const char*>(mirror)) {
    fill_wop_args(d, wa);
    a.lat_off = a.tgt_off = 0;
    a.y_f32 = -1;
  }
  static int padded_groups(int width) { return 4 * groups128(width); }
  void need(int parity, int width) { Qp[parity] = std::max(Qp[parity], padded_groups(width)); }
  int alloc(int bytes) {
    const int o = lds;
    lds += (bytes + 15) & ~15;
    return o;
  }
  void layout_images() {
    for (int p = 0; p < 2; ++p) {
      const int q = std::max(Qp[p], 4);
      pl[p] = alloc(std::max(q * (2048 + 64), extra[p]));
      sc[p] = pl[p] + q * 2048;
    }
  }
  static int f32_pitch(int n) { return 128 * groups128(n) + 4; }   // whole passes (every lane has a slot) + 4
  const char* fwd_units(int l) const { return wop + wa.units_begin + wa.fwd_unit[l] * WOP_UNIT; }
  const char* bwd_units(int l) const { return wop + wa.units_begin + wa.bwd_unit[l] * WOP_UNIT; }
  // one layer in one direction; returns the GEMM's index or a negative error.  src_tanh: the source image holds tanh
  // outputs (written at the fixed scale by the forward epilogue of the layer below)
  int add_gemm(int l, bool bwd, int src_par, bool src_tanh, int dst_par, const float* aux, int aux_rs, float* out, int sflags, const AopTensor& aop) {
    if (ng >= MAXG) return emx::fail(EMX_EUNSUPPORTED, "%s: more than %d GEMMs in one launch", who, MAXG);
    const int K = desc->width[l], N = desc->width[l + 1];
    const int kred = bwd ? N : K, nout = bwd ? K : N;
    GemmRec& G = a.g[ng];
    G.P = groups128(nout);
    G.C = groups128(kred);
    G.n = nout;
    G.sflags = sflags | (bwd ? S_BWD : 0);
    G.aux = aux;
    G.aux_rs = aux_rs;
    G.ws = reinterpret_cast<const float*>(wop + (long)l * sizeof(WopHeader));
    G.src_pl = pl[src_par];
    G.src_lo = padded_groups(kred) * 1024;
    G.src_sc = sc[src_par];
    G.src_rs = src_tanh ? TANH_RS : 0.f;
    G.dst_pl = dst_par >= 0 ? pl[dst_par] : -1;
    G.dst_lo = padded_groups(nout) * 1024;
    G.dst_sc = dst_par >= 0 ? sc[dst_par] : -1;
    G.dst_f32 = G.h_f32 = -1;
    G.dst_f32_pitch = G.h_f32_pitch = 0;
    G.out = out;
    G.aop = aop;
    return ng++;
  }
};

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
  a.cost.vec4 = (!cost->periodic_output && cost->variant != EMX_MEAN_NORM) ? 1 : 0;   // (the column-wise cost phase of the row-block kernels: cost_cols)
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
  // (the operand ring is two units deep in every build: a three-deep ring bought nothing at C2 -- the kernels are bound by
  //  instruction issue -- and cost registers; profiles/r04_nb_ab.txt)
  int rc;
  // several row blocks per CU and LDS for two of them: the 128-register build, two workgroups per CU
  constexpr int c128_min_blocks = 512;
  if (grid >= c128_min_blocks && 2 * (lds + 1024) <= 160 * 1024) {
    rc = set_lds(reinterpret_cast<const void*>(&chain_kernel_c128<MODE, 2>), lds, b.who);
    if (rc) return rc;
    hipLaunchKernelGGL((chain_kernel_c128<MODE, 2>), dim3(grid), dim3(NTC), lds, st, b.a.in, b.a.top_h, b.a.cost.target, b.a.ld_in, b.a.in_cols, b.a.row0,
                       b.a.row_end, b.a.cost.ldt, b.a.cost.n, b.a.in_Q, b.a.cost.ws != nullptr ? 1 : 0, b.a);
    return emx::check_launch(what);
  }
  // every GEMM one unit (all widths of the range <= 128): the SMALL build
  bool small = true;
  for (int i = 0; i < b.ng; ++i) small = small && b.a.g[i].P == 1 && b.a.g[i].C == 1;
  if (small) {
    rc = set_lds(reinterpret_cast<const void*>(&chain_kernel_small<MODE, 2>), lds, b.who);
    if (rc) return rc;
    hipLaunchKernelGGL((chain_kernel_small<MODE, 2>), dim3(grid), dim3(NTC), lds, st, b.a.in, b.a.top_h, b.a.cost.target, b.a.ld_in, b.a.in_cols, b.a.row0,
                       b.a.row_end, b.a.cost.ldt, b.a.cost.n, b.a.in_Q, b.a.cost.ws != nullptr ? 1 : 0, b.a);
    return emx::check_launch(what);
  }
  rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<MODE, 2>), lds, b.who);
  if (rc) return rc;
  hipLaunchKernelGGL((chain_kernel<MODE, 2>), dim3(grid), dim3(NTC), lds, st, b.a.in, b.a.top_h, b.a.cost.target, b.a.ld_in, b.a.in_cols, b.a.row0,
                     b.a.row_end, b.a.cost.ldt, b.a.cost.n, b.a.in_Q, b.a.cost.ws != nullptr ? 1 : 0, b.a);
  return emx::check_launch(what);
}

}  // namespace

extern "C" {

size_t emx_mlp_mirror_floats(const emx_mlp_desc* desc) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL) return 0;
  WopArgs wa;
  fill_wop_args(desc, wa);
  return (size_t)(wop_total_bytes(wa) / 4);
}

int emx_mlp_mirror(const emx_mlp_desc* desc, const float* params, float* mirror, emx_stream_t stream) {
  int rc = check_desc(desc, "emx_mlp_mirror");
  if (rc) return rc;
  EMX_REQUIRE(params && mirror, "emx_mlp_mirror: null pointer argument");
  WopArgs wa;
  fill_wop_args(desc, wa);
  // the zero padding of the layers (to whole 128 x 128 units) is written here, once; the optimizer rewrites real blocks only
  if (hipMemsetAsync(mirror, 0, (size_t)wop_total_bytes(wa), (hipStream_t)stream) != hipSuccess) return emx::check_launch("emx_mlp_mirror (memset)");
  hipLaunchKernelGGL(mlp_mirror_max_kernel, dim3((unsigned)desc->n_layers), dim3(256), 0, (hipStream_t)stream, wa, params,
                     reinterpret_cast<char*>(mirror));
  rc = emx::check_launch("mlp_mirror_max_kernel");
  if (rc) return rc;
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

}  // extern "C"

namespace {
constexpr size_t CHAIN_LDS_STATIC = 1280;   // the chain kernels' static exchange areas (set_lds)
inline bool chain_lds_fits(size_t dynamic_bytes) { return dynamic_bytes + CHAIN_LDS_STATIC <= 160 * 1024; }

// Layers [layer_begin, layer_end) as ONE launch of the row-block chain kernel.  dry_run: nothing is launched; EMX_OK if the
// range's LDS images fit, EMX_EUNSUPPORTED (no message) if not -- how the segmented passes below cut a range into launches.
int forward_chain(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                  int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, void* aop, int aop_rows,
                  emx_stream_t stream, bool dry_run) {
  const char* who = "emx_mlp_forward";
  int rc;
  ChainBuilder b(desc, mirror, who);
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  const int nf = layer_end - layer_begin;
  const int wout = desc->width[layer_end];
  for (int st = 0; st < nf; ++st) b.need(st & 1, desc->width[layer_begin + st]);
  // the cost's float32 copy of the output rows lives where the image of parity nf & 1 was: the last layer reads the other
  // parity, and every reader of this one is behind the barrier in front of the last layer
  if (cost) b.extra[nf & 1] = R * ChainBuilder::f32_pitch(wout) * 4;
  // a range that ends inside the network (a data-parallel rank's encoder pass) still leaves the fragments of its last
  // layer for the weight gradients of the layer above: they come out of the split, i.e. of a destination image
  const bool tail_frags = aop && layer_end < desc->n_layers;
  EMX_REQUIRE(!(tail_frags && cost), "%s: the cost belongs to the network's last layer", who);
  if (tail_frags) b.need(nf & 1, wout);
  b.layout_images();
  for (int st = 0; st < nf; ++st) {
    const int l = layer_begin + st;
    const bool last = st == nf - 1;
    // the fragments of h_l are the X operand of layer l + 1's weight gradient: not needed for the network's last layer
    const AopTensor ao = (aop && l + 1 < desc->n_layers) ? aop_tensor(aop, AL, 1 + l) : AopTensor{nullptr, nullptr, 0};
    const int gi = b.add_gemm(l, false, st & 1, st > 0 && act_bounded(desc->act[l - 1]), (!last || tail_frags) ? (st + 1) & 1 : -1, params + desc->b_off[l], 0,
                              h[l], act_flags(desc->act[l]), ao);
    if (gi < 0) return gi;
  }
  ChainArgs& a = b.a;
  a.ng = a.ng_fwd = b.ng;
  if (cost) {
    rc = fill_cost(a, cost, wout, cost->dY, who);
    if (rc) return rc;
    EMX_REQUIRE(cost->dY, "%s: incomplete cost description", who);
    GemmRec& G = a.g[b.ng - 1];
    G.dst_f32_pitch = a.y_pitch = ChainBuilder::f32_pitch(wout);
    G.dst_f32 = a.y_f32 = b.pl[nf & 1];
    a.tgt_off = b.alloc(R * cost->n * 4);
  }
  a.units = b.fwd_units(layer_begin);
  a.units_bwd = a.units;
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.in_Q = groups32(a.in_cols);
  a.in_Qp = ChainBuilder::padded_groups(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = (aop && layer_begin == 0) ? aop_tensor(aop, AL, 0) : AopTensor{nullptr, nullptr, 0};
  a.row0 = row_begin;
  a.row_end = row_end;
  if (dry_run) return chain_lds_fits((size_t)b.lds) ? EMX_OK : EMX_EUNSUPPORTED;
  return launch_chain<CHAIN_FWD>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (forward)");
}

// A forward range with a layer whose source rows do not fit the LDS images: cut into the longest runs of layers the chain kernel
// takes in one launch, and single layers through the streaming kernel (wide.h) where not even one layer fits.  The float32 rows
// h[l] hand a segment's result to the next one (they must be kept); every segment leaves the weight-gradient fragments of its last
// layer, the first one those of the range's input -- as the one launch would.
// plan_only: walk the cuts with dry runs and check every hand-over row, issue nothing -- the callers run this first, so that
// EMX_EUNSUPPORTED is always "decided in front of the launch: nothing has been issued" (include/emx.h), also inside a stream capture.
int forward_segmented(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in, int ld_in,
                      int row_begin, int row_end, float* const* h, void* aop, int aop_rows, emx_stream_t stream, bool plan_only) {
  const char* who = "emx_mlp_forward";
  WopArgs wa;
  fill_wop_args(desc, wa);
  const char* wop = reinterpret_cast<const char*>(mirror);
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  const float* cur = in;
  int ld = ld_in;
  int l = layer_begin;
  while (l < layer_end) {
    int m = layer_end;
    while (m > l && forward_chain(desc, params, mirror, l, m, cur, ld, row_begin, row_end, h, nullptr, aop, aop_rows, stream, true) != EMX_OK) --m;
    if (m > l) {
      const int rc = plan_only ? EMX_OK : forward_chain(desc, params, mirror, l, m, cur, ld, row_begin, row_end, h, nullptr, aop, aop_rows, stream, false);
      if (rc) return rc;
    } else if (plan_only) {
      m = l + 1;
    } else {
      m = l + 1;
      WideArgs w{};
      w.src = cur;
      w.ld_src = ld;
      w.K = desc->width[l];
      w.src_aop = (aop && l == 0) ? aop_tensor(aop, AL, 0) : AopTensor{nullptr, nullptr, 0};
      w.units = wop + wa.units_begin + wa.fwd_unit[l] * WOP_UNIT;
      w.ws = reinterpret_cast<const float*>(wop + (long)l * sizeof(WopHeader));
      w.P = groups128(desc->width[l + 1]);
      w.C = groups128(desc->width[l]);
      w.N = desc->width[l + 1];
      w.bias = params + desc->b_off[l];
      w.act = desc->act[l];
      w.out = h[l];
      w.ld_out = desc->width[l + 1];
      // the fragments of h_l are the X operand of layer l + 1's weight gradient: not needed behind the network's last layer
      w.out_aop = (aop && l + 1 < desc->n_layers) ? aop_tensor(aop, AL, 1 + l) : AopTensor{nullptr, nullptr, 0};
      w.row0 = row_begin;
      w.row_end = row_end;
      const int rc = launch_wide<false>(w, (hipStream_t)stream, who);
      if (rc) return rc;
    }
    if (m < layer_end && h[m - 1] == nullptr)
      return emx::fail(EMX_EUNSUPPORTED, "%s: a layer range this wide runs as several launches, which hand the float32 rows of the layers between them "
                       "on: this plan does not keep h[%d]", who, m - 1);
    cur = h[m - 1];
    ld = desc->width[m];
    l = m;
  }
  return EMX_OK;
}
}  // namespace

extern "C" {

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
  // the whole range in one launch where its LDS images fit; with the auto cost aboard only so (EMX_EUNSUPPORTED tells the caller
  // to run the cost as a kernel of its own behind the plain forward pass); a range too wide for one launch: in segments
  rc = forward_chain(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, cost, aop, aop_rows, stream, false);
  if (rc != EMX_EUNSUPPORTED || cost) return rc;      // (EMX_EUNSUPPORTED is decided in front of the launch: nothing has been issued)
  rc = forward_segmented(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, aop, aop_rows, stream, true);
  if (rc) return rc;                                   // (a missing hand-over row at ANY cut: refused before the first segment is issued)
  return forward_segmented(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, aop, aop_rows, stream, false);
}

int emx_mlp_forward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                    int ld_in, int row_begin, int row_end, float* const* h, void* aop, int aop_rows, emx_stream_t stream) {
  return emx_mlp_forward_cost(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, nullptr, aop, aop_rows, stream);
}

}  // extern "C"

namespace {
// Layers [layer_begin, layer_end) backward as ONE launch of the row-block chain kernel (dry_run: as forward_chain).
int backward_chain(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* dtop,
                   int ld_top, const float* dadd, int layer_add, const emx_latent_grad* latent, float* dbelow, int row_begin,
                   int row_end, float* const* h, float* const* delta, void* aop, int aop_rows, emx_stream_t stream, bool dry_run, bool lean = false) {
  // lean (the segments of backward_segmented): no LDS image is set aside for a result that has none -- dh below the range leaves
  // through `dbelow` only.  The one-launch path keeps its layout (and with it the builds its launches select).
  const char* who = "emx_mlp_backward";
  ChainBuilder b(desc, mirror, who);
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
  auto sflags = [&](int l) { return act_flags(desc->act[l]) | ((join && l == layer_add) ? S_JOIN : 0); };
  auto dop = [&](int l) { return aop ? aop_tensor(aop, AL, 1 + n + l) : AopTensor{nullptr, nullptr, 0}; };
  const int top = layer_end - 1;
  a.top_h = h[top];
  a.top_out = delta[top];
  a.top_flags = sflags(top);
  // boundary j = 0 is the top (width[layer_end]); GEMM of layer l: delta_l W_l^T -> dh_{l-1} -> delta_{l-1} (or dbelow)
  const int lo = dbelow ? layer_begin : layer_begin + 1;
  b.need(0, desc->width[layer_end]);
  for (int l = top; l >= lo; --l)
    if (!(lean && l == layer_begin)) b.need((top - l + 1) & 1, desc->width[l]);
  b.layout_images();
  for (int l = top; l >= lo; --l) {
    const int Kin = desc->width[l];
    const int st = top - l;
    const bool last = l == lo;
    int gi;
    if (l > layer_begin) {
      // delta_{l-1} is an operand of the next GEMM and / or of the weight gradients: both need the split, i.e. an image
      const AopTensor ao = dop(l - 1);
      gi = b.add_gemm(l, true, st & 1, false, (!last || ao.frag) ? (st + 1) & 1 : -1, desc->act[l - 1] ? h[l - 1] : params,
                      desc->act[l - 1] ? Kin : 0, delta[l - 1], sflags(l - 1), ao);
    } else {
      gi = b.add_gemm(l, true, st & 1, false, -1, params, 0, dbelow, 0, AopTensor{nullptr, nullptr, 0});
    }
    if (gi < 0) return gi;
  }
  if (b.ng == 0) {
    // a one-layer range without dbelow: only delta_top is formed (while the row block is staged); the pipeline still wants
    // one unit: the layer's own, with nothing kept
    const int gi = b.add_gemm(top, true, 0, false, -1, params, 0, nullptr, 0, AopTensor{nullptr, nullptr, 0});
    if (gi < 0) return gi;
    a.g[gi].P = a.g[gi].C = 1;
  }
  a.ng = a.ng_fwd = b.ng;
  if (join) a.lat_off = b.alloc(R * a.join_n * 4);
  a.units = b.bwd_units(top);
  a.units_bwd = a.units;
  a.in = dtop;
  a.ld_in = ld_top;
  a.in_cols = desc->width[layer_end];
  a.in_Q = groups32(a.in_cols);
  a.in_Qp = ChainBuilder::padded_groups(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = dop(top);
  a.row0 = row_begin;
  a.row_end = row_end;
  if (dry_run) return chain_lds_fits((size_t)b.lds) ? EMX_OK : EMX_EUNSUPPORTED;
  return launch_chain<CHAIN_BWD>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (backward)");
}

// A backward range whose top (or an inner boundary) is too wide for the LDS images: from the top down, the longest runs of layers
// the chain kernel takes in one launch, and single layers through the streaming kernel (wide.h).  Between two segments the raw
// d loss / d h of the boundary layer travels in that layer's delta rows (they must be kept): the upper segment leaves it there as
// its `dbelow`, the lower one takes it as its `dtop` and overwrites it, element by element, with delta = dh * act'(h) [+ join].
int backward_segmented(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* dtop, int ld_top,
                       const float* dadd, int layer_add, const emx_latent_grad* latent, float* dbelow, int row_begin, int row_end,
                       float* const* h, float* const* delta, void* aop, int aop_rows, emx_stream_t stream, bool plan_only) {
  const char* who = "emx_mlp_backward";   // (plan_only: as forward_segmented)
  WopArgs wa;
  fill_wop_args(desc, wa);
  const char* wop = reinterpret_cast<const char*>(mirror);
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  const int n = desc->n_layers;
  const bool join = dadd || latent;
  const float* cur = dtop;
  int ld = ld_top;
  int top_end = layer_end;                    // the segment in hand ends here (exclusive); cur = d loss / d h_{top_end - 1}
  while (top_end > layer_begin) {
    // the hand-over below a segment that starts at layer m > layer_begin: dh of layer m - 1, in delta[m - 1]
    auto below = [&](int m) -> float* { return m == layer_begin ? dbelow : delta[m - 1]; };
    auto seg_join = [&](int m) { return join && layer_add >= m && layer_add < top_end; };
    int m = layer_begin;
    bool no_room = false;
    for (; m < top_end; ++m) {
      if (m > layer_begin && delta[m - 1] == nullptr) {   // (no room for the hand-over at this boundary)
        no_room = true;
        continue;
      }
      const bool sj = seg_join(m);
      if (backward_chain(desc, params, mirror, m, top_end, cur, ld, sj ? dadd : nullptr, sj ? layer_add : 0, sj ? latent : nullptr, below(m), row_begin,
                         row_end, h, delta, aop, aop_rows, stream, true, true) == EMX_OK)
        break;
    }
    if (m < top_end) {
      const bool sj = seg_join(m);
      const int rc = plan_only ? EMX_OK : backward_chain(desc, params, mirror, m, top_end, cur, ld, sj ? dadd : nullptr, sj ? layer_add : 0,
                                                         sj ? latent : nullptr, below(m), row_begin, row_end, h, delta, aop, aop_rows, stream, false, true);
      if (rc) return rc;
    } else {
      // not even the top layer alone: its source delta_top = dtop * act'(h_top) is streamed
      const int top = top_end - 1;
      m = top;
      float* out = below(top);
      if (no_room || (top > layer_begin && out == nullptr))
        return emx::fail(EMX_EUNSUPPORTED, "%s: a layer range this wide runs as several launches, which hand d loss / d h on in the delta rows of "
                         "the layers between them: this plan keeps none (delta[%d] is null)", who, top - 1);
      EMX_REQUIRE(!(join && layer_add == top), "%s: rows cannot be joined into a layer %d columns wide (the streaming kernel takes none)", who,
                  desc->width[top + 1]);
      EMX_REQUIRE(!desc->act[top] || h[top], "%s: activations of layer %d are needed for its derivative", who, top);
      WideArgs w{};
      w.src = cur;
      w.ld_src = ld;
      w.K = desc->width[top + 1];
      w.src_h = desc->act[top] ? h[top] : nullptr;
      w.src_act = desc->act[top];
      w.src_out = delta[top];
      w.src_aop = aop ? aop_tensor(aop, AL, 1 + n + top) : AopTensor{nullptr, nullptr, 0};
      w.units = wop + wa.units_begin + wa.bwd_unit[top] * WOP_UNIT;
      w.ws = reinterpret_cast<const float*>(wop + (long)top * sizeof(WopHeader));
      w.P = groups128(desc->width[top]);
      w.C = groups128(desc->width[top + 1]);
      w.N = out ? desc->width[top] : 0;
      w.out = out;
      w.ld_out = desc->width[top];
      w.row0 = row_begin;
      w.row_end = row_end;
      const int rc = plan_only ? EMX_OK : launch_wide<true>(w, (hipStream_t)stream, who);
      if (rc) return rc;
    }
    if (m > layer_begin) {
      cur = delta[m - 1];
      ld = desc->width[m];
    }
    top_end = m;
  }
  return EMX_OK;
}
}  // namespace

extern "C" {

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
  rc = backward_chain(desc, params, mirror, layer_begin, layer_end, dtop, ld_top, dadd, layer_add, latent, dbelow, row_begin, row_end, h, delta, aop, aop_rows,
                      stream, false);
  if (rc != EMX_EUNSUPPORTED) return rc;              // (decided in front of the launch: nothing has been issued)
  rc = backward_segmented(desc, params, mirror, layer_begin, layer_end, dtop, ld_top, dadd, layer_add, latent, dbelow, row_begin, row_end, h, delta, aop,
                          aop_rows, stream, true);
  if (rc) return rc;                                   // (a missing hand-over row at ANY cut: refused before the first segment is issued)
  return backward_segmented(desc, params, mirror, layer_begin, layer_end, dtop, ld_top, dadd, layer_add, latent, dbelow, row_begin, row_end, h, delta, aop,
                            aop_rows, stream, false);
}

}  // extern "C"

namespace {
// what emx_mlp_forward_backward_tiles adds to the fused launch
struct TilesHost {
  const float* X;
  int B, D, ldx;
  float periodicity, data_span;
  const float* sig;
  void* sk_ws;
  size_t sk_ws_bytes;
  void* split_ws;
  float* zpub;
  unsigned* sync;
  // join (emx_mlp_forward_backward_tiles_join): the launch also runs the backward units below the latent layer -- the chain workgroups
  // wait for the tiles' partial sums and add them to dh of layer latent_layer themselves.  latent: NULL = no join
  int latent_layer;
  const emx_latent_grad* latent;
};

int device_cus();
int current_device();

template <bool FAST, int NCH, int KIND, bool SMALL>
int launch_chain_tiles_impl(ChainBuilder& b, const emx::sk::TileRole& tr, int grid, size_t lds, hipStream_t st) {
  int rc = set_lds(reinterpret_cast<const void*>(&chain_tiles_kernel<2, FAST, NCH, KIND, SMALL>), lds, b.who);
  if (rc) return rc;
  // The launch holds in-launch waits: a tile workgroup waits until ALL chain workgroups have published the latent rows, and (join)
  // every chain workgroup waits until ALL tile workgroups have reported.  Chain workgroups are dispatched first (lower indices), so
  // progress needs every chain workgroup AND at least one tile workgroup resident at once -- asked of the runtime for THIS
  // instantiation with THIS launch's LDS, on THIS device (a partitioned device, a larger batch through the C ABI): otherwise the
  // caller gets EMX_EUNSUPPORTED and keeps the separate launches.
  {
    static thread_local const void* cached_fn = nullptr;
    static thread_local size_t cached_lds = 0;
    static thread_local int cached_blocks = 0, cached_dev = -1;
    const void* fn = reinterpret_cast<const void*>(&chain_tiles_kernel<2, FAST, NCH, KIND, SMALL>);
    const int dev = current_device();
    if (cached_fn != fn || cached_lds != lds || cached_dev != dev) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, chain_tiles_kernel<2, FAST, NCH, KIND, SMALL>, NTC, lds) != hipSuccess) nb = 0;
      cached_fn = fn, cached_lds = lds, cached_blocks = nb, cached_dev = dev;
    }
    const long capacity = (long)cached_blocks * device_cus();
    if ((long)tr.n_chain + 1 > capacity)
      return emx::fail(EMX_EUNSUPPORTED, "%s: %d dense-layer workgroups + one tile workgroup do not fit the device at once (%d per CU x %d CUs): "
                       "the tiles cannot ride in this launch", b.who, tr.n_chain, cached_blocks, device_cus());
  }
  hipLaunchKernelGGL((chain_tiles_kernel<2, FAST, NCH, KIND, SMALL>), dim3(grid), dim3(NTC), lds, st, b.a.in, b.a.top_h, b.a.cost.target, b.a.ld_in, b.a.in_cols,
                     b.a.row0, b.a.row_end, b.a.cost.ldt, b.a.cost.n, b.a.in_Q, b.a.cost.ws != nullptr ? 1 : 0, b.a, tr);
  return emx::check_launch("mlp chain kernel (forward + cost + backward, all-pairs tiles aboard)");
}

template <bool FAST, int NCH, int KIND = 0>
int launch_chain_tiles(ChainBuilder& b, const emx::sk::TileRole& tr, int grid, size_t lds, hipStream_t st) {
  bool small = true;   // every GEMM one unit: the SMALL build of the dense-layer workgroups
  for (int i = 0; i < b.ng; ++i) small = small && b.a.g[i].P == 1 && b.a.g[i].C == 1;
  if (small) return launch_chain_tiles_impl<FAST, NCH, KIND, true>(b, tr, grid, lds, st);
  return launch_chain_tiles_impl<FAST, NCH, KIND, false>(b, tr, grid, lds, st);
}

std::atomic<long long> g_tchain_launches{0};

// The fused launch of a SMALL network as the transposed, register-resident chain (tchain.h).  EMX_EUNSUPPORTED (no message): the
// network, the cost or the number of row blocks is not what that kernel is built for -- the row-block chain kernel takes it.
int try_tchain(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in, int ld_in, int row_begin,
               int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin, float* const* delta, float* dbelow, void* aop, int aop_rows,
               const char* who, emx_stream_t stream) {
  // Taken while the launch is ONE round of workgroups (row blocks <= compute units: 117 KB of LDS, one workgroup per unit) -- there
  // a step with it is 2 us shorter than with the row-block chain kernel (63.6 against 65.6 us at 4096 x 128, profiles/r05_tchain_ab.txt);
  // beyond, two row-block workgroups share a unit and that kernel wins (8192 rows: 32.7 against 36 us for the launch alone).
  // EMX_TCHAIN=1: whenever the network qualifies; EMX_TCHAIN=0: never.
  // (read per call -- a call is one launch recorded into a graph, not a step: tests compare the two kernels inside one process)
  const char* const env = getenv("EMX_TCHAIN");
  const int mode = env ? atoi(env) : -1;
  const bool enabled = mode == 1 || (mode < 0 && (row_end - row_begin + R - 1) / R <= device_cus());
  const int n = desc->n_layers, top = layer_end - 1;
  if (!enabled || layer_begin != 0 || layer_end != n || n > TC_MAXL) return EMX_EUNSUPPORTED;
  for (int l = 0; l <= n; ++l)
    if (desc->width[l] > 128) return EMX_EUNSUPPORTED;
  for (int l = 0; l < n; ++l)
    if (desc->act[l] != ACT_NONE && desc->act[l] != ACT_TANH) return EMX_EUNSUPPORTED;
  if (cost->periodic_output || cost->variant == EMX_MEAN_NORM || (std::isfinite(cost->periodicity) && cost->periodicity > 0.f)) return EMX_EUNSUPPORTED;
  if (!(cost->target && cost->ws && cost->n > 0 && cost->ldt >= cost->n && cost->B > 0 && cost->off_sin >= 0 && cost->off_sin + cost->n <= desc->width[n]))
    return EMX_EUNSUPPORTED;   // (the row-block path reports what is wrong with the description)
  TChainArgs a{};
  int keep_slot[TC_MAXL];
  int kept = 0;
  for (int l = 0; l < n; ++l) {
    keep_slot[l] = -1;
    if (l >= bwd_begin && l < top && desc->act[l]) {
      if (kept == TC_KEEP) return EMX_EUNSUPPORTED;
      keep_slot[l] = kept++;
    }
  }
  WopArgs wa;
  fill_wop_args(desc, wa);
  const char* wop = reinterpret_cast<const char*>(mirror);
  AopLayout AL;
  if (aop) aop_layout(desc, aop_rows, AL);
  auto tensor = [&](int id) { return aop ? aop_tensor(aop, AL, id) : AopTensor{nullptr, nullptr, 0}; };
  auto rsw_of = [&](int l) { return reinterpret_cast<const float*>(wop + (long)l * sizeof(WopHeader)); };
  int ns = 0;
  for (int l = 0; l < n; ++l) {   // forward steps
    TStep& S = a.steps[ns++];
    S.bias = params + desc->b_off[l];
    S.rsw = rsw_of(l);
    S.width = desc->width[l + 1];
    S.act = desc->act[l];
    S.keep = keep_slot[l];
    if (l < top) {
      S.kind = TS_FWD;
      S.rows = h[l];
      const AopTensor t = tensor(1 + l);               // the fragments of h_l are the X operand of layer l + 1's weight gradient
      S.frag = t.frag, S.rscale = t.rscale, S.CG = t.CG;
    } else {
      S.kind = TS_FWD_TOP;                             // the output rows turn into d cost / d output = delta_top in place
      S.rows = delta[top];
      const AopTensor t = tensor(1 + n + top);
      S.frag = t.frag, S.rscale = t.rscale, S.CG = t.CG;
      a.top_rows = h[top];
    }
  }
  a.nl = n;
  const int lo = dbelow ? bwd_begin : bwd_begin + 1;
  for (int l = top; l >= lo; --l) {   // backward steps: the unit of layer l, result = delta_{l-1} (or dh below the range)
    TStep& S = a.steps[ns++];
    S.rsw = rsw_of(l);
    S.width = desc->width[l];
    if (l > bwd_begin) {
      S.kind = TS_BWD;
      S.act = desc->act[l - 1];
      S.keep = keep_slot[l - 1];
      S.rows = delta[l - 1];
      const AopTensor t = tensor(1 + n + l - 1);
      S.frag = t.frag, S.rscale = t.rscale, S.CG = t.CG;
    } else {
      S.kind = TS_BWD_BELOW;
      S.act = ACT_NONE;
      S.keep = -1;
      S.rows = dbelow;
    }
  }
  a.nsteps = ns;
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[0];
  {
    const AopTensor t = tensor(0);
    a.in_frag = t.frag, a.in_rscale = t.rscale, a.in_CG = t.CG;
  }
  a.units_fwd = wop + wa.units_begin + wa.fwd_unit[0] * WOP_UNIT;
  a.units_bwd = wop + wa.units_begin + wa.bwd_unit[top] * WOP_UNIT;
  a.target = cost->target;
  a.cost_ws = cost->ws;
  a.ldt = cost->ldt;
  a.cost_n = cost->n;
  a.cost_off = cost->off_sin;
  a.cost_variant = cost->variant;
  {
    CostArgs c{};
    fill_cost_scalars(c, cost->B, cost->n, desc->width[n], cost->ldt, cost->off_sin, cost->off_cos, 0, cost->periodicity, cost->variant, cost->coef);
    a.gnorm = c.gnorm;
  }
  a.row0 = row_begin;
  a.row_end = row_end;
  int rc = set_lds(reinterpret_cast<const void*>(&tchain_fused_kernel), TC_LDS, who, 0);
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  g_tchain_launches.fetch_add(1, std::memory_order_relaxed);
  hipLaunchKernelGGL(tchain_fused_kernel, dim3(grid), dim3(512), TC_LDS, (hipStream_t)stream, a);
  return emx::check_launch("transposed chain kernel (forward + cost + backward)");
}

int forward_backward_impl(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                          int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                          float* const* delta, float* dbelow, void* aop, int aop_rows, const TilesHost* tiles, const char* who,
                          emx_stream_t stream) {
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
  if (tiles == nullptr) {
    rc = try_tchain(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, cost, bwd_begin, delta, dbelow, aop, aop_rows, who, stream);
    if (rc != EMX_EUNSUPPORTED) return rc;      // (launched, or a real error; EMX_EUNSUPPORTED here only means: not a network for that kernel)
  }
  ChainBuilder b(desc, mirror, who);
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
    EMX_REQUIRE(!(last && ao.frag), "%s: the fused launch ends at the network's last layer", who);
    const int gi = b.add_gemm(l, false, st & 1, st > 0 && act_bounded(desc->act[l - 1]), last ? -1 : (st + 1) & 1, params + desc->b_off[l], 0, h[l],
                              act_flags(desc->act[l]), ao);
    if (gi < 0) return gi;
    // float32 copy of h_l in LDS: the cost's input (last layer), the activation derivative of the backward units, or the rows
    // the tile workgroups of the launch are waiting for (the latent layer, bwd_begin - 1)
    const bool for_bwd = desc->act[l] && l >= bwd_begin && l < top;
    const bool for_tiles = tiles != nullptr && l == (tiles->latent ? tiles->latent_layer : bwd_begin - 1);
    if (for_tiles) a.g[gi].sflags |= S_PUB;
    if (last || for_bwd || for_tiles) {
      GemmRec& G = a.g[gi];
      G.dst_f32_pitch = hpitch[l] = ChainBuilder::f32_pitch(desc->width[l + 1]);
      G.dst_f32 = hcopy[l] = b.alloc(R * G.dst_f32_pitch * 4);
    }
  }
  a.ng_fwd = b.ng;
  a.y_f32 = hcopy[top];
  a.y_pitch = hpitch[top];
  for (int l = top; l >= lo; --l) {
    const int st = top - l;
    const bool last = l == lo;
    int gi;
    if (l > bwd_begin) {
      // delta_{l-1} = (delta_l W_l^T) * act'(h_{l-1}); h_{l-1} is still in its float32 LDS copy
      const AopTensor ao = dop(l - 1);
      gi = b.add_gemm(l, true, st & 1, false, (!last || ao.frag) ? (st + 1) & 1 : -1, params, 0, delta[l - 1], act_flags(desc->act[l - 1]), ao);
      if (gi < 0) return gi;
      a.g[gi].h_f32 = hcopy[l - 1];
      a.g[gi].h_f32_pitch = hpitch[l - 1];
      if (tiles != nullptr && tiles->latent && l - 1 == tiles->latent_layer) {
        // dh of the latent layer: the all-pairs cost's gradient joins here, once the tile workgroups of the launch have left it
        a.g[gi].sflags |= S_JOIN | S_WAIT;
        a.lg = *tiles->latent;
        a.join_n = desc->width[l];
        a.lat_off = b.alloc(R * a.join_n * 4);
      }
    } else {
      gi = b.add_gemm(l, true, st & 1, false, -1, params, 0, dbelow, 0, AopTensor{nullptr, nullptr, 0});
      if (gi < 0) return gi;
    }
  }
  a.ng = b.ng;
  rc = fill_cost(a, cost, wout, delta[top], who);
  if (rc) return rc;
  a.tgt_off = b.alloc(R * cost->n * 4);
  a.top_pl = b.pl[0];
  a.top_sc = b.sc[0];
  a.top_Q = groups32(wout);
  a.top_Qp = ChainBuilder::padded_groups(wout);
  a.top_aop = dop(top);
  a.units = b.fwd_units(layer_begin);
  a.units_bwd = b.bwd_units(top);
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.in_Q = groups32(a.in_cols);
  a.in_Qp = ChainBuilder::padded_groups(a.in_cols);
  a.in_pl = b.pl[0];
  a.in_sc = b.sc[0];
  a.in_aop = (aop && layer_begin == 0) ? aop_tensor(aop, AL, 0) : AopTensor{nullptr, nullptr, 0};
  a.row0 = row_begin;
  a.row_end = row_end;
  if (tiles == nullptr)
    return launch_chain<CHAIN_FUSED>(b, row_begin, row_end, (size_t)b.lds, (hipStream_t)stream, "mlp chain kernel (forward + cost + backward)");
  // ---- the all-pairs tiles as extra workgroups of this launch (sk_tile.h)
  const int latent_layer = tiles->latent ? tiles->latent_layer : bwd_begin - 1;
  EMX_REQUIRE(layer_begin == 0 && latent_layer >= 0 && latent_layer < layer_end - 1, "%s: the tiles need the encoder's layers in the launch", who);
  if (tiles->latent) {
    EMX_REQUIRE(bwd_begin <= latent_layer && tiles->latent->partials && tiles->latent->Z && tiles->latent->n_slots > 0 && tiles->latent->n_slots <= 64,
                "%s: the join needs the backward range to reach below the latent layer and a latent-gradient description of at most 64 slots", who);
    EMX_REQUIRE(a.join_n > 0 && !desc->act[latent_layer], "%s: the join is built for a linear latent layer inside the backward range", who);
  }
  EMX_REQUIRE(row_begin == 0 && row_end == tiles->B, "%s: the tiles run over the whole batch (rows [0, %d))", who, tiles->B);
  EMX_REQUIRE(tiles->zpub && tiles->sync, "%s: null pointer argument (tile role)", who);
  rc = emx::ensure_counters_zeroed(tiles->sync, (tiles->latent ? 8 : 4) * sizeof(unsigned), (hipStream_t)stream, who);
  if (rc) return rc;
  emx::sk::TileRole tr{};
  int vblocks = 0, nch = 0, kind = 0;
  bool fast = false;
  rc = emx::sk::make_tile_role(tiles->X, tiles->B, tiles->D, tiles->ldx, h[latent_layer], desc->width[latent_layer + 1], tiles->periodicity, tiles->data_span,
                               tiles->sig, tiles->sk_ws, tiles->sk_ws_bytes, tiles->split_ws, &tr, &vblocks, &nch, &fast, &kind, who);
  if (rc) return rc;
  const int n_chain = (row_end - row_begin + R - 1) / R;
  tr.zpub = tiles->zpub;
  tr.sync = tiles->sync;
  tr.n_chain = n_chain;
  tr.join = tiles->latent ? 1 : 0;
  // persistent tile workgroups: one per CU beside its chain workgroup (more would wait for the chain workgroups to leave), each
  // walking its share of the tile pairs -- in the XCD-aware order 32 per XCD list (pairs k, k + 32, ... of the list)
  constexpr int forced_per_xcd = 0;
  const int per_xcd = forced_per_xcd ? forced_per_xcd : std::max(1, device_cus() / 8);
  if (tr.a.xcd_nbg > 0) tr.n_tile_wgs = 8 * std::min(per_xcd, tr.a.xcd_nbg * (tr.a.xcd_nbg + 1) / 2);
  else tr.n_tile_wgs = std::min(8 * per_xcd, (vblocks + 1) / 2);
  const size_t lds = std::max((size_t)b.lds, (size_t)emx::sk::TILE_ROLE_LDS_BYTES);
  int grid = n_chain + tr.n_tile_wgs;
  // (timing experiments only -- wrong results: EMX_TILES_DEBUG=1 launches the chain workgroups alone, =2 tile workgroups that do not wait)
  constexpr int dbg = 0;
  if (dbg == 1) grid = n_chain;
  if (dbg == 2) tr.n_chain = 0;
  if (dbg == 3) {   // the tile workgroups alone
    tr.n_chain = 0;
    grid = tr.n_tile_wgs;
    b.a.row_end = b.a.row0;
  }
  if (dbg == 4) tr.n_tile_wgs = std::min(tr.n_tile_wgs, 8);   // eight tile workgroups beside the chain workgroups: what do they cost the chain?
  hipStream_t st = (hipStream_t)stream;
  static_assert(sizeof(ChainArgs) + sizeof(emx::sk::TileRole) + 56 <= 4096, "kernel arguments of chain_tiles_kernel");
  if (kind != 0) {   // periodic rows: 32 x 32 VALU tiles (float form / 32-bit phases), any row length
    if (kind == 2) return fast ? launch_chain_tiles<true, 0, 2>(b, tr, grid, lds, st) : launch_chain_tiles<false, 0, 2>(b, tr, grid, lds, st);
    return fast ? launch_chain_tiles<true, 0, 1>(b, tr, grid, lds, st) : launch_chain_tiles<false, 0, 1>(b, tr, grid, lds, st);
  }
  if (!fast) {   // general sigmoid exponents (e.g. the cube example's (0.2, 3, 6 | 1, 2, 6)): rows of up to 128 features
    switch (nch) {
      case 1: return launch_chain_tiles<false, 1>(b, tr, grid, lds, st);
      case 2: return launch_chain_tiles<false, 2>(b, tr, grid, lds, st);
      case 4: return launch_chain_tiles<false, 4>(b, tr, grid, lds, st);
      default: return emx::fail(EMX_EUNSUPPORTED, "%s: the tile role with general sigmoid exponents is built for rows of up to 128 features", who);
    }
  }
  switch (nch) {
    case 1: return launch_chain_tiles<true, 1>(b, tr, grid, lds, st);
    case 2: return launch_chain_tiles<true, 2>(b, tr, grid, lds, st);
    case 4: return launch_chain_tiles<true, 4>(b, tr, grid, lds, st);
    case 8: return launch_chain_tiles<true, 8>(b, tr, grid, lds, st);
    default: return emx::fail(EMX_EUNSUPPORTED, "%s: the tile role is built for rows of 32, 64, 128 or 256 features (got %d chunks)", who, nch);
  }
}
}  // namespace

extern "C" {

int64_t emx_transposed_chain_count(void) { return (int64_t)g_tchain_launches.load(std::memory_order_relaxed); }

int emx_mlp_forward_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                             int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                             float* const* delta, float* dbelow, void* aop, int aop_rows, emx_stream_t stream) {
  return forward_backward_impl(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, cost, bwd_begin, delta, dbelow, aop,
                               aop_rows, nullptr, "emx_mlp_forward_backward", stream);
}

int emx_mlp_forward_backward_tiles(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                                   int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                                   float* const* delta, float* dbelow, void* aop, int aop_rows, const float* X, int D, int ldx, float periodicity,
                                   float data_span, const float sig[6], void* sk_ws, size_t sk_ws_bytes, void* split_ws, float* zpub, uint32_t* sync,
                                   emx_stream_t stream) {
  TilesHost t{X, row_end - row_begin, D, ldx, periodicity, data_span, sig, sk_ws, sk_ws_bytes, split_ws, zpub, sync, -1, nullptr};
  return forward_backward_impl(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, cost, bwd_begin, delta, dbelow, aop,
                               aop_rows, &t, "emx_mlp_forward_backward_tiles", stream);
}

int emx_mlp_forward_backward_tiles_join(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                                        int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                                        float* const* delta, float* dbelow, void* aop, int aop_rows, const float* X, int D, int ldx, float periodicity,
                                        float data_span, const float sig[6], void* sk_ws, size_t sk_ws_bytes, void* split_ws, float* zpub,
                                        uint32_t* sync, int latent_layer, const emx_latent_grad* latent, emx_stream_t stream) {
  EMX_REQUIRE(latent, "emx_mlp_forward_backward_tiles_join: null latent-gradient description");
  TilesHost t{X, row_end - row_begin, D, ldx, periodicity, data_span, sig, sk_ws, sk_ws_bytes, split_ws, zpub, sync, latent_layer, latent};
  return forward_backward_impl(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, cost, bwd_begin, delta, dbelow, aop,
                               aop_rows, &t, "emx_mlp_forward_backward_tiles_join", stream);
}

// rows per slice: EMX_WGRAD_SLICE_ROWS, doubled until at most EMX_WGRAD_MAX_SLICES slices remain -- every slice costs a
// write and a read of the whole gradient (partials), so their number must not grow with the batch
static int wgrad_slice_rows(int n_rows) {
  int r = EMX_WGRAD_SLICE_ROWS;
  while ((long)r * EMX_WGRAD_MAX_SLICES < n_rows) r *= 2;
  // small batches: shorter slices, so that a few hundred rows still spread over eight slices' worth of workgroups
  constexpr int min_rows = 32;
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
  a.h.has_fin = 0;
  if (finish) {
    int rcf = fill_finish_args(a.fin, finish->jobs, finish->n_jobs, finish->terms, finish->n_terms, finish->history, finish->term_history,
                               finish->capacity, finish->hstep, finish->cursor, finish->counter2, "emx_mlp_wgrad (finish)");
    if (rcf) return rcf;
    a.h.has_fin = 1;
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
  a.h.nl = layer_end - layer_begin;
  for (int i = 0; i <= MAXL; ++i) a.h.tile_begin[i] = INT_MAX;
  int tiles = 0;
  long lo = -1, hi = 0;
  for (int l = layer_begin; l < layer_end; ++l) {
    WgLayer& L = a.lay[l - layer_begin];
    const AopTensor X = aop_tensor(const_cast<void*>(aop), AL, l);          // id 0: the network input, id l = 1 + (l - 1): h_{l-1}
    const AopTensor Dl = aop_tensor(const_cast<void*>(aop), AL, 1 + n + l);
    L = WgLayer{X.frag, X.rscale, Dl.frag, Dl.rscale, (long)desc->w_off[l], (long)desc->b_off[l], desc->width[l], desc->width[l + 1], X.CG, Dl.CG};
    a.h.tile_begin[l - layer_begin] = tiles;
    tiles += ((desc->width[l] + 127) / 128) * ((desc->width[l + 1] + 127) / 128);
    const long wl = desc->w_off[l], wh = desc->b_off[l] + desc->width[l + 1];
    lo = (lo < 0) ? wl : std::min(lo, wl);
    hi = std::max(hi, wh);
  }
  a.h.partial = ws;
  a.h.n_params = n_params;
  const int rps = wgrad_slice_rows(nrows);
  a.h.rb0 = row_begin / 16;
  a.h.rb_end = (row_end + 15) / 16;
  a.h.rbs_per_slice = rps / 16;
  const int slices = emx_mlp_wgrad_slices(nrows);
  hipStream_t st = (hipStream_t)stream;
  static_assert(FIN_NT == NT, "the scalar epilogue shares the weight-gradient kernel's workgroup size");
  unsigned tbw[7];
  for (int i = 0; i < 7; ++i) {
    auto t16 = [&](int k) { return (unsigned)std::min(a.h.tile_begin[std::min(k, MAXL)], 0xffff) & 0xffffu; };
    tbw[i] = t16(2 * i) | (t16(2 * i + 1) << 16);
  }
  EMX_REQUIRE(tiles < 0xffff, "%s: %d output blocks", who, tiles);
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(tiles + a.h.has_fin, slices), dim3(NT), 0, st, tbw[0], tbw[1], tbw[2], tbw[3], tbw[4], tbw[5], tbw[6],
                     a.h.rb0, a.h.rb_end, a.h.rbs_per_slice, a.h.has_fin, tiles + a.h.has_fin, a.h.partial, a.h.n_params, a);
  rc = emx::check_launch("mlp_wgrad_kernel");
  if (rc) return rc;
  if (!grad) return EMX_OK;   // the caller sums the slices itself (emx_adam_step_slices)
  const long nn = hi - lo;
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((unsigned)((nn + NT - 1) / NT)), dim3(NT), 0, st, ws, (long)n_params, slices, lo, hi, grad);
  return emx::check_launch("mlp_wgrad_reduce_kernel");
}


namespace {
// compute units of the CURRENT device (a process may drive several devices, or partitions of one: the answer is kept per device
// ordinal, not per process)
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = 0;
  }
  return dev;
}
int device_cus() {
  constexpr int MAXDEV = 64;
  static std::atomic<int> cus[MAXDEV];   // 0: not asked yet
  const int dev = current_device();
  if (dev < 0 || dev >= MAXDEV) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
    return v;
  }
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
}  // namespace

// 32 x 32 blocks of all kernels; 0 if the launch is not made for this network / batch
static int wgrad_local_shape(const emx_mlp_desc* desc, int n_rows, const emx_next_batch* next, int* blocks_out, int* gather_out, int* splits_out) {
  constexpr int max_rows = 4096;
  constexpr int max_blocks = 1000;
  if (n_rows > max_rows) return 0;
  int blocks = 0;
  for (int l = 0; l < desc->n_layers; ++l) blocks += ((desc->width[l] + 31) / 32) * ((desc->width[l + 1] + 31) / 32);
  if (blocks > std::min(max_blocks, (int)EMX_REDUCE_WS)) return 0;
  int gather = 0;
  if (next) {
    if (next->adc_head) gather = (std::min(next->B, 2048) + 1) / 2 + (next->D + 31) / 32;   // (as emx_adam_step: row workgroups, then the means of D columns)
    else if (gather_vec4(next)) gather = (next->B + 2 * 4 * (NTL / 64) - 1) / (2 * 4 * (NTL / 64));   // four row pairs per wave, one pass
    else gather = std::min((next->B + NTL / 64 - 1) / (NTL / 64), 512);                            // a row per wave and pass
  }
  // the rows of a block over several workgroups: what bounds the launch is the bytes in flight, i.e. the number of CUs at work --
  // as many parts as fit the device beside the epilogue, at least 32 row blocks (four per wave) each, at most two (C2, 72 blocks:
  // 0.0735 / 0.0714 / 0.0725-0.0732 ms per step with 1 / 2 / 3 parts -- every part's share goes through memory)
  constexpr int max_splits = 2;   // (the last arriver adds at most four shares)
  const int nrb = (n_rows + 15) / 16;
  // at most 32 gather workgroups, two passes each at 4096 rows: 65.1-65.6 against 65.9-66.1 us per step with 64 (24 the same, 16: 66.0,
  // 8: 77.5; same box, alternating, profiles/r05_tail_gather_ab.txt) -- their loads compete with the block workgroups' fragment requests
  constexpr int gather_cap = 32;
  if (next && !next->adc_head) gather = std::min(gather, gather_cap);
  constexpr bool forced = false;   // (an experiment may ask for more workgroups than CUs: nobody waits for a sibling)
  int splits = std::min(std::min(max_splits, (nrb + 31) / 32), forced ? 4 : (std::max(device_cus(), 1) - 1) / std::max(blocks, 1));
  splits = std::max(splits, 1);
  *blocks_out = blocks;
  *gather_out = gather;
  *splits_out = splits;
  return 1;
}

size_t emx_mlp_wgrad_adam_local_ws_floats(const emx_mlp_desc* desc, int n_rows) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL || n_rows < 1) return 0;
  int b, g, sp;
  if (!wgrad_local_shape(desc, n_rows, nullptr, &b, &g, &sp)) return 0;
  return sp > 1 ? (size_t)sp * (size_t)b * WL_SHARE : 0;
}

int emx_mlp_wgrad_adam_local_fits(const emx_mlp_desc* desc, int n_rows, const emx_next_batch* next) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL || n_rows < 1) return 0;
  int b, g, sp;
  return wgrad_local_shape(desc, n_rows, next, &b, &g, &sp);
}

int emx_mlp_wgrad_adam_local(const emx_mlp_desc* desc, float* params, int64_t n_params, const void* aop, int aop_rows, int row_begin, int row_end,
                             float* ws, size_t ws_floats, const emx_finish_desc* finish, float* m, float* v, const float* l2coef, float lr, float beta1, float beta2, float eps,
                             float clip, float* reg_ws, float* mirror, const emx_next_batch* next, uint32_t* sync, emx_stream_t stream) {
  const char* who = "emx_mlp_wgrad_adam_local";
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && aop && m && v && reg_ws && mirror && sync && finish, "%s: null pointer argument", who);
  {
    const int rz = emx::ensure_counters_zeroed(sync, (4 + 1024) * sizeof(unsigned), (hipStream_t)stream, who);
    if (rz) return rz;
  }
  EMX_REQUIRE(finish->hstep && finish->cursor && finish->counter2, "%s: the epilogue's counters (history step, batch cursor, optimizer step) are required", who);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end, "%s: bad rows", who);
  rc = check_aop(aop, aop_rows, row_begin, row_end, who);
  if (rc) return rc;
  rc = check_next(next, who);
  if (rc) return rc;
  const int nrows = row_end - row_begin;
  int blocks = 0, gather = 0, splits = 1;
  if (!wgrad_local_shape(desc, nrows, next, &blocks, &gather, &splits))
    return emx::fail(EMX_EUNSUPPORTED, "%s: not made for this network / batch (emx_mlp_wgrad_adam_local_fits): use emx_mlp_wgrad + emx_adam_step_slices", who);
  const size_t need = splits > 1 ? (size_t)splits * blocks * WL_SHARE : 0;
  if (need > 0 && (!ws || ws_floats < need)) return emx::fail(EMX_EWORKSPACE, "%s: workspace %zu < %zu floats", who, ws ? ws_floats : (size_t)0, need);
  WgradLocalDev a{};
  rc = fill_finish_args(a.fin, finish->jobs, finish->n_jobs, finish->terms, finish->n_terms, finish->history, finish->term_history, finish->capacity,
                        finish->hstep, nullptr, nullptr, "emx_mlp_wgrad_adam_local (finish)");
  if (rc) return rc;
  a.t.reg_job = -1;
  for (int j = 0; j < finish->n_jobs; ++j)
    if (finish->jobs[j].src == reg_ws) a.t.reg_job = j;
  AopLayout AL;
  aop_layout(desc, aop_rows, AL);
  const int n = desc->n_layers;
  int begin[MAXL + 1], nblk[MAXL];
  for (int i = 0; i <= MAXL; ++i) begin[i] = 0xffff;
  int count = 0, largest = 1;
  for (int l = 0; l < n; ++l) {
    const AopTensor X = aop_tensor(const_cast<void*>(aop), AL, l);
    const AopTensor Dl = aop_tensor(const_cast<void*>(aop), AL, 1 + n + l);
    a.lay[l] = WgLayer{X.frag, X.rscale, Dl.frag, Dl.rscale, (long)desc->w_off[l], (long)desc->b_off[l], desc->width[l], desc->width[l + 1], X.CG, Dl.CG};
    EMX_REQUIRE(X.CG == (desc->width[l] + 31) / 32 && Dl.CG == (desc->width[l + 1] + 31) / 32, "%s: unexpected fragment layout", who);
    nblk[l] = X.CG * Dl.CG;
    a.t.block_begin[l] = (short)count;
    count += nblk[l];
    largest = std::max(largest, nblk[l]);
  }
  EMX_REQUIRE(count == blocks && blocks < 0x7fff, "%s: %d blocks", who, blocks);
  // virtual indices: per_xcd slots per XCD (at most the 32 CUs of one), a kernel's blocks in one run, aligned so that a run of up
  // to per_xcd blocks lies inside one XCD
  auto pow2ceil = [](int x) { int p = 1; while (p < x) p *= 2; return p; };
  // EMX_WGRAD_LOCAL_XCD: 0 blocks in layer order, workgroup = block; 2 / 4 (default 4): groups of that many consecutive blocks --
  // one row of a 128-wide kernel's blocks, which read the same in-feature fragments -- on one XCD, the groups dealt evenly over the
  // XCDs (C2: 0.0717 -> 0.0713 ms per step; pairs 0.0722); 1: a whole kernel's blocks on one XCD -- half the HBM traffic and a
  // LONGER launch (profiles/r03_tail_adam_ab.txt)
  constexpr int xcd_aware = 4;
  int per_xcd = std::min(32, std::max((blocks + 7) / 8, pow2ceil(largest)));
  int n_slots = 0;
  if (xcd_aware != 1) {
    const int G = (xcd_aware == 2 || xcd_aware == 4) ? xcd_aware : 0;
    per_xcd = -G;
    n_slots = G ? (blocks + 8 * G - 1) / (8 * G) * (8 * G) : blocks;
    for (int l = 0; l < n; ++l) begin[l] = a.t.block_begin[l];
  }
  for (; xcd_aware == 1; per_xcd = std::min(per_xcd * 2, 1 << 14)) {
    int pos = 0;
    for (int l = 0; l < n; ++l) {
      const int al = xcd_aware ? std::min(per_xcd, pow2ceil(nblk[l])) : 1;
      pos = (pos + al - 1) / al * al;
      begin[l] = pos;
      pos += nblk[l];
    }
    if (pos <= 8 * per_xcd || per_xcd >= (1 << 14)) {
      n_slots = 8 * per_xcd;
      break;
    }
  }
  EMX_REQUIRE(n_slots < 0xffff && begin[n - 1] + nblk[n - 1] <= n_slots, "%s: %d block slots", who, n_slots);
  unsigned tbw[7];
  for (int i = 0; i < 7; ++i) {
    auto t16 = [&](int k) { return (unsigned)std::min(begin[std::min(k, MAXL)], 0xffff) & 0xffffu; };
    tbw[i] = t16(2 * i) | (t16(2 * i + 1) << 16);
  }
  a.ad = make_adam(params, nullptr, nullptr, 0, 0, m, v, l2coef, n_params, lr, beta1, beta2, eps, clip, finish->counter2, reg_ws, desc, mirror, next);
  a.ad.late = 1;          // the regulariser shares left behind are those of the NEW weights ...
  a.ad.step_ahead = 1;    // ... while *step still counts the updates before this one
  a.ad.next.batch_step = 1;
  a.t.sync = sync;
  a.t.cursor = finish->cursor;
  a.t.opt_step = finish->counter2;
  a.t.gather_wgs = gather;
  a.t.n_counted = splits * blocks + gather;
  a.t.splits = splits;
  a.t.shares = ws;
  constexpr unsigned spin = WG_TAIL_SPIN_LIMIT;
  a.t.spin_limit = spin;
  hipLaunchKernelGGL(mlp_wgrad_local_kernel, dim3(splits * n_slots + 1 + gather), dim3(NTL), 0, (hipStream_t)stream, tbw[0], tbw[1], tbw[2], tbw[3], tbw[4], tbw[5],
                     tbw[6], row_begin / 16, (row_end + 15) / 16, blocks, per_xcd, n_slots, (const int64_t*)finish->counter2, (const char*)mirror, a);
  return emx::check_launch("mlp_wgrad_local_kernel");
}

}  // extern "C"

#ifdef EMX_CHAIN_TIMING
extern "C" int emx_debug_read(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg), sizeof(emx_dbg)); }
#endif

#ifdef TC_STAMP
extern "C" int emx_debug_tc_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tc_stamps), sizeof(unsigned long long) * 8 * 32 * 8);
}
#endif
