// K12: the dense encoder/decoder as fused row-block kernels on the exact-f32 matrix cores (gfx950).
//
// Replaces the per-layer Keras Dense ops of SequentialModel (encodermap/models/models.py:3179-3220: y = act(x W + b),
// W is [in, out]) and their autodiff (tape.gradient, models.py:3392-3393).  The layers are tiny (<= 3e5 parameters)
// and B is a few thousand rows, so per-layer GEMM launches are launch- and tail-bound (hipBLASLt: ~36 launches,
// 330 us per step at B=4096, D=128).  Here:
//
//   mlp_forward_kernel   one workgroup = 16 batch rows pushed through a RANGE of layers; the activations of the
//                        row block ping-pong between two LDS images, every layer's output is also stored to HBM
//                        for the backward pass.  Eight waves (two per SIMD) own one 16-column MFMA tile each;
//                        v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain, MI355X_MICROARCH.md) with A fragments read
//                        from LDS as one ds_read_b128 per four MFMAs and B fragments streamed from the L2-resident
//                        weights (64-B coalesced row segments), software-pipelined one K-step ahead.
//   mlp_backward_kernel  same row-block structure in reverse: delta_l = dh_l * act'(h_l) is stored for the weight
//                        gradients, dh_{l-1} = delta_l W_l^T uses 16-byte row loads of W_l (the k index of an MFMA
//                        may be assigned to lanes freely, so no transposed copy of the weights is needed).
//   mlp_wgrad_kernel     dW_l = h_{l-1}^T delta_l and db_l for ALL layers in one launch, split over row slices;
//                        mlp_wgrad_reduce_kernel adds the slice partials in fixed order (deterministic).
//
// Lane maps (cdna_hip_programming.md section 3): 16x16x4: A[m = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15],
// D[row = 4*(lane>>4) + reg][col = lane&15].  k-assignment used here: in K-step q lane group g = lane>>4 supplies
// k = 16q + 4g + i for the i-th of four consecutive MFMAs, so a lane's four A (or W-row) values are contiguous.
#include "cost.h"
#include "emx_common.h"
#include "finish.h"
#include "mirror.h"

namespace {

using namespace emx;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;           // weight-gradient kernels
constexpr int NTC = 512;          // row-block chain kernels: 8 waves, two per SIMD, one 16-column MFMA tile each
constexpr int R = 16;            // batch rows per workgroup
constexpr int MAXL = EMX_MAX_LAYERS;

struct MlpDev {
  int n_layers;
  int width[MAXL + 1];
  int act[MAXL];
  const float* W[MAXL];
  const float* b[MAXL];
  float* h[MAXL];        // activations, row stride = width[l+1]
  float* delta[MAXL];    // d loss / d pre-activation, row stride = width[l+1]
};

// Clamped (never masked) operand load.  No mask is needed: the other MFMA operand comes from an LDS image that is
// zero beyond the valid K range, and output columns beyond N are discarded in the epilogue, so out-of-range lanes
// may read any finite in-bounds value.  (A select on a loaded value makes hipcc emit an exec-masked load followed
// by s_waitcnt vmcnt(0) per element, which serialises the K loop on memory latency.)
__device__ __forceinline__ float ld_clamped(const float* __restrict__ p, int r, int c, int nr, int nc, int ld) {
  return p[(long)min(r, nr - 1) * ld + min(c, nc - 1)];
}

// tanh(x) = 1 - 2 / (e^{2x} + 1): two transcendental + three plain VALU instructions, absolute error <= ~1.5e-7 over the
// whole range (v_exp_f32 and v_rcp_f32 are 1-ulp; e = inf gives rcp = 0 -> 1, e = 0 gives -1), which is the float32
// resolution of the activations.  The epilogue of a layer is VALU-bound (two waves per SIMD finish their MFMAs together),
// so the ~50-instruction library tanhf would cost more than the layer's MFMAs.
__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);        // e^{2x}
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// ---------------------------------------------------------------------------------------------------------
// Row-block chain kernel (forward and backward-data share it).
//
// One workgroup = 16 batch rows pushed through a range of layers; the row block's activations (forward) or deltas
// (backward) ping-pong between two LDS images.  Eight waves, two per SIMD, one 16-column MFMA tile each
// (v_mfma_f32_16x16x4_f32: A = 16 rows x 4 k from LDS, B = 4 k x 16 columns of weights).
//
// The weights are the only HBM/L2 stream and every workgroup needs all of them (64 KB per 128x128 layer and CU), so
// the kernel is built around that stream:
//   * operand store ("mirror", emx_mlp_mirror): per layer and direction the weights are kept in the exact order the
//     waves consume them -- 1-KB blocks (tile of 16 output columns, K-step of 16), lane (g, m) owning the 16 bytes
//     {B[16q + 4g + i][16t + m], i < 4} -- so one global_load_dwordx4 per wave fetches eight whole 128-B lines
//     (the [in, out] kernel itself is read with half-used lines in either direction);
//   * measured load-to-use time of that stream with 256 workgroups pulling the same lines is 5-8 thousand cycles
//     (profiles/r01_chain_timeline.txt), a multiple of one layer's MFMA time (2048 cycles per SIMD).  The host
//     therefore flattens the layer range into UNITS (layer, pass of 128 columns, chunk of 128 k; one unit = up to
//     32 MFMAs per wave) and the kernel keeps the operands of NB - 1 units in flight in a register ring;
//   * vmcnt is one in-order counter for loads AND stores, so everything a unit will need from memory (operand
//     pieces, bias or h values for its epilogue) travels in one package NB - 1 units ahead, results are stored one
//     unit late (from registers, at the head of the next request burst), and no other vector-memory instruction
//     exists in the loop: every wait is for the OLDEST package only.
// ---------------------------------------------------------------------------------------------------------
#ifndef EMX_CHAIN_NB
#define EMX_CHAIN_NB 3
#endif
constexpr int MAXU = 40;   // units per launch
constexpr int NB = EMX_CHAIN_NB;      // operand ring depth: packages of NB - 1 units ahead in flight (measured at C2: 3 is 3 % faster than 4 or 2)

struct UnitLoad {       // what a request burst needs to fetch a unit's package (32 B, one s_load_dwordx8)
  int op_off;           // float offset of the operand block (first tile of the pass, first K-step of the chunk)
  int Q;                // K-steps per tile of this operand (tile pitch in 256-float blocks)
  int kq_tiles;         // K-steps in this chunk, 1..8 | tiles of this pass that exist, 1..8 (waves beyond them re-read the last one) << 8
  int col0;             // first output column of the pass
  int n;                // number of columns of aux (clamp bound)
  int rstride;          // row stride of aux: 0 = one broadcast row (a bias), n = one value per batch row (h rows)
  const float* aux;     // epilogue operand: the layer's bias (forward), h rows for the activation derivative (backward)
};
struct UnitExec {       // what the MFMA phase and the epilogue need (48 B)
  int kq_cols;          // kq | valid output columns of this pass << 8
  int src_off;          // LDS float offset of the chunk's first column in row 0 of the source image
  int dst_off;          // column offset of the pass in the destination image and in the output rows
  int flags;            // U_* | S_* << 8
  float* out;           // [*, n] result rows (h_l forward; delta_{l-1} or dbelow backward)
  int n;
  int src_pitch;        // row pitch of the source image
  int dst_base, dst_pitch;   // destination image: LDS float offset of row 0, row pitch
  int h_base1, h_pitch;      // image of h for the activation derivative: offset + 1 (0: h comes through aux), row pitch
};
// LDS images: R rows each, every image as wide as the widest layer it ever holds (whole 128-column passes + 4: the
// passes are written whole, and +4 keeps b128 rows conflict-free).  Image 0 starts at offset 0 and receives the row block.
constexpr int MAXIMG = 12;
struct ImgGeom {
  int n = 0, width[MAXIMG] = {}, pitch[MAXIMG] = {}, base[MAXIMG] = {}, total = 0;
  void need(int i, int w) {
    n = std::max(n, i + 1);
    width[i] = std::max(width[i], w);
  }
  void layout() {
    total = 0;
    for (int i = 0; i < n; ++i) {
      pitch[i] = ((std::max(width[i], 1) + 127) / 128) * 128 + 4;
      base[i] = total;
      total += R * pitch[i];
    }
  }
};
enum { U_FIRST = 1, U_LAST = 2, U_LAYER_END = 4, U_BWD = 8 };
enum { S_ACT = 1, S_JOIN = 2 };

struct ChainArgs {
  UnitLoad ul[MAXU];
  UnitExec ux[MAXU];
  int nu;
  int pitch;            // row pitch of image 0
  const float* mbase;   // operand store
  const float* in;      // forward: input rows; backward: d loss / d h_top
  int ld_in, in_cols;
  int row0, row_end;
  // backward only: the top of the range becomes delta_top = (in [+ join]) * act'(h_top), stored to top_out
  const float* top_h;
  float* top_out;
  int top_flags;
  // join: rows added to dh of one layer (S_JOIN): dadd and / or the latent gradient of the all-pairs cost
  const float* dadd;
  emx_latent_grad lg;
  int join_n;           // width of the joined layer (0: no join)
  // forward only, optional auto cost on the rows of the last layer (replaces emx_periodic_cost_fwd_bwd behind it):
  // cost.dY receives d cost / d output, cost.ws[workgroup] the workgroup's share of the cost
  CostArgs cost;
  int img_floats;       // floats of all LDS images (2 ping-pong images; fused mode: one per forward layer + 2)
  int cost_base, cost_pitch;   // image that holds the rows the cost is evaluated on
  // fused mode (forward, cost, backward through the upper layers in one launch): the cost runs behind unit cost_at and
  // leaves d cost / d output in the image at top_base, the source of the first backward unit
  int cost_at, top_base, top_pitch;
};

#ifdef EMX_CHAIN_TIMING   // in-kernel timeline (tools/abl_timing.py); stamps go to LDS so that they do not touch vmcnt
__device__ long long emx_dbg[8192];
#define TSTAMP(slot) do { if (MODE != CHAIN_BWD && lane == 0 && (slot) < 64) dbg_lds[wave * 64 + (slot)] = clock64(); } while (0)
#else
#define TSTAMP(slot)
#endif

template <int V>
struct IC {
  static constexpr int value = V;
};

enum { CHAIN_FWD = 0, CHAIN_BWD = 1, CHAIN_FUSED = 2 };

template <int MODE>
__global__ __launch_bounds__(NTC) void chain_kernel(const ChainArgs a) {
  constexpr bool BWD = MODE == CHAIN_BWD;
  extern __shared__ __align__(16) float lds[];
  __shared__ float cred[NTC / 64];
#ifdef EMX_CHAIN_TIMING
  __shared__ long long dbg_lds[512];
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int n0 = wave * 16;
  const int r0 = a.row0 + blockIdx.x * R;
  const int pitch = a.pitch;
  const int nu = a.nu;
  float* lat = lds + a.img_floats;   // [R][join_n] rows joined into one layer's dh (backward)
  float* tgt = lds + a.img_floats;   // [R][cost.n] target rows of the auto cost (forward), fetched with the row block
  constexpr int NE = MODE == CHAIN_BWD ? 4 : 1;   // epilogue operands per lane: h values of 4 rows (backward) or one bias (the fused
                                                  // launch's backward units take h from the LDS images)
  TSTAMP(0);

  float b[NB][8][4];
  float e[NB][NE];

  // one unit's package: its epilogue operands + 8 operand pieces.  Addresses are clamped, never masked or branched
  // around, so that every package is the same instruction sequence.
  auto piece_ptr = [&](const UnitLoad& L) { return a.mbase + L.op_off + (long)(min(wave, (L.kq_tiles >> 8) - 1) * L.Q) * 256 + lane * 4; };
  auto request_piece = [&](const UnitLoad& L, const float* __restrict__ p, int q, float (&bq)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + min(q, (L.kq_tiles & 0xff) - 1) * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = v[i];
  };
  auto request_epi = [&](const UnitLoad& L, float (&ee)[NE]) {
    const int ccol = min(L.col0 + n0 + m, L.n - 1);
#pragma unroll
    for (int reg = 0; reg < NE; ++reg) ee[reg] = L.aux[(long)min(r0 + 4 * g + reg, a.row_end - 1) * L.rstride + ccol];
  };

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

  // stage the row block: thread -> (row = tid / 32, columns 4 (tid % 32) + 128 i), zero beyond the input width.
  // The first 128 columns are requested before the operand packages and written to LDS after them.
  const int sr = tid >> 5, sK = a.in_cols;
  const bool srok = r0 + sr < a.row_end;
  const long sgrow = (long)min(r0 + sr, a.row_end - 1);
  auto stage_load = [&](int c, float (&v)[4]) {
    const float* __restrict__ src = a.in + sgrow * a.ld_in;
    if ((((a.ld_in | sK) & 3) == 0) && c < sK) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = t[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = src[min(c + i, sK - 1)];
    }
    if (BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cc = min(c + i, sK - 1);
        if (a.top_flags & S_JOIN) v[i] += lat[sr * sK + cc];
        if (a.top_flags & S_ACT) {
          const float hv = a.top_h[sgrow * sK + cc];
          v[i] *= fmaf(-hv, hv, 1.f);
        }
      }
    }
  };
  auto stage_store = [&](int c, const float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = srok && c + i < sK;
      if (BWD && ok) a.top_out[sgrow * sK + c + i] = v[i];
      lds[sr * pitch + c + i] = ok ? v[i] : 0.f;   // pitch % 4 == 0
    }
  };
  const int sc0 = 4 * (tid & 31);
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  float tv[4] = {0.f, 0.f, 0.f, 0.f};
  const bool tgt_global = !BWD && a.cost.dY != nullptr;
  if (MODE == CHAIN_FUSED) {   // the image that receives d cost / d output is the A operand of the first backward unit: zero padded
    for (int el = tid; el < R * a.top_pitch; el += NTC) lds[a.top_base + el] = 0.f;
  }
  auto target_load = [&](int c, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a.cost.target[sgrow * a.cost.ldt + min(c + i, a.cost.n - 1)];
  };
  auto target_store = [&](int c, const float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c + i < a.cost.n) tgt[sr * a.cost.n + c + i] = v[i];
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // results of the previous epilogue, stored at the head of the next iteration through a buffer resource whose bounds
  // check drops the lanes and rows that do not exist (no exec-masked branch around a vector-memory instruction)
  float pend[4] = {0.f, 0.f, 0.f, 0.f};
  const float* pend_base = a.ux[0].out;
  int pend_bytes = 0, pend_n = 0;
  unsigned pend_voff = 0x80000000u;
  UnitExec X = a.ux[0];
  UnitLoad Lq = a.ul[0];

  // auto cost on the last forward layer's rows, which are complete in LDS image cost_img: 32 threads per row.
  // to_lds: d cost / d output also goes to image top_img (the source of the fused backward units).
  auto cost_phase = [&](const bool to_lds) {
    const CostArgs& cst = a.cost;
    const int row = tid >> 5, l32 = tid & 31;
    const bool rok = r0 + row < a.row_end;
    const long grow = (long)min(r0 + row, a.row_end - 1);
    const float* yrow = lds + a.cost_base + row * a.cost_pitch;
    const float* trow = tgt + row * cst.n;
    float* drow = rok ? cst.dY + grow * cst.ldy : nullptr;
    float* lrow = (to_lds && rok) ? lds + a.top_base + row * a.top_pitch : nullptr;   // rows beyond the block stay zero
    float lsum = 0.f;
    if (cst.variant == EMX_MEAN_NORM) {
      float ss = 0.f;
      for (int k = l32; k < cst.n; k += 32) {
        float sn, cs, delta, dd;
        cost_elem(cst, yrow, trow, k, sn, cs, delta, dd);
        ss = fmaf(delta, delta, ss);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);   // stays inside the row's 32 lanes
      const float nrm = sqrtf(ss);
      const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
      if (l32 == 0 && rok) lsum = nrm;
      for (int k = l32; k < cst.n; k += 32) {
        float sn, cs, delta, dd;
        cost_elem(cst, yrow, trow, k, sn, cs, delta, dd);
        cost_store(cst, drow, nullptr, k, cst.gnorm * delta * inv * dd, sn, cs);
        cost_store(cst, lrow, nullptr, k, cst.gnorm * delta * inv * dd, sn, cs);
      }
    } else {
      for (int k = l32; k < cst.n; k += 32) {
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
        cost_store(cst, drow, nullptr, k, cst.gnorm * gq, sn, cs);
        cost_store(cst, lrow, nullptr, k, cst.gnorm * gq, sn, cs);
      }
    }
    const float tsum = block_sum<NTC>(lsum, cred);
    if (tid == 0) cst.ws[blockIdx.x] = tsum * cst.gnorm;
  };

  // Iteration ui requests the package of unit ui + NB - 1 into ring slot r and executes unit ui from slot r + 1; the
  // first NB - 1 iterations (ui < 0) only request.  Every iteration issues the SAME sequence of vector-memory
  // instructions (4 stores, NE epilogue operands, 8 pieces, the pieces one per MFMA group), with scalar branches only
  // around code that has none -- that is what lets the compiler's s_waitcnt insertion count exactly, so that the wait in
  // front of an MFMA group is for that group's piece of the OLDEST package and nothing younger.
  auto unit = [&](auto RC, const int ui) {
    constexpr int r = decltype(RC)::value;
    constexpr int rx = (r + 1) % NB;
    const bool run = ui >= 0 && ui < nu;
    const bool req = ui + NB - 1 < nu;
    const int kq = run ? (X.kq_cols & 0xff) : 0, cols = X.kq_cols >> 8;
    const bool mine = n0 < cols;

    {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend_base), 0, pend_bytes, 0x00020000);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pend[reg]), rs, pend_voff + reg * pend_n * 4, 0, 0);
      pend_bytes = 0;
    }
    if (ui == -(NB - 1)) {
      // the row block itself travels ahead of the first package and reaches LDS behind it
      if (sc0 < pitch) stage_load(sc0, sv);
      if (tgt_global && sc0 < a.cost.n) target_load(sc0, tv);
      TSTAMP(1);
    }
    request_epi(Lq, e[r]);
    // beyond the last unit the pieces degenerate to one shared line
    const float* __restrict__ pp = req ? piece_ptr(Lq) : a.mbase;
    const float* src = lds + X.src_off + m * X.src_pitch + 4 * g;
    if (run && (X.flags & U_FIRST)) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (run) TSTAMP(3 + 5 * ui);
    f32x4 av = *reinterpret_cast<const f32x4*>(src);   // A pieces are read one group ahead (any in-image address is readable)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 avn = *reinterpret_cast<const f32x4*>(src + 16 * (q < 7 ? q + 1 : 7));
      if (mine && q < kq) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b[rx][q][i], acc, 0, 0, 0);
      }
      request_piece(Lq, pp, q, b[r][q]);
      __builtin_amdgcn_sched_barrier(0);
      av = avn;
    }
    // the next iteration's records: scalar loads issued here land behind the epilogue (issued at the top of the iteration,
    // their s_waitcnt lgkmcnt(0) -- scalar loads return out of order -- would sit in front of the first ds_read)
    const UnitLoad Ln = a.ul[max(0, min(ui + NB, nu - 1))];
    const UnitExec Xn = a.ux[max(0, min(ui + 1, nu - 1))];
    Lq = Ln;

    if (ui == -(NB - 1)) {
      if (sc0 < pitch) stage_store(sc0, sv);
      for (int c = sc0 + 128; c < pitch; c += 128) {
        float v[4];
        stage_load(c, v);
        stage_store(c, v);
      }
      if (tgt_global) {
        if (sc0 < a.cost.n) target_store(sc0, tv);
        for (int c = sc0 + 128; c < a.cost.n; c += 128) {
          float v[4];
          target_load(c, v);
          target_store(c, v);
        }
      }
      __syncthreads();
      TSTAMP(2);
    }
    if (!run) return;
    TSTAMP(5 + 5 * ui);

    // ---- epilogue (every wave: the whole 128-column pass of the destination image is written, zeros where the
    //      layer is narrower, because it is the next layer's A operand)
    if (X.flags & U_LAST) {
      const int sflags = X.flags >> 8;
      const int col = X.dst_off + n0 + m;
      const bool cok = n0 + m < cols;
      const bool ubwd = MODE == CHAIN_BWD || (MODE == CHAIN_FUSED && (X.flags & U_BWD));
      const int hb1 = X.h_base1;   // 0: activations from the epilogue operands, else from the LDS image at hb1 - 1
      float* dst = lds + X.dst_base + col;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 4 * g + reg;
        float v = acc[reg];
        if (!ubwd) {
          v += e[rx][0];
          if (sflags & S_ACT) v = tanh_f32(v);
        } else {
          if (sflags & S_JOIN) v += lat[row * X.n + min(col, X.n - 1)];
          if (sflags & S_ACT) {
            const float hv = (MODE == CHAIN_FUSED && hb1) ? lds[hb1 - 1 + row * X.h_pitch + col] : e[rx][reg % NE];
            v *= fmaf(-hv, hv, 1.f);
          }
        }
        v = (cok && r0 + row < a.row_end) ? v : 0.f;
        dst[row * X.dst_pitch] = v;
        pend[reg] = v;
      }
      pend_base = X.out + (long)r0 * X.n;
      pend_n = X.n;
      pend_bytes = max(0, min(R, a.row_end - r0)) * X.n * 4;
      pend_voff = cok ? (unsigned)((4 * g * X.n + col) * 4) : 0x80000000u;
    }
    TSTAMP(6 + 5 * ui);
    if (X.flags & U_LAYER_END) __syncthreads();
    if (MODE == CHAIN_FUSED && ui == a.cost_at) {   // the forward part is complete: cost, d cost / d output -> HBM and image top_img
      cost_phase(true);
      __syncthreads();
    }
    TSTAMP(7 + 5 * ui);
    X = Xn;
  };

  // (left by `break` right behind the last unit: an iteration that only requests would still wait for the registers it
  //  overwrites, a full load latency for nothing)
  for (int ui = -(NB - 1);; ui += NB) {
    unit(IC<0>{}, ui);
    if (ui + 1 >= nu) break;
    unit(IC<1>{}, ui + 1);
    if (ui + 2 >= nu) break;
    if constexpr (NB >= 3) {
      unit(IC<2>{}, ui + 2);
      if (ui + 3 >= nu) break;
    }
    if constexpr (NB >= 4) {
      unit(IC<3>{}, ui + 3);
      if (ui + 4 >= nu) break;
    }
  }
  {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend_base), 0, pend_bytes, 0x00020000);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pend[reg]), rs, pend_voff + reg * pend_n * 4, 0, 0);
  }
  TSTAMP(60);
  if (MODE == CHAIN_FWD && a.cost.dY != nullptr) cost_phase(false);
  TSTAMP(61);
#ifdef EMX_CHAIN_TIMING
  __syncthreads();
  if (MODE != CHAIN_BWD && (blockIdx.x == 0 || blockIdx.x == 100)) emx_dbg[(blockIdx.x ? 2048 : 0) + tid] = dbg_lds[tid];
#endif
}

// gather form of mirror.h's layout (the scatter form lives in the Adam kernel)
__global__ __launch_bounds__(256) void mlp_mirror_kernel(const MirrorArgs ma, const float* __restrict__ params, float* __restrict__ mirror) {
  const long el = (long)blockIdx.x * 256 + threadIdx.x;
  if (el >= ma.begin[ma.n_layers]) return;
  int l = 0;
  while (el >= ma.begin[l + 1]) ++l;
  const int K = ma.width[l], N = ma.width[l + 1];
  long rel = el - ma.begin[l];
  const long blk = operand_block(K, N);
  const int dir = rel >= blk;
  rel -= dir * blk;
  const int nout = dir ? K : N, kred = dir ? N : K;
  const int Q = (kred + 15) / 16;
  const int block = (int)(rel >> 8), in = (int)(rel & 255);
  const int t = block / Q, q = block - t * Q;
  const int ln = in >> 2, i = in & 3;
  const int j = 16 * t + (ln & 15), k = 16 * q + 4 * (ln >> 4) + i;
  float v = 0.f;
  if (j < nout && k < kred) v = dir ? params[ma.w_off[l] + (long)j * N + k] : params[ma.w_off[l] + (long)k * N + j];
  mirror[el] = v;
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients: partial[slice][l] (+)= X_l^T delta_l over the slice's rows, X_l = (l == lx0 ? x_in : h_{l-1}).
// grid = (blocks of all layers) x n_slices; one workgroup = one 128x128 output block of one layer (each wave a
// 64x64 quadrant = 4x4 MFMA tiles), rows [row0 + slice*rows_per_slice, ...).  db is produced by the blocks of
// the first tile row.  Partials are stored densely per slice in the flat-parameter layout (w_off/b_off).
// ---------------------------------------------------------------------------------------------------------
struct WgradDev {
  MlpDev d;
  int l0, l1;
  const float* xin;
  int ld_xin;
  long w_off[MAXL], b_off[MAXL];
  int tile_begin[MAXL + 1];  // prefix sum of 64x64 output blocks per layer (relative to l0)
  float* partial;            // [n_slices][n_params]
  long n_params;
  int row0, row_end, rows_per_slice;
  int has_fin;               // the last blockIdx.x column is not a dW block: its y = 0 workgroup runs the step's scalar epilogue
  FinishArgs fin;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef EMX_CHAIN_TIMING   // per-workgroup timeline of the weight-gradient kernel (tools/abl_wgrad.py)
__device__ long long emx_dbg_wg[3 * 1024 + 64];
#define WSTAMP(k) do { if (threadIdx.x == 0) { const int bid_ = blockIdx.y * gridDim.x + blockIdx.x; \
    if (bid_ < 1024 && ((k) == 0 || (k) == 9)) emx_dbg_wg[64 + 3 * bid_ + ((k) ? 1 : 0)] = wall_clock64(); \
    if (bid_ == 0) emx_dbg_wg[k] = clock64(); } } while (0)
#else
#define WSTAMP(k)
#endif

// three waves per SIMD: 640 workgroups x 4 waves must all be resident at once (2.5 per SIMD); the scalar epilogue that rides
// along would otherwise raise the allocation to 198 registers = two waves (16 -> 22 us)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_wgrad_kernel(const WgradDev a) {
  // One workgroup = one 64x64 block of one layer's dW over one row slice (128 rows); each wave a 32x32 quadrant = 2x2
  // MFMA tiles.  The MFMA's m/n indices are free labels: lane slot m of tile u stands for feature 2m + u (n, v likewise),
  // so ONE 8-byte load per operand and 4-row step feeds both tiles of that operand, and the two tiles v = 0, 1 of a slot
  // pair form a float2 of adjacent output columns for the store.
  // (Decomposition: parallelism = blocks x slices, partial traffic = slices x parameters.  The kernel is bound by the
  //  latency of its operand loads -- the activations and deltas were written by the previous kernels on other XCDs --
  //  so it wants several workgroups per CU: 64x64 blocks x 32 slices = 640 workgroups, 128 MFMAs per wave; measured
  //  15.8 us against 20.1 us for 64x64 x 16 slices (320 workgroups) and 17.5 us for 128x128 x 32 slices (192 workgroups).)
  __shared__ float fin_red[EMX_MAX_SUM_JOBS][FIN_NT / 64];
  __shared__ float fin_tl[64];
  if (a.has_fin && blockIdx.x == gridDim.x - 1) {
    // The step's scalar epilogue (cost sums, history, counters) rides along as one extra workgroup: its inputs -- the
    // cost shares of the forward launch, the tile sums of the all-pairs launch, the regulariser shares the previous
    // optimizer launch left for the current weights -- are complete before this launch starts, and a launch of its own
    // costs ~6.5 us of a ~95 us step.
    if (blockIdx.y == 0) finish_step_body<emx::FQ_EMBEDDED>(a.fin, fin_red, fin_tl);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  WSTAMP(0);
  int l = a.l0;
  while (l + 1 < a.l1 && (int)blockIdx.x >= a.tile_begin[l + 1 - a.l0]) ++l;
  const int tb = blockIdx.x - a.tile_begin[l - a.l0];
  const int Kin = a.d.width[l], N = a.d.width[l + 1];
  const int nbn = (N + 63) >> 6;
  const int bi = tb / nbn, bj = tb - bi * nbn;
  const int i0 = bi * 64 + (wave >> 1) * 32;   // input-feature offset of this wave's quadrant
  const int j0 = bj * 64 + (wave & 1) * 32;    // output-feature offset
  const float* __restrict__ X = (l == a.l0) ? a.xin : a.d.h[l - 1];
  const int ldx = (l == a.l0) ? a.ld_xin : Kin;
  const float* __restrict__ Dl = a.d.delta[l];
  const int rs = a.row0 + blockIdx.y * a.rows_per_slice;
  const int re = min(rs + a.rows_per_slice, a.row_end);
  float* __restrict__ out = a.partial + (long)blockIdx.y * a.n_params;

  f32x4 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 bsum = {0.f, 0.f};
  const bool do_bias = (bi == 0) && ((wave >> 1) == 0);
  const bool quad_active = i0 < Kin && j0 < N;   // wave-uniform

  // 8-byte operand loads need 2 | width and an even row stride; otherwise two clamped scalar loads.  The decision is made
  // ONCE around the whole row loop: decided per load, the two variants write the same registers from different branches
  // and the compiler fences every load with a wait for the previous ones.
  const bool vx = ((ldx & 1) == 0) && ((Kin & 1) == 0), vd = (N & 1) == 0;
  const int cx = i0 + 2 * m, cd = j0 + 2 * m;
  auto ldvec = [&](const float* __restrict__ base, int ld, int width, int col, int row) -> f32x2 {
    return *reinterpret_cast<const f32x2*>(base + (long)row * ld + min(col, width - 2));   // clamped: out-of-range slots are discarded at the store
  };
  auto ldsca = [&](const float* __restrict__ base, int ld, int width, int col, int row) -> f32x2 {
    const float* p = base + (long)row * ld;
    return f32x2{p[min(col, width - 1)], p[min(col + 1, width - 1)]};
  };
  constexpr int US = 8;   // 4-row steps per batch: 32 MFMAs behind 16 operand loads
  if (quad_active) {
    // Two register sets, two batches per loop iteration: set A is the only loop-carried one and is reloaded (for the
    // batch after next) right behind the MFMAs that consumed it, set B is loaded and consumed inside the iteration.  All
    // loads are unconditional with clamped rows; rows beyond the slice are zeroed in the delta operand.
    f32x2 xA[US], dA[US], xB[US], dB[US];
#define WG_LOAD(LD, xo, dout, r)                                                              \
  _Pragma("unroll") for (int s2 = 0; s2 < US; ++s2) {                                        \
    const int row_ = (r) + 4 * s2 + g;                                                        \
    const int rowc_ = min(row_, re - 1);                                                      \
    xo[s2] = LD(X, ldx, Kin, cx, rowc_);                                                      \
    dout[s2] = LD(Dl, N, N, cd, rowc_) * (row_ < re ? 1.f : 0.f);                             \
  }
#define WG_MMA(xi, di)                                                                        \
  _Pragma("unroll") for (int s2 = 0; s2 < US; ++s2) {                                        \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                            \
      _Pragma("unroll") for (int v = 0; v < 2; ++v)                                          \
        acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi[s2][u], di[s2][v], acc[u][v], 0, 0, 0); \
    bsum += di[s2];                                                                           \
  }
#define WG_ROWS(LD)                                   \
  WG_LOAD(LD, xA, dA, rs)                             \
  WSTAMP(1);                                          \
  for (int r = rs; r < re; r += 8 * US) {             \
    WG_LOAD(LD, xB, dB, r + 4 * US)                   \
    WG_MMA(xA, dA)                                    \
    WG_LOAD(LD, xA, dA, r + 8 * US)                   \
    WG_MMA(xB, dB)                                    \
    if (r == rs) WSTAMP(2);                           \
  }                                                   \
  WSTAMP(3);
    if (vx && vd) {
      WG_ROWS(ldvec)
    } else {
      WG_ROWS(ldsca)
    }
#undef WG_ROWS
#undef WG_LOAD
#undef WG_MMA
    // store: tile (u, v), D[row slot 4g + reg][col slot m] -> dW[i0 + 2(4g + reg) + u][j0 + 2m + v]
    const bool vst = (N & 1) == 0 && ((a.w_off[l] & 1) == 0);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ii = i0 + 2 * (4 * g + reg) + u, jj = j0 + 2 * m;
        if (ii < Kin && jj < N) {
          float* q = out + a.w_off[l] + (long)ii * N + jj;
          if (vst) {
            *reinterpret_cast<f32x2*>(q) = f32x2{acc[u][0][reg], acc[u][1][reg]};
          } else {
            q[0] = acc[u][0][reg];
            if (jj + 1 < N) q[1] = acc[u][1][reg];
          }
        }
      }
  }
  if (do_bias && j0 < N) {
    // column sums: lane (m, g) holds the sums of columns j0 + 2m, + 1 over the rows of its lane group
    f32x2 sfull = bsum;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      float t = sfull[v];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      sfull[v] = t;
    }
    if (g == 0) {
#pragma unroll
      for (int v = 0; v < 2; ++v)
        if (cd + v < N) out[a.b_off[l] + cd + v] = sfull[v];
    }
  }
  WSTAMP(9);
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

int fill_dev(const emx_mlp_desc* desc, float* params, float* const* h, float* const* delta, MlpDev& d, const char* who) {
  EMX_REQUIRE(desc && params, "%s: null descriptor or parameter buffer", who);
  EMX_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= MAXL, "%s: %d layers not supported (1..%d)", who, desc->n_layers, MAXL);
  d.n_layers = desc->n_layers;
  for (int l = 0; l <= desc->n_layers; ++l) {
    EMX_REQUIRE(desc->width[l] >= 1, "%s: layer width must be >= 1", who);
    d.width[l] = desc->width[l];
  }
  for (int l = 0; l < desc->n_layers; ++l) {
    d.act[l] = desc->act[l];
    d.W[l] = params + desc->w_off[l];
    d.b[l] = params + desc->b_off[l];
    d.h[l] = h ? h[l] : nullptr;
    d.delta[l] = delta ? delta[l] : nullptr;
  }
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

int check_desc(const emx_mlp_desc* desc, const char* who) {
  EMX_REQUIRE(desc, "%s: null descriptor", who);
  EMX_REQUIRE(desc->n_layers >= 1 && desc->n_layers <= MAXL, "%s: %d layers not supported (1..%d)", who, desc->n_layers, MAXL);
  for (int l = 0; l <= desc->n_layers; ++l) EMX_REQUIRE(desc->width[l] >= 1, "%s: layer width must be >= 1", who);
  return EMX_OK;
}

// float offset of layer l's operand blocks in the operand store (forward block first, backward block second)
long mirror_begin(const emx_mlp_desc* desc, int l) {
  long o = 0;
  for (int i = 0; i < l; ++i) o += 2 * operand_block(desc->width[i], desc->width[i + 1]);
  return o;
}

// Appends the units of one GEMM: `kred` reduction indices, `nout` output columns, operand block at float offset `op`;
// `aux` with row stride `rstride` (0: a bias row) feeds the epilogue; LDS images src -> dst, himg + 1 = image of the
// activations for the derivative (0: they come through aux).
int add_units(ChainArgs& a, ImgGeom& G, long op, int kred, int nout, const float* aux, int rstride, float* out, int sflags, int uflags,
              int src_img, int dst_img, int himg1, const char* who) {
  if (std::max(std::max(src_img, dst_img), himg1 - 1) >= MAXIMG) return emx::fail(EMX_EUNSUPPORTED, "%s: more than %d LDS images", who, MAXIMG);
  G.need(src_img, kred);
  G.need(dst_img, nout);
  if (himg1) G.need(himg1 - 1, nout);
  const int Q = (kred + 15) / 16, T = (nout + 15) / 16;
  const int P = (nout + 127) / 128, Cn = (kred + 127) / 128;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < Cn; ++c) {
      if (a.nu >= MAXU) return emx::fail(EMX_EUNSUPPORTED, "%s: layer range needs more than %d 128x128 units", who, MAXU);
      const long off = op + ((long)(8 * p) * Q + 8 * c) * 256;
      if (off + 8L * Q * 256 > 0x7fffffffL) return emx::fail(EMX_EUNSUPPORTED, "%s: operand store too large", who);
      const int kq = std::min(8, Q - 8 * c);
      a.ul[a.nu] = UnitLoad{(int)off, Q, kq | (std::min(8, T - 8 * p) << 8), 128 * p, nout, rstride, aux};
      a.ux[a.nu] = UnitExec{kq | (std::min(128, nout - 128 * p) << 8), 128 * c, 128 * p,
                            (c == 0 ? U_FIRST : 0) | (c == Cn - 1 ? U_LAST : 0) | (c == Cn - 1 && p == P - 1 ? U_LAYER_END : 0) | uflags | (sflags << 8), out,
                            nout, src_img, dst_img, 0, himg1, 0};   // image ids until finish_images() turns them into offsets and pitches
      ++a.nu;
    }
  return EMX_OK;
}

// Lays the images out and turns the image ids of the unit records into LDS offsets and row pitches.
void finish_images(ChainArgs& a, ImgGeom& G) {
  G.layout();
  for (int u = 0; u < std::max(a.nu, 1); ++u) {
    UnitExec& x = a.ux[u];
    const int si = x.src_pitch, di = x.dst_base, hi1 = x.h_base1;
    x.src_off += G.base[si];
    x.src_pitch = G.pitch[si];
    x.dst_base = G.base[di];
    x.dst_pitch = G.pitch[di];
    x.h_base1 = hi1 ? G.base[hi1 - 1] + 1 : 0;
    x.h_pitch = hi1 ? G.pitch[hi1 - 1] : 0;
  }
  a.pitch = G.pitch[0];
  a.img_floats = G.total;
}

}  // namespace

extern "C" {

size_t emx_mlp_mirror_floats(const emx_mlp_desc* desc) {
  if (!desc || desc->n_layers < 1 || desc->n_layers > MAXL) return 0;
  return (size_t)mirror_begin(desc, desc->n_layers);
}

int emx_mlp_mirror(const emx_mlp_desc* desc, const float* params, float* mirror, emx_stream_t stream) {
  int rc = check_desc(desc, "emx_mlp_mirror");
  if (rc) return rc;
  EMX_REQUIRE(params && mirror, "emx_mlp_mirror: null pointer argument");
  MirrorArgs ma{};
  fill_mirror_args(desc, ma);
  const long n = ma.begin[desc->n_layers];
  hipLaunchKernelGGL(mlp_mirror_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ma, params, mirror);
  return emx::check_launch("mlp_mirror_kernel");
}

int emx_mlp_row_blocks(int n_rows) { return std::max((n_rows + R - 1) / R, 1); }

int emx_mlp_forward_cost(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                         int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, emx_stream_t stream) {
  int rc = check_desc(desc, "emx_mlp_forward");
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && h && in, "emx_mlp_forward: null pointer argument");
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "emx_mlp_forward: bad layer range");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_in >= desc->width[layer_begin], "emx_mlp_forward: bad rows / stride");
  ChainArgs a{};
  ImgGeom G;
  for (int l = layer_begin; l < layer_end; ++l) {
    EMX_REQUIRE(h[l], "emx_mlp_forward: activation buffer of layer %d is null", l);
    const int st = l - layer_begin;
    rc = add_units(a, G, mirror_begin(desc, l), desc->width[l], desc->width[l + 1], params + desc->b_off[l], 0, h[l], desc->act[l] ? S_ACT : 0, 0,
                   st & 1, (st + 1) & 1, 0, "emx_mlp_forward");
    if (rc) return rc;
  }
  G.need(1, 1);
  finish_images(a, G);
  const int cimg = (layer_end - layer_begin) & 1;
  a.cost_base = G.base[cimg];
  a.cost_pitch = G.pitch[cimg];
  if (cost) {
    const int wout = desc->width[layer_end];
    EMX_REQUIRE(cost->target && cost->dY && cost->ws && cost->n > 0 && cost->ldt >= cost->n && cost->B > 0, "emx_mlp_forward: incomplete cost description");
    EMX_REQUIRE(cost->variant >= EMX_MEAN_ABS && cost->variant <= EMX_MEAN_NORM, "emx_mlp_forward: cost variant %d not available", cost->variant);
    EMX_REQUIRE(cost->off_sin >= 0 && cost->off_sin + cost->n <= wout &&
                    (!cost->periodic_output || (cost->off_cos >= 0 && cost->off_cos + cost->n <= wout && std::isfinite(cost->periodicity))),
                "emx_mlp_forward: cost column window outside the output row");
    a.cost.target = cost->target;
    a.cost.dY = cost->dY;
    a.cost.ws = cost->ws;
    fill_cost_scalars(a.cost, cost->B, cost->n, wout, cost->ldt, cost->off_sin, cost->off_cos, cost->periodic_output, cost->periodicity,
                      cost->variant, cost->coef);
  }
  a.mbase = mirror;
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.row0 = row_begin;
  a.row_end = row_end;
  const size_t lds = ((size_t)a.img_floats + (cost ? (size_t)R * cost->n : 0)) * sizeof(float);
  rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<CHAIN_FWD>), lds, "emx_mlp_forward");
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  hipLaunchKernelGGL(chain_kernel<CHAIN_FWD>, dim3(grid), dim3(NTC), lds, (hipStream_t)stream, a);
  return emx::check_launch("mlp chain kernel (forward)");
}

int emx_mlp_forward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                    int ld_in, int row_begin, int row_end, float* const* h, emx_stream_t stream) {
  return emx_mlp_forward_cost(desc, params, mirror, layer_begin, layer_end, in, ld_in, row_begin, row_end, h, nullptr, stream);
}

int emx_mlp_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* dtop,
                     int ld_top, const float* dadd, int layer_add, const emx_latent_grad* latent, float* dbelow, int row_begin,
                     int row_end, float* const* h, float* const* delta, emx_stream_t stream) {
  int rc = check_desc(desc, "emx_mlp_backward");
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && dtop && h && delta, "emx_mlp_backward: null pointer argument");
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "emx_mlp_backward: bad layer range");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_top >= desc->width[layer_end], "emx_mlp_backward: bad rows / stride");
  EMX_REQUIRE((!dadd && !latent) || (layer_add >= layer_begin && layer_add < layer_end), "emx_mlp_backward: layer_add outside the layer range");
  ChainArgs a{};
  ImgGeom G;
  const bool join = dadd || latent;
  if (latent) {
    a.lg = *latent;
    EMX_REQUIRE(a.lg.partials && a.lg.Z && a.lg.n_slots > 0, "emx_mlp_backward: incomplete latent gradient descriptor");
  }
  if (join) a.join_n = desc->width[layer_add + 1];
  a.dadd = dadd;
  for (int l = layer_begin; l < layer_end; ++l) EMX_REQUIRE(h[l] && delta[l], "emx_mlp_backward: buffer of layer %d is null", l);
  auto sflags = [&](int l) { return (desc->act[l] ? S_ACT : 0) | ((join && l == layer_add) ? S_JOIN : 0); };
  // top of the range: delta_top from dtop
  const int top = layer_end - 1;
  a.top_h = h[top];
  a.top_out = delta[top];
  a.top_flags = sflags(top);
  // GEMM of layer l: delta_l W_l^T -> dh_{l-1} -> delta_{l-1} (or dbelow at the bottom of the range)
  const int lo = dbelow ? layer_begin : layer_begin + 1;
  for (int l = top; l >= lo; --l) {
    const int Kin = desc->width[l], N = desc->width[l + 1];
    const long op = mirror_begin(desc, l) + operand_block(Kin, N);
    const int st = top - l;
    if (l > layer_begin)
      rc = add_units(a, G, op, N, Kin, desc->act[l - 1] ? h[l - 1] : delta[l - 1], Kin, delta[l - 1], sflags(l - 1), U_BWD, st & 1, (st + 1) & 1, 0,
                     "emx_mlp_backward");
    else
      rc = add_units(a, G, op, N, Kin, dbelow, Kin, dbelow, 0, U_BWD, st & 1, (st + 1) & 1, 0, "emx_mlp_backward");
    if (rc) return rc;
  }
  if (a.nu == 0) {   // a one-layer range without dbelow: only delta_top is formed; the kernel still wants one readable record
    a.ul[0] = UnitLoad{0, 1, 1 | (1 << 8), 0, 1, 0, mirror};
    a.ux[0] = UnitExec{1 | (1 << 8), 0, 0, 0, delta[top], 1, 0, 1, 0, 0, 0};
  }
  G.need(0, desc->width[layer_end]);   // the row block of dtop
  G.need(1, 1);
  finish_images(a, G);
  a.mbase = mirror;
  a.in = dtop;
  a.ld_in = ld_top;
  a.in_cols = desc->width[layer_end];
  a.row0 = row_begin;
  a.row_end = row_end;
  const size_t lds = ((size_t)a.img_floats + (size_t)R * a.join_n) * sizeof(float);
  rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<CHAIN_BWD>), lds, "emx_mlp_backward");
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  hipLaunchKernelGGL(chain_kernel<CHAIN_BWD>, dim3(grid), dim3(NTC), lds, (hipStream_t)stream, a);
  return emx::check_launch("mlp chain kernel (backward)");
}

int emx_mlp_forward_backward(const emx_mlp_desc* desc, float* params, const float* mirror, int layer_begin, int layer_end, const float* in,
                             int ld_in, int row_begin, int row_end, float* const* h, const emx_auto_cost* cost, int bwd_begin,
                             float* const* delta, float* dbelow, emx_stream_t stream) {
  const char* who = "emx_mlp_forward_backward";
  int rc = check_desc(desc, who);
  if (rc) return rc;
  EMX_REQUIRE(params && mirror && h && in && cost && delta, "%s: null pointer argument", who);
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "%s: bad layer range", who);
  EMX_REQUIRE(layer_begin <= bwd_begin && bwd_begin < layer_end, "%s: backward range [%d, %d) outside the forward range", who, bwd_begin, layer_end);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_in >= desc->width[layer_begin], "%s: bad rows / stride", who);
  const int top = layer_end - 1, wout = desc->width[layer_end];
  EMX_REQUIRE(!desc->act[top], "%s: the fused launch needs a linear top layer (d cost / d output is its delta)", who);
  EMX_REQUIRE(cost->target && cost->ws && cost->n > 0 && cost->ldt >= cost->n && cost->B > 0, "%s: incomplete cost description", who);
  EMX_REQUIRE(cost->variant >= EMX_MEAN_ABS && cost->variant <= EMX_MEAN_NORM, "%s: cost variant %d not available", who, cost->variant);
  EMX_REQUIRE(cost->off_sin >= 0 && cost->off_sin + cost->n <= wout &&
                  (!cost->periodic_output || (cost->off_cos >= 0 && cost->off_cos + cost->n <= wout && std::isfinite(cost->periodicity))),
              "%s: cost column window outside the output row", who);
  EMX_REQUIRE(cost->dY == delta[top] || cost->dY == nullptr, "%s: d cost / d output is delta[layer_end - 1]; pass that buffer (or NULL) as cost->dY", who);
  ChainArgs a{};
  ImgGeom G;
  const int nf = layer_end - layer_begin;           // forward layers: images 0 .. nf (image i = input of forward layer i)
  for (int l = layer_begin; l < layer_end; ++l) {
    EMX_REQUIRE(h[l], "%s: activation buffer of layer %d is null", who, l);
    const int st = l - layer_begin;
    rc = add_units(a, G, mirror_begin(desc, l), desc->width[l], desc->width[l + 1], params + desc->b_off[l], 0, h[l], desc->act[l] ? S_ACT : 0, 0, st,
                   st + 1, 0, who);
    if (rc) return rc;
  }
  a.cost_at = a.nu - 1;
  const int cost_img = nf, top_img = nf + 1;         // d cost / d output; the backward units ping-pong between nf + 1 and nf + 2
  EMX_REQUIRE(nf + 3 <= MAXIMG, "%s: too many layers for one launch", who);
  G.need(top_img, wout);
  G.need(nf + 2, 1);
  for (int l = top; l >= bwd_begin; --l) {
    EMX_REQUIRE(delta[l], "%s: delta buffer of layer %d is null", who, l);
    const int Kin = desc->width[l], N = desc->width[l + 1];
    const long op = mirror_begin(desc, l) + operand_block(Kin, N);
    const int st = top - l, src = nf + 1 + (st & 1), dst = nf + 1 + ((st + 1) & 1);
    if (l > bwd_begin) {
      // delta_{l-1} = (delta_l W_l^T) * act'(h_{l-1}); h_{l-1} is still in LDS image (l - 1 - layer_begin) + 1
      // (the epilogue operand slot of these units is not used: any readable row, here the head of the operand store)
      rc = add_units(a, G, op, N, Kin, mirror, 0, delta[l - 1], desc->act[l - 1] ? S_ACT : 0, U_BWD, src, dst, (l - layer_begin) + 1, who);
    } else {
      if (!dbelow) break;
      rc = add_units(a, G, op, N, Kin, mirror, 0, dbelow, 0, U_BWD, src, dst, 0, who);
    }
    if (rc) return rc;
  }
  a.cost.target = cost->target;
  a.cost.dY = delta[top];
  a.cost.ws = cost->ws;
  fill_cost_scalars(a.cost, cost->B, cost->n, wout, cost->ldt, cost->off_sin, cost->off_cos, cost->periodic_output, cost->periodicity,
                    cost->variant, cost->coef);
  finish_images(a, G);
  a.cost_base = G.base[cost_img];
  a.cost_pitch = G.pitch[cost_img];
  a.top_base = G.base[top_img];
  a.top_pitch = G.pitch[top_img];
  a.mbase = mirror;
  a.in = in;
  a.ld_in = ld_in;
  a.in_cols = desc->width[layer_begin];
  a.row0 = row_begin;
  a.row_end = row_end;
  const size_t lds = ((size_t)a.img_floats + (size_t)R * cost->n) * sizeof(float);
  rc = set_lds(reinterpret_cast<const void*>(&chain_kernel<CHAIN_FUSED>), lds, who);
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  hipLaunchKernelGGL(chain_kernel<CHAIN_FUSED>, dim3(grid), dim3(NTC), lds, (hipStream_t)stream, a);
  return emx::check_launch("mlp chain kernel (forward + cost + backward)");
}

// rows per slice: EMX_WGRAD_SLICE_ROWS, doubled until at most EMX_WGRAD_MAX_SLICES slices remain -- every slice costs a
// write and a read of the whole gradient (partials), so their number must not grow with the batch
static int wgrad_slice_rows(int n_rows) {
  int r = EMX_WGRAD_SLICE_ROWS;
  while ((long)r * EMX_WGRAD_MAX_SLICES < n_rows) r *= 2;
  // small batches: shorter slices, so that a few hundred rows still spread over eight slices' worth of workgroups (the
  // reduction over a slice's rows is a serial chain of 4-row steps per workgroup)
  static const int min_rows = getenv("EMX_WGRAD_MIN_SLICE") ? std::max(4, atoi(getenv("EMX_WGRAD_MIN_SLICE"))) : 32;
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

int emx_mlp_wgrad(const emx_mlp_desc* desc, float* params, int64_t n_params, int layer_begin, int layer_end, const float* xin,
                  int ld_xin, int row_begin, int row_end, float* const* h, float* const* delta, float* grad, float* ws,
                  size_t ws_floats, const emx_finish_desc* finish, emx_stream_t stream) {
  WgradDev a;
  a.has_fin = 0;
  if (finish) {
    int rcf = fill_finish_args(a.fin, finish->jobs, finish->n_jobs, finish->terms, finish->n_terms, finish->history, finish->term_history,
                               finish->capacity, finish->hstep, finish->cursor, finish->counter2, "emx_mlp_wgrad (finish)");
    if (rcf) return rcf;
    a.has_fin = 1;
  }
  int rc = fill_dev(desc, params, h, delta, a.d, "emx_mlp_wgrad");
  if (rc) return rc;
  EMX_REQUIRE(xin && h && delta && ws, "emx_mlp_wgrad: null pointer argument");
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "emx_mlp_wgrad: bad layer range");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_xin >= desc->width[layer_begin], "emx_mlp_wgrad: bad rows / stride");
  const int nrows = row_end - row_begin;
  if (ws_floats < emx_mlp_wgrad_ws_floats(n_params, nrows))
    return emx::fail(EMX_EWORKSPACE, "emx_mlp_wgrad: workspace %zu < %zu floats", ws_floats, emx_mlp_wgrad_ws_floats(n_params, nrows));
  a.l0 = layer_begin;
  a.l1 = layer_end;
  a.xin = xin;
  a.ld_xin = ld_xin;
  int tiles = 0;
  long lo = -1, hi = 0;
  for (int l = layer_begin; l < layer_end; ++l) {
    EMX_REQUIRE(delta[l] && (l == layer_begin || h[l - 1]), "emx_mlp_wgrad: buffer of layer %d is null", l);
    a.w_off[l] = desc->w_off[l];
    a.b_off[l] = desc->b_off[l];
    a.tile_begin[l - layer_begin] = tiles;
    tiles += ((desc->width[l] + 63) / 64) * ((desc->width[l + 1] + 63) / 64);
    const long wl = desc->w_off[l], wh = desc->b_off[l] + desc->width[l + 1];
    lo = (lo < 0) ? wl : std::min(lo, wl);
    hi = std::max(hi, wh);
  }
  a.tile_begin[layer_end - layer_begin] = tiles;
  a.partial = ws;
  a.n_params = n_params;
  a.row0 = row_begin;
  a.row_end = row_end;
  a.rows_per_slice = wgrad_slice_rows(nrows);
  const int slices = emx_mlp_wgrad_slices(nrows);
  hipStream_t st = (hipStream_t)stream;
  static_assert(FIN_NT == NT, "the scalar epilogue shares the weight-gradient kernel's workgroup size");
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(tiles + a.has_fin, slices), dim3(NT), 0, st, a);
  rc = emx::check_launch("mlp_wgrad_kernel");
  if (rc) return rc;
  if (!grad) return EMX_OK;   // the caller sums the slices itself (emx_adam_tick)
  const long n = hi - lo;
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, ws, (long)n_params, slices, lo, hi, grad);
  return emx::check_launch("mlp_wgrad_reduce_kernel");
}

}  // extern "C"

#ifdef EMX_CHAIN_TIMING
extern "C" int emx_debug_read(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg), sizeof(emx_dbg)); }
#endif

#ifdef EMX_CHAIN_TIMING
extern "C" int emx_debug_read_wg(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg_wg), sizeof(emx_dbg_wg)); }
#endif
