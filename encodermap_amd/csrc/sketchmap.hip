// K1/K2: fused sketch-map sigmoid pairwise-distance cost, forward + latent gradient (gfx950).
//
// Replaces the TensorFlow op chain of
//   pairwise_dist / pairwise_dist_periodic / sigmoid   encodermap/misc/distances.py:66-88,144-176,179-255
//   sigmoid_loss_func                                  encodermap/loss_functions/loss_functions.py:351-367
// which materialises >= 10 [B,B] (periodic: [B,B,D]) temporaries, by ONE tiled all-pairs kernel plus a
// small fixed-order reduction kernel.
//
// Layout / algorithm
//   * The B x B pair matrix is cut into TM x TN tiles, one 256-thread workgroup (4 waves) per tile.  With the
//     full batch owned by this rank only tiles with tj >= ti are launched (i<->j symmetry): an off-diagonal
//     tile counts twice in the loss and produces both the row sums (gradient of rows in ti) and the column
//     sums (gradient of rows in tj).  With a data-parallel row block the rectangle rows x [0,B) is launched
//     and only row sums are produced.
//   * K loop: 32-wide chunks of the feature axis are staged global -> registers -> LDS (128-B coalesced
//     row segments, ds_write_b128), the next chunk's global loads are in flight while the current one is
//     consumed.  LDS rows are padded to 36 floats so that the interleaved row assignment (row = t + 16 r)
//     reads conflict-free ds_read_b128 fragments.  Each thread owns an RM x RN register micro-tile of
//     squared-distance accumulators (8x8 for 128-tiles, 4x4 for 64-tiles): per pair-element one v_sub +
//     one v_fma (Euclidean) or sub, sub|abs|, min, fma (periodic).  FP32 VALU bound, not HBM bound:
//     algorithmic HBM traffic is 4*B*(D+2L) bytes per launch.
//   * Epilogue (registers): d^2 -> sigmoid complements E = (1 + c (d/sig)^a)^(-b/a), f = E_l - E_h
//     (= S_h - S_l without the 1-x cancellation), loss += f^2, g = f * S_l'(d_l)/d_l, row/column gradient
//     sums.  The default parameter family (a_h=12,b_h=6,a_l=2,b_l=6) needs one v_rsq and one v_rcp per pair;
//     any other parameters take the exp2/log2 path.
//   * Per-tile partial sums go to a workspace slot that exactly one tile writes; the finish kernel adds the
//     slots in index order (no float atomics -> bitwise reproducible), applies the scale and the optional
//     center cost (loss_functions.py:422-449) and writes dZ.
//
// Deviations from the reference arithmetic, all below fp32 resolution of the loss: distances are formed by
// direct differences instead of the Gram identity (no cancellation); the periodic path omits the +1e-12
// regularisers of distances.py:169-175 (they only keep TF's sqrt gradient finite; X carries no gradient here).
#include "emx_common.h"
#include "sk_tile.h"
#include "split.h"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace emx;
using namespace emx::sk;

constexpr int NT = 256;
constexpr int KSPLIT_MAX_NB = 10;      // batches of up to 640 rows may split the reduction of a tile over several workgroups
constexpr int KSPLIT_BLOCKS = 512;     // ... over at most this many in total



// (leading scalars: tile decode and the first row loads, preloaded into scalar registers at wave launch; 56 bytes, the struct follows)
constexpr int SK_LT_ANY = 9;   // LT of sk_tile_kernel: any latent width, walked in chunks of eight columns
template <int TM, int TN, int LT, int PMODE, bool FASTSIG>
__global__ __launch_bounds__(NT) void sk_tile_kernel(const float* p_X, int p_ldx, int p_D, int p_B, int p_row0, int p_row_end, int p_nbc, int p_symm,
                                                     int p_xcd_nbg, int p_vec4, int p_deal_world, int p_deal_rank, int p_pad, const SkArgs a) {
  constexpr int RM = TM / 16, RN = TN / 16;
  constexpr int LMAX = LT == SK_LT_ANY ? 8 : (LT ? LT : 4);
  constexpr int STAGE = (TM + TN) * LS;
  constexpr int PART = 16 * (TM + TN) * LMAX;
  constexpr int SMEM = (STAGE > PART ? STAGE : PART) + 8;
  __shared__ __align__(16) float smem[SMEM];
  float* Xi_s = smem;
  float* Xj_s = smem + TM * LS;

  const int L = LT == 2 ? 2 : a.L;   // LT: 2 = exactly two latent columns (compile time); 0 / 8 = up to 4 / 8 columns at run time, zero padded
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;

  (void)p_pad;
  int ti, tj, tslot = blockIdx.x;
  if (p_xcd_nbg > 0) {   // XCD-aware order (xcd_tile): every XCD's L2 sees half of the row panels instead of all of them
    if (!xcd_tile(blockIdx.x, p_xcd_nbg, p_nbc, ti, tj, tslot)) return;
  } else if (p_symm) {
    tri_decode(blockIdx.x, p_nbc, ti, tj);
  } else {
    ti = blockIdx.x / p_nbc;
    tj = blockIdx.x - ti * p_nbc;
  }
  if (p_deal_world > 1 && tslot % p_deal_world != p_deal_rank) {   // (dealt_away with the preloaded copies)
    if (threadIdx.x == 0) a.tile_loss[tslot] = 0.f;
    return;
  }
  const int i0 = p_row0 + ti * TM;
  const int j0 = tj * TN;
  const bool offdiag = p_symm && (ti != tj);

  float acc[RM][RN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RN; ++c) acc[r][c] = 0.f;

  // ---- K loop -------------------------------------------------------------------------------
  const int kq = tid & 7, rb = tid >> 3;
  constexpr int QI = TM / 32, QJ = TN / 32;
  float4 pi[QI], pj[QJ];
  const bool vec4 = p_vec4 != 0;
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < QI; ++q) {
      int row = i0 + rb + 32 * q;
      pi[q] = load_row4(p_X, row, p_ldx, k0 + 4 * kq, p_D, row < p_row_end, vec4);
    }
#pragma unroll
    for (int q = 0; q < QJ; ++q) {
      int row = j0 + rb + 32 * q;
      pj[q] = load_row4(p_X, row, p_ldx, k0 + 4 * kq, p_D, row < p_B, vec4);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < p_D; k0 += KC) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QI; ++q)
      *reinterpret_cast<float4*>(&Xi_s[(rb + 32 * q) * LS + 4 * kq]) = PMODE == 2 ? phase4(pi[q], a.rp_hi, a.rp_lo) : pi[q];
#pragma unroll
    for (int q = 0; q < QJ; ++q)
      *reinterpret_cast<float4*>(&Xj_s[(rb + 32 * q) * LS + 4 * kq]) = PMODE == 2 ? phase4(pj[q], a.rp_hi, a.rp_lo) : pj[q];
    __syncthreads();
    if (k0 + KC < p_D) gload(k0 + KC);
#pragma unroll 2
    for (int kk = 0; kk < KC; kk += 4) {
      float4 av[RM];
#pragma unroll
      for (int r = 0; r < RM; ++r) av[r] = *reinterpret_cast<const float4*>(&Xi_s[(ty + 16 * r) * LS + kk]);
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        const float4 bv = *reinterpret_cast<const float4*>(&Xj_s[(tx + 16 * c) * LS + kk]);
#pragma unroll
        for (int r = 0; r < RM; ++r) acc[r][c] = acc4<PMODE>(av[r], bv, acc[r][c], a.period);
      }
    }
  }
  __syncthreads();  // staging LDS is reused for the partial sums below
  if constexpr (PMODE == 2) {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) acc[r][c] *= a.phase_unit2;
  }

  // ---- epilogue, any latent width (LT = SK_LT_ANY; the reference stacks any n_neurons, models.py:3133-3160): the low-d squared
  //      distances over ALL columns first, eight at a time; then, chunk by chunk, the gradient's partial sums as below
  if constexpr (LT == SK_LT_ANY) {
    float wr[RM], wc[RN];
    long rowi[RM], rowj[RN];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int row = i0 + ty + 16 * r;
      wr[r] = row < a.row_end ? 1.f : 0.f;
      rowi[r] = (long)min(row, a.B - 1) * L;
    }
#pragma unroll
    for (int c = 0; c < RN; ++c) {
      const int row = j0 + tx + 16 * c;
      wc[c] = row < a.B ? 1.f : 0.f;
      rowj[c] = (long)min(row, a.B - 1) * L;
    }
    float dl2[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) dl2[r][c] = 0.f;
    for (int l0 = 0; l0 < L; l0 += 8) {
      float zi[RM][8], zj[RN][8];
#pragma unroll
      for (int l = 0; l < 8; ++l) {
#pragma unroll
        for (int r = 0; r < RM; ++r) zi[r][l] = l0 + l < L ? a.Z[rowi[r] + l0 + l] : 0.f;
#pragma unroll
        for (int c = 0; c < RN; ++c) zj[c][l] = l0 + l < L ? a.Z[rowj[c] + l0 + l] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c)
#pragma unroll
          for (int l = 0; l < 8; ++l) {
            const float dz = zi[r][l] - zj[c][l];
            dl2[r][c] = fmaf(dz, dz, dl2[r][c]);
          }
    }
    float g[RM][RN];
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        const float Eh = sig_E_high<FASTSIG>(acc[r][c], a.sh);
        float El, G;
        sig_E_low<FASTSIG>(dl2[r][c], a.sl, El, G);
        const float f = (El - Eh) * (wr[r] * wc[c]);
        lsum = fmaf(f, f, lsum);
        g[r][c] = f * G;
      }
    float* red = smem + SMEM - 8;
    const float tl = block_sum<NT>(lsum, red);
    if (tid == 0) a.tile_loss[tslot] = offdiag ? 2.f * tl : tl;
    float* rpart = smem;                       // [TM][8][16]
    float* cpart = smem + 16 * TM * 8;         // [TN][8][16]
    for (int l0 = 0; l0 < L; l0 += 8) {
      float zi[RM][8], zj[RN][8], gi[RM][8];
#pragma unroll
      for (int l = 0; l < 8; ++l) {
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          zi[r][l] = l0 + l < L ? a.Z[rowi[r] + l0 + l] : 0.f;
          gi[r][l] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < RN; ++c) zj[c][l] = l0 + l < L ? a.Z[rowj[c] + l0 + l] : 0.f;
      }
      __syncthreads();   // (the chunk in front has been folded)
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        float gj[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) gj[l] = 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
          for (int l = 0; l < 8; ++l) {
            const float dz = zi[r][l] - zj[c][l];
            gi[r][l] = fmaf(g[r][c], dz, gi[r][l]);
            gj[l] = fmaf(-g[r][c], dz, gj[l]);
          }
        if (offdiag) {
#pragma unroll
          for (int l = 0; l < 8; ++l) cpart[((tx + 16 * c) * 8 + l) * 16 + ty] = gj[l];
        }
      }
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int l = 0; l < 8; ++l) rpart[((ty + 16 * r) * 8 + l) * 16 + tx] = gi[r][l];
      __syncthreads();
      for (int e = tid; e < TM * 8; e += NT) {
        const int rl = e >> 3, l = l0 + (e & 7);
        if (l < L) {
          const float4* p = reinterpret_cast<const float4*>(&rpart[e * 16]);
          const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
          const float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                          (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
          a.P[((long)tj * a.nrows_pad + (i0 - a.row0) + rl) * L + l] = s;
        }
      }
      if (offdiag) {
        for (int e = tid; e < TN * 8; e += NT) {
          const int rl = e >> 3, l = l0 + (e & 7);
          if (l < L) {
            const float4* p = reinterpret_cast<const float4*>(&cpart[e * 16]);
            const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
            const float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                            (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
            a.P[((long)ti * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = s;   // symm => row0 == 0
          }
        }
      }
    }
    return;
  }

  // ---- epilogue -----------------------------------------------------------------------------
  float zi[RM][LMAX], zj[RN][LMAX], wr[RM], wc[RN];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    int row = i0 + ty + 16 * r;
    bool ok = row < a.row_end;
    wr[r] = ok ? 1.f : 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) zi[r][l] = (ok && l < L) ? a.Z[(long)row * L + l] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < RN; ++c) {
    int row = j0 + tx + 16 * c;
    bool ok = row < a.B;
    wc[c] = ok ? 1.f : 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) zj[c][l] = (ok && l < L) ? a.Z[(long)row * L + l] : 0.f;
  }

  float* rpart = smem;                       // [TM][LMAX][16]
  float* cpart = smem + 16 * TM * LMAX;      // [TN][LMAX][16]
  float gi[RM][LMAX];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gi[r][l] = 0.f;
  float lsum = 0.f;

#pragma unroll
  for (int c = 0; c < RN; ++c) {
    float gj[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gj[l] = 0.f;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      float dz[LMAX];
      float dl2 = 0.f;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        dz[l] = zi[r][l] - zj[c][l];
        dl2 = fmaf(dz[l], dz[l], dl2);
      }
      float Eh = sig_E_high<FASTSIG>(acc[r][c], a.sh);
      float El, G;
      sig_E_low<FASTSIG>(dl2, a.sl, El, G);
      float f = (El - Eh) * (wr[r] * wc[c]);
      lsum = fmaf(f, f, lsum);
      float g = f * G;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        gi[r][l] = fmaf(g, dz[l], gi[r][l]);
        gj[l] = fmaf(-g, dz[l], gj[l]);
      }
    }
    if (offdiag) {
#pragma unroll
      for (int l = 0; l < LMAX; ++l) cpart[((tx + 16 * c) * LMAX + l) * 16 + ty] = gj[l];
    }
  }
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) rpart[((ty + 16 * r) * LMAX + l) * 16 + tx] = gi[r][l];

  // tile loss: deterministic block reduction (own small LDS area behind the partials)
  float* red = smem + SMEM - 8;
  float tl = block_sum<NT>(lsum, red);  // contains __syncthreads -> partials visible
  if (tid == 0) a.tile_loss[tslot] = offdiag ? 2.f * tl : tl;

  // fold the 16 partials per (row, l) in fixed order and publish to this tile's private slot
  for (int e = tid; e < TM * LMAX; e += NT) {
    int rl = e / LMAX, l = e - rl * LMAX;
    if (l < L) {
      const float4* p = reinterpret_cast<const float4*>(&rpart[e * 16]);
      float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
      float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
      int row_local = (i0 - a.row0) + rl;
      a.P[((long)tj * a.nrows_pad + row_local) * L + l] = s;
    }
  }
  if (offdiag) {
    for (int e = tid; e < TN * LMAX; e += NT) {
      int rl = e / LMAX, l = e - rl * LMAX;
      if (l < L) {
        const float4* p = reinterpret_cast<const float4*>(&cpart[e * 16]);
        float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
        float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                  (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
        int row_local = (j0 - a.row0) + rl;  // symm => row0 == 0
        a.P[((long)ti * a.nrows_pad + row_local) * L + l] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Euclidean tiles on the matrix cores: d2_ij = |x_i|^2 + |x_j|^2 - 2 x_i.x_j  -- the reference's own Gram
// formulation (encodermap/misc/distances.py:216-227) -- with the cross term on v_mfma_f32_32x32x2_f32 (exact
// f32: bitwise an fmaf chain, same 157 TFLOP/s peak as the VALU but one MFMA covers 2048 multiply-adds and leaves
// the VALU to the sigmoid epilogue of the co-resident tiles).  Rows are shifted by the first row of the batch
// before the products (distances are translation invariant; a common offset of the data would otherwise dominate
// the cancellation), the diagonal is exactly 0 and d2 is clamped at 0 like distances.py:229.
// 64x64 tile per workgroup, one 32x32 quadrant per wave.  k-assignment: in each group of 8 k the lane half
// h = lane>>5 supplies k = 8c + 4h + i to the i-th of four MFMAs, so A and B fragments are single ds_read_b128.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef EMX_CHAIN_TIMING   // per-workgroup timeline of the all-pairs kernel (tools/abl_tiles.py)
__device__ long long emx_dbg_sk[4 * 32 + 3 * 4096];
#define KSTAMP(k) do { if (threadIdx.x == 0) { if ((blockIdx.x & 511) == 0 && blockIdx.x < 2048) emx_dbg_sk[(blockIdx.x >> 9) * 32 + (k)] = clock64(); \
    if (blockIdx.x < 4096 && ((k) == 0 || (k) == 23)) { emx_dbg_sk[128 + 3 * blockIdx.x + ((k) ? 1 : 0)] = wall_clock64(); if (k) emx_dbg_sk[128 + 3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(0x1804 /* HW_ID... */); } } } while (0)
#else
#define KSTAMP(k)
#endif

template <int LT, bool FASTSIG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_vgpr(64))) void sk_tile_mfma_kernel(const SkArgs a) {
  // Resource budget: <= 64 VGPRs and <= 20 KB LDS per workgroup so that 8 tiles (8 waves per SIMD) are resident per
  // CU: a tile alternates an MFMA phase (K loop) and a VALU phase (sigmoid epilogue), and only co-resident tiles in
  // different phases keep both pipes busy (measured: 4 resident tiles leave MFMA at 30 % and VALU at 40 %).
  constexpr int T = 64;
  constexpr int LMAX = LT ? LT : 4;
  constexpr int RP = 36;                                // 32 pre-reduced column-lane partials per (row, l) + pad
  constexpr int STAGE = 2 * T * LS;                     // Xi, Xj images
  constexpr int PART = T * LMAX * RP;                   // row partials (reuse the staging area)
  constexpr int MAIN = STAGE > PART ? STAGE : PART;
  constexpr int SMEM = MAIN + 2 * T + T * LMAX + T * LMAX * 2 + 8;   // + norms + latent rows of the tile + column partials
  __shared__ __align__(16) float smem[SMEM];
  float* Xi_s = smem;
  float* Xj_s = smem + T * LS;
  float* ni_s = smem + MAIN;
  float* nj_s = ni_s + T;
  float* zi_s = nj_s + T;                               // [T][LMAX]
  float* cpart = zi_s + T * LMAX;                       // [T][LC][2]
  float* red = cpart + T * LMAX * 2;

  const int L = LT == 2 ? 2 : a.L;   // LT: 2 = exactly two latent columns (compile time); 0 / 8 = up to 4 / 8 columns at run time, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;     // quadrant of this wave
  const int lr = lane & 31, h = lane >> 5;

  int ti, tj;
  if (a.symm) {
    tri_decode(blockIdx.x, a.nbc, ti, tj);
  } else {
    ti = blockIdx.x / a.nbc;
    tj = blockIdx.x - ti * a.nbc;
  }
  if (dealt_away(a, blockIdx.x)) return;
  const int i0 = a.row0 + ti * T;
  const int j0 = tj * T;
  const bool offdiag = a.symm && (ti != tj);

  KSTAMP(0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // latent rows of the tile's i side -> LDS (read back as broadcasts in the epilogue instead of living in registers)
  for (int e = tid; e < T * LMAX; e += NT) {
    const int rl = e / LMAX, l = e - rl * LMAX;
    const int ig = min(i0 + rl, a.B - 1);
    zi_s[e] = (l < L) ? a.Z[(long)ig * L + l] : 0.f;
  }

  // ---- K loop: stage (x - x_ref) chunks, accumulate row norms, MFMA ---------------------------------
  const int kq = tid & 7, rb = tid >> 3;
  float4 pi[2], pj[2];
  float pni[2] = {0.f, 0.f}, pnj[2] = {0.f, 0.f};
  auto gload = [&](int k0) {
    const int k = k0 + 4 * kq;
    const float4 xr = load_row4(a.X, 0, a.ldx, k, a.D, true, true);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int row = i0 + rb + 32 * q;
      float4 v = load_row4(a.X, min(row, a.B - 1), a.ldx, k, a.D, true, true);
      float w = (row < a.row_end && k < a.D) ? 1.f : 0.f;
      pi[q] = make_float4((v.x - xr.x) * w, (v.y - xr.y) * w, (v.z - xr.z) * w, (v.w - xr.w) * w);
      row = j0 + rb + 32 * q;
      v = load_row4(a.X, min(row, a.B - 1), a.ldx, k, a.D, true, true);
      w = (row < a.B && k < a.D) ? 1.f : 0.f;
      pj[q] = make_float4((v.x - xr.x) * w, (v.y - xr.y) * w, (v.z - xr.z) * w, (v.w - xr.w) * w);
    }
  };
  gload(0);
  KSTAMP(1);
  for (int k0 = 0; k0 < a.D; k0 += KC) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      *reinterpret_cast<float4*>(&Xi_s[(rb + 32 * q) * LS + 4 * kq]) = pi[q];
      *reinterpret_cast<float4*>(&Xj_s[(rb + 32 * q) * LS + 4 * kq]) = pj[q];
      pni[q] = fmaf(pi[q].x, pi[q].x, fmaf(pi[q].y, pi[q].y, fmaf(pi[q].z, pi[q].z, fmaf(pi[q].w, pi[q].w, pni[q]))));
      pnj[q] = fmaf(pj[q].x, pj[q].x, fmaf(pj[q].y, pj[q].y, fmaf(pj[q].z, pj[q].z, fmaf(pj[q].w, pj[q].w, pnj[q]))));
    }
    __syncthreads();
    KSTAMP(2 + 2 * (k0 / KC));
    gload(k0 + KC < a.D ? k0 + KC : k0);   // unconditional: see sk_tile_f16_kernel
#pragma unroll
    for (int c = 0; c < KC / 8; ++c) {
      const float4 av = *reinterpret_cast<const float4*>(&Xi_s[(wm * 32 + lr) * LS + 8 * c + 4 * h]);
      const float4 bv = *reinterpret_cast<const float4*>(&Xj_s[(wn * 32 + lr) * LS + 8 * c + 4 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    KSTAMP(3 + 2 * (k0 / KC));
  }
  // row norms: the 8 lanes that share a row (kq = 0..7) add their partials in a fixed order
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float s = pni[q], t = pnj[q];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      t += __shfl_xor(t, o, 64);
    }
    if (kq == 0) {
      ni_s[rb + 32 * q] = s;
      nj_s[rb + 32 * q] = t;
    }
  }
  __syncthreads();   // norms visible; all MFMA operand reads done: the staging area becomes the row-partial area
  KSTAMP(20);

  // ---- epilogue: this lane owns column j = wn*32 + lr and the 16 rows (reg&3) + 8(reg>>2) + 4h of the quadrant
  const int jl = wn * 32 + lr;
  const int jg = j0 + jl;
  const bool jok = jg < a.B;
  const float nj = nj_s[jl];
  float zj[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) zj[l] = (jok && l < L) ? a.Z[(long)jg * L + l] : 0.f;

  float* rpart = smem;   // [T][LMAX][RP]
  float gj[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) gj[l] = 0.f;
  float lsum = 0.f;
  // interior tiles (the vast majority) need no validity masks and no diagonal fix-up
  const bool edge = (i0 + T > a.row_end) || (j0 + T > a.B) || (i0 == j0);
  auto pair = [&](int r, bool masked) {
    const int il = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    float dz[LMAX];
    float dl2 = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      dz[l] = zi_s[il * LMAX + l] - zj[l];
      dl2 = fmaf(dz[l], dz[l], dl2);
    }
    float d2 = fmaxf(fmaf(-2.f, acc[r], ni_s[il] + nj), 0.f);
    if (masked) d2 = (i0 + il == jg) ? 0.f : d2;
    const float Eh = sig_E_high<FASTSIG>(d2, a.sh);
    float El, G;
    sig_E_low<FASTSIG>(dl2, a.sl, El, G);
    float f = El - Eh;
    if (masked) f = (i0 + il < a.row_end && jok) ? f : 0.f;
    lsum = fmaf(f, f, lsum);
    const float g = f * G;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      float v = g * dz[l];
      gj[l] -= v;
      v += __shfl_xor(v, 1, 64);                       // neighbouring columns pre-reduced in registers
      if ((lr & 1) == 0) rpart[(il * LMAX + l) * RP + (jl >> 1)] = v;
    }
  };
  if (edge) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pair(r, true);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) pair(r, false);
  }
  if (offdiag) {
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      float s = sum_xor32(gj[l]);   // the two row halves of the quadrant
      if (h == 0) cpart[(jl * LMAX + l) * 2 + wm] = s;
    }
  }
  KSTAMP(21);
  float tl = block_sum<NT>(lsum, red);   // contains __syncthreads -> partials visible
  KSTAMP(22);
  if (tid == 0) a.tile_loss[blockIdx.x] = offdiag ? 2.f * tl : tl;

  for (int e = tid; e < T * LMAX; e += NT) {
    int rl = e / LMAX, l = e - rl * LMAX;
    if (l < L) {
      const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 8; u += 4) {
        float4 s0 = p[u], s1 = p[u + 1], s2 = p[u + 2], s3 = p[u + 3];
        s += (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
             (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
      }
      a.P[((long)tj * a.nrows_pad + (i0 - a.row0) + rl) * L + l] = s;
    }
  }
  if (offdiag) {
    for (int e = tid; e < T * LMAX; e += NT) {
      int rl = e / LMAX, l = e - rl * LMAX;
      if (l < L) a.P[((long)ti * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = cpart[e * 2] + cpart[e * 2 + 1];
    }
  }
  KSTAMP(23);
}

// The same tile with the Gram product on the 16-bit matrix cores (split.h): operands are the pre-split f16 planes of the rows.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: an array of them stays in scratch memory)

// NCH > 0: the number of 32-feature chunks is a compile-time constant and the K loop is fully unrolled -- with a rolled loop
// the prefetched pieces are loop-carried registers, the compiler loads into temporaries and copies them into place at the
// loop's end, and that copy waits for the data one instruction after it was requested.
template <int LT, bool FASTSIG, int NCH>
// (leading scalars: what the tile decode and the first operand loads need, preloaded into scalar registers at wave launch --
//  -amdgpu-kernarg-preload-count; 56 bytes, the struct follows)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(LT > 8 ? 3 : 8, LT > 8 ? 3 : 8))) void sk_tile_f16_kernel(
    const uint16_t* p_planes, long p_plane_stride, int p_xcd_nbg, int p_nbc, int p_symm, int p_deal_world, int p_deal_rank, int p_row0, int p_B,
    int p_split_B, int p_split_ld, int p_kmode_ksplit, const SkArgs a) {
  // (the compiler sizes its register budget for the occupancy a 64-KB LDS would allow; gfx950 has 160 KB, so the target
  //  is stated: 8 tiles resident per CU: <= 64 VGPRs, <= 20 KB LDS)
  constexpr int T = 64;
  constexpr int LMAX = LT ? LT : 4;
  // LT = 8: latent widths 5 .. 8 (run-time width, zero padded).  Every consumer plans latents beyond four columns on the 32-tile
  // layout of the workspace (pick_tile: their VALU tiles only fit the static LDS limit at 32 x 32); this 64 x 64 tile therefore
  // WRITES that layout (OUT32) -- its row sums into slot 2 tj of the 32-row blocks it covers, zeros into slot 2 tj + 1, its cost
  // into the slot of 32-tile (2 ti, 2 tj) and zeros into its other 32-tiles' -- as the 128 x 128 macro-tiles write the 64-tile
  // layout (below).  The sums the consumers form over the slots are the 32-tile pass's up to the association of the additions.
  constexpr bool OUT32 = LT > 4;
  // LT = 12 / 16: latent widths 9 .. 12 / .. 16 (run-time width, zero padded): the epilogue walks the latent EIGHT columns at a time (LC) -- the
  // squared low-d distances of the lane's sixteen pairs over all columns first, then the pairs' coefficients (they replace the Gram
  // accumulators), then per chunk of columns the gradient partials through the same reduction area as the eight-column form.
  constexpr int LC = LT > 8 ? 8 : LMAX;                 // latent columns per epilogue pass
  constexpr int RP = 20;                                // 16 pre-reduced column-lane partials per (row, l) + pad
  constexpr int STAGE = 1024 * 4;                       // 2 sides x 2 planes x 64 rows x 32 k f16, as 16-byte pieces
  constexpr int PART = T * LC * RP;                     // row partials (reuse the staging area)
  constexpr int MAIN = STAGE > PART ? STAGE : PART;
  constexpr int SMEM = MAIN + 4 * T + T * LMAX + T * LC * 2 + 8;   // + (norm, 1/scale) of 2 T rows + latent rows of the tile + column partials
  __shared__ __align__(16) float smem[SMEM];
  float2* nr_s = reinterpret_cast<float2*>(smem + MAIN);   // [2 T] (|x|^2, 1/s): rows of the i side, then of the j side
  float* zi_s = smem + MAIN + 4 * T;                    // [T][LMAX]
  float* cpart = zi_s + T * LMAX;                       // [T][LMAX][2]
  float* red = cpart + T * LC * 2;

  const int L = (LT >= 1 && LT <= 4) ? LT : a.L;   // LT: 1 .. 4 = exactly that many latent columns (compile time); 0 = up to 4 columns at run time, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;     // quadrant of this wave
  const int lr = lane & 31, h = lane >> 5;

  const int kmode = NCH == 0 ? (p_kmode_ksplit >> 16) : 0;
  const int ksplit_ = p_kmode_ksplit & 0xffff;
  const int tile_id = kmode == 3 ? (int)blockIdx.x / ksplit_ : (int)blockIdx.x;
  int ti, tj, tslot = tile_id;
  if (p_xcd_nbg > 0) {
    if (!xcd_tile(blockIdx.x, p_xcd_nbg, p_nbc, ti, tj, tslot)) return;
  } else if (p_symm) {
    tri_decode(tile_id, p_nbc, ti, tj);
  } else {
    ti = tile_id / p_nbc;
    tj = tile_id - ti * p_nbc;
  }
  if (p_deal_world > 1 && tslot % p_deal_world != p_deal_rank) {   // (dealt_away with the preloaded copies)
    if (threadIdx.x == 0) a.tile_loss[tslot] = 0.f;
    return;
  }
  const int i0 = p_row0 + ti * T;
  const int j0 = tj * T;
  const bool offdiag = p_symm && (ti != tj);

  KSTAMP(0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // ---- K loop: 32 reduction indices per chunk; the four operand planes of the tile's 64 + 64 rows are staged as 16-byte
  //      pieces (8 f16) -- one contiguous 4-KB block per side and plane in the chunk-major planes -- through an LDS image
  //      that is XOR-swizzled so that the ds_read_b128 of 16 consecutive rows covers all 64 banks; 2 x 3 MFMAs per chunk
  SplitRows sp = a.split;      // (norms, 1 / scale: used behind the first loads)
  sp.planes = const_cast<uint16_t*>(p_planes);
  sp.plane_stride = p_plane_stride;
  sp.B = p_split_B;
  sp.ld = p_split_ld;
  const uint16_t* gp[4];
  int lo[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + NT * u;                       // piece: side (i / j), plane, row, quarter q
    const int side = e >> 9, plane = (e >> 8) & 1, row = (e & 255) >> 2, q = e & 3;
    const int grow = min((side ? j0 : i0) + row, p_B - 1);
    gp[u] = sp.planes + plane * sp.plane_stride + (long)grow * SPLIT_KC + 8 * q;
    lo[u] = ((side * 2 + plane) * 64 + row) * 4 + (q ^ ((row >> 2) & 3));
  }
  u32x4 pre[4];
  const long chunk_stride = (long)sp.B * SPLIT_KC;
  const int nch = NCH > 0 ? NCH : sp.ld / SPLIT_KC;
  const int ra = wm * 32 + lr, rbj = wn * 32 + lr;
  u32x4* buf = reinterpret_cast<u32x4*>(smem);
  // (no lambdas around `pre`: captured by reference it stays an addressable array, i.e. in scratch memory)
#define SK_GLOAD(cc)                                                                                       \
  {                                                                                                        \
    const long goff = (long)min((cc), nch - 1) * chunk_stride;                                             \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) pre[u] = *reinterpret_cast<const u32x4*>(gp[u] + goff); \
  }
#define SK_CHUNK(cc, more)                                                                                              \
  {                                                                                                                     \
    __syncthreads();                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) buf[lo[u]] = pre[u];                                                  \
    __syncthreads();                                                                                                    \
    if ((cc) < 4) KSTAMP(2 + 2 * (cc));                                                                                 \
    if (more) SK_GLOAD((cc) + 1)                                                                                        \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
      const int qa = (2 * t + h) ^ ((ra >> 2) & 3), qb = (2 * t + h) ^ ((rbj >> 2) & 3);                                \
      const f16x8 ah = __builtin_bit_cast(f16x8, buf[ra * 4 + qa]), al = __builtin_bit_cast(f16x8, buf[(64 + ra) * 4 + qa]);          \
      const f16x8 bh = __builtin_bit_cast(f16x8, buf[(128 + rbj) * 4 + qb]), bl = __builtin_bit_cast(f16x8, buf[(192 + rbj) * 4 + qb]); \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0); /* small products first */                    \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);                                               \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);                                               \
    }                                                                                                                   \
    if ((cc) < 4) KSTAMP(3 + 2 * (cc));                                                                                 \
  }
  if (kmode == 0) SK_GLOAD(0)   // ahead of the small prologue loads
  KSTAMP(1);
  // latent rows of the tile's i side -> LDS (read back as broadcasts in the epilogue instead of living in registers).  Behind
  // the first chunk's requests: the latent was written by the launch in front of this one, its first use costs a full trip to
  // memory, and in front of the requests that trip was what every tile's first chunk waited for.
  for (int e = tid; e < T * LMAX; e += NT) {
    const int rl = e / LMAX, l = e - rl * LMAX;
    const int ig = min(i0 + rl, a.B - 1);
    zi_s[e] = (l < L) ? a.Z[(long)ig * L + l] : 0.f;
  }
  for (int e = tid; e < 2 * T; e += NT) {             // (norm, 1/scale) of both sides' rows
    const int grow = min((e < T ? i0 : j0 - T) + e, a.B - 1);
    nr_s[e] = make_float2(sp.norms[grow], sp.rscale[grow]);
  }
  if constexpr (NCH > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) SK_CHUNK(c, c + 1 < NCH)
  } else if (kmode == 0) {
    for (int c = 0; c < nch; ++c) SK_CHUNK(c, true)   // (the last request re-reads its own chunk)
  } else {
    // kmode 3, ONE launch for the split reduction: the workgroup leaves its share of the chunks in kpart (sc1 stores, drained), takes a
    // ticket of its tile, and the LAST of the tile's workgroups adds the shares in the order of the chunks and runs the epilogue.  Nobody waits; the last one leaves the ticket word at zero for the next launch.
    const int sidx = (int)blockIdx.x - tile_id * a.ksplit;
    const int c0 = (int)((long)sidx * nch / a.ksplit), c1 = (int)((long)(sidx + 1) * nch / a.ksplit);
    if (c0 < c1) {
      SK_GLOAD(c0)
      for (int c = c0; c < c1; ++c) SK_CHUNK(c, true)
    }
    float* dst = a.kpart + ((long)blockIdx.x * 4 + wave) * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) emx::sk::store_sc1(dst + 64 * r, acc[r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) red[0] = __uint_as_float(__hip_atomic_fetch_add(&a.ktickets[tile_id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const unsigned ticket = __float_as_uint(red[0]);
    if (ticket != (unsigned)a.ksplit - 1u) return;
    if (tid == 0) __hip_atomic_store(&a.ktickets[tile_id], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* src = a.kpart + ((long)tile_id * a.ksplit * 4 + wave) * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s2 = 0; s2 < a.ksplit; ++s2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += emx::sk::load_sc1(src + (long)s2 * 4096 + 64 * r);
  }
#undef SK_CHUNK
#undef SK_GLOAD
  __syncthreads();   // norms visible; all MFMA operand reads done: the staging area becomes the row-partial area
  KSTAMP(20);

  // ---- epilogue: this lane owns column j = wn*32 + lr and the 16 rows (reg&3) + 8(reg>>2) + 4h of the quadrant
  const int jl = wn * 32 + lr;
  const int jg = j0 + jl;
  const bool jok = jg < a.B;
  const float2 nrj = nr_s[T + jl];
  const float nj = nrj.x, cj = -2.f * nrj.y;
  if constexpr (LT > 8) {
    // ---- latent widths 9 .. 16, eight columns at a time
    const float* zjp = a.Z + (long)min(jg, a.B - 1) * L;   // the latent row of column j (re-read per chunk: a few L2 hits)
    float* rpart = smem;                                   // [T][LC][RP]
    const bool edge = (i0 + T > a.row_end) || (j0 + T > a.B) || (i0 == j0);
    const int ib = wm * 32 + 4 * h;
    const float2* nr_b = nr_s + ib;
    const float* zi_b = zi_s + ib * LMAX;
    float* rp_b = rpart + ib * LC * RP + (jl >> 2);
    float dl2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dl2[r] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < LMAX; c0 += LC) {
      float zj[LC];
#pragma unroll
      for (int l = 0; l < LC; ++l) zj[l] = (jok && c0 + l < L) ? zjp[c0 + l] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int l = 0; l < LC; ++l) {
          if (c0 + l < LMAX) {   // (LT = 12: the second chunk is four columns wide)
            const float d = zi_b[ro * LMAX + c0 + l] - zj[l];
            dl2[r] = fmaf(d, d, dl2[r]);
          }
        }
      }
    }
    // the pairs' coefficients g = f G (they take the Gram accumulators' registers) and the tile's cost
    float lsum = 0.f;
    auto coef = [&](int r, bool masked) {
      const int ro = (r & 3) + 8 * (r >> 2);
      const float2 nri = nr_b[ro];
      float d2 = fmaf(acc[r] * nri.y, cj, nri.x + nj);
      if (!FASTSIG) d2 = fmaxf(d2, 0.f);
      if (masked) d2 = (i0 + ib + ro == jg) ? 0.f : d2;
      const float Eh = sig_E_high<FASTSIG>(d2, a.sh);
      float El, G;
      sig_E_low<FASTSIG>(dl2[r], a.sl, El, G);
      float f = El - Eh;
      if (masked) f = (i0 + ib + ro < a.row_end && jok) ? f : 0.f;
      lsum = fmaf(f, f, lsum);
      acc[r] = f * G;
    };
    if (edge) {
#pragma unroll
      for (int r = 0; r < 16; ++r) coef(r, true);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) coef(r, false);
    }
    const float tl = block_sum<NT>(lsum, red);   // (contains __syncthreads: the staging area is free)
    const int nbc = a.nbc, nbr = a.nbr;
    if (tid < 4) {   // the 32-tile layout: this tile's cost into the slot of 32-tile (2 ti, 2 tj), zeros into its other 32-tiles'
      const int si = 2 * ti + (tid >> 1), sj = 2 * tj + (tid & 1);
      const float val = tid == 0 ? (offdiag ? 2.f * tl : tl) : 0.f;
      if (p_symm) {
        if (si <= sj && sj < nbc) a.tile_loss[si * nbc - si * (si - 1) / 2 + (sj - si)] = val;
      } else if (si < nbr && sj < nbc) {
        a.tile_loss[si * nbc + sj] = val;
      }
    }
    const int ir = i0 - a.row0;
    auto chunk = [&](const int c0, auto W_) __attribute__((always_inline)) {
      constexpr int W = decltype(W_)::value;                                // columns c0 .. c0 + W - 1 (W <= LC)
      if (c0 >= L) return;                                                  // (workgroup-uniform)
      float zj[W], gj[W];
#pragma unroll
      for (int l = 0; l < W; ++l) {
        zj[l] = (jok && c0 + l < L) ? zjp[c0 + l] : 0.f;
        gj[l] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int l = 0; l < W; ++l) {
          float v = acc[r] * (zi_b[ro * LMAX + c0 + l] - zj[l]);
          gj[l] -= v;
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
          rp_b[(ro * LC + l) * RP] = v;
        }
      }
      if (offdiag) {
#pragma unroll
        for (int l = 0; l < W; ++l) {
          const float sgj = sum_xor32(gj[l]);   // the two row halves of the quadrant
          if (h == 0) cpart[(jl * LC + l) * 2 + wm] = sgj;
        }
      }
      __syncthreads();
      for (int e = tid; e < T * LC; e += NT) {
        const int rl = e / LC, l = c0 + (e - rl * LC);
        if (l < L && l < c0 + W && ir + rl < a.nrows_pad) {
          const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
          const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
          const float sum = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                            (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
          a.P[((long)(2 * tj) * a.nrows_pad + ir + rl) * L + l] = sum;
          if (2 * tj + 1 < nbc) a.P[((long)(2 * tj + 1) * a.nrows_pad + ir + rl) * L + l] = 0.f;
        }
      }
      if (offdiag) {
        for (int e = tid; e < T * LC; e += NT) {
          const int rl = e / LC, l = c0 + (e - rl * LC);
          if (l < L && l < c0 + W && j0 - a.row0 + rl < a.nrows_pad) {
            a.P[((long)(2 * ti) * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = cpart[e * 2] + cpart[e * 2 + 1];
            a.P[((long)(2 * ti + 1) * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = 0.f;
          }
        }
      }
      __syncthreads();   // (the next chunk overwrites the partial areas)
    };
    if constexpr (LMAX - LC == LC) {   // two chunks of the same width: ONE copy of the code
#pragma unroll 1
      for (int c0 = 0; c0 < LMAX; c0 += LC) chunk(c0, std::integral_constant<int, LC>{});
    } else {
      chunk(0, std::integral_constant<int, LC>{});
      chunk(LC, std::integral_constant<int, LMAX - LC>{});
    }
    return;
  }
  float zj[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) zj[l] = (jok && l < L) ? a.Z[(long)jg * L + l] : 0.f;

  float* rpart = smem;   // [T][LMAX][RP]
  float gj[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) gj[l] = 0.f;
  float lsum = 0.f;
  // interior tiles (the vast majority) need no validity masks and no diagonal fix-up
  const bool edge = (i0 + T > a.row_end) || (j0 + T > a.B) || (i0 == j0);
  // Row r of this lane is wm*32 + 4h + ro(r), ro(r) = (r & 3) + 8 (r >> 2): one base address per array, the pair's row
  // as an immediate offset.
  const int ib = wm * 32 + 4 * h;
  const float2* nr_b = nr_s + ib;
  const float* zi_b = zi_s + ib * LMAX;
  float* rp_b = rpart + ib * LMAX * RP + (jl >> 2);
  constexpr bool ROWS4 = LMAX <= 3;
  float vrow[4][LMAX];
  auto pair = [&](int r, bool masked) {
    const int ro = (r & 3) + 8 * (r >> 2);
    float dz[LMAX];
    float dl2 = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      dz[l] = zi_b[ro * LMAX + l] - zj[l];
      dl2 = fmaf(dz[l], dz[l], dl2);
    }
    const float2 nri = nr_b[ro];
    float d2 = fmaf(acc[r] * nri.y, cj, nri.x + nj);   // |xi|^2 + |xj|^2 - 2 (xs_i . xs_j) / (s_i s_j)
    if (!FASTSIG) d2 = fmaxf(d2, 0.f);                 // (the fast form takes an even power of d^2: no clamp at 0 needed)
    if (masked) d2 = (i0 + ib + ro == jg) ? 0.f : d2;
    const float Eh = sig_E_high<FASTSIG>(d2, a.sh);
    float El, G;
    sig_E_low<FASTSIG>(dl2, a.sl, El, G);
    float f = El - Eh;
    if (masked) f = (i0 + ib + ro < a.row_end && jok) ? f : 0.f;
    lsum = fmaf(f, f, lsum);
    const float g = f * G;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      float v = g * dz[l];
      gj[l] -= v;
      if constexpr (ROWS4) {
        vrow[r & 3][l] = v;
      } else {   // (four latent columns: the grouped form costs the kernel a resident tile per CU -- every pair reduced and stored by itself,
                 //  the four lanes of a quad hold the same sum and all store it to the same word)
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        rp_b[(ro * LMAX + l) * RP] = v;
      }
    }
  };
  // Row sums: four neighbouring columns are pre-reduced in registers by DPP quad permutations (lane ^ 1, lane ^ 2), not LDS
  // bpermutes.  The four pairs r = 4 q4 .. 4 q4 + 3 of a lane are four CONSECUTIVE rows: the quad reduces them as a transposing
  // butterfly -- lane q of the quad ends with the quad's sum for row q (the same additions in the same association as reducing each
  // pair by itself: (own pair of lanes) + (the other pair)) -- and ONE store per four pairs and latent column leaves the wave
  // instead of four (a 4-byte LDS store costs the LDS pipe four cycles whatever it carries).
  const bool odd1 = (lr & 1) != 0, odd2 = (lr & 2) != 0;
  float* rp_q = rp_b + (lr & 3) * LMAX * RP;
  auto rows4 = [&](int q4) __attribute__((always_inline)) {
    if constexpr (!ROWS4) return;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      const float v0 = vrow[0][l], v1 = vrow[1][l], v2 = vrow[2][l], v3 = vrow[3][l];
      const float s01 = odd1 ? v1 : v0, o01 = odd1 ? v0 : v1, s23 = odd1 ? v3 : v2, o23 = odd1 ? v2 : v3;
      const float a2 = s01 + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, o01), 0xB1, 0xF, 0xF, true));
      const float b2 = s23 + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, o23), 0xB1, 0xF, 0xF, true));
      const float s4 = odd2 ? b2 : a2, o4 = odd2 ? a2 : b2;
      rp_q[(8 * q4 * LMAX + l) * RP] = s4 + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, o4), 0x4E, 0xF, 0xF, true));
    }
  };
  if (edge) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
      for (int r = 4 * q4; r < 4 * q4 + 4; ++r) pair(r, true);
      rows4(q4);
    }
  } else {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
      for (int r = 4 * q4; r < 4 * q4 + 4; ++r) pair(r, false);
      rows4(q4);
    }
  }
  if (offdiag) {
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      float s = sum_xor32(gj[l]);   // the two row halves of the quadrant
      if (h == 0) cpart[(jl * LMAX + l) * 2 + wm] = s;
    }
  }
  KSTAMP(21);
  float tl = block_sum<NT>(lsum, red);   // contains __syncthreads -> partials visible
  KSTAMP(22);
  if constexpr (OUT32) {
    // ---- the 32-tile layout (a.nbc, a.nrows_pad, a.P, a.tile_loss are the 32-tile plan's): this tile covers the 32-tiles
    //      (2 ti + p, 2 tj + q); symmetric pass: those with 2 ti + p <= 2 tj + q
    const int nbc = a.nbc, nbr = a.nbr;
    if (tid < 4) {
      const int si = 2 * ti + (tid >> 1), sj = 2 * tj + (tid & 1);
      const float val = tid == 0 ? (offdiag ? 2.f * tl : tl) : 0.f;
      if (p_symm) {
        if (si <= sj && sj < nbc) a.tile_loss[si * nbc - si * (si - 1) / 2 + (sj - si)] = val;
      } else if (si < nbr && sj < nbc) {
        a.tile_loss[si * nbc + sj] = val;
      }
    }
    const int ir = i0 - a.row0;   // first row of the tile among the rows the plan handles
    for (int e = tid; e < T * LMAX; e += NT) {
      const int rl = e / LMAX, l = e - rl * LMAX;
      if (l < L && ir + rl < a.nrows_pad) {
        const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
        const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
        const float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                        (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
        a.P[((long)(2 * tj) * a.nrows_pad + ir + rl) * L + l] = s;
        if (2 * tj + 1 < nbc) a.P[((long)(2 * tj + 1) * a.nrows_pad + ir + rl) * L + l] = 0.f;
      }
    }
    if (offdiag) {
      for (int e = tid; e < T * LMAX; e += NT) {
        const int rl = e / LMAX, l = e - rl * LMAX;
        if (l < L && j0 - a.row0 + rl < a.nrows_pad) {
          a.P[((long)(2 * ti) * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = cpart[e * 2] + cpart[e * 2 + 1];
          a.P[((long)(2 * ti + 1) * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = 0.f;
        }
      }
    }
    return;
  }
  if (tid == 0) a.tile_loss[tslot] = offdiag ? 2.f * tl : tl;

  for (int e = tid; e < T * LMAX; e += NT) {
    int rl = e / LMAX, l = e - rl * LMAX;
    if (l < L) {
      const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
      const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
      const float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                      (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
      a.P[((long)tj * a.nrows_pad + (i0 - a.row0) + rl) * L + l] = s;
    }
  }
  if (offdiag) {
    for (int e = tid; e < T * LMAX; e += NT) {
      int rl = e / LMAX, l = e - rl * LMAX;
      if (l < L) a.P[((long)ti * a.nrows_pad + (j0 - a.row0) + rl) * L + l] = cpart[e * 2] + cpart[e * 2 + 1];
    }
  }
  KSTAMP(23);
}

// ---------------------------------------------------------------------------------------------------------------------
// Large batches (B >= 8192): 128 x 128 MACRO-TILES.  With 64 x 64 tiles the kernel stages 131 KB per tile at D = 256 -- 4.3 GB
// from the L2s into LDS per launch at B = 16384, 12.9 TB/s, i.e. the rate at which the L2s can fill LDS is what bounds it
// (48 executed FLOP per staged byte).  A 128 x 128 tile per workgroup, a 64 x 64 quadrant (2 x 2 matrix-instruction tiles, four
// accumulators) per wave, stages (128 + 128) rows for four times the products: half the bytes per product, half the LDS reads.
// The OUTPUT keeps the 64-tile layout every consumer plans with (make_plan: slot s of row block ib): a macro-tile (TI, TJ)
// writes its row sums into slot 2 TJ (rows of TI), its column sums into slot 2 TI (rows of TJ), zeros into the odd slots
// beside them, and its cost into the slot of 64-tile (2 TI, 2 TJ) (zeros into its other 64-tiles' slots): the sums the
// consumers form over the slots are those of the 64-tile pass up to the association of the additions.  Two latent columns, not
// for dealt tiles (the deal is by 64-tile).
// Measured at B = 16384, D = 256 (profiles/r03_c5_counters.json): 335 -> 257 us; matrix pipe 35 %, VALU 51 % busy, L2 hit rate 90 %
// -- the two pipes take turns (products, then 64 sigmoid pairs per lane) in every workgroup, and three workgroups per CU do not
// interleave them enough (both busy together in a fifth of the matrix cycles; waves waiting for an instruction's result 45 % of
// their time: profiles/r03_c5_stalls.json).  Also built, measured, removed: a persistent, software-pipelined form -- one workgroup
// of eight waves per CU walking its tiles with TWO accumulator sets per wave, the products of tile k + 1 issued between the
// epilogue rows of tile k (two matrix instructions, then a row of pairs; operand pieces three chunks ahead in a register ring):
// 612-747 us per step against 452 -- with two waves per SIMD and one pair per lane and row the epilogue is a string of dependent
// instructions, where this kernel's sixteen unrolled pairs give the scheduler sixteen independent chains; the XCD's tiles walked in 8 x 8 sub-blocks (16 shared panels instead
// of a whole row of them; 457.8 against 450.5 us per step: the planes are served by the L2s either way) and the rolled K loop at
// 128 registers, four workgroups per CU (451.9 against 453.4 us).
// ---------------------------------------------------------------------------------------------------------------------
template <bool FASTSIG, int NCH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void sk_tile_f16_big_kernel(
    const uint16_t* p_planes, long p_plane_stride, int p_xcd_nbg, int p_nbc2, int p_B, int p_split_B, int p_split_ld, const SkArgs a) {
  constexpr int T = 128, LMAX = 2, L = 2, RP = 20;
  constexpr int STAGE = 2048 * 4;                       // 2 sides x 2 planes x 128 rows x 32 k f16, as 16-byte pieces
  constexpr int SMEM = STAGE + 4 * T + T * LMAX + T * LMAX * 2 + 8;
  __shared__ __align__(16) float smem[SMEM];
  float2* nr_s = reinterpret_cast<float2*>(smem + STAGE);   // [2 T] (|x|^2, 1/s): rows of the i side, then of the j side
  float* zi_s = smem + STAGE + 4 * T;                   // [T][LMAX]
  float* cpart = zi_s + T * LMAX;                       // [T][LMAX][2]
  float* red = cpart + T * LMAX * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;     // 64 x 64 quadrant of this wave
  const int lr = lane & 31, h = lane >> 5;

  int TI, TJ, tmacro;
  if (p_xcd_nbg > 0) {
    if (!xcd_tile(blockIdx.x, p_xcd_nbg, p_nbc2, TI, TJ, tmacro)) return;
  } else {
    tri_decode(blockIdx.x, p_nbc2, TI, TJ);
  }
  const int i0 = TI * T, j0 = TJ * T;
  const bool offdiag = TI != TJ;

  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;

  const uint16_t* gp[8];
  int lo[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + NT * u;                        // piece: side (i / j), plane, row, quarter q
    const int side = e >> 10, plane = (e >> 9) & 1, row = (e & 511) >> 2, q = e & 3;
    const int grow = min((side ? j0 : i0) + row, p_B - 1);
    gp[u] = p_planes + plane * p_plane_stride + (long)grow * SPLIT_KC + 8 * q;
    lo[u] = ((side * 2 + plane) * 128 + row) * 4 + (q ^ ((row >> 2) & 3));
  }
  u32x4 pre[8];
  const long chunk_stride = (long)p_split_B * SPLIT_KC;
  const int nch = NCH > 0 ? NCH : p_split_ld / SPLIT_KC;
  u32x4* buf = reinterpret_cast<u32x4*>(smem);
#define SKB_GLOAD(cc)                                                                                      \
  {                                                                                                        \
    const long goff = (long)min((cc), nch - 1) * chunk_stride;                                             \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) pre[u] = *reinterpret_cast<const u32x4*>(gp[u] + goff); \
  }
#define SKB_CHUNK(cc, more)                                                                                             \
  {                                                                                                                     \
    __syncthreads();                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) buf[lo[u]] = pre[u];                                                  \
    __syncthreads();                                                                                                    \
    if (more) SKB_GLOAD((cc) + 1)                                                                                       \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
      f16x8 ah[2], al[2], bh[2], bl[2];                                                                                 \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                   \
        const int ra = wm * 64 + 32 * u + lr, rb = wn * 64 + 32 * u + lr;                                               \
        const int qa = (2 * t + h) ^ ((ra >> 2) & 3), qb = (2 * t + h) ^ ((rb >> 2) & 3);                               \
        ah[u] = __builtin_bit_cast(f16x8, buf[ra * 4 + qa]);                                                            \
        al[u] = __builtin_bit_cast(f16x8, buf[(128 + ra) * 4 + qa]);                                                    \
        bh[u] = __builtin_bit_cast(f16x8, buf[(256 + rb) * 4 + qb]);                                                    \
        bl[u] = __builtin_bit_cast(f16x8, buf[(384 + rb) * 4 + qb]);                                                    \
      }                                                                                                                 \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bl[v], acc[u][v], 0, 0, 0); /* small products first */ \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u], bh[v], acc[u][v], 0, 0, 0);                           \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bh[v], acc[u][v], 0, 0, 0);                           \
    }                                                                                                                   \
  }
  SKB_GLOAD(0)
  for (int e = tid; e < T * LMAX; e += NT) {
    const int rl = e / LMAX, l = e - rl * LMAX;
    const int ig = min(i0 + rl, a.B - 1);
    zi_s[e] = a.Z[(long)ig * L + l];
  }
  for (int e = tid; e < 2 * T; e += NT) {             // (norm, 1/scale) of both sides' rows
    const int grow = min((e < T ? i0 : j0 - T) + e, a.B - 1);
    nr_s[e] = make_float2(a.split.norms[grow], a.split.rscale[grow]);
  }
  if constexpr (NCH > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) SKB_CHUNK(c, c + 1 < NCH)
  } else {
    for (int c = 0; c < nch; ++c) SKB_CHUNK(c, true)   // (the last request re-reads its own chunk)
  }
#undef SKB_CHUNK
#undef SKB_GLOAD
  __syncthreads();   // all MFMA operand reads done: the staging area becomes the row-partial area

  // ---- epilogue: this lane owns the columns 64 wn + 32 v + lr and, per (u, v), the 16 rows 64 wm + 32 u + (r & 3) + 8 (r >> 2) + 4 h
  float zj[2][LMAX], nj[2], cj[2];
  bool jok[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int jl = wn * 64 + 32 * v + lr, jg = j0 + jl;
    jok[v] = jg < a.B;
    const float2 nrj = nr_s[T + jl];
    nj[v] = nrj.x;
    cj[v] = -2.f * nrj.y;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) zj[v][l] = jok[v] ? a.Z[(long)jg * L + l] : 0.f;
  }
  float* rpart = smem;   // [T][LMAX][RP]
  float gj[2][LMAX];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gj[v][l] = 0.f;
  float lsum = 0.f;
  const bool edge = (i0 + T > a.row_end) || (j0 + T > a.B) || (i0 == j0);
  auto rows16 = [&](auto U_, bool masked) __attribute__((always_inline)) {
    constexpr int u = decltype(U_)::value;
    const int ib = wm * 64 + 32 * u + 4 * h;
    const float2* nr_b = nr_s + ib;
    const float* zi_b = zi_s + ib * LMAX;
    float* rp_b = rpart + ib * LMAX * RP + ((wn * 32 + lr) >> 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = (r & 3) + 8 * (r >> 2);
      const float2 nri = nr_b[ro];
      float rowg[LMAX] = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        float dz[LMAX];
        float dl2 = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
          dz[l] = zi_b[ro * LMAX + l] - zj[v][l];
          dl2 = fmaf(dz[l], dz[l], dl2);
        }
        float d2 = fmaf(acc[u][v][r] * nri.y, cj[v], nri.x + nj[v]);
        if (!FASTSIG) d2 = fmaxf(d2, 0.f);
        if (masked) d2 = (i0 + ib + ro == j0 + wn * 64 + 32 * v + lr) ? 0.f : d2;
        const float Eh = sig_E_high<FASTSIG>(d2, a.sh);
        float El, G;
        sig_E_low<FASTSIG>(dl2, a.sl, El, G);
        float f = El - Eh;
        if (masked) f = (i0 + ib + ro < a.row_end && jok[v]) ? f : 0.f;
        lsum = fmaf(f, f, lsum);
        const float g = f * G;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
          const float x = g * dz[l];
          gj[v][l] -= x;
          rowg[l] += x;
        }
      }
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        float x = rowg[l];
        x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
        x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
        rp_b[(ro * LMAX + l) * RP] = x;
      }
    }
  };
  if (edge) {
    rows16(std::integral_constant<int, 0>{}, true);
    rows16(std::integral_constant<int, 1>{}, true);
  } else {
    rows16(std::integral_constant<int, 0>{}, false);
    rows16(std::integral_constant<int, 1>{}, false);
  }
  if (offdiag) {
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        const float s = sum_xor32(gj[v][l]);   // the two row halves of the sub-tiles
        if (h == 0) cpart[((wn * 64 + 32 * v + lr) * LMAX + l) * 2 + wm] = s;
      }
  }
  const float tl = block_sum<NT>(lsum, red);   // contains __syncthreads -> partials visible
  // ---- outputs in the 64-tile layout: nbc = 64-row blocks of the batch; the macro-tile's four (three) 64-tiles
  const int nbc = a.nbc;
  if (tid < 4) {
    const int ti = 2 * TI + (tid >> 1), tj = 2 * TJ + (tid & 1);
    if (ti <= tj && tj < nbc) a.tile_loss[ti * nbc - ti * (ti - 1) / 2 + (tj - ti)] = tid == 0 ? (offdiag ? 2.f * tl : tl) : 0.f;
  }
  for (int e = tid; e < T * LMAX; e += NT) {
    const int rl = e / LMAX, l = e - rl * LMAX;
    const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
    const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
    const float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                    (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
    if (i0 + rl < a.nrows_pad) {
      a.P[((long)(2 * TJ) * a.nrows_pad + i0 + rl) * L + l] = s;
      if (2 * TJ + 1 < nbc) a.P[((long)(2 * TJ + 1) * a.nrows_pad + i0 + rl) * L + l] = 0.f;
    }
  }
  if (offdiag) {
    for (int e = tid; e < T * LMAX; e += NT) {
      const int rl = e / LMAX, l = e - rl * LMAX;
      if (j0 + rl < a.nrows_pad) {
        a.P[((long)(2 * TI) * a.nrows_pad + j0 + rl) * L + l] = cpart[e * 2] + cpart[e * 2 + 1];
        a.P[((long)(2 * TI + 1) * a.nrows_pad + j0 + rl) * L + l] = 0.f;
      }
    }
  }
}

struct FinArgs {
  const float* P;
  const float* tile_loss;
  const float* Z;
  float* dZ;
  float* dist_loss;
  float* center_loss;
  int B, L, row0, row_end, nrows_pad, nslots, ntiles, symm, tile;
  float gscale;      // -4 * scale * (b_l c_l / sig_l^2) / B^2
  float lscale;      // scale / B^2
  float cgrad;       // 2 * center_scale / (B*L)
  float cscale;      // center_scale / (B*L)
};

// grid: ceil(nrows*L/256) blocks for dZ + 1 block for the scalar losses
__global__ __launch_bounds__(NT) void sk_finish_kernel(const FinArgs a) {
  __shared__ float red[8];
  const int nrows = a.row_end - a.row0;
  const int nelem = nrows * a.L;
  if (blockIdx.x == gridDim.x - 1) {
    float s = 0.f;
    for (int t = threadIdx.x; t < a.ntiles; t += NT) s += a.tile_loss[t];
    float tot = block_sum<NT>(s, red);
    if (threadIdx.x == 0) *a.dist_loss = tot * a.lscale;
    if (a.center_loss) {
      float c = 0.f;
      for (int e = threadIdx.x; e < nelem; e += NT) {
        float z = a.Z[(long)a.row0 * a.L + e];
        c = fmaf(z, z, c);
      }
      float ct = block_sum<NT>(c, red);
      if (threadIdx.x == 0) *a.center_loss = ct * a.cscale;
    }
    return;
  }
  int e = blockIdx.x * NT + threadIdx.x;
  if (e >= nelem) return;
  int row_local = e / a.L;
  // every slot of this row was written by exactly one tile; 8 independent partial sums keep 8 loads in flight,
  // the association is fixed -> deterministic
  const long stride = (long)a.nrows_pad * a.L;
  const float* __restrict__ P = a.P + e;
  float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int sl = 0;
  for (; sl + 8 <= a.nslots; sl += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] += P[(long)(sl + u) * stride];
  }
  for (int u = 0; sl < a.nslots; ++sl, ++u) part[u] += P[(long)sl * stride];
  float s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  (void)row_local;
  long gidx = (long)a.row0 * a.L + e;
  float g = a.gscale * s;
  if (a.center_loss) g = fmaf(a.cgrad, a.Z[gidx], g);
  a.dZ[gidx] = g;
}

// Data parallel with dealt tiles: d cost / d z of ALL rows from the tiles THIS rank computed -- for row block ib the slot s
// holds the partial of tile (min(ib, s), max(ib, s)), owned iff its triangle index % world == rank -- plus the center
// cost's gradient on the rank's own rows.  The sum over ranks (one all-reduce of B x L floats) is the full gradient.
struct DealtArgs {
  const float* P;
  const float* Z;
  float* dZ;
  int B, L, nbc, nrows_pad, tile, rank, world, own_begin, own_end;
  float gscale, cgrad;
};
__global__ __launch_bounds__(NT) void sk_reduce_dealt_kernel(const DealtArgs a) {
  const int e = blockIdx.x * NT + threadIdx.x;
  if (e >= a.B * a.L) return;
  const int row = e / a.L, ib = row / a.tile;
  const long stride = (long)a.nrows_pad * a.L;
  float part[4] = {0.f, 0.f, 0.f, 0.f};   // fixed association -> deterministic
  for (int s = 0; s < a.nbc; ++s) {
    const int ti = min(ib, s), tj = max(ib, s);
    const int t = ti * a.nbc - ti * (ti - 1) / 2 + (tj - ti);
    if (t % a.world == a.rank) part[s & 3] += a.P[(long)s * stride + e];
  }
  float g = a.gscale * ((part[0] + part[1]) + (part[2] + part[3]));
  if (row >= a.own_begin && row < a.own_end) g = fmaf(a.cgrad, a.Z[e], g);
  a.dZ[e] = g;
}

SigDev make_sig(float sig, float aa, float bb) {
  SigDev s;
  s.inv_sig2 = 1.f / (sig * sig);
  s.c = (float)(std::pow(2.0, (double)aa / bb) - 1.0);
  s.k6 = (float)(std::pow(std::pow(2.0, (double)aa / bb) - 1.0, 1.0 / 6.0) / ((double)sig * sig));
  s.c_inv = (float)((std::pow(2.0, (double)aa / bb) - 1.0) / ((double)sig * sig));
  s.half_a = 0.5f * aa;
  s.nba = -bb / aa;
  s.half_am2 = 0.5f * (aa - 2.f);
  s.vmax = (float)std::pow(2.0, 100.0 / std::max(0.5 * aa, 1.0));
  s.a_is_2 = (aa == 2.f);
  return s;
}

#define SK_F16_LAUNCH(kern, grid_, args_)                                                                                              \
  hipLaunchKernelGGL(kern, grid_, dim3(NT), 0, st, (const uint16_t*)(args_).split.planes, (long)(args_).split.plane_stride, (args_).xcd_nbg,  \
                     (args_).nbc, (args_).symm, (args_).deal_world, (args_).deal_rank, (args_).row0, (args_).B, (args_).split.B,              \
                     (args_).split.ld, ((args_).kmode << 16) | ((args_).ksplit & 0xffff), args_)

template <int T, int LT, int PMODE, bool FAST>
void launch_tiles(const SkArgs& a, int ntiles, hipStream_t st) {
  hipLaunchKernelGGL((sk_tile_kernel<T, T, LT, PMODE, FAST>), dim3(ntiles), dim3(NT), 0, st, a.X, a.ldx, a.D, a.B, a.row0, a.row_end, a.nbc, a.symm,
                     a.xcd_nbg, a.vec4, a.deal_world, a.deal_rank, 0, a);
}

template <int T, int LT>
void dispatch2(const SkArgs& a, int ntiles, bool periodic, bool fast, hipStream_t st) {
  if (periodic && a.phase_ok) {
    if (fast) launch_tiles<T, LT, 2, true>(a, ntiles, st);
    else launch_tiles<T, LT, 2, false>(a, ntiles, st);
  } else if (periodic) {
    if (fast) launch_tiles<T, LT, 1, true>(a, ntiles, st);
    else launch_tiles<T, LT, 1, false>(a, ntiles, st);
  } else {
    if (fast) launch_tiles<T, LT, 0, true>(a, ntiles, st);
    else launch_tiles<T, LT, 0, false>(a, ntiles, st);
  }
}

constexpr int SMALL_TILE_MAX_B = 512;   // batches up to this size run the VALU tiles at 32 x 32

// floats of the per-tile partial sums + tile costs at the front of the workspace (the larger of the 64- and 32-tile layouts)
size_t ws_front_floats(int B, int L) {
  const size_t nb = (size_t)(B + 63) / 64;
  const size_t f64 = nb * nb * 64 * (size_t)L + nb * nb + 64;
  if (B > SMALL_TILE_MAX_B && L <= 4) return f64;   // (latent widths 5..8 may run the 32 x 32 tiles at any batch size: pick_tile)
  const size_t n32 = (size_t)(B + 31) / 32;
  return std::max(f64, n32 * n32 * 32 * (size_t)L + n32 * n32 + 64);
}

int pick_tile(int B, int nrows, bool gram, bool periodic, int L) {
  if (L > 4) return 32;   // latent widths 5 and more run the 32 x 32 VALU tiles (their partial-sum area fits the static LDS limit at that size
                          // only); like the rule below a function of what every entry point that plans the workspace layout knows
  if (periodic && B <= SMALL_TILE_MAX_B) return 32;   // (a function of B and the periodicity only: every entry point that plans
                                                      //  the workspace layout knows both)
  // 64x64 for every large batch.  The MFMA (Gram) tiles exist in that size only, and the VALU kernel's 128x128 instantiation
  // (half the LDS traffic per pair-element) needs 256 VGPRs + 12 AGPRs: one wave per SIMD, where the VALU issues 0.15
  // instructions per clock instead of 0.35 at four waves -- measured at B = 8192, D = 512 periodic: 2.19 ms against 1.40 ms.
  // Small batches on the VALU tiles (periodic data): 32x32, because a 256-row batch is ten 64-tiles on ten CUs -- 36 tiles
  // of a quarter of the pairs each finish sooner (the default EncoderMap parameters are such a batch).
  (void)nrows;
  (void)gram;
  return 64;
}

__global__ void pairwise_dist_kernel(const float* __restrict__ X, int B, int D, int ldx, float period, int periodic,
                                     float* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= B) return;
  float acc = 0.f;
  for (int k = 0; k < D; ++k) {
    float d = X[(long)i * ldx + k] - X[(long)j * ldx + k];
    if (periodic) {
      d = fabsf(d);
      d = fminf(d, period - d);
    }
    acc = fmaf(d, d, acc);
  }
  out[(long)i * B + j] = sqrtf(acc);
}

}  // namespace

extern "C" {

size_t emx_sketchmap_ws_bytes(int B, int L) {
  if (B <= 0 || L <= 0) return 0;
  size_t nb = (size_t)(B + 63) / 64;
  // + the Gram partials of the split reduction (small batches: at most KSPLIT_BLOCKS workgroups x 64 x 64 floats)
  // in FRONT of everything, wherever (B, L) puts the rest: one ticket word per tile of that reduction (ZERO before the first use; the
  // launches leave them at zero, and no launch of any shape writes anything else there)
  return (KSPLIT_BLOCKS + ws_front_floats(B, L) + (nb <= KSPLIT_MAX_NB ? (size_t)KSPLIT_BLOCKS * 4096 : 0)) * sizeof(float);
}

}  // extern "C" (ws_bytes)

namespace {

struct SkPlan {
  SkArgs a;
  FinArgs f;
  int ntiles, T, nblk_finish;
  bool periodic, fast, gram;
};

int make_plan(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, const float sig[6],
              float scale, float center_scale, int row_begin, int row_end, float* dist_loss, float* center_loss,
              float* dZ, void* ws, size_t ws_bytes, SkPlan& pl, const char* who, void* split_ws = nullptr, float data_span = 0.f) {
  EMX_REQUIRE(X && Z && ws && sig, "%s: null pointer argument", who);
  EMX_REQUIRE(B > 0 && D > 0 && ldx >= D, "%s: bad shape B=%d D=%d ldx=%d", who, B, D, ldx);
  EMX_REQUIRE(L >= 1 && L <= 4096, "%s: latent dimension %d", who, L);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && row_end <= B, "%s: bad row block [%d,%d) of %d", who, row_begin,
              row_end, B);
  EMX_REQUIRE(periodicity > 0.f, "%s: periodicity must be > 0 (INFINITY for none)", who);
  for (int i = 0; i < 6; ++i) EMX_REQUIRE(sig[i] > 0.f, "%s: sigmoid parameters must be > 0", who);
  if (ws_bytes < emx_sketchmap_ws_bytes(B, L))
    return emx::fail(EMX_EWORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, emx_sketchmap_ws_bytes(B, L));
  const bool symm = (row_begin == 0 && row_end == B);
  const int nrows = row_end - row_begin;
  pl.periodic = std::isfinite(periodicity);
  pl.gram = true;
  const bool vec4ok = (D % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const bool split_ok = split_ws != nullptr && pl.gram && !pl.periodic;
  const int T = pick_tile(B, nrows, (pl.gram && !pl.periodic && vec4ok) || split_ok, pl.periodic, L);
  pl.T = T;
  pl.fast = sig[1] == 12.f && sig[2] == 6.f && sig[4] == 2.f && sig[5] == 6.f;
  SkArgs& a = pl.a;
  a.X = X;
  a.Z = Z;
  a.B = B;
  a.D = D;
  a.ldx = ldx;
  a.L = L;
  a.row0 = row_begin;
  a.row_end = row_end;
  a.nbr = (nrows + T - 1) / T;
  a.nbc = (B + T - 1) / T;
  a.nrows_pad = a.nbr * T;
  a.symm = symm;
  a.xcd_nbg = 0;
  a.split = split_ok ? split_view(split_ws, B, D) : SplitRows{};
  a.vec4 = (D % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  a.period = periodicity;
  // Periodic data that span less than 1.5 periods (angles in [-pi, pi) or [0, 360): every set the reference is used on): all
  // |x_i - x_j| < 1.5 p, where the reference's min(|d|, p - |d|) squared IS the squared wrapped difference, which the
  // fixed-point tiles compute.  A wider or unknown span (data_span <= 0) keeps the float form, which follows the reference's
  // formula for any |d|.  EMX_SK_PHASE=0 forces the float form.
  constexpr bool phase_allowed = true;
  a.phase_ok = (pl.periodic && phase_allowed && data_span > 0.f && data_span < 1.5f * periodicity) ? 1 : 0;
  {
    const double rp = pl.periodic ? 1.0 / (double)periodicity : 0.0;
    a.rp_hi = (float)rp;
    a.rp_lo = (float)(rp - (double)a.rp_hi);
    a.phase_unit2 = pl.periodic ? (float)(((double)periodicity / 4294967296.0) * ((double)periodicity / 4294967296.0)) : 0.f;
  }
  a.sh = make_sig(sig[0], sig[1], sig[2]);
  a.sl = make_sig(sig[3], sig[4], sig[5]);
  pl.ntiles = symm ? a.nbc * (a.nbc + 1) / 2 : a.nbr * a.nbc;
  const int nslots = a.nbc;
  float* wsf = static_cast<float*>(ws) + KSPLIT_BLOCKS;   // (behind the ticket words of the split reduction)
  a.P = wsf;
  a.tile_loss = wsf + (size_t)nslots * a.nrows_pad * L;
  a.ksplit = 1;
  a.kmode = 0;
  a.deal_rank = 0;
  a.deal_world = 1;
  {
    const size_t nb = (size_t)(B + 63) / 64;
    a.kpart = nb <= (size_t)KSPLIT_MAX_NB ? wsf + ws_front_floats(B, L) : nullptr;
    a.ktickets = static_cast<unsigned*>(ws);
  }
  FinArgs& f = pl.f;
  f.P = a.P;
  f.tile_loss = a.tile_loss;
  f.Z = Z;
  f.dZ = dZ;
  f.dist_loss = dist_loss;
  f.center_loss = center_loss;
  f.B = B;
  f.L = L;
  f.row0 = row_begin;
  f.row_end = row_end;
  f.nrows_pad = a.nrows_pad;
  f.nslots = nslots;
  f.ntiles = pl.ntiles;
  f.symm = symm;
  f.tile = T;
  const double B2 = (double)B * (double)B;
  const double gl = (double)sig[5] * (std::pow(2.0, (double)sig[4] / sig[5]) - 1.0) / ((double)sig[3] * sig[3]);
  f.gscale = (float)(-4.0 * scale * gl / B2);
  f.lscale = (float)(scale / B2);
  f.cgrad = (float)(2.0 * center_scale / ((double)B * L));
  f.cscale = (float)(center_scale / ((double)B * L));
  pl.nblk_finish = (nrows * L + NT - 1) / NT + 1;
  return EMX_OK;
}

int launch_tiles_plan(const SkPlan& pl, hipStream_t st) {
  const SkArgs& a = pl.a;
  if (a.split.planes != nullptr && pl.T == 64) {
    const int nch = a.split.ld / SPLIT_KC;
    SkArgs b = a;
    b.xcd_nbg = 0;
    int grid = pl.ntiles;
    constexpr bool xcd_order = true;
    if (a.symm && a.nbc >= 16 && xcd_order) {   // XCD-aware tile order (xcd_tile): 8 x the largest share, the surplus blocks leave at once
      b.xcd_nbg = (a.nbc + 3) / 4;
      grid = 8 * b.xcd_nbg * (b.xcd_nbg + 1);
    }
    // few tiles, many chunks (a small batch of very long rows, e.g. the pair-distance vectors of the ADC model): the tiles
    // alone would occupy a handful of CUs -- split each tile's reduction over workgroups (first launch), add the partials
    // up and run the epilogue (second launch)
    constexpr int ksplit_cap = 16;   // (measured at the ADC shape: 32 -> 180.8, 16 -> 178.3, 8 -> 179.8 us per step)
    const int ksplit = (b.kpart && b.xcd_nbg == 0 && nch >= 16) ? std::min(std::min(ksplit_cap, nch / 4), KSPLIT_BLOCKS / std::max(pl.ntiles, 1)) : 1;
    if (ksplit > 1) {
      b.ksplit = ksplit;
      // one launch (kmode 3: the last workgroup of a tile adds the shares and runs the epilogue; its tickets live in memory of the
      // library's, one zeroed block per workspace address: emx_common.h)
      {
        int rz = EMX_OK;
        b.ktickets = emx::owned_tickets(b.ktickets, (size_t)KSPLIT_BLOCKS * sizeof(unsigned), st, "emx_sketchmap_tiles (split reduction)", &rz);
        if (rz) return rz;
      }
      {
        b.kmode = 3;
        const dim3 gk(pl.ntiles * ksplit);
        if (a.L == 2 && pl.fast) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 0>), gk, b);
        else if (a.L == 2) SK_F16_LAUNCH((sk_tile_f16_kernel<2, false, 0>), gk, b);
        else if (pl.fast) SK_F16_LAUNCH((sk_tile_f16_kernel<0, true, 0>), gk, b);
        else SK_F16_LAUNCH((sk_tile_f16_kernel<0, false, 0>), gk, b);
        const int rck = emx::check_launch("sk_tile_f16_kernel (split reduction)");
        if (rck) return rck;
      }
      return EMX_OK;
    }
    // large batches: 128 x 128 macro-tiles writing the 64-tile layout (sk_tile_f16_big_kernel); not for dealt tiles
    constexpr int big_min_b = 8192;
    if (a.L == 2 && a.symm && a.deal_world <= 1 && a.B >= big_min_b && a.row0 == 0 && a.row_end == a.B) {
      const int nbc2 = (a.B + 127) / 128;
      int nbg2 = 0, grid2 = nbc2 * (nbc2 + 1) / 2;
      if (nbc2 >= 16 && xcd_order) {
        nbg2 = (nbc2 + 3) / 4;
        grid2 = 8 * nbg2 * (nbg2 + 1);
      }
#define SK_BIG_LAUNCH(kern) hipLaunchKernelGGL(kern, dim3(grid2), dim3(NT), 0, st, (const uint16_t*)a.split.planes, (long)a.split.plane_stride, nbg2, \
                                               nbc2, a.B, a.split.B, a.split.ld, a)
      if (pl.fast && nch == 8) SK_BIG_LAUNCH((sk_tile_f16_big_kernel<true, 8>));
      else if (pl.fast && nch == 4) SK_BIG_LAUNCH((sk_tile_f16_big_kernel<true, 4>));
      else if (pl.fast) SK_BIG_LAUNCH((sk_tile_f16_big_kernel<true, 0>));
      else SK_BIG_LAUNCH((sk_tile_f16_big_kernel<false, 0>));
#undef SK_BIG_LAUNCH
      return emx::check_launch("sk_tile_f16_big_kernel");
    }
    if (a.L == 2 && pl.fast && nch == 4) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 4>), dim3(grid), b);
    else if (a.L == 1 && pl.fast && nch == 4) SK_F16_LAUNCH((sk_tile_f16_kernel<1, true, 4>), dim3(grid), b);   // (a one-column latent at compile time: the run-time width
    else if (a.L == 1 && pl.fast && nch == 8) SK_F16_LAUNCH((sk_tile_f16_kernel<1, true, 8>), dim3(grid), b);   //  instantiation runs four tiles per CU, 24.6 against 17 us at the C2 shape)
    else if (a.L == 3 && pl.fast && nch == 4) SK_F16_LAUNCH((sk_tile_f16_kernel<3, true, 4>), dim3(grid), b);   // three latent columns at compile time (the run-time
    else if (a.L == 3 && pl.fast && nch == 8) SK_F16_LAUNCH((sk_tile_f16_kernel<3, true, 8>), dim3(grid), b);   //  width instantiation below needs 102 registers: four tiles per CU)
    else if (a.L == 4 && pl.fast && nch == 4) SK_F16_LAUNCH((sk_tile_f16_kernel<4, true, 4>), dim3(grid), b);
    else if (a.L == 4 && pl.fast && nch == 8) SK_F16_LAUNCH((sk_tile_f16_kernel<4, true, 8>), dim3(grid), b);
    else if (a.L == 2 && pl.fast && nch == 8) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 8>), dim3(grid), b);
    else if (a.L == 2 && pl.fast && nch == 2) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 2>), dim3(grid), b);
    else if (a.L == 2 && pl.fast && nch == 1) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 1>), dim3(grid), b);
    else if (a.L == 2 && pl.fast) SK_F16_LAUNCH((sk_tile_f16_kernel<2, true, 0>), dim3(grid), b);
    else if (a.L == 2) SK_F16_LAUNCH((sk_tile_f16_kernel<2, false, 0>), dim3(grid), b);
    else if (pl.fast) SK_F16_LAUNCH((sk_tile_f16_kernel<0, true, 0>), dim3(grid), b);
    else SK_F16_LAUNCH((sk_tile_f16_kernel<0, false, 0>), dim3(grid), b);
    return emx::check_launch("sk_tile_f16_kernel");
  }
  if (!pl.periodic && a.vec4 && pl.T == 64 && pl.gram && a.L <= 4) {
    if (a.L == 2) {
      if (pl.fast) hipLaunchKernelGGL((sk_tile_mfma_kernel<2, true>), dim3(pl.ntiles), dim3(NT), 0, st, a);
      else hipLaunchKernelGGL((sk_tile_mfma_kernel<2, false>), dim3(pl.ntiles), dim3(NT), 0, st, a);
    } else {
      if (pl.fast) hipLaunchKernelGGL((sk_tile_mfma_kernel<0, true>), dim3(pl.ntiles), dim3(NT), 0, st, a);
      else hipLaunchKernelGGL((sk_tile_mfma_kernel<0, false>), dim3(pl.ntiles), dim3(NT), 0, st, a);
    }
    return emx::check_launch("sk_tile_mfma_kernel");
  }
  if (a.split.planes != nullptr && pl.T == 32 && !pl.periodic && a.L > 4 && a.L <= 16 && a.deal_world <= 1 && a.B >= 64) {
    // latent widths 5 .. 16 on split rows: the Gram tiles of the f16 matrix cores at 64 x 64, writing the 32-tile layout every
    // consumer of these widths plans with (sk_tile_f16_kernel: OUT32) -- the 32 x 32 VALU tiles they replace cost 7x the two-column
    // pass at 4096 rows (113-126 us against 17).  Not for dealt tiles: the deal is by 32-tile.
    const int nch = a.split.ld / SPLIT_KC;
    const int nbc64 = (a.B + 63) / 64, nbr64 = (a.row_end - a.row0 + 63) / 64;
    int grid = a.symm ? nbc64 * (nbc64 + 1) / 2 : nbr64 * nbc64;
    int xnbg = 0;
    constexpr bool xcd_order = true;
    if (a.symm && nbc64 >= 16 && xcd_order) {
      xnbg = (nbc64 + 3) / 4;
      grid = 8 * xnbg * (xnbg + 1);
    }
#define SK_F16_LAUNCH32(kern)                                                                                                                      \
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, st, (const uint16_t*)a.split.planes, (long)a.split.plane_stride, xnbg, nbc64, a.symm, 1, 0, a.row0, \
                     a.B, a.split.B, a.split.ld, 0, a)
    // (five and six columns: the six-column instantiation -- the epilogue's latent part is a third of its instructions at this width)
    // (nine to sixteen columns: the epilogue walks the latent eight columns at a time, LT = 16)
    if (a.L > 8 && a.L <= 12 && pl.fast && nch == 4) SK_F16_LAUNCH32((sk_tile_f16_kernel<12, true, 4>));   // (chunks of eight and four columns)
    else if (a.L > 8 && a.L <= 12 && pl.fast && nch == 8) SK_F16_LAUNCH32((sk_tile_f16_kernel<12, true, 8>));
    else if (a.L > 8 && pl.fast && nch == 4) SK_F16_LAUNCH32((sk_tile_f16_kernel<16, true, 4>));
    else if (a.L > 8 && pl.fast && nch == 8) SK_F16_LAUNCH32((sk_tile_f16_kernel<16, true, 8>));
    else if (a.L > 8 && pl.fast) SK_F16_LAUNCH32((sk_tile_f16_kernel<16, true, 0>));
    else if (a.L > 8) SK_F16_LAUNCH32((sk_tile_f16_kernel<16, false, 0>));
    else if (a.L <= 6 && pl.fast && nch == 4) SK_F16_LAUNCH32((sk_tile_f16_kernel<6, true, 4>));
    else if (a.L <= 6 && pl.fast && nch == 8) SK_F16_LAUNCH32((sk_tile_f16_kernel<6, true, 8>));
    else if (pl.fast && nch == 4) SK_F16_LAUNCH32((sk_tile_f16_kernel<8, true, 4>));
    else if (pl.fast && nch == 8) SK_F16_LAUNCH32((sk_tile_f16_kernel<8, true, 8>));
    else if (pl.fast) SK_F16_LAUNCH32((sk_tile_f16_kernel<8, true, 0>));
    else SK_F16_LAUNCH32((sk_tile_f16_kernel<8, false, 0>));
#undef SK_F16_LAUNCH32
    return emx::check_launch("sk_tile_f16_kernel (latent widths 5 .. 16)");
  }
  if (pl.T == 32) {
    if (a.L == 2) dispatch2<32, 2>(a, pl.ntiles, pl.periodic, pl.fast, st);
    else if (a.L > 8) dispatch2<32, SK_LT_ANY>(a, pl.ntiles, pl.periodic, pl.fast, st);   // any width, eight columns at a time
    else if (a.L > 4) dispatch2<32, 8>(a, pl.ntiles, pl.periodic, pl.fast, st);
    else dispatch2<32, 0>(a, pl.ntiles, pl.periodic, pl.fast, st);
  } else if (pl.T != 64) {
    return emx::fail(EMX_EUNSUPPORTED, "emx_sketchmap_tiles: no %d x %d tile kernel", pl.T, pl.T);
  } else {
    SkArgs b = a;
    int grid = pl.ntiles;
    constexpr bool xcd_order = true;
    if (a.symm && a.nbc >= 16 && xcd_order) {   // as for the f16 tiles: 8 x the largest share, surplus blocks leave at once
      b.xcd_nbg = (a.nbc + 3) / 4;
      grid = 8 * b.xcd_nbg * (b.xcd_nbg + 1);
    }
    if (a.L == 2) dispatch2<64, 2>(b, grid, pl.periodic, pl.fast, st);
    else dispatch2<64, 0>(b, grid, pl.periodic, pl.fast, st);
  }
  return emx::check_launch("sk_tile_kernel");
}

int launch_finish_plan(const SkPlan& pl, hipStream_t st) {
  hipLaunchKernelGGL(sk_finish_kernel, dim3(pl.nblk_finish), dim3(NT), 0, st, pl.f);
  return emx::check_launch("sk_finish_kernel");
}

}  // namespace

// The plan of emx_sketchmap_tiles (full batch, Euclidean, split rows) as the description of the tile ROLE of the fused
// dense-layer launch (sk_tile.h, mlp.hip: emx_mlp_forward_backward_tiles).  EMX_EUNSUPPORTED outside the shapes that role is
// built for: the caller keeps the separate launches.
namespace emx {
namespace sk {
int make_tile_role(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span, const float sig[6], void* ws,
                   size_t ws_bytes, void* split_ws, TileRole* out, int* virtual_blocks, int* n_chunks, bool* fast, int* kind, const char* who) {
  EMX_REQUIRE(out && virtual_blocks && n_chunks && fast && kind, "%s: null pointer argument (tile role)", who);
  SkPlan pl;
  int rc = make_plan(X, B, D, ldx, Z, L, periodicity, sig, 1.f, 0.f, 0, B, nullptr, nullptr, nullptr, ws, ws_bytes, pl, who, split_ws, data_span);
  if (rc) return rc;
  out->a = pl.a;
  out->a.xcd_nbg = 0;
  *virtual_blocks = pl.ntiles;
  *fast = pl.fast;
  if (pl.periodic) {
    // periodic rows: the 32 x 32 VALU tiles (batches of up to 512 rows, pick_tile)
    if (L != 2 || pl.T != 32) return emx::fail(EMX_EUNSUPPORTED, "%s: the periodic tile role runs two-column latents on 32 x 32 tiles (batches of up to 512 rows)", who);
    *kind = pl.a.phase_ok ? 2 : 1;
    *n_chunks = 0;
    return EMX_OK;
  }
  if (L != 2 || pl.T != 64 || pl.a.split.planes == nullptr) return emx::fail(EMX_EUNSUPPORTED, "%s: the tile role runs two-column latents on split rows", who);
  *kind = 0;
  constexpr bool xcd_order = true;
  if (pl.a.nbc >= 16 && xcd_order) {   // (as launch_tiles_plan)
    out->a.xcd_nbg = (pl.a.nbc + 3) / 4;
    *virtual_blocks = 8 * out->a.xcd_nbg * (out->a.xcd_nbg + 1);
  }
  *n_chunks = pl.a.split.ld / SPLIT_KC;
  return EMX_OK;
}
}  // namespace sk
}  // namespace emx

__global__ __launch_bounds__(NT) void sk_split_kernel(const float* __restrict__ X, int B, int D, int ldx, const SplitRows sp) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * NT + threadIdx.x) >> 6, nwaves = (gridDim.x * NT) >> 6;
  for (int b = wave; b < B; b += nwaves) split_row(sp, X + (long)b * ldx, X, D, b, lane);
}

// long rows (thousands of features): one workgroup per row instead of one wave -- the two passes over a row are a chain of
// dependent round trips per lane, and a small batch of long rows leaves most of the chip idle with one wave per row
constexpr int SPW_NT = 1024;   // (a 256-row batch is 256 workgroups: each brings a full CU's worth of waves)
__global__ __launch_bounds__(SPW_NT) void sk_split_wide_kernel(const float* __restrict__ X, int D, int ldx, const SplitRows s) {
  __shared__ float red_n[SPW_NT / 64], red_m[SPW_NT / 64];
  const long b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* __restrict__ src = X + b * ldx;
  float nrm = 0.f, mx = 0.f;
  for (int k = tid; k < D; k += SPW_NT) {
    const float v = src[k] - X[k];
    nrm = fmaf(v, v, nrm);
    mx = fmaxf(mx, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  nrm = wave_sum(nrm);
  if ((tid & 63) == 0) {
    red_n[tid >> 6] = nrm;
    red_m[tid >> 6] = mx;
  }
  __syncthreads();
  nrm = 0.f;
  mx = 0.f;
#pragma unroll
  for (int w = 0; w < SPW_NT / 64; w += 4) {   // fixed order
    nrm += (red_n[w] + red_n[w + 1]) + (red_n[w + 2] + red_n[w + 3]);
    mx = fmaxf(mx, fmaxf(fmaxf(red_m[w], red_m[w + 1]), fmaxf(red_m[w + 2], red_m[w + 3])));
  }
  // (same scale rule as emx::split_row)
  int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;
  E = mx > 0.f ? min(max(E, -100), 100) : 14;
  const float sc = __uint_as_float((uint32_t)(127 + 14 - E) << 23), rs = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
  for (int k = tid; k < s.ld; k += SPW_NT) {
    const float v = k < D ? (src[k] - X[k]) * sc : 0.f;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    uint16_t* o = s.planes + emx::split_index(s, b, k);
    o[0] = emx::f16_bits(h);
    o[s.plane_stride] = emx::f16_bits(l);
  }
  if (tid == 0) {
    s.norms[b] = nrm;
    s.rscale[b] = rs;
  }
}

extern "C" {

size_t emx_sketchmap_split_bytes(int B, int D) { return (B > 0 && D > 0) ? split_bytes(B, D) : 0; }

int emx_sketchmap_split(const float* X, int B, int D, int ldx, void* split_ws, emx_stream_t stream) {
  EMX_REQUIRE(X && split_ws && B > 0 && D > 0 && ldx >= D, "emx_sketchmap_split: bad arguments");
  EMX_REQUIRE((reinterpret_cast<uintptr_t>(split_ws) & 15) == 0, "emx_sketchmap_split: workspace must be 16-byte aligned");
  const SplitRows sp = split_view(split_ws, B, D);
  if (D >= 1024 && B <= 8192) {
    hipLaunchKernelGGL(sk_split_wide_kernel, dim3((unsigned)B), dim3(SPW_NT), 0, (hipStream_t)stream, X, D, ldx, sp);
    return emx::check_launch("sk_split_wide_kernel");
  }
  hipLaunchKernelGGL(sk_split_kernel, dim3((unsigned)std::min((B + 3) / 4, 4096)), dim3(NT), 0, (hipStream_t)stream, X, B, D, ldx, sp);
  return emx::check_launch("sk_split_kernel");
}

int emx_sketchmap_fwd_bwd(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span,
                          const float sig[6], float scale, float center_scale, int row_begin, int row_end,
                          float* dist_loss, float* center_loss, float* dZ, void* ws, size_t ws_bytes,
                          emx_stream_t stream) {
  EMX_REQUIRE(dist_loss && dZ, "emx_sketchmap_fwd_bwd: null pointer argument");
  SkPlan pl;
  int rc = make_plan(X, B, D, ldx, Z, L, periodicity, sig, scale, center_scale, row_begin, row_end, dist_loss, center_loss, dZ,
                     ws, ws_bytes, pl, "emx_sketchmap_fwd_bwd", nullptr, data_span);
  if (rc) return rc;
  rc = launch_tiles_plan(pl, (hipStream_t)stream);
  if (rc) return rc;
  return launch_finish_plan(pl, (hipStream_t)stream);
}

int emx_sketchmap_tiles(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span,
                        const float sig[6], int row_begin, int row_end, void* ws, size_t ws_bytes, void* split_ws,
                        emx_stream_t stream) {
  SkPlan pl;
  int rc = make_plan(X, B, D, ldx, Z, L, periodicity, sig, 1.f, 0.f, row_begin, row_end, nullptr, nullptr, nullptr, ws, ws_bytes,
                     pl, "emx_sketchmap_tiles", split_ws, data_span);
  if (rc) return rc;
  return launch_tiles_plan(pl, (hipStream_t)stream);
}

int emx_sketchmap_finish(int B, int D, const float* Z, int L, float periodicity, const float sig[6], float scale, float center_scale,
                         int row_begin, int row_end, float* dist_loss, float* center_loss, float* dZ, void* ws,
                         size_t ws_bytes, emx_stream_t stream) {
  EMX_REQUIRE(dist_loss && dZ, "emx_sketchmap_finish: null pointer argument");
  SkPlan pl;
  int rc = make_plan(Z /* X unused by the finish pass */, B, D, D, Z, L, periodicity, sig, scale, center_scale, row_begin, row_end,
                     dist_loss, center_loss, dZ, ws, ws_bytes, pl, "emx_sketchmap_finish");
  if (rc) return rc;
  return launch_finish_plan(pl, (hipStream_t)stream);
}

/* Data parallel, symmetric: the all-pairs pass over the WHOLE batch with its upper-triangle tiles dealt
 * to the ranks -- every rank computes 1 / world of the pair work (the rectangle rows x [0, B) of a row-block rank is twice
 * that) and keeps row AND column partials of its tiles. */
int emx_sketchmap_tiles_dealt(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span,
                              const float sig[6], void* ws, size_t ws_bytes, void* split_ws, int rank, int world, emx_stream_t stream) {
  EMX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "emx_sketchmap_tiles_dealt: bad rank %d of %d", rank, world);
  SkPlan pl;
  int rc = make_plan(X, B, D, ldx, Z, L, periodicity, sig, 1.f, 0.f, 0, B, nullptr, nullptr, nullptr, ws, ws_bytes, pl,
                     "emx_sketchmap_tiles_dealt", split_ws, data_span);
  if (rc) return rc;
  pl.a.deal_rank = rank;
  pl.a.deal_world = world;
  return launch_tiles_plan(pl, (hipStream_t)stream);
}

int emx_sketchmap_reduce_dealt(int B, int D, const float* Z, int L, float periodicity, const float sig[6], float scale, float center_scale,
                               int own_begin, int own_end, int rank, int world, float* dZ, void* ws, size_t ws_bytes, emx_stream_t stream) {
  EMX_REQUIRE(dZ && world >= 1 && rank >= 0 && rank < world, "emx_sketchmap_reduce_dealt: bad arguments");
  EMX_REQUIRE(0 <= own_begin && own_begin <= own_end && own_end <= B, "emx_sketchmap_reduce_dealt: bad own rows");
  SkPlan pl;
  int rc = make_plan(Z, B, D, D, Z, L, periodicity, sig, scale, center_scale, 0, B, nullptr, nullptr, nullptr, ws, ws_bytes, pl,
                     "emx_sketchmap_reduce_dealt");
  if (rc) return rc;
  DealtArgs a{pl.f.P, Z, dZ, B, L, pl.a.nbc, pl.f.nrows_pad, pl.T, rank, world, own_begin, own_end, pl.f.gscale, pl.f.cgrad};
  hipLaunchKernelGGL(sk_reduce_dealt_kernel, dim3((unsigned)((B * L + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, a);
  return emx::check_launch("sk_reduce_dealt_kernel");
}

int emx_sketchmap_describe(int B, int D, int ldx, const float* X, const float* Z, int L, float periodicity, const float sig[6],
                           float scale, float center_scale, int row_begin, int row_end, void* ws, size_t ws_bytes,
                           emx_sketchmap_partials* out) {
  EMX_REQUIRE(out, "emx_sketchmap_describe: null output");
  SkPlan pl;
  int rc = make_plan(X, B, D, ldx, Z, L, periodicity, sig, scale, center_scale, row_begin, row_end, nullptr, nullptr, nullptr, ws, ws_bytes,
                     pl, "emx_sketchmap_describe");
  if (rc) return rc;
  out->grad.partials = pl.f.P;
  out->grad.Z = Z;
  out->grad.slot_stride = (int64_t)pl.f.nrows_pad * L;
  out->grad.n_slots = pl.f.nslots;
  out->grad.row_base = row_begin;
  out->grad.gscale = pl.f.gscale;
  out->grad.center_grad = pl.f.cgrad;
  out->tile_loss = pl.f.tile_loss;
  out->n_tiles = pl.ntiles;
  out->loss_scale = pl.f.lscale;
  out->center_loss_scale = pl.f.cscale;
  return EMX_OK;
}

int emx_pairwise_dist(const float* X, int B, int D, int ldx, float periodicity, float* out, emx_stream_t stream) {
  EMX_REQUIRE(X && out && B > 0 && D > 0 && ldx >= D, "emx_pairwise_dist: bad arguments");
  dim3 grid((B + 255) / 256, B);
  hipLaunchKernelGGL(pairwise_dist_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, B, D, ldx, periodicity,
                     std::isfinite(periodicity) ? 1 : 0, out);
  return emx::check_launch("pairwise_dist_kernel");
}

}  // extern "C"

#ifdef EMX_CHAIN_TIMING
extern "C" int emx_debug_read_sk(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg_sk), sizeof(emx_dbg_sk)); }
#endif
