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
#include "emx_common.h"

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

// tanh with |error| <= ~1e-7 (the float32 resolution of the activations): odd polynomial near zero, 1 - 2/(e^{2x}+1)
// elsewhere.  ~14 VALU instructions instead of the ~50 of the library tanhf, which matters at one wave per SIMD.
__device__ __forceinline__ float tanh_f32(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float p = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * ax);        // e^{2|x|}
  const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  return ax < 0.25f ? p : copysignf(t, x);
}

constexpr int UQ = 8;   // K-steps (16 k each) per pipeline unit: 32 B-fragment registers in flight per unit

// One pipeline unit = (layer, 32-column pass of this wave, chunk of UQ K-steps).  Units are walked in order across
// layer boundaries so that the B fragments (weights, independent of the activations) of the NEXT unit are always in
// flight while the current unit's MFMAs, epilogue and barrier run.
struct Unit {
  int l, n0, c;
  bool valid;
};

template <bool BWD>
__device__ __forceinline__ int unit_cols(const MlpDev& d, int l) {   // width of the GEMM's output (N for fwd, Kin for bwd)
  return BWD ? d.width[l] : d.width[l + 1];
}
template <bool BWD>
__device__ __forceinline__ int unit_kq(const MlpDev& d, int l) {     // K-steps of the GEMM's reduction axis
  return ((BWD ? d.width[l + 1] : d.width[l]) + 15) >> 4;
}

// forward walks layers upward, backward walks them downward (and skips the data GEMM of layer lo)
template <bool BWD>
__device__ __forceinline__ Unit first_unit(const MlpDev& d, int l, int lo, int hi, int wave) {
  Unit u{l, wave * 16, 0, false};
  while (BWD ? (u.l > lo) : (u.l < hi)) {
    if (u.n0 < unit_cols<BWD>(d, u.l)) {
      u.valid = true;
      return u;
    }
    u.l += BWD ? -1 : 1;
  }
  return u;
}
template <bool BWD>
__device__ __forceinline__ Unit next_unit(const MlpDev& d, Unit u, int lo, int hi, int wave) {
  const int nch = (unit_kq<BWD>(d, u.l) + UQ - 1) / UQ;
  if (u.c + 1 < nch) return Unit{u.l, u.n0, u.c + 1, true};
  if (u.n0 + 128 < unit_cols<BWD>(d, u.l)) return Unit{u.l, u.n0 + 128, 0, true};
  return first_unit<BWD>(d, u.l + (BWD ? -1 : 1), lo, hi, wave);
}

// B fragments of a forward unit: W[k][n], k = 16(UQ c + q) + 4g + i, n = n0 + 16t + m.  Uniform row base + per-lane
// offset (-> global_load with an SGPR base); rows are clamped only in the ragged last chunk.
__device__ __forceinline__ void load_unit_fwd(const MlpDev& d, const Unit u, int m, int g, float (&b)[UQ][4]) {
  const int K = d.width[u.l], N = d.width[u.l + 1];
  const float* __restrict__ W = d.W[u.l];
  const int off0 = 4 * g * N + min(u.n0 + m, N - 1);
  const int k0 = 16 * UQ * u.c;
  if (k0 + 16 * UQ <= K) {
#pragma unroll
    for (int q = 0; q < UQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* rb = W + (long)(k0 + 16 * q + i) * N;   // wave-uniform
        b[q][i] = rb[off0];
      }
  } else {
    const int c0 = min(u.n0 + m, N - 1);
#pragma unroll
    for (int q = 0; q < UQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* rb = W + (long)min(k0 + 16 * q + 4 * g + i, K - 1) * N;   // clamped: the A side is zero there
        b[q][i] = rb[c0];
      }
  }
}

// B fragments of a backward-data unit: 16-byte pieces of W rows: W[j][k..k+3], j = n0 + 16t + m (input feature),
// k = 16(UQ c + q) + 4g (output feature)
__device__ __forceinline__ void load_unit_bwd(const MlpDev& d, const Unit u, int m, int g, float (&b)[UQ][4]) {
  const int Kin = d.width[u.l], N = d.width[u.l + 1];
  const float* __restrict__ W = d.W[u.l];
  const float* p0 = W + (long)min(u.n0 + m, Kin - 1) * N;
  const int k0 = 16 * UQ * u.c + 4 * g;
  if ((N & 3) == 0) {   // wave-uniform
#pragma unroll
    for (int q = 0; q < UQ; ++q) {
      const int kk = min(k0 + 16 * q, N - 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p0 + kk);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b[q][i] = w0[i];
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < UQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = min(k0 + 16 * q + i, N - 1);
        b[q][i] = p0[kk];
      }
  }
}

__device__ __forceinline__ void mma_unit(const float* src, int pitch, int m, int g, int c, int Kq, const float (&b)[UQ][4], f32x4& acc0) {
#pragma unroll
  for (int q = 0; q < UQ; ++q) {
    const int kq = UQ * c + q;
    if (kq < Kq) {   // wave-uniform
      const f32x4 a = *reinterpret_cast<const f32x4*>(&src[m * pitch + 16 * kq + 4 * g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[q][i], acc0, 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: layers [l0, l1) on rows [row0, row_end); `in` is [*, width[l0]] with row stride ld_in
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTC) void mlp_forward_kernel(const MlpDev d, int l0, int l1, const float* __restrict__ in, int ld_in,
                                                         int row0, int row_end, int pitch) {
  extern __shared__ __align__(16) float lds[];
  // (LDS images are addressed as lds + offset everywhere: a pointer array would decay to generic pointers and
  //  flat_load, whose out-of-order return forces s_waitcnt vmcnt(0) in front of every MFMA group)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int r0 = row0 + blockIdx.x * R;

  float bcur[UQ][4], bnxt[UQ][4];
  Unit cur = first_unit<false>(d, l0, l0, l1, wave);
  if (cur.valid) load_unit_fwd(d, cur, m, g, bcur);

  // stage the input rows (zero padded to the pitch, rows beyond the block zero)
  {
    const int K = d.width[l0];
    for (int e = tid; e < R * pitch; e += NTC) {
      int r = e / pitch, c = e - r * pitch;
      float v = 0.f;
      if (c < K && r0 + r < row_end) v = in[(long)(r0 + r) * ld_in + c];
      lds[e] = v;
    }
  }
  __syncthreads();

  int buf = 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
  for (int l = l0; l < l1; ++l) {
    const int N = d.width[l + 1];
    const int Kq = (d.width[l] + 15) >> 4;
    const int nch = (Kq + UQ - 1) / UQ;
    const float* src = lds + buf * (R * pitch);
    float* dst = lds + (buf ^ 1) * (R * pitch);
    while (cur.valid && cur.l == l) {
      const Unit nxt = next_unit<false>(d, cur, l0, l1, wave);
      if (nxt.valid) load_unit_fwd(d, nxt, m, g, bnxt);
      mma_unit(src, pitch, m, g, cur.c, Kq, bcur, acc0);
      if (cur.c == nch - 1) {
        // epilogue: bias, activation, to LDS (next layer's input, zero beyond N) and to HBM
        const float* __restrict__ bias = d.b[l];
        float* hout = d.h[l];
        const bool act = d.act[l] != 0;
        {
          const int col = cur.n0 + m;
          const float bv = bias[min(col, N - 1)];
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * g + reg;
            float v = acc0[reg] + bv;
            if (act) v = tanh_f32(v);
            v = col < N ? v : 0.f;
            dst[row * pitch + col] = v;   // col < round16(N) <= pitch - 4
            if (col < N && r0 + row < row_end) hout[(long)(r0 + row) * N + col] = v;
          }
        }
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < UQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) bcur[q][i] = bnxt[q][i];
      cur = nxt;
    }
    __syncthreads();
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward (data): for l = l1-1 .. l0:  delta_l = dh_l * act'(h_l)  (stored),  dh_{l-1} = delta_l W_l^T.
// `dtop` [*, width[l1]] (row stride ld_top) is d loss / d h_{l1-1}; `dadd` (may be null) [*, width[ladd+1]] is
// added to dh_ladd before its activation derivative (the latent gradient of the distance/center costs).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTC) void mlp_backward_kernel(const MlpDev d, int l0, int l1, const float* __restrict__ dtop, int ld_top,
                                                          const float* __restrict__ dadd, int ladd, const emx_latent_grad lg,
                                                          float* __restrict__ dbelow, int row0, int row_end, int pitch) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int r0 = row0 + blockIdx.x * R;

  float bcur[UQ][4], bnxt[UQ][4];
  const int lo = dbelow ? l0 - 1 : l0;   // with `dbelow` the data GEMM of layer l0 runs too (dh below the range is stored)
  Unit cur = first_unit<true>(d, l1 - 1, lo, l1, wave);
  if (cur.valid) load_unit_bwd(d, cur, m, g, bcur);

  // dh of the top layer -> LDS
  {
    const int N = d.width[l1];
    for (int e = tid; e < R * pitch; e += NTC) {
      int r = e / pitch, c = e - r * pitch;
      float v = 0.f;
      if (c < N && r0 + r < row_end) v = dtop[(long)(r0 + r) * ld_top + c];
      lds[e] = v;
    }
  }
  __syncthreads();

  int buf = 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
  for (int l = l1 - 1; l >= l0; --l) {
    const int Kin = d.width[l], N = d.width[l + 1];  // layer maps Kin -> N
    float* dh = lds + buf * (R * pitch);
    // delta_l = (dh_l [+ dadd]) * act'(h_l), in place in LDS, and stored to HBM for the weight gradients
    {
      const float* __restrict__ h = d.h[l];
      float* __restrict__ dl = d.delta[l];
      const bool act = d.act[l] != 0;
      const bool add = dadd != nullptr && l == ladd;
      for (int e = tid; e < R * N; e += NTC) {
        int r = e / N, c = e - r * N;
        float v = dh[r * pitch + c];
        const bool ok = r0 + r < row_end;
        if (ok) {
          const long gi = (long)(r0 + r) * N + c;
          if (add) v += dadd[gi];
          if (lg.partials != nullptr && l == ladd) {
            // latent gradient of the all-pairs cost: fixed-order sum of the per-tile slots (8 loads in flight)
            const float* __restrict__ P = lg.partials + (long)(r0 + r - lg.row_base) * N + c;
            float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int sidx = 0;
            for (; sidx + 8 <= lg.n_slots; sidx += 8) {
#pragma unroll
              for (int u = 0; u < 8; ++u) part[u] += P[(long)(sidx + u) * lg.slot_stride];
            }
            for (int u = 0; sidx < lg.n_slots; ++sidx, ++u) part[u] += P[(long)sidx * lg.slot_stride];
            const float ssum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            v += fmaf(lg.center_grad, lg.Z[gi], lg.gscale * ssum);
          }
          if (act) {
            float hv = h[gi];
            v *= fmaf(-hv, hv, 1.f);
          }
          dl[gi] = v;
        } else {
          v = 0.f;
        }
        dh[r * pitch + c] = v;
      }
    }
    __syncthreads();
    if (l == l0 && !dbelow) break;  // no dh needed below the first layer of the range
    // dh_{l-1}[row][j] = sum_k delta[row][k] W[j][k]  (k over the N outputs, j over the Kin inputs)
    const int Kq = (N + 15) >> 4;
    const int nch = (Kq + UQ - 1) / UQ;
    const float* src = lds + buf * (R * pitch);
    float* dst = lds + (buf ^ 1) * (R * pitch);
    while (cur.valid && cur.l == l) {
      const Unit nxt = next_unit<true>(d, cur, lo, l1, wave);
      if (nxt.valid) load_unit_bwd(d, nxt, m, g, bnxt);
      mma_unit(src, pitch, m, g, cur.c, Kq, bcur, acc0);
      if (cur.c == nch - 1) {
        {
          const int col = cur.n0 + m;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float v = acc0[reg];
            dst[(4 * g + reg) * pitch + col] = col < Kin ? v : 0.f;
            if (l == l0 && col < Kin && r0 + 4 * g + reg < row_end) dbelow[(long)(r0 + 4 * g + reg) * Kin + col] = v;
          }
        }
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < UQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) bcur[q][i] = bnxt[q][i];
      cur = nxt;
    }
    __syncthreads();
    buf ^= 1;
  }
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
  int tile_begin[MAXL + 1];  // prefix sum of 128x128 output blocks per layer (relative to l0)
  float* partial;            // [n_slices][n_params]
  long n_params;
  int row0, row_end, rows_per_slice;
};

__global__ __launch_bounds__(NT) void mlp_wgrad_kernel(const WgradDev a) {
  // One workgroup = one 128x128 block of one layer's dW over one row slice; each wave a 64x64 quadrant = 4x4 MFMA
  // tiles.  The MFMA's m/n indices are free labels: lane slot m of tile u stands for feature 4m + u (n, v likewise), so
  // ONE 16-byte load per operand and row feeds all four tiles of that operand (2 loads per 16 MFMAs), and the four
  // tiles (u, v = 0..3) of a slot pair form a float4 of adjacent output columns for the store.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  int l = a.l0;
  while (l + 1 < a.l1 && (int)blockIdx.x >= a.tile_begin[l + 1 - a.l0]) ++l;
  const int tb = blockIdx.x - a.tile_begin[l - a.l0];
  const int Kin = a.d.width[l], N = a.d.width[l + 1];
  const int nbn = (N + 127) >> 7;
  const int bi = tb / nbn, bj = tb - bi * nbn;
  const int i0 = bi * 128 + (wave >> 1) * 64;   // input-feature offset of this wave's quadrant
  const int j0 = bj * 128 + (wave & 1) * 64;    // output-feature offset
  const float* __restrict__ X = (l == a.l0) ? a.xin : a.d.h[l - 1];
  const int ldx = (l == a.l0) ? a.ld_xin : Kin;
  const float* __restrict__ Dl = a.d.delta[l];
  const int rs = a.row0 + blockIdx.y * a.rows_per_slice;
  const int re = min(rs + a.rows_per_slice, a.row_end);
  float* __restrict__ out = a.partial + (long)blockIdx.y * a.n_params;

  f32x4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (bi == 0) && ((wave >> 1) == 0);
  const bool quad_active = i0 < Kin && j0 < N;   // wave-uniform

  // 16-byte operand loads need 4 | width and an aligned row stride; otherwise four clamped scalar loads
  const bool vx = ((ldx & 3) == 0) && ((Kin & 3) == 0), vd = (N & 3) == 0;
  const int cx = i0 + 4 * m, cd = j0 + 4 * m;
  auto ldrow = [&](const float* __restrict__ base, int ld, int width, int col, bool vec, int row) -> f32x4 {
    const float* p = base + (long)row * ld;
    f32x4 v;
    if (vec) {
      v = *reinterpret_cast<const f32x4*>(p + min(col, width - 4));   // clamped: out-of-range slots are discarded at the store
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = p[min(col + i, width - 1)];
    }
    return v;
  };
  constexpr int US = 4;   // 4-row steps per batch; the next batch's loads are issued before this batch's 64 MFMAs
  if (quad_active) {
    f32x4 xa[US], db[US], xn[US], dn[US];
    auto load_batch = [&](int r, f32x4 (&xo)[US], f32x4 (&dout)[US]) {
#pragma unroll
      for (int s2 = 0; s2 < US; ++s2) {
        const int row = min(r + 4 * s2 + g, re - 1);
        xo[s2] = ldrow(X, ldx, Kin, cx, vx, row);
        dout[s2] = ldrow(Dl, N, N, cd, vd, row);
      }
    };
    load_batch(rs, xn, dn);
    for (int r = rs; r < re; r += 4 * US) {
#pragma unroll
      for (int s2 = 0; s2 < US; ++s2) {
        xa[s2] = xn[s2];
        db[s2] = dn[s2];
      }
      if (r + 4 * US < re) load_batch(r + 4 * US, xn, dn);
      if (r + 4 * US > re) {   // ragged tail (wave-uniform): rows beyond the slice contribute nothing
#pragma unroll
        for (int s2 = 0; s2 < US; ++s2) {
          const float ok = (r + 4 * s2 + g < re) ? 1.f : 0.f;
          db[s2] *= ok;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < US; ++s2) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s2][u], db[s2][v], acc[u][v], 0, 0, 0);
        bsum += db[s2];
      }
    }
    // store: tile (u, v), D[row slot 4g + reg][col slot m] -> dW[i0 + 4(4g + reg) + u][j0 + 4m + v]
    const bool vst = (N & 3) == 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ii = i0 + 4 * (4 * g + reg) + u, jj = j0 + 4 * m;
        if (ii < Kin && jj < N) {
          float* q = out + a.w_off[l] + (long)ii * N + jj;
          if (vst && ((a.w_off[l] & 3) == 0)) {
            *reinterpret_cast<f32x4*>(q) = f32x4{acc[u][0][reg], acc[u][1][reg], acc[u][2][reg], acc[u][3][reg]};
          } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (jj + v < N) q[v] = acc[u][v][reg];
          }
        }
      }
  }
  if (do_bias && j0 < N) {
    // column sums: lane (m, g) holds the sums of columns j0 + 4m .. + 3 over the rows of its lane group
    f32x4 sfull = bsum;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float t = sfull[v];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      sfull[v] = t;
    }
    if (g == 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (cd + v < N) out[a.b_off[l] + cd + v] = sfull[v];
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

int pitch_for(const emx_mlp_desc* desc, int l0, int l1) {
  int w = 0;
  for (int l = l0; l <= l1; ++l) w = std::max(w, desc->width[l]);
  return ((w + 15) / 16) * 16 + 4;   // tiles write up to the next multiple of 16; +4 keeps b128 rows conflict-free
}

int set_lds(const void* fn, size_t bytes, const char* who) {
  if (bytes > 160 * 1024) return emx::fail(EMX_EUNSUPPORTED, "%s: layer width too large for the LDS row block (%zu bytes)", who, bytes);
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return emx::fail(EMX_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, bytes, hipGetErrorString(e));
  }
  return EMX_OK;
}

}  // namespace

extern "C" {

int emx_mlp_forward(const emx_mlp_desc* desc, float* params, int layer_begin, int layer_end, const float* in, int ld_in,
                    int row_begin, int row_end, float* const* h, emx_stream_t stream) {
  MlpDev d;
  int rc = fill_dev(desc, params, h, nullptr, d, "emx_mlp_forward");
  if (rc) return rc;
  EMX_REQUIRE(in && h, "emx_mlp_forward: null pointer argument");
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "emx_mlp_forward: bad layer range");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_in >= desc->width[layer_begin], "emx_mlp_forward: bad rows / stride");
  for (int l = layer_begin; l < layer_end; ++l) EMX_REQUIRE(h[l], "emx_mlp_forward: activation buffer of layer %d is null", l);
  const int pitch = pitch_for(desc, layer_begin, layer_end);
  const size_t lds = (size_t)2 * R * pitch * sizeof(float);
  rc = set_lds(reinterpret_cast<const void*>(&mlp_forward_kernel), lds, "emx_mlp_forward");
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid), dim3(NTC), lds, (hipStream_t)stream, d, layer_begin, layer_end, in, ld_in, row_begin,
                     row_end, pitch);
  return emx::check_launch("mlp_forward_kernel");
}

int emx_mlp_backward(const emx_mlp_desc* desc, float* params, int layer_begin, int layer_end, const float* dtop, int ld_top,
                     const float* dadd, int layer_add, const emx_latent_grad* latent, float* dbelow, int row_begin, int row_end,
                     float* const* h, float* const* delta, emx_stream_t stream) {
  MlpDev d;
  int rc = fill_dev(desc, params, h, delta, d, "emx_mlp_backward");
  if (rc) return rc;
  EMX_REQUIRE(dtop && h && delta, "emx_mlp_backward: null pointer argument");
  EMX_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= desc->n_layers, "emx_mlp_backward: bad layer range");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && ld_top >= desc->width[layer_end], "emx_mlp_backward: bad rows / stride");
  EMX_REQUIRE((!dadd && !latent) || (layer_add >= layer_begin && layer_add < layer_end), "emx_mlp_backward: layer_add outside the layer range");
  emx_latent_grad lg{};
  if (latent) {
    lg = *latent;
    EMX_REQUIRE(lg.partials && lg.Z && lg.n_slots > 0, "emx_mlp_backward: incomplete latent gradient descriptor");
  }
  for (int l = layer_begin; l < layer_end; ++l) EMX_REQUIRE(h[l] && delta[l], "emx_mlp_backward: buffer of layer %d is null", l);
  const int pitch = pitch_for(desc, layer_begin, layer_end);
  const size_t lds = (size_t)2 * R * pitch * sizeof(float);
  rc = set_lds(reinterpret_cast<const void*>(&mlp_backward_kernel), lds, "emx_mlp_backward");
  if (rc) return rc;
  const int grid = (row_end - row_begin + R - 1) / R;
  hipLaunchKernelGGL(mlp_backward_kernel, dim3(grid), dim3(NTC), lds, (hipStream_t)stream, d, layer_begin, layer_end, dtop, ld_top, dadd,
                     layer_add, lg, dbelow, row_begin, row_end, pitch);
  return emx::check_launch("mlp_backward_kernel");
}

int emx_mlp_wgrad_slices(int n_rows) { return std::max((n_rows + EMX_WGRAD_SLICE_ROWS - 1) / EMX_WGRAD_SLICE_ROWS, 1); }

size_t emx_mlp_wgrad_ws_floats(int64_t n_params, int n_rows) {
  int slices = (n_rows + EMX_WGRAD_SLICE_ROWS - 1) / EMX_WGRAD_SLICE_ROWS;
  return (size_t)std::max(slices, 1) * (size_t)n_params;
}

int emx_mlp_wgrad(const emx_mlp_desc* desc, float* params, int64_t n_params, int layer_begin, int layer_end, const float* xin,
                  int ld_xin, int row_begin, int row_end, float* const* h, float* const* delta, float* grad, float* ws,
                  size_t ws_floats, emx_stream_t stream) {
  WgradDev a;
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
    tiles += ((desc->width[l] + 127) / 128) * ((desc->width[l + 1] + 127) / 128);
    const long wl = desc->w_off[l], wh = desc->b_off[l] + desc->width[l + 1];
    lo = (lo < 0) ? wl : std::min(lo, wl);
    hi = std::max(hi, wh);
  }
  a.tile_begin[layer_end - layer_begin] = tiles;
  a.partial = ws;
  a.n_params = n_params;
  a.row0 = row_begin;
  a.row_end = row_end;
  a.rows_per_slice = EMX_WGRAD_SLICE_ROWS;
  const int slices = (nrows + EMX_WGRAD_SLICE_ROWS - 1) / EMX_WGRAD_SLICE_ROWS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(tiles, slices), dim3(NT), 0, st, a);
  rc = emx::check_launch("mlp_wgrad_kernel");
  if (rc) return rc;
  if (!grad) return EMX_OK;   // the caller sums the slices itself (emx_adam_tick)
  const long n = hi - lo;
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, ws, (long)n_params, slices, lo, hi, grad);
  return emx::check_launch("mlp_wgrad_reduce_kernel");
}

}  // extern "C"
