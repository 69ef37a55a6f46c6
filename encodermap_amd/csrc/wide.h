// One Dense layer whose SOURCE rows are too wide for the row-block kernels' LDS images (included by mlp.hip, inside its namespace).
//
// The row-block chain kernels keep a layer's whole source -- 16 rows x K columns as two f16 planes -- in LDS: 64 K bytes, i.e.
// K <= ~2400 next to a second image.  The reference has no such limit (models/models.py:3133-3160 stacks Dense layers of any
// width; a periodic data set of 1300 features enters the network 2600 wide, an ADC model of a 450-residue protein ~2700 wide).
// A layer beyond it runs here, by itself: one workgroup per block of 16 rows STREAMS the source in windows of 256 columns --
// staged exactly as the chain kernels stage their row block (per (row, 32 columns) power-of-two scales, two f16 planes,
// transposed image; the same fragments for the weight gradients, aop.h) -- and accumulates up to eight passes of 128 output
// columns at a time in registers (wider outputs: the source is streamed again per group of eight passes).  The weights are the
// layer's units of the operand store (wop.h) in the direction asked for, so the optimizer and the weight-gradient kernels see
// nothing new.  Forward: out = act(src W + b).  Backward (the TOP layer of a backward range): the source is
// delta_top = dtop * act'(h_top), formed while it is staged (its rows and fragments kept if asked for), out = delta_top W^T,
// the raw d loss / d h of the layer below -- the caller's next launch applies that layer's act' and joins.
// Arithmetic: every float32 product as the three f16 matrix products hl + lh + hh of the split operands (22-bit operands, f32
// accumulation), each K-step unscaled as it is added -- the dynamic-scale path of the chain kernels, element for element.
#pragma once

struct WideArgs {
  const float* src;        // [*, K] float32 rows, row stride ld_src
  int ld_src, K;
  const float* src_h;      // backward: h_top [*, K] (row stride K) for the activation derivative; null: none
  int src_act;             // backward: activation code of the top layer (ACT_*)
  float* src_out;          // backward: delta_top rows out [*, K] (may be null)
  AopTensor src_aop;       // fragments of the staged source for the weight gradients (frag null: not kept)
  const char* units;       // the layer's first unit in this direction; unit (pass p, chunk c) = units + (p C + c) 64 KB
  const float* ws;         // the layer's reciprocal weight scale
  int P, C;                // passes of 128 output columns, chunks of 128 reduction indices
  int N;                   // output columns that exist (0: no product -- the source is staged for its rows / fragments only)
  const float* bias;       // forward: [N]; null: none
  int act;                 // forward: activation of the layer
  float* out;              // [*, N] float32 rows, row stride ld_out
  int ld_out;
  AopTensor out_aop;       // fragments of the output (frag null: not kept)
  int row0, row_end;
};

constexpr int WIDE_WIN = 256;                       // source columns per window (two chunks of 128 k)
constexpr int WIDE_PG = 8;                          // passes of 128 output columns held in registers at a time
constexpr int WIDE_PITCH = 128 * WIDE_PG + 4;       // floats per row of the output block in LDS
constexpr int WIDE_IMG = WIDE_WIN * 32;             // bytes of one plane of the window's image
constexpr int WIDE_LDS = 2 * WIDE_IMG + (WIDE_WIN / 32) * 64 + R * WIDE_PITCH * 4;

template <bool BWD>
__global__ __launch_bounds__(NTC) void wide_layer_kernel(const WideArgs a) {
  extern __shared__ __align__(16) char lds[];
  char* const img = lds;                                             // hi plane [256 columns][16 rows] f16, lo plane behind it
  float* const sct = reinterpret_cast<float*>(lds + 2 * WIDE_IMG);   // [8 groups][16 rows] reciprocal scales
  float* const outl = sct + (WIDE_WIN / 32) * 16;                    // [16][WIDE_PITCH] the pass group's output rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int sc_c = lane & 31, sc_hf = lane >> 5;
  const int r0 = a.row0 + blockIdx.x * R;
  const int rb = r0 >> 4;
  const int rows_valid = min(R, a.row_end - r0);
  const int nwin = (a.K + WIDE_WIN - 1) / WIDE_WIN;
  const int npg = a.N > 0 ? (a.P + WIDE_PG - 1) / WIDE_PG : 1;
  const float wrs = *a.ws;

  for (int pg = 0; pg < npg; ++pg) {
    const int p0 = pg * WIDE_PG;
    const int np = a.N > 0 ? min(WIDE_PG, a.P - p0) : 0;
    f32x4 acc[WIDE_PG];
#pragma unroll
    for (int p = 0; p < WIDE_PG; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int w = 0; w < nwin; ++w) {
      // ---- stage the window: wave = column group (32 columns), lane (c, hf) = column c, rows 8 hf .. 8 hf + 7
      {
        const int grp = (WIDE_WIN / 32) * w + wave;     // column group of the source
        const int col = 32 * grp + sc_c, cc = min(col, a.K - 1);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = 8 * sc_hf + e;
          const long grow = (long)min(r0 + row, a.row_end - 1);
          float y;
          if (BWD && pg > 0 && a.src_out != nullptr) {
            // a later pass group streams the source again: delta_top as this very thread stored it during the first one (src_out may
            // BE src -- the hand-over between two segments of a backward range happens in place -- and then holds delta, not dtop)
            y = a.src_out[grow * a.K + cc];
          } else {
            y = a.src[grow * a.ld_src + cc];
            if constexpr (BWD) {
              if (a.src_h != nullptr) {
                const float hv = a.src_h[grow * a.K + cc];
                y *= a.src_act == ACT_TANH ? fmaf(-hv, hv, 1.f) : act_deriv_from_output(a.src_act, hv);
              }
            }
          }
          v[e] = y * ((col < a.K && row < rows_valid) ? 1.f : 0.f);
          if constexpr (BWD) {
            if (pg == 0 && a.src_out != nullptr && col < a.K && row < rows_valid) a.src_out[(long)(r0 + row) * a.K + col] = v[e];
          }
        }
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
        for (int e = 0; e < 8; ++e) {
          const float t = v[e] * s[e];
          hi[e] = (_Float16)t;
          lo[e] = (_Float16)(t - (float)hi[e]);
        }
        const int k = 32 * wave + sc_c;                 // column inside the window
        *reinterpret_cast<half8*>(img + k * 32 + sc_hf * 16) = hi;
        *reinterpret_cast<half8*>(img + WIDE_IMG + k * 32 + sc_hf * 16) = lo;
        {
          float* dst = sct + wave * 16 + 8 * sc_hf;     // (every lane of a half writes the same 32 bytes)
          *reinterpret_cast<f32x4*>(dst) = f32x4{rs[0], rs[1], rs[2], rs[3]};
          *reinterpret_cast<f32x4*>(dst + 4) = f32x4{rs[4], rs[5], rs[6], rs[7]};
        }
        if (pg == 0 && a.src_aop.frag != nullptr && grp < a.src_aop.CG) {
          bm = max_xor32(bm);
          float sb, rsb;
          pow2_scale_bits<ACT_TOP>(bm, sb, rsb);
          half8 fh, fl;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const _Float16 ratio = (_Float16)(sb * rs[e]);
            fh[e] = hi[e] * ratio;
            fl[e] = lo[e] * ratio;
          }
          char* f = a.src_aop.frag + ((long)rb * a.src_aop.CG + grp) * 2048 + lane * 16;
          *reinterpret_cast<half8*>(f) = fh;
          *reinterpret_cast<half8*>(f + 1024) = fl;
          if (lane == 0) a.src_aop.rscale[(long)rb * a.src_aop.CG + grp] = rsb;
        }
      }
      __syncthreads();
      // ---- products: the window's (up to) two chunks against the units of the pass group
      if (np > 0) {
        for (int cc2 = 0; cc2 < 2; ++cc2) {
          const int c = 2 * w + cc2;
          if (c >= a.C) break;
          half8 ah[4], al[4];
          f32x4 rsa[4];
          const char* apl = img + cc2 * 4096 + lane * 8;
          const char* asc = reinterpret_cast<const char*>(sct) + cc2 * 256 + g * 16;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const half4 h0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + s * 1024));
            const half4 h1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + s * 1024 + 512));
            const half4 l0 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + WIDE_IMG + s * 1024));
            const half4 l1 = lds_read_tr16(reinterpret_cast<const _Float16*>(apl + WIDE_IMG + s * 1024 + 512));
            ah[s] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            al[s] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            rsa[s] = *reinterpret_cast<const f32x4*>(asc + s * 64);
          }
          // the wave's pieces of unit (p, c): [wave][K-step][plane] x 1 KB; the next pass's are requested under this pass's products
          half8 wp[2][4][2];
          auto pieces = [&](const int slot, const int p) __attribute__((always_inline)) {
            const char* u = a.units + ((long)(p0 + p) * a.C + c) * WOP_UNIT + wave * 8192 + lane * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              wp[slot][s][0] = *reinterpret_cast<const half8*>(u + s * 2048);
              wp[slot][s][1] = *reinterpret_cast<const half8*>(u + s * 2048 + 1024);
            }
          };
          pieces(0, 0);
#pragma unroll
          for (int p = 0; p < WIDE_PG; ++p) {
            if (p < np) {
              if (p + 1 < np) pieces((p + 1) & 1, p + 1);
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], wp[p & 1][s][1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s], wp[p & 1][s][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s], wp[p & 1][s][0], d, 0, 0, 0);
                acc[p] = d * rsa[s] + acc[p];
              }
            }
          }
        }
      }
      __syncthreads();   // (the next window overwrites the image)
    }
    if (np == 0) continue;

    // ---- the pass group's rows: lane (n, g) of wave w holds rows 4 g .. 4 g + 3 of column 128 p + 16 w + n
#pragma unroll
    for (int p = 0; p < WIDE_PG; ++p) {
      if (p < np) {
        const int col = 128 * (p0 + p) + 16 * wave + n;
        const float bv = (!BWD && a.bias != nullptr) ? a.bias[min(col, a.N - 1)] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float x = fmaf(acc[p][rr], wrs, bv);
          if constexpr (!BWD) {
            if (a.act == ACT_TANH) x = tanh_f32(x);
            else if (a.act != ACT_NONE) x = act_apply(a.act, x);
          }
          outl[(4 * g + rr) * WIDE_PITCH + 128 * p + 16 * wave + n] = x;
        }
      }
    }
    __syncthreads();
    const int c0 = 128 * p0, ncols = min(128 * np, a.N - c0);
    if (a.out != nullptr) {
      for (int i = tid; i < R * ncols; i += NTC) {
        const int r = i / ncols, cidx = i - r * ncols;
        if (r < rows_valid) a.out[(long)(r0 + r) * a.ld_out + c0 + cidx] = outl[r * WIDE_PITCH + cidx];
      }
    }
    if (a.out_aop.frag != nullptr) {
      for (int q = wave; 32 * q < ncols; q += NWC) {
        const int grp = 4 * p0 + q;
        const int cl = 32 * q + sc_c;
        float v[8], s[8], rs[8];
        unsigned bm = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = 8 * sc_hf + e;
          v[e] = (cl < ncols && row < rows_valid) ? outl[row * WIDE_PITCH + min(cl, ncols - 1)] : 0.f;
          unsigned m = row16_max(abs_bits(v[e]));
          m = max_xor16(m);
          pow2_scale_bits<ACT_TOP>(m, s[e], rs[e]);
          bm = max(bm, m);
        }
        bm = max_xor32(bm);
        float sb, rsb;
        pow2_scale_bits<ACT_TOP>(bm, sb, rsb);
        half8 fh, fl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = v[e] * s[e];
          const _Float16 hi = (_Float16)t, lo = (_Float16)(t - (float)hi);
          const _Float16 ratio = (_Float16)(sb * rs[e]);
          fh[e] = hi * ratio;
          fl[e] = lo * ratio;
        }
        if (grp < a.out_aop.CG) {
          char* f = a.out_aop.frag + ((long)rb * a.out_aop.CG + grp) * 2048 + lane * 16;
          *reinterpret_cast<half8*>(f) = fh;
          *reinterpret_cast<half8*>(f + 1024) = fl;
          if (lane == 0) a.out_aop.rscale[(long)rb * a.out_aop.CG + grp] = rsb;
        }
      }
    }
    __syncthreads();   // (the next pass group rewrites outl)
  }
}

template <bool BWD>
int launch_wide(const WideArgs& a, hipStream_t st, const char* who);
