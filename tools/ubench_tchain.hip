// The TRANSPOSED, register-resident chain of Dense layers (docs/LAB_NOTEBOOK.md section 8, plan for the dense layers), as a microbenchmark:
// H^T = W^T X^T on v_mfma_f32_16x16x32_f16 with the weights as the A operand and the 16 batch rows of a block as the N dimension.
// D comes out as lane (n = batch row, g) holding out-features 16 t + 4 g .. + 3 of tile t; with the k order
//     kk(g, e) = 4 g + e (e < 4), 16 + 4 g + (e - 4) (e >= 4)
// inside a K-step of 32 (csrc/wop.h's order) the accumulators of tiles 2 s and 2 s + 1 in lane (n, g) ARE that lane's eight k of
// K-step s of the next layer: bias, tanh and the f16 split happen in place -- no LDS image, no barrier, no cross-lane move.
// Every float32 product is the three f16 products hl + lh + hh of operands split into two planes (22 bits), as in the library.
//   form 1: ONE wave carries a 16-row block through all layers (96 matrix instructions per 128 x 128 layer);
//   form 2: TWO waves split the K-steps (wave w: K-steps 2 w, 2 w + 1), exchange 16 accumulator registers through LDS and each
//           finalises the four tiles whose features are its own K-steps of the next layer.
// 256 workgroups (one row block per CU: the regime of B = 4096), LAYERS layers of 128 x 128 with tanh; checked against float64.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_tchain.hip -o tools/ubench_tchain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 128;            // layer width
constexpr int LAYERS = 12;
constexpr float SX = 16384.f;     // fixed scale of the activations (tanh outputs and inputs in [-1, 1])

__host__ __device__ inline int kk(int g, int e) { return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4); }

__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// pieces: [layer][tile t (8)][K-step s (4)][plane (2)][lane (64)] x 16 bytes; rsw[layer] = 1 / weight scale; bias [layer][128]
template <int NWAVES, int wave>
__device__ __forceinline__ void tchain_body(const half8* __restrict__ pieces, const float* __restrict__ rsw, const float* __restrict__ bias,
                                            const float* __restrict__ X, float* __restrict__ Y, long long* __restrict__ cycles, f32x4* xch) {
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  constexpr int KS0 = NWAVES == 2 ? 2 : 4;   // K-steps per wave
  const int s_base = NWAVES == 2 ? 2 * wave : 0;
  // the block's rows as B operands of the wave's K-steps
  half8 bh[KS0], bl[KS0];
#pragma unroll
  for (int s = 0; s < KS0; ++s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = X[row * W + 32 * (s_base + s) + kk(g, e)] * SX;
      bh[s][e] = (_Float16)t;
      bl[s][e] = (_Float16)(t - (float)bh[s][e]);
    }
  }
  // the weight pieces of a tile travel one tile ahead of their use in a register ring of two (KS0 K-steps x 2 planes each)
  half8 wb[2][KS0][2];
  auto request = [&](const int slot, const int l, const int t) __attribute__((always_inline)) {
    const half8* P = pieces + ((long)min(l, LAYERS - 1) * 8 + t) * 4 * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      wb[slot][s][0] = P[((s_base + s) * 2 + 0) * 64];
      wb[slot][s][1] = P[((s_base + s) * 2 + 1) * 64];
    }
  };
  request(0, 0, 0);
  const long long t0 = clock64();
  f32x4 out[8];
  for (int l = 0; l < LAYERS; ++l) {
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < 7) request((t + 1) & 1, l, t + 1);
      else request(0, l + 1, 0);                       // (the next layer's first tile, under this layer's last products and epilogue)
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t & 1][s][0], bl[s], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t & 1][s][1], bh[s], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t & 1][s][0], bh[s], acc[t], 0, 0, 0);
      }
    }
    const float cm = rsw[l] * (1.f / SX);
    if constexpr (NWAVES == 2) {
      // hand the partner the partial sums of ITS tiles (wave 0 finalises tiles 0-3 = K-steps 0, 1 of the next layer, wave 1 tiles 4-7)
      f32x4* mine = xch + wave * 4 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; ++t) mine[t * 64] = acc[4 * (1 - wave) + t];
      __syncthreads();
      const f32x4* theirs = xch + (1 - wave) * 4 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[4 * wave + t] += theirs[t * 64];
      __syncthreads();
    }
    constexpr int T0 = NWAVES == 2 ? 4 : 8;
    const int t_base = NWAVES == 2 ? 4 * wave : 0;
#pragma unroll
    for (int t = 0; t < T0; ++t) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + l * W + 16 * (t_base + t) + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = tanh_f32(fmaf(acc[t_base + t][r], cm, b4[r]));
    }
    // in place: the next layer's B operands (tiles 2 s, 2 s + 1 of this wave -> its K-step s)
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = out[2 * s + (e >> 2)][e & 3] * SX;
        bh[s][e] = (_Float16)t;
        bl[s][e] = (_Float16)(t - (float)bh[s][e]);
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
  constexpr int T0 = NWAVES == 2 ? 4 : 8;
  const int t_base = NWAVES == 2 ? 4 * wave : 0;
#pragma unroll
  for (int t = 0; t < T0; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[row * W + 16 * (t_base + t) + 4 * g + r] = out[t][r];
}

// form 4: FOUR waves per row block, wave w = K-step w of every tile (24 matrix instructions per layer) and owner of tiles 2 w, 2 w + 1
// (= its K-step of the next layer).  A wave's 16 pieces of a layer are requested a whole layer ahead (register ring of two layers);
// the partial sums of the six tiles a wave does not own go to their owners through LDS (one barrier of four waves per layer).
template <int wave, bool NOSTREAM>
__device__ __forceinline__ void tchain4_body(const half8* __restrict__ pieces, const float* __restrict__ rsw, const float* __restrict__ bias,
                                             const float* __restrict__ X, float* __restrict__ Y, long long* __restrict__ cycles, f32x4* xch) {
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  half8 bh, bl;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = X[row * W + 32 * wave + kk(g, e)] * SX;
    bh[e] = (_Float16)t;
    bl[e] = (_Float16)(t - (float)bh[e]);
  }
  half8 wb[2][8][2];
  auto request = [&](const int slot, const int l) __attribute__((always_inline)) {
    const half8* P = pieces + (long)min(l, LAYERS - 1) * 8 * 4 * 2 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      wb[slot][t][0] = P[((t * 4 + wave) * 2 + 0) * 64];
      wb[slot][t][1] = P[((t * 4 + wave) * 2 + 1) * 64];
    }
  };
  request(0, 0);
  const long long t0 = clock64();
  f32x4 out[2];
  auto layer = [&](const int l, auto SLOT_) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT_)::value;
    if (!NOSTREAM) request(1 - slot, l + 1);
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[NOSTREAM ? 0 : slot][t][0], bl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[NOSTREAM ? 0 : slot][t][1], bh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[NOSTREAM ? 0 : slot][t][0], bh, acc[t], 0, 0, 0);
    }
    // xch[layer parity][owner][source][tile of the owner (2)][lane]: two copies, so ONE barrier per layer orders everything (a wave
    // can be at most one layer ahead of the slowest reader of the other copy)
    f32x4* const xc = xch + slot * (4 * 4 * 2 * 64);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o == wave) continue;
      xc[((o * 4 + wave) * 2 + 0) * 64 + lane] = acc[2 * o];
      xc[((o * 4 + wave) * 2 + 1) * 64 + lane] = acc[2 * o + 1];
    }
    __syncthreads();
    f32x4 s0 = acc[2 * wave], s1 = acc[2 * wave + 1];
#pragma unroll
    for (int src = 0; src < 4; ++src) {
      if (src == wave) continue;
      s0 += xc[((wave * 4 + src) * 2 + 0) * 64 + lane];
      s1 += xc[((wave * 4 + src) * 2 + 1) * 64 + lane];
    }
    const float cm = rsw[l] * (1.f / SX);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + l * W + 16 * (2 * wave) + 4 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + l * W + 16 * (2 * wave + 1) + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      out[0][r] = tanh_f32(fmaf(s0[r], cm, b0[r]));
      out[1][r] = tanh_f32(fmaf(s1[r], cm, b1[r]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = out[e >> 2][e & 3] * SX;
      bh[e] = (_Float16)t;
      bl[e] = (_Float16)(t - (float)bh[e]);
    }
  };
  for (int l = 0; l < LAYERS; l += 2) {
    layer(l, std::integral_constant<int, 0>{});
    layer(l + 1, std::integral_constant<int, 1>{});
  }
  const long long t1 = clock64();
  if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[row * W + 16 * (2 * wave + t) + 4 * g + r] = out[t][r];
}

template <bool NOSTREAM>   // NOSTREAM: every layer multiplies with the FIRST layer's pieces, requested once (wrong results: what the weight stream costs)
__global__ __launch_bounds__(256) void tchain4(const half8* __restrict__ pieces, const float* __restrict__ rsw, const float* __restrict__ bias,
                                               const float* __restrict__ X, float* __restrict__ Y, long long* __restrict__ cycles) {
  __shared__ __align__(16) f32x4 xch[2 * 4 * 4 * 2 * 64];   // 64 KB: two copies by layer parity
  const int w = threadIdx.x >> 6;
  if (w == 0) tchain4_body<0, NOSTREAM>(pieces, rsw, bias, X, Y, cycles, xch);
  else if (w == 1) tchain4_body<1, NOSTREAM>(pieces, rsw, bias, X, Y, cycles, xch);
  else if (w == 2) tchain4_body<2, NOSTREAM>(pieces, rsw, bias, X, Y, cycles, xch);
  else tchain4_body<3, NOSTREAM>(pieces, rsw, bias, X, Y, cycles, xch);
}

template <int NWAVES>
__global__ __launch_bounds__(64 * NWAVES) void tchain(const half8* __restrict__ pieces, const float* __restrict__ rsw, const float* __restrict__ bias,
                                                      const float* __restrict__ X, float* __restrict__ Y, long long* __restrict__ cycles) {
  __shared__ __align__(16) f32x4 xch[NWAVES == 2 ? 2 * 4 * 64 : 1];   // two waves: the partial accumulators handed to the partner
  if (NWAVES == 1 || threadIdx.x < 64) tchain_body<NWAVES, 0>(pieces, rsw, bias, X, Y, cycles, xch);   // (the wave index as a constant: register arrays stay registers)
  else tchain_body<NWAVES, NWAVES == 2 ? 1 : 0>(pieces, rsw, bias, X, Y, cycles, xch);
}

int main() {
  const int NB = 256, ROWS = NB * 16;
  std::vector<float> Wt((size_t)LAYERS * W * W), B((size_t)LAYERS * W), X((size_t)ROWS * W), rsw(LAYERS);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : Wt) v = rnd() * 0.15f;
  for (auto& v : B) v = rnd() * 0.1f;
  for (auto& v : X) v = rnd();
  std::vector<_Float16> pieces((size_t)LAYERS * 8 * 4 * 2 * 64 * 8);
  for (int l = 0; l < LAYERS; ++l) {
    float mx = 0.f;
    for (int i = 0; i < W * W; ++i) mx = std::fmax(mx, std::fabs(Wt[(size_t)l * W * W + i]));
    int ex;
    std::frexp(mx, &ex);
    const float sw = std::ldexp(1.f, 13 - ex);
    rsw[l] = 1.f / sw;
    for (int t = 0; t < 8; ++t)
      for (int s = 0; s < 4; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int m = lane & 15, g = lane >> 4;
            const float v = Wt[(size_t)l * W * W + (size_t)(32 * s + kk(g, e)) * W + 16 * t + m] * sw;   // W[k][out feature]
            const _Float16 h = (_Float16)v, lo = (_Float16)(v - (float)h);
            pieces[((((size_t)l * 8 + t) * 4 + s) * 2 + 0) * 512 + lane * 8 + e] = h;
            pieces[((((size_t)l * 8 + t) * 4 + s) * 2 + 1) * 512 + lane * 8 + e] = lo;
          }
  }
  // float64 reference
  std::vector<double> ref(X.begin(), X.end()), nxt((size_t)ROWS * W);
  for (int l = 0; l < LAYERS; ++l) {
    for (int r = 0; r < ROWS; ++r)
      for (int m = 0; m < W; ++m) {
        double a = B[(size_t)l * W + m];
        for (int k = 0; k < W; ++k) a += ref[(size_t)r * W + k] * Wt[(size_t)l * W * W + (size_t)k * W + m];
        nxt[(size_t)r * W + m] = std::tanh(a);
      }
    ref.swap(nxt);
  }
  half8* dP;
  float *dR, *dB, *dX, *dY;
  long long* dC;
  hipMalloc(&dP, pieces.size() * 2);
  hipMalloc(&dR, LAYERS * 4);
  hipMalloc(&dB, B.size() * 4);
  hipMalloc(&dX, X.size() * 4);
  hipMalloc(&dY, X.size() * 4);
  hipMalloc(&dC, NB * 8);
  hipMemcpy(dP, pieces.data(), pieces.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dR, rsw.data(), LAYERS * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int form = 1; form <= 4; ++form) {
    std::vector<float> Y(X.size());
    std::vector<long long> cyc(NB);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      if (form == 1) hipLaunchKernelGGL(tchain<1>, dim3(NB), dim3(64), 0, 0, dP, dR, dB, dX, dY, dC);
      else if (form == 2) hipLaunchKernelGGL(tchain<2>, dim3(NB), dim3(128), 0, 0, dP, dR, dB, dX, dY, dC);
      else if (form == 3) hipLaunchKernelGGL(tchain4<false>, dim3(NB), dim3(256), 0, 0, dP, dR, dB, dX, dY, dC);
      else hipLaunchKernelGGL(tchain4<true>, dim3(NB), dim3(256), 0, 0, dP, dR, dB, dX, dY, dC);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0) best = std::fmin(best, ms);
    }
    hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dC, NB * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < Y.size(); ++i) err = std::fmax(err, std::fabs(Y[i] - ref[i]));
    long long cs = 0;
    for (auto c : cyc) cs += c;
    printf("%s: kernel %.2f us, %.0f cycles per layer inside the wave (mean over %d row blocks), max |y - float64| %.2e\n",
           form == 1 ? "one wave per row block  " : (form == 2 ? "two waves per row block " : (form == 3 ? "four waves per row block" : "four waves, NO weight stream (wrong results)")), best * 1e3, (double)cs / NB / LAYERS, NB, err);
  }
  return 0;
}
