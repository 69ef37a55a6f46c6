// What the all-pairs tile kernels share with the fused dense-layer launch that runs the tiles beside the decoder (mlp.hip,
// chain_tiles_kernel): the argument record, the two sigmoid forms and the tile orders.  See sketchmap.hip for the algorithm.
#pragma once
#include "emx_common.h"
#include <type_traits>

#include "split.h"

namespace emx {
namespace sk {

struct SigDev {
  float k6;         // fast high-d form (a = 12, b = 6): inv_sig2 * c^(1/6), so that c v^6 = (k6 d^2)^6
  float c_inv;      // fast low-d form (a = 2): c / sig^2
  float inv_sig2;   // 1/sig^2
  float c;          // 2^(a/b) - 1
  float half_a;     // a/2
  float nba;        // -b/a
  float half_am2;   // (a-2)/2
  float vmax;       // clamp on (d/sig)^2 so that v^(a/2) stays finite
  int a_is_2;
};

struct SkArgs {
  const float* X;
  const float* Z;
  float* P;          // [nslots][nrows_pad][L]
  float* tile_loss;  // [ntiles]
  int B, D, ldx, L;
  int row0, row_end;  // rows handled: [row0,row_end)
  int nbr, nbc;       // tile rows / tile columns
  int nrows_pad;
  int symm;
  int vec4;
  float period;
  float rp_hi, rp_lo, phase_unit2;   // fixed-point periodic tiles: 1 / period in two floats, (period / 2^32)^2
  int phase_ok;                      // the data span less than 1.5 periods: the fixed-point tiles may run
  SigDev sh, sl;
  SplitRows split;   // optional f16 planes + norms of the rows (split.h): selects the f16 Gram kernel
  int xcd_nbg;       // > 0: XCD-aware tile order of the f16 kernel (xcd_tile), tile columns per group
  // few tiles, many features (B <= 640): the reduction of a tile is split over ksplit workgroups of the launch (kmode 3: each leaves
  // the Gram partial of its chunks in kpart and takes a ticket; the last one adds them up in a fixed order and runs the epilogue).
  // kmode 0: one workgroup per tile does both.
  int ksplit, kmode;
  float* kpart;      // [tiles][ksplit][4 waves][16][64]
  unsigned* ktickets; // [tiles] kmode 3: the split reduction in ONE launch -- workgroups of a tile that have left their share (zero between launches)
  // data parallel: the upper-triangle tiles are DEALT to the ranks -- tile t (row-major triangle index)
  // belongs to rank t % deal_world; a workgroup of another rank's tile leaves at once (its cost slot zeroed).  0 / 1: off.
  int deal_rank, deal_world;
};
__device__ __forceinline__ bool dealt_away(const SkArgs& a, int tslot) {
  if (a.deal_world <= 1 || tslot % a.deal_world == a.deal_rank) return false;
  if (threadIdx.x == 0) a.tile_loss[tslot] = 0.f;
  return true;
}

template <int N>
__device__ __forceinline__ float powi(float v) {
  if constexpr (N == 0) return 1.f;
  else if constexpr (N == 1) return v;
  else if constexpr (N % 2 == 0) { float h = powi<N / 2>(v); return h * h; }
  else return v * powi<N - 1>(v);
}

// complement of the sketch-map sigmoid, E(v) = (1 + c v^(a/2))^(-b/a), v = (d/sig)^2   (distances.py:86)
template <bool FAST>
__device__ __forceinline__ float sig_E_high(float d2, const SigDev& s) {
  if constexpr (FAST) {  // a = 12, b = 6: E = (1 + w^6)^(-1/2), w = d^2 c^(1/6) / sig^2 -- four instructions in front of the rsq; an
                         // overflowing w^6 gives q = inf, rsq = 0: the correct limit; a d^2 that rounding made slightly negative
                         // enters through an even power, so the caller needs no clamp at 0
    const float w = d2 * s.k6;
    const float w3 = w * w * w;
    return __builtin_amdgcn_rsqf(fmaf(w3, w3, 1.f));
  } else {
    float v = fminf(d2 * s.inv_sig2, s.vmax);
    float t = __builtin_amdgcn_exp2f(s.half_a * __builtin_amdgcn_logf(v));
    float q = fmaf(s.c, t, 1.f);
    return __builtin_amdgcn_exp2f(s.nba * __builtin_amdgcn_logf(q));
  }
}

// low-d: E and G = S'(d)/d / (b c / sig^2) = v^((a-2)/2) * q^(-b/a-1)
template <bool FAST>
__device__ __forceinline__ void sig_E_low(float d2, const SigDev& s, float& E, float& G) {
  if constexpr (FAST) {  // a = 2, b = 6
    float q = fmaf(s.c_inv, d2, 1.f);
    float r = __builtin_amdgcn_rcpf(q);
    E = r * r * r;
    G = E * r;
  } else {
    float v = fminf(d2 * s.inv_sig2, s.vmax);
    float lv = __builtin_amdgcn_logf(v);
    float t = __builtin_amdgcn_exp2f(s.half_a * lv);
    float q = fmaf(s.c, t, 1.f);
    E = __builtin_amdgcn_exp2f(s.nba * __builtin_amdgcn_logf(q));
    float um = s.a_is_2 ? 1.f : __builtin_amdgcn_exp2f(s.half_am2 * lv);
    G = um * E * __builtin_amdgcn_rcpf(q);
    G = (v > 0.f) ? G : 0.f;  // reference mask: coincident latent points carry no gradient
  }
}

// upper-triangle linear index -> (ti, tj), ti <= tj < nb, row-major
__device__ __forceinline__ void tri_decode(int t, int nb, int& ti, int& tj) {
  float fn = 2.f * nb + 1.f;
  int i = (int)((fn - sqrtf(fn * fn - 8.f * (float)t)) * 0.5f);
  i = max(0, min(i, nb - 1));
  while (i > 0 && (i * nb - i * (i - 1) / 2) > t) --i;
  while ((i + 1) * nb - (i + 1) * i / 2 <= t) ++i;
  ti = i;
  tj = i + (t - (i * nb - i * (i - 1) / 2));
}

// XCD-aware order of the upper-triangle tiles.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one,
// observed, speed only -- nothing below depends on it for correctness), each XCD has its own L2, and a tile needs the
// planes of its row panel and of its column panel.  In row-major triangle order every XCD ends up pulling ALL panels
// through the fabric (8 x 2 MB at C2, while all 2048 resident tiles stage at once).  Here the panels form four groups
// and the triangle ten group pairs: the six off-diagonal pairs (nbg^2 tiles each) go to the blocks with b % 8 = 0..5, the
// four diagonal ones (nbg (nbg + 1) / 2 tiles each) two by two to b % 8 = 6, 7 -- every XCD touches half of the panels,
// 256 / 272 tiles at C2.  Blocks beyond an XCD's share (and tiles beyond a ragged last group) leave at once.
// Returns false for such a block; t = the tile's index in row-major triangle order (its tile_loss slot).
__device__ __forceinline__ bool xcd_tile(int bid, int nbg, int nbc, int& ti, int& tj, int& t) {
  const int x = bid & 7, k = bid >> 3;
  int gi, gj, li, lj;
  if (x < 6) {
    gi = x < 3 ? 0 : (x < 5 ? 1 : 2);
    gj = x < 3 ? x + 1 : (x < 5 ? x - 1 : 3);
    if (k >= nbg * nbg) return false;
    li = k / nbg;
    lj = k - li * nbg;
  } else {
    const int T2 = nbg * (nbg + 1) / 2;
    if (k >= 2 * T2) return false;
    const int second = k >= T2 ? 1 : 0;
    gi = gj = 2 * (x - 6) + second;
    tri_decode(k - second * T2, nbg, li, lj);
  }
  ti = gi * nbg + li;
  tj = gj * nbg + lj;
  if (ti >= nbc || tj >= nbc) return false;
  t = ti * nbc - ti * (ti - 1) / 2 + (tj - ti);
  return true;
}


// ---- the VALU tiles' staging and pair-element arithmetic (sk_tile_kernel in sketchmap.hip, tile32_pair_role below)
constexpr int KC = 32;       // feature chunk staged per iteration
constexpr int LS = KC + 4;   // padded LDS row stride (floats)

__device__ __forceinline__ float abs_minus(float x, float h) {   // |x| - h
  float r;
  asm("v_sub_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "v"(h));
  return r;
}
__device__ __forceinline__ float minus_abs(float h, float x) {   // h - |x|
  float r;
  asm("v_sub_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(h), "v"(x));
  return r;
}

// PMODE of the VALU tiles: 0 Euclidean, 1 periodic (float form, any data), 2 periodic on fixed-point phases (data that span
// less than 1.5 periods, phase_u32 below)
template <int PMODE>
__device__ __forceinline__ float acc4(const float4& a, const float4& b, float acc, float period) {
  if constexpr (PMODE == 2) {
    // the staged values are phases: x / period as a 32-bit fraction of a turn.  The 32-bit difference wraps by itself and,
    // read as a signed integer, IS the wrapped difference in [-1/2, 1/2) turns: sub, convert, fma -- three instructions
    // per pair-element instead of four (the sum is scaled by (period / 2^32)^2 once per pair, in front of the epilogue)
    const uint4 ua = *reinterpret_cast<const uint4*>(&a), ub = *reinterpret_cast<const uint4*>(&b);
    const float d0 = (float)(int)(ua.x - ub.x), d1 = (float)(int)(ua.y - ub.y), d2 = (float)(int)(ua.z - ub.z),
                d3 = (float)(int)(ua.w - ub.w);
    acc = fmaf(d0, d0, acc);
    acc = fmaf(d1, d1, acc);
    acc = fmaf(d2, d2, acc);
    acc = fmaf(d3, d3, acc);
  } else if constexpr (PMODE == 1) {
    // min(|d|, p - |d|) = p/2 - | |d| - p/2 |: three subtractions, two of them with a |.| source modifier, instead of two
    // and a v_min_f32, which issues at half the rate of v_sub_f32 on gfx950 (profiles/r01_valu_issue_rates.txt).  Written
    // as instructions: left to itself the compiler pairs the subtractions into v_pk_add_f32, which has no source
    // modifiers, and spends a v_and_b32 per |.|.
    const float h = 0.5f * period;
    float d0 = abs_minus(a.x - b.x, h), d1 = abs_minus(a.y - b.y, h), d2 = abs_minus(a.z - b.z, h), d3 = abs_minus(a.w - b.w, h);
    d0 = minus_abs(h, d0);
    d1 = minus_abs(h, d1);
    d2 = minus_abs(h, d2);
    d3 = minus_abs(h, d3);
    acc = fmaf(d0, d0, acc);
    acc = fmaf(d1, d1, acc);
    acc = fmaf(d2, d2, acc);
    acc = fmaf(d3, d3, acc);
  } else {
    float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    acc = fmaf(d0, d0, acc);
    acc = fmaf(d1, d1, acc);
    acc = fmaf(d2, d2, acc);
    acc = fmaf(d3, d3, acc);
  }
  return acc;
}

// x / period as a 32-bit fraction of a turn (wrapping), to 2^-32 of a period: t = x * (1 / period) in two floats, its distance
// from the nearest integer is exact in float32, the rounding error of the product and the low part of 1 / period are added
// as a second integer.  (Exactly half a turn saturates to 2^31 - 1: one unit.)
__device__ __forceinline__ unsigned phase_u32(float x, float rp_hi, float rp_lo) {
  const float t = x * rp_hi;
  const float fh = t - rintf(t);
  const float fl = fmaf(x, rp_lo, fmaf(x, rp_hi, -t));
  return (unsigned)((int)(fh * 4294967296.f) + (int)rintf(fl * 4294967296.f));
}
__device__ __forceinline__ float4 phase4(const float4& v, float rp_hi, float rp_lo) {
  uint4 u = make_uint4(phase_u32(v.x, rp_hi, rp_lo), phase_u32(v.y, rp_hi, rp_lo), phase_u32(v.z, rp_hi, rp_lo), phase_u32(v.w, rp_hi, rp_lo));
  return *reinterpret_cast<float4*>(&u);
}

__device__ __forceinline__ float4 load_row4(const float* __restrict__ X, long row, int ldx, int k, int D, bool rowvalid,
                                            bool vec4) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rowvalid && k < D) {
    const float* p = X + row * (long)ldx + k;
    if (vec4) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      v.x = p[0];
      if (k + 1 < D) v.y = p[1];
      if (k + 2 < D) v.z = p[2];
      if (k + 3 < D) v.w = p[3];
    }
  }
  return v;
}


// ---------------------------------------------------------------------------------------------------------------------
// The all-pairs tiles as a ROLE inside the fused dense-layer launch (mlp.hip: chain_tiles_kernel).
//
// The tile pass needs the batch rows (split planes, left by the optimizer launch of the step in front) and the latent rows --
// which the first three of the fused launch's fifteen units produce.  As a launch of its own it ran behind the WHOLE fused launch
// (decoder forward, auto cost, decoder backward: ~15 of its ~24 us at B = 4096) although it depends on none of that.  Here the
// tiles are extra workgroups of the fused launch: workgroups [0, n_chain) push their 16-row blocks through the layers and PUBLISH
// their latent rows when the encoder's last layer is done (write-through stores, then one agent-scope add to a counter);
// workgroups [n_chain, ...) each take TWO tiles (a 512-thread workgroup = two 4-wave halves in lockstep: the barriers are
// workgroup-wide), run the Gram products -- which need no latent -- at once, wait for the counter, and run the sigmoid epilogue
// under the decoder half of the chain workgroups.  Per tile the arithmetic, the order of every addition and the slots the
// partial sums go to are those of sk_tile_f16_kernel<2, FASTSIG, NCH>: the results are bit-identical to the separate launches.
//
// Progress: the chain workgroups never wait for anything, and the dispatcher hands out workgroups in index order (per XCD), so
// every chain workgroup is resident before a tile workgroup of its XCD can occupy a slot; the spin below is nevertheless
// BOUNDED -- on expiry the workgroup raises sync[2] and carries on (wrong numbers, reported by the host, never a hang).
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility): every published byte is stored with sc1 by ONE wave, which
// drains its stores (s_waitcnt vmcnt(0)) before its lane 0 adds to the counter at agent scope; the consumer polls the counter
// with sc1 loads from one lane, a workgroup barrier follows, and EVERY load of the published rows is an sc1 load to registers.
// ---------------------------------------------------------------------------------------------------------------------
struct TileRole {
  SkArgs a;             // as for the stand-alone launch: planes, norms, partial-sum slots, sigmoid constants, XCD tile order
  const float* zpub;    // [B][2] latent rows as published by the chain workgroups of THIS launch
  unsigned* sync;       // [0] chain workgroups that have published, [1] tile workgroups that are done, [2] spin expired (sticky)
  int n_chain;          // chain workgroups of the launch
  int n_tile_wgs;       // tile workgroups of the launch
  int join;             // the chain workgroups of the launch wait for the tiles' partial sums (sync[3]: all tile workgroups are done;
                        // sync[4]: chain workgroups that have collected their rows) and run the encoder's backward units themselves
};
constexpr int TILE_ROLE_SMEM_FLOATS = 4096 + 4 * 64 + 64 * 2 + 64 * 2 * 2 + 8;   // one tile (sk_tile_f16_kernel<2, ...>: 18976 bytes)
constexpr int TILE_ROLE_LDS_BYTES = 2 * TILE_ROLE_SMEM_FLOATS * 4;
constexpr unsigned TILE_ROLE_SPIN_LIMIT = 1u << 16;                                // x ~0.5 us of sleep: tens of milliseconds

#ifndef EMX_TILE_EXP
#define EMX_TILE_EXP 0   // timing experiments only (wrong results): 1 = no pair epilogue, 2 = no Gram products
#endif
#ifndef EMX_TILE_PFX
#define EMX_TILE_PFX 3   // chunks of the next tile in flight under the epilogue of the current one
#endif
typedef float rf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 rf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float load_sc1(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// w: index of this tile workgroup (512 threads) among the launch's n_tile_wgs.  NCH: 32-feature chunks of a row (compile time, as in
// sk_tile_f16_kernel).  The workgroup is PERSISTENT: it walks its share of the tile pairs -- pair k of its XCD's list, k = w / 8,
// w / 8 + n_tile_wgs / 8, ... -- and keeps the next pair's operand planes IN FLIGHT under the current pair's sigmoid epilogue (all
// chunks of a tile are requested at once, into registers: one trip to the L2s per tile, hidden behind the epilogue in front of it;
// as a launch of its own the pass hides these trips behind seven other resident tiles, which a CU shared with a chain workgroup
// does not have room for).
template <bool FASTSIG, int NCH>
__device__ __forceinline__ void tile_pair_role(char* lds, const TileRole& tr, const int w) {
  constexpr int T = 64, LMAX = 2, L = 2, RP = 20, HNT = 256;
  constexpr int MAIN = 4096;
  // The operand pieces of a tile travel through a register ring of RING chunk slots (chunk c in slot c % RING): the first PFX chunks
  // of the NEXT tile are requested when the current tile's products are done and are in flight under its epilogue (more slots
  // alive there would not fit the 128 registers next to the epilogue's own); the others are requested at the start of the tile's
  // products -- one trip, partly behind the first chunks' products -- and, for rows of more than 128 features, slot by slot as
  // the slots come free.
  constexpr int RING = NCH < 4 ? NCH : 4;
  constexpr int PFX = NCH < EMX_TILE_PFX ? NCH : EMX_TILE_PFX;
  const SkArgs& a = tr.a;
  // (half and wave are uniform over a wave: said so, the tile coordinates and everything derived from them live in scalar registers)
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  float* smem = reinterpret_cast<float*>(lds) + half * TILE_ROLE_SMEM_FLOATS;
  float2* nr_s = reinterpret_cast<float2*>(smem + MAIN);   // [2 T] (|x|^2, 1/s): rows of the i side, then of the j side
  float* zi_s = smem + MAIN + 4 * T;                       // [T][LMAX]
  float* cpart = zi_s + T * LMAX;                          // [T][LMAX][2]
  float* red = cpart + T * LMAX * 2;
  const int wm = wave >> 1, wn = wave & 1;     // quadrant of this wave
  const int lr = lane & 31, h = lane >> 5;
  const SplitRows& sp = a.split;
  const long chunk_stride = (long)sp.B * SPLIT_KC;
  ru32x4* buf = reinterpret_cast<ru32x4*>(smem);
  const int ra = wm * 32 + lr, rbj = wn * 32 + lr;

  // this workgroup's pairs: k = k0, k0 + kstep, ... < n_pairs
  const bool xcd = a.xcd_nbg > 0;
  const int xq = w & 7;
  const int n_tiles = a.nbc * (a.nbc + 1) / 2;
  const int n_pairs = xcd ? a.xcd_nbg * (a.xcd_nbg + 1) / 2 : (n_tiles + 1) / 2;     // (XCD order: pairs per XCD list)
  const int k0 = xcd ? (w >> 3) : w, kstep = xcd ? (tr.n_tile_wgs >> 3) : tr.n_tile_wgs;

  // tile of pair k for this half: packed as ti | tj << 10 | valid << 20 (scalars; tslot follows from ti, tj)
  auto tile_of = [&](int k) __attribute__((always_inline)) -> int {
    int ti = 0, tj = 0, tslot = 0;
    bool valid;
    if (xcd) {
      valid = xcd_tile((2 * k + half) * 8 + xq, a.xcd_nbg, a.nbc, ti, tj, tslot);
    } else {
      tslot = 2 * k + half;
      valid = tslot < n_tiles;
      if (valid) tri_decode(tslot, a.nbc, ti, tj);
    }
    if (!valid) ti = tj = 0;   // (keeps the half in step with the other one's barriers; nothing of it is stored)
    return __builtin_amdgcn_readfirstlane(ti | (tj << 10) | ((valid ? 1 : 0) << 20));
  };

  // piece e = tid + 256 u of a chunk: side u >> 1 (i / j), plane u & 1, row tid >> 2, quarter tid & 3 -- the same for every tile
  const int prow = tid >> 2, pq = tid & 3;
  const int lo0 = prow * 4 + (pq ^ ((prow >> 2) & 3));     // + 256 u: ((side * 2 + plane) * 64 + row) * 4 + (q ^ ((row >> 2) & 3))
  ru32x4 pre[RING][4];
  float2 nr_pre = make_float2(0.f, 0.f);
  auto request = [&](const int t, auto C_) __attribute__((always_inline)) {   // chunk C of tile t into its ring slot
    constexpr int c = decltype(C_)::value;
    const long goff = (long)c * chunk_stride;
    const uint16_t* gi = sp.planes + (long)min((t & 1023) * T + prow, a.B - 1) * SPLIT_KC + 8 * pq + goff;
    const uint16_t* gj = sp.planes + (long)min(((t >> 10) & 1023) * T + prow, a.B - 1) * SPLIT_KC + 8 * pq + goff;
    pre[c % RING][0] = *reinterpret_cast<const ru32x4*>(gi);
    pre[c % RING][1] = *reinterpret_cast<const ru32x4*>(gi + sp.plane_stride);
    pre[c % RING][2] = *reinterpret_cast<const ru32x4*>(gj);
    pre[c % RING][3] = *reinterpret_cast<const ru32x4*>(gj + sp.plane_stride);
  };
  auto request_head = [&](const int t) __attribute__((always_inline)) {   // the chunks that travel under the epilogue in front
    static_assert(PFX <= 4, "request_head");
    request(t, std::integral_constant<int, 0>{});
    if constexpr (PFX > 1) request(t, std::integral_constant<int, 1>{});
    if constexpr (PFX > 2) request(t, std::integral_constant<int, (PFX > 2 ? 2 : 0)>{});
    if constexpr (PFX > 3) request(t, std::integral_constant<int, (PFX > 3 ? 3 : 0)>{});
  };
  auto request_norms = [&](const int t) __attribute__((always_inline)) {
    if (tid < 2 * T) {
      const int grow = min((tid < T ? (t & 1023) * T : ((t >> 10) & 1023) * T - T) + tid, a.B - 1);
      nr_pre = make_float2(sp.norms[grow], sp.rscale[grow]);
    }
  };
  // the latent rows a tile's epilogue needs, per thread: element tid of the i side's [T][2] block (threads 0 .. 127) and the two
  // columns of this lane's j row -- sc1 loads of what the chain workgroups published
  auto request_z = [&](const int t, float& zi, float (&zj)[LMAX]) __attribute__((always_inline)) {
    const int ig = min((t & 1023) * T + (tid >> 1), a.B - 1);
    zi = tid < T * LMAX ? load_sc1(tr.zpub + (long)ig * L + (tid & 1)) : 0.f;
    const int jg = ((t >> 10) & 1023) * T + wn * 32 + lr;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) zj[l] = jg < a.B ? load_sc1(tr.zpub + (long)jg * L + l) : 0.f;
  };
  auto mfma_chunk = [&](rf32x16& acc) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qa = (2 * t + h) ^ ((ra >> 2) & 3), qb = (2 * t + h) ^ ((rbj >> 2) & 3);
      const rf16x8 ah = __builtin_bit_cast(rf16x8, buf[ra * 4 + qa]), al = __builtin_bit_cast(rf16x8, buf[(64 + ra) * 4 + qa]);
      const rf16x8 bh = __builtin_bit_cast(rf16x8, buf[(128 + rbj) * 4 + qb]), bl = __builtin_bit_cast(rf16x8, buf[(192 + rbj) * 4 + qb]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);   // small products first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  };
  // chunk C of the tile: (C == 0: the ring's remaining slots are requested first) write, barrier, products, then the slot is
  // requested again for chunk C + RING
  auto gram = [&](const int t, rf32x16& acc, auto C_, auto&& self) __attribute__((always_inline)) -> void {
    constexpr int c = decltype(C_)::value;
    if constexpr (c < NCH) {
      if constexpr (c == 0) {
        if constexpr (PFX < RING) request(t, std::integral_constant<int, (PFX < RING ? PFX : 0)>{});
        if constexpr (PFX + 1 < RING) request(t, std::integral_constant<int, (PFX + 1 < RING ? PFX + 1 : 0)>{});
      }
      __syncthreads();   // (c = 0: the epilogue in front has finished with the area)
#pragma unroll
      for (int u = 0; u < 4; ++u) buf[lo0 + 256 * u] = pre[c % RING][u];
      if (c == 0 && tid < 2 * T) nr_s[tid] = nr_pre;
      __syncthreads();
      if constexpr (c + RING < NCH) request(t, std::integral_constant<int, (c + RING < NCH ? c + RING : 0)>{});
      mfma_chunk(acc);
      self(t, acc, std::integral_constant<int, c + 1>{}, self);
    }
  };
  bool latent_ready = false;
  if (k0 >= n_pairs) goto report;
  {
    int cur = tile_of(k0);
    request_head(cur);
    request_norms(cur);
    float zi_c = 0.f, zj_c[LMAX] = {0.f, 0.f};
    bool z_c = false;
    for (int k = k0; k < n_pairs; k += kstep) {
      const int cti = cur & 1023, ctj = (cur >> 10) & 1023;
      const bool cvalid = (cur >> 20) != 0;
      const int i0 = cti * T, j0 = ctj * T;
      const bool offdiag = cti != ctj;
      rf32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // ---- Gram products: the chunks through the LDS image (as sk_tile_f16_kernel: write, barrier, six instructions)
#if EMX_TILE_EXP != 2
      gram(cur, acc, std::integral_constant<int, 0>{}, gram);
#else
      __syncthreads();
      if (tid < 2 * T) nr_s[tid] = nr_pre;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = pre[0][r & 3][r >> 2] * 1e-9f;
      __syncthreads();
#endif
      // ---- the next pair's planes go on their way now: they travel under this pair's epilogue
      const int kn = k + kstep;
      int nxt = cur;
      float zi_n = 0.f, zj_n[LMAX] = {0.f, 0.f};
      bool z_n = false;
      if (kn < n_pairs) {
        nxt = tile_of(kn);
        request_head(nxt);
        request_norms(nxt);
        if (latent_ready) {   // (the latent rows are all there: the next tile's go on their way with its planes)
          request_z(nxt, zi_n, zj_n);
          z_n = true;
        }
      }
      // ---- the latent rows: published by the chain workgroups of this launch (first pair only: then they are all there)
      if (!latent_ready) {
        if (threadIdx.x == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(&tr.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tr.n_chain) {
            __builtin_amdgcn_s_sleep(20);
            if (++spins > TILE_ROLE_SPIN_LIMIT) {
              __hip_atomic_store(&tr.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
        }
        latent_ready = true;
      }
      __syncthreads();   // the counter has been seen; all MFMA operand reads are done: the staging area becomes the row-partial area
      // ---- epilogue: this lane owns column j = wn*32 + lr and the 16 rows (reg&3) + 8(reg>>2) + 4h of the quadrant
      const int jl = wn * 32 + lr;
      const int jg = j0 + jl;
      const bool jok = jg < a.B;
      float zj[LMAX];
      if (!z_c) request_z(cur, zi_c, zj_c);   // (first pair: the rows could not be asked for before they were published)
      if (tid < T * LMAX) zi_s[tid] = zi_c;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) zj[l] = zj_c[l];
      __syncthreads();   // latent rows of the i side visible
      const float2 nrj = nr_s[T + jl];
      const float nj = nrj.x, cj = -2.f * nrj.y;

      float* rpart = smem;   // [T][LMAX][RP]
      float gj[LMAX];
#pragma unroll
      for (int l = 0; l < LMAX; ++l) gj[l] = 0.f;
      float lsum = 0.f;
      const bool edge = (i0 + T > a.row_end) || (j0 + T > a.B) || (i0 == j0);
      const int ib = wm * 32 + 4 * h;
      const float2* nr_b = nr_s + ib;
      const float* zi_b = zi_s + ib * LMAX;
      float* rp_b = rpart + ib * LMAX * RP + (jl >> 2);
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
        if (!FASTSIG) d2 = fmaxf(d2, 0.f);
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
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
          rp_b[(ro * LMAX + l) * RP] = v;
        }
      };
#if EMX_TILE_EXP != 1
      if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pair(r, true);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) pair(r, false);
      }
#else
      lsum = acc[0] + acc[5] + zj[0] + nj + cj + zi_b[0] + nr_b[0].x;
      (void)pair;
      (void)rp_b;
#endif
      if (offdiag) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
          float s = sum_xor32(gj[l]);   // the two row halves of the quadrant
          if (h == 0) cpart[(jl * LMAX + l) * 2 + wm] = s;
        }
      }
      // tile cost: block_sum<256>'s order (wave sums, then waves 0..3) on this half's four waves
      {
        const float v = wave_sum(lsum);
        if (lane == 0) red[wave] = v;
      }
      __syncthreads();   // partials and wave sums visible
      if (tid == 0 && cvalid) {
        float tl = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) tl += red[i];
        a.tile_loss[cti * a.nbc - cti * (cti - 1) / 2 + (ctj - cti)] = offdiag ? 2.f * tl : tl;   // (row-major triangle index)
      }
      if (cvalid) {
        for (int e = tid; e < T * LMAX; e += HNT) {
          const int rl = e / LMAX, l = e - rl * LMAX;
          const float4* p = reinterpret_cast<const float4*>(&rpart[e * RP]);
          const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
          const float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) +
                          (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
          store_sc1(&a.P[((long)ctj * a.nrows_pad + i0 + rl) * L + l], s);
        }
        if (offdiag) {
          for (int e = tid; e < T * LMAX; e += HNT) {
            const int rl = e / LMAX, l = e - rl * LMAX;
            store_sc1(&a.P[((long)cti * a.nrows_pad + j0 + rl) * L + l], cpart[e * 2] + cpart[e * 2 + 1]);
          }
        }
      }
      cur = nxt;
      zi_c = zi_n;
      zj_c[0] = zj_n[0];
      zj_c[1] = zj_n[1];
      z_c = z_n;
    }
  }
report:
  // the last tile workgroup to finish resets the launch's counters (every chain workgroup has published by then: a tile
  // workgroup with work saw the full count) -- and, when the chain workgroups wait for the tiles' partial sums (join: stored with
  // sc1 above, drained here by every wave before the workgroup counts itself done), tells them
  if (tr.join) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(&tr.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)tr.n_tile_wgs - 1) {
      __hip_atomic_store(&tr.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&tr.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tr.join) __hip_atomic_store(&tr.sync[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The same role for PERIODIC batches of up to 512 rows (the default EncoderMap configuration: B = 256, periodic): the 32 x 32 VALU
// tiles of sk_tile_kernel<32, 32, 2, PMODE, FASTSIG> -- direct (wrapped) differences staged from the batch rows themselves, PMODE
// 1 the float form, 2 the 32-bit phases -- two tiles per 512-thread workgroup in lockstep, the pair loop (which needs no latent)
// at once, the epilogue behind the published latent rows.  Same arithmetic, order and slots as the stand-alone kernel.
constexpr int TILE32_SMEM_FLOATS = (32 + 32) * LS + 8;   // one tile: staging (2304 floats) >= its partial sums (2048)

template <int PMODE, bool FASTSIG>
__device__ __forceinline__ void tile32_pair_role(char* lds, const TileRole& tr, const int w) {
  constexpr int T = 32, RM = 2, RN = 2, LMAX = 2, L = 2, HNT = 256;
  const SkArgs& a = tr.a;
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  float* smem = reinterpret_cast<float*>(lds) + half * TILE32_SMEM_FLOATS;
  float* Xi_s = smem;
  float* Xj_s = smem + T * LS;
  float* red = smem + TILE32_SMEM_FLOATS - 8;
  const int tx = tid & 15, ty = tid >> 4;
  const int kq = tid & 7, rb = tid >> 3;
  const bool vec4 = a.vec4 != 0;
  const int n_tiles = a.nbc * (a.nbc + 1) / 2, n_pairs = (n_tiles + 1) / 2;
  bool latent_ready = false;
  for (int k = w; k < n_pairs; k += tr.n_tile_wgs) {
    int ti = 0, tj = 0;
    const int tslot = 2 * k + half;
    const bool valid = tslot < n_tiles;
    if (valid) tri_decode(tslot, a.nbc, ti, tj);   // (an odd tile count: the last pair's second half idles in step, nothing of it is stored)
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
    const int i0 = ti * T, j0 = tj * T;
    const bool offdiag = ti != tj;
    float acc[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) acc[r][c] = 0.f;
    // ---- K loop (as sk_tile_kernel)
    float4 pi, pj;
    auto gload = [&](int k0) __attribute__((always_inline)) {
      pi = load_row4(a.X, i0 + rb, a.ldx, k0 + 4 * kq, a.D, i0 + rb < a.row_end, vec4);
      pj = load_row4(a.X, j0 + rb, a.ldx, k0 + 4 * kq, a.D, j0 + rb < a.B, vec4);
    };
    gload(0);
    for (int k0 = 0; k0 < a.D; k0 += KC) {
      __syncthreads();   // (first chunk: the pair in front has folded its partial sums out of the area)
      *reinterpret_cast<float4*>(&Xi_s[rb * LS + 4 * kq]) = PMODE == 2 ? phase4(pi, a.rp_hi, a.rp_lo) : pi;
      *reinterpret_cast<float4*>(&Xj_s[rb * LS + 4 * kq]) = PMODE == 2 ? phase4(pj, a.rp_hi, a.rp_lo) : pj;
      __syncthreads();
      if (k0 + KC < a.D) gload(k0 + KC);
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
    if constexpr (PMODE == 2) {
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c) acc[r][c] *= a.phase_unit2;
    }
    // ---- the latent rows: published by the chain workgroups of this launch (first pair only)
    if (!latent_ready) {
      if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&tr.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tr.n_chain) {
          __builtin_amdgcn_s_sleep(20);
          if (++spins > TILE_ROLE_SPIN_LIMIT) {
            __hip_atomic_store(&tr.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      latent_ready = true;
    }
    __syncthreads();  // the counter has been seen; staging LDS is reused for the partial sums below
    // ---- epilogue (as sk_tile_kernel, LT = 2)
    float zi[RM][LMAX], zj[RN][LMAX], wr[RM], wc[RN];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int row = i0 + ty + 16 * r;
      const bool ok = row < a.row_end;
      wr[r] = ok ? 1.f : 0.f;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) zi[r][l] = ok ? load_sc1(tr.zpub + (long)row * L + l) : 0.f;
    }
#pragma unroll
    for (int c = 0; c < RN; ++c) {
      const int row = j0 + tx + 16 * c;
      const bool ok = row < a.B;
      wc[c] = ok ? 1.f : 0.f;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) zj[c][l] = ok ? load_sc1(tr.zpub + (long)row * L + l) : 0.f;
    }
    float* rpart = smem;                       // [T][LMAX][16]
    float* cpart = smem + 16 * T * LMAX;       // [T][LMAX][16]
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
        const float Eh = sig_E_high<FASTSIG>(acc[r][c], a.sh);
        float El, G;
        sig_E_low<FASTSIG>(dl2, a.sl, El, G);
        const float f = (El - Eh) * (wr[r] * wc[c]);
        lsum = fmaf(f, f, lsum);
        const float g = f * G;
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
    // tile cost: block_sum<256>'s order on this half's four waves
    {
      const float v = wave_sum(lsum);
      if (lane == 0) red[wave] = v;
    }
    __syncthreads();   // partials and wave sums visible
    if (tid == 0 && valid) {
      float tl = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) tl += red[i];
      a.tile_loss[tslot] = offdiag ? 2.f * tl : tl;
    }
    if (valid) {
      if (tid < T * LMAX) {
        const int rl = tid / LMAX, l = tid - rl * LMAX;
        const float4* p = reinterpret_cast<const float4*>(&rpart[tid * 16]);
        const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
        const float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                        (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
        store_sc1(&a.P[((long)tj * a.nrows_pad + i0 + rl) * L + l], s);
      }
      if (offdiag && tid < T * LMAX) {
        const int rl = tid / LMAX, l = tid - rl * LMAX;
        const float4* p = reinterpret_cast<const float4*>(&cpart[tid * 16]);
        const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
        const float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w)) +
                        (((s2.x + s2.y) + (s2.z + s2.w)) + ((s3.x + s3.y) + (s3.z + s3.w)));
        store_sc1(&a.P[((long)ti * a.nrows_pad + j0 + rl) * L + l], s);
      }
    }
  }
  if (tr.join) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(&tr.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)tr.n_tile_wgs - 1) {
      __hip_atomic_store(&tr.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&tr.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tr.join) __hip_atomic_store(&tr.sync[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// host side (sketchmap.hip): the tile plan of emx_sketchmap_tiles for the full batch on split rows, as a role description
// kind: 0 = Gram tiles on split rows (Euclidean), 1 / 2 = 32 x 32 VALU tiles of periodic rows, float form / 32-bit phases
int make_tile_role(const float* X, int B, int D, int ldx, const float* Z, int L, float periodicity, float data_span, const float sig[6], void* ws,
                   size_t ws_bytes, void* split_ws, TileRole* out, int* virtual_blocks, int* n_chunks, bool* fast, int* kind, const char* who);

}  // namespace sk
}  // namespace emx
