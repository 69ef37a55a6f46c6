// The fused dense-layer launch (forward + auto cost + decoder backward) of SMALL networks as a TRANSPOSED, register-resident chain
// (included by mlp.hip, inside its namespace).
//
// The row-block chain kernels (chain_body) compute H = X W with the batch rows as the M dimension of v_mfma_f32_16x16x32_f16: a
// layer's output leaves the matrix unit with lane = FEATURE, the next layer wants lane = ROW, so every layer goes accumulator ->
// LDS image -> barrier of eight waves -> transposed read (~3 000 cycles per 128 x 128 layer, most of them that round trip).
// Here H^T = W^T X^T: the weight pieces of the operand store are used AS THEY STAND as the A operand (lane (m, g) of a piece holds
// the eight k of wop.h's kk(g, e) for out-feature m -- the same registers a B fragment holds for column n), the 16 batch rows of a
// block are the N dimension, and D comes out as lane (n = row, g) holding out-features 16 t + 4 g .. + 3 of tile t.  With
// kk(g, e) = 4 g + e, 16 + 4 g + (e - 4) the accumulators of tiles 2 s and 2 s + 1 in lane (n, g) ARE the eight k that lane
// supplies for K-step s of the next layer: bias, activation and the f16 split happen in place.  EIGHT waves carry a row block, two
// per SIMD (with one the waves stalled on the issue of their own weight requests with nobody to run in the gaps): wave (w, hf) runs
// K-step w of tiles 4 hf .. 4 hf + 3 (12 matrix instructions per layer) and leaves its partial sums in LDS ([tile][K-step], one
// barrier per layer, two copies of the area by layer parity); BOTH waves of a pair (w, 0), (w, 1) then add the four K-steps' shares of
// tiles 2 w, 2 w + 1 -- the operand of their K-step of the next layer, which each of them needs -- and finalise them identically;
// that 16 x 32 block is exactly one block of the weight-gradient fragments (aop.h): wave hf of the pair leaves plane hf, transposed
// through a wave-private LDS area (ds_read_b64_tr_b16) and stores tile hf of the pair's two as float32 rows.  A wave's 8 pieces of a layer travel a whole layer ahead in a register ring of two,
// requested in quarters between the phases of the step.  The backward GEMMs are the same picture on the backward units (delta^T_{l-1} = W_l delta^T_l), the
// activation derivative taken from the layer's output kept in the SAME register layout.  The prologue is written for latency: the
// step records come through the scalar cache (wave-uniform indices), every memory request of the launch's head -- input rows,
// targets, the first layer's pieces, all steps' biases and weight scales (-> LDS tables) -- is in flight before the first wait, the
// launch's scalars are read once (docs/LAB_NOTEBOOK.md section 4 round 5 (d) has the stamps that asked for each of these).
// The default fused launch for: every width of the range <= 128 (one unit per layer), tanh / linear layers, a non-periodic output with
// mean_abs or mean_square, the whole network forward and the backward pass down to bwd_begin, at most one row block per compute unit
// (mlp.hip: try_tchain).  Arithmetic: three f16 products per float32 product on operands split into two planes with a power-of-two
// scale per (row, 32 features) -- the chain kernels' precision class, another order of summation.
#pragma once

constexpr int TC_MAXL = EMX_MAX_LAYERS;
constexpr int TC_KEEP = 4;   // layer outputs kept for the backward pass's activation derivatives (LDS, 8 KB each)

// One step of the chain = one unit of the operand store; everything a step's epilogue needs, as ONE 64-byte record (a single scalar
// load at the head of the step: fetched field by field from tables, each behind its own wait, the lookups were 1 200 of a step's
// 5 000 cycles).
enum { TS_FWD = 0, TS_FWD_TOP = 1, TS_BWD = 2, TS_BWD_BELOW = 3 };
struct TStep {
  const float* bias;     // forward: the layer's bias
  float* rows;           // float32 rows of the step's result (h_l, delta_{l-1}, or dh below the range); NULL: not kept
  char* frag;            // its weight-gradient fragments; NULL: not kept
  float* rscale;
  const float* rsw;      // the layer's reciprocal weight scale (header of the operand store)
  int width;             // columns of the result that exist
  int act;               // forward: the layer's activation; backward: the activation whose derivative the result is multiplied with
  int keep;              // forward: LDS slot the output is kept in for the backward pass (-1: none); backward: the slot to read
  int CG;                // column groups of the fragment tensor
  int kind;              // TS_*
  int pad;
};
static_assert(sizeof(TStep) == 64, "TStep is one s_load_dwordx16");

struct TChainArgs {
  TStep steps[2 * TC_MAXL];
  int nsteps, nl;
  const float* in;
  int ld_in, in_cols;
  char* in_frag;           // fragments of the input (tensor 0); NULL: not kept
  float* in_rscale;
  int in_CG, pad0;
  const char* units_fwd;   // unit of layer 0; layer l: + l * 64 KB
  const char* units_bwd;   // unit of the top layer in the backward direction; layer top - j: + j * 64 KB
  const float* target;
  float* top_rows;         // the output rows h_top as float32 rows; NULL: not kept
  float* cost_ws;
  int ldt, cost_n, cost_off, cost_variant;
  float gnorm;
  int row0, row_end, pad1;
};

constexpr int TC_XCH = 2 * 8 * 4 * 64 * 16;               // bytes: [parity][tile][K-step][lane] x 16
constexpr int TC_KEEP_BYTES = TC_KEEP * 4 * 8 * 64 * 4;   // [slot][wave][e][lane] floats
constexpr int TC_FRAG = 8 * 1024;                         // a wave's transposition area: one plane [32 features][16 rows] f16
constexpr int TC_TAB = 2 * TC_MAXL * (128 + 32) * 4;      // every step's bias (128 floats) and reciprocal weight scale (x 32)
constexpr int TC_LDS = TC_XCH + TC_KEEP_BYTES + TC_FRAG + 64 + TC_TAB;

// (the argument record is indexed at run time -- layer tables -- so it is read through the kernel-argument segment itself, scalar
//  loads from constant memory: indexing the by-value struct makes hipcc copy all of it to scratch)
typedef const TChainArgs __attribute__((address_space(4))) * TChainK;
#ifdef TC_STAMP
__device__ unsigned long long g_tc_stamps[8][32][8];
#define TCS(step, k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && lane == 0) g_tc_stamps[wave8][step][k] = t_; } while (0)
#define TCSV(step, k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TCS(step, k); } while (0)
#else
#define TCSV(step, k) do {} while (0)
#define TCS(step, k) do {} while (0)
#endif

// (ONE copy of the body for the eight waves -- the wave index and the pair's half are scalar registers, the exchange goes through LDS
//  addresses computed from them: 15 KB of code.  The first form instantiated the body per wave, ~80 KB for a 64-KB instruction cache
//  that the step's other three launches have just used.)
__device__ __forceinline__ void tchain_wave(const TChainK ka, char* lds) {
  const auto& a = *ka;
  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave8 & 3;          // the K-step this wave multiplies, and the pair of tiles (= 32 features) it finalises
  const int hf = wave8 >> 2;           // its half of the tiles in the products: tiles 4 hf .. 4 hf + 3
  const int n = lane & 15, g = lane >> 4;
  const int r0 = a.row0 + blockIdx.x * R;
  const int rb = r0 >> 4;
  const int rows_valid = min(R, a.row_end - r0);
  const bool rok = n < rows_valid;
  const long grow = (long)min(r0 + n, a.row_end - 1);
  f32x4* const xch = reinterpret_cast<f32x4*>(lds);
  float* const keep = reinterpret_cast<float*>(lds + TC_XCH);
  char* const fstage = lds + TC_XCH + TC_KEEP_BYTES + wave8 * 1024;   // the wave's own transposition area: one plane [32 features][16 rows] f16
  float* const red = reinterpret_cast<float*>(lds + TC_XCH + TC_KEEP_BYTES + TC_FRAG);
  float* const btab = red + 16;                        // [step][128]
  float* const rtab = btab + 2 * TC_MAXL * 128;        // [step][32 copies]
  // feature (column) of element e of this lane inside the wave's block of 32: tiles 2 w and 2 w + 1, rows 4 g .. 4 g + 3 of D
  auto fcol = [&](int e) { return 32 * wave + 16 * (e >> 2) + 4 * g + (e & 3); };

  half8 bh, bl;        // the operand of the wave's K-step: eight k of this lane's row
  float brs = 1.f;     // its reciprocal scale (per row and 32 features)
  unsigned bmax = 0;   // bit pattern of the row's largest |value| over those 32 features
  bool bfixed = false; // the operand holds tanh outputs at the fixed scale 2^14 (no maxima, partial sums need no unscaling)
  auto make_operand = [&](const float (&v)[8], const bool fixed) __attribute__((always_inline)) {
    float s = 16384.f;
    bfixed = fixed;
    if (fixed) {
      brs = TANH_RS;
    } else {
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) m = max(m, abs_bits(v[e]));
      m = max_xor32(max_xor16(m));                     // the row's maximum over the four lanes (g) that hold its 32 features
      bmax = m;
      pow2_scale_bits<ACT_TOP>(m, s, brs);
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      unsigned ph, pl;
      split_f16_pair(v[e] * s, v[e + 1] * s, ph, pl);   // (three instructions a pair: emx_common.h)
      const half2v vh = __builtin_bit_cast(half2v, ph), vl = __builtin_bit_cast(half2v, pl);
      bh[e] = vh[0], bh[e + 1] = vh[1], bl[e] = vl[0], bl[e + 1] = vl[1];
    }
  };
  // The wave's 16 x 32 block of the tensor the CURRENT operand was made from, as weight-gradient fragments (aop.h): the row-scaled
  // planes shifted to the block's scale (a power of two <= 1 per row: exact while the low plane stays above 2^-24, as in the
  // chain kernels' stage_finish), written row-major [row][16 features] per tile (one 8-byte LDS store per plane and tile) and read
  // back TRANSPOSED (ds_read_b64_tr_b16: lane i of a 16-lane group receives feature i of four rows): lane (feature, hh) ends with
  // rows 8 hh .. 8 hh + 7 of its feature -- one fragment lane.  The wave's own LDS area; nobody else touches it.
  auto emit_fragments = [&](const AopTensor& ao) __attribute__((always_inline)) {
    if (ao.frag == nullptr || wave >= ao.CG) return;
    // (the pair's two waves hold the same operand: wave hf leaves plane hf)
    float rsb = TANH_RS;
    half8 fp = hf ? bl : bh;
    if (!bfixed) {
      const unsigned bm = row16_max(bmax);             // the block's maximum: over the 16 rows (lanes n of a DPP row; every g agrees)
      float sb;
      pow2_scale_bits<ACT_TOP>(bm, sb, rsb);
      const _Float16 ratio = (_Float16)(sb * brs);
#pragma unroll
      for (int e = 0; e < 8; ++e) fp[e] = fp[e] * ratio;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)   // tile q of the pair's two: [16 rows][16 features] f16 = 512 bytes
      *reinterpret_cast<half4*>(fstage + q * 512 + n * 32 + g * 8) = half4{fp[4 * q], fp[4 * q + 1], fp[4 * q + 2], fp[4 * q + 3]};
    __builtin_amdgcn_wave_barrier();
    // group G = lane >> 4 of the wave: tile G & 1, rows 8 (G >> 1) ..; lane 4 qq + p of the group addresses row qq, features 4 p ..
    const int G = lane >> 4, qq = (lane & 15) >> 2, p = lane & 3;
    const char* src = fstage + (G & 1) * 512 + (8 * (G >> 1) + qq) * 32 + p * 8;
    const half4 h0 = lds_read_tr16(reinterpret_cast<const _Float16*>(src));
    const half4 h1 = lds_read_tr16(reinterpret_cast<const _Float16*>(src + 4 * 32));
    __builtin_amdgcn_wave_barrier();
    // lane' = lane: feature 16 (G & 1) + (lane & 15) = lane & 31, hh = G >> 1 = lane >> 5: the fragment's lane map
    char* f = ao.frag + ((long)rb * ao.CG + wave) * 2048 + hf * 1024 + lane * 16;
    *reinterpret_cast<half8*>(f) = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    if (lane == 0 && hf == 0) ao.rscale[(long)rb * ao.CG + wave] = rsb;
  };
  // the wave's block as float32 rows [*, width]
  auto store_rows = [&](const float (&v)[8], float* dst, int width) __attribute__((always_inline)) {
    if (dst == nullptr || !rok) return;
    // (the pair's two waves hold the same block: wave hf stores tile hf of the two.  A tile's four features of a lane are contiguous:
    //  one 16-byte store where the row length allows it)
    const int c = 32 * wave + 16 * hf + 4 * g;
    const f32x4 t = hf ? f32x4{v[4], v[5], v[6], v[7]} : f32x4{v[0], v[1], v[2], v[3]};
    if (c + 3 < width && (width & 3) == 0 && (reinterpret_cast<unsigned long long>(dst) & 15) == 0) {   // (the C API takes any row pointer)
      *reinterpret_cast<f32x4*>(dst + grow * width + c) = t;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c + r < width) dst[grow * width + c + r] = t[r];
    }
  };

  // ---- everything the steps' epilogues read from memory, once, side by side at the head of the launch (a row block is a serial
  //      chain of ~10 steps on one CU: whatever a step asks memory for is a trip to L2 on its critical path).  Thread (step, j) of a
  //      pass of 16 steps fetches its record's fields as ordinary memory (one trip), then four biases (a second): two trips in all.
  const int nsteps = a.nsteps;
  // (the launch's scalars ONCE, here: read at their places of use the compiler fetched them again per element, each fetch a scalar
  //  load with a full wait inside a branch -- the cost phase alone took 3 800 cycles)
  const int cost_off = a.cost_off, cost_n = a.cost_n, cost_variant = a.cost_variant, in_cols = a.in_cols;
  const float gnorm = a.gnorm;
  float* const top_rows = a.top_rows;
  TCS(30, 0);
  half8 wb[2][4][2];   // a register ring of two steps: the wave's pieces of its four tiles
  auto unit_of = [&](int step) -> const char* {
    const int s = min(step, nsteps - 1);
    return s < a.nl ? a.units_fwd + (long)s * WOP_UNIT : a.units_bwd + (long)(s - a.nl) * WOP_UNIT;
  };
  // A wave's 16 pieces of the NEXT step, a quarter at a time: the four waves of a workgroup pull 64 KB per step through the CU's
  // 64-byte-per-clock load path (>= 1 024 cycles) -- requested in one burst at the head of the step the waves stall on issue behind
  // their own requests (the products phase took 1 200 cycles instead of 400); in quarters between the phases the stream runs beside
  // the work
  auto request_q = [&](const int slot, const int step, const int q) __attribute__((always_inline)) {
    const char* u = unit_of(step) + (4 * hf) * 8192 + wave * 2048 + lane * 16;
    wb[slot][q][0] = *reinterpret_cast<const half8*>(u + q * 8192);
    wb[slot][q][1] = *reinterpret_cast<const half8*>(u + q * 8192 + 1024);
  };
  // the block's input rows (operand of K-step `wave`, fragments of tensor 0) and the cost's targets: 16 bytes per lane and tile where
  // the rows allow it (a lane's four features of a tile are contiguous) -- as single floats the 16 requests of a wave touch 16 lines
  // each, 2 000 cycles of the CU's load path before the first product
  auto load_block = [&](const float* base, const long ld, const int off, const int ncol, float (&x)[8]) __attribute__((always_inline)) {
    const bool vec = ((ld | off | ncol) & 3) == 0 && (reinterpret_cast<unsigned long long>(base) & 15) == 0 && ncol >= 4;
    if (vec) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = 32 * wave + 16 * q + 4 * g - off;
        const f32x4 t = *reinterpret_cast<const f32x4*>(base + grow * ld + min(max(c, 0), ncol - 4));
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * q + r] = t[r];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = base[grow * ld + min(max(fcol(e) - off, 0), ncol - 1)];   // (clamped: the VALUES are masked)
    }
  };
  // The tables: wave w8 fetches steps w8, w8 + 8, w8 + 16 -- the step is wave-uniform, so the records come through the scalar cache
  // (the argument segment), ahead of everything; the biases (two per lane and step) and scales are the one vector trip, requested
  // behind the input rows and the first layer's pieces.  No conditions: indices beyond the last step repeat it (same values, same
  // places), so that all of it is one basic block and every request is in flight before the first wait.
  struct Trip { const float* bp; const float* rp; int kind, w, st; float rv; };
  static_assert(2 * TC_MAXL <= 8 * 3, "the tables of 2 TC_MAXL steps are filled by 3 trips of the 8 waves");
  Trip tr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int st = min(wave8 + 8 * i, nsteps - 1);
    const auto& K = a.steps[st];
    tr[i] = Trip{K.bias, K.rsw, K.kind, K.width, st, 0.f};
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) tr[i].rv = *tr[i].rp;
  float tv[8], x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) tv[e] = 0.f;
  load_block(a.in, a.ld_in, 0, in_cols, x);
  if (a.target != nullptr) load_block(a.target, a.ldt, cost_off, cost_n, tv);
#pragma unroll
  for (int q = 0; q < 4; ++q) request_q(0, 0, q);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const Trip& t = tr[i];
    const bool has = t.kind <= TS_FWD_TOP && t.bp != nullptr;
    const float* b2 = has ? t.bp : t.rp;                // (an address that exists either way: no conditional loads)
    const int wm = has ? t.w - 1 : 0;
    const float b0 = b2[min(lane, wm)], b1 = b2[min(64 + lane, wm)];
    const float rv = t.rv;
    btab[t.st * 128 + lane] = has ? b0 : 0.f;
    btab[t.st * 128 + 64 + lane] = has ? b1 : 0.f;
    rtab[t.st * 32 + (lane & 31)] = rv;
  }
  TCS(30, 3);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (rok & (fcol(e) < in_cols)) ? x[e] : 0.f;
  TCS(30, 4);
  make_operand(x, false);
  TCS(30, 5);
  float lsum = 0.f;
  // (no barrier here: the tables are read behind the first step's barrier; the input's fragments leave behind its products)

  auto step_fn = [&](const int step, auto SLOT_) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT_)::value;
    if (step >= nsteps) return;                       // (workgroup-uniform)
    TCS(step, 0);
    // the step's record (scalar cache; its biases and weight scale wait in the prologue's LDS tables: nothing of a step's epilogue is
    // a trip to memory)
    TStep S;
    {
      const auto& K = a.steps[step];                  // (constant address space: field by field, the loads merge into one)
      S.bias = K.bias, S.rows = K.rows, S.frag = K.frag, S.rscale = K.rscale, S.rsw = K.rsw;
      S.width = K.width, S.act = K.act, S.keep = K.keep, S.CG = K.CG, S.kind = K.kind;
    }
    request_q(1 - slot, step + 1, 0);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[slot][t][0], bl, acc[t], 0, 0, 0);   // small products first
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[slot][t][1], bh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[slot][t][0], bh, acc[t], 0, 0, 0);
    }
    // the wave's partial sums of its four tiles (K-step `wave`) -> LDS, [tile][K-step][lane]; unscaled by this K-step's row scale first
    f32x4* const xc = xch + slot * (8 * 4 * 64);
    const bool fixed_in = bfixed;                      // (workgroup-uniform: every wave's operand came from the same layer)
    if (!fixed_in) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] *= brs;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) xc[((4 * hf + t) * 4 + wave) * 64 + lane] = acc[t];
    request_q(1 - slot, step + 1, 1);
    if (slot == 0 && step == 0) emit_fragments(AopTensor{a.in_frag, a.in_rscale, a.in_CG});   // (the operand is the input's until the epilogue)
    TCS(step, 1);
    __syncthreads();
    TCS(step, 2);
    // the step's bias and weight scale from the tables of the prologue (written in front of this barrier at the latest)
    const float rsw = rtab[step * 32];
    float ev[8];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(btab + step * 128 + 32 * wave + 4 * g);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(btab + step * 128 + 32 * wave + 16 + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[e] = b0[e], ev[4 + e] = b1[e];
    }
    // both waves of a pair finalise tiles 2 wave and 2 wave + 1: the four K-steps' shares in the order 0, 1, 2, 3
    f32x4 s0 = xc[((2 * wave) * 4 + 0) * 64 + lane], s1 = xc[((2 * wave + 1) * 4 + 0) * 64 + lane];
#pragma unroll
    for (int src = 1; src < 4; ++src) {
      s0 += xc[((2 * wave) * 4 + src) * 64 + lane];
      s1 += xc[((2 * wave + 1) * 4 + src) * 64 + lane];
    }
    request_q(1 - slot, step + 1, 2);
    TCS(step, 3);
    float v[8];
    const float cm = fixed_in ? rsw * TANH_RS : rsw;
    // (b) a full block of a full-width layer needs no masks
    const bool full = rows_valid == R && S.width >= 32 * wave + 32;
    if (S.kind <= TS_FWD_TOP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = fmaf(e < 4 ? s0[e & 3] : s1[e & 3], cm, ev[e]);
        if (S.act == ACT_TANH) x = tanh_f32(x);
        v[e] = x;
      }
      if (!full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (rok & (fcol(e) < S.width)) ? v[e] : 0.f;
      }
      if (S.kind == TS_FWD) {
        store_rows(v, S.rows, S.width);
        if (S.keep >= 0) {
          f32x4* k = reinterpret_cast<f32x4*>(keep) + ((S.keep * 4 + wave) * 2) * 64 + lane;
          k[64 * hf] = hf ? f32x4{v[4], v[5], v[6], v[7]} : f32x4{v[0], v[1], v[2], v[3]};
        }
      } else {
        store_rows(v, top_rows, S.width);                // (the output itself, where somebody reads it)
        // ---- the auto cost on the output rows (the step's rows pointer is d cost / d output's: delta_top replaces the output in
        //      place).  Selects, no branches: both
        //      variants' terms are three instructions
        const bool abs_variant = cost_variant == EMX_MEAN_ABS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = fcol(e);
          const bool live = rok & (c >= cost_off) & (c < cost_off + cost_n);
          const float diff = v[e] - tv[e];
          const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
          const float term = abs_variant ? fabsf(diff) : diff * diff;
          const float q = abs_variant ? sgn : 2.f * diff;
          lsum += (live & (hf == 0)) ? term : 0.f;
          v[e] = live ? gnorm * q : 0.f;
        }
        {   // the row block's cost share: wave sums, then the four waves in turn
          const float ws = wave_sum(lsum);
          if (lane == 0 && hf == 0) red[wave] = ws;
        }
        store_rows(v, S.rows, S.width);
      }
    } else {
      // backward: dh of the layer below, times the activation derivative of that layer's output (kept in this register layout)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = (e < 4 ? s0[e & 3] : s1[e & 3]) * cm;
      }
      if (!full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (rok & (fcol(e) < S.width)) ? v[e] : 0.f;
      }
      if (S.act == ACT_TANH) {
        const f32x4* k = reinterpret_cast<const f32x4*>(keep) + ((S.keep * 4 + wave) * 2) * 64 + lane;
        const f32x4 h0 = k[0], h1 = k[64];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float hv = e < 4 ? h0[e & 3] : h1[e & 3];
          v[e] *= fmaf(-hv, hv, 1.f);
        }
      }
      store_rows(v, S.rows, S.width);
    }
    request_q(1 - slot, step + 1, 3);
    TCS(step, 4);
    make_operand(v, S.kind == TS_FWD && S.act == ACT_TANH);
    TCS(step, 5);
    emit_fragments(AopTensor{S.frag, S.rscale, S.CG});
    TCS(step, 6);
  };
  for (int step = 0; step < nsteps; step += 2) {
    step_fn(step, IC_<0>{});
    step_fn(step + 1, IC_<1>{});
  }
  TCS(31, 0);
  __syncthreads();
  if (threadIdx.x == 0 && a.cost_ws != nullptr) a.cost_ws[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) * gnorm;
}

__global__ __launch_bounds__(512) void tchain_fused_kernel(const TChainArgs) {
  extern __shared__ __align__(16) char lds[];
  const TChainK ka = (TChainK)__builtin_amdgcn_kernarg_segment_ptr();
  tchain_wave(ka, lds);
}
