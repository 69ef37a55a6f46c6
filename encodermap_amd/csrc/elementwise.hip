// Data feed, periodic transforms, row-mean costs, center cost, fused Adam and step bookkeeping (gfx950).
// Every kernel here is HBM- or launch-latency-bound: rows are walked by whole waves with 16-byte lanes
// where the layout allows it, scalar losses use fixed-order two-pass sums (no float atomics).
#include <mutex>
#include <unordered_map>
#include <vector>

#include "emx_common.h"
#include "cost.h"
#include "wop.h"
#include "split.h"
#include "adc_head.h"

namespace {

using namespace emx;

constexpr float TWO_PI = 6.283185307179586f;
constexpr int NT = 256;

thread_local char g_err[512] = "";

}  // namespace

namespace emx {
char* error_buffer() { return g_err; }
long long& launch_counter() {
  static long long n = 0;
  return n;
}

namespace {
std::mutex g_counters_mu;
std::unordered_map<const void*, size_t> g_counters_seen;   // pointer -> bytes zeroed
}  // namespace

int ensure_counters_zeroed(void* words, size_t bytes, hipStream_t stream, const char* who) {
  if (words == nullptr || bytes == 0) return EMX_OK;
  std::lock_guard<std::mutex> lock(g_counters_mu);
  auto it = g_counters_seen.find(words);
  if (it != g_counters_seen.end() && it->second >= bytes) return EMX_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone)
    return fail(EMX_EINVAL, "%s: the counter words at %p are new to the library and the stream is capturing -- run the call once outside the "
                "capture first (a warm-up step), so that the words are zeroed in front of a launch and not on every replay", who, words);
  if (hipMemsetAsync(words, 0, bytes, stream) != hipSuccess) {
    (void)hipGetLastError();
    return fail(EMX_EINVAL, "%s: could not zero the counter words at %p (%zu bytes): not device memory?", who, words, bytes);
  }
  if (g_counters_seen.size() > 65536) g_counters_seen.clear();   // (a long-lived process that keeps allocating: start over, a memset each)
  g_counters_seen[words] = bytes;
  return EMX_OK;
}
}  // namespace emx

namespace emx {
namespace {
// Ticket words of the split reductions (emx_sketchmap_tiles, kmode 3): library-owned device memory, one block per scratch address.
// Blocks are never handed back to the runtime while the process lives (a launch in flight may still count in one); a block whose
// scratch address was forgotten (emx_forget_counters) waits in the pool and is ZEROED AGAIN, in stream order, for its next owner.
std::unordered_map<const void*, std::pair<unsigned*, size_t>> g_tickets;   // scratch address -> (block, bytes)
std::vector<std::pair<unsigned*, size_t>> g_ticket_pool;                   // blocks without an owner, free to be handed out again
std::vector<std::pair<unsigned*, size_t>> g_ticket_parked;                 // blocks without an owner that are NEVER handed out again

// A block goes back to the pool only when its OWNER says so (emx_forget_counters(scratch): the caller's statement that no launch in
// flight and no captured graph still addresses the scratch buffer -- include/emx.h).  A block the LIBRARY takes away from an address
// on its own -- too small for a later call with the same scratch, or emx_forget_counters(NULL) -- is parked for the life of the
// process instead: a hipGraph captured earlier, or a launch in flight on another stream, may still hold its pointer, and handed to
// another scratch address two owners would share ticket words (a few KB per such event; there is no cap on live owners any more).
void retire_tickets_locked(const void* scratch, bool reusable) {
  if (scratch == nullptr) {
    for (auto& kv : g_tickets) g_ticket_parked.push_back(kv.second);
    g_tickets.clear();
    return;
  }
  auto it = g_tickets.find(scratch);
  if (it == g_tickets.end()) return;
  (reusable ? g_ticket_pool : g_ticket_parked).push_back(it->second);
  g_tickets.erase(it);
}
}  // namespace

// The ticket block of `scratch`.  First sight of an address (per process, or after emx_forget_counters): a block is taken from the
// pool -- or allocated with hipMalloc, which is why first sight must not happen inside a stream capture, and should not while
// ANOTHER thread captures in global mode (hipMalloc is not a capturable call) -- and zeroed on `stream` in front of the launch.
unsigned* owned_tickets(const void* scratch, size_t bytes, hipStream_t stream, const char* who, int* rc) {
  *rc = EMX_OK;
  std::lock_guard<std::mutex> lock(g_counters_mu);
  auto it = g_tickets.find(scratch);
  if (it != g_tickets.end() && it->second.second >= bytes) return it->second.first;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) {
    *rc = fail(EMX_EINVAL, "%s: the scratch buffer at %p is new to the library and the stream is capturing -- run the call once outside the "
               "capture first (a warm-up step): its ticket words are allocated at first sight", who, scratch);
    return nullptr;
  }
  if (it != g_tickets.end()) retire_tickets_locked(scratch, false);     // (too small for this call: parked, a graph may still hold it)
  unsigned* block = nullptr;
  size_t have = 0;
  for (size_t i = 0; i < g_ticket_pool.size(); ++i)
    if (g_ticket_pool[i].second >= bytes && (block == nullptr || g_ticket_pool[i].second < have)) {   // smallest block that fits
      block = g_ticket_pool[i].first;
      have = g_ticket_pool[i].second;
    }
  if (block != nullptr) {
    for (size_t i = 0; i < g_ticket_pool.size(); ++i)
      if (g_ticket_pool[i].first == block) {
        g_ticket_pool.erase(g_ticket_pool.begin() + (long)i);
        break;
      }
  } else {
    have = bytes;
    if (hipMalloc(reinterpret_cast<void**>(&block), bytes) != hipSuccess) {
      (void)hipGetLastError();
      *rc = fail(EMX_EWORKSPACE, "%s: could not allocate %zu bytes of ticket words", who, bytes);
      return nullptr;
    }
  }
  if (hipMemsetAsync(block, 0, have, stream) != hipSuccess) {
    (void)hipGetLastError();
    g_ticket_pool.emplace_back(block, have);
    *rc = fail(EMX_EWORKSPACE, "%s: could not zero %zu bytes of ticket words", who, have);
    return nullptr;
  }
  g_tickets[scratch] = std::make_pair(block, have);
  return block;
}
}  // namespace emx

extern "C" void emx_forget_counters(void* words) {
  std::lock_guard<std::mutex> lock(emx::g_counters_mu);
  if (words == nullptr) emx::g_counters_seen.clear();
  else emx::g_counters_seen.erase(words);
  emx::retire_tickets_locked(words, true);   // (the owner's word that nothing addresses the block any more: zeroed again for its next owner)
}

namespace {

// ---------------------------------------------------------------------------------------------------
// K13 + K4: gather rows by index, optionally emit [sin | cos]      autoencoder.py:845-856, models.py:3347-3351
// one wave per row; lanes stride the feature axis (coalesced 256-B segments of the gathered row)
// ---------------------------------------------------------------------------------------------------
#ifdef EMX_CHAIN_TIMING   // in-kernel timelines of the two scalar-ish kernels of a step (tools/abl_finish.py)
__device__ long long emx_dbg_fin[64];
#define FSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) emx_dbg_fin[k] = clock64(); } while (0)
#define ASTAMP(k, blk) do { if (threadIdx.x == 0 && (int)blockIdx.x == (blk)) emx_dbg_fin[k] = wall_clock64(); } while (0)
#endif

}  // namespace
#ifndef ASTAMP
#define ASTAMP(k, blk)
#endif
#include "finish.h"   // (after FSTAMP)
#include "optimizer.h"
namespace {

// K13: the shuffle of the data feed as a keyed bijection (include/emx.h, emx_permute; host twin: layout.feistel_permutation)
struct PermArgs {
  int64_t* out;
  long n;
  int abits, bbits;        // bits of the left / right half-word
  unsigned key[6];
};
__host__ __device__ inline unsigned mix32(unsigned h) {   // murmur3 finaliser
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(NT) void permute_kernel(const PermArgs a) {
  const long i = (long)blockIdx.x * NT + threadIdx.x;
  if (i >= a.n) return;
  const unsigned long long amask = (1ull << a.abits) - 1, bmask = (1ull << a.bbits) - 1;
  unsigned long long x = (unsigned long long)i;
  do {   // cycle walking: the network permutes [0, 2^(abits + bbits)), which is < 2 n values
    unsigned long long l = x >> a.bbits, r = x & bmask;   // l: abits wide, r: bbits wide
#pragma unroll
    for (int k = 0; k < 6; k += 2) {
      l = (l ^ mix32((unsigned)r ^ a.key[k])) & amask;
      r = (r ^ mix32((unsigned)l ^ a.key[k + 1])) & bmask;
    }
    x = (l << a.bbits) | r;
  } while (x >= (unsigned long long)a.n);
  a.out[i] = (int64_t)x;
}

// (GatherArgs, gather_rows_body, gather_rows2_body: optimizer.h)

__global__ __launch_bounds__(NT) void gather_rows_kernel(const GatherArgs g) {
  gather_rows_body(g, (blockIdx.x * NT + threadIdx.x) >> 6, (gridDim.x * NT) >> 6, threadIdx.x & 63);
}

__global__ __launch_bounds__(NT) void periodic_in_kernel(const float* __restrict__ x, int B, int D, int ldx, float ang_scale,
                                                         float* __restrict__ out, int ldo) {
  long e = (long)blockIdx.x * NT + threadIdx.x;
  long n = (long)B * D;
  for (; e < n; e += (long)gridDim.x * NT) {
    int b = (int)(e / D), k = (int)(e - (long)b * D);
    float s, c;
    sincos_bounded(x[(long)b * ldx + k] * ang_scale, &s, &c);
    out[(long)b * ldo + k] = s;
    out[(long)b * ldo + D + k] = c;
  }
}

__global__ __launch_bounds__(NT) void periodic_out_kernel(const float* __restrict__ y, int B, int n, int ldy, int off_sin,
                                                          int off_cos, float out_scale, float* __restrict__ out, int ldo) {
  long e = (long)blockIdx.x * NT + threadIdx.x;
  long tot = (long)B * n;
  for (; e < tot; e += (long)gridDim.x * NT) {
    int b = (int)(e / n), k = (int)(e - (long)b * n);
    const float* row = y + (long)b * ldy;
    out[(long)b * ldo + k] = atan2f(row[off_sin + k], row[off_cos + k]) * out_scale;
  }
}

// ---------------------------------------------------------------------------------------------------
// K5: row-mean cost on periodic differences, fused with PeriodicOutput (atan2) forward + backward.
// one wave per row (two passes over the row for mean_norm); per-block partial -> ws, final sum below.
// ---------------------------------------------------------------------------------------------------
// (bid of nblk: this workgroup among those that work on the cost `a` -- the whole grid, or one job's share of a batched launch)
__device__ __forceinline__ void periodic_cost_body(const CostArgs& a, const int bid, const int nblk, float (&red)[8]) {
  // long rows (a.wide): a row per workgroup, the columns over all its threads -- a wave per row walks 45 dependent
  // iterations over a 2850-column row while most of the chip idles behind a 256-row batch
  // (a.wide = S workgroups per row: workgroup bid takes the columns (bid % S) NT + threadIdx.x, + S NT, ... of row bid / S)
  const int lane = a.wide ? (bid % a.wide) * NT + threadIdx.x : threadIdx.x & 63;
  const int kstep = a.wide ? a.wide * NT : 64;
  const int wave = a.wide ? bid / a.wide : (bid * NT + threadIdx.x) >> 6;
  const int nwaves = a.wide ? nblk / a.wide : (nblk * NT) >> 6;
  float lsum = 0.f;
  for (int b = a.row_begin + wave; b < a.row_end; b += nwaves) {
    const float* yrow = a.y + (long)b * a.ldy;
    const float* trow = a.target + (long)b * a.ldt;
    float* drow = a.dY ? a.dY + (long)b * a.ldy : nullptr;
    const float* addrow = a.dpred_add ? a.dpred_add + (long)b * a.n : nullptr;
    const float gn = a.coef_scale ? a.gnorm * *a.coef_scale : a.gnorm;
    if (a.variant == EMX_MEAN_NORM) {
      float ss = 0.f;
      for (int k = lane; k < a.n; k += 64) {
        float s, c, delta, dd;
        cost_elem(a, yrow, trow, k, s, c, delta, dd);
        ss = fmaf(delta, delta, ss);
      }
      ss = wave_sum(ss);
      float nrm = sqrtf(ss);
      float inv = nrm > 0.f ? 1.f / nrm : 0.f;
      if (lane == 0) lsum += nrm;
      for (int k = lane; k < a.n; k += 64) {
        float s, c, delta, dd;
        cost_elem(a, yrow, trow, k, s, c, delta, dd);
        cost_store(a, drow, addrow, k, gn * delta * inv * dd, s, c);
      }
    } else {
      for (int k = lane; k < a.n; k += kstep) {
        float s, c, delta, dd;
        cost_elem(a, yrow, trow, k, s, c, delta, dd);
        float g;
        if (a.variant == EMX_MEAN_ABS) {
          lsum += fabsf(delta);
          g = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
        } else {
          lsum = fmaf(delta, delta, lsum);
          g = 2.f * delta * dd;
        }
        cost_store(a, drow, addrow, k, gn * g, s, c);
      }
    }
  }
  float t = block_sum<NT>(lsum, red);
  if (threadIdx.x == 0) a.ws[bid] = a.prescale ? t * a.gnorm * (a.coef_scale ? *a.coef_scale : 1.f) : t;
  if (a.prescale && bid == 0)   // unused scratch entries read as zero by the summing job
    for (int i = nblk + threadIdx.x; i < EMX_REDUCE_WS; i += NT) a.ws[i] = 0.f;
}

__global__ __launch_bounds__(NT) void periodic_cost_kernel(const CostArgs a) {
  __shared__ float red[8];
  periodic_cost_body(a, blockIdx.x, gridDim.x, red);
}

// several row-mean costs of one step in one launch (emx_periodic_cost_batch): job j owns the workgroups [first[j], first[j + 1])
struct CostBatch {
  CostArgs job[EMX_MAX_COST_JOBS];
  int first[EMX_MAX_COST_JOBS + 1];
  int n_jobs;
};
__global__ __launch_bounds__(NT) void periodic_cost_batch_kernel(const CostBatch b) {
  __shared__ float red[8];
  int j = 0;
  while (j + 1 < b.n_jobs && (int)blockIdx.x >= b.first[j + 1]) ++j;
  periodic_cost_body(b.job[j], blockIdx.x - b.first[j], b.first[j + 1] - b.first[j], red);
}

// *out = scale * sum_{i<n} ws[i], fixed order
__global__ __launch_bounds__(NT) void final_sum_kernel(const float* __restrict__ ws, int n, float scale, const float* __restrict__ dev_scale,
                                                       float* __restrict__ out) {
  __shared__ float red[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += NT) s += ws[i];
  float t = block_sum<NT>(s, red);
  if (threadIdx.x == 0) *out = t * scale * (dev_scale ? *dev_scale : 1.f);
}

__global__ __launch_bounds__(NT) void center_cost_kernel(const float* __restrict__ Z, int L, int row_begin, int row_end,
                                                         float lscale, float gscale, float* __restrict__ loss,
                                                         float* __restrict__ dZ, int accumulate) {
  __shared__ float red[8];
  const int n = (row_end - row_begin) * L;
  const long base = (long)row_begin * L;
  float s = 0.f;
  for (int e = threadIdx.x; e < n; e += NT) {
    float z = Z[base + e];
    s = fmaf(z, z, s);
    float g = gscale * z;
    dZ[base + e] = accumulate ? dZ[base + e] + g : g;
  }
  float t = block_sum<NT>(s, red);
  if (threadIdx.x == 0) *loss = t * lscale;
}

// (AdamArgs, adam_one, grad4, grad1, adam_store_block: optimizer.h)

// Launched with one wave per workgroup: the update of ~1e5 parameters reads 36 streams (32 gradient slices, w, m, v,
// l2coef), ~150 KB per 1024 parameters, and a CU takes 64 B/clk -- on 65 CUs (256-thread workgroups) the loads alone were
// 2300+ cycles of L1 fill per CU; one wave per workgroup spreads them over every CU.
// The leading scalars (56 bytes, the struct follows) are preloaded into scalar registers at wave launch
// (-amdgpu-kernarg-preload-count): a workgroup knows its role, and the pointers of its first loads -- the step counter, the
// batch cursor -- without a trip to the argument segment; the plane's record (its address depends on the workgroup index
// alone) is the only trip in front of an optimizer wave's operand loads, where the role fields used to be a first one.
// (ADCH: the instantiation whose extra workgroups can do the head of the next ADC step, emx_next_batch::adc_head -- kept out of the
//  other one, which the EncoderMap step waits for: the branch costs that launch 0.3 us even when it is not taken)
template <bool ADCH>
__global__ __launch_bounds__(NT) void adam_kernel(int p_adam_blocks, int p_has_hist, int p_next_vec4, int p_batch_step, const int64_t* p_step,
                                                  const char* p_mirror, const int64_t* p_cursor, const int64_t* p_idx, int p_B, int p_grid,
                                                  const AdamArgs a) {   // (p_grid = gridDim.x: the hidden argument would be a trip of its own)
  constexpr int EXPLICIT_BYTES = 56;
  __shared__ float red[NT / 64];
  constexpr int nt = 64;   // = ADAM_NT, the launch's workgroup size (blockDim.x is a hidden argument: a trip)
  ASTAMP(32, 0);
  ASTAMP(40, p_adam_blocks);
  ASTAMP(42, p_grid - 2);
  if (p_has_hist && (int)blockIdx.x == p_grid - 1) {
    if (threadIdx.x == 0) {
      const long hs = *a.hstep - 1;
      float tl[64];
      for (int i = 0; i < a.row_n; ++i) tl[i] = a.row_terms[i];
      float tot = 0.f;
      for (int i = 0; i < a.row_n; ++i) tot += tl[i];
      if (hs >= 0 && hs < a.capacity) {
        if (a.history) a.history[hs] = tot;
        if (a.term_history)
          for (int i = 0; i < a.row_n; ++i) a.term_history[hs * a.row_n + i] = tl[i];
      }
    }
    return;
  }
  if ((int)blockIdx.x >= p_adam_blocks) {
    // the optimizer is a short, latency-bound stream over ~1e5 parameters; the rows of the NEXT batch are gathered
    // beside it, on other CUs, instead of in a launch of their own at the head of the next step
    const int gb = blockIdx.x - p_adam_blocks, ngb = p_grid - p_adam_blocks - (p_has_hist ? 1 : 0);
    GatherArgs gl = a.next;   // (cursor, index block and batch shape from the preloaded scalars: the cursor load goes out at once)
    gl.cursor = p_cursor;
    gl.idx = p_idx;
    gl.B = p_B;
    gl.batch_step = p_batch_step;
    if constexpr (ADCH) {
      // the head of the next ADC step (emx_adc_batch_head's work, adc_head.h): virtual workgroups of 256 threads for the rows (four
      // of this launch's one-wave workgroups each), then workgroups of 4 columns x 16 row phases for the batch means
      const emx::AdcHeadArgs& h = *gl.adc_head;
      const long batch = *p_cursor + p_batch_step;
      if (h.n_batches > 0 && batch >= h.n_batches) return;
      const int64_t* __restrict__ idx = p_idx + batch * (long)p_B;
      const int row_wgs = 4 * h.row_blocks;
      if (gb < row_wgs) {
        emx::adc_head_rows(h, idx, gb >> 2, h.row_blocks, (gb & 3) * 64 + (int)threadIdx.x, p_batch_step);
      } else if (h.feat_mean != nullptr) {
        __shared__ float part[emx::ADC_CM_PH][4];
        emx::adc_head_means<4>(h, idx, (gb - row_wgs) * 4, threadIdx.x, part);
      }
      return;
    }
    if (p_next_vec4) gather_rows2_body(gl, (gb * nt + threadIdx.x) >> 6, (ngb * nt) >> 6, threadIdx.x & 63);
    else gather_rows_body(gl, (gb * nt + threadIdx.x) >> 6, (ngb * nt) >> 6, threadIdx.x & 63);
    ASTAMP(41, p_adam_blocks);
    ASTAMP(43, p_grid - 2);
    return;
  }
  const int nblk = p_adam_blocks;
  FSTAMP(16);
  const long step_now = *p_step;   // requested first, consumed behind the parameter loads (one round trip, not two)
  // alpha = lr * sqrt(1 - b2^t) / (1 - b1^t)   (Keras 2.13-2.15 Adam.update_step); b^t = 2^(t log2 b) on the hardware
  // exp2 (the library powf is several hundred instructions in front of the first load of a latency-bound kernel)
  auto alpha_of = [&](long st) {
    const float t = (float)(st + (a.late ? 0 : 1));
    return a.lr * sqrtf(1.f - __builtin_amdgcn_exp2f(t * a.log2_b2)) / (1.f - __builtin_amdgcn_exp2f(t * a.log2_b1));
  };
  float reg = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(a.w) | reinterpret_cast<uintptr_t>(a.g) | reinterpret_cast<uintptr_t>(a.m) |
                     reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.l2coef) | reinterpret_cast<uintptr_t>(a.gpart) |
                     (uintptr_t)(a.gstride * 4)) & 15) == 0;
  if (p_mirror != nullptr) {
    __shared__ __align__(16) _Float16 img[512];
    typedef const void __attribute__((address_space(4))) * KArgs;
    typedef const char __attribute__((address_space(4))) * KBytes;
    reg = adam_store_block<false>(a, (KArgs)((KBytes)__builtin_amdgcn_kernarg_segment_ptr() + EXPLICIT_BYTES), p_mirror, (int)blockIdx.x,
                                  threadIdx.x & 63, step_now, img);
  } else {
    const long n4 = a.n >> 2;
    long start = 0;
    if (vec) {
      for (long i = (long)blockIdx.x * nt + threadIdx.x; i < n4; i += (long)nblk * nt) {
        float4 w = reinterpret_cast<float4*>(a.w)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float4 l = a.l2coef ? reinterpret_cast<const float4*>(a.l2coef)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = grad4(a, i);
        const float alpha = alpha_of(step_now);
        FSTAMP(17);
        adam_one(w.x, g.x, m.x, v.x, l.x, a, alpha, reg);
        adam_one(w.y, g.y, m.y, v.y, l.y, a, alpha, reg);
        adam_one(w.z, g.z, m.z, v.z, l.z, a, alpha, reg);
        adam_one(w.w, g.w, m.w, v.w, l.w, a, alpha, reg);
        reinterpret_cast<float4*>(a.w)[i] = w;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        FSTAMP(18);
      }
      start = n4 << 2;
    }
    for (long i = start + (long)blockIdx.x * nt + threadIdx.x; i < a.n; i += (long)nblk * nt) {
      float w = a.w[i], m = a.m[i], v = a.v[i];
      adam_one(w, grad1(a, i), m, v, a.l2coef ? a.l2coef[i] : 0.f, a, alpha_of(step_now), reg);
      a.w[i] = w;
      a.m[i] = m;
      a.v[i] = v;
    }
  }
  FSTAMP(19);
  ASTAMP(36, 0);
  reg = wave_sum(reg);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = reg;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tsum = 0.f;
    for (int wv = 0; wv < nt / 64; ++wv) tsum += red[wv];
    a.ws[blockIdx.x] = tsum;
  }
  FSTAMP(20);
  ASTAMP(37, 0);
  if (blockIdx.x == 0)
    for (int i = nblk + threadIdx.x; i < EMX_REDUCE_WS; i += nt) a.ws[i] = 0.f;
}

// regulariser shares of the CURRENT weights in the layout the optimizer leaves them in (ws[block], zeros up to
// EMX_REDUCE_WS): the start-up of the late-bookkeeping scheme (emx_l2_partials)
__global__ __launch_bounds__(NT) void l2_partials_kernel(const float* __restrict__ w, const float* __restrict__ l2coef, long n, float* __restrict__ ws) {
  __shared__ float red[NT / 64];
  const int nt = blockDim.x;
  float reg = 0.f;
  if (l2coef)
    for (long i = (long)blockIdx.x * nt + threadIdx.x; i < n; i += (long)gridDim.x * nt) reg = fmaf(0.5f * l2coef[i] * w[i], w[i], reg);
  reg = wave_sum(reg);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = reg;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int wv = 0; wv < nt / 64; ++wv) t += red[wv];
    ws[blockIdx.x] = t;
  }
  if (blockIdx.x == 0)
    for (int i = gridDim.x + threadIdx.x; i < EMX_REDUCE_WS; i += nt) ws[i] = 0.f;
}

__global__ void step_tick_kernel(const float* __restrict__ terms, int n_terms, float* __restrict__ history,
                                 float* __restrict__ term_history, long capacity, int64_t* __restrict__ step,
                                 int64_t* __restrict__ cursor, int64_t* __restrict__ counter2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long s = *step;
  float tot = 0.f;
  for (int i = 0; i < n_terms; ++i) tot += terms[i];
  if (s >= 0 && s < capacity) {
    if (history) history[s] = tot;
    if (term_history)
      for (int i = 0; i < n_terms; ++i) term_history[s * n_terms + i] = terms[i];
  }
  *step = s + 1;
  if (cursor) *cursor += 1;
  if (counter2) *counter2 += 1;
}

__global__ __launch_bounds__(FIN_NT) void finish_step_kernel(const FinishArgs a) {
  __shared__ float red[EMX_MAX_SUM_JOBS][FIN_NT / 64];
  __shared__ float tl[64];
  finish_step_body<emx::FQ>(a, red, tl);
}

inline int grid_for(long work_items, int per_block) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > EMX_REDUCE_WS) g = EMX_REDUCE_WS;
  return (int)g;
}

}  // namespace

extern "C" {

int emx_version(void) { return EMX_VERSION; }
const char* emx_last_error(void) { return g_err; }
const char* emx_target_arch(void) { return "gfx950"; }
int64_t emx_launch_count(void) { return (int64_t)emx::launch_counter(); }

int emx_permute(int64_t* out, int64_t n, uint64_t seed, uint64_t epoch, emx_stream_t stream) {
  EMX_REQUIRE(out && n > 0, "emx_permute: bad arguments");
  EMX_REQUIRE(n <= (1ll << 62), "emx_permute: n too large");
  PermArgs a{};
  a.out = out;
  a.n = (long)n;
  int bits = 2;
  while ((1ull << bits) < (unsigned long long)n) ++bits;
  a.abits = bits / 2;
  a.bbits = bits - a.abits;
  EMX_REQUIRE(a.bbits <= 32, "emx_permute: n too large");
  for (int k = 0; k < 6; ++k)
    a.key[k] = mix32((unsigned)(seed & 0xffffffffu) ^ mix32((unsigned)(seed >> 32) + 0x9e3779b9u * (unsigned)(k + 1)) ^
                     mix32((unsigned)(epoch & 0xffffffffu) + 0x7f4a7c15u * (unsigned)(k + 1)) ^ mix32((unsigned)(epoch >> 32) ^ 0x165667b1u));
  hipLaunchKernelGGL(permute_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, a);
  return emx::check_launch("permute_kernel");
}

int emx_gather_rows(const float* data, int64_t N, int D, const int64_t* idx, const int64_t* cursor, int B, float periodicity,
                    float* xb, float* xin, void* split_ws, emx_stream_t stream) {
  EMX_REQUIRE(data && idx && (xb || xin), "emx_gather_rows: null pointer argument");
  EMX_REQUIRE(N > 0 && D > 0 && B > 0, "emx_gather_rows: bad shape N=%ld D=%d B=%d", (long)N, D, B);
  EMX_REQUIRE(!xin || (std::isfinite(periodicity) && periodicity > 0.f), "emx_gather_rows: sin/cos output needs a finite periodicity");
  GatherArgs g{data, D, idx, cursor, B, 0, 0, xin ? (periodicity == TWO_PI ? 1.f : TWO_PI / periodicity) : 1.f, xb, xin, SplitRows{}};
  if (split_ws) g.split = split_view(split_ws, B, D);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(B, 4)), dim3(NT), 0, (hipStream_t)stream, g);
  return emx::check_launch("gather_rows_kernel");
}

int emx_periodic_in(const float* x, int B, int D, int ldx, float periodicity, float* out, int ldo, emx_stream_t stream) {
  EMX_REQUIRE(x && out && B > 0 && D > 0 && ldx >= D && ldo >= 2 * D, "emx_periodic_in: bad arguments");
  EMX_REQUIRE(std::isfinite(periodicity) && periodicity > 0.f, "emx_periodic_in: needs a finite periodicity");
  float ang = periodicity == TWO_PI ? 1.f : TWO_PI / periodicity;
  hipLaunchKernelGGL(periodic_in_kernel, dim3(grid_for((long)B * D, NT)), dim3(NT), 0, (hipStream_t)stream, x, B, D, ldx, ang, out,
                     ldo);
  return emx::check_launch("periodic_in_kernel");
}

int emx_periodic_out(const float* y, int B, int n, int ldy, int off_sin, int off_cos, float periodicity, float* out, int ldo,
                     emx_stream_t stream) {
  EMX_REQUIRE(y && out && B > 0 && n > 0 && ldo >= n, "emx_periodic_out: bad arguments");
  EMX_REQUIRE(off_sin >= 0 && off_cos >= 0 && off_sin + n <= ldy && off_cos + n <= ldy, "emx_periodic_out: column window outside the row");
  EMX_REQUIRE(std::isfinite(periodicity) && periodicity > 0.f, "emx_periodic_out: needs a finite periodicity");
  float sc = periodicity == TWO_PI ? 1.f : periodicity / TWO_PI;
  hipLaunchKernelGGL(periodic_out_kernel, dim3(grid_for((long)B * n, NT)), dim3(NT), 0, (hipStream_t)stream, y, B, n, ldy, off_sin,
                     off_cos, sc, out, ldo);
  return emx::check_launch("periodic_out_kernel");
}

namespace {
// validated description of one row-mean cost; *grid = its workgroups
int fill_cost_job(CostArgs& a, int* grid, const float* y, int ldy, int off_sin, int off_cos, int periodic_output, const float* target, int ldt,
                  int B, int n, float periodicity, int variant, float coef, const float* coef_scale, const float* dpred_add, int row_begin,
                  int row_end, float* loss, float* dY, float* ws, const char* who) {
  EMX_REQUIRE(y && target && ws, "%s: null pointer argument", who);
  EMX_REQUIRE(B > 0 && n > 0 && ldt >= n, "%s: bad shape", who);
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && row_end <= B, "%s: bad row block", who);
  EMX_REQUIRE(variant >= EMX_MEAN_ABS && variant <= EMX_MEAN_NORM, "%s: cost variant %d not available", who, variant);
  // ldy == 0 broadcasts one prediction row to every batch row (forward only: the batch-mean angles of MeanAngles)
  EMX_REQUIRE(ldy > 0 || !dY, "%s: a broadcast prediction (ldy == 0) cannot receive a gradient", who);
  EMX_REQUIRE(off_sin >= 0 && (ldy == 0 || off_sin + n <= ldy) && (!periodic_output || (off_cos >= 0 && (ldy == 0 || off_cos + n <= ldy))),
              "%s: column window outside the row", who);
  EMX_REQUIRE(!periodic_output || std::isfinite(periodicity), "%s: periodic output needs a finite periodicity", who);
  a.y = y;
  a.target = target;
  a.dpred_add = dpred_add;
  a.coef_scale = coef_scale;
  a.loss = loss;
  a.dY = dY;
  a.ws = ws;
  fill_cost_scalars(a, B, n, ldy, ldt, off_sin, off_cos, periodic_output, periodicity, variant, coef);
  a.row_begin = row_begin;
  a.row_end = row_end;
  a.prescale = loss ? 0 : 1;
  // one workgroup per row: long rows, or a few hundred rows of a few hundred columns (a wave per row leaves most CUs idle
  // behind a 256-row batch, with an atan2 per element)
  a.wide = ((n >= 1024 || (n >= 128 && row_end - row_begin <= EMX_REDUCE_WS)) && variant != EMX_MEAN_NORM) ? 1 : 0;
  if (a.wide) {
    // a few hundred long rows: several workgroups per row, as long as every row keeps its own workgroups
    const int rows = row_end - row_begin;
    const int S = std::max(1, std::min(EMX_REDUCE_WS / rows, (n + 2 * NT - 1) / (2 * NT)));
    a.wide = rows <= EMX_REDUCE_WS ? S : 1;
    *grid = std::min(rows, EMX_REDUCE_WS) * a.wide;
  } else {
    *grid = grid_for(row_end - row_begin, 4);
  }
  return EMX_OK;
}
}  // namespace

int emx_periodic_cost_fwd_bwd(const float* y, int ldy, int off_sin, int off_cos, int periodic_output, const float* target, int ldt,
                              int B, int n, float periodicity, int variant, float coef, const float* coef_scale,
                              const float* dpred_add, int row_begin, int row_end, float* loss, float* dY, float* ws,
                              emx_stream_t stream) {
  CostArgs a;
  int grid = 0;
  int rc = fill_cost_job(a, &grid, y, ldy, off_sin, off_cos, periodic_output, target, ldt, B, n, periodicity, variant, coef, coef_scale, dpred_add,
                         row_begin, row_end, loss, dY, ws, "emx_periodic_cost_fwd_bwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(periodic_cost_kernel, dim3(grid), dim3(NT), 0, st, a);
  rc = emx::check_launch("periodic_cost_kernel");
  if (rc || !loss) return rc;
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NT), 0, st, ws, grid, a.gnorm, coef_scale, loss);
  return emx::check_launch("final_sum_kernel");
}

int emx_periodic_cost_batch(const emx_cost_job* jobs, int n_jobs, int B, int row_begin, int row_end, emx_stream_t stream) {
  EMX_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= EMX_MAX_COST_JOBS, "emx_periodic_cost_batch: 1 .. %d jobs", EMX_MAX_COST_JOBS);
  CostBatch b{};
  b.n_jobs = n_jobs;
  b.first[0] = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const emx_cost_job& q = jobs[j];
    int grid = 0;
    int rc = fill_cost_job(b.job[j], &grid, q.y, q.ldy, q.off_sin, q.off_cos, q.periodic_output, q.target, q.ldt, B, q.n, q.periodicity, q.variant,
                           q.coef, q.coef_scale, q.dpred_add, row_begin, row_end, nullptr, q.dY, q.ws, "emx_periodic_cost_batch");
    if (rc) return rc;
    for (int i = 0; i < j; ++i) EMX_REQUIRE(jobs[i].ws != q.ws, "emx_periodic_cost_batch: jobs %d and %d share their scratch", i, j);
    b.first[j + 1] = b.first[j] + grid;
  }
  hipLaunchKernelGGL(periodic_cost_batch_kernel, dim3(b.first[n_jobs]), dim3(NT), 0, (hipStream_t)stream, b);
  return emx::check_launch("periodic_cost_batch_kernel");
}

int emx_center_cost_fwd_bwd(const float* Z, int B, int L, float scale, int row_begin, int row_end, float* loss, float* dZ,
                            int accumulate, emx_stream_t stream) {
  EMX_REQUIRE(Z && loss && dZ && B > 0 && L > 0, "emx_center_cost_fwd_bwd: bad arguments");
  EMX_REQUIRE(0 <= row_begin && row_begin < row_end && row_end <= B, "emx_center_cost_fwd_bwd: bad row block");
  const double bl = (double)B * L;
  hipLaunchKernelGGL(center_cost_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, Z, L, row_begin, row_end, (float)(scale / bl),
                     (float)(2.0 * scale / bl), loss, dZ, accumulate);
  return emx::check_launch("center_cost_kernel");
}

namespace {
constexpr int ADAM_NT = 64;   // one wave per workgroup (see adam_kernel, which has the number built in)
inline int gather_blocks(const emx_next_batch* next) {   // two rows per wave, or one
  if (!next) return 0;
  if (next->adc_head) return 4 * std::min(next->B, 2048) + (next->D + 3) / 4;   // (D: the feature columns whose means are wanted, 0: none)
  return (int)std::min<long>(gather_vec4(next) ? (next->B + 1) / 2 : next->B, 8192);
}

// workgroups (one wave each) of the optimizer part of the launch: float4 lanes over the flat buffer, or -- with a weight
// operand store -- one per 32 x 32 kernel block and per 64 bias elements
inline int adam_grid(const AdamArgs& a, int64_t n) {
  if (a.mirror) return 64 * a.n_planes;   // (<= EMX_REDUCE_WS: 16 planes of 64)
  return grid_for((n + 3) / 4, 64);
}

#define ADAM_LAUNCH_(K, grid_, st_, a_)                                                                                                   \
  hipLaunchKernelGGL(K, dim3(grid_), dim3(ADAM_NT), 0, st_, (a_).adam_blocks, (a_).hstep != nullptr ? 1 : 0, (a_).next_vec4,                    \
                     (a_).next.batch_step, (a_).step, (const char*)(a_).mirror, (a_).next.cursor, (a_).next.idx, (a_).next.B, (int)(grid_),          \
                     a_)
#define ADAM_LAUNCH(grid_, st_, a_)                                                                                                       \
  do {                                                                                                                                    \
    if ((a_).next.adc_head != nullptr) ADAM_LAUNCH_(adam_kernel<true>, grid_, st_, a_);                                                     \
    else ADAM_LAUNCH_(adam_kernel<false>, grid_, st_, a_);                                                                                  \
  } while (0)

}  // namespace

int emx_adam_step(float* w, const float* g, float* m, float* v, const float* l2coef, int64_t n, float lr, float beta1, float beta2,
                  float eps, float clip, const int64_t* step, float* reg_loss, float* ws, const emx_mlp_desc* mirror_desc, float* mirror,
                  const emx_next_batch* next, int bookkeeping_done, emx_stream_t stream) {
  int rcn = check_next(next, "emx_adam_step");
  if (rcn) return rcn;
  EMX_REQUIRE(!bookkeeping_done || !reg_loss, "emx_adam_step: with bookkeeping_done the regulariser shares belong to the next step (reg_loss must be NULL)");
  EMX_REQUIRE(w && g && m && v && step && ws, "emx_adam_step: null pointer argument");
  EMX_REQUIRE(!mirror || (mirror_desc && mirror_desc->n_layers >= 1 && mirror_desc->n_layers <= EMX_MAX_LAYERS), "emx_adam_step: mirror without a valid descriptor");
  EMX_REQUIRE(n > 0, "emx_adam_step: empty parameter buffer");
  AdamArgs a = make_adam(w, g, nullptr, 0, 0, m, v, l2coef, n, lr, beta1, beta2, eps, clip, step, ws, mirror_desc, mirror, next);
  const int grid = adam_grid(a, n);
  a.adam_blocks = grid;
  a.late = bookkeeping_done ? 1 : 0;
  if (a.late) a.next.batch_step = 0;
  hipStream_t st = (hipStream_t)stream;
  ADAM_LAUNCH(grid + gather_blocks(next), st, a);
  int rc = emx::check_launch("adam_kernel");
  if (rc) return rc;
  if (reg_loss) {
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NT), 0, st, ws, grid, 1.f, (const float*)nullptr, reg_loss);
    rc = emx::check_launch("final_sum_kernel");
  }
  return rc;
}

int emx_adam_step_history(float* w, const float* g, float* m, float* v, const float* l2coef, int64_t n, float lr, float beta1, float beta2,
                          float eps, float clip, const int64_t* step, float* ws, const emx_mlp_desc* mirror_desc, float* mirror,
                          const emx_next_batch* next, const float* terms, int n_terms, float* history, float* term_history,
                          int64_t capacity, const int64_t* hstep, emx_stream_t stream) {
  int rcn = check_next(next, "emx_adam_step_history");
  if (rcn) return rcn;
  EMX_REQUIRE(w && g && m && v && step && ws && terms && hstep, "emx_adam_step_history: null pointer argument");
  EMX_REQUIRE(!mirror || (mirror_desc && mirror_desc->n_layers >= 1 && mirror_desc->n_layers <= EMX_MAX_LAYERS), "emx_adam_step_history: mirror without a valid descriptor");
  EMX_REQUIRE(n > 0 && n_terms > 0 && n_terms <= 64 && capacity >= 0, "emx_adam_step_history: bad sizes");
  AdamArgs a = make_adam(w, g, nullptr, 0, 0, m, v, l2coef, n, lr, beta1, beta2, eps, clip, step, ws, mirror_desc, mirror, next);
  a.adam_blocks = adam_grid(a, n);
  a.late = 1;
  a.next.batch_step = 0;
  a.row_terms = terms;
  a.row_n = n_terms;
  a.history = history;
  a.term_history = term_history;
  a.capacity = (long)capacity;
  a.hstep = hstep;
  ADAM_LAUNCH(a.adam_blocks + gather_blocks(next) + 1, (hipStream_t)stream, a);
  return emx::check_launch("adam_kernel");
}

int emx_adam_step_slices(float* w, const float* g_partials, int n_slices, int64_t partial_stride, float* m, float* v, const float* l2coef,
                         int64_t n, float lr, float beta1, float beta2, float eps, float clip, const int64_t* step, float* ws,
                         const emx_mlp_desc* mirror_desc, float* mirror, const emx_next_batch* next, int bookkeeping_done,
                         emx_stream_t stream) {
  int rcn = check_next(next, "emx_adam_step_slices");
  if (rcn) return rcn;
  EMX_REQUIRE(w && g_partials && m && v && step && ws, "emx_adam_step_slices: null pointer argument");
  EMX_REQUIRE(!mirror || (mirror_desc && mirror_desc->n_layers >= 1 && mirror_desc->n_layers <= EMX_MAX_LAYERS), "emx_adam_step_slices: mirror without a valid descriptor");
  EMX_REQUIRE(n > 0 && n_slices > 0 && partial_stride >= n, "emx_adam_step_slices: bad slice layout");
  AdamArgs a = make_adam(w, nullptr, g_partials, n_slices, partial_stride, m, v, l2coef, n, lr, beta1, beta2, eps, clip, step, ws, mirror_desc,
                         mirror, next);
  a.adam_blocks = adam_grid(a, n);
  a.late = bookkeeping_done ? 1 : 0;
  if (a.late) a.next.batch_step = 0;
  ADAM_LAUNCH(a.adam_blocks + gather_blocks(next), (hipStream_t)stream, a);
  return emx::check_launch("adam_kernel");
}

int emx_l2_partials(const float* w, const float* l2coef, int64_t n, float* ws, emx_stream_t stream) {
  EMX_REQUIRE(w && ws && n > 0, "emx_l2_partials: bad arguments");
  hipLaunchKernelGGL(l2_partials_kernel, dim3(grid_for((n + 3) / 4, ADAM_NT)), dim3(ADAM_NT), 0, (hipStream_t)stream, w, l2coef, (long)n, ws);
  return emx::check_launch("l2_partials_kernel");
}

int emx_adam_tick(float* w, const float* g, const float* g_partials, int n_slices, int64_t partial_stride, float* m, float* v,
                  const float* l2coef, int64_t n, float lr, float beta1, float beta2, float eps, float clip, int64_t* opt_step,
                  float* terms, int n_terms, int reg_slot, float* history, float* term_history, int64_t capacity, int64_t* hstep,
                  int64_t* cursor, float* ws, const emx_mlp_desc* mirror_desc, float* mirror, emx_stream_t stream) {
  EMX_REQUIRE(w && m && v && opt_step && terms && hstep && ws, "emx_adam_tick: null pointer argument");
  EMX_REQUIRE(!mirror || (mirror_desc && mirror_desc->n_layers >= 1 && mirror_desc->n_layers <= EMX_MAX_LAYERS), "emx_adam_tick: mirror without a valid descriptor");
  EMX_REQUIRE((g != nullptr) != (g_partials != nullptr), "emx_adam_tick: pass either the gradient or its slice partials");
  EMX_REQUIRE(!g_partials || (n_slices > 0 && partial_stride >= n), "emx_adam_tick: bad slice layout");
  EMX_REQUIRE(n > 0 && n_terms > 0 && n_terms <= 64 && reg_slot >= 0 && reg_slot < n_terms, "emx_adam_tick: bad sizes");
  AdamArgs a = make_adam(w, g, g_partials, n_slices, partial_stride, m, v, l2coef, n, lr, beta1, beta2, eps, clip, opt_step, ws, mirror_desc, mirror);
  a.adam_blocks = adam_grid(a, n);
  // Three plain launches on the stream.  (An in-kernel hand-off -- last workgroup finishes -- was measured slower: its
  // agent-scope release has to write back the megabytes of L2 lines the weight-gradient kernel has just dirtied.)
  const int grid = a.adam_blocks;
  hipStream_t st = (hipStream_t)stream;
  ADAM_LAUNCH(grid, st, a);
  int rc = emx::check_launch("adam_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NT), 0, st, ws, grid, 1.f, (const float*)nullptr, terms + reg_slot);
  rc = emx::check_launch("final_sum_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, st, terms, n_terms, history, term_history, (long)capacity, hstep, cursor,
                     opt_step);
  return emx::check_launch("step_tick_kernel");
}

int emx_finish_step(const emx_sum_job* jobs, int n_jobs, float* terms, int n_terms, float* history, float* term_history,
                    int64_t capacity, int64_t* hstep, int64_t* cursor, int64_t* counter2, emx_stream_t stream) {
  FinishArgs a;
  int rc = fill_finish_args(a, jobs, n_jobs, terms, n_terms, history, term_history, capacity, hstep, cursor, counter2, "emx_finish_step");
  if (rc) return rc;
  hipLaunchKernelGGL(finish_step_kernel, dim3(1), dim3(FIN_NT), 0, (hipStream_t)stream, a);
  return emx::check_launch("finish_step_kernel");
}

int emx_step_tick(const float* terms, int n_terms, float* history, float* term_history, int64_t capacity, int64_t* step,
                  int64_t* cursor, int64_t* counter2, emx_stream_t stream) {
  EMX_REQUIRE(terms && step && n_terms > 0 && n_terms <= 64, "emx_step_tick: bad arguments");
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, n_terms, history, term_history, (long)capacity,
                     step, cursor, counter2);
  return emx::check_launch("step_tick_kernel");
}

}  // extern "C"

#ifdef EMX_CHAIN_TIMING
extern "C" int emx_debug_read_fin(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg_fin), sizeof(emx_dbg_fin)); }
#endif
