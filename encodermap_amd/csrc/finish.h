// The step's scalar epilogue (emx_finish_step): shared by its stand-alone kernel (elementwise.hip) and the extra workgroup of
// the weight-gradient kernel (mlp.hip).
#pragma once
#include "emx_common.h"

namespace emx {

constexpr int FIN_NT = 256;

#ifndef FSTAMP   // in-kernel timeline hooks, defined by elementwise.hip in EMX_CHAIN_TIMING builds (tools/abl_finish.py)
#define FSTAMP(k)
#endif

struct FinishArgs {
  emx_sum_job jobs[EMX_MAX_SUM_JOBS];
  int n_jobs;
  float* terms;
  int n_terms;
  float* history;
  float* term_history;
  long capacity;
  int64_t* hstep;
  int64_t* cursor;
  int64_t* counter2;
  const float* terms_base;   // 16-byte aligned, >= 16 readable bytes
};

// One workgroup, one memory round trip for all the step's scalars: every job's first 8192 elements are requested (16-byte
// loads, four jobs at a time) before anything is added, one block-wide reduction serves all jobs, and the bookkeeping of
// step_tick_kernel works from values fetched at kernel entry.  A single workgroup is bound by round trips (~2.7 us each
// here: freshly written lines of other XCDs, cold TLB) and by its own instruction count, not by bandwidth; job after job,
// each with its own reduction and a serial tail, was 9 us.  Fixed order throughout.
constexpr int FQ = 8;   // 16-byte loads per thread and job in the common round (stand-alone launch: everything in flight at once)
constexpr int FQ_EMBEDDED = 2;   // ... when it rides along with the weight gradients: its latency hides behind them, its
                                 // registers count against every workgroup of that launch (4 x FQ float4 per thread)

// the body of finish_step_kernel, for one workgroup of FIN_NT threads (also run by one extra workgroup of the
// weight-gradient kernel, so that a single-GPU step needs no launch of its own for it)
template <int FQ>
__device__ __forceinline__ void finish_step_body(const FinishArgs& a, float (&red)[EMX_MAX_SUM_JOBS][FIN_NT / 64], float (&tl)[64],
                                                 const int shift_job = -1, const long shift = 0) {   // (job shift_job reads src + shift floats)
  constexpr int NT = FIN_NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  FSTAMP(0);
  // scalars of the bookkeeping, requested up front
  long hs = 0, cur = 0, cnt = 0;
  if (tid == 0) {
    if (a.hstep) hs = *a.hstep;
    if (a.cursor) cur = *a.cursor;
    if (a.counter2) cnt = *a.counter2;
  }
  if (tid < a.n_terms) tl[tid] = a.terms[tid];
  float dsc[EMX_MAX_SUM_JOBS];
#pragma unroll
  for (int j = 0; j < EMX_MAX_SUM_JOBS; ++j) dsc[j] = (j < a.n_jobs && a.jobs[j].dev_scale && tid == 0) ? *a.jobs[j].dev_scale : 1.f;

  float part[EMX_MAX_SUM_JOBS];
#pragma unroll
  for (int g0 = 0; g0 < EMX_MAX_SUM_JOBS; g0 += 4) {
    if (g0 >= a.n_jobs) {   // uniform
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) part[g0 + jj] = 0.f;
      continue;
    }
    float4 v[4][FQ];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = g0 + jj;
      // absent and unaligned jobs re-read the first 16 bytes of `terms` with weight 0: no branch around any load
      const float* const srcj = a.jobs[j < a.n_jobs ? j : 0].src + (j == shift_job ? shift : 0);
      const bool vec = j < a.n_jobs && (reinterpret_cast<uintptr_t>(srcj) & 15) == 0;
      const float4* __restrict__ src4 = reinterpret_cast<const float4*>(vec ? srcj : a.terms_base);
      const long n4 = vec ? a.jobs[j].n >> 2 : 0;
#pragma unroll
      for (int u = 0; u < FQ; ++u) v[jj][u] = src4[min((long)tid + u * NT, max(n4, 1L) - 1)];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = g0 + jj;
      const bool live = j < a.n_jobs;
      emx_sum_job job = a.jobs[live ? j : 0];
      if (j == shift_job) job.src += shift;
      const bool vec = live && (reinterpret_cast<uintptr_t>(job.src) & 15) == 0;
      const long n = live ? job.n : 0, n4 = vec ? n >> 2 : 0;
      const bool sq = job.square != 0;
      float acc[FQ];
#pragma unroll
      for (int u = 0; u < FQ; ++u) {
        const float4 x = v[jj][u];
        const float wq = (long)tid + u * NT < n4 ? 1.f : 0.f;
        acc[u] = (sq ? (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w) : (x.x + x.y) + (x.z + x.w)) * wq;
      }
      float p;
      if constexpr (FQ == 8) {
        p = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
      } else {
        p = 0.f;
#pragma unroll
        for (int u = 0; u < FQ; ++u) p += acc[u];
      }
      for (long i = (long)tid + FQ * NT; i < n4; i += NT) {   // long jobs: the rest, strided
        const float4 x = reinterpret_cast<const float4*>(job.src)[i];
        p += sq ? (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w) : (x.x + x.y) + (x.z + x.w);
      }
      for (long i = 4 * n4 + tid; i < n; i += NT) {             // tail (or an unaligned job)
        const float x = job.src[i];
        p += sq ? x * x : x;
      }
      part[j] = wave_sum(p);
    }
  }
  FSTAMP(2);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < EMX_MAX_SUM_JOBS; ++j) red[j][wave] = part[j];
  }
  __syncthreads();
  FSTAMP(3);
  if (tid == 0) {
    for (int j = 0; j < a.n_jobs; ++j) {
      float tot = 0.f;
      for (int w = 0; w < NT / 64; ++w) tot += red[j][w];
      tot *= a.jobs[j].scale * dsc[j];
      a.terms[a.jobs[j].slot] = tot;
      tl[a.jobs[j].slot] = tot;
    }
    FSTAMP(4);
    if (a.hstep) {
      float tot = 0.f;
      for (int i = 0; i < a.n_terms; ++i) tot += tl[i];
      if (hs >= 0 && hs < a.capacity) {
        if (a.history) a.history[hs] = tot;
        if (a.term_history)
          for (int i = 0; i < a.n_terms; ++i) a.term_history[hs * a.n_terms + i] = tl[i];
      }
      *a.hstep = hs + 1;
      if (a.cursor) *a.cursor = cur + 1;
      if (a.counter2) *a.counter2 = cnt + 1;
    }
    FSTAMP(5);
  }
}


// host: validated copy of the caller's description
inline int fill_finish_args(FinishArgs& a, const emx_sum_job* jobs, int n_jobs, float* terms, int n_terms, float* history, float* term_history,
                            int64_t capacity, int64_t* hstep, int64_t* cursor, int64_t* counter2, const char* who) {
  EMX_REQUIRE(terms && n_terms > 0 && n_terms <= 64 && n_jobs >= 0 && n_jobs <= EMX_MAX_SUM_JOBS && (n_jobs == 0 || jobs), "%s: bad arguments", who);
  a.jobs[0] = emx_sum_job{terms, 0, 1.f, nullptr, 0, 0};
  EMX_REQUIRE((reinterpret_cast<uintptr_t>(terms) & 15) == 0 && n_terms >= 4, "%s: terms must be 16-byte aligned, >= 4 entries", who);
  a.terms_base = terms;
  for (int j = 0; j < n_jobs; ++j) {
    EMX_REQUIRE((jobs[j].src || jobs[j].n == 0) && jobs[j].n >= 0 && jobs[j].slot >= 0 && jobs[j].slot < n_terms, "%s: bad job %d", who, j);
    a.jobs[j] = jobs[j];
    if (jobs[j].n == 0) a.jobs[j].src = terms;
  }
  a.n_jobs = n_jobs;
  a.terms = terms;
  a.n_terms = n_terms;
  a.history = history;
  a.term_history = term_history;
  a.capacity = (long)capacity;
  a.hstep = hstep;
  a.cursor = cursor;
  a.counter2 = counter2;
  return EMX_OK;
}

}  // namespace emx
