// VALU issue-rate microbenchmark for gfx950: decides how the sketch-map inner loop is written.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { FMA, PKFMA, SUB, SUBABS, MINF, PKADD, PKMUL, EXP, LOG, RSQ, RCP, SQRT, CVTI, SUBU, RNDNE, MIX_EUC, MIX_EUC_PK, MIX_PER, MIX_INT, SUBCLAMP, MIX_FLOAT3, MIX_RELU, MIX_RELU_PK, MIX_FLOAT3_PK, NOPS };
const char* NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_sub_f32", "v_sub_f32 |abs|(vop3)", "v_min_f32", "v_pk_add_f32", "v_pk_mul_f32",
  "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "v_cvt_f32_i32", "v_sub_u32", "v_rndne_f32",
  "mix: sub+fma", "mix: pk_add+pk_fma", "mix: sub,sub|abs|,min,fma", "mix: sub_u32,cvt,fma",
  "v_sub_f32 |abs| clamp (vop3)", "mix: sub,sub|abs|,sub|abs|,fma", "mix: sub,sub|abs|clamp,add", "mix: 2x(sub,sub|abs|clamp),pk_add",
  "mix: 2x(sub,sub|abs|,sub|abs|),pk_fma"};
// elements per "instruction group" for reporting
template <int OP>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float x, float y) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f2 px = {x, y}, py = {y, x};
  int i0 = threadIdx.x, i1 = i0 * 3, i2 = i0 * 5, i3 = i0 * 7, i4 = i0 * 11, i5 = i0 * 13, i6 = i0 * 17, i7 = i0 * 19;
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == FMA) {
#define X(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##n) : "v"(x), "v"(y));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == PKFMA) {
#define X(n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p##n) : "v"(px), "v"(py));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == SUB) {
#define X(n) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##n) : "v"(x));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == SUBABS) {
#define X(n) asm volatile("v_sub_f32 %0, %1, |%0|" : "+v"(a##n) : "v"(x));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == MINF) {
#define X(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##n) : "v"(x));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == PKADD) {
#define X(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "v"(px));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == PKMUL) {
#define X(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(px));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == EXP) {
#define X(n) asm volatile("v_exp_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == LOG) {
#define X(n) asm volatile("v_log_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == RSQ) {
#define X(n) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == RCP) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == SQRT) {
#define X(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == CVTI) {
#define X(n) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a##n) : "v"(i##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == SUBU) {
#define X(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(i##n) : "v"(i0));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == RNDNE) {
#define X(n) asm volatile("v_rndne_f32 %0, %0" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == MIX_EUC) {  // 8 elements: sub + fma each = 16 instr
      float t;
#define X(n) asm volatile("v_sub_f32 %1, %2, %3\n v_fma_f32 %0, %1, %1, %0" : "+v"(a##n), "=&v"(t) : "v"(x), "v"(y));
      REP8(X)
#undef X
    } else if constexpr (OP == MIX_EUC_PK) {  // 16 elements: 8 pk_add + 8 pk_fma = 16 instr
      f2 t;
#define X(n) asm volatile("v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p##n), "=&v"(t) : "v"(px), "v"(py));
      REP8(X)
#undef X
    } else if constexpr (OP == MIX_PER) {  // 4 elements: 16 instr
      float t, u;
#define X(n) asm volatile("v_sub_f32 %1, %3, %4\n v_sub_f32 %2, %5, |%1|\n v_min_f32 %1, |%1|, %2\n v_fma_f32 %0, %1, %1, %0" : "+v"(a##n), "=&v"(t), "=&v"(u) : "v"(x), "v"(y), "v"(6.2831853f));
      X(0) X(1) X(2) X(3)
#undef X
    } else if constexpr (OP == SUBCLAMP) {
#define X(n) asm volatile("v_sub_f32_e64 %0, |%0|, 0.5 clamp" : "+v"(a##n));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == MIX_FLOAT3) {  // the float form of the periodic tiles: 4 elements, 16 instr
      float t, u;
#define X(n) asm volatile("v_sub_f32 %1, %3, %4\n v_sub_f32_e64 %2, |%1|, %5\n v_sub_f32_e64 %1, %5, |%2|\n v_fma_f32 %0, %1, %1, %0" : "+v"(a##n), "=&v"(t), "=&v"(u) : "v"(x), "v"(y), "v"(3.14159265f));
      X(0) X(1) X(2) X(3)
#undef X
    } else if constexpr (OP == MIX_RELU) {  // Gram + correction form: 5.33 elements, 16 instr
      float t, u;
#define X(n) asm volatile("v_sub_f32 %1, %3, %4\n v_sub_f32_e64 %2, |%1|, 0.5 clamp\n v_add_f32 %0, %0, %2" : "+v"(a##n), "=&v"(t), "=&v"(u) : "v"(x), "v"(y));
      X(0) X(1) X(2) X(3) X(4)
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(y));
#undef X
    } else if constexpr (OP == MIX_RELU_PK) {  // as the compiler writes it: 2 elements = sub, sub, clamp, clamp, pk_add: 15 instr + 1
      f2 t; float u, w;
#define X(n) asm volatile("v_sub_f32 %2, %4, %5\n v_sub_f32 %3, %5, %4\n v_sub_f32_e64 %2, |%2|, 0.5 clamp\n v_sub_f32_e64 %3, |%3|, 0.5 clamp\n v_pk_add_f32 %0, %0, %1" : "+v"(p##n), "=&v"(t), "=&v"(u), "=&v"(w) : "v"(x), "v"(y));
      X(0) X(1) X(2)
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
#undef X
    } else if constexpr (OP == MIX_FLOAT3_PK) {  // float form as compiled: 2 elements = 2 sub, 4 sub|abs|, pk_fma: 14 instr + 2
      f2 t; float u, w;
#define X(n) asm volatile("v_sub_f32 %2, %4, %5\n v_sub_f32 %3, %5, %4\n v_sub_f32_e64 %2, |%2|, %6\n v_sub_f32_e64 %3, |%3|, %6\n v_sub_f32_e64 %2, %6, |%2|\n v_sub_f32_e64 %3, %6, |%3|\n v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p##n), "=&v"(t), "=&v"(u), "=&v"(w) : "v"(x), "v"(y), "v"(3.14159265f));
      X(0) X(1)
      asm volatile("v_sub_f32 %0, %1, %2\n v_sub_f32 %0, %2, %1" : "=v"(u) : "v"(x), "v"(y));
#undef X
    } else if constexpr (OP == MIX_INT) {  // ~5.33 elements: 16 instr (3 per element)
      float t; int u;
#define X(n) asm volatile("v_sub_u32 %2, %3, %4\n v_cvt_f32_i32 %1, %2\n v_fma_f32 %0, %1, %1, %0" : "+v"(a##n), "=&v"(t), "=&v"(u) : "v"(i##n), "v"(i0));
      X(0) X(1) X(2) X(3) X(4)
      asm volatile("v_sub_u32 %0, %1, %2" : "=v"(u) : "v"(i1), "v"(i0));
#undef X
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x +
            (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
  if (r == 12345.678f) out[0] = r;
}

template <int OP>
double run(int blocks, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.9999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

template <int OP>
void report(float* out) {
  const int iters = 20000;
  for (int wpb : {1, 2, 4, 8}) {  // blocks per CU (each block = 4 waves = 1 per SIMD)
    int blocks = 256 * wpb;
    double s = run<OP>(blocks, iters, out);
    double winstr = (double)blocks * 4 * iters * 16;  // wave-instructions
    double per_simd_cycle = winstr / (256.0 * 4) / (s * 2.4e9);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f wave-instr/clk/SIMD (@2.4GHz)  %.2f T lane-ops/s\n", NAMES[OP], wpb, s * 1e3, per_simd_cycle,
           winstr * 64 / s * 1e-12);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  report<FMA>(out); report<PKFMA>(out); report<SUB>(out); report<SUBABS>(out); report<MINF>(out); report<PKADD>(out); report<PKMUL>(out);
  report<EXP>(out); report<LOG>(out); report<RSQ>(out); report<RCP>(out); report<SQRT>(out); report<CVTI>(out); report<SUBU>(out); report<RNDNE>(out);
  report<MIX_EUC>(out); report<MIX_EUC_PK>(out); report<MIX_PER>(out); report<MIX_INT>(out);
  report<SUBCLAMP>(out); report<MIX_FLOAT3>(out); report<MIX_RELU>(out); report<MIX_RELU_PK>(out); report<MIX_FLOAT3_PK>(out);
  return 0;
}
