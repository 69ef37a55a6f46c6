// Weight-stream microbenchmark for the row-block chain kernel (gfx950): every workgroup streams the SAME `units` x 64 KB
// from L2 into registers, a 1-KB piece per wave-instruction, `DEPTH` units in flight -- what does a CU get, and how does
// it depend on the order in which the CUs of an XCD walk the pieces?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: every workgroup the same order; 1: the wave's 16 pieces rotated by the workgroup's index inside its XCD;
// 2: rotation + waves of a workgroup take each other's pieces in XCD-dependent order (piece p of the unit: (p + 7 x) % 64)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ w, int units, float* out, int active) {
  if ((int)blockIdx.x >= active) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = blockIdx.x >> 3;   // index of the workgroup among those of its XCD
  f4 ring[DEPTH][16];
  f4 acc = {0, 0, 0, 0};
  auto piece_off = [&](int u, int i) -> long {
    int p;
    if (MODE == 0) p = wave * 16 + i;
    else if (MODE == 1) p = wave * 16 + ((i + x) & 15);
    else p = (wave * 16 + i + 7 * x) & 63;
    return (long)u * 65536 + p * 1024 + lane * 16;
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[d][i] = *reinterpret_cast<const f4*>(w + piece_off(d, i));
  auto step = [&](auto RC, int u) {
    constexpr int r = decltype(RC)::value;
    constexpr int rx = (r + 1) % DEPTH;
    const int un = min(u + DEPTH - 1, units - 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc += ring[rx][i];
      ring[r][i] = *reinterpret_cast<const f4*>(w + piece_off(un, i));
    }
  };
  for (int u = 0;; u += DEPTH) {
    step(std::integral_constant<int, DEPTH - 1>{}, u);
    if (u + 1 >= units) break;
    step(std::integral_constant<int, 0>{}, u + 1);
    if (u + 2 >= units) break;
    if constexpr (DEPTH >= 3) {
      step(std::integral_constant<int, 1>{}, u + 2);
      if (u + 3 >= units) break;
    }
    if constexpr (DEPTH >= 4) {
      step(std::integral_constant<int, 2>{}, u + 3);
      if (u + 4 >= units) break;
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int MODE, int DEPTH>
void run(const char* w, int units, float* out, int active, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, w, units, out, active);
  hipDeviceSynchronize();
  const int reps = 50;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, w, units, out, active);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s depth %d  active CUs %3d  units %2d : %7.2f us per launch, %6.3f us per unit, %6.1f GB/s per CU, %5.2f TB/s chip\n", name, DEPTH, active,
         units, us, us / units, 65536.0 * units / us * 1e-3, 65536.0 * units * active / us * 1e-6);
}

int main() {
  const int units = 10;
  char* w;
  float* out;
  hipMalloc(&w, 65536 * 64);
  hipMalloc(&out, 64);
  hipMemset(w, 0, 65536 * 64);
  for (int active : {256, 128, 64, 32, 8}) {
    run<0, 3>(w, units, out, active, "same order");
    run<1, 3>(w, units, out, active, "rotated inside the wave");
    run<2, 3>(w, units, out, active, "rotated over the unit");
  }
  run<0, 2>(w, units, out, 256, "same order");
  run<0, 4>(w, units, out, 256, "same order");
  run<2, 4>(w, units, out, 256, "rotated over the unit");
  run<0, 3>(w, 40, out, 256, "same order");
  run<2, 3>(w, 40, out, 256, "rotated over the unit");
  return 0;
}
