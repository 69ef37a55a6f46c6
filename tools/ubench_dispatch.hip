// What does it cost to DISPATCH a grid?  Empty kernels (every wave leaves at once) of the all-pairs tile kernel's shapes -- 2080 workgroups
// of 256 threads with 19 KB of LDS and 64 registers -- and of coarser ones, 50 launches per hipGraph replay (no host in the loop).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o tools/ubench_dispatch ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT, int LDSF>
__global__ __launch_bounds__(NT) void empty(float* out, int never) {
  __shared__ float lds[LDSF > 0 ? LDSF : 1];
  if (never) {
    lds[threadIdx.x] = (float)never;
    __syncthreads();
    out[blockIdx.x * NT + threadIdx.x] = lds[(threadIdx.x * 7) % (LDSF > 0 ? LDSF : 1)];
  }
}

template <int NT, int LDSF>
float time_grid(int grid, float* out, hipStream_t st) {
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((empty<NT, LDSF>), dim3(grid), dim3(NT), 0, st, out, 0);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  return best * 1e3f / 50.f;
}

int main() {
  float* out;
  hipMalloc(&out, 4200 * 1024 * 4);
  hipStream_t st;
  hipStreamCreate(&st);
  printf("empty kernels, us per launch inside a graph of 50\n");
  printf("grid    1 x  64, no LDS        %6.2f\n", time_grid<64, 0>(1, out, st));
  printf("grid  256 x 256, no LDS        %6.2f\n", time_grid<256, 0>(256, out, st));
  printf("grid  528 x 256, 19 KB LDS     %6.2f\n", time_grid<256, 4744>(528, out, st));
  printf("grid 1176 x 256, 19 KB LDS     %6.2f\n", time_grid<256, 4744>(1176, out, st));
  printf("grid 2080 x 256, no LDS        %6.2f\n", time_grid<256, 0>(2080, out, st));
  printf("grid 2080 x 256, 19 KB LDS     %6.2f\n", time_grid<256, 4744>(2080, out, st));
  printf("grid 4160 x 256, 19 KB LDS     %6.2f\n", time_grid<256, 4744>(4160, out, st));
  printf("grid 1040 x 512, 38 KB LDS     %6.2f\n", time_grid<512, 9488>(1040, out, st));
  printf("grid  520 x1024, 38 KB LDS     %6.2f\n", time_grid<1024, 9488>(520, out, st));
  printf("grid  256 x 512, 38 KB LDS     %6.2f\n", time_grid<512, 9488>(256, out, st));
  printf("grid  512 x 512, 38 KB LDS     %6.2f\n", time_grid<512, 9488>(512, out, st));
  printf("grid 8320 x  64, 4.7 KB LDS    %6.2f\n", time_grid<64, 1186>(8320, out, st));
  return 0;
}
