// Kernels whose only effect is to walk 0 .. 56 KB of straight-line code on every CU (and so to evict that much of the CUs' instruction
// caches): tools/time_icache.py launches them between launches of the fused dense-layer kernel
// to separate the instruction-cache part of "first touch" from the data part.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/icache_probe.hip -o tools/libicache_probe.so
#include <hip/hip_runtime.h>

#define CODE_KERNEL(NAME, REPT)                                                              \
  __global__ __launch_bounds__(64) void NAME(int* out) {                                      \
    int x = threadIdx.x;                                                                       \
    asm volatile(".rept " #REPT "\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x)); /* 4 bytes each */ \
    if (x == -1) out[0] = x;                                                                   \
  }
CODE_KERNEL(code_0k, 16)
CODE_KERNEL(code_8k, 2000)
CODE_KERNEL(code_16k, 4000)
CODE_KERNEL(code_24k, 6000)
CODE_KERNEL(code_32k, 8000)
CODE_KERNEL(code_40k, 10000)
CODE_KERNEL(code_48k, 12000)
CODE_KERNEL(code_56k, 14000)
static int* buf = nullptr;
// kb: 0, 8, 16, .. 56 KB of straight-line code walked by `wgs` single-wave workgroups
extern "C" int probe_launch(int kb, int wgs, void* stream) {
  if (!buf) (void)hipMalloc(&buf, 64);
  hipStream_t st = (hipStream_t)stream;
  switch (kb) {
    case 0: hipLaunchKernelGGL(code_0k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 8: hipLaunchKernelGGL(code_8k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 16: hipLaunchKernelGGL(code_16k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 24: hipLaunchKernelGGL(code_24k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 32: hipLaunchKernelGGL(code_32k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 40: hipLaunchKernelGGL(code_40k, dim3(wgs), dim3(64), 0, st, buf); break;
    case 48: hipLaunchKernelGGL(code_48k, dim3(wgs), dim3(64), 0, st, buf); break;
    default: hipLaunchKernelGGL(code_56k, dim3(wgs), dim3(64), 0, st, buf); break;
  }
  return (int)hipGetLastError();
}
