// A kernel whose only effect is to walk ~56 KB of straight-line code on every CU (and so to evict the CUs' instruction caches of
// whatever ran before), and one that does nothing: tools/time_icache.py launches them between launches of the fused dense-layer kernel
// to separate the instruction-cache part of "first touch" from the data part.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/icache_probe.hip -o tools/libicache_probe.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void bigcode_kernel(int* out) {
  int x = threadIdx.x;
  asm volatile(".rept 14000\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x));   // 14000 x 4 bytes
  if (x == -1) out[0] = x;
}
__global__ __launch_bounds__(64) void tiny_kernel(int* out) {
  int x = threadIdx.x;
  asm volatile(".rept 16\n v_add_u32 %0, %0, 1\n .endr" : "+v"(x));
  if (x == -1) out[0] = x;
}
static int* buf = nullptr;
extern "C" int probe_launch(int big, int wgs, void* stream) {
  if (!buf) hipMalloc(&buf, 64);
  if (big) hipLaunchKernelGGL(bigcode_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, buf);
  else hipLaunchKernelGGL(tiny_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, buf);
  return (int)hipGetLastError();
}
