// Does a line written by one launch survive in the writer XCD's L2 for the next launch?  Kernel A: workgroup i writes chunk i
// (64 KB).  Kernel B: workgroup i reads chunk (i + shift) % n -- shift 0: the reader runs on the XCD that wrote the chunk
// (workgroups are dealt to the 8 XCDs round-robin), shift 1: on the next XCD, shift 8: the same XCD, another CU.  B is
// latency-bound on purpose: one dependent 16-byte load per lane after the other (a chain of `hops` trips), the time of B per
// hop is the latency of a trip.  Also: B after B (nothing rewritten in between: clean lines).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_xcd_affinity.hip -o tools/ubench_xcd_affinity ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int CHUNK = 65536;   // bytes per workgroup

__global__ __launch_bounds__(256) void write_kernel(unsigned* buf, unsigned seed) {
  unsigned* p = buf + (long)blockIdx.x * (CHUNK / 4);
  // a permutation cycle inside the chunk: element e holds the index of the next 16-byte piece to visit
  for (int e = threadIdx.x; e < CHUNK / 16; e += 256) {
    const unsigned next = (e * 97u + 31u + seed) % (CHUNK / 16);
    p[e * 4] = next;
    p[e * 4 + 1] = p[e * 4 + 2] = p[e * 4 + 3] = seed;
  }
}

__global__ __launch_bounds__(64) void chase_kernel(const unsigned* buf, int n, int shift, int hops, unsigned* out) {
  const unsigned* p = buf + (long)((blockIdx.x + shift) % n) * (CHUNK / 4);
  unsigned e = threadIdx.x;
  for (int h = 0; h < hops; ++h) e = __builtin_nontemporal_load(p + 4 * e) % (CHUNK / 16) ^ 0;   // (plain dependent loads)
  out[blockIdx.x * 64 + threadIdx.x] = e;
}

__global__ __launch_bounds__(64) void chase_plain_kernel(const unsigned* buf, int n, int shift, int hops, unsigned* out) {
  const unsigned* p = buf + (long)((blockIdx.x + shift) % n) * (CHUNK / 4);
  unsigned e = threadIdx.x;
  for (int h = 0; h < hops; ++h) e = p[4 * e] % (CHUNK / 16);
  out[blockIdx.x * 64 + threadIdx.x] = e;
}

int main() {
  const int n = 256, hops = 64, reps = 50;
  unsigned *buf, *out;
  hipMalloc(&buf, (size_t)n * CHUNK);
  hipMalloc(&out, (size_t)n * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipStream_t st;
  hipStreamCreate(&st);
  for (int mode = 0; mode < 2; ++mode) {          // 0: B after A (fresh, dirty lines), 1: B after B (clean lines)
    for (int shift : {0, 8, 1, 3}) {
      float tb = 0.f;
      for (int r = 0; r < reps + 5; ++r) {
        if (mode == 0) hipLaunchKernelGGL(write_kernel, dim3(n), dim3(256), 0, st, buf, (unsigned)r);
        else hipLaunchKernelGGL(chase_plain_kernel, dim3(n), dim3(64), 0, st, buf, n, shift, hops, out);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(chase_plain_kernel, dim3(n), dim3(64), 0, st, buf, n, shift, hops, out);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 5) tb += ms;
      }
      printf("%s, workgroup b reads the chunk workgroup b + %d wrote: %.2f us per launch of %d dependent trips (%.0f ns per trip incl. launch)\n",
             mode == 0 ? "after the WRITER launch" : "after a READER launch (same shift)", shift, tb / reps * 1e3, hops, tb / reps * 1e6 / hops);
    }
  }
  return 0;
}
