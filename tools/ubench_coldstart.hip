// What do the first memory accesses of a launch cost?  One wave per workgroup stamps (s_memtime) around: a kernel-argument field
// beyond the preloaded registers (scalar load), a field in another 64-byte line of the argument segment, the same line again, a vector
// load of a line of device memory the previous launch wrote, a line next to it, a dependent load through a pointer table, and a
// stretch of straight-line code -- in the first launch of a graph of launches separated by a cache-trashing kernel, and in the next
// launch right behind it.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coldstart.hip -o tools/ubench_coldstart ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big {
  float* out;
  unsigned long long* stamps;
  const float* data;
  const float* const* table;
  int pad[16];
  int f0[64];      // + 32 + 64 bytes: beyond the 16 preloaded registers
  int f1[256];
  int f2[64];
};

#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) st[k] = t_; } while (0)

__global__ __launch_bounds__(64) void probe(const Big b) {
  typedef const Big __attribute__((address_space(4))) * BK;
  const BK k = (BK)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned long long* st = k->stamps + blockIdx.x * 16;
  STAMP(0);
  STAMP(1);                                    // (the stamp's own cost)
  const int a0 = k->f0[3];
  STAMP(2);                                    // first argument line
  const int a1 = k->f1[200];
  STAMP(3);                                    // another line
  const int a2 = k->f1[201];
  STAMP(4);                                    // the same line again
  const float d0 = k->data[blockIdx.x * 4096 + threadIdx.x];
  STAMP(5);                                    // device memory, first touch
  const float d1 = k->data[blockIdx.x * 4096 + 64 + threadIdx.x];
  STAMP(6);                                    // the next lines
  const float* p = k->table[blockIdx.x];
  const float d2 = p[threadIdx.x];
  STAMP(7);                                    // pointer table -> data (two dependent trips)
  float acc = d0 + d1 + d2 + (float)(a0 + a1 + a2);
#pragma unroll
  for (int i = 0; i < 256; ++i) acc = __builtin_fmaf(acc, 1.0001f, (float)i);   // 256 dependent instructions of straight-line code: 2 KB
  STAMP(8);
  const int a3 = k->f2[5];
  STAMP(9);                                    // one more argument line, late
  k->out[blockIdx.x * 64 + threadIdx.x] = acc + a3;
  STAMP(10);
}

__global__ void trash(float* buf, long n) {   // everyone reads and writes a lot of other memory: caches and TLBs forget the probe's
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) buf[i] = buf[i] * 1.0001f + 1.f;
}
__global__ void writer(float* data, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) data[i] = (float)i;
}

int main() {
  const int WG = 16;
  Big b{};
  float *out, *data, *buf;
  const float** table;
  unsigned long long* stamps;
  hipMalloc(&out, WG * 64 * 4);
  hipMalloc(&data, WG * 4096 * 4 + (1 << 20));
  hipMalloc(&buf, 256l << 20);
  hipMalloc(&table, WG * 8);
  hipMalloc(&stamps, 4 * WG * 16 * 8);
  std::vector<const float*> ht(WG);
  for (int i = 0; i < WG; ++i) ht[i] = data + WG * 4096 + i * 8192;
  hipMemcpy(table, ht.data(), WG * 8, hipMemcpyHostToDevice);
  b.out = out, b.data = data, b.table = (const float* const*)table;
  hipStream_t s;
  hipStreamCreate(&s);
  for (int mode = 0; mode < 2; ++mode) {
    hipGraph_t g;
    hipGraphExec_t ge;
    if (mode == 1) hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int rep = 0; rep < (mode ? 1 : 3); ++rep) {
      hipLaunchKernelGGL(writer, dim3(256), dim3(256), 0, s, data, (long)WG * 4096 + (1 << 18));
      hipLaunchKernelGGL(trash, dim3(2048), dim3(256), 0, s, buf, (long)(256l << 18));
      for (int j = 0; j < 4; ++j) {
        b.stamps = stamps + j * WG * 16;
        hipLaunchKernelGGL(probe, dim3(WG), dim3(64), 0, s, b);
      }
    }
    if (mode == 1) {
      hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int rep = 0; rep < 3; ++rep) hipGraphLaunch(ge, s);
    }
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(4 * WG * 16);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%s: cycles between stamps, workgroup 0 / 5 / 15 (stamp cost, arg line, another arg line, same line, device line, next line, table->data, 256 instr, late arg line, store)\n",
           mode ? "inside a graph" : "eager launches");
    for (int j = 0; j < 4; ++j)
      for (int w : {0, 5, 15}) {
        printf("  launch %d wg %2d:", j, w);
        for (int q = 1; q <= 10; ++q) printf(" %6llu", h[(j * WG + w) * 16 + q] - h[(j * WG + w) * 16 + q - 1]);
        printf("\n");
      }
  }
  return 0;
}
