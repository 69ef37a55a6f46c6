// Where does the dispatcher put the workgroups of a launch that fills the chip exactly once?  Every workgroup records its XCC,
// shader engine, shader array, CU, and the time it started; the grid and the footprint (256 threads, 19 KB of LDS, <= 64
// registers) are those of the all-pairs tile kernel at B = 4096 (2080 workgroups on 256 CUs x 8 slots).  The question behind it:
// which workgroups share a CU -- their operand loads could hit in that CU's vector L1.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_placement.hip -o tools/ubench_placement ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void probe(Rec* out, int spin) {
  __shared__ float lds[19 * 256];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) {
    lds[(threadIdx.x * 19 + i) % (19 * 256)] = v;
    __syncthreads();
    v = v * 1.0001f + lds[(threadIdx.x * 7 + i) % (19 * 256)];
  }
  if (threadIdx.x == 0) out[blockIdx.x] = Rec{hw, xcc, t0, wall_clock64() + (unsigned long long)(v == 123.f)};
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 2080, spin = argc > 2 ? atoi(argv[2]) : 2000;
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * grid);
  std::vector<Rec> h(grid);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, spin);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (auto& r : h) tmin = r.t0 < tmin ? r.t0 : tmin;
  // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
  std::map<unsigned, std::vector<int>> by_cu;
  printf("# blockIdx xcc se sh cu start_ticks(100MHz) dur_ticks\n");
  for (int b = 0; b < grid; ++b) {
    const Rec& r = h[b];
    const unsigned cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 15;
    if (b < 96 || b % 97 == 0) printf("%d %u %u %u %u %llu %llu\n", b, xcc, se, sh, cu, r.t0 - tmin, r.t1 - r.t0);
    by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("# distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 40) break;
    printf("xcc %u se %u sh %u cu %u :", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
    for (int b : kv.second) printf(" %d(k=%d)", b, b >> 3);
    printf("\n");
  }
  // is blockIdx % 8 the XCC?
  int same = 0;
  for (int b = 0; b < grid; ++b) same += ((h[b].xcc_id & 15) == (h[0].xcc_id & 15)) == ((b & 7) == 0);
  printf("# blockIdx %% 8 decides the XCC for %d of %d workgroups\n", same, grid);
  // k-differences between the workgroups of one CU, histogram
  std::map<int, int> hist;
  for (auto& kv : by_cu)
    for (size_t i = 1; i < kv.second.size(); ++i) hist[(kv.second[i] >> 3) - (kv.second[i - 1] >> 3)]++;
  printf("# k-stride between consecutive workgroups of a CU (k = blockIdx >> 3): ");
  for (auto& kv : hist) printf("%d:%d ", kv.first, kv.second);
  printf("\n");
  return 0;
}
