// Do the vector unit, the matrix pipe and the LDS pipe of a CU work at the same time when DIFFERENT waves of a SIMD use them?
// Eight waves per SIMD (2048 workgroups of 256 threads, <= 64 registers, as the all-pairs tile kernel): the waves of every other
// WORKGROUP run a vector loop (independent fused multiply-adds + one rsq per eight), the others a matrix loop
// (v_mfma_f32_32x32x16_f16 back to back), an LDS loop (ds_write_b128 + 2 ds_read_b128, the tile kernel's staging mix), or both.
// Times of the roles alone and together: together = max(...) means the pipes overlap across waves, = sum means they do not.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// role bits: 1 = vector loop on even workgroups, 2 = matrix loop on odd workgroups, 4 = LDS loop on odd workgroups
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void roles(float* out, int roles_mask, int n_valu, int n_mfma, int n_lds) {
  __shared__ __align__(16) u32x4 buf[1024];   // 16 KB
  const int tid = threadIdx.x;
  const bool even = ((blockIdx.x >> 3) & 1) == 0;      // (blockIdx & 7 is the XCD: alternate inside an XCD's list)
  float acc = 0.f;
  if (even) {
    if (roles_mask & 1) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 1.f + tid * 1e-3f + i;
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = fmaf(v[i], 1.0001f, 0.5f);
          v[i] = fmaf(v[i], v[i], 1.f);
          v[i] = fmaf(v[i], 0.999f, -0.25f);
          v[i] = v[i] * 0.5f;
        }
        v[it & 7] = __builtin_amdgcn_rsqf(v[it & 7] * v[it & 7] + 1.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
  } else {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(tid * 1e-3f + e), b[e] = (_Float16)(1.f + e);
    u32x4 w = {(unsigned)tid, 1u, 2u, 3u};
    const int n_it = max(n_mfma, n_lds);
    for (int it = 0; it < n_it; ++it) {
      if ((roles_mask & 4) && it < n_lds) {
        buf[(tid * 5 + it) & 1023] = w;
        const u32x4 r0 = buf[(tid + 17 * it) & 1023], r1 = buf[(tid * 3 + it) & 1023];
        w = w + r0 + r1;
      }
      if ((roles_mask & 2) && it < n_mfma) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
      }
    }
    acc = c[0] + c[7] + (float)w[0];
  }
  if (acc == 12345.678f) out[blockIdx.x * 256 + tid] = acc;
}

int main() {
  float* out;
  hipMalloc(&out, 2080 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int NV = 60, NM = 96, NL = 96;
  auto run = [&](int mask) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(roles, dim3(2048), dim3(256), 0, 0, out, mask, NV, NM, NL);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    return best * 100.f;   // us per launch
  };
  const char* names[8] = {"nothing", "vector", "matrix", "vector + matrix", "lds", "vector + lds", "matrix + lds", "vector + matrix + lds"};
  for (int mask = 0; mask < 8; ++mask) printf("%-24s %8.2f us per launch\n", names[mask], run(mask));
  return 0;
}
