// Is split_f16_pair (v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16) bit-identical to the compiler's form?  hipcc --offload-arch=gfx950 -O3
// -ffp-contract=on -I encodermap_amd/csrc -I include tools/experiments/split_pair_check.hip -o build_ab/split_pair_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
__device__ void pair_asm(float v0, float v1, unsigned& hi, unsigned& lo) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}
__device__ void pair_plain(float v0, float v1, unsigned& hi, unsigned& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const _Float16 a = (_Float16)v0, b = (_Float16)v1;
  const _Float16 la = (_Float16)(v0 - (float)a), lb = (_Float16)(v1 - (float)b);
  hi = __builtin_bit_cast(unsigned, h2{a, b}), lo = __builtin_bit_cast(unsigned, h2{la, lb});
}
__global__ void k(const float* x, unsigned* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h0, l0, h1, l1;
  pair_asm(x[2 * i], x[2 * i + 1], h0, l0);
  pair_plain(x[2 * i], x[2 * i + 1], h1, l1);
  o[4 * i] = h0, o[4 * i + 1] = l0, o[4 * i + 2] = h1, o[4 * i + 3] = l1;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const int e = (int)((s >> 8) % 44) - 28;          // magnitudes 2^-28 .. 2^15
    s = s * 1664525u + 1013904223u;
    const float m = 1.f + (float)(s >> 9) / 8388608.f;
    x[i] = ((s & 1) ? -1.f : 1.f) * ldexpf(m, e);
  }
  x[0] = 0.f; x[1] = -0.f; x[2] = 16384.f; x[3] = -16384.f; x[4] = 65504.f; x[5] = 6.1e-5f; x[6] = 5.96e-8f; x[7] = 3e-8f;
  float* dx; unsigned* d_o;
  hipMalloc(&dx, n * 4); hipMalloc(&d_o, n * 8);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, d_o, n);
  std::vector<unsigned> o(2 * n);
  hipMemcpy(o.data(), d_o, n * 8, hipMemcpyDeviceToHost);
  long bad_hi = 0, bad_lo = 0; int shown = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (o[4 * i] != o[4 * i + 2]) ++bad_hi;
    if (o[4 * i + 1] != o[4 * i + 3]) {
      ++bad_lo;
      if (shown++ < 12) printf("x = %.9g %.9g: hi %08x lo asm %08x plain %08x\n", x[2 * i], x[2 * i + 1], o[4 * i], o[4 * i + 1], o[4 * i + 3]);
    }
  }
  printf("pairs %d: hi differs %ld, lo differs %ld\n", n / 2, bad_hi, bad_lo);
  return 0;
}
