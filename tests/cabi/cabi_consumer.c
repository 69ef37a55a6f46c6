/* A plain-C consumer of libemx.so (include/emx.h): no torch, no C++, device memory through the HIP runtime API only.
 * Test infrastructure (tests/test_cabi_c.py builds it with gcc):
 *   cabi_consumer            -> prints version / target, checks the argument validation path (no GPU needed)
 *   cabi_consumer --gpu      -> known answers of the reference's own tests through the ABI on device 0:
 *       pairwise_dist_periodic([[1/8,1/2],[7/8,1/2]], 1) == [[0,1/4],[1/4,0]],  periodicity inf -> 6/8
 *                                                                 (tests/test_pairwise_distances.py:140-153)
 *       sketch-map cost of two coincident point sets is 0, center cost of a latent of ones == center scale
 *                                                                 (tests/test_losses.py:815-907)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emx.h"

#define CHECK_HIP(x)                                                                  \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)
#define CHECK_EMX(x)                                                                  \
  do {                                                                                \
    int r_ = (x);                                                                     \
    if (r_ != EMX_OK) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #x, r_, emx_last_error());                    \
      return 3;                                                                       \
    }                                                                                 \
  } while (0)

static int close_to(float a, float b, float tol) { return fabsf(a - b) <= tol; }

int main(int argc, char** argv) {
  printf("emx_version %d target %s\n", emx_version(), emx_target_arch());
  if (emx_version() != EMX_VERSION || strcmp(emx_target_arch(), "gfx950") != 0) return 1;
  /* rejected before any launch: null pointers -> EMX_EINVAL with a message (legal without a GPU) */
  if (emx_pairwise_dist(NULL, 2, 2, 2, INFINITY, NULL, NULL) != EMX_EINVAL || strlen(emx_last_error()) == 0) return 1;
  if (emx_sketchmap_ws_bytes(256, 2) == 0) return 1;
  if (argc < 2 || strcmp(argv[1], "--gpu") != 0) {
    printf("ok (no GPU part)\n");
    return 0;
  }

  CHECK_HIP(hipSetDevice(0));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  const float x_h[4] = {1.f / 8, 1.f / 2, 7.f / 8, 1.f / 2};
  float *x, *out;
  float o[4];
  CHECK_HIP(hipMalloc((void**)&x, sizeof x_h));
  CHECK_HIP(hipMalloc((void**)&out, sizeof o));
  CHECK_HIP(hipMemcpy(x, x_h, sizeof x_h, hipMemcpyHostToDevice));
  CHECK_EMX(emx_pairwise_dist(x, 2, 2, 2, 1.0f, out, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  printf("periodic 1: %g %g %g %g\n", o[0], o[1], o[2], o[3]);
  if (!(close_to(o[0], 0.f, 1e-6f) && close_to(o[1], 0.25f, 1e-6f) && close_to(o[2], 0.25f, 1e-6f) && close_to(o[3], 0.f, 1e-6f))) return 4;
  CHECK_EMX(emx_pairwise_dist(x, 2, 2, 2, INFINITY, out, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  printf("euclidean: %g %g %g %g\n", o[0], o[1], o[2], o[3]);
  if (!(close_to(o[1], 0.75f, 1e-6f) && close_to(o[2], 0.75f, 1e-6f) && close_to(o[0], 0.f, 1e-6f))) return 4;

  /* sketch-map cost: latent == high-d points and identical sigmoids -> 0; center cost of ones == its scale */
  enum { B = 300, D = 2, L = 2 };
  static float xs[B * D], ones[B * L];
  unsigned s = 12345u;
  for (int i = 0; i < B * D; ++i) {
    s = s * 1664525u + 1013904223u;
    xs[i] = (float)(s >> 8) / 16777216.f * 4.f;
  }
  for (int i = 0; i < B * L; ++i) ones[i] = 1.f;
  const float sig[6] = {1.f, 2.f, 6.f, 1.f, 2.f, 6.f};
  float *X, *Z, *dZ, *loss, *closs;
  void* ws;
  const size_t wsb = emx_sketchmap_ws_bytes(B, L);
  CHECK_HIP(hipMalloc((void**)&X, sizeof xs));
  CHECK_HIP(hipMalloc((void**)&Z, sizeof xs));
  CHECK_HIP(hipMalloc((void**)&dZ, sizeof xs));
  CHECK_HIP(hipMalloc((void**)&loss, 4));
  CHECK_HIP(hipMalloc((void**)&closs, 4));
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_HIP(hipMemset(ws, 0, wsb));   /* emx.h: zero once before the first use */
  CHECK_HIP(hipMemcpy(X, xs, sizeof xs, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(Z, xs, sizeof xs, hipMemcpyHostToDevice));
  CHECK_EMX(emx_sketchmap_fwd_bwd(X, B, D, D, Z, L, INFINITY, 0.f /* data span: unknown */, sig, 500.f, 1e-4f, 0, B, loss, closs, dZ, ws, wsb, st));
  CHECK_HIP(hipStreamSynchronize(st));
  float l = -1.f, c = -1.f;
  CHECK_HIP(hipMemcpy(&l, loss, 4, hipMemcpyDeviceToHost));
  printf("sketch-map cost of identical point sets: %g\n", l);
  if (!(l >= 0.f && l < 1e-9f)) return 5;
  CHECK_HIP(hipMemcpy(Z, ones, sizeof ones, hipMemcpyHostToDevice));
  CHECK_EMX(emx_sketchmap_fwd_bwd(X, B, D, D, Z, L, INFINITY, 0.f /* data span: unknown */, sig, 500.f, 1e-4f, 0, B, loss, closs, dZ, ws, wsb, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(&c, closs, 4, hipMemcpyDeviceToHost));
  printf("center cost of a latent of ones: %g\n", c);
  if (!close_to(c, 1e-4f, 1e-9f)) return 5;
  printf("ok\n");
  return 0;
}
