/* A plain-C consumer of libemx.so (include/emx.h): no torch, no C++, device memory through the HIP runtime API only.
 * Test infrastructure (tests/test_cabi_c.py builds it with gcc):
 *   cabi_consumer            -> prints version / target, checks the argument validation path (no GPU needed)
 *   cabi_consumer --gpu      -> known answers of the reference's own tests through the ABI on device 0:
 *       pairwise_dist_periodic([[1/8,1/2],[7/8,1/2]], 1) == [[0,1/4],[1/4,0]],  periodicity inf -> 6/8
 *                                                                 (tests/test_pairwise_distances.py:140-153)
 *       sketch-map cost of two coincident point sets is 0, center cost of a latent of ones == center scale
 *                                                                 (tests/test_losses.py:815-907)
 *       emx_mlp_infer: an encoder D -> 128 -> 128 -> 2 (tanh, tanh, linear) on 1000 rows against the same layers in double here --
 *       D = 20 (resident pieces) and D = 300 (one wide end: staged pieces); autoencoder.py:1110-1136, models.py:3331-3359
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

/* an encoder of three Dense layers through emx_mlp_infer against a double loop; returns 0 when every latent value agrees to 5e-6 */
static int check_infer(int D, hipStream_t st) {
  enum { H = 128, Lz = 2, ROWS = 1000 };
  emx_mlp_desc desc;
  memset(&desc, 0, sizeof desc);
  desc.n_layers = 3;
  desc.width[0] = D, desc.width[1] = H, desc.width[2] = H, desc.width[3] = Lz;
  desc.act[0] = 1, desc.act[1] = 1, desc.act[2] = 0;   /* tanh, tanh, linear */
  long off = 0;
  for (int l = 0; l < 3; ++l) {
    desc.w_off[l] = off, off += (long)desc.width[l] * desc.width[l + 1];
    desc.b_off[l] = off, off += desc.width[l + 1];
  }
  float* p_h = (float*)malloc(off * sizeof(float));
  float* x_h = (float*)malloc((size_t)ROWS * D * sizeof(float));
  float z_h[ROWS * Lz];
  unsigned s = 777u + (unsigned)D;
  for (long i = 0; i < off; ++i) {
    s = s * 1664525u + 1013904223u;
    p_h[i] = ((float)(s >> 8) / 16777216.f - 0.5f) * 0.3f;
  }
  for (long i = 0; i < (long)ROWS * D; ++i) {
    s = s * 1664525u + 1013904223u;
    x_h[i] = ((float)(s >> 8) / 16777216.f - 0.5f) * 4.f;
  }
  float *p_d, *x_d, *z_d;
  CHECK_HIP(hipMalloc((void**)&p_d, off * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&x_d, (size_t)ROWS * D * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&z_d, sizeof z_h));
  CHECK_HIP(hipMemcpy(p_d, p_h, off * sizeof(float), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(x_d, x_h, (size_t)ROWS * D * sizeof(float), hipMemcpyHostToDevice));
  CHECK_EMX(emx_mlp_infer(&desc, p_d, 0, 3, x_d, D, 0.f, ROWS, z_d, Lz, 0.f, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(z_h, z_d, sizeof z_h, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int r = 0; r < ROWS; ++r) {
    double a[300], b[H];
    for (int k = 0; k < D; ++k) a[k] = x_h[(long)r * D + k];
    int K = D;
    for (int l = 0; l < 3; ++l) {
      const int N = desc.width[l + 1];
      for (int n = 0; n < N; ++n) {
        double acc = p_h[desc.b_off[l] + n];
        for (int k = 0; k < K; ++k) acc += a[k] * (double)p_h[desc.w_off[l] + (long)k * N + n];
        b[n] = desc.act[l] ? tanh(acc) : acc;
      }
      for (int n = 0; n < N; ++n) a[n] = b[n];
      K = N;
    }
    for (int n = 0; n < Lz; ++n) {
      const double e = fabs((double)z_h[r * Lz + n] - a[n]);
      if (e > worst) worst = e;
    }
  }
  printf("emx_mlp_infer, %d -> 128 -> 128 -> 2 on %d rows: largest |z - double| %.3g\n", D, (int)ROWS, worst);
  free(p_h), free(x_h);
  hipFree(p_d), hipFree(x_d), hipFree(z_d);
  return worst < 5e-6 ? 0 : 6;
}

int main(int argc, char** argv) {
  printf("emx_version %d target %s\n", emx_version(), emx_target_arch());
  if (emx_version() != EMX_VERSION || strcmp(emx_target_arch(), "gfx950") != 0) return 1;
  /* rejected before any launch: null pointers -> EMX_EINVAL with a message (legal without a GPU) */
  if (emx_pairwise_dist(NULL, 2, 2, 2, INFINITY, NULL, NULL) != EMX_EINVAL || strlen(emx_last_error()) == 0) return 1;
  if (emx_sketchmap_ws_bytes(256, 2) == 0) return 1;
  {
    /* emx_mlp_infer: a layer range outside the network and a row stride below the row length are refused before any launch */
    emx_mlp_desc d;
    memset(&d, 0, sizeof d);
    d.n_layers = 1, d.width[0] = 4, d.width[1] = 2;
    float dummy[8] = {0};
    if (emx_mlp_infer(&d, dummy, 0, 2, dummy, 4, 0.f, 1, dummy, 2, 0.f, NULL) != EMX_EINVAL) return 1;
    if (emx_mlp_infer(&d, dummy, 0, 1, dummy, 3, 0.f, 1, dummy, 2, 0.f, NULL) != EMX_EINVAL || strlen(emx_last_error()) == 0) return 1;
    if (emx_mlp_infer(&d, dummy, 0, 1, NULL, 4, 0.f, 0, NULL, 2, 0.f, NULL) != EMX_OK) return 1;   /* no rows: nothing to do */
  }
  {
    /* emx_p2p_*: a rank outside the world and a chunk count beyond EMX_P2P_MAX_CHUNKS are refused before any device call */
    void* ctx = NULL;
    char handle[256];
    if (emx_p2p_handle_bytes() <= 0 || emx_p2p_handle_bytes() > (int)sizeof handle) return 1;
    if (emx_p2p_create(2, 2, 100, 8, &ctx, handle) != EMX_EINVAL || ctx != NULL) return 1;
    if (emx_p2p_create(1, 0, 100, EMX_P2P_MAX_CHUNKS + 1, &ctx, handle) != EMX_EINVAL) return 1;
    if (emx_p2p_allreduce(NULL, NULL, 1, 1.f, NULL) != EMX_EINVAL) return 1;
  }
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
  {
    int rc = check_infer(20, st);
    if (rc) return rc;
    rc = check_infer(300, st);
    if (rc) return rc;
  }
  {
    /* the one-shot all-reduce with the one rank a single process is: out = scale * in, through the whole protocol (region, handle,
     * connect, publish / flag / read) */
    enum { NP = 1000 };
    void* ctx = NULL;
    char handle[256];
    float v_h[NP], *v_d;
    for (int i = 0; i < NP; ++i) v_h[i] = (float)(i - 500) * 0.25f;
    CHECK_HIP(hipMalloc((void**)&v_d, sizeof v_h));
    CHECK_HIP(hipMemcpy(v_d, v_h, sizeof v_h, hipMemcpyHostToDevice));
    CHECK_EMX(emx_p2p_create(1, 0, NP, 4, &ctx, handle));
    CHECK_EMX(emx_p2p_connect(ctx, handle));
    CHECK_EMX(emx_p2p_allreduce(ctx, v_d, NP, 0.5f, st));
    CHECK_EMX(emx_p2p_allreduce(ctx, v_d, NP, 0.5f, st));
    CHECK_HIP(hipStreamSynchronize(st));
    int gave_up = -1;
    CHECK_EMX(emx_p2p_status(ctx, &gave_up));
    float w_h[NP];
    CHECK_HIP(hipMemcpy(w_h, v_d, sizeof w_h, hipMemcpyDeviceToHost));
    for (int i = 0; i < NP; ++i)
      if (w_h[i] != v_h[i] * 0.25f) return 7;
    if (gave_up != 0) return 7;
    CHECK_EMX(emx_p2p_destroy(ctx));
    hipFree(v_d);
    printf("emx_p2p_allreduce with one rank: twice 0.5 x\n");
  }
  printf("ok\n");
  return 0;
}
