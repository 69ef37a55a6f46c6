// ADC path: dihedral <-> Cartesian back-mapping and per-frame pair distances (gfx950).
//
//   K8  emx_colmean                    MeanAngles / mean bond lengths     encodermap/models/layers.py:971,1153-1161
//   K6  emx_chain_in_plane_fwd/bwd     chain_in_plane                     encodermap/encodermap_tf1/backmapping.py:97-119
//   K7  emx_dihedrals_to_cartesian_*   dihedrals_to_cartesian_tf(_layers) encodermap/encodermap_tf1/backmapping.py:164-214,
//                                                                         encodermap/misc/backmapping.py:179-309,1873-1912
//                                      rotation_matrix                    encodermap/encodermap_tf1/misc.py:286-304
//   K3  emx_pdist_flat_fwd/bwd         PairwiseDistances / pairwise_dist(flat=True)   layers.py:1252-1267, distances.py:235-253
//
// K7 algorithm.  The reference twists the planar chain with A/2 sequential steps per direction, each rotating the
// whole remaining tail about the current bond (O(A^2) per frame, hundreds of tiny launches).  The same result is a
// composition of rigid transforms along the chain: with the ORIGINAL (planar) coordinates P_k, step i is the rotation
// N_i by the dihedral d_i about the original bond (P_{i+1}, P_{i+2}) through P_{i+2}, and
//     out_k = A_{k-3} P_k,   A_i = A_{i-1} o N_i,   A_{-1} = identity            (k >= 3; out_k = P_k for k < 3)
// (induction: the axis of step i, taken from the already twisted chain, is A_{i-1} applied to the original bond).
// One thread walks one direction of one frame: O(A) work, nothing materialised.  The reference's row-vector
// "(x - o) @ R(u, -d)" equals the usual right-handed rotation by +d about u acting on column vectors.
// The backward pass walks the chain in reverse, recovering A_{i-1} = A_i o N_i^{-1} instead of storing it, and
// accumulates the adjoints of the transform, of the dihedral and of the planar coordinates.
#include "adc_head.h"
#include "cost.h"
#include "emx_common.h"
#include "split.h"

namespace {

using namespace emx;
constexpr int NT = 256;
constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.283185307179586f;

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct M3 {
  float m[3][3];
};
__device__ __forceinline__ V3 mul(const M3& A, V3 v) {
  return V3{A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3& A, V3 v) {  // A^T v
  return V3{A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
__device__ __forceinline__ M3 mulBT(const M3& A, const M3& B) {  // A B^T
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[j][0] + A.m[i][1] * B.m[j][1] + A.m[i][2] * B.m[j][2];
  return C;
}
__device__ __forceinline__ M3 mulAT(const M3& A, const M3& B) {  // A^T B
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[0][i] * B.m[0][j] + A.m[1][i] * B.m[1][j] + A.m[2][i] * B.m[2][j];
  return C;
}
__device__ __forceinline__ M3 identity() {
  M3 I;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) I.m[i][j] = (i == j) ? 1.f : 0.f;
  return I;
}
// right-handed rotation by angle (c = cos, s = sin) about the unit axis u (Rodrigues, encodermap_tf1/misc.py:286-304 transposed)
__device__ __forceinline__ M3 rodrigues(V3 u, float c, float s) {
  const float t = 1.f - c;
  M3 Q;
  Q.m[0][0] = c + t * u.x * u.x;
  Q.m[0][1] = t * u.x * u.y - s * u.z;
  Q.m[0][2] = t * u.x * u.z + s * u.y;
  Q.m[1][0] = t * u.x * u.y + s * u.z;
  Q.m[1][1] = c + t * u.y * u.y;
  Q.m[1][2] = t * u.y * u.z - s * u.x;
  Q.m[2][0] = t * u.x * u.z - s * u.y;
  Q.m[2][1] = t * u.y * u.z + s * u.x;
  Q.m[2][2] = c + t * u.z * u.z;
  return Q;
}

// ---------------------------------------------------------------------------------------------------------
// K8: column mean.  block = 64 columns x 4 row phases; out[k] = mean_b x[b, k]
// ---------------------------------------------------------------------------------------------------------
constexpr int CM_PH = 16;   // row phases per workgroup: 64 columns x 16 phases = 1024 threads (a few hundred rows, a few hundred
                            // columns: four phases left 64 dependent-ish rows per thread on four workgroups)
__global__ __launch_bounds__(64 * CM_PH) void colmean_kernel(const float* __restrict__ x, int B, int n, int ldx, float* __restrict__ out) {
  __shared__ float part[CM_PH][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    int b = ph;
    for (; b + 3 * CM_PH < B; b += 4 * CM_PH) {   // 4 independent loads in flight per thread
      s0 += x[(long)b * ldx + c];
      s1 += x[(long)(b + CM_PH) * ldx + c];
      s2 += x[(long)(b + 2 * CM_PH) * ldx + c];
      s3 += x[(long)(b + 3 * CM_PH) * ldx + c];
    }
    for (; b < B; b += CM_PH) s0 += x[(long)b * ldx + c];
  }
  part[ph][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < CM_PH; q += 4)
      t += (part[q][threadIdx.x] + part[q + 1][threadIdx.x]) + (part[q + 2][threadIdx.x] + part[q + 3][threadIdx.x]);
    out[c] = t / (float)B;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K6: chain_in_plane.  One thread per frame; O(A) recurrence:
//   P_0 = 0, phi = 0, s = +1;  P_{i+1} = P_i + len_i (cos phi, s sin phi, 0);  phi <- pi - theta_i - phi;  s <- -s
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void chain_fwd_kernel(const float* __restrict__ len, const float* __restrict__ ang, int lda, int B,
                                                       int A, float* __restrict__ xyz) {
  const int b = blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const float* th = ang + (long)b * lda;
  float* o = xyz + (long)b * A * 3;
  float x = 0.f, y = 0.f, phi = 0.f, sg = 1.f;
  o[0] = 0.f;
  o[1] = 0.f;
  o[2] = 0.f;
  for (int i = 0; i < A - 1; ++i) {
    float sn, cs;
    sincosf(phi, &sn, &cs);
    x += len[i] * cs;
    y += len[i] * sn * sg;
    o[3 * (i + 1)] = x;
    o[3 * (i + 1) + 1] = y;
    o[3 * (i + 1) + 2] = 0.f;
    if (i < A - 2) phi = PI_F - th[i] - phi;
    sg = -sg;
  }
}

// backward: dphi_j/dtheta_i = (-1)^(i+j) (i < j); dL/dphi_j = len_j * (-sin phi_j * GX_{j+1} + s_j cos phi_j * GY_{j+1}),
// GX/GY = suffix sums of the coordinate gradients.  Reverse sweep, phi recovered by inverting the recurrence.
__global__ __launch_bounds__(NT) void chain_bwd_kernel(const float* __restrict__ len, const float* __restrict__ ang, int lda,
                                                       const float* __restrict__ dxyz, int B, int A, float* __restrict__ dang, int ldd) {
  const int b = blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const float* th = ang + (long)b * lda;
  const float* g = dxyz + (long)b * A * 3;
  float* d = dang + (long)b * ldd;
  // phi_{A-2} by running the recurrence forward once (cheap: no trigonometry)
  float phi = 0.f;
  for (int i = 0; i < A - 2; ++i) phi = PI_F - th[i] - phi;
  float sg = ((A - 2) & 1) ? -1.f : 1.f;
  float GX = 0.f, GY = 0.f, acc = 0.f;   // acc = sum_{j > i} (-1)^(i+j) dL/dphi_j, maintained as acc <- -(acc) + ...
  for (int j = A - 2; j >= 0; --j) {
    GX += g[3 * (j + 1)];
    GY += g[3 * (j + 1) + 1];
    float sn, cs;
    sincosf(phi, &sn, &cs);
    const float dphi = len[j] * (-sn * GX + sg * cs * GY);
    // for theta_{j-1}: sum_{k >= j} (-1)^{(j-1)+k} dphi_k = -(dphi_j - sum_{k>j} (-1)^{j+k}...)  -> recurrence acc_j = dphi_j - acc_{j+1}
    acc = dphi - acc;
    if (j >= 1) {
      d[j - 1] = -acc;
      phi = PI_F - th[j - 1] - phi;   // phi_{j-1} = pi - theta_{j-1} - phi_j
    }
    sg = -sg;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K7: one thread = one direction (0: right part, 1: left part) of one frame.
//   split s = A/2.  right: atoms s-1, s, s+1, ... A-1 with dihedrals d[s-1 ...];   left: atoms s+1, s, s-1, ... 0 with
//   dihedrals d[s-2], d[s-3], ... d[0]   (encodermap_tf1/backmapping.py:176-181; iteration counts models.py:666-671).
//   Local chain index c = 0.. maps to atom  a(c) = dir ? s + 1 - c : s - 1 + c,  local dihedral i to
//   dih(i) = dir ? s - 2 - i : s - 1 + i.  Local atoms 0..2 stay in place; the right pass owns atoms s-1..s+1.
// ---------------------------------------------------------------------------------------------------------
struct D2C {
  int s, A;
  __device__ __forceinline__ int natoms(int dir) const { return dir ? s + 2 : A - (s - 1); }
  __device__ __forceinline__ int atom(int dir, int c) const { return dir ? s + 1 - c : s - 1 + c; }
  __device__ __forceinline__ int dih(int dir, int i) const { return dir ? s - 2 - i : s - 1 + i; }
};

__global__ __launch_bounds__(NT) void d2c_fwd_kernel(const float* __restrict__ dih, int ldd, float offset, const float* __restrict__ plane,
                                                     int B, int A, float* __restrict__ out) {
  const int t = blockIdx.x * NT + threadIdx.x;
  const int b = t >> 1, dir = t & 1;
  if (b >= B) return;
  const D2C g{A / 2, A};
  const float* P = plane + (long)b * A * 3;
  const float* dd = dih + (long)b * ldd;
  float* O = out + (long)b * A * 3;
  const int n = g.natoms(dir);
  // fixed atoms (the left pass leaves s-1..s+1 to the right pass)
  if (dir == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < n) st3(O + 3 * g.atom(0, c), ld3(P + 3 * g.atom(0, c)));
  }
  M3 R = identity();
  V3 tr{0.f, 0.f, 0.f};
  V3 p1 = n > 1 ? ld3(P + 3 * g.atom(dir, 1)) : V3{0, 0, 0};
  V3 p2 = n > 2 ? ld3(P + 3 * g.atom(dir, 2)) : V3{0, 0, 0};
  for (int i = 0; i + 3 < n; ++i) {
    // N_i: rotation by d_i about u = unit(p2 - p1) through o = p2
    V3 w = p2 - p1;
    const float inv = rsqrtf(dot(w, w));
    const V3 u = inv * w;
    float sn, cs;
    sincosf(dd[g.dih(dir, i)] + offset, &sn, &cs);
    const M3 Q = rodrigues(u, cs, sn);
    const V3 c = p2 - mul(Q, p2);
    tr = mul(R, c) + tr;
    R = mul(R, Q);
    const V3 p3 = ld3(P + 3 * g.atom(dir, i + 3));
    st3(O + 3 * g.atom(dir, i + 3), mul(R, p3) + tr);
    p1 = p2;
    p2 = p3;
  }
}

// backward: gradients wrt the dihedrals and the planar coordinates.  dplane must be zero-initialised by the host
// wrapper (both directions add into the shared atoms s-1..s+1 -> done by ONE thread per frame walking both directions).
__global__ __launch_bounds__(NT) void d2c_bwd_kernel(const float* __restrict__ dih, int ldd, float offset, const float* __restrict__ plane,
                                                     const float* __restrict__ dout, int B, int A, float* __restrict__ ddih, int ldg,
                                                     float* __restrict__ dplane) {
  const int b = blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const D2C g{A / 2, A};
  const float* P = plane + (long)b * A * 3;
  const float* dd = dih + (long)b * ldd;
  const float* G = dout + (long)b * A * 3;
  float* DD = ddih + (long)b * ldg;
  float* DP = dplane + (long)b * A * 3;
  for (int k = 0; k < A; ++k) st3(DP + 3 * k, V3{0.f, 0.f, 0.f});
  // atoms s-1, s, s+1 pass straight through
  for (int c = 0; c < 3; ++c) {
    const int a = g.atom(0, c);
    if (a < A && a >= 0) st3(DP + 3 * a, ld3(G + 3 * a));
  }
  for (int dir = 0; dir < 2; ++dir) {
    const int n = g.natoms(dir);
    const int nd = n - 3;   // number of dihedral steps
    if (nd <= 0) continue;
    // forward sweep to the final transform A_{nd-1}
    M3 R = identity();
    V3 tr{0.f, 0.f, 0.f};
    for (int i = 0; i < nd; ++i) {
      const V3 p1 = ld3(P + 3 * g.atom(dir, i + 1)), p2 = ld3(P + 3 * g.atom(dir, i + 2));
      V3 w = p2 - p1;
      const V3 u = rsqrtf(dot(w, w)) * w;
      float sn, cs;
      sincosf(dd[g.dih(dir, i)] + offset, &sn, &cs);
      const M3 Q = rodrigues(u, cs, sn);
      tr = mul(R, p2 - mul(Q, p2)) + tr;
      R = mul(R, Q);
    }
    // reverse sweep
    M3 Rb;   // adjoint of R_i
    V3 tb{0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rb.m[r][c] = 0.f;
    for (int i = nd - 1; i >= 0; --i) {
      // out_{i+3} = R_i P_{i+3} + t_i
      const int a3 = g.atom(dir, i + 3);
      const V3 gk = ld3(G + 3 * a3), p3 = ld3(P + 3 * a3);
      Rb.m[0][0] += gk.x * p3.x; Rb.m[0][1] += gk.x * p3.y; Rb.m[0][2] += gk.x * p3.z;
      Rb.m[1][0] += gk.y * p3.x; Rb.m[1][1] += gk.y * p3.y; Rb.m[1][2] += gk.y * p3.z;
      Rb.m[2][0] += gk.z * p3.x; Rb.m[2][1] += gk.z * p3.y; Rb.m[2][2] += gk.z * p3.z;
      tb = tb + gk;
      {
        const V3 add = mulT(R, gk);
        float* q = DP + 3 * a3;
        q[0] += add.x; q[1] += add.y; q[2] += add.z;
      }
      // step i: R_i = R_{i-1} Q_i,  t_i = R_{i-1} c_i + t_{i-1},  c_i = o - Q_i o,  o = P_{i+2}
      const int a1 = g.atom(dir, i + 1), a2 = g.atom(dir, i + 2);
      const V3 p1 = ld3(P + 3 * a1), o = ld3(P + 3 * a2);
      V3 w = o - p1;
      const float inv = rsqrtf(dot(w, w));
      const V3 u = inv * w;
      float sn, cs;
      sincosf(dd[g.dih(dir, i)] + offset, &sn, &cs);
      const M3 Q = rodrigues(u, cs, sn);
      const V3 c = o - mul(Q, o);
      // recover R_{i-1}, t_{i-1}
      const M3 Rp = mulBT(R, Q);
      const V3 tp = tr - mul(Rp, c);
      // adjoints
      M3 Qb = mulAT(Rp, Rb);           // from R_i = Rp Q
      const V3 cb = mulT(Rp, tb);      // from t_i = Rp c + tp
      M3 Rpb = mulBT(Rb, Q);           // Rb Q^T
      Rpb.m[0][0] += tb.x * c.x; Rpb.m[0][1] += tb.x * c.y; Rpb.m[0][2] += tb.x * c.z;
      Rpb.m[1][0] += tb.y * c.x; Rpb.m[1][1] += tb.y * c.y; Rpb.m[1][2] += tb.y * c.z;
      Rpb.m[2][0] += tb.z * c.x; Rpb.m[2][1] += tb.z * c.y; Rpb.m[2][2] += tb.z * c.z;
      // c = o - Q o
      V3 ob = cb - mulT(Q, cb);
      Qb.m[0][0] -= cb.x * o.x; Qb.m[0][1] -= cb.x * o.y; Qb.m[0][2] -= cb.x * o.z;
      Qb.m[1][0] -= cb.y * o.x; Qb.m[1][1] -= cb.y * o.y; Qb.m[1][2] -= cb.y * o.z;
      Qb.m[2][0] -= cb.z * o.x; Qb.m[2][1] -= cb.z * o.y; Qb.m[2][2] -= cb.z * o.z;
      // Q(u, d) = cs I + sn [u]x + (1 - cs) u u^T
      const float trQ = Qb.m[0][0] + Qb.m[1][1] + Qb.m[2][2];
      const V3 ax{Qb.m[2][1] - Qb.m[1][2], Qb.m[0][2] - Qb.m[2][0], Qb.m[1][0] - Qb.m[0][1]};   // <Qb, [v]x> = v . ax
      const V3 Qu = mul(Qb, u), QTu = mulT(Qb, u);
      const float uQu = dot(u, Qu);
      DD[g.dih(dir, i)] = -sn * trQ + cs * dot(u, ax) + sn * uQu;
      V3 ub = sn * ax + (1.f - cs) * (Qu + QTu);
      // u = w / |w|
      const V3 wb = inv * (ub - dot(u, ub) * u);
      ob = ob + wb;
      {
        float* q = DP + 3 * a2;
        q[0] += ob.x; q[1] += ob.y; q[2] += ob.z;
        float* q1 = DP + 3 * a1;
        q1[0] -= wb.x; q1[1] -= wb.y; q1[2] -= wb.z;
      }
      R = Rp;
      tr = tp;
      Rb = Rpb;
      // tb carries over unchanged (t_i = ... + t_{i-1})
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K7, one workgroup per frame (two waves: wave 0 = right part, wave 1 = left part), the frame and every per-step
// quantity in LDS.  The chain is a product of rigid transforms whose FACTORS do not depend on each other -- step i turns
// about an axis of the PLANAR input chain -- so only the 3x3 / 3-vector recurrences are serial:
//   phase A (lanes over the steps)  Q_i, c_i from the planar atoms and the dihedral (sin/cos, Rodrigues)
//   phase B (one lane, LDS only)    R_i = R_{i-1} Q_i, t_i = R_{i-1} c_i + t_{i-1};  backward also, independent of it and
//                                   interleaved with it, the adjoint recurrence  Rb_{i-1} = Rb_i Q_i^T + tb_i (x) c_i + g (x) p
//   phase C (lanes over the steps)  out_{i+3} = R_i P_{i+3} + t_i;  backward: the step's local adjoints (d dihedral, and the
//                                   step's three contributions to d plane, kept per step)
//   phase D (lanes over the atoms)  backward: every atom adds the contributions it receives in a fixed order
// The thread-per-frame kernels above (global loads and library sincosf inside the serial loop, 256 threads on the
// whole chip) took 97 us forward and 667 us backward at B = 256, A = 228; they remain for frames that do not fit LDS.
// ---------------------------------------------------------------------------------------------------------
#ifdef EMX_D2C_TIMING   // per-phase timeline of workgroup 0 (tools/abl_d2c.py)
__device__ long long emx_dbg_d2c[64];
#define DSTAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && threadIdx.x < (BWD ? 256 : 128)) emx_dbg_d2c[(BWD ? 32 : 0) + 8 * (threadIdx.x >> 6) + (k)] = clock64(); } while (0)
__device__ long long emx_dbg_af[32];   // the fused frame passes (tools/abl_d2c.py): wave 0 of workgroup 0, forward 0.., backward 16..
#define AFSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) emx_dbg_af[k] = clock64(); } while (0)
#else
#define DSTAMP(k)
#define AFSTAMP(k)
#endif
template <int V>
struct RingIdx {
  static constexpr int value = V;
};
template <int V>
struct DppC {
  static constexpr int value = V;
};
constexpr int FR_NT_FWD = 128, FR_NT_BWD = 256;   // backward: waves 2, 3 run the adjoint recurrence of the two parts

__device__ __forceinline__ void lds_put9(float* d, const M3& m) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) d[3 * r + c] = m.m[r][c];
}
__device__ __forceinline__ M3 lds_get9(const float* d) {
  M3 m;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) m.m[r][c] = d[3 * r + c];
  return m;
}
__device__ __forceinline__ M3 outer(V3 a, V3 b) {
  M3 m;
  m.m[0][0] = a.x * b.x; m.m[0][1] = a.x * b.y; m.m[0][2] = a.x * b.z;
  m.m[1][0] = a.y * b.x; m.m[1][1] = a.y * b.y; m.m[1][2] = a.y * b.z;
  m.m[2][0] = a.z * b.x; m.m[2][1] = a.z * b.y; m.m[2][2] = a.z * b.z;
  return m;
}
__device__ __forceinline__ M3 add(const M3& a, const M3& b) {
  M3 m;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) m.m[r][c] = a.m[r][c] + b.m[r][c];
  return m;
}

__device__ __forceinline__ M3 zero3() {
  M3 m;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) m.m[r][c] = 0.f;
  return m;
}

// rigid transform x -> R x + t; compose(P, X) = P o X (X first)
struct Aff {
  M3 R;
  V3 t;
};
__device__ __forceinline__ Aff aff_identity() { return Aff{identity(), V3{0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Aff compose(const Aff& P, const Aff& X) { return Aff{mul(P.R, X.R), mul(P.R, X.t) + P.t}; }

// the value of another lane by DPP (row_shr:n 0x110 + n, row_bcast:15 0x142, row_bcast:31 0x143, wave_shr:1 0x138); lanes
// without a source lane, and rows outside the row mask, receive `neutral`
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_from(float v, float neutral) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, neutral), __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ Aff aff_from(const Aff& x) {
  Aff p;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) p.R.m[r][c] = dpp_from<CTRL, ROWS>(x.R.m[r][c], r == c ? 1.f : 0.f);
  p.t = V3{dpp_from<CTRL, ROWS>(x.t.x, 0.f), dpp_from<CTRL, ROWS>(x.t.y, 0.f), dpp_from<CTRL, ROWS>(x.t.z, 0.f)};
  return p;
}
// inclusive prefix sum over the lanes of a wave
__device__ __forceinline__ float wave_scan_add(float v) {
  v += dpp_from<0x111, 0xF>(v, 0.f);
  v += dpp_from<0x112, 0xF>(v, 0.f);
  v += dpp_from<0x114, 0xF>(v, 0.f);
  v += dpp_from<0x118, 0xF>(v, 0.f);
  v += dpp_from<0x142, 0xA>(v, 0.f);
  v += dpp_from<0x143, 0xC>(v, 0.f);
  return v;
}
__device__ __forceinline__ float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
// product of the transforms of all lanes below this one (lane 0: the identity), lower lanes on the left
__device__ __forceinline__ Aff aff_scan_exclusive(Aff x) {
  x = compose(aff_from<0x111, 0xF>(x), x);
  x = compose(aff_from<0x112, 0xF>(x), x);
  x = compose(aff_from<0x114, 0xF>(x), x);
  x = compose(aff_from<0x118, 0xF>(x), x);
  x = compose(aff_from<0x142, 0xA>(x), x);
  x = compose(aff_from<0x143, 0xC>(x), x);
  return aff_from<0x138, 0xF>(x);
}
// the same scan with additions, over 9 + 3 floats
__device__ __forceinline__ void sum_scan_exclusive(M3 a, V3 v, M3& ea, V3& ev) {
  auto round = [&](auto CT, auto RM) __attribute__((always_inline)) {
    constexpr int C = decltype(CT)::value, RW = decltype(RM)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a.m[r][c] += dpp_from<C, RW>(a.m[r][c], 0.f);
    v = V3{v.x + dpp_from<C, RW>(v.x, 0.f), v.y + dpp_from<C, RW>(v.y, 0.f), v.z + dpp_from<C, RW>(v.z, 0.f)};
  };
  round(DppC<0x111>{}, DppC<0xF>{});
  round(DppC<0x112>{}, DppC<0xF>{});
  round(DppC<0x114>{}, DppC<0xF>{});
  round(DppC<0x118>{}, DppC<0xF>{});
  round(DppC<0x142>{}, DppC<0xA>{});
  round(DppC<0x143>{}, DppC<0xC>{});
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) ea.m[r][c] = dpp_from<0x138, 0xF>(a.m[r][c], 0.f);
  ev = V3{dpp_from<0x138, 0xF>(v.x, 0.f), dpp_from<0x138, 0xF>(v.y, 0.f), dpp_from<0x138, 0xF>(v.z, 0.f)};
}

// LDS floats of the per-frame kernels: the frame (and its gradient), then 12 (forward) / 36 (backward) per step
__host__ __device__ inline long d2c_frame_floats(int A, bool bwd) { return (long)(bwd ? 6 : 3) * A + (long)(bwd ? 36 : 12) * (A > 4 ? A - 2 : 2); }

// (the body of the per-frame kernels, also run by the fused ADC frame pass below: NTB threads of one workgroup, every one of them
//  calls it; wave w works on part w & 1, waves >= 2 help in the phases that run over the steps.  P, dd, dout may be LDS or HBM.)
template <bool BWD, int NTB>
__device__ __forceinline__ void d2c_frame_body(const float* __restrict__ dd, float offset, const float* __restrict__ P, const float* __restrict__ dout_row,
                                               int A, float* __restrict__ out_row, float* __restrict__ out_lds, float* __restrict__ ddih_row,
                                               float* __restrict__ dplane_row, float* __restrict__ sm) {
  constexpr int FR_NT = NTB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = wave & 1, role = wave >> 1;   // role 1 (backward only): the adjoint recurrence in phase B
  constexpr int LSTEP = NTB / 2;                // lanes that share one part's steps in phases A and C
  const int slane = lane + 64 * role;
  const D2C g{A / 2, A};
  float* Pf = sm;                        // [A][3] planar chain
  float* Gf = sm + 3 * A;                // [A][3] d loss / d out (backward)
  float* S = sm + (BWD ? 6 : 3) * A;     // per step: [Q 9 | c 3] ([R 9 | t 3] after phase B), backward + [R 9 | tb 3] + [Rb 9 | -]
  constexpr int SW = BWD ? 36 : 12;
  const int n = g.natoms(dir), nd = max(n - 3, 0);
  const int nd0 = max(g.natoms(0) - 3, 0);
  float* Sd = S + (long)(dir ? nd0 : 0) * SW;
  DSTAMP(0);
  // (coordinates relative to the middle atom, the one both parts start from: the map commutes with translations, and a
  //  step's translation part is as large as its axis is far from the origin -- float32 rounding scales with that distance)
  const V3 ctr = ld3(P + 3 * g.s);
  for (int e = tid; e < 3 * A; e += FR_NT) {
    const int c3 = e % 3;
    Pf[e] = P[e] - (c3 == 0 ? ctr.x : c3 == 1 ? ctr.y : ctr.z);
    if (BWD) Gf[e] = dout_row[e];
  }
  __syncthreads();
  DSTAMP(1);
  // ---- phase A
  for (int i = slane; i < nd; i += LSTEP) {
    const V3 p1 = ld3(Pf + 3 * g.atom(dir, i + 1)), p2 = ld3(Pf + 3 * g.atom(dir, i + 2));
    const V3 w = p2 - p1;
    const V3 u = rsqrtf(dot(w, w)) * w;
    float sn, cs;
    emx::sincos_bounded(dd[g.dih(dir, i)] + offset, &sn, &cs);
    const M3 Q = rodrigues(u, cs, sn);
    lds_put9(Sd + (long)i * SW, Q);
    st3(Sd + (long)i * SW + 9, p2 - mul(Q, p2));
  }
  __syncthreads();
  DSTAMP(2);
  // ---- phase B: the recurrences as scans over the lanes.  [R_i | t_i] is the prefix product M_0 ... M_i of the steps' rigid
  //      transforms M = [[Q, c], [0, 1]], an associative product: lane l folds its run of m = ceil(nd / 64) consecutive steps
  //      (local prefixes stay in LDS), the lanes' totals go through an inclusive wave scan (row_shr 1, 2, 4, 8, row_bcast 15,
  //      31: a lane without a partner receives the identity, whose product is exact), and every local prefix is multiplied from
  //      the left by what precedes its lane.  12 + 2 m products of 3 x 4 transforms per lane instead of nd dependent steps on
  //      one lane (160 cycles each: 18 000 of the forward kernel's 21 700 cycles at 228 atoms).
  //      Backward: the adjoint recurrence  V_j = V_{j+1} M_{j+1}^T + g_j (x) [p_j | 1]  unrolls to
  //      V_j = [ (sum_{k >= j} g_k (x) (out_k - t_j)) R_j | sum_{k >= j} g_k ]  (out_k = R_k p_k + t_k, the step's output atom):
  //      a plain suffix sum of g (x) [out | 1] over the steps -- the same scan with additions, lanes in descending step order
  //      -- followed by one change of frame per step.
  if (role == 0 && nd > 0) {
    const int m = (nd + 63) >> 6;
    const int s0 = lane * m;
    constexpr int RO = BWD ? 12 : 0, TO = BWD ? 33 : 9;   // where a step's [R | t] lives
    Aff acc = aff_identity();
    for (int s = 0; s < m; ++s) {
      const int k = s0 + s;
      if (k < nd) {
        float* q = Sd + (long)k * SW;
        const Aff X{lds_get9(q), ld3(q + 9)};
        acc = compose(acc, X);
        lds_put9(q + RO, acc.R);
        st3(q + TO, acc.t);
      }
    }
    const Aff E = aff_scan_exclusive(acc);
    float* O = out_row;
    for (int s = 0; s < m; ++s) {
      const int k = s0 + s;
      if (k < nd) {
        float* q = Sd + (long)k * SW;
        const Aff Lk{lds_get9(q + RO), ld3(q + TO)};
        const Aff Sk = compose(E, Lk);
        const int a3 = g.atom(dir, k + 3);
        const V3 o = mul(Sk.R, ld3(Pf + 3 * a3)) + Sk.t;
        if (BWD) {
          lds_put9(q + RO, Sk.R);
          st3(q + TO, Sk.t);
          lds_put9(q + 24, outer(ld3(Gf + 3 * a3), o));
        } else {
          st3(O + 3 * a3, o + ctr);
          if (out_lds) st3(out_lds + 3 * a3, o + ctr);
        }
      }
    }
    if (BWD) {
      // (the moments were written by the lanes of the ascending order, they are read in the descending one: same wave, LDS
      //  operations complete in order)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      M3 ym = zero3();
      V3 yv{0.f, 0.f, 0.f};
      for (int s = 0; s < m; ++s) {
        const int e = s0 + s;
        if (e < nd) {
          float* q = Sd + (long)(nd - 1 - e) * SW;
          ym = add(ym, lds_get9(q + 24));
          yv = yv + ld3(Gf + 3 * g.atom(dir, nd - 1 - e + 3));
          lds_put9(q + 24, ym);
          st3(q + 21, yv);
        }
      }
      M3 em;
      V3 ev;
      sum_scan_exclusive(ym, yv, em, ev);
      for (int s = 0; s < m; ++s) {
        const int e = s0 + s;
        if (e < nd) {
          float* q = Sd + (long)(nd - 1 - e) * SW;
          const M3 Ym = add(em, lds_get9(q + 24));
          const V3 tb = ev + ld3(q + 21);
          const V3 tj = ld3(q + TO);
          M3 C = Ym;
          C.m[0][0] -= tb.x * tj.x; C.m[0][1] -= tb.x * tj.y; C.m[0][2] -= tb.x * tj.z;
          C.m[1][0] -= tb.y * tj.x; C.m[1][1] -= tb.y * tj.y; C.m[1][2] -= tb.y * tj.z;
          C.m[2][0] -= tb.z * tj.x; C.m[2][1] -= tb.z * tj.y; C.m[2][2] -= tb.z * tj.z;
          lds_put9(q + 24, mul(C, lds_get9(q + RO)));
          st3(q + 21, tb);
        }
      }
    }
  }
  DSTAMP(3);
  __syncthreads();
  DSTAMP(4);
  // ---- phase C
  if (!BWD) {
    if (wave == 0 && lane < 3 && lane < n) {
      const V3 p = ld3(P + 3 * g.atom(0, lane));
      st3(out_row + 3 * g.atom(0, lane), p);
      if (out_lds) st3(out_lds + 3 * g.atom(0, lane), p);
    }
    return;   // (the steps' atoms were written by phase B)
  }
  for (int i = slane; i < nd; i += LSTEP) {
    float* q = Sd + (long)i * SW;
    const M3 Q = lds_get9(q);
    const V3 c = ld3(q + 9);
    const M3 Ri = lds_get9(q + 12);
    const M3 Rp = i > 0 ? lds_get9(q - SW + 12) : identity();
    const M3 Rb = lds_get9(q + 24);
    const V3 tb = ld3(q + 21);
    const int a1 = g.atom(dir, i + 1), a2 = g.atom(dir, i + 2), a3 = g.atom(dir, i + 3);
    const V3 gk = ld3(Gf + 3 * a3);
    const V3 p1 = ld3(Pf + 3 * a1), o = ld3(Pf + 3 * a2);
    const V3 w = o - p1;
    const float inv = rsqrtf(dot(w, w));
    const V3 u = inv * w;
    float sn, cs;
    emx::sincos_bounded(dd[g.dih(dir, i)] + offset, &sn, &cs);
    M3 Qb = mulAT(Rp, Rb);           // from R_i = Rp Q
    const V3 cb = mulT(Rp, tb);      // from t_i = Rp c + tp
    V3 ob = cb - mulT(Q, cb);        // c = o - Q o
    Qb.m[0][0] -= cb.x * o.x; Qb.m[0][1] -= cb.x * o.y; Qb.m[0][2] -= cb.x * o.z;
    Qb.m[1][0] -= cb.y * o.x; Qb.m[1][1] -= cb.y * o.y; Qb.m[1][2] -= cb.y * o.z;
    Qb.m[2][0] -= cb.z * o.x; Qb.m[2][1] -= cb.z * o.y; Qb.m[2][2] -= cb.z * o.z;
    // Q(u, d) = cs I + sn [u]x + (1 - cs) u u^T
    const float trQ = Qb.m[0][0] + Qb.m[1][1] + Qb.m[2][2];
    const V3 ax{Qb.m[2][1] - Qb.m[1][2], Qb.m[0][2] - Qb.m[2][0], Qb.m[1][0] - Qb.m[0][1]};   // <Qb, [v]x> = v . ax
    const V3 Qu = mul(Qb, u), QTu = mulT(Qb, u);
    const float uQu = dot(u, Qu);
    ddih_row[g.dih(dir, i)] = -sn * trQ + cs * dot(u, ax) + sn * uQu;
    const V3 ub = sn * ax + (1.f - cs) * (Qu + QTu);
    const V3 wb = inv * (ub - dot(u, ub) * u);   // u = w / |w|
    ob = ob + wb;
    (void)c;
    // the step's contributions to d plane, in the slot of Q: [to atom i+3 | to atom i+2 | to atom i+1]
    st3(q, mulT(Ri, gk));
    st3(q + 3, ob);
    st3(q + 6, V3{-wb.x, -wb.y, -wb.z});
  }
  DSTAMP(5);
  __syncthreads();
  // ---- phase D: every atom collects what it receives: pass-through (atoms s-1 .. s+1), then right part, then left part
  float* DP = dplane_row;
  const int nd1 = max(g.natoms(1) - 3, 0);
  for (int a = tid; a < A; a += FR_NT) {
    V3 acc{0.f, 0.f, 0.f};
    if (a >= g.s - 1 && a <= g.s + 1) acc = ld3(Gf + 3 * a);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int cidx = d ? g.s + 1 - a : a - (g.s - 1);   // local chain index of the atom in this direction
      const int ndd = d ? nd1 : nd0;
      const float* Sx = S + (long)(d ? nd0 : 0) * SW;
      if (cidx < 0) continue;
#pragma unroll
      for (int r = 3; r >= 1; --r) {   // as step (cidx - r)'s contribution number (3 - r)
        const int i = cidx - r;
        if (i >= 0 && i < ndd) acc = acc + ld3(Sx + (long)i * SW + 3 * (3 - r));
      }
    }
    st3(DP + 3 * a, acc);
  }
  DSTAMP(6);
}

template <bool BWD>
__global__ __launch_bounds__(BWD ? FR_NT_BWD : FR_NT_FWD) void d2c_frame_kernel(const float* __restrict__ dih, int ldd, float offset, const float* __restrict__ plane,
                                                          const float* __restrict__ dout, int A, float* __restrict__ out,
                                                          float* __restrict__ ddih, int ldg, float* __restrict__ dplane) {
  extern __shared__ float sm[];
  const long b = blockIdx.x;
  d2c_frame_body<BWD, BWD ? FR_NT_BWD : FR_NT_FWD>(dih + b * ldd, offset, plane + b * A * 3, BWD ? dout + b * A * 3 : nullptr, A,
                                                    BWD ? nullptr : out + b * A * 3, nullptr, BWD ? ddih + b * ldg : nullptr,
                                                    BWD ? dplane + b * A * 3 : nullptr, sm);
}

// K6, one wave per frame: the angle recurrence is one subtraction per step (serial, LDS), sin/cos and the partial sums of
// the steps run over the lanes (a wave-wide inclusive scan per 64 steps).
// (the body, run by ONE wave; th: the frame's angles, o: its plane row in HBM, ol: an LDS copy of it or null)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void chain_frame_body(const float* __restrict__ len, const float* __restrict__ th, int A, float* __restrict__ sm,
                                                 float* __restrict__ o, float* __restrict__ ol, const int lane) {
  for (int i = lane; i < A - 2; i += 64) sm[i + 1] = PI_F - th[i];
  wave_sync();
  // phi_i = c_i - phi_{i-1}, phi_0 = 0  <=>  (-1)^i phi_i = sum_{k = 1..i} (-1)^k c_k: a prefix sum across the lanes, 64 steps
  // at a time with a carry, instead of A - 2 dependent LDS round trips on one lane
  {
    float carry = 0.f;
    for (int i0 = 1; i0 < A - 1; i0 += 64) {
      const int i = i0 + lane;
      float v = i < A - 1 ? ((i & 1) ? -sm[i] : sm[i]) : 0.f;
      v = wave_scan_add(v);   // (DPP: six dependent shuffles through LDS per scan were most of this kernel)
      const float t = carry + v;
      if (i < A - 1) sm[i] = (i & 1) ? -t : t;
      carry += lane63(v);
    }
    if (lane == 0) sm[0] = 0.f;
  }
  wave_sync();
  if (lane == 0) {
    st3(o, V3{0.f, 0.f, 0.f});
    if (ol) st3(ol, V3{0.f, 0.f, 0.f});
  }
  float cx = 0.f, cy = 0.f;   // carry of the partial sums
  for (int i0 = 0; i0 < A - 1; i0 += 64) {
    const int i = i0 + lane;
    float dx = 0.f, dy = 0.f;
    if (i < A - 1) {
      float sn, cs;
      emx::sincos_bounded(sm[i], &sn, &cs);
      dx = len[i] * cs;
      dy = (i & 1) ? -len[i] * sn : len[i] * sn;
    }
    dx = wave_scan_add(dx);
    dy = wave_scan_add(dy);
    if (i < A - 1) {
      st3(o + 3 * (i + 1), V3{cx + dx, cy + dy, 0.f});
      if (ol) st3(ol + 3 * (i + 1), V3{cx + dx, cy + dy, 0.f});
    }
    cx += lane63(dx);
    cy += lane63(dy);
  }
}
__global__ __launch_bounds__(64) void chain_frame_kernel(const float* __restrict__ len, const float* __restrict__ ang, int lda, int A,
                                                         float* __restrict__ xyz) {
  extern __shared__ float sm[];   // [A] phi
  const int b = blockIdx.x;
  chain_frame_body(len, ang + (long)b * lda, A, sm, xyz + (long)b * A * 3, nullptr, threadIdx.x);
}

// chain_in_plane backward, ONE wave per frame: phi as in the forward body; GX, GY (suffix sums of the coordinate gradients) and
// the alternating suffix sum of dL/dphi as wave scans over the steps in descending order, 64 at a time with a carry -- the
// thread-per-frame kernel above walks A - 2 dependent steps with a library sincosf each on 256 threads of the whole chip.
//   dL/dphi_j = len_j (-sin phi_j GX_j + (-1)^j cos phi_j GY_j);   (-1)^j acc_j = sum_{k >= j} (-1)^k dL/dphi_k;   dtheta_{j-1} = -acc_j
__device__ __forceinline__ void chain_bwd_body(const float* __restrict__ len, const float* __restrict__ th, const float* __restrict__ g, int A,
                                               float* __restrict__ sm, float* __restrict__ d, const int lane) {
  for (int i = lane; i < A - 2; i += 64) sm[i + 1] = PI_F - th[i];
  wave_sync();
  {
    float carry = 0.f;
    for (int i0 = 1; i0 < A - 1; i0 += 64) {
      const int i = i0 + lane;
      float v = i < A - 1 ? ((i & 1) ? -sm[i] : sm[i]) : 0.f;
      v = wave_scan_add(v);
      const float t = carry + v;
      if (i < A - 1) sm[i] = (i & 1) ? -t : t;
      carry += lane63(v);
    }
    if (lane == 0) sm[0] = 0.f;
  }
  wave_sync();
  const int n = A - 1;
  float cX = 0.f, cY = 0.f, cS = 0.f;
  for (int e0 = 0; e0 < n; e0 += 64) {
    const int e = e0 + lane, j = n - 1 - e;
    const bool ok = e < n;
    float gx = ok ? g[3 * (j + 1)] : 0.f, gy = ok ? g[3 * (j + 1) + 1] : 0.f;
    gx = wave_scan_add(gx);
    gy = wave_scan_add(gy);
    float alt = 0.f;
    if (ok) {
      float sn, cs;
      emx::sincos_bounded(sm[j], &sn, &cs);
      const float sgn = (j & 1) ? -1.f : 1.f;
      const float dphi = len[j] * (-sn * (cX + gx) + sgn * cs * (cY + gy));
      alt = sgn * dphi;
    }
    const float S = cS + wave_scan_add(alt);
    if (ok && j >= 1) d[j - 1] = (j & 1) ? S : -S;   // -acc_j, acc_j = (-1)^j S
    cX += lane63(gx);
    cY += lane63(gy);
    cS = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S), 63));
  }
}
__global__ __launch_bounds__(64) void chain_bwd_frame_kernel(const float* __restrict__ len, const float* __restrict__ ang, int lda,
                                                             const float* __restrict__ dxyz, int A, float* __restrict__ dang, int ldd) {
  extern __shared__ float sm[];   // [A] phi
  const long b = blockIdx.x;
  chain_bwd_body(len, ang + b * lda, dxyz + b * A * 3, A, sm, dang + b * ldd, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// K3: per-frame flat pair distances of the selected atoms (start + step * c, c < C), strict upper triangle, row-major
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_decode(long p, int C, int& i, int& j) {
  // p = i*C - i(i+1)/2 + (j - i - 1)
  float fc = (float)C - 0.5f;
  int ii = (int)(fc - sqrtf(fmaxf(fc * fc - 2.f * (float)p, 0.f)));
  ii = max(0, min(ii, C - 2));
  while (ii > 0 && (long)ii * C - (long)ii * (ii + 1) / 2 > p) --ii;
  while ((long)(ii + 1) * C - (long)(ii + 1) * (ii + 2) / 2 <= p) ++ii;
  i = ii;
  j = (int)(p - ((long)ii * C - (long)ii * (ii + 1) / 2)) + ii + 1;
}

// Pairs without a decode: row i of the strict upper triangle (C - 1 - i pairs) and row C - 2 - i (i + 1 pairs) together are C
// slots, so slot s of the ceil((C - 1) / 2) x C rectangle is  f = s / C, c = s % C  ->  (i, j) = (f, f + 1 + c) for c < C - 1 - f,
// else (C - 2 - f, C - 1 - f + c - (C - 1 - f) ...) of the mirrored row; the middle row (C even) has only its first part.  One
// float multiply and a correction instead of a square root and two divergent search loops per pair.
__device__ __forceinline__ bool pair_slot(int s, int C, float rC, int& i, int& j) {
  int f = (int)(((float)s + 0.5f) * rC);
  int c = s - f * C;
  if (c < 0) {
    --f;
    c += C;
  } else if (c >= C) {
    ++f;
    c -= C;
  }
  const int n0 = C - 1 - f;          // pairs of row f
  if (c < n0) {
    i = f;
    j = f + 1 + c;
    return true;
  }
  i = C - 2 - f;
  j = i + 1 + (c - n0);
  return i != f;                     // (the middle row is not its own mirror)
}
__device__ __forceinline__ int pair_slots(int C) { return ((C - 1 + 1) >> 1) * C; }
__device__ __forceinline__ int pair_index(int i, int j, int C) { return ((i * (2 * C - i - 1)) >> 1) + (j - i - 1); }

// thread t of nt: the pairs of the slots t, t + nt, ... of the atoms s0 + st c held in sx ([*][3])
__device__ __forceinline__ void pdist_pairs(const float* __restrict__ sx, int s0, int st, int C, long P, int ldo, float* __restrict__ o, int t, int nt) {
  if (P < (1L << 30)) {
    const float rC = 1.f / (float)C;
    const int ns = pair_slots(C);
    for (int s = t; s < ns; s += nt) {
      int i, j;
      if (!pair_slot(s, C, rC, i, j)) continue;
      const float* a = sx + 3 * (s0 + st * i);
      const float* c = sx + 3 * (s0 + st * j);
      const float dx = a[0] - c[0], dy = a[1] - c[1], dz = a[2] - c[2];
      o[pair_index(i, j, C)] = sqrtf(dx * dx + dy * dy + dz * dz);
    }
  } else {
    for (long p = t; p < P; p += nt) {
      int i, j;
      pair_decode(p, C, i, j);
      const float* a = sx + 3 * (s0 + st * i);
      const float* c = sx + 3 * (s0 + st * j);
      const float dx = a[0] - c[0], dy = a[1] - c[1], dz = a[2] - c[2];
      o[p] = sqrtf(dx * dx + dy * dy + dz * dz);
    }
  }
  for (long p = P + t; p < ldo; p += nt) o[p] = 0.f;   // padding columns (row stride rounded up) stay zero
}

constexpr int PD_NT = 1024;   // threads per frame: a 256-frame batch is 256 workgroups, one per CU -- each should bring all its waves
__global__ __launch_bounds__(PD_NT) void pdist_fwd_kernel(const float* __restrict__ xyz, int A, int start, int step, int C, long P, int ldo,
                                                       float* __restrict__ out) {
  extern __shared__ float sx[];   // [C][3]
  const int b = blockIdx.x;
  const float* X = xyz + (long)b * A * 3;
  for (int e = threadIdx.x; e < 3 * C; e += PD_NT) {
    int c = e / 3, k = e - 3 * c;
    sx[e] = X[3 * (start + step * c) + k];
  }
  __syncthreads();
  pdist_pairs(sx, 0, 1, C, P, ldo, out + (long)b * ldo, threadIdx.x, PD_NT);
}

// T threads per selected atom (T a power of two <= 8, consecutive lanes) gather its C-1 pair terms, j = q, q + T, ..., and add
// their shares in a fixed tree (deterministic, no atomics); the frame's gradient row is staged in LDS first when it fits
// (stage_g) -- read in place, every thread walks its dependent scattered global loads.  (One thread per atom left 76 of 256
// threads with 75 serial terms each at the ADC shape.)
// (the part behind the staging: sx the selected atoms, g the frame's gradient row, D the frame's [A][3] result -- HBM or LDS)
__device__ __forceinline__ void pdist_bwd_body(const float* __restrict__ sx, const float* __restrict__ g, int A, int start, int step, int C,
                                               float* __restrict__ D, int accumulate, int T) {
  if (!accumulate) {
    // atoms that are not selected receive zero
    for (int e = threadIdx.x; e < 3 * A; e += PD_NT) {
      int a = e / 3;
      bool sel = a >= start && (a - start) % step == 0 && (a - start) / step < C;
      if (!sel) D[e] = 0.f;
    }
  }
  const int q = threadIdx.x & (T - 1), per_pass = PD_NT / T;
  for (int i0 = 0; i0 < C; i0 += per_pass) {          // uniform trip count: every lane takes part in the shuffles
    const int i = i0 + threadIdx.x / T;
    const bool ok = i < C;
    const int ic = ok ? i : C - 1;
    const float xi = sx[3 * ic], yi = sx[3 * ic + 1], zi = sx[3 * ic + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int j = q; j < C; j += T) {
      if (j == ic) continue;
      const int lo = min(ic, j), hi = max(ic, j);
      const long p = (long)lo * C - (long)lo * (lo + 1) / 2 + (hi - lo - 1);
      const float dx = xi - sx[3 * j], dy = yi - sx[3 * j + 1], dz = zi - sx[3 * j + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      const float w = d2 > 0.f ? g[p] * rsqrtf(d2) : 0.f;   // reference mask: zero gradient on coincident atoms
      gx = fmaf(w, dx, gx);
      gy = fmaf(w, dy, gy);
      gz = fmaf(w, dz, gz);
    }
    for (int o = 1; o < T; o <<= 1) {
      gx += __shfl_xor(gx, o, 64);
      gy += __shfl_xor(gy, o, 64);
      gz += __shfl_xor(gz, o, 64);
    }
    if (ok && q == 0) {
      float* qd = D + 3 * (start + step * i);
      if (accumulate) {
        qd[0] += gx; qd[1] += gy; qd[2] += gz;
      } else {
        qd[0] = gx; qd[1] = gy; qd[2] = gz;
      }
    }
  }
}
__global__ __launch_bounds__(PD_NT) void pdist_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ dout, int A, int start,
                                                          int step, int C, int ldo, float* __restrict__ dxyz, int accumulate, int stage_g,
                                                          int T) {
  extern __shared__ float sx[];
  const int b = blockIdx.x;
  const float* X = xyz + (long)b * A * 3;
  for (int e = threadIdx.x; e < 3 * C; e += PD_NT) {
    int c = e / 3, k = e - 3 * c;
    sx[e] = X[3 * (start + step * c) + k];
  }
  const float* g = dout + (long)b * ldo;
  if (stage_g) {
    float* sg = sx + 3 * C;
    const int P = C * (C - 1) / 2;
    for (int e = threadIdx.x; e < P; e += PD_NT) sg[e] = g[e];
    g = sg;
  }
  __syncthreads();
  pdist_bwd_body(sx, g, A, start, step, C, dxyz + (long)b * A * 3, accumulate, T);
}

// ---------------------------------------------------------------------------------------------------------
// The forward half of the ADC model's back-mapping as ONE launch, a workgroup per frame (emx_adc_frame_fwd): PeriodicOutput of
// the angle / dihedral blocks, chain_in_plane, dihedrals_to_cartesian and the pair distances of the back-mapped AND of the input
// frame -- five launches of 4-6 us each at a 256-frame batch, every one of them a chain of dependent trips to memory for a few
// kilobytes per frame.  Here the frame stays in LDS between the stages (HBM receives what the backward pass and the costs read
// later); the input frame's pair distances do not depend on the model and run beside the serial chain construction.
// The stages are the bodies of the kernels above: same arithmetic, same results.
// ---------------------------------------------------------------------------------------------------------
constexpr int AF_NT = 1024;
struct AdcFrameArgs {
  const float* Y;          // [B, ldy] decoder output
  int ldy;
  float out_scale;         // periodicity / 2 pi
  int n_ang, ang_sin, ang_cos;   // angle block of Y (ang_sin < 0: the angles are the batch means in mean_ang)
  const float* mean_ang;
  float* ang_pred;         // [B, A - 2] (written unless the means are used)
  int dih_sin, dih_cos;
  float* dih_pred;         // [B, A - 3]
  const float* mean_len;   // [A - 1]
  float offset;
  int A;
  float* plane;            // [B, A, 3]
  float* xyz;              // [B, A, 3]
  const float* cart;       // [B, A, 3] input frames
  int start, step, C, ldo;
  float* inp_pair;         // [B, ldo] (null: not wanted)
  float* out_pair;         // [B, ldo]
  int split_wide;          // the rows' norms in the order of emx_sketchmap_split's long-row kernel (else: of its wave-per-row kernel)
  emx::SplitRows sp;       // planes null: no split.  Else emx_sketchmap_split of inp_pair: its rows minus row 0 as two f16 planes
};
__host__ __device__ inline long adc_frame_floats(int A, int C, bool split) {
  return (long)A + 2 * (long)A + 6 * (long)A + 3 * (long)C + d2c_frame_floats(A, false) +
         (split ? 3 * (long)C + emx::split_ld((int)((long)C * (C - 1) / 2)) : 0);
}

__device__ __forceinline__ void adc_frame_fwd_body(const AdcFrameArgs& a, float* sm) {
  const int A = a.A, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long b = blockIdx.x;
  float* phi = sm;                 // [A]
  float* ang_l = phi + A;          // [A] angles of the frame
  float* dih_l = ang_l + A;        // [A] dihedrals
  float* Pl = dih_l + A;           // [A][3] plane
  float* Xl = Pl + 3 * A;          // [A][3] back-mapped frame
  float* sxc = Xl + 3 * A;         // [C][3] selected atoms of the input frame
  float* fr = sxc + 3 * a.C;       // the frame kernel's own buffers
  const bool split = a.sp.planes != nullptr;
  float* sxr = fr + d2c_frame_floats(A, false);   // [C][3] selected atoms of input frame 0, the reference row of the split
  float* vrow = sxr + 3 * a.C;                    // [P] this frame's input pair distances minus frame 0's
  __shared__ float red_n[AF_NT / 64], red_m[AF_NT / 64];
  const float* row = a.Y + b * a.ldy;
  const long P = (long)a.C * (a.C - 1) / 2;
  AFSTAMP(0);
  // ---- PeriodicOutput; the input frame's selected atoms
  for (int k = tid; k < A - 2; k += AF_NT) {
    float v;
    if (a.ang_sin >= 0) {
      v = atan2f(row[a.ang_sin + k], row[a.ang_cos + k]) * a.out_scale;
      a.ang_pred[b * (A - 2) + k] = v;
    } else {
      v = a.mean_ang[k];
    }
    ang_l[k] = v;
  }
  for (int k = tid; k < A - 3; k += AF_NT) {
    const float v = atan2f(row[a.dih_sin + k], row[a.dih_cos + k]) * a.out_scale;
    a.dih_pred[b * (A - 3) + k] = v;
    dih_l[k] = v;
  }
  if (a.inp_pair) {
    const float* X = a.cart + b * A * 3;
    for (int e = tid; e < 3 * a.C; e += AF_NT) {
      const int c = e / 3, k = e - 3 * c;
      sxc[e] = X[3 * (a.start + a.step * c) + k];
      if (split) sxr[e] = a.cart[3 * (a.start + a.step * c) + k];
    }
  }
  __syncthreads();
  AFSTAMP(1);
  // ---- chain_in_plane on wave 0; the other waves: pair distances of the input frame
  if (wave == 0) {
    chain_frame_body(a.mean_len, ang_l, A, phi, a.plane + b * A * 3, Pl, lane);
  } else if (a.inp_pair && !split) {
    pdist_pairs(sxc, 0, 1, a.C, P, a.ldo, a.inp_pair + b * a.ldo, tid - 64, AF_NT - 64);
  } else if (a.inp_pair) {
    // (pdist_pairs, and beside every distance the same pair of frame 0: the split works on the difference)
    float* o = a.inp_pair + b * a.ldo;
    const float rC = 1.f / (float)a.C;
    const int ns = pair_slots(a.C);
    for (int sl = tid - 64; sl < ns; sl += AF_NT - 64) {
      int i, j;
      if (!pair_slot(sl, a.C, rC, i, j)) continue;
      const int p = pair_index(i, j, a.C);
      const float* u = sxc + 3 * i;
      const float* w = sxc + 3 * j;
      const float dx = u[0] - w[0], dy = u[1] - w[1], dz = u[2] - w[2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      const float* ur = sxr + 3 * i;
      const float* wr = sxr + 3 * j;
      const float ex = ur[0] - wr[0], ey = ur[1] - wr[1], ez = ur[2] - wr[2];
      o[p] = d;
      vrow[p] = d - sqrtf(ex * ex + ey * ey + ez * ez);
    }
    for (long p = P + tid - 64; p < a.ldo; p += AF_NT - 64) o[p] = 0.f;
  }
  AFSTAMP(2);
  __syncthreads();
  AFSTAMP(3);
  // ---- dihedrals_to_cartesian
  d2c_frame_body<false, AF_NT>(dih_l, a.offset, Pl, nullptr, A, a.xyz + b * A * 3, Xl, nullptr, nullptr, fr);
  __syncthreads();
  AFSTAMP(4);
  // ---- pair distances of the back-mapped frame
  pdist_pairs(Xl, a.start, a.step, a.C, P, a.ldo, a.out_pair + b * a.ldo, tid, AF_NT);
  AFSTAMP(5);
  if (!split) return;
  // ---- sk_split_wide_kernel on the row of differences: its partition of the row over 1024 threads, its order of additions
  //      (a.split_wide; short rows: the order of emx::split_row, a wave per row)
  float nrm = 0.f, mx = 0.f;
  const int kstep = a.split_wide ? AF_NT : 64;
  if (a.split_wide || wave == 0) {
    for (int k = a.split_wide ? tid : lane; k < (int)P; k += kstep) {
      const float v = vrow[k];
      nrm = fmaf(v, v, nrm);
      mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    nrm = emx::wave_sum(nrm);
    if (lane == 0) {
      red_n[wave] = nrm;
      red_m[wave] = mx;
    }
  }
  __syncthreads();
  if (a.split_wide) {
    nrm = 0.f;
    mx = 0.f;
#pragma unroll
    for (int w = 0; w < AF_NT / 64; w += 4) {   // fixed order
      nrm += (red_n[w] + red_n[w + 1]) + (red_n[w + 2] + red_n[w + 3]);
      mx = fmaxf(mx, fmaxf(fmaxf(red_m[w], red_m[w + 1]), fmaxf(red_m[w + 2], red_m[w + 3])));
    }
  } else {
    nrm = red_n[0];
    mx = red_m[0];
  }
  int E = (int)((__float_as_uint(mx) >> 23) & 0xff) - 126;   // (the scale rule of emx::split_row)
  E = mx > 0.f ? min(max(E, -100), 100) : 14;
  const float sc = __uint_as_float((uint32_t)(127 + 14 - E) << 23), rs = __uint_as_float((uint32_t)(127 - 14 + E) << 23);
  for (int k = 2 * tid; k < a.sp.ld; k += 2 * AF_NT) {   // (ld is a multiple of 32: a pair never straddles a chunk)
    const float v0 = k < (int)P ? vrow[k] * sc : 0.f, v1 = k + 1 < (int)P ? vrow[k + 1] * sc : 0.f;
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
    uint16_t* o = a.sp.planes + emx::split_index(a.sp, b, k);
    *reinterpret_cast<uint32_t*>(o) = (uint32_t)emx::f16_bits(h0) | ((uint32_t)emx::f16_bits(h1) << 16);
    *reinterpret_cast<uint32_t*>(o + a.sp.plane_stride) = (uint32_t)emx::f16_bits(l0) | ((uint32_t)emx::f16_bits(l1) << 16);
  }
  if (tid == 0) {
    a.sp.norms[b] = nrm;
    a.sp.rscale[b] = rs;
  }
  AFSTAMP(6);
}

// ---------------------------------------------------------------------------------------------------------
// ... and the way back (emx_adc_frame_bwd): the cartesian cost on the frame's pair distances (emx_periodic_cost_fwd_bwd on
// out_pair against inp_pair), the pair distances' backward pass and the backward pass of dihedrals_to_cartesian, a workgroup
// per frame -- three launches of 6-10 us.  d cost / d pair distances and d cost / d xyz never leave LDS.  The cost's workgroup
// shares land in ws exactly as the stand-alone launch leaves them (its partition of a row: S = 4 workgroups of 256 threads,
// here the four quarters of this workgroup), so that the step's scalar epilogue adds up the same numbers in the same order.
// ---------------------------------------------------------------------------------------------------------
struct AdcFrameBwdArgs {
  emx::CostArgs cost;      // y = out_pair, target = inp_pair, n = P; dY unused (the gradient row stays in LDS)
  const float* xyz;        // [B, A, 3] back-mapped frames
  const float* dih;        // [B, A - 3]
  const float* plane;      // [B, A, 3]
  float offset;
  int A, start, step, C, T;
  float* ddih;             // [B, ldg]
  int ldg;
  float* dplane;           // [B, A, 3]
  // optional: up to four row-mean costs on row b of other blocks (emx_periodic_cost_batch: the angle-like costs of the decoder output;
  // one of them adds d cost / d dihedrals of this launch to its gradient), each the stand-alone launch's one-workgroup-per-row
  // form -- 256 threads, here quarter j of the workgroup
  emx::CostArgs job[EMX_MAX_COST_JOBS];
  int n_jobs;
  // optional (models with backbone angles): chain_in_plane's backward pass on d cost / d plane of this launch, in front of the jobs
  const float* len;        // [A - 1] mean bond lengths
  const float* ang;        // [B, A - 2] angles (null: skipped)
  float* dang;             // [B, ldd]
  int ldd;
};
__host__ __device__ inline long adc_frame_bwd_floats(int A, int C) {
  return 3 * (long)C + (long)C * (C - 1) / 2 + 3 * (long)A + d2c_frame_floats(A, true) + A;
}
static_assert(AF_NT == PD_NT, "the pair-distance bodies run on the whole workgroup");

__device__ __forceinline__ void adc_frame_bwd_body(const AdcFrameBwdArgs& a, float* sm, float* red) {
  const int A = a.A, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long b = blockIdx.x;
  const int P = a.C * (a.C - 1) / 2;
  float* sx = sm;                 // [C][3] selected atoms of the back-mapped frame
  float* sg = sx + 3 * a.C;       // [P] d cost / d pair distances
  float* dx = sg + P;             // [A][3] d cost / d xyz
  float* fr = dx + 3 * A;         // the frame kernel's buffers
  AFSTAMP(16);
  {
    const float* X = a.xyz + b * A * 3;
    for (int e = tid; e < 3 * a.C; e += AF_NT) {
      const int c = e / 3, k = e - 3 * c;
      sx[e] = X[3 * (a.start + a.step * c) + k];
    }
  }
  // ---- the cost on row b (periodic_cost_body with four workgroups per row: quarter q of this workgroup is its workgroup 4 b + q)
  {
    const emx::CostArgs& c = a.cost;
    const float* yrow = c.y + b * c.ldy;
    const float* trow = c.target + b * c.ldt;
    const float gn = c.coef_scale ? c.gnorm * *c.coef_scale : c.gnorm;
    float lsum = 0.f;
    for (int k = tid; k < c.n; k += AF_NT) {
      float s, cs, delta, dd;
      emx::cost_elem(c, yrow, trow, k, s, cs, delta, dd);
      float g;
      if (c.variant == EMX_MEAN_ABS) {
        lsum += fabsf(delta);
        g = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
      } else {
        lsum = fmaf(delta, delta, lsum);
        g = 2.f * delta * dd;
      }
      emx::cost_store(c, sg, nullptr, k, gn * g, s, cs);
    }
    lsum = emx::wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();   // (also: sx and sg are complete)
    if ((tid & 255) == 0) {
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) r += red[wave + i];
      c.ws[4 * b + (tid >> 8)] = r * c.gnorm * (c.coef_scale ? *c.coef_scale : 1.f);
    }
    if (b == 0)   // unused scratch entries read as zero by the summing job
      for (int i = 4 * (int)gridDim.x + tid; i < EMX_REDUCE_WS; i += AF_NT) c.ws[i] = 0.f;
  }
  AFSTAMP(17);
  // ---- pair distances, backward: d cost / d xyz of the frame
  pdist_bwd_body(sx, sg, A, a.start, a.step, a.C, dx, 0, a.T);
  AFSTAMP(18);
  __syncthreads();
  AFSTAMP(19);
  // ---- dihedrals_to_cartesian, backward
  d2c_frame_body<true, AF_NT>(a.dih + b * (A - 3), a.offset, a.plane + b * A * 3, dx, A, nullptr, nullptr, a.ddih + b * a.ldg, a.dplane + b * A * 3, fr);
  AFSTAMP(20);
  if (a.ang != nullptr) {
    // ---- chain_in_plane, backward (one wave; d cost / d plane of this row was written above)
    __syncthreads();
    if (wave == 0) chain_bwd_body(a.len, a.ang + b * (A - 2), a.dplane + b * A * 3, A, fr + d2c_frame_floats(A, true), a.dang + b * a.ldd, lane);
  }
  if (a.n_jobs == 0) return;
  // ---- the row-mean costs of the batch launch on row b: periodic_cost_body with one workgroup of 256 threads per row, quarter
  //      by quarter (d cost / d dihedrals of this row was written above: visible behind the barrier)
  __syncthreads();
  const int q = wave >> 2, t256 = tid & 255;
  float lsum = 0.f;
  auto run = [&](const emx::CostArgs& c) __attribute__((always_inline)) {
    const float* yrow = c.y + b * c.ldy;
    const float* trow = c.target + b * c.ldt;
    float* drow = c.dY ? c.dY + b * c.ldy : nullptr;
    const float* addrow = c.dpred_add ? c.dpred_add + b * c.n : nullptr;
    const float gn = c.coef_scale ? c.gnorm * *c.coef_scale : c.gnorm;
    for (int k = t256; k < c.n; k += 256) {
      float sn, cs, delta, dd;
      emx::cost_elem(c, yrow, trow, k, sn, cs, delta, dd);
      float g;
      if (c.variant == EMX_MEAN_ABS) {
        lsum += fabsf(delta);
        g = (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f)) * dd;
      } else {
        lsum = fmaf(delta, delta, lsum);
        g = 2.f * delta * dd;
      }
      emx::cost_store(c, drow, addrow, k, gn * g, sn, cs);
    }
  };
  if (q == 0) run(a.job[0]);
  else if (q == 1 && a.n_jobs > 1) run(a.job[1]);
  else if (q == 2 && a.n_jobs > 2) run(a.job[2]);
  else if (q == 3 && a.n_jobs > 3) run(a.job[3]);
  lsum = emx::wave_sum(lsum);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  auto finish = [&](const emx::CostArgs& c) __attribute__((always_inline)) {
    if (t256 == 0) {
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) r += red[wave + i];
      c.ws[b] = r * c.gnorm * (c.coef_scale ? *c.coef_scale : 1.f);
    }
    if (b == 0)
      for (int i = (int)gridDim.x + t256; i < EMX_REDUCE_WS; i += 256) c.ws[i] = 0.f;
  };
  if (q == 0) finish(a.job[0]);
  else if (q == 1 && a.n_jobs > 1) finish(a.job[1]);
  else if (q == 2 && a.n_jobs > 2) finish(a.job[2]);
  else if (q == 3 && a.n_jobs > 3) finish(a.job[3]);
}

__global__ __launch_bounds__(AF_NT) void adc_frame_fwd_kernel(const AdcFrameArgs a) {
  extern __shared__ float sm[];
  adc_frame_fwd_body(a, sm);
}
__global__ __launch_bounds__(AF_NT) void adc_frame_bwd_kernel(const AdcFrameBwdArgs a) {
  extern __shared__ float sm[];
  __shared__ float red[AF_NT / 64];
  adc_frame_bwd_body(a, sm, red);
}
// Both halves in ONE launch (emx_adc_frame_fwd_bwd): nothing between them depends on another frame -- the workgroup that has
// back-mapped frame b runs the cartesian cost and the way back for frame b from what it has just written (drained, then a barrier;
// the lines were never in this CU's L1), and the second launch's ramp and first trips go.
__global__ __launch_bounds__(AF_NT) void adc_frame_both_kernel(const AdcFrameArgs f, const AdcFrameBwdArgs g) {
  extern __shared__ float sm[];
  __shared__ float red[AF_NT / 64];
  adc_frame_fwd_body(f, sm);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  adc_frame_bwd_body(g, sm, red);
}

// ---------------------------------------------------------------------------------------------------------
// Featurisation of a backbone chain ("next" row 2): bond lengths, bond angles and dihedrals of
// consecutive atoms, the quantities the reference obtains from mdtraj's C kernels (encodermap/loading/features.py:
// 150-156, 1133, 1394) and feeds to the ADC model.  Formulas: lengths encodermap_tf1/moldata.py:270-271, dihedral
// atan2(|b2| b1.(b2 x b3), (b1 x b2).(b2 x b3)) encodermap/misc/backmapping.py:344-352.
// One thread per (frame, atom): coalesced across atoms within a frame.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ __launch_bounds__(NT) void featurize_kernel(const float* __restrict__ xyz, long N, int A, float* __restrict__ dist,
                                                       float* __restrict__ ang, float* __restrict__ dih) {
  const long e = (long)blockIdx.x * NT + threadIdx.x;
  if (e >= N * A) return;
  const long f = e / A;
  const int i = (int)(e - f * A);
  const float* P = xyz + f * A * 3;
  if (i + 1 < A) {
    const V3 p0 = ld3(P + 3 * i), p1 = ld3(P + 3 * (i + 1));
    const V3 b1 = p1 - p0;
    if (dist) dist[f * (A - 1) + i] = sqrtf(dot(b1, b1));
    if (i + 2 < A) {
      const V3 p2 = ld3(P + 3 * (i + 2));
      const V3 b2 = p2 - p1;
      if (ang) {
        const V3 u = p0 - p1;
        const float c = dot(u, b2) * rsqrtf(dot(u, u) * dot(b2, b2));
        ang[f * (A - 2) + i] = acosf(fminf(fmaxf(c, -1.f), 1.f));
      }
      if (i + 3 < A && dih) {
        const V3 p3 = ld3(P + 3 * (i + 3));
        const V3 b3 = p3 - p2;
        const V3 c1 = cross(b2, b3), c2 = cross(b1, b2);
        dih[f * (A - 3) + i] = atan2f(dot(b1, c1) * sqrtf(dot(b2, b2)), dot(c1, c2));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// All-atom back-mapping: set every torsion of every frame to its target by rotating the
// atoms on the far side of the torsion's central bond -- the serial loops of encodermap/misc/backmapping.py:1668-1690
// (central dihedrals) and :1722-1745 (side chains), whose numba variant is `parallel_rotation_application` :383-405.
// The torsions of one frame depend on each other (each rotation moves the atoms later torsions are measured on), the
// frames do not: ONE WAVE PER FRAME, the frame in LDS, torsion after torsion.  The far sides arrive as a lane-major bit
// table, far_mask[j][w][lane] bit k <-> atom (32 w + k) 64 + lane: one coalesced 256-byte load per torsion and wave
// (requested one torsion ahead), no index lists, and lane l only ever touches atoms l, l + 64, ... of the LDS frame.
//   current = atan2(|b2| b1.(b2 x b3), (b1 x b2).(b2 x b3))                      misc/rotate.py:547-581 (its sin / cos only)
//   p' = pivot + R (p - pivot),  R = cos a I + (1 - cos a) d d^T + sin a [d]_x   transformations.rotation_matrix, as
//   called at backmapping.py:1683-1684 with a = target - current, d = unit(x[q2] - x[q1]), pivot = x[q1]
// ---------------------------------------------------------------------------------------------------------
constexpr int AA_NT = 64;

__global__ __launch_bounds__(AA_NT) void allatom_rotate_kernel(const float* __restrict__ tmpl, long tmpl_stride, int A,
                                                               const int* __restrict__ quads, const unsigned* __restrict__ far_mask, int W,
                                                               int J, const float* __restrict__ targets, long ldt, float* __restrict__ out) {
  extern __shared__ float fr[];   // [A][3]
  const int lane = threadIdx.x;
  const long f = blockIdx.x;
  const float* __restrict__ src = tmpl + f * tmpl_stride;
  for (int e = lane; e < 3 * A; e += AA_NT) fr[e] = src[e];
  __syncthreads();
  const float* __restrict__ tg = targets + f * ldt;
  const unsigned* __restrict__ mrow = far_mask + lane;
  __shared__ float tsin[AA_NT], tcos[AA_NT];   // sin / cos of the next 64 targets, one per lane
  unsigned m_next = J > 0 ? mrow[0] : 0u;
  for (int j = 0; j < J; ++j) {
    if ((j & (AA_NT - 1)) == 0) {
      float s_, c_;
      emx::sincos_bounded(tg[min(j + lane, J - 1)], &s_, &c_);
      tsin[lane] = s_;
      tcos[lane] = c_;
      __syncthreads();
    }
    const unsigned m0 = m_next;
    m_next = mrow[(long)min(j + 1, J - 1) * W * AA_NT];
    const float st = tsin[j & (AA_NT - 1)], ct = tcos[j & (AA_NT - 1)];
    const int q0 = quads[4 * j], q1 = quads[4 * j + 1], q2 = quads[4 * j + 2], q3 = quads[4 * j + 3];
    const V3 p0 = ld3(fr + 3 * q0), p1 = ld3(fr + 3 * q1), p2 = ld3(fr + 3 * q2), p3 = ld3(fr + 3 * q3);
    const V3 b1 = p1 - p0, b2 = p2 - p1, b3 = p3 - p2;
    const V3 c1 = cross(b2, b3), c2 = cross(b1, b2);
    const float l2 = dot(b2, b2);
    // current = atan2(y, x) is never formed: only sin / cos of (target - current) are needed, and
    // (cos, sin)(current) = (x, y) / hypot(x, y)   [atan2(0, 0) = 0]
    const float y = dot(b1, c1) * sqrtf(l2), x = dot(c1, c2);
    const float h2 = fmaf(x, x, y * y);
    const float rh = rsqrtf(h2);
    const float cc = h2 > 0.f ? x * rh : 1.f, sc = h2 > 0.f ? y * rh : 0.f;
    const float sa = fmaf(st, cc, -ct * sc), ca = fmaf(ct, cc, st * sc);
    const V3 d = rsqrtf(l2) * b2;
    const float oc = 1.f - ca;
    // rows of R
    const V3 r0{fmaf(oc * d.x, d.x, ca), fmaf(oc * d.x, d.y, -sa * d.z), fmaf(oc * d.x, d.z, sa * d.y)};
    const V3 r1{fmaf(oc * d.y, d.x, sa * d.z), fmaf(oc * d.y, d.y, ca), fmaf(oc * d.y, d.z, -sa * d.x)};
    const V3 r2{fmaf(oc * d.z, d.x, -sa * d.y), fmaf(oc * d.z, d.y, sa * d.x), fmaf(oc * d.z, d.z, ca)};
    for (int w = 0; w < W; ++w) {
      unsigned m = w == 0 ? m0 : mrow[((long)j * W + w) * AA_NT];
      const int base = w * 32 * AA_NT + lane;
      while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        float* a = fr + 3 * (base + k * AA_NT);
        const V3 v = ld3(a) - p1;
        st3(a, V3{p1.x + dot(r0, v), p1.y + dot(r1, v), p1.z + dot(r2, v)});
      }
    }
    __syncthreads();
  }
  float* __restrict__ dst = out + f * (long)A * 3;
  for (int e = lane; e < 3 * A; e += AA_NT) dst[e] = fr[e];
}

__global__ void soft_start_kernel(const int64_t* __restrict__ step, float a, float b, float c, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float sc = c;
  if (a >= 0.f) {
    const float t = (float)*step;
    if (t < a) sc = 0.f;
    else if (t <= b) sc = c / (b - a) * (t - a);
  }
  *out = sc;
}

// The head of an ADC step in one launch: the batch's feature rows and Cartesian rows by index, the sin/cos encoder input of
// the angular blocks, and the soft-start scale of this step (a workgroup per row; thread 0 of the launch does the scalar).
__global__ __launch_bounds__(NT) void adc_head_kernel(const emx::AdcHeadArgs a) {
  const int64_t* __restrict__ idx = a.idx + (a.cursor ? *a.cursor * (long)a.B : 0);
  if (blockIdx.x >= a.row_blocks) {
    // emx_colmean of the batch's feature rows, read through the index (the row copies of this launch are not waited for)
    __shared__ float part[emx::ADC_CM_PH][NT / emx::ADC_CM_PH];
    emx::adc_head_means<NT / emx::ADC_CM_PH>(a, idx, (blockIdx.x - a.row_blocks) * (NT / emx::ADC_CM_PH), threadIdx.x, part);
    return;
  }
  // a workgroup per row: a few hundred columns per row and only a few hundred rows per batch -- a wave per row walks ~35
  // dependent iterations behind its three hops (cursor, index, row) while most of the chip idles
  emx::adc_head_rows(a, idx, blockIdx.x, a.row_blocks, threadIdx.x);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Training metrics of the ADC model
// ---------------------------------------------------------------------------------------------------------
// clashes: number of atom pairs i < j of a frame closer than `thr`                     callbacks/metrics.py:505-522
//   (count_nonzero(pairwise_dist(y, flat=True) < d, axis=1)); one workgroup per frame, the frame in LDS
__global__ __launch_bounds__(NT) void clash_count_kernel(const float* __restrict__ xyz, int A, float thr2, long long* __restrict__ counts) {
  extern __shared__ float fr[];   // [A][3]
  __shared__ int red[NT / 64];
  const float* __restrict__ f = xyz + (long)blockIdx.x * A * 3;
  for (int e = threadIdx.x; e < 3 * A; e += NT) fr[e] = f[e];
  __syncthreads();
  int c = 0;
  for (int i = 0; i < A - 1; ++i) {
    const float xi = fr[3 * i], yi = fr[3 * i + 1], zi = fr[3 * i + 2];
    for (int j = i + 1 + threadIdx.x; j < A; j += NT) {
      const float dx = fr[3 * j] - xi, dy = fr[3 * j + 1] - yi, dz = fr[3 * j + 2] - zi;
      c += fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < thr2 ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    counts[blockIdx.x] = t;
  }
}

// eigenvalues of a symmetric positive semi-definite 3x3 matrix (cyclic Jacobi, float64: the work is negligible)
__device__ inline void eig3_psd(double m[3][3], double ev[3]) {
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(m[0][1]) + fabs(m[0][2]) + fabs(m[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(m[p][q]) < 1e-300) continue;
        const double theta = (m[q][q] - m[p][p]) / (2.0 * m[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; ++k) {   // columns p, q
          const double a = m[k][p], b = m[k][q];
          m[k][p] = cs * a - sn * b;
          m[k][q] = sn * a + cs * b;
        }
        for (int k = 0; k < 3; ++k) {   // rows p, q
          const double a = m[p][k], b = m[q][k];
          m[p][k] = cs * a - sn * b;
          m[q][k] = sn * a + cs * b;
        }
      }
  }
  ev[0] = m[0][0];
  ev[1] = m[1][1];
  ev[2] = m[2][2];
}

// weighted Kabsch RMSD of frame pairs, one wave per pair                               callbacks/metrics.py:71-153 (kabsch_weighted, rmsd)
//   weights = tile(w3, A / 3); translate: every ATOM is shifted by the mean of its own three coordinates (the
//   reference's `a -= mean(a, axis=-1)`, metrics.py:149-151); msd = (PSQ + QSQ) iw - 2 (s1 + s2 +- s3), the sign of the
//   smallest singular value following det(C) (= det(V) det(W^T) of the SVD)
__global__ __launch_bounds__(NT) void kabsch_rmsd_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int A, float w0,
                                                         float w1, float w2, int translate, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int fidx = (blockIdx.x * NT + threadIdx.x) >> 6;
  if (fidx >= B) return;
  const float* __restrict__ P = a + (long)fidx * A * 3;
  const float* __restrict__ Q = b + (long)fidx * A * 3;
  // float64 sums: msd is a small difference of large sums (identical frames must give 0, not sqrt(eps) * size)
  double C[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}}, cp[3] = {0., 0., 0.}, cq[3] = {0., 0., 0.};
  double psq = 0., qsq = 0., wsum = 0.;
  auto wsumd = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int j = lane; j < A; j += 64) {
    const int r3 = j % 3;
    const double w = r3 == 0 ? w0 : (r3 == 1 ? w1 : w2);
    double p[3] = {P[3 * j], P[3 * j + 1], P[3 * j + 2]}, q[3] = {Q[3 * j], Q[3 * j + 1], Q[3 * j + 2]};
    if (translate) {
      const double mp = (p[0] + p[1] + p[2]) / 3., mq = (q[0] + q[1] + q[2]) / 3.;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        p[i] -= mp;
        q[i] -= mq;
      }
    }
    wsum += w;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      cp[i] = fma(p[i], w, cp[i]);
      cq[i] = fma(q[i], w, cq[i]);
      psq = fma(p[i] * p[i], w, psq);
      qsq = fma(q[i] * q[i], w, qsq);
#pragma unroll
      for (int k = 0; k < 3; ++k) C[i][k] = fma(p[i] * q[k], w, C[i][k]);
    }
  }
  wsum = wsumd(wsum);
  psq = wsumd(psq);
  qsq = wsumd(qsq);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    cp[i] = wsumd(cp[i]);
    cq[i] = wsumd(cq[i]);
#pragma unroll
    for (int k = 0; k < 3; ++k) C[i][k] = wsumd(C[i][k]);
  }
  if (lane != 0) return;
  const double iw = 1.0 / wsum;   // 3 / sum of the [A, 3] weight array
  double Cd[3][3], M[3][3];
  double PSQ = psq, QSQ = qsq;
  for (int i = 0; i < 3; ++i) {
    PSQ -= cp[i] * cp[i] * iw;
    QSQ -= cq[i] * cq[i] * iw;
    for (int k = 0; k < 3; ++k) Cd[i][k] = (C[i][k] - cp[i] * cq[k] * iw) * iw;
  }
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) M[i][k] = Cd[0][i] * Cd[0][k] + Cd[1][i] * Cd[1][k] + Cd[2][i] * Cd[2][k];   // C^T C
  double ev[3];
  eig3_psd(M, ev);
  double s[3] = {sqrt(fmax(ev[0], 0.0)), sqrt(fmax(ev[1], 0.0)), sqrt(fmax(ev[2], 0.0))};
  const double det = Cd[0][0] * (Cd[1][1] * Cd[2][2] - Cd[1][2] * Cd[2][1]) - Cd[0][1] * (Cd[1][0] * Cd[2][2] - Cd[1][2] * Cd[2][0]) +
                     Cd[0][2] * (Cd[1][0] * Cd[2][1] - Cd[1][1] * Cd[2][0]);
  const double smin = fmin(s[0], fmin(s[1], s[2]));
  double ssum = s[0] + s[1] + s[2];
  if (det < 0.0) ssum -= 2.0 * smin;
  const double msd = (PSQ + QSQ) * iw - 2.0 * ssum;
  out[fidx] = (float)sqrt(fmax(msd, 0.0));
}


// per-step record of the two metrics (Keras logs a metric's result() every epoch = step): mean clashes -- truncated like
// tf.reduce_mean of an int64 tensor, metrics.py:521 -- and mean RMSD, written at index *hstep (before the step's
// bookkeeping advances it)
__global__ __launch_bounds__(NT) void metric_tick_kernel(const long long* __restrict__ counts, const float* __restrict__ rmsd, int B,
                                                         float* __restrict__ clash_hist, float* __restrict__ rmsd_hist, long capacity,
                                                         const int64_t* __restrict__ hstep) {
  __shared__ float redf[NT / 64];
  __shared__ long long redi[NT / 64];
  long long c = 0;
  float r = 0.f;
  for (int i = threadIdx.x; i < B; i += NT) {
    if (counts) c += counts[i];
    if (rmsd) r += rmsd[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c += __shfl_xor(c, o, 64);
    r += __shfl_xor(r, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    redi[threadIdx.x >> 6] = c;
    redf[threadIdx.x >> 6] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long ct = 0;
    float rt = 0.f;
    for (int w = 0; w < NT / 64; ++w) {
      ct += redi[w];
      rt += redf[w];
    }
    const long s = *hstep;
    if (s >= 0 && s < capacity) {
      if (counts && clash_hist) clash_hist[s] = (float)(ct / B);
      if (rmsd && rmsd_hist) rmsd_hist[s] = rt / (float)B;
    }
  }
}

extern "C" {

int emx_colmean(const float* x, int B, int n, int ldx, float* out, emx_stream_t stream) {
  EMX_REQUIRE(x && out && B > 0 && n > 0 && ldx >= n, "emx_colmean: bad arguments");
  hipLaunchKernelGGL(colmean_kernel, dim3((n + 63) / 64), dim3(64 * CM_PH), 0, (hipStream_t)stream, x, B, n, ldx, out);
  return emx::check_launch("colmean_kernel");
}

int emx_chain_in_plane_fwd(const float* lengths, const float* angles, int lda, int B, int A, float* xyz, emx_stream_t stream) {
  EMX_REQUIRE(lengths && angles && xyz && B > 0 && A >= 3 && (lda >= A - 2 || lda == 0), "emx_chain_in_plane_fwd: bad arguments (A >= 3)");
  if ((size_t)A * sizeof(float) <= 48 * 1024) {
    hipLaunchKernelGGL(chain_frame_kernel, dim3(B), dim3(64), (size_t)A * sizeof(float), (hipStream_t)stream, lengths, angles, lda, A, xyz);
    return emx::check_launch("chain_frame_kernel");
  }
  hipLaunchKernelGGL(chain_fwd_kernel, dim3((B + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, lengths, angles, lda, B, A, xyz);
  return emx::check_launch("chain_fwd_kernel");
}

int emx_chain_in_plane_bwd(const float* lengths, const float* angles, int lda, const float* dxyz, int B, int A, float* dangles, int ldd,
                           emx_stream_t stream) {
  EMX_REQUIRE(lengths && angles && dxyz && dangles && B > 0 && A >= 3 && (lda >= A - 2 || lda == 0) && ldd >= A - 2, "emx_chain_in_plane_bwd: bad arguments");
  if ((size_t)A * sizeof(float) <= 48 * 1024) {
    hipLaunchKernelGGL(chain_bwd_frame_kernel, dim3(B), dim3(64), (size_t)A * sizeof(float), (hipStream_t)stream, lengths, angles, lda, dxyz, A,
                       dangles, ldd);
    return emx::check_launch("chain_bwd_frame_kernel");
  }
  hipLaunchKernelGGL(chain_bwd_kernel, dim3((B + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, lengths, angles, lda, dxyz, B, A, dangles,
                     ldd);
  return emx::check_launch("chain_bwd_kernel");
}

int emx_dihedrals_to_cartesian_fwd(const float* dihedrals, int ldd, float dihedral_offset, const float* plane, int B, int A, float* out,
                                   emx_stream_t stream) {
  EMX_REQUIRE(dihedrals && plane && out && B > 0 && A >= 4 && ldd >= A - 3, "emx_dihedrals_to_cartesian_fwd: bad arguments (A >= 4)");
  if (const size_t lds = (size_t)d2c_frame_floats(A, false) * sizeof(float); lds <= 160 * 1024) {
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&d2c_frame_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return emx::fail(EMX_ELAUNCH, "emx_dihedrals_to_cartesian_fwd: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(d2c_frame_kernel<false>, dim3(B), dim3(FR_NT_FWD), lds, (hipStream_t)stream, dihedrals, ldd, dihedral_offset, plane, nullptr, A,
                       out, nullptr, 0, nullptr);
    return emx::check_launch("d2c_frame_kernel (forward)");
  }
  hipLaunchKernelGGL(d2c_fwd_kernel, dim3((2 * B + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, dihedrals, ldd, dihedral_offset, plane, B,
                     A, out);
  return emx::check_launch("d2c_fwd_kernel");
}

int emx_dihedrals_to_cartesian_bwd(const float* dihedrals, int ldd, float dihedral_offset, const float* plane, const float* dout, int B, int A,
                                   float* ddihedrals, int ldg, float* dplane, emx_stream_t stream) {
  EMX_REQUIRE(dihedrals && plane && dout && ddihedrals && dplane && B > 0 && A >= 4 && ldd >= A - 3 && ldg >= A - 3,
              "emx_dihedrals_to_cartesian_bwd: bad arguments (A >= 4)");
  if (const size_t lds = (size_t)d2c_frame_floats(A, true) * sizeof(float); lds <= 160 * 1024) {
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&d2c_frame_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return emx::fail(EMX_ELAUNCH, "emx_dihedrals_to_cartesian_bwd: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(d2c_frame_kernel<true>, dim3(B), dim3(FR_NT_BWD), lds, (hipStream_t)stream, dihedrals, ldd, dihedral_offset, plane, dout, A,
                       nullptr, ddihedrals, ldg, dplane);
    return emx::check_launch("d2c_frame_kernel (backward)");
  }
  hipLaunchKernelGGL(d2c_bwd_kernel, dim3((B + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, dihedrals, ldd, dihedral_offset, plane, dout, B,
                     A, ddihedrals, ldg, dplane);
  return emx::check_launch("d2c_bwd_kernel");
}

int emx_pdist_flat_fwd(const float* xyz, int B, int A, int start, int step, int C, float* out, int ldo, emx_stream_t stream) {
  EMX_REQUIRE(xyz && out && B > 0 && A > 0 && C >= 2 && step >= 1 && start >= 0, "emx_pdist_flat_fwd: bad arguments");
  EMX_REQUIRE(start + (long)step * (C - 1) < A, "emx_pdist_flat_fwd: atom selection outside the frame");
  const long P = (long)C * (C - 1) / 2;
  EMX_REQUIRE(ldo >= P, "emx_pdist_flat_fwd: output row stride %d < %ld pairs", ldo, P);
  const size_t lds = (size_t)3 * C * sizeof(float);
  if (lds > 64 * 1024) return emx::fail(EMX_EUNSUPPORTED, "emx_pdist_flat_fwd: %d atoms do not fit the LDS frame buffer", C);
  hipLaunchKernelGGL(pdist_fwd_kernel, dim3(B), dim3(PD_NT), lds, (hipStream_t)stream, xyz, A, start, step, C, P, ldo, out);
  return emx::check_launch("pdist_fwd_kernel");
}

static int fill_frame_fwd(const char* who, const float* Y, int ldy, int B, float periodicity, int ang_sin, int ang_cos, const float* mean_angles,
                          float* ang_pred, int dih_sin, int dih_cos, float* dih_pred, const float* mean_lengths, float dihedral_offset, int A,
                          float* plane, float* xyz, const float* cart, int start, int step, int C, float* inp_pair, float* out_pair, int ldo,
                          void* split_ws, AdcFrameArgs& a, size_t& lds) {
  EMX_REQUIRE(Y && dih_pred && mean_lengths && plane && xyz && out_pair && B > 0 && A >= 4, "%s: bad arguments (A >= 4)", who);
  EMX_REQUIRE((ang_sin >= 0 && ang_cos >= 0 && ang_pred) || (ang_sin < 0 && mean_angles), "%s: the angles come from Y (columns + ang_pred) or from mean_angles", who);
  EMX_REQUIRE(ang_sin < 0 || (ang_sin + A - 2 <= ldy && ang_cos + A - 2 <= ldy), "%s: angle columns outside the row", who);
  EMX_REQUIRE(dih_sin >= 0 && dih_cos >= 0 && dih_sin + A - 3 <= ldy && dih_cos + A - 3 <= ldy, "%s: dihedral columns outside the row", who);
  EMX_REQUIRE(std::isfinite(periodicity) && periodicity > 0.f, "%s: needs a finite periodicity", who);
  EMX_REQUIRE(C >= 2 && step >= 1 && start >= 0 && start + (long)step * (C - 1) < A, "%s: atom selection outside the frame", who);
  const long P = (long)C * (C - 1) / 2;
  EMX_REQUIRE(ldo >= P, "%s: output row stride %d < %ld pairs", who, ldo, P);
  EMX_REQUIRE(!inp_pair || cart, "%s: the input frames are needed for their pair distances", who);
  EMX_REQUIRE(!split_ws || (inp_pair && (reinterpret_cast<uintptr_t>(split_ws) & 15) == 0 && P < (1L << 30)),
              "%s: the split needs the input pair block and a 16-byte aligned workspace", who);
  lds = (size_t)adc_frame_floats(A, C, split_ws != nullptr) * sizeof(float);
  if (lds > 64 * 1024) return emx::fail(EMX_EUNSUPPORTED, "%s: a frame of %d atoms does not fit the LDS buffers of the fused pass (use the single stages)", who, A);
  a.Y = Y;
  a.ldy = ldy;
  a.out_scale = periodicity == TWO_PI_F ? 1.f : periodicity / TWO_PI_F;
  a.n_ang = A - 2;
  a.ang_sin = ang_sin;
  a.ang_cos = ang_cos;
  a.mean_ang = mean_angles;
  a.ang_pred = ang_pred;
  a.dih_sin = dih_sin;
  a.dih_cos = dih_cos;
  a.dih_pred = dih_pred;
  a.mean_len = mean_lengths;
  a.offset = dihedral_offset;
  a.A = A;
  a.plane = plane;
  a.xyz = xyz;
  a.cart = cart;
  a.start = start;
  a.step = step;
  a.C = C;
  a.ldo = ldo;
  a.inp_pair = inp_pair;
  a.out_pair = out_pair;
  a.sp = split_ws ? emx::split_view(split_ws, B, (int)P) : emx::SplitRows{};
  a.split_wide = P >= 1024 && B <= 8192;   // (the rule of emx_sketchmap_split)
  return EMX_OK;
}

int emx_adc_frame_fwd(const float* Y, int ldy, int B, float periodicity, int ang_sin, int ang_cos, const float* mean_angles, float* ang_pred,
                      int dih_sin, int dih_cos, float* dih_pred, const float* mean_lengths, float dihedral_offset, int A, float* plane, float* xyz,
                      const float* cart, int start, int step, int C, float* inp_pair, float* out_pair, int ldo, void* split_ws,
                      emx_stream_t stream) {
  AdcFrameArgs a;
  size_t lds = 0;
  const int rc = fill_frame_fwd("emx_adc_frame_fwd", Y, ldy, B, periodicity, ang_sin, ang_cos, mean_angles, ang_pred, dih_sin, dih_cos, dih_pred,
                                mean_lengths, dihedral_offset, A, plane, xyz, cart, start, step, C, inp_pair, out_pair, ldo, split_ws, a, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(adc_frame_fwd_kernel, dim3(B), dim3(AF_NT), lds, (hipStream_t)stream, a);
  return emx::check_launch("adc_frame_fwd_kernel");
}

static int fill_frame_bwd(const char* who, const float* out_pair, const float* inp_pair, int ldo, int variant, float coef, const float* coef_scale,
                          float* ws, const float* xyz, const float* dihedrals, float dihedral_offset, const float* plane, int B, int A, int start,
                          int step, int C, float* ddihedrals, int ldg, float* dplane, const float* mean_lengths, const float* angles,
                          float* dangles, int ldd, const emx_cost_job* jobs, int n_jobs, AdcFrameBwdArgs& a, size_t& lds) {
  EMX_REQUIRE(!angles || (mean_lengths && dangles && ldd >= A - 2), "%s: chain_in_plane's backward pass needs the mean lengths and dangles [B, >= A-2]", who);
  EMX_REQUIRE(out_pair && inp_pair && ws && xyz && dihedrals && plane && ddihedrals && dplane && B > 0 && A >= 4 && ldg >= A - 3, "%s: bad arguments (A >= 4)", who);
  EMX_REQUIRE(C >= 2 && step >= 1 && start >= 0 && start + (long)step * (C - 1) < A, "%s: atom selection outside the frame", who);
  const long P = (long)C * (C - 1) / 2;
  EMX_REQUIRE(ldo >= P, "%s: row stride %d < %ld pairs", who, ldo, P);
  EMX_REQUIRE(variant >= EMX_MEAN_ABS && variant <= EMX_MEAN_NORM, "%s: cost variant %d not available", who, variant);
  // the stand-alone cost launch puts S = min(EMX_REDUCE_WS / B, ceil(P / 512)) workgroups of 256 threads on a row; the fused pass
  // is that partition for S = 4 (a few hundred frames with a few thousand pair distances each)
  const bool four = variant != EMX_MEAN_NORM && B <= EMX_REDUCE_WS / 4 && P > 3 * 512;
  if (!four)
    return emx::fail(EMX_EUNSUPPORTED, "%s: B = %d frames x %ld pair distances (variant %d) is outside the fused pass (use the single stages)", who, B, P, variant);
  lds = (size_t)adc_frame_bwd_floats(A, C) * sizeof(float);
  if (lds > 160 * 1024) return emx::fail(EMX_EUNSUPPORTED, "%s: a frame of %d atoms does not fit the LDS buffers of the fused pass (use the single stages)", who, A);
  a = AdcFrameBwdArgs{};
  a.cost.y = out_pair;
  a.cost.target = inp_pair;
  a.cost.dpred_add = nullptr;
  a.cost.coef_scale = coef_scale;
  a.cost.dY = nullptr;
  a.cost.ws = ws;
  a.cost.loss = nullptr;
  emx::fill_cost_scalars(a.cost, B, (int)P, ldo, ldo, 0, 0, 0, INFINITY, variant, coef);
  a.cost.row_begin = 0;
  a.cost.row_end = B;
  a.cost.prescale = 1;
  a.cost.wide = 4;
  a.xyz = xyz;
  a.dih = dihedrals;
  a.plane = plane;
  a.offset = dihedral_offset;
  a.A = A;
  a.start = start;
  a.step = step;
  a.C = C;
  int T = 8;
  while (T > 1 && (long)C * T > PD_NT) T >>= 1;
  a.T = T;
  a.ddih = ddihedrals;
  a.ldg = ldg;
  a.dplane = dplane;
  a.len = mean_lengths;
  a.ang = angles;
  a.dang = dangles;
  a.ldd = ldd;
  EMX_REQUIRE(n_jobs >= 0 && n_jobs <= EMX_MAX_COST_JOBS && (n_jobs == 0 || jobs), "%s: 0 .. %d cost jobs", who, EMX_MAX_COST_JOBS);
  a.n_jobs = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const emx_cost_job& q = jobs[j];
    EMX_REQUIRE(q.y && q.target && q.ws && q.n > 0 && q.ldt >= q.n, "%s: cost job %d: incomplete", who, j);
    EMX_REQUIRE(q.variant >= EMX_MEAN_ABS && q.variant <= EMX_MEAN_NORM, "%s: cost job %d: variant %d not available", who, j, q.variant);
    EMX_REQUIRE(q.ldy > 0 || !q.dY, "%s: cost job %d: a broadcast prediction (ldy == 0) cannot receive a gradient", who, j);
    EMX_REQUIRE(q.off_sin >= 0 && (q.ldy == 0 || q.off_sin + q.n <= q.ldy) &&
                    (!q.periodic_output || (q.off_cos >= 0 && (q.ldy == 0 || q.off_cos + q.n <= q.ldy) && std::isfinite(q.periodicity))),
                "%s: cost job %d: column window outside the row", who, j);
    for (int i = 0; i < j; ++i) EMX_REQUIRE(jobs[i].ws != q.ws, "%s: cost jobs %d and %d share their scratch", who, i, j);
    EMX_REQUIRE(q.ws != ws, "%s: cost job %d shares its scratch with the cartesian cost", who, j);
    // the stand-alone batch launch gives such a job one workgroup of 256 threads per row exactly when 128 <= n <= 512 (B <= 256 here)
    if (q.variant == EMX_MEAN_NORM || q.n < 128 || q.n > 512)
      return emx::fail(EMX_EUNSUPPORTED, "%s: cost job %d (n = %d, variant %d) is outside the fused pass (use emx_periodic_cost_batch)", who, j, q.n, q.variant);
    emx::CostArgs& c = a.job[j];
    c.y = q.y;
    c.target = q.target;
    c.dpred_add = q.dpred_add;
    c.coef_scale = q.coef_scale;
    c.dY = q.dY;
    c.ws = q.ws;
    c.loss = nullptr;
    emx::fill_cost_scalars(c, B, q.n, q.ldy, q.ldt, q.off_sin, q.off_cos, q.periodic_output, q.periodicity, q.variant, q.coef);
    c.row_begin = 0;
    c.row_end = B;
    c.prescale = 1;
    c.wide = 1;
  }
  return EMX_OK;
}

int emx_adc_frame_bwd(const float* out_pair, const float* inp_pair, int ldo, int variant, float coef, const float* coef_scale, float* ws,
                      const float* xyz, const float* dihedrals, float dihedral_offset, const float* plane, int B, int A, int start, int step, int C,
                      float* ddihedrals, int ldg, float* dplane, const float* mean_lengths, const float* angles, float* dangles, int ldd,
                      const emx_cost_job* jobs, int n_jobs, emx_stream_t stream) {
  const char* who = "emx_adc_frame_bwd";
  AdcFrameBwdArgs a;
  size_t lds = 0;
  const int rc = fill_frame_bwd(who, out_pair, inp_pair, ldo, variant, coef, coef_scale, ws, xyz, dihedrals, dihedral_offset, plane, B, A, start, step, C,
                                ddihedrals, ldg, dplane, mean_lengths, angles, dangles, ldd, jobs, n_jobs, a, lds);
  if (rc) return rc;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&adc_frame_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return emx::fail(EMX_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
  hipLaunchKernelGGL(adc_frame_bwd_kernel, dim3(B), dim3(AF_NT), lds, (hipStream_t)stream, a);
  return emx::check_launch("adc_frame_bwd_kernel");
}

int emx_adc_frame_fwd_bwd(const float* Y, int ldy, int B, float periodicity, int ang_sin, int ang_cos, const float* mean_angles, float* ang_pred,
                          int dih_sin, int dih_cos, float* dih_pred, const float* mean_lengths, float dihedral_offset, int A, float* plane,
                          float* xyz, const float* cart, int start, int step, int C, float* inp_pair, float* out_pair, int ldo, void* split_ws,
                          int variant, float coef, const float* coef_scale, float* ws, float* ddihedrals, int ldg, float* dplane, float* dangles,
                          int ldd, const emx_cost_job* jobs, int n_jobs, emx_stream_t stream) {
  const char* who = "emx_adc_frame_fwd_bwd";
  EMX_REQUIRE(inp_pair, "%s: the cartesian cost needs the input frames' pair distances (inp_pair)", who);
  AdcFrameArgs f;
  AdcFrameBwdArgs g;
  size_t lds_f = 0, lds_g = 0;
  int rc = fill_frame_fwd(who, Y, ldy, B, periodicity, ang_sin, ang_cos, mean_angles, ang_pred, dih_sin, dih_cos, dih_pred, mean_lengths, dihedral_offset,
                          A, plane, xyz, cart, start, step, C, inp_pair, out_pair, ldo, split_ws, f, lds_f);
  if (rc) return rc;
  rc = fill_frame_bwd(who, out_pair, inp_pair, ldo, variant, coef, coef_scale, ws, xyz, dih_pred, dihedral_offset, plane, B, A, start, step, C, ddihedrals,
                      ldg, dplane, mean_lengths, ang_sin >= 0 ? ang_pred : nullptr, dangles, ldd, jobs, n_jobs, g, lds_g);
  if (rc) return rc;
  const size_t lds = std::max(lds_f, lds_g);
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&adc_frame_both_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return emx::fail(EMX_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
  static_assert(sizeof(AdcFrameArgs) + sizeof(AdcFrameBwdArgs) <= 4096, "kernel arguments of adc_frame_both_kernel");
  hipLaunchKernelGGL(adc_frame_both_kernel, dim3(B), dim3(AF_NT), lds, (hipStream_t)stream, f, g);
  return emx::check_launch("adc_frame_both_kernel");
}

int emx_pdist_flat_bwd(const float* xyz, const float* dout, int B, int A, int start, int step, int C, int ldo, float* dxyz, int accumulate,
                       emx_stream_t stream) {
  EMX_REQUIRE(xyz && dout && dxyz && B > 0 && A > 0 && C >= 2 && step >= 1 && start >= 0, "emx_pdist_flat_bwd: bad arguments");
  EMX_REQUIRE(start + (long)step * (C - 1) < A, "emx_pdist_flat_bwd: atom selection outside the frame");
  EMX_REQUIRE(ldo >= (long)C * (C - 1) / 2, "emx_pdist_flat_bwd: gradient row stride too small");
  size_t lds = (size_t)3 * C * sizeof(float);
  if (lds > 64 * 1024) return emx::fail(EMX_EUNSUPPORTED, "emx_pdist_flat_bwd: %d atoms do not fit the LDS frame buffer", C);
  const size_t with_g = lds + (size_t)C * (C - 1) / 2 * sizeof(float);
  const int stage_g = with_g <= 64 * 1024;
  if (stage_g) lds = with_g;
  int T = 8;
  while (T > 1 && (long)C * T > PD_NT) T >>= 1;
  hipLaunchKernelGGL(pdist_bwd_kernel, dim3(B), dim3(PD_NT), lds, (hipStream_t)stream, xyz, dout, A, start, step, C, ldo, dxyz, accumulate, stage_g,
                     T);
  return emx::check_launch("pdist_bwd_kernel");
}

int emx_backbone_features(const float* xyz, int64_t N, int A, float* distances, float* angles, float* dihedrals, emx_stream_t stream) {
  EMX_REQUIRE(xyz && N > 0 && A >= 2 && (distances || angles || dihedrals), "emx_backbone_features: bad arguments");
  const long n = (long)N * A;
  hipLaunchKernelGGL(featurize_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, xyz, (long)N, A, distances, angles,
                     dihedrals);
  return emx::check_launch("featurize_kernel");
}

#ifdef EMX_D2C_TIMING
extern "C" int emx_debug_read_af(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg_af), sizeof(emx_dbg_af)); }
extern "C" int emx_debug_read_d2c(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emx_dbg_d2c), sizeof(emx_dbg_d2c)); }
#endif

int emx_allatom_mask_words(int A) { return A > 0 ? ((A + 2047) / 2048) * 64 : 0; }

int emx_allatom_backmap(const float* xyz0, int64_t frame_stride, int64_t n_frames, int A, const int32_t* quads, const uint32_t* far_mask,
                        int J, const float* targets, int64_t ldt, float* out, emx_stream_t stream) {
  EMX_REQUIRE(xyz0 && out && n_frames > 0 && A >= 4 && J >= 0, "emx_allatom_backmap: bad arguments");
  EMX_REQUIRE(J == 0 || (quads && far_mask && targets && ldt >= J), "emx_allatom_backmap: %d torsions without their tables", J);
  EMX_REQUIRE(frame_stride == 0 || frame_stride >= (int64_t)3 * A, "emx_allatom_backmap: frame stride %lld < 3 A", (long long)frame_stride);
  const size_t lds = (size_t)3 * A * sizeof(float);
  if (lds > 64 * 1024) return emx::fail(EMX_EUNSUPPORTED, "emx_allatom_backmap: %d atoms do not fit the LDS frame buffer", A);
  EMX_REQUIRE(n_frames <= 0x7fffffffLL, "emx_allatom_backmap: too many frames for one launch");
  hipLaunchKernelGGL(allatom_rotate_kernel, dim3((unsigned)n_frames), dim3(AA_NT), lds, (hipStream_t)stream, xyz0, (long)frame_stride, A, quads,
                     far_mask, (A + 2047) / 2048, J, targets, (long)ldt, out);
  return emx::check_launch("allatom_rotate_kernel");
}

int emx_soft_start_scale(const int64_t* step, float soft_a, float soft_b, float scale_c, float* out, emx_stream_t stream) {
  EMX_REQUIRE(step && out, "emx_soft_start_scale: null pointer argument");
  EMX_REQUIRE(soft_a < 0.f || soft_b > soft_a, "emx_soft_start_scale: soft start needs b > a");
  hipLaunchKernelGGL(soft_start_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, soft_a, soft_b, scale_c, out);
  return emx::check_launch("soft_start_kernel");
}

namespace {
int fill_adc_head(emx::AdcHeadArgs& a, const float* feat_data, const float* cart_data, int64_t N, int F, int C3, const int64_t* idx, const int64_t* cursor,
                  int B, float* feat, float* cart, const int* part_off, const int* part_n, const int* part_col, int n_parts, float periodicity, float* xin,
                  int ld_xin, const int64_t* step, float soft_a, float soft_b, float scale_c, float* cart_scale, float* feat_mean, const char* who) {
  EMX_REQUIRE(feat_data && cart_data && feat && cart && xin, "%s: null pointer argument", who);
  EMX_REQUIRE(N > 0 && F > 0 && C3 > 0 && B > 0 && n_parts >= 1 && n_parts <= 4 && part_off && part_n && part_col, "%s: bad shape", who);
  EMX_REQUIRE(std::isfinite(periodicity) && periodicity > 0.f, "%s: the angular blocks need a finite periodicity", who);
  EMX_REQUIRE(!cart_scale || (step && (soft_a < 0.f || soft_b > soft_a)), "%s: soft start needs the step counter and b > a", who);
  a = emx::AdcHeadArgs{};
  for (int p = 0; p < n_parts; ++p) {
    EMX_REQUIRE(part_off[p] >= 0 && part_n[p] > 0 && part_off[p] + part_n[p] <= F && part_col[p] >= 0 && part_col[p] + 2 * part_n[p] <= ld_xin,
                "%s: block %d outside the feature row or the encoder input", who, p);
    a.off[p] = part_off[p];
    a.n[p] = part_n[p];
    a.col[p] = part_col[p];
  }
  a.feat_data = feat_data;
  a.cart_data = cart_data;
  a.idx = idx;
  a.cursor = cursor;
  a.feat = feat;
  a.cart = cart;
  a.xin = xin;
  a.B = B;
  a.F = F;
  a.C3 = C3;
  a.ld_xin = ld_xin;
  a.n_parts = n_parts;
  const float two_pi = 6.283185307179586f;
  a.ang_scale = periodicity == two_pi ? 1.f : two_pi / periodicity;
  a.step = step;
  a.soft_a = soft_a;
  a.soft_b = soft_b;
  a.scale_c = scale_c;
  a.cart_scale = cart_scale;
  a.feat_mean = feat_mean;
  a.row_blocks = std::min(B, 2048);
  return EMX_OK;
}
}  // namespace

int emx_adc_batch_head(const float* feat_data, const float* cart_data, int64_t N, int F, int C3, const int64_t* idx, const int64_t* cursor,
                       int B, float* feat, float* cart, const int* part_off, const int* part_n, const int* part_col, int n_parts,
                       float periodicity, float* xin, int ld_xin, const int64_t* step, float soft_a, float soft_b, float scale_c,
                       float* cart_scale, float* feat_mean, emx_stream_t stream) {
  EMX_REQUIRE(idx, "emx_adc_batch_head: null pointer argument");
  emx::AdcHeadArgs a;
  int rc = fill_adc_head(a, feat_data, cart_data, N, F, C3, idx, cursor, B, feat, cart, part_off, part_n, part_col, n_parts, periodicity, xin, ld_xin,
                         step, soft_a, soft_b, scale_c, cart_scale, feat_mean, "emx_adc_batch_head");
  if (rc) return rc;
  const int grid = a.row_blocks + (feat_mean ? (F + NT / CM_PH - 1) / (NT / CM_PH) : 0);
  hipLaunchKernelGGL(adc_head_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a);
  return emx::check_launch("adc_head_kernel");
}

size_t emx_adc_head_bytes(void) { return sizeof(emx::AdcHeadArgs); }

int emx_adc_head_describe(const float* feat_data, const float* cart_data, int64_t N, int F, int C3, int B, float* feat, float* cart,
                          const int* part_off, const int* part_n, const int* part_col, int n_parts, float periodicity, float* xin, int ld_xin,
                          const int64_t* step, float soft_a, float soft_b, float scale_c, float* cart_scale, float* feat_mean, int n_batches,
                          void* record) {
  EMX_REQUIRE(record && n_batches > 0, "emx_adc_head_describe: null record / no batches");
  emx::AdcHeadArgs a;
  int rc = fill_adc_head(a, feat_data, cart_data, N, F, C3, nullptr, nullptr, B, feat, cart, part_off, part_n, part_col, n_parts, periodicity, xin,
                         ld_xin, step, soft_a, soft_b, scale_c, cart_scale, feat_mean, "emx_adc_head_describe");
  if (rc) return rc;
  a.n_batches = n_batches;
  a.batch_step = 1;
  if (hipMemcpy(record, &a, sizeof(a), hipMemcpyHostToDevice) != hipSuccess) return emx::fail(EMX_ELAUNCH, "emx_adc_head_describe: cannot write the record");
  return EMX_OK;
}

int emx_clash_count(const float* xyz, int B, int A, float clash_distance, int64_t* counts, emx_stream_t stream) {
  EMX_REQUIRE(xyz && counts && B > 0 && A >= 2 && clash_distance >= 0.f, "emx_clash_count: bad arguments");
  const size_t lds = (size_t)3 * A * sizeof(float);
  if (lds > 64 * 1024) return emx::fail(EMX_EUNSUPPORTED, "emx_clash_count: %d atoms do not fit the LDS frame buffer", A);
  hipLaunchKernelGGL(clash_count_kernel, dim3(B), dim3(NT), lds, (hipStream_t)stream, xyz, A, clash_distance * clash_distance,
                     reinterpret_cast<long long*>(counts));
  return emx::check_launch("clash_count_kernel");
}

int emx_kabsch_rmsd(const float* a, const float* b, int B, int A, const float weights3[3], int translate, float* rmsd, emx_stream_t stream) {
  EMX_REQUIRE(a && b && rmsd && weights3 && B > 0 && A >= 3, "emx_kabsch_rmsd: bad arguments");
  EMX_REQUIRE(A % 3 == 0, "emx_kabsch_rmsd: the weights repeat over N, CA, C -- the number of atoms must be a multiple of 3 (got %d)", A);
  hipLaunchKernelGGL(kabsch_rmsd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(NT), 0, (hipStream_t)stream, a, b, B, A, weights3[0], weights3[1],
                     weights3[2], translate, rmsd);
  return emx::check_launch("kabsch_rmsd_kernel");
}

int emx_metric_tick(const int64_t* counts, const float* rmsd, int B, float* clash_hist, float* rmsd_hist, int64_t capacity, const int64_t* hstep,
                    emx_stream_t stream) {
  EMX_REQUIRE(B > 0 && hstep && (counts || rmsd), "emx_metric_tick: bad arguments");
  EMX_REQUIRE((!counts || clash_hist) && (!rmsd || rmsd_hist), "emx_metric_tick: a metric without its history buffer");
  hipLaunchKernelGGL(metric_tick_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(counts), rmsd, B, clash_hist,
                     rmsd_hist, (long)capacity, hstep);
  return emx::check_launch("metric_tick_kernel");
}

}  // extern "C"
