// One-shot all-reduce of a SMALL buffer over xGMI peer mappings (gfx950): the step's [gradient | cost terms] buffer, 267 KB at C2.
//   replaces: nothing in the reference (single process: encodermap/autoencoder/autoencoder.py:741-748); in this engine the RCCL
//             all-reduce of engine.DataParallel (one collective per step).
//
// Why.  A ring or tree collective of 267 KB over eight GPUs is latency, not bandwidth: the ranks synchronise several times and a
// launch costs what a launch costs (RCCL: ~25 us and more at this size, where the whole step is ~65 us).  xGMI is point to point --
// every GPU reaches every peer's memory directly -- so the shortest exchange is ONE hop: every rank publishes its buffer in memory
// its peers have mapped, signals, and then READS all peers' buffers itself (seven links in parallel, 267 KB each: ~2 us at the
// links' rate) and adds them in rank order -- every rank the same additions in the same order: bit-identical results everywhere,
// which the replicas' weights depend on.
//
// Memory.  Per rank ONE allocation of uncached (fine-grained) device memory, exported by hipIpcGetMemHandle and opened by every peer:
//   [flags: world x EMX_P2P_CHUNKS words, epoch of the chunk's last publication by that rank][slot 0][slot 1]  (slots: n floats each)
// Uncached, so that what a kernel wrote is in memory -- visible to a peer's loads over the fabric -- once it is released at system
// scope, and so that a flag written by a peer is seen by the spinning reader.
//
// One launch, `chunks` workgroups; workgroup c owns floats [c, c + 1) * n / chunks:
//   (1) copy the chunk of the caller's buffer into the local slot (epoch & 1); behind a barrier, write the epoch into flags[rank][c] of
//       EVERY peer (and its own) -- a release at system scope;
//   (2) wait until flags[p][c] >= epoch for every p (the local copy of the flags: peers write, this rank polls its own memory) -- an
//       acquire at system scope per poll;
//   (3) out[i] = scale * (((slot_0[i] + slot_1[i]) + slot_2[i]) + ...) over the ranks in order, reading the peers' slots directly.
// No barrier at the end: slots alternate by epoch parity, and a rank can only publish epoch e + 2 into the slot of epoch e after
// its own launch e + 1 has seen every peer's flag e + 1 -- which a peer writes at the START of its launch e + 1, behind its launch e,
// i.e. behind its last read of that slot.  The epoch lives in device memory per workgroup (a launch increments its own word), so the
// launch can be captured in a hipGraph and replayed.  A wait that does not end (a peer died) gives up after ~20 s of polling and sets a
// sticky word the host reads (emx_p2p_status): the results from then on are invalid, nothing hangs.
#include <cstring>

#include "emx_common.h"

namespace {

using namespace emx;

constexpr int P2P_MAX_WORLD = 16;
constexpr int P2P_NT = 256;
constexpr unsigned P2P_SPIN_LIMIT = 5000000u;   // x s_sleep(127) ~ 4 us: ~20 s

struct P2PArgs {
  char* peer[P2P_MAX_WORLD];     // every rank's region as mapped HERE ([rank] = the local one)
  float* buf;                    // the caller's buffer: input and result
  unsigned* epoch;               // [chunks] device words of this rank (ordinary memory)
  unsigned* gave_up;             // sticky
  long n, slot_bytes;
  int world, rank, chunks, flags_bytes;
  float scale;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(P2P_NT) void p2p_allreduce_kernel(const P2PArgs a) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const unsigned e = a.epoch[c] + 1u;
  const long per = ((a.n + a.chunks - 1) / a.chunks + 3) & ~3L;       // floats per chunk, a multiple of four
  const long lo = (long)c * per, hi = lo + per < a.n ? lo + per : a.n;
  const long slot_off = a.flags_bytes + (long)(e & 1u) * a.slot_bytes;
  // (1) publish the chunk
  {
    float* const mine = reinterpret_cast<float*>(a.peer[a.rank] + slot_off);
    for (long i = lo + 4 * tid; i < hi; i += 4 * P2P_NT) {
      if (i + 4 <= hi) {
        *reinterpret_cast<f32x4*>(mine + i) = *reinterpret_cast<const f32x4*>(a.buf + i);
      } else {
        for (long j = i; j < hi; ++j) mine[j] = a.buf[j];
      }
    }
  }
  // (the chunk's stores -- every thread's -- happen before the barrier, the barrier before the flag's release at system scope: the
  //  release is cumulative.  An explicit __threadfence_system() of all 256 threads in front of it was 5 us of a one-rank launch.)
#ifdef P2P_EXTRA_FENCES
  __threadfence_system();
#endif
  __syncthreads();
  if (tid < a.world) {
    unsigned* const f = reinterpret_cast<unsigned*>(a.peer[tid]) + (long)a.rank * a.chunks + c;
    __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (2) every peer's chunk is there
  if (tid < a.world) {
    const unsigned* const f = reinterpret_cast<const unsigned*>(a.peer[a.rank]) + (long)tid * a.chunks + c;
    unsigned spins = 0;
    // (epochs only grow; the comparison survives their wrap-around after 4e9 launches)
    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      if (++spins > P2P_SPIN_LIMIT) {
        __hip_atomic_store(a.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(127);
    }
  }
  __syncthreads();      // (the polling threads' acquires at system scope happen before everybody's reads of the peers' slots)
#ifdef P2P_EXTRA_FENCES
  __threadfence_system();
#endif
  // (3) the sum over the ranks, in rank order
  for (long i = lo + 4 * tid; i < hi; i += 4 * P2P_NT) {
    if (i + 4 <= hi) {
      f32x4 s = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.peer[0] + slot_off) + i);
      for (int p = 1; p < a.world; ++p) s += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.peer[p] + slot_off) + i);
      *reinterpret_cast<f32x4*>(a.buf + i) = s * a.scale;
    } else {
      for (long j = i; j < hi; ++j) {
        float s = reinterpret_cast<const float*>(a.peer[0] + slot_off)[j];
        for (int p = 1; p < a.world; ++p) s += reinterpret_cast<const float*>(a.peer[p] + slot_off)[j];
        a.buf[j] = s * a.scale;
      }
    }
  }
  if (tid == 0) a.epoch[c] = e;
}

struct P2PCtx {
  int world, rank, chunks;
  long n, slot_bytes, flags_bytes, bytes;
  char* local;                    // this rank's region
  char* peer[P2P_MAX_WORLD];      // as mapped here
  bool opened[P2P_MAX_WORLD];
  unsigned* epoch;                // [chunks] + the sticky word
  bool connected;
};

}  // namespace

extern "C" {

int emx_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int emx_p2p_create(int world, int rank, int64_t n_floats, int chunks, void** ctx_out, void* handle_out) {
  const char* who = "emx_p2p_create";
  EMX_REQUIRE(ctx_out && handle_out, "%s: null argument", who);
  EMX_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "%s: rank %d of %d (at most %d ranks)", who, rank, world, P2P_MAX_WORLD);
  EMX_REQUIRE(n_floats >= 1 && n_floats <= (1L << 28), "%s: %lld floats", who, (long long)n_floats);
  EMX_REQUIRE(chunks >= 1 && chunks <= EMX_P2P_MAX_CHUNKS, "%s: %d chunks (1 .. %d)", who, chunks, EMX_P2P_MAX_CHUNKS);
  P2PCtx* c = new P2PCtx();
  c->world = world, c->rank = rank, c->chunks = chunks, c->n = n_floats, c->connected = false;
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return fail(EMX_ELAUNCH, "%s: no device", who);
  }
  c->flags_bytes = (((long)world * chunks * 4) + 4095) & ~4095L;
  c->slot_bytes = ((n_floats * 4) + 4095) & ~4095L;
  c->bytes = c->flags_bytes + 2 * c->slot_bytes;
  for (int p = 0; p < P2P_MAX_WORLD; ++p) c->peer[p] = nullptr, c->opened[p] = false;
  void* mem = nullptr;
  if (hipExtMallocWithFlags(&mem, (size_t)c->bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return fail(EMX_ELAUNCH, "%s: hipExtMallocWithFlags(%lld bytes, uncached)", who, (long long)c->bytes);
  }
  c->local = static_cast<char*>(mem);
  if (hipMemset(mem, 0, (size_t)c->bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&c->epoch), (chunks + 1) * 4) != hipSuccess ||
      hipMemset(c->epoch, 0, (chunks + 1) * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(mem);
    delete c;
    return fail(EMX_ELAUNCH, "%s: device memory", who);
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, mem) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(mem);
    (void)hipFree(c->epoch);
    delete c;
    return fail(EMX_ELAUNCH, "%s: hipIpcGetMemHandle", who);
  }
  std::memcpy(handle_out, &h, sizeof h);
  c->peer[rank] = c->local;
  *ctx_out = c;
  return EMX_OK;
}

int emx_p2p_connect(void* ctx, const void* handles) {
  const char* who = "emx_p2p_connect";
  EMX_REQUIRE(ctx && handles, "%s: null argument", who);
  P2PCtx* c = static_cast<P2PCtx*>(ctx);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank || c->opened[p]) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)p * sizeof h, sizeof h);
    void* mapped = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(EMX_ELAUNCH, "%s: hipIpcOpenMemHandle of rank %d's region: %s", who, p, hipGetErrorString(e));
    }
    c->peer[p] = static_cast<char*>(mapped);
    c->opened[p] = true;
  }
  c->connected = true;
  return EMX_OK;
}

int emx_p2p_allreduce(void* ctx, float* buf, int64_t n, float scale, emx_stream_t stream) {
  const char* who = "emx_p2p_allreduce";
  EMX_REQUIRE(ctx && buf, "%s: null argument", who);
  P2PCtx* c = static_cast<P2PCtx*>(ctx);
  EMX_REQUIRE(c->connected || c->world == 1, "%s: emx_p2p_connect first", who);
  EMX_REQUIRE(n >= 1 && n <= c->n, "%s: %lld floats, the regions hold %lld", who, (long long)n, (long long)c->n);
  EMX_REQUIRE((reinterpret_cast<unsigned long long>(buf) & 15) == 0, "%s: the buffer must be 16-byte aligned", who);
  P2PArgs a{};
  for (int p = 0; p < c->world; ++p) a.peer[p] = c->peer[p];
  a.buf = buf;
  a.epoch = c->epoch;
  a.gave_up = c->epoch + c->chunks;
  a.n = n;
  a.slot_bytes = c->slot_bytes;
  a.world = c->world, a.rank = c->rank, a.chunks = c->chunks, a.flags_bytes = (int)c->flags_bytes;
  a.scale = scale;
  hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(c->chunks), dim3(P2P_NT), 0, (hipStream_t)stream, a);
  return check_launch(who);
}

int emx_p2p_status(void* ctx, int* gave_up) {
  const char* who = "emx_p2p_status";
  EMX_REQUIRE(ctx && gave_up, "%s: null argument", who);
  P2PCtx* c = static_cast<P2PCtx*>(ctx);
  unsigned v = 0;
  if (hipMemcpy(&v, c->epoch + c->chunks, 4, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return fail(EMX_ELAUNCH, "%s: hipMemcpy", who);
  }
  *gave_up = (int)v;
  return EMX_OK;
}

int emx_p2p_destroy(void* ctx) {
  if (!ctx) return EMX_OK;
  P2PCtx* c = static_cast<P2PCtx*>(ctx);
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (c->opened[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  (void)hipFree(c->local);
  (void)hipFree(c->epoch);
  delete c;
  return EMX_OK;
}

}  // extern "C"
