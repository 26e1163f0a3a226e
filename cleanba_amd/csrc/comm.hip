// comm.hip — everything that crosses a GPU boundary, behind the C ABI.
//
//  (1) RCCL collectives among the learner GPUs (replaces jax.pmap / jax.lax.pmean, ppo:628,649-660): librccl is bound at run time
//      (dlopen), so the library has no link-time RCCL dependency and uses the SAME copy as whatever else lives in the process.  One
//      communicator slot per purpose; the gradient all-reduce runs on the context's own communication stream, its dense + heads tail
//      (95 % of the bytes) as soon as the backward pass has produced it, the conv head after the backward pass; the learner stream
//      joins before the optimizer.  The host (any language) only has to carry the 128-byte unique id from rank 0 to the others.
//  (2) Peer writes for split actor / learner topologies (replaces jax.device_put_sharded, ppo:358-363, and device_put(params)
//      back, ppo:721-725): a learner exports HIP IPC handles of its ring fields, the actor maps them and writes each learner's
//      [T+1][E/L] column shard STRAIGHT into that ring with one strided 2-D copy per field on a side stream (xGMI peer write, no
//      staging, bytes moved = the shard); learner 0 writes new parameters straight into the actor's versioned parameter buffers the same
//      way.  "Delivered" notifications are host messages (the reference blocks on queue.put/get at the same points).
#include "cbm_ctx.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <string>
#include <unistd.h>

// ------------------------------------------------------------------------------------------ RCCL binding
namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load(const char* path) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return 0;
  const char* cands[] = {path, getenv("CBM_RCCL_PATH"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* h = nullptr;
  std::string tried;
  for (const char* p : cands) {
    if (!p || !*p) continue;
    h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    tried += std::string(p) + ": " + dlerror() + "; ";
  }
  if (!h) { cbm_set_error("cannot load librccl (%s)", tried.c_str()); return -1; }
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
    cbm_set_error("librccl is missing one of ncclGetUniqueId/CommInitRank/CommDestroy/AllReduce/GetErrorString");
    return -1;
  }
  g_rccl = r;
  return 0;
}
}  // namespace

#define CBM_NCCL(call)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess) {                                                                             \
      cbm_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__);      \
      return -1;                                                                                         \
    }                                                                                                    \
  } while (0)

__global__ void comm_scale_f32_kernel(float* x, int64_t n, float f) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= f;
}
__global__ void comm_scale_f64_kernel(double* x, int n, double f) {
  if ((int)threadIdx.x < n) x[threadIdx.x] *= f;
}
// ------------------------------------------------------------------------------------------ native all-reduce (no RCCL)
// One kernel per collective, working on the peers' buffers through their IPC mappings (or plain pointers when the peer context lives in this
// process).  Synchronisation = flag words in per-rank signal blocks: block b of rank r announces phase p of collective #seq by storing seq
// into sig[peer][p][b][r] of EVERY peer (system-scope release store) and waits until its own sig[r][p][b][*] all carry >= seq.  Flags only
// grow, so nothing is ever reset; block b only ever talks to block b of the peers, and a kernel has NAT_BLOCKS (<< what the chip holds) blocks, so
// the waiting kernels of several ranks that share ONE GPU are co-resident and cannot starve each other.
//   two-shot (flat gradient, in place):  [start] everybody's input is final -> rank r sums slice r of all peers in rank order 0..n-1 (PEER READS) and
//       stores the sum into slice r of all peers (PEER WRITES; only rank r ever touches slice r between the two barriers) -> [end] all slices landed.
//   one-shot (statistics, f64 scratch): [start] -> every rank reads all peers, keeps the result in registers -> [end] everybody has read -> write own.
// Deterministic by construction: one fixed summation order per element, independent of timing (ppo:30 asks XLA for the same).
#define NAT_BLOCKS 64
#define NAT_THREADS 512
#define NAT_SMALL_MAX 4096
#define NAT_SIG_WORDS (2 * NAT_BLOCKS * CBM_NATIVE_MAX_RANKS)
struct NatArgs {
  void* data[CBM_NATIVE_MAX_RANKS];
  uint32_t* sig[CBM_NATIVE_MAX_RANKS];
  int nranks, rank;
  int64_t off, n;                     // element range of this collective inside the registered buffer
  uint32_t seq;
  int* err;                           // page-locked host word: 1 / 2 = the start / end wait timed out
  unsigned long long timeout_ticks;   // wall_clock64() ticks (100 MHz)
};
struct NatBlob {   // what a rank publishes (CBM_NATIVE_BLOB_BYTES): IPC handles for other processes, plain pointers for contexts of the same process
  uint8_t ipc[CBM_NATIVE_BUFS][CBM_IPC_HANDLE_BYTES];
  uint64_t ptr[CBM_NATIVE_BUFS];
  uint32_t pid, device;
};
static_assert(sizeof(NatBlob) <= CBM_NATIVE_BLOB_BYTES, "blob size");

static __device__ __forceinline__ void nat_signal_and_wait(const NatArgs& a, int phase) {
  __syncthreads();
  const int t = threadIdx.x;
  if (t < a.nranks) {
    const size_t row = ((size_t)phase * NAT_BLOCKS + blockIdx.x) * CBM_NATIVE_MAX_RANKS;
    __hip_atomic_store(a.sig[t] + row + a.rank, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = a.sig[a.rank] + row + t;
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > a.timeout_ticks) { *(volatile int*)a.err = 1 + phase; break; }
    }
  }
  __syncthreads();
}

// NR > 0: the rank count is a compile-time constant — the NR peer loads of an element are all requested before the first add (one fabric round trip
// per element instead of NR); NR = 0: any count up to CBM_NATIVE_MAX_RANKS, one load after the other
template <int NR>
__global__ __launch_bounds__(NAT_THREADS) void nat_allreduce_f32_kernel(const NatArgs a) {
  __threadfence_system();
  nat_signal_and_wait(a, 0);
  __threadfence_system();               // acquire for every thread: no load below may be served by anything fetched before the peers signalled
  const int N = NR > 0 ? NR : a.nranks;
  const int64_t chunk = (((a.n + N - 1) / N) + 3) & ~(int64_t)3;
  const int64_t lo = chunk * a.rank < a.n ? chunk * a.rank : a.n;
  const int64_t hi = lo + chunk < a.n ? lo + chunk : a.n;
  const int64_t tid = (int64_t)blockIdx.x * NAT_THREADS + threadIdx.x, nth = (int64_t)NAT_BLOCKS * NAT_THREADS;
  const int64_t nvec = (a.off & 3) == 0 ? (hi - lo) >> 2 : 0;   // slices start at multiples of 4 floats: vector path when the range does too
  for (int64_t i = tid; i < nvec; i += nth) {
    const int64_t e = a.off + lo + 4 * i;
    float4 sum;
    if constexpr (NR > 0) {
      float4 v[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) v[r] = *reinterpret_cast<const float4*>((const float*)a.data[r] + e);
      sum = v[0];
#pragma unroll
      for (int r = 1; r < NR; ++r) { sum.x = sum.x + v[r].x; sum.y = sum.y + v[r].y; sum.z = sum.z + v[r].z; sum.w = sum.w + v[r].w; }   // rank order
#pragma unroll
      for (int r = 0; r < NR; ++r) *reinterpret_cast<float4*>((float*)a.data[r] + e) = sum;
    } else {
      sum = *reinterpret_cast<const float4*>((const float*)a.data[0] + e);
      for (int r = 1; r < N; ++r) {
        const float4 v = *reinterpret_cast<const float4*>((const float*)a.data[r] + e);
        sum.x = sum.x + v.x; sum.y = sum.y + v.y; sum.z = sum.z + v.z; sum.w = sum.w + v.w;
      }
      for (int r = 0; r < N; ++r) *reinterpret_cast<float4*>((float*)a.data[r] + e) = sum;
    }
  }
  for (int64_t i = lo + 4 * nvec + tid; i < hi; i += nth) {
    const int64_t e = a.off + i;
    float sum = ((const float*)a.data[0])[e];
    for (int r = 1; r < N; ++r) sum = sum + ((const float*)a.data[r])[e];
    for (int r = 0; r < N; ++r) ((float*)a.data[r])[e] = sum;
  }
  __threadfence_system();               // release: the peer writes above are out before the flag is
  nat_signal_and_wait(a, 1);
}

template <class Tv, int OP>   // OP: 0 sum, 1 max, 2 min; one block, n <= NAT_SMALL_MAX
__global__ __launch_bounds__(1024) void nat_allreduce_small_kernel(const NatArgs a) {
  __threadfence_system();
  nat_signal_and_wait(a, 0);
  __threadfence_system();
  Tv keep[NAT_SMALL_MAX / 1024];
#pragma unroll
  for (int j = 0; j < NAT_SMALL_MAX / 1024; ++j) {
    const int64_t i = threadIdx.x + 1024 * j;
    Tv sum = 0;
    if (i < a.n) {
      sum = ((const Tv*)a.data[0])[a.off + i];
      for (int r = 1; r < a.nranks; ++r) {
        const Tv v = ((const Tv*)a.data[r])[a.off + i];
        sum = OP == 0 ? sum + v : (OP == 1 ? (v > sum ? v : sum) : (v < sum ? v : sum));
      }
    }
    keep[j] = sum;
  }
  __threadfence_system();
  nat_signal_and_wait(a, 1);            // every rank has read every input: the in-place results may go out
#pragma unroll
  for (int j = 0; j < NAT_SMALL_MAX / 1024; ++j) {
    const int64_t i = threadIdx.x + 1024 * j;
    if (i < a.n) ((Tv*)a.data[a.rank])[a.off + i] = keep[j];
  }
}

static unsigned long long nat_timeout_ticks() {
  const char* e = getenv("CBM_NATIVE_TIMEOUT_S");
  const double sec = e && atof(e) > 0 ? atof(e) : 120.0;
  return (unsigned long long)(sec * 1e8);
}
// which registered buffer does [buf, buf + n*elt) live in?  0 grads, 1 statistics, 2 f64 scratch
static int nat_locate(cbm_ctx* c, const void* buf, int64_t nbytes, int* which_buf, int64_t* off_bytes) {
  const struct { const void* base; int64_t bytes; } reg[3] = {{c->grads, c->P * 4}, {c->stats_dev, (int64_t)c->stat_rows * 8 * 4}, {c->comm_scratch, CBM_COMM_SCRATCH * 8}};
  for (int b = 0; b < 3; ++b) {
    const int64_t d = (const char*)buf - (const char*)reg[b].base;
    if (d >= 0 && d + nbytes <= reg[b].bytes) { *which_buf = b; *off_bytes = d; return 0; }
  }
  cbm_set_error("native all-reduce: the buffer is not one of the registered ones (gradients, statistics, scratch)");
  return -1;
}
static NatArgs nat_args(CbmComm& k, int b, int64_t off, int64_t n) {
  NatArgs a;
  memset(&a, 0, sizeof(a));
  for (int r = 0; r < k.nranks; ++r) { a.data[r] = k.nat_peer[b][r]; a.sig[r] = (uint32_t*)k.nat_peer[3][r]; }
  a.nranks = k.nranks; a.rank = k.rank; a.off = off; a.n = n; a.seq = ++k.nat_seq; a.err = k.nat_err; a.timeout_ticks = nat_timeout_ticks();
  return a;
}
static int nat_allreduce_f32(cbm_ctx* c, CbmComm& k, float* buf, int64_t n, hipStream_t st) {
  if (n <= 0) return 0;
  int b = 0; int64_t offb = 0;
  if (nat_locate(c, buf, n * 4, &b, &offb)) return -1;
  const NatArgs a = nat_args(k, b, offb / 4, n);
  if (n <= NAT_SMALL_MAX) hipLaunchKernelGGL((nat_allreduce_small_kernel<float, 0>), dim3(1), dim3(1024), 0, st, a);
  else {
    const dim3 g(NAT_BLOCKS), t(NAT_THREADS);
    switch (k.nranks) {
      case 2: hipLaunchKernelGGL(nat_allreduce_f32_kernel<2>, g, t, 0, st, a); break;
      case 3: hipLaunchKernelGGL(nat_allreduce_f32_kernel<3>, g, t, 0, st, a); break;
      case 4: hipLaunchKernelGGL(nat_allreduce_f32_kernel<4>, g, t, 0, st, a); break;
      case 6: hipLaunchKernelGGL(nat_allreduce_f32_kernel<6>, g, t, 0, st, a); break;
      case 8: hipLaunchKernelGGL(nat_allreduce_f32_kernel<8>, g, t, 0, st, a); break;
      default: hipLaunchKernelGGL(nat_allreduce_f32_kernel<0>, g, t, 0, st, a);
    }
  }
  CBM_HIP(hipGetLastError());
  return 0;
}
// a flag wait timed out inside a native collective (a peer died or never joined): reported by the next call that synchronises with the device
int cbm_comm_check_native(cbm_ctx* c) {
  for (int i = 0; i < CBM_COMM_SLOTS; ++i) {
    CbmComm& k = c->comms[i];
    if (k.native && k.nat_err && *(volatile int*)k.nat_err) {
      cbm_set_error("native all-reduce (communicator %d, rank %d of %d): the %s wait timed out — a peer rank died or never reached collective #%u",
                    i, k.rank, k.nranks, *(volatile int*)k.nat_err == 1 ? "start" : "end", k.nat_seq);
      return -1;
    }
  }
  return 0;
}

// all-reduce(SUM) of fp32 data on `st`: RCCL, the native kernels, or the self-test communicator's "n identical ranks"
static int comm_allreduce_f32(cbm_ctx* c, CbmComm& k, float* buf, int64_t n, hipStream_t st) {
  if (k.loopback) {
    if (n > 0) comm_scale_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(buf, n, (float)k.nranks);
    return 0;
  }
  if (k.native) return nat_allreduce_f32(c, k, buf, n, st);
  CBM_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)k.comm, st));
  return 0;
}

static int comm_check(cbm_ctx* c, int which) {
  if (which < 0 || which >= CBM_COMM_SLOTS) { cbm_set_error("communicator slot %d outside [0,%d)", which, CBM_COMM_SLOTS); return -1; }
  if (!c->comms[which].nranks) { cbm_set_error("communicator %d is not initialised (cbm_comm_init)", which); return -1; }
  return 0;
}

extern "C" int cbm_comm_load(const char* librccl_path) { return rccl_load(librccl_path); }

extern "C" int cbm_comm_unique_id(uint8_t id[CBM_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == CBM_COMM_ID_BYTES, "ncclUniqueId size");
  if (rccl_load(nullptr)) return -1;
  ncclUniqueId u;
  CBM_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int cbm_comm_init(cbm_ctx* c, int32_t which, const uint8_t id[CBM_COMM_ID_BYTES], int32_t nranks, int32_t rank) {
  if (which < 0 || which >= CBM_COMM_SLOTS) { cbm_set_error("communicator slot %d outside [0,%d)", which, CBM_COMM_SLOTS); return -1; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { cbm_set_error("bad rank %d of %d", rank, nranks); return -1; }
  if (c->comms[which].nranks) { cbm_set_error("communicator %d already initialised", which); return -1; }
  if (rccl_load(nullptr)) return -1;
  CBM_HIP(hipSetDevice(c->cfg.device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  CBM_NCCL(g_rccl.CommInitRank(&comm, nranks, u, rank));
  c->comms[which].comm = comm;
  c->comms[which].nranks = nranks;
  c->comms[which].rank = rank;
  return 0;
}

extern "C" int cbm_comm_init_loopback(cbm_ctx* c, int32_t which, int32_t nranks) {
  if (which < 0 || which >= CBM_COMM_SLOTS || nranks < 1) { cbm_set_error("bad loopback communicator"); return -1; }
  if (c->comms[which].nranks) { cbm_set_error("communicator %d already initialised", which); return -1; }
  c->comms[which].nranks = nranks;
  c->comms[which].rank = 0;
  c->comms[which].loopback = true;
  return 0;
}

extern "C" int cbm_comm_native_export(cbm_ctx* c, int32_t which, uint8_t blob[CBM_NATIVE_BLOB_BYTES]) {
  if (which < 0 || which >= CBM_COMM_SLOTS) { cbm_set_error("communicator slot %d outside [0,%d)", which, CBM_COMM_SLOTS); return -1; }
  CBM_HIP(hipSetDevice(c->cfg.device));
  CbmComm& k = c->comms[which];
  if (!k.nat_sig_local) {
    // the signal block: uncached device memory when the runtime offers it (flags are polled by other GPUs), plain device memory otherwise
    // (the flag accesses are system-scope atomics either way)
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, NAT_SIG_WORDS * 4, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); CBM_HIP(hipMalloc(&p, NAT_SIG_WORDS * 4)); }
    CBM_HIP(hipMemset(p, 0, NAT_SIG_WORDS * 4));
    CBM_HIP(hipDeviceSynchronize());
    k.nat_sig_local = p;
    CBM_HIP(hipHostMalloc((void**)&k.nat_err, 64, hipHostMallocDefault));
    *k.nat_err = 0;
  }
  NatBlob b;
  memset(&b, 0, sizeof(b));
  void* const bufs[CBM_NATIVE_BUFS] = {c->grads, c->stats_dev, c->comm_scratch, k.nat_sig_local};
  for (int i = 0; i < CBM_NATIVE_BUFS; ++i) {
    hipIpcMemHandle_t h;
    CBM_HIP(hipIpcGetMemHandle(&h, bufs[i]));
    memcpy(b.ipc[i], &h, sizeof(h));
    b.ptr[i] = (uint64_t)(uintptr_t)bufs[i];
  }
  b.pid = (uint32_t)getpid();
  b.device = (uint32_t)c->cfg.device;
  memset(blob, 0, CBM_NATIVE_BLOB_BYTES);
  memcpy(blob, &b, sizeof(b));
  return 0;
}

extern "C" int cbm_comm_native_init(cbm_ctx* c, int32_t which, int32_t nranks, int32_t rank, const uint8_t* blobs) {
  if (which < 0 || which >= CBM_COMM_SLOTS) { cbm_set_error("communicator slot %d outside [0,%d)", which, CBM_COMM_SLOTS); return -1; }
  if (nranks < 1 || nranks > CBM_NATIVE_MAX_RANKS || rank < 0 || rank >= nranks) { cbm_set_error("native communicator: rank %d of %d (at most %d ranks)", rank, nranks, CBM_NATIVE_MAX_RANKS); return -1; }
  CbmComm& k = c->comms[which];
  if (k.nranks) { cbm_set_error("communicator %d already initialised", which); return -1; }
  if (!k.nat_sig_local) { cbm_set_error("cbm_comm_native_export must be called on this context first"); return -1; }
  CBM_HIP(hipSetDevice(c->cfg.device));
  void* const own[CBM_NATIVE_BUFS] = {c->grads, c->stats_dev, c->comm_scratch, k.nat_sig_local};
  for (int r = 0; r < nranks; ++r) {
    NatBlob b;
    memcpy(&b, blobs + (size_t)r * CBM_NATIVE_BLOB_BYTES, sizeof(b));
    for (int i = 0; i < CBM_NATIVE_BUFS; ++i) {
      if (r == rank) { k.nat_peer[i][r] = own[i]; continue; }
      if (b.pid == (uint32_t)getpid()) { k.nat_peer[i][r] = (void*)(uintptr_t)b.ptr[i]; continue; }   // a context of this process: no IPC needed (nor possible)
      hipIpcMemHandle_t h;
      memcpy(&h, b.ipc[i], sizeof(h));
      void* p = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        cbm_set_error("native communicator: hipIpcOpenMemHandle of rank %d's buffer %d (GPU %u) failed on GPU %d: %s (needs HSA_ENABLE_IPC_MODE_LEGACY=0 in "
                      "every process and a peer path between the GPUs)", r, i, b.device, c->cfg.device, hipGetErrorString(e));
        return -1;
      }
      k.nat_peer[i][r] = p;
      k.nat_mapped[i][r] = true;
    }
  }
  k.native = true;
  k.nranks = nranks;
  k.rank = rank;
  k.nat_seq = 0;
  return 0;
}

extern "C" const char* cbm_comm_backend(cbm_ctx* c, int32_t which) {
  if (which < 0 || which >= CBM_COMM_SLOTS || !c->comms[which].nranks) return "";
  return c->comms[which].loopback ? "loopback" : (c->comms[which].native ? "native" : "rccl");
}

extern "C" int cbm_comm_size(cbm_ctx* c, int32_t which) {
  if (which < 0 || which >= CBM_COMM_SLOTS) return 0;
  return c->comms[which].nranks;
}

int cbm_comm_destroy_all(cbm_ctx* c) {
  for (int i = 0; i < CBM_COMM_SLOTS; ++i) {
    CbmComm& k = c->comms[i];
    if (k.comm) { g_rccl.CommDestroy((ncclComm_t)k.comm); k.comm = nullptr; }
    for (int b = 0; b < CBM_NATIVE_BUFS; ++b)
      for (int r = 0; r < CBM_NATIVE_MAX_RANKS; ++r)
        if (k.nat_mapped[b][r]) { (void)hipIpcCloseMemHandle(k.nat_peer[b][r]); k.nat_mapped[b][r] = false; }
    if (k.nat_sig_local) { (void)hipFree(k.nat_sig_local); k.nat_sig_local = nullptr; }
    if (k.nat_err) { (void)hipHostFree(k.nat_err); k.nat_err = nullptr; }
    k.native = false;
    k.nranks = 0;
  }
  return 0;
}

// all-reduce of a few host doubles (barriers, max-over-ranks timing): staged through the context's scratch on the communication stream
// (one stream per communicator), blocking; call it from the thread that drives the learner
extern "C" int cbm_comm_allreduce_f64(cbm_ctx* c, int32_t which, double* host_inout, int32_t n, int32_t op) {
  if (comm_check(c, which)) return -1;
  if (n < 1 || n > CBM_COMM_SCRATCH) { cbm_set_error("cbm_comm_allreduce_f64 carries 1..%d values", CBM_COMM_SCRATCH); return -1; }
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipMemcpyAsync(c->comm_scratch, host_inout, (size_t)n * 8, hipMemcpyHostToDevice, c->cstream));
  if (c->comms[which].loopback) {
    if (op == 0) comm_scale_f64_kernel<<<dim3(1), dim3(64), 0, c->cstream>>>(c->comm_scratch, n, (double)c->comms[which].nranks);
  } else if (c->comms[which].native) {
    const NatArgs a = nat_args(c->comms[which], 2, 0, n);
    if (op == 1) hipLaunchKernelGGL((nat_allreduce_small_kernel<double, 1>), dim3(1), dim3(1024), 0, c->cstream, a);
    else if (op == 2) hipLaunchKernelGGL((nat_allreduce_small_kernel<double, 2>), dim3(1), dim3(1024), 0, c->cstream, a);
    else hipLaunchKernelGGL((nat_allreduce_small_kernel<double, 0>), dim3(1), dim3(1024), 0, c->cstream, a);
  } else {
    CBM_NCCL(g_rccl.AllReduce(c->comm_scratch, c->comm_scratch, (size_t)n, ncclFloat64, op == 1 ? ncclMax : (op == 2 ? ncclMin : ncclSum),
                              (ncclComm_t)c->comms[which].comm, c->cstream));
  }
  CBM_HIP(hipMemcpyAsync(host_inout, c->comm_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->cstream));
  CBM_HIP(hipStreamSynchronize(c->cstream));
  return cbm_comm_check_native(c);
}

extern "C" int cbm_comm_barrier(cbm_ctx* c, int32_t which) {
  double one = 1.0;
  return cbm_comm_allreduce_f64(c, which, &one, 1, 0);
}

// pmean(grads) over the CBM_COMM_LEARNERS communicator (ppo:628), overlapped with the backward pass.  Call after
// cbm_learner_minibatch_grad (which only enqueues).  The flat gradient is laid out conv1 | conv2 | conv3 | dense | heads and produced
// from the back: the tail [w[3], P) is final at tail_ev, long before the conv dgrad / wgrad kernels finish.  Both pieces go through the
// communication stream (one stream per communicator: no cross-stream ordering is asked of RCCL), the learner stream waits for it.
// *grad_div = number of ranks (the mean is taken inside the optimizer kernel); 1 and no work when the communicator does not exist.
int cbm_learner_allreduce_grads_impl(cbm_ctx* c, float* grad_div) {
  CbmComm& k = c->comms[CBM_COMM_LEARNERS];
  if (grad_div) *grad_div = k.nranks ? (float)k.nranks : 1.0f;
  if (!k.nranks) return 0;
  const int64_t tail = c->L.w[3];
  const bool timed = c->comm_prof_on && c->comm_prof_n < CBM_COMM_PROF_MAX;
  hipEvent_t* ev = timed ? &c->comm_prof_ev[4 * c->comm_prof_n] : nullptr;
  if (c->comm_overlap) {
    CBM_HIP(hipStreamWaitEvent(c->cstream, c->tail_ev, 0));
    if (timed) CBM_HIP(hipEventRecord(ev[0], c->cstream));
    if (comm_allreduce_f32(c, k, c->grads + tail, c->P - tail, c->cstream)) return -1;
    if (timed) CBM_HIP(hipEventRecord(ev[1], c->cstream));
    CBM_HIP(hipEventRecord(c->bwd_ev, c->lstream));          // the backward pass is complete on the learner stream here
    if (timed) CBM_HIP(hipEventRecord(ev[2], c->lstream));
    CBM_HIP(hipStreamWaitEvent(c->cstream, c->bwd_ev, 0));
    if (comm_allreduce_f32(c, k, c->grads, tail, c->cstream)) return -1;
  } else {
    CBM_HIP(hipEventRecord(c->bwd_ev, c->lstream));
    if (timed) CBM_HIP(hipEventRecord(ev[2], c->lstream));
    CBM_HIP(hipStreamWaitEvent(c->cstream, c->bwd_ev, 0));
    if (timed) CBM_HIP(hipEventRecord(ev[0], c->cstream));
    if (comm_allreduce_f32(c, k, c->grads, c->P, c->cstream)) return -1;
    if (timed) CBM_HIP(hipEventRecord(ev[1], c->cstream));
  }
  CBM_HIP(hipEventRecord(c->ext_ev, c->cstream));
  CBM_HIP(hipStreamWaitEvent(c->lstream, c->ext_ev, 0));
  if (timed) { CBM_HIP(hipEventRecord(ev[3], c->lstream)); c->comm_prof_n += 1; }
  return 0;
}
extern "C" int cbm_learner_allreduce_grads(cbm_ctx* c, float* grad_div) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  return cbm_learner_allreduce_grads_impl(c, grad_div);
}

// jax.lax.pmean of the loss statistics over the learners (ppo:649-653): SUM here, the host divides by the rank count when it reads them
int cbm_learner_allreduce_stats_impl(cbm_ctx* c) {
  CbmComm& k = c->comms[CBM_COMM_LEARNERS];
  if (!k.nranks) return 0;
  CBM_HIP(hipEventRecord(c->bwd_ev, c->lstream));
  CBM_HIP(hipStreamWaitEvent(c->cstream, c->bwd_ev, 0));
  if (comm_allreduce_f32(c, k, c->stats_dev, (int64_t)c->stat_rows * 8, c->cstream)) return -1;
  CBM_HIP(hipEventRecord(c->ext_ev, c->cstream));
  CBM_HIP(hipStreamWaitEvent(c->lstream, c->ext_ev, 0));
  return 0;
}

// timing of the gradient all-reduce for bench.py: per minibatch, (a) duration of the tail all-reduce on the communication stream and
// (b) the EXPOSED time on the learner stream = end of the backward pass -> optimizer may start
extern "C" int cbm_comm_profile(cbm_ctx* c, int32_t on) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (on && !c->comm_prof_created) {
    for (int i = 0; i < 4 * CBM_COMM_PROF_MAX; ++i) CBM_HIP(hipEventCreate(&c->comm_prof_ev[i]));
    c->comm_prof_created = true;
  }
  c->comm_prof_on = on != 0;
  c->comm_prof_n = 0;
  return 0;
}
extern "C" int cbm_comm_profile_read(cbm_ctx* c, double* tail_ms, double* exposed_ms, int32_t* count) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  CBM_HIP(hipStreamSynchronize(c->cstream));
  double a = 0.0, b = 0.0;
  for (int i = 0; i < c->comm_prof_n; ++i) {
    float ms = 0.0f;
    CBM_HIP(hipEventElapsedTime(&ms, c->comm_prof_ev[4 * i], c->comm_prof_ev[4 * i + 1]));
    a += ms;
    CBM_HIP(hipEventElapsedTime(&ms, c->comm_prof_ev[4 * i + 2], c->comm_prof_ev[4 * i + 3]));
    b += ms;
  }
  if (tail_ms) *tail_ms = a;
  if (exposed_ms) *exposed_ms = b;
  if (count) *count = c->comm_prof_n;
  c->comm_prof_n = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------ peer writes (split topologies)
extern "C" int cbm_ipc_export(cbm_ctx* c, const char* name, int32_t ring_index, uint8_t handle[CBM_IPC_HANDLE_BYTES]) {
  static_assert(sizeof(hipIpcMemHandle_t) == CBM_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  CBM_HIP(hipSetDevice(c->cfg.device));
  void* p = nullptr;
  int64_t n = 0;
  if (cbm_buffer(c, name, ring_index, &p, &n)) return -1;
  if (!p) { cbm_set_error("buffer '%s' is not allocated in this context", name); return -1; }
  hipIpcMemHandle_t h;
  CBM_HIP(hipIpcGetMemHandle(&h, p));
  memcpy(handle, &h, sizeof(h));
  return 0;
}
extern "C" int cbm_ipc_open(cbm_ctx* c, const uint8_t handle[CBM_IPC_HANDLE_BYTES], void** dev_ptr) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  const hipError_t e = hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    cbm_set_error("hipIpcOpenMemHandle failed on GPU %d: %s — the peer's buffer cannot be mapped here (needs HSA_ENABLE_IPC_MODE_LEGACY=0 in BOTH "
                  "processes and peer access between the two GPUs)", c->cfg.device, hipGetErrorString(e));
    return -1;
  }
  // The mapping exists.  Diagnostic only (never validated on a multi-GPU node, and ROCm may report an imported pointer's device
  // either way): when the runtime says this GPU has no peer path to the owner, say which two devices — once, on stderr — and keep the mapping;
  // if the path really is missing, the first copy through it fails with the runtime's own error and this hint is already in the log.
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, *dev_ptr) == hipSuccess && at.device != c->cfg.device) {
    int can = 1;
    static std::atomic<bool> warned{false};
    if (hipDeviceCanAccessPeer(&can, c->cfg.device, at.device) == hipSuccess && !can && !warned.exchange(true))
      fprintf(stderr, "cleanba_mi: hipDeviceCanAccessPeer(GPU %d -> GPU %d) = 0: the split topology writes shards / parameters peer to peer and needs an "
                      "xGMI or PCIe P2P path between every actor GPU and every learner GPU of a group\n", c->cfg.device, at.device);
  } else {
    (void)hipGetLastError();
  }
  return 0;
}
extern "C" int cbm_ipc_close(cbm_ctx* c, void* dev_ptr) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipIpcCloseMemHandle(dev_ptr));
  return 0;
}

// One learner's column shard of one committed rollout, written straight into that learner's ring entry (ppo:358-363: jnp.split along the
// env axis + device_put_sharded).  Source: columns [slot*E + li*El, +El) of every [T+1][B] field of the slot's current ring entry;
// destination: columns [dst_col0, +El) of fields with dst_cols columns per row (the learner's hstack of its slots' shards, ppo:587).  One
// strided 2-D copy per field on the io stream, ordered after the rollout's commit event; the call only enqueues (cbm_io_sync waits).
extern "C" int cbm_actor_ship_shard(cbm_ctx* c, int32_t slot, int32_t ring_index, int32_t li, int32_t n_learners, const cbm_peer_ring* dst,
                                    int32_t dst_cols, int32_t dst_col0) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (slot < 0 || slot >= c->S || ring_index < 0 || ring_index >= c->cfg.ring_depth) { cbm_set_error("bad slot / ring index"); return -1; }
  if (n_learners < 1 || c->E % n_learners || li < 0 || li >= n_learners) { cbm_set_error("local_num_envs must split evenly over the learners"); return -1; }
  const size_t El = (size_t)c->E / n_learners, col0 = (size_t)slot * c->E + (size_t)li * El, B = (size_t)c->Bdev, T1 = (size_t)c->T1;
  if (dst_col0 < 0 || (size_t)dst_col0 + El > (size_t)dst_cols) { cbm_set_error("destination columns [%d,+%zu) outside %d", dst_col0, El, dst_cols); return -1; }
  RingEntry& R = c->ring[ring_index];
  std::lock_guard<std::mutex> lk(c->io_mu);
  CBM_HIP(hipStreamWaitEvent(c->iostream, R.ready[slot], 0));
  struct F { void* d; const void* s; size_t elem; };
  const F fs[] = {{dst->obs, R.obs, CBM_FRAME}, {dst->actions, R.actions, 4}, {dst->logprobs, R.logprobs, 4}, {dst->values, R.values, 4},
                  {dst->rewards, R.rewards, 4}, {dst->dones, R.dones, 1}, {dst->firststeps, R.firststeps, 1},
                  {dst->logits, R.logits, (size_t)c->A * 4}};
  for (const F& f : fs) {
    if (!f.d) continue;   // a field the algorithm does not use (PPO: logits / firststeps; IMPALA: logprobs / values)
    CBM_HIP(hipMemcpy2DAsync((uint8_t*)f.d + (size_t)dst_col0 * f.elem, (size_t)dst_cols * f.elem, (const uint8_t*)f.s + col0 * f.elem, B * f.elem,
                             El * f.elem, T1, hipMemcpyDeviceToDevice, c->iostream));
  }
  return 0;
}
extern "C" int cbm_io_sync(cbm_ctx* c) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipStreamSynchronize(c->iostream));
  return 0;
}

// learner 0 -> its actor (ppo:721-725): parameter version v = (updates done) goes straight into the actor's version buffer v % 3, which
// the actor cannot be reading (rollout v+1 reads v-1, rollout v+2 waits for v).  Blocks until the write has landed.
extern "C" int cbm_params_push(cbm_ctx* c, void* const peer_versions[3]) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int v = c->updates_done;
  if (v < 1) { cbm_set_error("cbm_params_push before the first update"); return -1; }
  CBM_HIP(hipMemcpyAsync(peer_versions[v % NPV], c->params, (size_t)c->P * 4, hipMemcpyDeviceToDevice, c->lstream));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
// actor side: version updates_done+1 has been written into actor_params[v % 3] by the learner (and has landed)
extern "C" int cbm_params_mark_published(cbm_ctx* c) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int v = c->updates_done + 1;
  CBM_HIP(hipEventRecord(c->params_ready[v % NPV], c->lstream));   // the learner stream of an actor-only context is idle: completes at once
  cbm_publish(c, c->updates_done, v);
  return 0;
}
