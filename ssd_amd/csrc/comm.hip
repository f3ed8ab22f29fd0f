// One-shot full-mesh all-reduce for the latency-bound tensor-parallel sums of the decode path.
//
// The reference all-reduces a [T, hidden] bf16 tensor twice per layer through NCCL (ssd/layers/linear.py:195-199,
// ssd/layers/embed_head.py:53-56): 161 collectives of <= 128 KiB per 70B forward.  A ring is the wrong shape for that
// on MI355X: xGMI is a full mesh (7 links per GPU), so every rank can read all peers' buffers at once and reduce
// locally -- one hop, no serialized ring steps, and a fixed rank order makes the result bitwise identical on every
// rank and run.
//
// Mechanism (per call, `epoch` = per-workgroup call counter kept on the device, so hipGraph replays stay in step):
//   1. copy my input slice into my staging slot (epoch & 1) with system-scope stores, system-scope release;
//   2. write `epoch` into my entry of every peer's flag array (remote stores over xGMI);
//   3. spin (bounded) until every peer's entry in MY flag array reaches `epoch`, acquire;
//   4. read all ranks' slots (remote loads), sum in fp32 in rank order, write bf16 to the output.
// Slots are double-buffered by epoch parity: a rank can only overwrite slot s at call e+2 after passing the flag
// exchange of call e+1, which every peer enters only after finishing its reads of call e.
// Buffers are fine-grained device memory shared through hipIpc handles; nothing here allocates per call.
// A spin that exceeds its budget sets an error word instead of hanging; the caller then falls back to RCCL.
#include "common.h"
#include <cstring>

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_BLOCKS = 8;
constexpr int AR_THREADS = 512;

struct ArPeers {
  unsigned long long* slot[AR_MAX_RANKS];   // each rank's staging area: 2 slots of slot_elems bf16 (as 8-byte words)
  unsigned int* flags[AR_MAX_RANKS];        // each rank's flag array [AR_BLOCKS][AR_MAX_RANKS]
};

__device__ __forceinline__ void st_sys64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(AR_THREADS)
allreduce_bf16_kernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out, long n8,
                      long slot_words, ArPeers peers, int rank, int world, unsigned int* __restrict__ counters,
                      unsigned int* __restrict__ err, long spin_budget, int gather) {
  __shared__ unsigned int s_epoch;
  __shared__ int s_fail;
  const int blk = blockIdx.x;
  if (threadIdx.x == 0) { s_epoch = counters[blk] + 1; s_fail = 0; }
  __syncthreads();
  const unsigned int epoch = s_epoch;
  const long per = (n8 + gridDim.x - 1) / gridDim.x;
  const long i0 = blk * per, i1 = min(n8, i0 + per);
  unsigned long long* my = peers.slot[rank] + (long)(epoch & 1u) * slot_words;
  for (long i = i0 + threadIdx.x; i < i1; i += AR_THREADS) st_sys64(my + i, in[i]);
  __syncthreads();
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system-scope release
  __syncthreads();
  if (threadIdx.x < world) {
    const int peer = threadIdx.x;
    __hip_atomic_store(peers.flags[peer] + blk * AR_MAX_RANKS + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int* mine = peers.flags[rank] + blk * AR_MAX_RANKS + peer;
    long spins = 0;
    // epochs are compared with wrap-around-safe signed difference
    while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      // fast polls first (latency), then ~1 us polls: ranks can be skewed by SECONDS of host work (hipGraph capture,
      // Python GC), which must not read as a failure -- the budget only exists so that a dead peer cannot hang the GPU
      if (spins < 4096) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
      if (++spins > spin_budget) { s_fail = 1; break; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                                  // system-scope acquire
    counters[blk] = epoch;
    if (s_fail) atomicExch(err, 1u);
  }
  __syncthreads();
  if (s_fail) return;
  if (gather) {   // all-gather of opaque 8-byte words: out[r][i] = rank r's in[i]
    for (long i = i0 + threadIdx.x; i < i1; i += AR_THREADS)
      for (int r = 0; r < world; ++r) out[(long)r * n8 + i] = ld_sys64(peers.slot[r] + (long)(epoch & 1u) * slot_words + i);
    return;
  }
  for (long i = i0 + threadIdx.x; i < i1; i += AR_THREADS) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int r = 0; r < world; ++r) {
      const unsigned long long v = ld_sys64(peers.slot[r] + (long)(epoch & 1u) * slot_words + i);
      a0 += bf2f((unsigned)(v & 0xffffu)); a1 += bf2f((unsigned)((v >> 16) & 0xffffu));
      a2 += bf2f((unsigned)((v >> 32) & 0xffffu)); a3 += bf2f((unsigned)(v >> 48));
    }
    out[i] = (unsigned long long)pack_bf2(a0, a1) | ((unsigned long long)pack_bf2(a2, a3) << 32);
  }
}

// ---- host-side helpers (setup time only; never called on the hot path) ----
extern "C" int ssd_comm_alloc(void** out, long bytes) {
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) return SSD_ERR_LAUNCH;
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess) return SSD_ERR_LAUNCH;
  if (hipDeviceSynchronize() != hipSuccess) return SSD_ERR_LAUNCH;
  *out = p;
  return SSD_OK;
}
extern "C" int ssd_comm_free(void* p) { return hipFree(p) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH; }
extern "C" int ssd_comm_ipc_export(void* p, void* handle64) {
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) return SSD_ERR_LAUNCH;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
  memcpy(handle64, &h, 64);
  return SSD_OK;
}
extern "C" int ssd_comm_ipc_open(const void* handle64, void** out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return SSD_ERR_LAUNCH;
  *out = p;
  return SSD_OK;
}
extern "C" int ssd_comm_ipc_close(void* p) { return hipIpcCloseMemHandle(p) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH; }

// in / out: bf16 [n] device buffers (n % 4 == 0, 8-byte aligned; in == out allowed); slots / flags: per-rank pointers
// (own allocation for `rank`, IPC-opened mappings for the peers); slot_elems: capacity of one staging slot in bf16
// elements; counters: uint32[AR_BLOCKS] zero-initialised device memory; err: uint32 device word, 1 after a spin timeout.
extern "C" int ssd_allreduce_bf16(const void* in, void* out, long n, int rank, int world, void* const* slots,
                                  void* const* flags, long slot_elems, void* counters, void* err, long spin_budget,
                                  void* stream) {
  if (world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || n <= 0 || (n & 3) || n > slot_elems) return SSD_ERR_SHAPE;
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) {
    peers.slot[r] = (unsigned long long*)(r < world ? slots[r] : nullptr);
    peers.flags[r] = (unsigned int*)(r < world ? flags[r] : nullptr);
  }
  const long n8 = n / 4;
  int blocks = (int)((n8 + 2047) / 2048);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_BLOCKS) blocks = AR_BLOCKS;
  hipLaunchKernelGGL(allreduce_bf16_kernel, dim3(blocks), dim3(AR_THREADS), 0, (hipStream_t)stream,
                     (const unsigned long long*)in, (unsigned long long*)out, n8, slot_elems / 4, peers, rank, world,
                     (unsigned int*)counters, (unsigned int*)err, spin_budget, 0);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// All-gather of n8 opaque 8-byte words per rank through the same staging slots / flags / counters:
// out [world][n8] (in must not alias out).  Used for the vocab-parallel argmax (value, index) exchange.
extern "C" int ssd_allgather_u64(const void* in, void* out, long n8, int rank, int world, void* const* slots,
                                 void* const* flags, long slot_elems, void* counters, void* err, long spin_budget,
                                 void* stream) {
  if (world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || n8 <= 0 || n8 * 4 > slot_elems) return SSD_ERR_SHAPE;
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) {
    peers.slot[r] = (unsigned long long*)(r < world ? slots[r] : nullptr);
    peers.flags[r] = (unsigned int*)(r < world ? flags[r] : nullptr);
  }
  int blocks = (int)((n8 + 2047) / 2048);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_BLOCKS) blocks = AR_BLOCKS;
  hipLaunchKernelGGL(allreduce_bf16_kernel, dim3(blocks), dim3(AR_THREADS), 0, (hipStream_t)stream,
                     (const unsigned long long*)in, (unsigned long long*)out, n8, slot_elems / 4, peers, rank, world,
                     (unsigned int*)counters, (unsigned int*)err, spin_budget, 1);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
