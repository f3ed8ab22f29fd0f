// One-shot full-mesh all-reduce for the latency-bound tensor-parallel sums of the decode path.
//
// The reference all-reduces a [T, hidden] bf16 tensor twice per layer through NCCL (ssd/layers/linear.py:195-199,
// ssd/layers/embed_head.py:53-56): 161 collectives of <= 128 KiB per 70B forward.  A ring is the wrong shape for that
// on MI355X: xGMI is a full mesh (7 links per GPU), so every rank can read all peers' buffers at once and reduce
// locally -- one hop, no serialized ring steps, and a fixed rank order makes the result bitwise identical on every
// rank and run.
//
// Mechanism (per call, `epoch` = call counter kept on the device, so hipGraph replays stay in step).  EVERY collective
// of this file -- plain all-reduce, fused all-reduce + norm, all-gather -- launches exactly AR_BLOCKS workgroups and
// every workgroup takes part in the flag exchange of every call, whether or not the call's partition gives it data:
// the AR_BLOCKS per-workgroup counters therefore all equal the number of collectives issued so far (one global epoch),
// however the message sizes and partitions of consecutive calls differ.
//   1. copy my input slice into my staging slot (epoch & 1) with system-scope stores, system-scope release;
//   2. write `epoch` into my entry of every peer's flag array (remote stores over xGMI);
//   3. spin (bounded) until every peer's entry in MY flag array reaches `epoch`, acquire;
//   4. read all ranks' slots (remote loads), sum in fp32 in rank order, write bf16 to the output.
// Slots are double-buffered by epoch parity: a rank can only overwrite slot s at call e+2 after passing the flag
// exchange of call e+1, which a peer's workgroup enters only after that peer's WHOLE kernel of call e has retired
// (kernels of one stream do not overlap) -- so the argument does not depend on which workgroup owned which words in
// call e, only on all workgroups agreeing on the epoch, which the fixed launch width guarantees.
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

// Publish protocol (guide: "payload write-through, drained, then ONE flag store; the consumer polls the flag and reads the
// payload with cache-bypassing loads"): the staged words are system-scope (write-through, sc0 sc1) stores and every reader
// uses system-scope loads, so no cache holds a stale or an unwritten copy on either side -- what must be ordered is only
// "all my payload stores have completed" before "my flag store is issued": every storing wave drains its store queue
// (s_waitcnt vmcnt(0), as inline asm so the compiler cannot drop or move it), the workgroup meets, then the flags go out.
// No release fence (an L2 write-back of everything the preceding GEMM left dirty) and no acquire fence (an L2 invalidate
// that the NEXT GEMM would pay for): measured on MI355X with the TP = 4 shard shapes, the fused all-reduce + norm launch
// went from 7.6 us to the figure in profiles/r03_tp_shard_per_kind.txt.
__device__ __forceinline__ void ar_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
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
  ar_publish_barrier();
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
    counters[blk] = epoch;           // (no acquire fence: every staged word is read with a system-scope load, see ar_publish_barrier)
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

// Fused: all-reduce of this rank's [T][H] partial (o_proj / down_proj output) + residual add + RMSNorm of the sum --
// the three launches RowParallelLinear.forward's all_reduce (linear.py:195-199) + RMSDNorm.add_norm_forward
// (layernorm.py:76-88) cost per half layer become one.  A workgroup owns whole rows (the norm needs the full row), so
// at most AR_BLOCKS workgroups read the peers' staged rows; per row the arithmetic and the reduction order are exactly
// those of allreduce_bf16_kernel followed by rmsnorm_kernel (norm.hip) -- results are bit-identical to the unfused pair.
constexpr int ARN_MAXH = 16384;    // = NORM_MAXH; ARN_THREADS = ssd_norm_threads(H): same chunk -> thread mapping, same
                                   // sum-of-squares order as rmsnorm_kernel<THREADS> (norm.hip)
template <int ARN_THREADS>
__global__ void __launch_bounds__(ARN_THREADS)
allreduce_add_rmsnorm_kernel(const unsigned long long* __restrict__ in, const u32x4_t* __restrict__ res_in,
                             u32x4_t* __restrict__ res_out, const u32x4_t* __restrict__ w, float eps,
                             u32x4_t* __restrict__ out_rows, u32x4_t* __restrict__ out_frag, int T, int H,
                             long slot_words, ArPeers peers, int rank, int world, unsigned int* __restrict__ counters,
                             unsigned int* __restrict__ err, long spin_budget) {
  __shared__ unsigned int s_epoch;
  __shared__ int s_fail;
  __shared__ float red[ARN_THREADS / 64];
  const int blk = blockIdx.x;
  if (threadIdx.x == 0) { s_epoch = counters[blk] + 1; s_fail = 0; }
  __syncthreads();
  const unsigned int epoch = s_epoch;
  const int rpb = (T + gridDim.x - 1) / gridDim.x;
  const int r0 = blk * rpb, r1 = min(T, r0 + rpb);
  const int H8 = H >> 3, KT = H >> 5;
  const long hw = H >> 2;                       // 8-byte words per row
  unsigned long long* my = peers.slot[rank] + (long)(epoch & 1u) * slot_words;
  for (long i = (long)r0 * hw + threadIdx.x; i < (long)r1 * hw; i += ARN_THREADS) st_sys64(my + i, in[i]);
  ar_publish_barrier();
  if (threadIdx.x < world) {
    const int peer = threadIdx.x;
    __hip_atomic_store(peers.flags[peer] + blk * AR_MAX_RANKS + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int* mine = peers.flags[rank] + blk * AR_MAX_RANKS + peer;
    long spins = 0;
    while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (spins < 4096) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
      if (++spins > spin_budget) { s_fail = 1; break; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counters[blk] = epoch;
    if (s_fail) atomicExch(err, 1u);
  }
  __syncthreads();
  if (s_fail) return;
  for (int row = r0; row < r1; ++row) {
    constexpr int ARN_MAXC = ARN_MAXH / 8 / ARN_THREADS;
    float v[ARN_MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < ARN_MAXC; ++i) {
      const int c = threadIdx.x + i * ARN_THREADS;
      if (c < H8) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
          const unsigned long long* sp = peers.slot[r] + (long)(epoch & 1u) * slot_words + (long)row * hw + 2 * c;
          const unsigned long long v0 = ld_sys64(sp), v1 = ld_sys64(sp + 1);
          a[0] += bf2f((unsigned)(v0 & 0xffffu)); a[1] += bf2f((unsigned)((v0 >> 16) & 0xffffu));
          a[2] += bf2f((unsigned)((v0 >> 32) & 0xffffu)); a[3] += bf2f((unsigned)(v0 >> 48));
          a[4] += bf2f((unsigned)(v1 & 0xffffu)); a[5] += bf2f((unsigned)((v1 >> 16) & 0xffffu));
          a[6] += bf2f((unsigned)((v1 >> 32) & 0xffffu)); a[7] += bf2f((unsigned)(v1 >> 48));
        }
        const u32x4_t rv = res_in[(size_t)row * H8 + c];
        u32x4_t ro;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the all-reduce delivers bf16 (one rounding of the fp32 rank-order sum); the norm then works in fp32
          const float lo = round_bf(a[2 * j]) + bf2f(rv[j] & 0xffffu);
          const float hi = round_bf(a[2 * j + 1]) + bf2f(rv[j] >> 16);
          v[i][2 * j] = lo; v[i][2 * j + 1] = hi;
          ro[j] = pack_bf2(lo, hi);
          ss += lo * lo; ss += hi * hi;
        }
        res_out[(size_t)row * H8 + c] = ro;
      }
    }
    ss = wave_sum(ss);
    __syncthreads();                       // red[] of the previous row has been consumed
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < ARN_THREADS / 64; ++i) tot += red[i];
    const float rs = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
    for (int i = 0; i < ARN_MAXC; ++i) {
      const int c = threadIdx.x + i * ARN_THREADS;
      if (c < H8) {
        const u32x4_t wv = w[c];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = (v[i][2 * j] * rs) * bf2f(wv[j] & 0xffffu);
          const float hi = (v[i][2 * j + 1] * rs) * bf2f(wv[j] >> 16);
          o[j] = pack_bf2(lo, hi);
        }
        if (out_rows) out_rows[(size_t)row * H8 + c] = o;
        if (out_frag) out_frag[frag_chunk(row, c, KT)] = o;
      }
    }
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
  hipLaunchKernelGGL(allreduce_bf16_kernel, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream,
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
  hipLaunchKernelGGL(allreduce_bf16_kernel, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream,
                     (const unsigned long long*)in, (unsigned long long*)out, n8, slot_elems / 4, peers, rank, world,
                     (unsigned int*)counters, (unsigned int*)err, spin_budget, 1);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// All-reduce of `in` ([T][H] bf16 partial sums) fused with the residual add and the RMSNorm that follow it in every
// decoder half-layer: res_out = bf16(allreduce(in) + res_in); out = bf16(x32 * rsqrt(mean(x32^2) + eps) * weight), written
// row-major (out_rows) and/or fragment-major (out_frag).  Same staging slots / flags / counters as ssd_allreduce_bf16
// (the calls may be interleaved freely as long as every rank issues the same sequence).  T*H <= slot_elems, H % 32 == 0.
extern "C" int ssd_allreduce_add_rmsnorm_bf16(const void* in, const void* res_in, void* res_out, const void* weight,
                                              float eps, void* out_rows, void* out_frag, int T, int H, int rank, int world,
                                              void* const* slots, void* const* flags, long slot_elems, void* counters,
                                              void* err, long spin_budget, void* stream) {
  if (world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || T <= 0 || H <= 0 || (H & 31) ||
      H > ARN_MAXH || (long)T * H > slot_elems)
    return SSD_ERR_SHAPE;
  if (!in || !res_in || !res_out || !weight) return SSD_ERR_ARG;
  ArPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) {
    peers.slot[r] = (unsigned long long*)(r < world ? slots[r] : nullptr);
    peers.flags[r] = (unsigned int*)(r < world ? flags[r] : nullptr);
  }
  if (ssd_norm_threads(H) == 1024) {
    hipLaunchKernelGGL(allreduce_add_rmsnorm_kernel<1024>, dim3(AR_BLOCKS), dim3(1024), 0, (hipStream_t)stream,
                       (const unsigned long long*)in, (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps,
                       (u32x4_t*)out_rows, (u32x4_t*)out_frag, T, H, slot_elems / 4, peers, rank, world,
                       (unsigned int*)counters, (unsigned int*)err, spin_budget);
  } else {
    hipLaunchKernelGGL(allreduce_add_rmsnorm_kernel<256>, dim3(AR_BLOCKS), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)in, (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps,
                       (u32x4_t*)out_rows, (u32x4_t*)out_frag, T, H, slot_elems / 4, peers, rank, world,
                       (unsigned int*)counters, (unsigned int*)err, spin_budget);
  }
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
