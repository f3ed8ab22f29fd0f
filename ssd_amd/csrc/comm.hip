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
#include <cstdlib>

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
// (Rounds 3-5 kept a fenced form behind SSD_AR_FENCES=1 -- a system-scope release before the flag stores and an acquire after the
// poll; it validated identically in every two-process run and was never needed: deleted in round 6 with its switch.)
__device__ __forceinline__ void ar_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// W: ranks the peer loops are unrolled over (1 / 2 / 4 / 8, the smallest >= world): every rank's word is requested before the
// first add, and no rank is read twice (a first version unrolled over all 8 whatever the world size: at world = 1 the fused
// all-reduce + norm launch went from 7.6 to 11.1 us, profiles/r04_tp_shard_per_kind_v1.txt)
template <int W>
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
    counters[blk] = epoch;           // (default: no acquire fence -- every staged word is read with a system-scope load, see ar_publish_barrier)
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
    // every rank's word is requested before the first add (a rolled loop is load -> wait -> add per rank: `world` dependent
    // remote round trips); ranks past `world` re-read the last one and are not added -- same sum, same (rank) order
    unsigned long long v[W];
#pragma unroll
    for (int r = 0; r < W; ++r) v[r] = ld_sys64(peers.slot[min(r, world - 1)] + (long)(epoch & 1u) * slot_words + i);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int r = 0; r < W; ++r)
      if (r < world) {
        a0 += bf2f((unsigned)(v[r] & 0xffffu)); a1 += bf2f((unsigned)((v[r] >> 16) & 0xffffu));
        a2 += bf2f((unsigned)((v[r] >> 32) & 0xffffu)); a3 += bf2f((unsigned)(v[r] >> 48));
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
template <int ARN_THREADS, int W>
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
        unsigned long long v0[W], v1[W];      // all ranks' words in flight at once (see allreduce_bf16_kernel)
#pragma unroll
        for (int r = 0; r < W; ++r) {
          const unsigned long long* sp = peers.slot[min(r, world - 1)] + (long)(epoch & 1u) * slot_words + (long)row * hw + 2 * c;
          v0[r] = ld_sys64(sp); v1[r] = ld_sys64(sp + 1);
        }
        const u32x4_t rv = res_in[(size_t)row * H8 + c];
#pragma unroll
        for (int r = 0; r < W; ++r)
          if (r < world) {
            a[0] += bf2f((unsigned)(v0[r] & 0xffffu)); a[1] += bf2f((unsigned)((v0[r] >> 16) & 0xffffu));
            a[2] += bf2f((unsigned)((v0[r] >> 32) & 0xffffu)); a[3] += bf2f((unsigned)(v0[r] >> 48));
            a[4] += bf2f((unsigned)(v1[r] & 0xffffu)); a[5] += bf2f((unsigned)((v1[r] >> 16) & 0xffffu));
            a[6] += bf2f((unsigned)((v1[r] >> 32) & 0xffffu)); a[7] += bf2f((unsigned)(v1[r] >> 48));
          }
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

// ---------------------------------------------------------------------------------------------------------------------
// Data-tagged granules (round 4): the same collectives with ONE hop instead of three dependent ones.
//
// The protocol above is stage -> drain -> flag -> (peer) poll -> remote READ: three dependent uncached trips before a byte of
// the sum exists, and its correctness rests on "all my payload stores completed before my flag store is issued".  Here every
// rank PUSHES its values into each peer's inbox as naturally aligned 8-byte granules {2 bf16 payload, 32-bit epoch tag} with
// one write-through store each, and then polls ITS OWN inbox (local reads) until every granule it needs carries this call's
// epoch.  An 8-byte store is single-copy atomic, so a reader sees a granule whole or not at all: there is no ordering
// between separate payload and flag stores to get right, no drain, no fence, no remote read round trip -- the guide's
// handoff-1to1 primitive (MI355X_MICROARCH.md: 0.8-1.0 us idle per hop against 1.7-2.5x for payload + flag).  Cost: 2x the
// bytes on the wire (a [8, 8192] bf16 message is 256 KiB per peer instead of 128), which is why messages above GR_MAX_ELEMS
// keep the flag protocol.  Epochs, counters, launch width and the double-buffering argument are those of the flag protocol
// (a rank overwrites the parity-e inbox regions at call e + 2, by which time every peer has retired call e): the two
// protocols may be interleaved freely.  inbox[r] = rank r's inbox: [2 parities][AR_MAX_RANKS sources][gr_cap granules].
// ---------------------------------------------------------------------------------------------------------------------
struct GrPeers {
  unsigned long long* inbox[AR_MAX_RANKS];
};

__device__ __forceinline__ unsigned long long gr_make(unsigned int payload, unsigned int epoch) {
  return (unsigned long long)payload | ((unsigned long long)epoch << 32);
}

// poll NG granules of every peer source (own rank skipped) until all carry `epoch`; returns false on a timeout
template <int NG, int W>
__device__ __forceinline__ bool gr_collect(const unsigned long long* mybox, long src_stride, long g0, int rank, int world,
                                           unsigned int epoch, long spin_budget, unsigned long long (&v)[W][NG]) {
  long spins = 0;
  while (true) {
    bool ok = true;
    if (W == 1) return true;        // no peer
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const int rr = min(r, world - 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) v[r][g] = ld_sys64(mybox + (long)rr * src_stride + g0 + g);
    }
#pragma unroll
    for (int r = 0; r < W; ++r)
      if (r < world && r != rank) {
#pragma unroll
        for (int g = 0; g < NG; ++g) ok = ok && ((unsigned int)(v[r][g] >> 32) == epoch);
      }
    if (ok) return true;
    if (spins < 4096) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
    if (++spins > spin_budget) return false;
  }
}

template <int W>
__global__ void __launch_bounds__(AR_THREADS)
allreduce_gr_kernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out, long n8, long gr_cap,
                    GrPeers peers, int rank, int world, unsigned int* __restrict__ counters, unsigned int* __restrict__ err,
                    long spin_budget) {
  __shared__ unsigned int s_epoch;
  __shared__ int s_fail;
  const int blk = blockIdx.x;
  const long per = (n8 + gridDim.x - 1) / gridDim.x;
  const long i0 = blk * per, i1 = min(n8, i0 + per);
  // this call's input words are requested before the epoch is known (the counter read is a dependent round trip of its own)
  constexpr int MAXI = 4;        // words per thread: n8 <= GR_MAX_ELEMS / 4 = 16384 over 8 x 512 threads
  unsigned long long w[MAXI];
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const long i = i0 + threadIdx.x + (long)k * AR_THREADS;
    w[k] = i < i1 ? in[i] : 0ull;
  }
  if (threadIdx.x == 0) { s_epoch = counters[blk] + 1; s_fail = 0; }
  __syncthreads();
  const unsigned int epoch = s_epoch;
  const long par = (long)(epoch & 1u) * AR_MAX_RANKS * gr_cap;
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const long i = i0 + threadIdx.x + (long)k * AR_THREADS;
    if (i < i1) {
      const unsigned long long g_lo = gr_make((unsigned int)w[k], epoch), g_hi = gr_make((unsigned int)(w[k] >> 32), epoch);
      for (int pr = 0; pr < world; ++pr)
        if (pr != rank) {
          unsigned long long* dst = peers.inbox[pr] + par + (long)rank * gr_cap + 2 * i;
          st_sys64(dst, g_lo);
          st_sys64(dst + 1, g_hi);
        }
    }
  }
  const unsigned long long* mybox = peers.inbox[rank] + par;
  bool fail = false;
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const long i = i0 + threadIdx.x + (long)k * AR_THREADS;
    if (i < i1 && !fail) {
      unsigned long long v[W][2];
      if (!gr_collect<2, W>(mybox, gr_cap, 2 * i, rank, world, epoch, spin_budget, v)) { fail = true; break; }
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int r = 0; r < W; ++r)
        if (r < world) {
          const unsigned int lo = r == rank ? (unsigned int)w[k] : (unsigned int)v[r][0];
          const unsigned int hi = r == rank ? (unsigned int)(w[k] >> 32) : (unsigned int)v[r][1];
          a0 += bf2f(lo & 0xffffu); a1 += bf2f(lo >> 16);
          a2 += bf2f(hi & 0xffffu); a3 += bf2f(hi >> 16);
        }
      out[i] = (unsigned long long)pack_bf2(a0, a1) | ((unsigned long long)pack_bf2(a2, a3) << 32);
    }
  }
  if (fail) s_fail = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    counters[blk] = epoch;
    if (s_fail) atomicExch(err, 1u);
  }
}

template <int ARN_THREADS, int W>
__global__ void __launch_bounds__(ARN_THREADS)
allreduce_add_rmsnorm_gr_kernel(const unsigned long long* __restrict__ in, const u32x4_t* __restrict__ res_in,
                                u32x4_t* __restrict__ res_out, const u32x4_t* __restrict__ w, float eps,
                                u32x4_t* __restrict__ out_rows, u32x4_t* __restrict__ out_frag, int T, int H, long gr_cap,
                                GrPeers peers, int rank, int world, unsigned int* __restrict__ counters,
                                unsigned int* __restrict__ err, long spin_budget) {
  __shared__ unsigned int s_epoch;
  __shared__ int s_fail;
  __shared__ float red[ARN_THREADS / 64];
  const int blk = blockIdx.x;
  const int rpb = (T + gridDim.x - 1) / gridDim.x;
  const int r0 = blk * rpb, r1 = min(T, r0 + rpb);
  const int H8 = H >> 3, KT = H >> 5;
  const long hw = H >> 2;                       // 8-byte words per row
  constexpr int ARN_MAXC = ARN_MAXH / 8 / ARN_THREADS;
  if (threadIdx.x == 0) { s_epoch = counters[blk] + 1; s_fail = 0; }
  __syncthreads();
  const unsigned int epoch = s_epoch;
  const long par = (long)(epoch & 1u) * AR_MAX_RANKS * gr_cap;
  const unsigned long long* mybox = peers.inbox[rank] + par;
  // ---- push: every chunk of every row this workgroup owns, to every peer ----
  for (int row = r0; row < r1; ++row) {
#pragma unroll
    for (int i = 0; i < ARN_MAXC; ++i) {
      const int c = threadIdx.x + i * ARN_THREADS;
      if (c < H8) {
        const unsigned long long* src = in + (long)row * hw + 2 * c;
        const unsigned long long x0 = src[0], x1 = src[1];
        const unsigned long long g[4] = {gr_make((unsigned int)x0, epoch), gr_make((unsigned int)(x0 >> 32), epoch),
                                         gr_make((unsigned int)x1, epoch), gr_make((unsigned int)(x1 >> 32), epoch)};
        for (int pr = 0; pr < world; ++pr)
          if (pr != rank) {
            unsigned long long* dst = peers.inbox[pr] + par + (long)rank * gr_cap + ((long)row * H8 + c) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) st_sys64(dst + q, g[q]);
          }
      }
    }
  }
  // ---- collect + add + norm, row by row (arithmetic and order of allreduce_add_rmsnorm_kernel) ----
  bool fail = false;
  for (int row = r0; row < r1; ++row) {
    float v[ARN_MAXC][8];
    u32x4_t wreg[ARN_MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < ARN_MAXC; ++i) {
      const int c = threadIdx.x + i * ARN_THREADS;
      if (c < H8 && !fail) {
        wreg[i] = w[c];
        const u32x4_t rv = res_in[(size_t)row * H8 + c];
        const unsigned long long* src = in + (long)row * hw + 2 * c;
        const unsigned long long x0 = src[0], x1 = src[1];
        unsigned long long gv[W][4];
        if (!gr_collect<4, W>(mybox, gr_cap, ((long)row * H8 + c) * 4, rank, world, epoch, spin_budget, gv)) { fail = true; }
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < W; ++r)
          if (r < world) {
            unsigned int u[4];
            if (r == rank) { u[0] = (unsigned int)x0; u[1] = (unsigned int)(x0 >> 32); u[2] = (unsigned int)x1; u[3] = (unsigned int)(x1 >> 32); }
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) u[q] = (unsigned int)gv[r][q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[2 * q] += bf2f(u[q] & 0xffffu); a[2 * q + 1] += bf2f(u[q] >> 16); }
          }
        u32x4_t ro;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
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
        const u32x4_t wv = wreg[i];
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
  if (fail) s_fail = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    counters[blk] = epoch;
    if (s_fail) atomicExch(err, 1u);
  }
}

// ---- host-side helpers (setup time only; never called on the hot path) ----
extern "C" int ssd_comm_alloc(void** out, long bytes) {
  if (!out || bytes <= 0) return SSD_ERR_ARG;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) return SSD_ERR_LAUNCH;
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess) return SSD_ERR_LAUNCH;
  if (hipDeviceSynchronize() != hipSuccess) return SSD_ERR_LAUNCH;
  *out = p;
  return SSD_OK;
}
extern "C" int ssd_comm_free(void* p) {
  if (!p) return SSD_ERR_ARG;
  return hipFree(p) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
extern "C" int ssd_comm_ipc_export(void* p, void* handle64) {
  if (!p || !handle64) return SSD_ERR_ARG;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) return SSD_ERR_LAUNCH;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
  memcpy(handle64, &h, 64);
  return SSD_OK;
}
extern "C" int ssd_comm_ipc_open(const void* handle64, void** out) {
  if (!handle64 || !out) return SSD_ERR_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return SSD_ERR_LAUNCH;
  *out = p;
  return SSD_OK;
}
extern "C" int ssd_comm_ipc_close(void* p) {
  if (!p) return SSD_ERR_ARG;
  return hipIpcCloseMemHandle(p) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

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
#define AR_GO(WV)                                                                                                        \
  hipLaunchKernelGGL(allreduce_bf16_kernel<WV>, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream,              \
                     (const unsigned long long*)in, (unsigned long long*)out, n8, slot_elems / 4, peers, rank, world,   \
                     (unsigned int*)counters, (unsigned int*)err, spin_budget, 0)
  if (world == 1) AR_GO(1); else if (world == 2) AR_GO(2); else if (world <= 4) AR_GO(4); else AR_GO(8);
#undef AR_GO
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
#define AR_GO(WV)                                                                                                        \
  hipLaunchKernelGGL(allreduce_bf16_kernel<WV>, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream,              \
                     (const unsigned long long*)in, (unsigned long long*)out, n8, slot_elems / 4, peers, rank, world,   \
                     (unsigned int*)counters, (unsigned int*)err, spin_budget, 1)
  if (world == 1) AR_GO(1); else if (world == 2) AR_GO(2); else if (world <= 4) AR_GO(4); else AR_GO(8);
#undef AR_GO
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
#define ARN_GO(TH, WV)                                                                                                   \
  hipLaunchKernelGGL((allreduce_add_rmsnorm_kernel<TH, WV>), dim3(AR_BLOCKS), dim3(TH), 0, (hipStream_t)stream,           \
                     (const unsigned long long*)in, (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, \
                     (u32x4_t*)out_rows, (u32x4_t*)out_frag, T, H, slot_elems / 4, peers, rank, world,                    \
                     (unsigned int*)counters, (unsigned int*)err, spin_budget)
#define ARN_W(TH) do { if (world == 1) ARN_GO(TH, 1); else if (world == 2) ARN_GO(TH, 2); else if (world <= 4) ARN_GO(TH, 4); else ARN_GO(TH, 8); } while (0)
  if (ssd_norm_threads(H) == 1024) ARN_W(1024); else ARN_W(256);
#undef ARN_W
#undef ARN_GO
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// ---- data-tagged granule variants (see the comment above allreduce_gr_kernel).  inboxes[r] = rank r's inbox (own allocation for
// `rank`, IPC-opened mappings for the peers): 2 * AR_MAX_RANKS * gr_cap granules of 8 bytes, zero-initialised; gr_cap = granules
// per (parity, source) region >= n / 2.  counters / err as above (the SAME counters as the flag protocol's calls).
extern "C" int ssd_allreduce_gr_bf16(const void* in, void* out, long n, int rank, int world, void* const* inboxes, long gr_cap,
                                     void* counters, void* err, long spin_budget, void* stream) {
  if (world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || n <= 0 || (n & 3) || n / 2 > gr_cap ||
      n / 4 > 4L * AR_BLOCKS * AR_THREADS)
    return SSD_ERR_SHAPE;
  GrPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) peers.inbox[r] = (unsigned long long*)(r < world ? inboxes[r] : nullptr);
#define GR_GO(WV)                                                                                                        \
  hipLaunchKernelGGL(allreduce_gr_kernel<WV>, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream,                \
                     (const unsigned long long*)in, (unsigned long long*)out, n / 4, gr_cap, peers, rank, world,        \
                     (unsigned int*)counters, (unsigned int*)err, spin_budget)
  if (world == 1) GR_GO(1); else if (world == 2) GR_GO(2); else if (world <= 4) GR_GO(4); else GR_GO(8);
#undef GR_GO
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_allreduce_add_rmsnorm_gr_bf16(const void* in, const void* res_in, void* res_out, const void* weight, float eps,
                                                 void* out_rows, void* out_frag, int T, int H, int rank, int world,
                                                 void* const* inboxes, long gr_cap, void* counters, void* err, long spin_budget,
                                                 void* stream) {
  if (world < 1 || world > AR_MAX_RANKS || rank < 0 || rank >= world || T <= 0 || H <= 0 || (H & 31) || H > ARN_MAXH ||
      (long)T * H / 2 > gr_cap)
    return SSD_ERR_SHAPE;
  if (!in || !res_in || !res_out || !weight) return SSD_ERR_ARG;
  GrPeers peers;
  for (int r = 0; r < AR_MAX_RANKS; ++r) peers.inbox[r] = (unsigned long long*)(r < world ? inboxes[r] : nullptr);
#define GRN_GO(TH, WV)                                                                                                   \
  hipLaunchKernelGGL((allreduce_add_rmsnorm_gr_kernel<TH, WV>), dim3(AR_BLOCKS), dim3(TH), 0, (hipStream_t)stream,        \
                     (const unsigned long long*)in, (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, \
                     (u32x4_t*)out_rows, (u32x4_t*)out_frag, T, H, gr_cap, peers, rank, world, (unsigned int*)counters,     \
                     (unsigned int*)err, spin_budget)
#define GRN_W(TH) do { if (world == 1) GRN_GO(TH, 1); else if (world == 2) GRN_GO(TH, 2); else if (world <= 4) GRN_GO(TH, 4); else GRN_GO(TH, 8); } while (0)
  if (ssd_norm_threads(H) == 1024) GRN_W(1024); else GRN_W(256);
#undef GRN_W
#undef GRN_GO
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
