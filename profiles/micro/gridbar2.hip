// Device-wide barrier inside a persistent kernel, second look (round 4).  gridbar.hip polled with ACQUIRE loads and arrived with a
// RELEASE add: on gfx950 every such poll carries a `buffer_inv sc1` (an L2 invalidate) and measured 10-30 us per barrier.  The
// norm-in-GEMM experiment (profiles/r04_nig_probe_v*.txt) showed that the invalidate is the cost, not the flag.  Here:
//   arrive  = (optional release fence) + RELAXED agent-scope add
//   wait    = RELAXED agent-scope (sc1) polls by one lane of the workgroup
//   data    = exchanged through agent-scope relaxed 4-byte loads / stores (sc1: coherent across the XCDs' L2s, no fence needed)
// Modes:  0 barrier only;  1 + each workgroup publishes 16 B and every workgroup reads the whole vector (gridDim * 16 B) after the
// barrier (checked);  2 = 1 + every workgroup streams `wkb` KiB of weights per round, the loads ISSUED BEFORE the wait and consumed
// after it (does the stream hide under the barrier?);  3 = 2 but the loads are issued AFTER the barrier (the cold start a kernel
// boundary forces).  Against: a hipGraph of `rounds` dependent kernels doing the same per-round work.
// hipcc --offload-arch=gfx950 -O3 -o gridbar2 gridbar2.hip && ./gridbar2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, AGENT);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, AGENT) < target) {
      if (++spins > 2000000L) { atomicExch(err, 1u); break; }
    }
  }
  __syncthreads();
  return true;
}

template <int MODE>
__global__ void __launch_bounds__(256) persistent(unsigned* counter, unsigned* vec /* [2][grid*4] */, const u32x4* __restrict__ W,
                                                  int wchunks /* 16-B chunks per thread per round */, int rounds, unsigned* err,
                                                  unsigned* sink) {
  const unsigned G = gridDim.x;
  unsigned acc = 0;
  u32x4 wacc = {0, 0, 0, 0};
  const size_t per_round = (size_t)G * blockDim.x * wchunks;
  for (int r = 0; r < rounds; ++r) {
    unsigned* cur = vec + (size_t)(r & 1) * G * 4;
    const u32x4* wp = W + (size_t)r * per_round + ((size_t)blockIdx.x * wchunks * blockDim.x) + threadIdx.x;
    u32x4 wv[8];
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i < wchunks) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * blockDim.x);
    }
    if (MODE >= 1 && threadIdx.x < 4)
      __hip_atomic_store(cur + blockIdx.x * 4 + threadIdx.x, (unsigned)(r * 131 + blockIdx.x * 4 + threadIdx.x), __ATOMIC_RELAXED, AGENT);
    if (MODE >= 1) __builtin_amdgcn_s_waitcnt(0);      // (stores issued; the counter add below is ordered behind them by vmcnt -- see note)
    grid_barrier(counter, (unsigned)(r + 1) * G, err);
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i < wchunks) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * blockDim.x);
    }
    if (MODE >= 1) {
      for (unsigned i = threadIdx.x; i < G * 4; i += blockDim.x) {
        const unsigned v = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, AGENT);
        if (v != (unsigned)(r * 131 + i)) atomicExch(err, 2u);
        acc += v;
      }
    }
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i < wchunks) wacc ^= wv[i];
    }
  }
  if (acc == 0x12345u || (wacc[0] ^ wacc[1] ^ wacc[2] ^ wacc[3]) == 0x777u) *sink = acc;
}

// the same round as one kernel of a graph: read the previous round's vector (plain loads: the boundary made it visible), stream the
// weights, publish this round's piece
template <int MODE>
__global__ void __launch_bounds__(256) round_kernel(unsigned* vec, const u32x4* __restrict__ W, int wchunks, int r, unsigned* err,
                                                    unsigned* sink) {
  const unsigned G = gridDim.x;
  unsigned acc = 0;
  u32x4 wacc = {0, 0, 0, 0};
  const size_t per_round = (size_t)G * blockDim.x * wchunks;
  const u32x4* wp = W + (size_t)r * per_round + ((size_t)blockIdx.x * wchunks * blockDim.x) + threadIdx.x;
  u32x4 wv[8];
  if (MODE >= 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < wchunks) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * blockDim.x);
  }
  if (MODE >= 1 && r > 0) {
    const unsigned* prev = vec + (size_t)((r - 1) & 1) * G * 4;
    for (unsigned i = threadIdx.x; i < G * 4; i += blockDim.x) {
      const unsigned v = prev[i];
      if (v != (unsigned)((r - 1) * 131 + i)) atomicExch(err, 2u);
      acc += v;
    }
  }
  if (MODE >= 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < wchunks) wacc ^= wv[i];
  }
  if (MODE >= 1 && threadIdx.x < 4) vec[(size_t)(r & 1) * G * 4 + blockIdx.x * 4 + threadIdx.x] = (unsigned)(r * 131 + blockIdx.x * 4 + threadIdx.x);
  if (acc == 0x12345u || (wacc[0] ^ wacc[1] ^ wacc[2] ^ wacc[3]) == 0x777u) *sink = acc;
}

// Mode 4: flag-array barrier (no atomics: workgroup b stores flags[b] = round + 1; one wave of every workgroup reads ALL flags --
// 16 B per lane -- until every one has reached the round), then the exchange of mode 1.
// Mode 5: no barrier at all: the exchanged words travel as 8-byte granules {value, round + 1} (the one-shot all-reduce's trick,
// csrc/comm.hip) and every workgroup polls the data itself until all tags match: publish -> visible in ONE round trip.  Two
// buffers alternate; a workgroup can run at most one round ahead of the slowest one, so a buffer is never overwritten while read.
template <int MODE>
__global__ void __launch_bounds__(256) persistent2(unsigned* flags /* [grid] */, unsigned long long* gr /* [2][grid*4] */,
                                                   unsigned* vec, const u32x4* __restrict__ W, int wchunks, int rounds,
                                                   unsigned* err, unsigned* sink) {
  const unsigned G = gridDim.x;
  unsigned acc = 0;
  u32x4 wacc = {0, 0, 0, 0};
  const size_t per_round = (size_t)G * blockDim.x * wchunks;
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (dead) break;          // a wait ran out: give up (the others run out too)
    const u32x4* wp = W + (size_t)r * per_round + ((size_t)blockIdx.x * wchunks * blockDim.x) + threadIdx.x;
    u32x4 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < wchunks) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * blockDim.x);
    if (MODE == 4) {
      unsigned* cur = vec + (size_t)(r & 1) * G * 4;
      if (threadIdx.x < 4)
        __hip_atomic_store(cur + blockIdx.x * 4 + threadIdx.x, (unsigned)(r * 131 + blockIdx.x * 4 + threadIdx.x), __ATOMIC_RELAXED, AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned)(r + 1), __ATOMIC_RELAXED, AGENT);
      if (threadIdx.x < 64) {
        long spins = 0;
        for (;;) {
          bool all = true;
          for (unsigned i = threadIdx.x; i < G; i += 64)
            all = all && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, AGENT) >= (unsigned)(r + 1);
          if (__all(all)) break;
          if (++spins > 20000L) { atomicExch(err, 1u); dead = 1; break; }
        }
      }
      __syncthreads();
      for (unsigned i = threadIdx.x; i < G * 4; i += blockDim.x) {
        const unsigned v = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, AGENT);
        if (v != (unsigned)(r * 131 + i)) atomicExch(err, 2u);
        acc += v;
      }
    } else {
      unsigned long long* cur = gr + (size_t)(r & 1) * G * 4;
      if (threadIdx.x < 4) {
        const unsigned v = (unsigned)(r * 131 + blockIdx.x * 4 + threadIdx.x);
        __hip_atomic_store(cur + blockIdx.x * 4 + threadIdx.x, ((unsigned long long)(r + 1) << 32) | v, __ATOMIC_RELAXED, AGENT);
      }
      for (unsigned i = threadIdx.x; i < G * 4; i += blockDim.x) {
        long spins = 0;
        unsigned long long g;
        while (((g = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, AGENT)) >> 32) != (unsigned long long)(r + 1)) {
          if (++spins > 20000L) { atomicExch(err, 1u); dead = 1; break; }
        }
        const unsigned v = (unsigned)g;
        if (v != (unsigned)(r * 131 + i)) atomicExch(err, 2u);
        acc += v;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < wchunks) wacc ^= wv[i];
  }
  if (acc == 0x12345u || (wacc[0] ^ wacc[1] ^ wacc[2] ^ wacc[3]) == 0x777u) *sink = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
static int run(int blocks, int wchunks, int rounds, unsigned* counter, unsigned* vec, const u32x4* W, unsigned* err, unsigned* sink,
               hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemsetAsync(counter, 0, 4, s));
    CK(hipEventRecord(e0, s));
    persistent<MODE><<<blocks, 256, 0, s>>>(counter, vec, W, wchunks, rounds, err, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  unsigned h_err = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
  const double mb = (double)blocks * 256 * wchunks * 16 / 1e6;
  printf("mode %d blocks=%4d weights/round %6.1f MB : persistent %6.2f us/round (err %u)", MODE, blocks, MODE >= 2 ? mb : 0.0,
         best * 1e3 / rounds, h_err);
  CK(hipMemset(err, 0, 4));
  if (MODE != 3) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < rounds; ++r) round_kernel<MODE><<<blocks, 256, 0, s>>>(vec, W, wchunks, r, err, sink);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
    printf(" ; graph of kernels %6.2f us/round (err %u)", best * 1e3 / rounds, h_err);
    CK(hipMemset(err, 0, 4));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  printf("\n"); fflush(stdout);
  return 0;
}

template <int MODE>
static int run2(int blocks, int wchunks, int rounds, unsigned* flags, unsigned long long* gr, unsigned* vec, const u32x4* W, unsigned* err,
                unsigned* sink, hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemsetAsync(flags, 0, 4096, s)); CK(hipMemsetAsync(gr, 0, 2 * 1024 * 4 * 8, s));
    CK(hipEventRecord(e0, s));
    persistent2<MODE><<<blocks, 256, 0, s>>>(flags, gr, vec, W, wchunks, rounds, err, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  unsigned h_err = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
  printf("mode %d blocks=%4d weights/round %6.1f MB : persistent %6.2f us/round (err %u)\n", MODE, blocks,
         (double)blocks * 256 * wchunks * 16 / 1e6, best * 1e3 / rounds, h_err);
  fflush(stdout);
  CK(hipMemset(err, 0, 4));
  return 0;
}

int main() {
  const int rounds = 64;
  unsigned *counter, *vec, *err, *sink; u32x4* W;
  const size_t wbytes = (size_t)rounds * 1024 * 256 * 8 * 16;      // 2 GiB: every round streams its own weights
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&vec, 2 * 1024 * 16)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&W, wbytes));
  CK(hipMemset(W, 1, wbytes)); CK(hipMemset(err, 0, 4)); CK(hipMemset(vec, 0, 2 * 1024 * 16));
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned* flags; unsigned long long* gr;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&gr, 2 * 1024 * 4 * 8));
  for (int blocks : {256, 512, 1024})
    for (int wchunks : {0, 2, 8}) {
      if (run2<4>(blocks, wchunks, rounds, flags, gr, vec, W, err, sink, s)) return 1;
      if (run2<5>(blocks, wchunks, rounds, flags, gr, vec, W, err, sink, s)) return 1;
    }
  for (int blocks : {256, 512, 1024}) {
    if (run<0>(blocks, 0, rounds, counter, vec, W, err, sink, s)) return 1;
    if (run<1>(blocks, 0, rounds, counter, vec, W, err, sink, s)) return 1;
    for (int wchunks : {2, 4, 8}) {
      if (run<2>(blocks, wchunks, rounds, counter, vec, W, err, sink, s)) return 1;
      if (run<3>(blocks, wchunks, rounds, counter, vec, W, err, sink, s)) return 1;
    }
  }
  return 0;
}
