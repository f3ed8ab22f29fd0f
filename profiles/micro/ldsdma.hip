// Loader / consumer weight stream through an LDS ring -- the skeleton of the engine the guide prices (MI355X_MICROARCH.md rows
// ldsdma-fill, gather-pass, engine-vs-launches) and the answer to what capped csrc/chain.hip this round: there the waves that stream
// the weights through their registers also poll the hand-offs, so every poll queues behind the stream.  Here ONE loader wave per
// workgroup owns the stream (`global_load_lds_dwordx4`: HBM -> LDS without registers, 16 x 1 KiB = one 16 KiB slot of an 8-slot ring,
// DEPTH fills outstanding), three consumer waves take slots out of the ring, and -- experiment 2 -- one consumer wave now and then
// gathers an 8 KB granule vector with agent-scope loads while the loader either keeps its DEPTH or is thinned to one outstanding
// fill for the duration of the gather.
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma ldsdma.hip && ./ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned long long u64;
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int RING = 131072;       // LDS bytes of the ring: NS slots of FL KiB (FL = 1 KiB wave loads per fill)

// every wait is bounded: a logic error must end the launch, not hang the box
__device__ __forceinline__ void spin_ge(volatile unsigned* p, unsigned target, unsigned* sink) {
  for (long i = 0; *p < target; ++i) {
    if (i > 4000000L) { *sink = 0xdeadu; break; }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int N> __device__ __forceinline__ void wait_vm() {       // at most N of this wave's loads still outstanding
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  if (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  if (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
}

// mode 0: stream only.  mode 1: consumer wave 3 gathers `gather_every` fills apart, loader keeps DEPTH.  mode 2: loader thinned to one
// outstanding fill while the gather runs.
template <int DEPTH, int FL, int LW>
__global__ void __launch_bounds__(256) engine(const u32x4* __restrict__ W, int nfills, const u64* __restrict__ granules, int mode,
                                              int gather_every, unsigned* sink, unsigned long long* gather_ticks, unsigned* gather_count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOT = FL * 1024, NS = RING / SLOT, NC = 4 - LW;
  static_assert((DEPTH - 1) * FL <= 48, "vmcnt is 6 bits");
  char* ring = smem;                                        // NS slots
  volatile unsigned* ready = reinterpret_cast<volatile unsigned*>(smem + NS * SLOT);          // [NS] fill index + 1 when landed
  volatile unsigned* freed = ready + NS;                                                       // [NS] fill index + 1 when consumed
  volatile unsigned* gathering = freed + NS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 2 * NS + 1) ready[threadIdx.x] = 0;
  __syncthreads();
  const u32x4* src = W + (size_t)blockIdx.x * nfills * (SLOT / 16);
  if (wave < LW) {
    // ---------------- loaders: fill f belongs to loader f % LW ----------------
    int f = wave;
    for (; f < nfills; f += LW) {
      const int slot = f % NS;
      if (f >= NS) spin_ge(freed + slot, (unsigned)(f - NS + 1), sink);
      if (mode == 2 && *gathering) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // thinned: nothing of ours in the queue
#pragma unroll
      for (int i = 0; i < FL; ++i)
        __builtin_amdgcn_global_load_lds(GLBP(src + (size_t)f * (SLOT / 16) + i * 64 + lane), LDSP(ring + slot * SLOT + i * 1024), 16, 0, 2 /* nt */);
      wait_vm<(DEPTH - 1) * FL>();                           // this loader's fill issued DEPTH - 1 of ITS fills ago has landed
      const int done = f - (DEPTH - 1) * LW;
      if (done >= 0 && lane == 0) ready[done % NS] = (unsigned)(done + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) for (int d = f - (DEPTH - 1) * LW; d < nfills; d += LW) if (d >= 0) ready[d % NS] = (unsigned)(d + 1);
  } else {
    // ---------------- consumers: fill f belongs to consumer LW + f % NC ----------------
    u32x4 acc = {0, 0, 0, 0};
    unsigned long long ticks = 0;
    unsigned ngather = 0;
    for (int f = wave - LW; f < nfills; f += NC) {
      const int slot = f % NS;
      spin_ge(ready + slot, (unsigned)(f + 1), sink);
      const u32x4* s = reinterpret_cast<const u32x4*>(ring + slot * SLOT);
#pragma unroll
      for (int i = 0; i < FL; ++i) acc ^= s[i * 64 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) freed[slot] = (unsigned)(f + 1);
      if (mode != 0 && wave == 3 && gather_every > 0 && ((f / NC) % gather_every) == gather_every - 1) {
        // one gather pass: 1024 granules = 8 KB, 16 agent-scope 8-byte loads per lane
        if (lane == 0) *gathering = 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = wall_clock64();
        asm volatile("" ::: "memory");
        u64 g = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) g ^= __hip_atomic_load(granules + i * 64 + lane, __ATOMIC_RELAXED, AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the pass ends when its last granule has arrived)
        acc[0] ^= (unsigned)g;
        const unsigned long long t1 = wall_clock64();
        if (lane == 0) *gathering = 0;
        ticks += t1 - t0; ++ngather;
      }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) *sink = 1;
    if (wave == 3 && lane == 0 && ngather) { atomicAdd(gather_ticks, ticks); atomicAdd(gather_count, ngather); }
  }
}

// the same bytes through registers (what csrc/gemm*.hip and chain.hip do): 4 waves, 8 x 16-byte loads in flight per lane
__global__ void __launch_bounds__(256) regstream(const u32x4* __restrict__ W, int nfills, unsigned* sink) {
  constexpr int SLOT = 16384;
  const u32x4* src = W + (size_t)blockIdx.x * nfills * (SLOT / 16) + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const int n = nfills * (SLOT / 16) / 256;     // 16-byte chunks per thread
  for (int i = 0; i < n; i += 8) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + (size_t)(i + k) * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int DEPTH, int FL = 16, int LW = 1>
static int run(const u32x4* W, int blocks, int nfills16, const u64* gr, int mode, int every, unsigned* sink, unsigned long long* gt, unsigned* gc,
               hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = RING + 64 * 4;
  const int nfills = nfills16 * 16 / FL, SLOT = FL * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(engine<DEPTH, FL, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best = 1e9;
  unsigned long long ticks = 0; unsigned cnt = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(gt, 0, 8, s)); CK(hipMemsetAsync(gc, 0, 4, s));
    CK(hipEventRecord(e0, s));
    engine<DEPTH, FL, LW><<<blocks, 256, lds, s>>>(W, nfills, gr, mode, every, sink, gt, gc);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) { best = ms; CK(hipMemcpy(&ticks, gt, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cnt, gc, 4, hipMemcpyDeviceToHost)); }
  }
  const double bytes = (double)blocks * nfills * SLOT;
  printf("ring %2d KiB slots, %d loader(s), DEPTH %d, mode %d: %7.1f us, %6.2f TB/s (%5.1f GB/s per workgroup)", FL, LW, DEPTH, mode, best * 1e3, bytes / (best * 1e-3) / 1e12,
         bytes / blocks / (best * 1e-3) / 1e9);
  if (cnt) printf("; gather pass (8 KB, one wave): %5.2f us average over %u", (double)ticks / cnt / 100.0, cnt);
  printf("\n"); fflush(stdout);
  return 0;
}

int main() {
  const int blocks = 256, nfills = 256;                     // (16 KiB units) 4 MiB per workgroup, 1 GiB per launch
  constexpr int SLOT = 16384;
  u32x4* W; u64* gr; unsigned* sink; unsigned long long* gt; unsigned* gc;
  CK(hipMalloc(&W, (size_t)blocks * nfills * SLOT)); CK(hipMemset(W, 1, (size_t)blocks * nfills * SLOT));
  CK(hipMalloc(&gr, 8192)); CK(hipMemset(gr, 0, 8192)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&gt, 8)); CK(hipMalloc(&gc, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s)); regstream<<<blocks, 256, 0, s>>>(W, nfills, sink); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("register stream (4 waves x 8 loads in flight per lane): %7.1f us, %6.2f TB/s\n", best * 1e3, (double)blocks * nfills * SLOT / (best * 1e-3) / 1e12);
  fflush(stdout);
  if (run<1>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s)) return 1;
  if (run<2>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s)) return 1;
  if (run<4>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s)) return 1;
  if (run<4>(W, blocks, nfills, gr, 1, 4, sink, gt, gc, s)) return 1;      // gathers beside the full-depth loader
  if (run<4>(W, blocks, nfills, gr, 2, 4, sink, gt, gc, s)) return 1;      // loader thinned while a gather runs
  if (run<2>(W, blocks, nfills, gr, 1, 4, sink, gt, gc, s)) return 1;
  if (run<2>(W, blocks, nfills, gr, 2, 4, sink, gt, gc, s)) return 1;
  // handshake amortised over 32 KiB; two loader waves (each every other fill) + two consumers
  if ((run<2, 32, 1>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s))) return 1;
  if ((run<2, 16, 2>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s))) return 1;
  if ((run<4, 16, 2>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s))) return 1;
  if ((run<2, 32, 2>(W, blocks, nfills, gr, 0, 0, sink, gt, gc, s))) return 1;
  if ((run<4, 16, 2>(W, blocks, nfills, gr, 1, 4, sink, gt, gc, s))) return 1;
  return 0;
}
