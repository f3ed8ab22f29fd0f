// Timing skeleton of a WHOLE single-token decoder layer chain on the loader / consumer engine (round 4, groundwork for the next
// round's draft-layer kernel; no arithmetic of the model, the DATA FLOW of it):
//   * 256 resident workgroups x 4 waves: 2 loader waves stream the workgroup's share of every phase's weights, in phase order and
//     without ever stopping at an edge, into a 128 KiB LDS ring (global_load_lds_dwordx4, nt; profiles/micro/ldsdma.hip: 6.9 TB/s);
//   * 2 consumer waves take the slots (ds_read_b128 + a stand-in reduction), and at the end of a phase consumer 0 publishes the
//     workgroup's outputs as data-tagged granules and gathers the next phase's input vector from ALL workgroups (csrc/chain.hip's
//     edges) before either consumer touches the next phase's slots;
//   * phases per layer = Llama-3.2-1B's: o_proj 32 KiB / workgroup -> 8 KB vector, gate_up 256 KiB -> 32 KB, down_proj 128 KiB ->
//     8 KB, QKV 48 KiB -> 12 KB, plus an attention stand-in edge (8 KB) with no weights.  464 KiB per workgroup and layer = 118.8 MB
//     per layer (the real layer: 121.6 MB).
// The loader runs ahead of the consumers by up to the ring (128 KiB = 5 us of this CU's share of the stream), so an edge shorter
// than that costs nothing: the question this answers is how close a layer gets to its 17.6 us of streaming.
//   hipcc --offload-arch=gfx950 -O3 -o layer_engine layer_engine.hip && ./layer_engine
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned long long u64;
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int GRID = 256, SLOT = 16384, NS = 8, FL = 16, LW = 2, NC = 2;
constexpr int NPH = 5;
// fills (16 KiB) per workgroup, granules published per workgroup, granules gathered per workgroup -- per phase
__constant__ int PH_FILLS[NPH] = {2, 16, 8, 3, 0};            // o_proj, gate_up, down_proj, QKV, attention stand-in
__constant__ int PH_PUB[NPH] = {4, 16, 4, 6, 4};              // x 256 workgroups = the vector below
__constant__ int PH_VEC[NPH] = {1024, 4096, 1024, 1536, 1024};
constexpr int FILLS_PER_LAYER = 2 + 16 + 8 + 3;

__device__ __forceinline__ bool spin_ge(volatile unsigned* p, unsigned target, unsigned* err) {
  for (long i = 0; *p < target; ++i) {
    if (i > 4000000L) { *err = 1; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

__global__ void __launch_bounds__(256) layer_engine(const u32x4* __restrict__ W, int layers, u64* gran /* [NPH][4096] */, unsigned* err,
                                                    unsigned* sink, int edges_on, unsigned long long* stats /* [8] ticks */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  volatile unsigned* ready = reinterpret_cast<volatile unsigned*>(smem + NS * SLOT);     // [NS] fill + 1 landed
  volatile unsigned* freed = ready + NS;                                                  // [NS] fill + 1 consumed
  volatile unsigned* phase_open = freed + NS;                                             // fills < *phase_open may be consumed
  volatile unsigned* consumed = phase_open + 1;                                           // [NC] fills consumed by consumer c
  volatile unsigned* arrive = consumed + NC;                                              // consumers at the edge (monotonic)
  volatile unsigned* gathered = arrive + 1;                                               // consumers done gathering (monotonic)
  volatile unsigned* gathering = gathered + 1;                                            // consumers currently inside a gather
  const bool thin_on = edges_on >= 3;
  if (edges_on >= 3) edges_on -= 2;                                                       // 3 = edges + thinning, 4 = no rendezvous + thinning
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 2 * NS + 1 + NC + 3) ready[threadIdx.x] = 0;
  __syncthreads();
  const int nfills = layers * FILLS_PER_LAYER;
  const u32x4* src = W + (size_t)blockIdx.x * nfills * (SLOT / 16);
  if (wave < LW) {
    // ---------------- loaders: fill f belongs to loader f % LW; two fills of its own outstanding ----------------
    int f = wave, marked = wave;
    unsigned long long t_stall = 0;
    for (; f < nfills; f += LW) {
      const int slot = f % NS;
      const unsigned long long ts = wall_clock64();
      if (f >= NS && !spin_ge(freed + slot, (unsigned)(f - NS + 1), err)) return;
      t_stall += wall_clock64() - ts;
#pragma unroll
      for (int i = 0; i < FL; ++i)
        __builtin_amdgcn_global_load_lds(GLBP(src + (size_t)f * (SLOT / 16) + i * 64 + lane), LDSP(ring + slot * SLOT + i * 1024), 16, 0, 2);
      // thinned while a consumer of this CU gathers (the guide's gather-pass row): one outstanding fill instead of two
      const bool thin = thin_on && *gathering;
      if (thin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      const int upto = thin ? f : f - LW;
      if (lane == 0) for (; marked <= upto; marked += LW) ready[marked % NS] = (unsigned)(marked + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) for (; marked < nfills; marked += LW) ready[marked % NS] = (unsigned)(marked + 1);
    if (lane == 0 && wave == 0) atomicAdd(stats + 0, t_stall);          // loader 0: waiting for a free slot
    return;
  }
  // ---------------- consumers ----------------
  const int c = wave - LW;
  u32x4 acc = {0, 0, 0, 0};
  unsigned fbase = 0;            // first fill of the current phase
  unsigned long long t_ready = 0, t_arrive = 0, t_gather = 0, t_sync = 0;
  for (int layer = 0; layer < layers; ++layer) {
    for (int ph = 0; ph < NPH; ++ph) {
      const unsigned nf = PH_FILLS[ph];
      // this phase's slots: fill f belongs to consumer f % NC
      for (unsigned f = fbase + ((c + NC - fbase % NC) % NC); f < fbase + nf; f += NC) {
        const int slot = f % NS;
        const unsigned long long tr = wall_clock64();
        if (!spin_ge(ready + slot, f + 1, err)) return;
        t_ready += wall_clock64() - tr;
        const u32x4* s = reinterpret_cast<const u32x4*>(ring + slot * SLOT);
#pragma unroll
        for (int i = 0; i < FL; ++i) acc ^= s[i * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) freed[slot] = f + 1;
      }
      fbase += nf;
      if (lane == 0) consumed[c] = fbase;
      if (!edges_on) continue;
      // the edge: both consumers arrive (LDS counter); consumer 0 publishes the workgroup's outputs; each consumer gathers HALF of
      // the vector (16 granules per lane and pass); both meet again before the next phase's slots (the gathered halves are the
      // next phase's B operand in LDS)
      const unsigned edge = (unsigned)(layer * NPH + ph), tag = edge + 1;
      u64* buf = gran + (size_t)ph * 4096;
      if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(arrive), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned long long ta = wall_clock64();
      if (c == 0) {
        if (!spin_ge(arrive, NC * (edge + 1), err)) return;
        const int npub = PH_PUB[ph];
        if (lane < npub) __hip_atomic_store(buf + blockIdx.x * npub + lane, ((u64)tag << 32) | (acc[0] & 0xffffu), __ATOMIC_RELAXED, AGENT);
      }
      const unsigned long long tg = wall_clock64();
      t_arrive += tg - ta;
      if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(gathering), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int nv = PH_VEC[ph], half = nv / NC;
      for (int base = c * half; base < (c + 1) * half; base += 1024) {
        long spins = 0;
        for (;;) {
          bool ok = true;
          u64 g = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int idx = base + i * 64 + lane;
            if (idx < (c + 1) * half) {
              const u64 v = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, AGENT);
              ok = ok && (unsigned)(v >> 32) == tag;
              g ^= v;
            }
          }
          if (__all(ok) || edges_on == 2) { acc[1] ^= (unsigned)g; break; }      // (2: one pass, no rendezvous: the mechanism alone)
          if (++spins > 2000000L) { *err = 2; return; }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_sub(const_cast<unsigned*>(gathering), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned long long ts2 = wall_clock64();
      t_gather += ts2 - tg;
      if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(gathered), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!spin_ge(gathered, NC * (edge + 1), err)) return;
      t_sync += wall_clock64() - ts2;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) *sink = 1;
  if (lane == 0 && c == 0) { atomicAdd(stats + 1, t_ready); atomicAdd(stats + 2, t_arrive); atomicAdd(stats + 3, t_gather); atomicAdd(stats + 4, t_sync); }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int layers = 16;
  const size_t wbytes = (size_t)GRID * layers * FILLS_PER_LAYER * SLOT;          // 1.9 GB
  u32x4* W; u64* gran; unsigned *err, *sink; unsigned long long* stats;
  CK(hipMalloc(&stats, 64));
  CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 1, wbytes));
  CK(hipMalloc(&gran, (size_t)NPH * 4096 * 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = NS * SLOT + 64 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_engine), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int edges = 0; edges < 5; ++edges) {
    float best = 1e9; unsigned h_err = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(gran, 0, (size_t)NPH * 4096 * 8, s)); CK(hipMemsetAsync(err, 0, 4, s)); CK(hipMemsetAsync(stats, 0, 64, s));
      CK(hipEventRecord(e0, s));
      layer_engine<<<GRID, 256, lds, s>>>(W, layers, gran, err, sink, edges, stats);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      unsigned e = 0; CK(hipStreamSynchronize(s)); CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); h_err |= e;
    }
    printf("%s: %d layers %8.1f us = %6.2f us per layer, %5.2f TB/s (err %u)\n", edges == 4 ? "no rendezvous, loaders THINNED during gathers" : edges == 3 ? "5 all-to-all edges, loaders THINNED during gathers" :
           edges == 2 ? "stream + 5 edges per layer WITHOUT the rendezvous (publish, one gather pass)" : edges ? "stream + 5 all-to-all edges per layer" : "stream only (no edges)",
           layers, best * 1e3, best * 1e3 / layers, (double)wbytes / (best * 1e-3) / 1e12, h_err);
    unsigned long long h[8]; CK(hipMemcpy(h, stats, 64, hipMemcpyDeviceToHost));
    const double k = 1.0 / 100.0 / GRID / layers;       // 100 MHz ticks -> us, per workgroup and layer (last repetition)
    printf("    per layer and workgroup (us): loader 0 waiting for a free slot %.2f | consumer 0: waiting for landed fills %.2f, for its partner %.2f, "
           "publish + gather %.2f, meeting after the gather %.2f\n", h[0] * k, h[1] * k, h[2] * k, h[3] * k, h[4] * k);
    fflush(stdout);
  }
  return 0;
}
