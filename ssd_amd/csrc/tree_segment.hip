// One decoder layer between two attention launches for 2..30 token rows as ONE resident launch:
//
//   o_proj -> (+residual, RMSNorm) -> gate_up + SiLU*mul -> down_proj -> (+residual, RMSNorm) -> the NEXT layer's QKV + RoPE + KV store
//
// the M-row form of chain.hip: the 24-branch tree-decode step of asynchronous speculation (reference
// ssd/engine/draft_runner.py:713-812 _decode_tree; MQ_LEN = (K+1) * F rows through LlamaDecoderLayer.forward,
// ssd/models/llama3.py:128-199) and the K+1-row glue decode (draft_runner.py:560-640).  Seven launches per layer become two
// (attention + this): the tree step was 7 x ~4.3 us of boundary + ramp on 17.6 us of weight streaming per 1B layer.
//
// MI355X design (what differs from the single-token chain).  At M = 24 a phase's output is 96 KB (o_proj / down_proj rows) or
// 384 KB (the activation), so an all-to-all edge cannot be a granule all-gather (2x the bytes, polled by every workgroup):
//   * a producer writes its finished bf16 tile with 16-byte WRITE-THROUGH stores (`buffer_store_dwordx4 ... sc1`), drains them
//     (s_waitcnt vmcnt(0)), meets its workgroup, and one thread stores the workgroup's flag word {generation, layer, phase};
//   * a consumer's wave 0 polls the 256 flag words (1 KB: one 16-byte agent-scope load per lane) while the other waves sit at an
//     LDS-only barrier with the NEXT phase's first weight tiles already in flight, then every wave reads the payload with `sc1`
//     loads (L2-served, never from this CU's L1: no acquire fence, no cache invalidate anywhere);
//   * every workgroup repeats the residual add + RMSNorm of all M rows itself (a wave owns whole rows: the row's sum of squares
//     is a wave reduction -- NOT ssd_rmsnorm's order, see ts_add_norm: tolerance-equal to the separate launches) and keeps x^ -- the B operand of gate_up / QKV -- in LDS (96 KB at M = 24);
//     the activation, too big for LDS, is read per k-tile from L2 in the fragment-major layout it was published in;
//   * o_proj / down_proj: row HALVES of a 16-row group per workgroup over the full K (256 units for the 1B: no split-K slabs);
//   * every global access is `buffer_load / buffer_store` with a scalar base, a scalar tile offset and a 32-bit lane offset: no
//     64-bit VALU address arithmetic, no integer division in front of the first load (2.5-3.3 us of front end in
//     the launches this replaces); the barriers inside the phases wait for LDS only (`s_waitcnt lgkmcnt(0); s_barrier`), so a
//     prefetched weight stream is never drained by a combine.
// Tags, generations, bounded waits and the error word are chain.hip's (ssd_chain_tick bumps the generation once per forward).
// Rounding points are those of the separate launches (bf16 after every projection, fp32 residual add, bf16 residual, fp32 norm
// with a single rounding); the fp32 summation ORDER inside o_proj / down_proj differs from the split-K slab path it replaces
// (tolerance-tested against it and against the oracle: tests/test_hip_tree_segment.py).
#include "common.h"
#include <type_traits>

typedef unsigned long long u64_t;
typedef __amdgpu_buffer_rsrc_t ts_rsrc_t;
#define TS_AGENT __HIP_MEMORY_SCOPE_AGENT
constexpr int TS_WAVES = 8, TS_THREADS = TS_WAVES * 64, TS_GRID = 256;
constexpr int TS_CRED_BYTES = TS_WAVES * 4 * 64 * 16;      // split-K combine area: 8 waves x (NT * MT <= 4) tiles
constexpr int TS_PLAIN = 0, TS_NT = 2, TS_SC1 = 16;        // gfx940+ cache-policy bits of the raw buffer intrinsics

struct TsParams {
  const void* a_frag;       // attention output of this layer, fragment-major [MT * 16][qn]
  const bf16_t* res_in;     // [M][h] residual entering the layer's attention add
  bf16_t* res_mid;          // [M][h] residual after the attention add (published write-through; the LAST layer's residual output)
  bf16_t* res_out;          // [M][h] residual after the MLP add (layers before the last)
  bf16_t* o_rows;           // [M][h] hand-off 1: o_proj output
  void* act_f;              // hand-off 2: SiLU(gate) * up, fragment-major [MT * 16][I]
  bf16_t* d_rows;           // [M][h] hand-off 3: down_proj output (the last layer's h_out)
  const void* Wo;
  const void* Wgu;
  const void* Wd;
  const void* Wqkv;         // NEXT layer's fused QKV (rotation-paired rows); null on the last layer
  const bf16_t* ln_post;
  const bf16_t* ln_next;
  const int64_t* positions;
  const float* cos_sin;
  const int32_t* slots;
  bf16_t* q_out;
  bf16_t* k_cache;
  bf16_t* v_cache;
  bf16_t* qkv_rows;         // alternative to the RoPE epilogue (models with a per-head q / k norm): the next layer's raw QKV rows [M][qkv_n]
  unsigned* flags;          // [3][TS_GRID]
  const unsigned* gen;
  unsigned* err;
  float eps;
  int M, h, qn, I, qkv_n, nh, nkv, hd, bs, layer;
  long spin_budget;
};

__device__ __forceinline__ ts_rsrc_t ts_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
template <int CP>
__device__ __forceinline__ u32x4_t ts_load(ts_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, CP);
}
template <int CP>
__device__ __forceinline__ void ts_store(u32x4_t v, ts_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, CP);
}
// Workgroup barrier that orders LDS traffic only: outstanding global loads (the prefetched weight tiles) stay in flight.
__device__ __forceinline__ void ts_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// End of a producing phase: every wave's write-through stores have left, the workgroup has met, one flag word goes out.
__device__ __forceinline__ void ts_publish(unsigned* flag, unsigned tag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, TS_AGENT);
}
// Wave 0 polls all TS_GRID flag words of a phase (4 per lane) until they carry `tag`; the other waves wait at the LDS barrier.
__device__ __forceinline__ void ts_wait(const unsigned* flags, unsigned tag, const TsParams& p, int wave, int lane) {
  if (wave == 0) {
    const u64_t* f = reinterpret_cast<const u64_t*>(flags) + lane * 2;
    const u64_t want = ((u64_t)tag << 32) | tag;
    long spins = 0;
    for (;;) {
      const u64_t v0 = __hip_atomic_load(f, __ATOMIC_RELAXED, TS_AGENT), v1 = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, TS_AGENT);
      if (__ballot(v0 == want && v1 == want) == ~0ull) break;
      if (++spins > p.spin_budget) { if (lane == 0) atomicExch(p.err, 1u); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  ts_barrier_lds();
}

// One workgroup's share of a skinny GEMM: NT adjacent 16-row groups of W (HALF: 8 rows of one group) x MT 16-token tiles over the
// full K; the k-tiles are dealt to the 8 waves in groups of U, two groups in flight per wave.  B operand: XS = 0 the LDS image of
// x^ (chunk (k8, m) at k8 * pitch + m), 1 fragment-major global memory written by an earlier launch, 2 fragment-major global
// memory published in THIS launch (sc1 loads).
template <int NT, int MT, int U, bool HALF, int XS>
struct TsGemm {
  ts_rsrc_t wr, xr;
  unsigned lane16, wbase, wstride, xstride;
  int kt0, kstep, nmain;
  bool wact, xact[MT];
  u32x4_t a[2][U][NT];
  u32x4_t x[2][U][MT];

  __device__ __forceinline__ void init(ts_rsrc_t w_rsrc, int g0, int KT, int half, ts_rsrc_t x_rsrc, int M, int wave, int lane) {
    wr = w_rsrc; xr = x_rsrc;
    lane16 = (unsigned)lane << 4;
    // tile offsets are wave-uniform and travel in the scalar offset operand (readfirstlane: the compiler must not build a
    // per-lane waterfall around the buffer instruction); the lane offset + the k-tile within a group are the vector / immediate part
    wbase = __builtin_amdgcn_readfirstlane((unsigned)g0 * (unsigned)KT << 10);
    wstride = xstride = __builtin_amdgcn_readfirstlane((unsigned)KT << 10);
    kstep = TS_WAVES * U;
    kt0 = __builtin_amdgcn_readfirstlane(wave * U);
    const int ngroups = KT / U;
    nmain = ngroups > wave ? (ngroups - wave + TS_WAVES - 1) / TS_WAVES : 0;
    wact = !HALF || (((lane & 15) >> 3) == half);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xact[mt] = mt * 16 + (lane & 15) < M;      // padding token rows are never loaded
  }
  template <int BUF>
  __device__ __forceinline__ void loadW(int kt) {
    const unsigned kb = __builtin_amdgcn_readfirstlane((unsigned)kt << 10);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (wact) v = ts_load<TS_NT>(wr, lane16 + ((unsigned)u << 10), __builtin_amdgcn_readfirstlane(wbase + nt * wstride + kb));
        a[BUF][u][nt] = v;
      }
  }
  template <int BUF>
  __device__ __forceinline__ void loadX(int kt) {
    if (XS == 0) return;
    const unsigned kb = __builtin_amdgcn_readfirstlane((unsigned)kt << 10);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (xact[mt]) v = ts_load<XS == 2 ? TS_SC1 : TS_PLAIN>(xr, lane16 + ((unsigned)u << 10), __builtin_amdgcn_readfirstlane(mt * xstride + kb));
        x[BUF][u][mt] = v;
      }
  }
  __device__ __forceinline__ void prefetchW() {
    const int nm = __builtin_amdgcn_readfirstlane(nmain);
    if (nm > 0) loadW<0>(kt0);
    if (nm > 1) loadW<1>(kt0 + kstep);
  }
  __device__ __forceinline__ void prefetchX() {
    const int nm = __builtin_amdgcn_readfirstlane(nmain);
    if (nm > 0) loadX<0>(kt0);
    if (nm > 1) loadX<1>(kt0 + kstep);
  }
  template <int BUF>
  __device__ __forceinline__ void stage(const u32x4_t* xlds, int pitch, f32x4_t (&acc)[NT][MT], int& kt, int it, int nm, int lane) {
    u32x4_t xb[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (XS != 0) {
          xb[u][mt] = x[BUF][u][mt];
        } else {
          u32x4_t o = {0u, 0u, 0u, 0u};
          if (xact[mt]) o = xlds[((kt + u) * 4 + (lane >> 4)) * pitch + mt * 16 + (lane & 15)];
          xb[u][mt] = o;
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(a[BUF][u][nt], xb[u][mt], acc[nt][mt]);
    if (it + 2 < nm) { loadW<BUF>(kt + 2 * kstep); loadX<BUF>(kt + 2 * kstep); }
    kt += kstep;
  }
  __device__ __forceinline__ void run(const u32x4_t* xlds, int pitch, f32x4_t (&acc)[NT][MT], int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // (wave-uniform by construction; said explicitly so that the k loop and its tile offsets stay on the scalar unit)
    const int nm = __builtin_amdgcn_readfirstlane(nmain);
    int kt = __builtin_amdgcn_readfirstlane(kt0);
    for (int it = 0; it < nm; it += 2) {
      stage<0>(xlds, pitch, acc, kt, it, nm, lane);
      if (it + 1 < nm) stage<1>(xlds, pitch, acc, kt, it + 1, nm, lane);
    }
  }
};

// Wave `mt` (< MT) finishes token tile mt of a row-half unit: fixed-order sum of the 8 waves' partial tiles, bf16, and one
// 16-byte write-through store per token row (8 features: the lane pair (q4, q4 + 1) of the accumulator layout).
template <int MT>
__device__ __forceinline__ void ts_store_rows(const f32x4_t* cred, int mt, ts_rsrc_t rows, int grp, int half, int M, int h, int lane) {
  f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) s += cred[(w * MT + mt) * 64 + lane];
  const unsigned lo0 = pack_bf2_hw(s[0], s[1]), lo1 = pack_bf2_hw(s[2], s[3]);
  const unsigned hi0 = __shfl_down(lo0, 16, 64), hi1 = __shfl_down(lo1, 16, 64);
  const int q4 = lane >> 4, m = mt * 16 + (lane & 15);
  if ((q4 & 1) == 0 && (q4 >> 1) == half && m < M)
    ts_store<TS_SC1>(u32x4_t{lo0, lo1, hi0, hi1}, rows, (unsigned)(m * h + grp * 16 + half * 8) * 2u, 0u);
}

// (projection output + residual) of all M rows -> the new bf16 residual (chunk c of every row by workgroup c), the row's sum of
// squares and x^ = bf16((x * rs) * w) into the LDS image.  Summation order: each lane first adds the chunk sums of its
// CPL lane-strided chunks, then one xor tree over the wave -- ssd_rmsnorm<256> at h = 2048 runs the xor tree per 64-chunk block and
// adds the 4 block totals in order, so rs can differ from the separate launches' in the last ulp: like the projections' fp32 sums,
// the segment's norm is tolerance-equal to them, not bit-identical (tests/test_hip_tree_segment.py holds it to the oracle).  Wave w
// owns rows w, w + 8, ...; two rows' loads are in flight at a time.  FIRST: a = o_proj rows (published), b = the layer's input
// residual (an earlier launch's), the result is published write-through (phase 4 of every workgroup re-reads it); else
// a = down_proj rows, b = that residual (both published), the result is the next launch's.
template <int CPL, bool FIRST>
__device__ __forceinline__ void ts_add_norm(const TsParams& p, u32x4_t* xlds, int pitch, ts_rsrc_t ra, ts_rsrc_t rb, ts_rsrc_t rdst,
                                            const bf16_t* ln_w, int wave, int lane) {
  const int M = p.M;
  const unsigned rowb = (unsigned)p.h * 2u, lane16 = (unsigned)lane << 4;
  const int b = blockIdx.x;
  u32x4_t wv[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) wv[j] = *reinterpret_cast<const u32x4_t*>(ln_w + (j * 64 + lane) * 8);
  for (int r0 = wave; r0 < M; r0 += 2 * TS_WAVES) {
    u32x4_t av[2][CPL], bv[2][CPL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + i * TS_WAVES;
      if (row < M) {
        const unsigned ro = __builtin_amdgcn_readfirstlane((unsigned)row * rowb);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          av[i][j] = ts_load<TS_SC1>(ra, lane16 + j * 1024u, ro);
          bv[i][j] = ts_load<FIRST ? TS_PLAIN : TS_SC1>(rb, lane16 + j * 1024u, ro);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + i * TS_WAVES;
      if (row < M) {
        float x[CPL][8], t = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          float cs = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[j][2 * e] = bf2f(av[i][j][e] & 0xffffu) + bf2f(bv[i][j][e] & 0xffffu);
            x[j][2 * e + 1] = bf2f(av[i][j][e] >> 16) + bf2f(bv[i][j][e] >> 16);
            cs += x[j][2 * e] * x[j][2 * e]; cs += x[j][2 * e + 1] * x[j][2 * e + 1];
          }
          t += cs;
        }
        t = wave_sum(t);
        const float rs = 1.0f / sqrtf(t / (float)p.h + p.eps);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int c = j * 64 + lane;
          if (c == b) {
            const u32x4_t ro = {pack_bf2_hw(x[j][0], x[j][1]), pack_bf2_hw(x[j][2], x[j][3]), pack_bf2_hw(x[j][4], x[j][5]), pack_bf2_hw(x[j][6], x[j][7])};
            ts_store<FIRST ? TS_SC1 : TS_PLAIN>(ro, rdst, (unsigned)c * 16u, __builtin_amdgcn_readfirstlane((unsigned)row * rowb));
          }
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2_hw((x[j][2 * e] * rs) * bf2f(wv[j][e] & 0xffffu), (x[j][2 * e + 1] * rs) * bf2f(wv[j][e] >> 16));
          xlds[c * pitch + row] = o;
        }
      }
    }
  }
  ts_barrier_lds();
}

template <int MT, int CPL>
__global__ void __launch_bounds__(TS_THREADS) tree_segment_kernel(const TsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int mcol = lane & 15, q4 = lane >> 4;
  const bool last = p.Wqkv == nullptr;
  const int M = p.M, pitch = p.M + 1;                         // odd chunk pitch: the image's column writes spread over the banks
  f32x4_t* cred = reinterpret_cast<f32x4_t*>(smem);
  u32x4_t* xlds = reinterpret_cast<u32x4_t*>(smem + TS_CRED_BYTES);
  const int KTS = last ? 15 : 14;                             // trace slots (profiling builds only)
  KTRACE(KTS, 0);
  const unsigned tag0 = ((*p.gen & 0xffffffu) << 8) | ((unsigned)p.layer << 2);
  const int KTq = p.qn >> 5, KTh = p.h >> 5, KTi = p.I >> 5;
  const size_t rows_bytes = (size_t)M * p.h * 2;
  const ts_rsrc_t r_o = ts_rsrc(p.o_rows, rows_bytes), r_d = ts_rsrc(p.d_rows, rows_bytes), r_mid = ts_rsrc(p.res_mid, rows_bytes);
  const ts_rsrc_t r_in = ts_rsrc(p.res_in, rows_bytes);
  const ts_rsrc_t r_act = ts_rsrc(p.act_f, (size_t)MT * 16 * p.I * 2), r_a = ts_rsrc(p.a_frag, (size_t)MT * 16 * p.qn * 2);

  // ---------------- phase 1: o_proj, half row groups ----------------
  {
    const int units = (p.h >> 4) * 2;
    const ts_rsrc_t rw = ts_rsrc(p.Wo, (size_t)p.h * p.qn * 2);
    TsGemm<1, MT, 4, true, 1> g;
    if (b < units) { g.init(rw, b >> 1, KTq, b & 1, r_a, M, wave, lane); g.prefetchW(); g.prefetchX(); }
    for (int u = b; u < units; u += TS_GRID) {
      f32x4_t acc[1][MT];
      g.run(nullptr, 0, acc, lane);
      if (u + TS_GRID < units) { g.init(rw, (u + TS_GRID) >> 1, KTq, (u + TS_GRID) & 1, r_a, M, wave, lane); g.prefetchW(); g.prefetchX(); }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) cred[(wave * MT + mt) * 64 + lane] = acc[0][mt];
      ts_barrier_lds();
      if (wave < MT) ts_store_rows<MT>(cred, wave, r_o, u >> 1, u & 1, M, p.h, lane);
      ts_barrier_lds();
    }
    KTRACE(KTS, 1);
    ts_publish(p.flags + b, tag0 | 1u);
  }
  KTRACE(KTS, 2);

  // ---------------- phase 2: add + norm, gate_up + SiLU * mul (the first unit's weights fly during the edge) ----------------
  {
    const int pairs = p.I >> 4;          // (gate, up) row-group pairs
    const ts_rsrc_t rw = ts_rsrc(p.Wgu, (size_t)2 * p.I * p.h * 2);
    TsGemm<2, MT, 4, false, 0> g;
    if (b < pairs) { g.init(rw, 2 * b, KTh, 0, rw, M, wave, lane); g.prefetchW(); }
    ts_wait(p.flags, tag0 | 1u, p, wave, lane);
    KTRACE(KTS, 3);
    ts_add_norm<CPL, true>(p, xlds, pitch, r_o, r_in, r_mid, p.ln_post, wave, lane);
    KTRACE(KTS, 4);
    const unsigned KT2 = (unsigned)p.I >> 5;
    for (int pr = b; pr < pairs; pr += TS_GRID) {
      f32x4_t acc[2][MT];
      g.run(xlds, pitch, acc, lane);
      if (pr + TS_GRID < pairs) { g.init(rw, 2 * (pr + TS_GRID), KTh, 0, rw, M, wave, lane); g.prefetchW(); }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cred[(wave * 2 * MT + nt * MT + mt) * 64 + lane] = acc[nt][mt];
      ts_barrier_lds();
      if (wave < MT) {
        const int mt = wave;
        f32x4_t gs = f32x4_t{0.f, 0.f, 0.f, 0.f}, us = gs;
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) { gs += cred[(w * 2 * MT + mt) * 64 + lane]; us += cred[(w * 2 * MT + MT + mt) * 64 + lane]; }
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gb = round_bf_hw(gs[r]), ub = round_bf_hw(us[r]);
          o[r] = (gb / (1.0f + __expf(-gb))) * ub;
        }
        const unsigned lo0 = pack_bf2_hw(o[0], o[1]), lo1 = pack_bf2_hw(o[2], o[3]);
        const unsigned hi0 = __shfl_down(lo0, 16, 64), hi1 = __shfl_down(lo1, 16, 64);
        // activation feature n = pr * 16 + q4 * 4 + r of token m: chunk (m, n / 8) of the fragment-major [MT * 16][I] image
        if ((q4 & 1) == 0 && mt * 16 + mcol < M) {
          const unsigned n8 = (unsigned)pr * 2u + (unsigned)(q4 >> 1);
          const unsigned chunk = (((unsigned)mt * KT2 + (n8 >> 2)) << 6) + (unsigned)mcol + ((n8 & 3u) << 4);
          ts_store<TS_SC1>(u32x4_t{lo0, lo1, hi0, hi1}, r_act, chunk << 4, 0u);
        }
      }
      ts_barrier_lds();
    }
    ts_publish(p.flags + TS_GRID + b, tag0 | 2u);
  }
  KTRACE(KTS, 5);

  // ---------------- phase 3: down_proj, half row groups (its first weight tiles fly during the edge) ----------------
  {
    const int units = (p.h >> 4) * 2;
    const ts_rsrc_t rw = ts_rsrc(p.Wd, (size_t)p.h * p.I * 2);
    TsGemm<1, MT, 4, true, 2> g;
    if (b < units) { g.init(rw, b >> 1, KTi, b & 1, r_act, M, wave, lane); g.prefetchW(); }
    ts_wait(p.flags + TS_GRID, tag0 | 2u, p, wave, lane);
    if (b < units) g.prefetchX();
    KTRACE(KTS, 6);
    for (int u = b; u < units; u += TS_GRID) {
      f32x4_t acc[1][MT];
      g.run(nullptr, 0, acc, lane);
      if (u + TS_GRID < units) { g.init(rw, (u + TS_GRID) >> 1, KTi, (u + TS_GRID) & 1, r_act, M, wave, lane); g.prefetchW(); g.prefetchX(); }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) cred[(wave * MT + mt) * 64 + lane] = acc[0][mt];
      ts_barrier_lds();
      if (wave < MT) ts_store_rows<MT>(cred, wave, r_d, u >> 1, u & 1, M, p.h, lane);
      ts_barrier_lds();
    }
    if (last) { KTRACE(KTS, 7); return; }     // h_out = the down_proj rows, res_mid = the residual: the final norm adds them
    ts_publish(p.flags + 2 * TS_GRID + b, tag0 | 3u);
  }
  KTRACE(KTS, 7);

  // ---------------- phase 4: the MLP add + the next layer's norm, its QKV + RoPE + KV store ----------------
  {
    const int groups = p.qkv_n >> 4;
    const ts_rsrc_t rw = ts_rsrc(p.Wqkv, (size_t)p.qkv_n * p.h * 2);
    const ts_rsrc_t r_out = ts_rsrc(p.res_out, rows_bytes);
    TsGemm<1, MT, 4, false, 0> g;
    if (b < groups) { g.init(rw, b, KTh, 0, rw, M, wave, lane); g.prefetchW(); }
    ts_wait(p.flags + 2 * TS_GRID, tag0 | 3u, p, wave, lane);
    KTRACE(KTS, 8);
    ts_add_norm<CPL, false>(p, xlds, pitch, r_d, r_mid, r_out, p.ln_next, wave, lane);
    KTRACE(KTS, 9);
    const int gph = p.hd >> 4, qk_groups = (p.nh + p.nkv) * gph, half = p.hd >> 1;
    for (int grp = b; grp < groups; grp += TS_GRID) {
      f32x4_t acc[1][MT];
      g.run(xlds, pitch, acc, lane);
      if (grp + TS_GRID < groups) { g.init(rw, grp + TS_GRID, KTh, 0, rw, M, wave, lane); g.prefetchW(); }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) cred[(wave * MT + mt) * 64 + lane] = acc[0][mt];
      ts_barrier_lds();
      if (wave < MT) {
        // the epilogue of gemm_fused.hip gemm_qkv_rope_m32_kernel for token tile `wave`
        const int mt = wave, m = mt * 16 + mcol, nrow = q4 * 4;
        f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) s += cred[(w * MT + mt) * 64 + lane];
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = round_bf_hw(s[r]);   // the reference stores qkv as bf16 before RoPE
        if (p.qkv_rows) {
          // Qwen3: RMSHeadNorm sits between the projection and the rotation and needs a whole head (8 row groups = 8 workgroups here):
          // the rows leave as the projection wrote them (rotation-paired order) and ssd_rope_store_kv goes on
          if (m < M)
            *reinterpret_cast<u32x2_t*>(p.qkv_rows + (size_t)m * p.qkv_n + grp * 16 + nrow) = u32x2_t{pack_bf2_hw(x[0], x[1]), pack_bf2_hw(x[2], x[3])};
        } else if (grp < qk_groups) {
          const int head = grp / gph, j = grp % gph;
          float other[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(x[r], 32, 64);   // rotation partner: rows i <-> i + 8
          if (m < M) {
            const int hi = q4 >> 1;
            const int d = j * 8 + (q4 & 1) * 4;
            const float* cs = p.cos_sin + (size_t)p.positions[m] * p.hd;
            const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(cs + d), s4 = *reinterpret_cast<const f32x4_t*>(cs + half + d);
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              yv[r] = hi ? __fadd_rn(__fmul_rn(x[r], c4[r]), __fmul_rn(other[r], s4[r]))
                         : __fsub_rn(__fmul_rn(x[r], c4[r]), __fmul_rn(other[r], s4[r]));
            const u32x2_t v = {pack_bf2_hw(yv[0], yv[1]), pack_bf2_hw(yv[2], yv[3])};
            const int dim = hi * half + d;
            if (head < p.nh) {
              *reinterpret_cast<u32x2_t*>(p.q_out + ((size_t)m * p.nh + head) * p.hd + dim) = v;
            } else {
              const int slot = p.slots[m];
              if (slot >= 0) {
                const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + (head - p.nh)) * p.bs + (slot % p.bs);
                *reinterpret_cast<u32x2_t*>(p.k_cache + rowi * p.hd + dim) = v;
              }
            }
          }
        } else if (m < M) {   // V: natural row order, straight to the paged cache
          const int vg = grp - qk_groups;
          const int kvh = vg / gph, dim = (vg % gph) * 16 + nrow;
          const int slot = p.slots[m];
          if (slot >= 0) {
            const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + kvh) * p.bs + (slot % p.bs);
            const u32x2_t v = {pack_bf2_hw(x[0], x[1]), pack_bf2_hw(x[2], x[3])};
            *reinterpret_cast<u32x2_t*>(p.v_cache + rowi * p.hd + dim) = v;
          }
        }
      }
      ts_barrier_lds();
    }
  }
  KTRACE(KTS, 10);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static size_t ts_lds_bytes(int M, int h) { return (size_t)TS_CRED_BYTES + (size_t)(h / 8) * (M + 1) * 16; }

typedef void (*ts_kernel_t)(const TsParams);
static ts_kernel_t ts_pick(int M, int h) {
  const bool two = M > 16;
  if (h == 2048) return two ? tree_segment_kernel<2, 4> : tree_segment_kernel<1, 4>;
  if (h == 1024) return two ? tree_segment_kernel<2, 2> : tree_segment_kernel<1, 2>;
  return nullptr;
}

// All TS_GRID workgroups must be resident at once (they wait for each other): asked of the runtime once per kernel variant.
static bool ts_resident(ts_kernel_t k, size_t lds) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), TS_THREADS, lds) != hipSuccess) return false;
  return (long)per_cu * cus >= TS_GRID;
}

extern "C" int ssd_tree_segment_workspace_bytes(int h, int I) {
  // [flags 3 x 256 words | o_proj rows 32 x h | down_proj rows 32 x h | residual-after-attention rows 32 x h | activation 32 x I]
  return 3 * TS_GRID * 4 + 3 * 32 * h * 2 + 32 * I * 2;
}

extern "C" int ssd_tree_segment_ok(int M, int h, int qn, int I, int qkv_n, int nh, int nkv, int hd) {
  if (M < 1 || M > 32 || (h != 1024 && h != 2048) || (qn & 127) || (I & 127) || I > 16384 || qn <= 0 || I <= 0) return SSD_ERR_SHAPE;
  if ((hd != 64 && hd != 128 && hd != 256) || qkv_n != (nh + 2 * nkv) * hd || qn != nh * hd) return SSD_ERR_SHAPE;
  if (ts_lds_bytes(M, h) > 160 * 1024) return SSD_ERR_SHAPE;
  return SSD_OK;
}

extern "C" int ssd_tree_segment(const void* a_frag, const void* res_in, void* res_out, void* h_out, const void* w_o, const void* w_gu,
                                const void* w_d, const void* w_qkv_next, const void* ln_post, const void* ln_next, float eps,
                                const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache,
                                void* v_cache, void* qkv_rows_next, int M, int h, int qn, int I, int qkv_n, int nh, int nkv, int hd,
                                int block_size, int layer, void* workspace, const void* gen, void* err, void* stream) {
  if (int rc = ssd_tree_segment_ok(M, h, qn, I, qkv_n, nh, nkv, hd)) return rc;
  if (!a_frag || !res_in || !res_out || !w_o || !w_gu || !w_d || !ln_post || !workspace || !gen || !err) return SSD_ERR_ARG;
  if (w_qkv_next && qkv_rows_next ? (!ln_next || positions || cos_sin || slots || q_out || k_cache || v_cache || h_out)
      : w_qkv_next ? (!ln_next || !positions || !cos_sin || !slots || !q_out || !k_cache || !v_cache || h_out) : (!h_out || qkv_rows_next))
    return SSD_ERR_ARG;
  if (res_out == res_in) return SSD_ERR_ARG;               // every workgroup re-reads res_in while chunk owners write res_out
  if (layer < 0 || layer > 63) return SSD_ERR_ARG;
  constexpr long budget = 200000L;         // polls of >= 64 clocks + one memory round trip each: >= 0.2 s before a wait gives up
  const ts_kernel_t kern = ts_pick(M, h);
  if (!kern) return SSD_ERR_SHAPE;
  const size_t lds = ts_lds_bytes(M, h);
  // per (device, kernel variant), asked once per device (at the largest image the variant can be given): the function attribute
  // and the occupancy answer belong to the device that is current at the call (a process-wide flag would skip both
  // on a second device).  0 = not asked yet, 1 = resident, 2 = refused.
  static unsigned char resident[SSD_MAX_DEVICES][4];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SSD_MAX_DEVICES) return SSD_ERR_LAUNCH;
  const int vi = (M > 16 ? 1 : 0) + (h == 1024 ? 2 : 0);
  if (resident[dev][vi] == 0) {
    int mmax = M > 16 ? 32 : 16;
    while (ts_lds_bytes(mmax, h) > 160 * 1024) --mmax;
    resident[dev][vi] = ts_resident(kern, ts_lds_bytes(mmax, h)) ? 1 : 2;
  }
  if (resident[dev][vi] != 1) return SSD_ERR_LAUNCH;
  TsParams p;
  char* ws = (char*)workspace;
  p.flags = (unsigned*)ws; ws += 3 * TS_GRID * 4;
  p.o_rows = (bf16_t*)ws; ws += (size_t)32 * h * 2;
  bf16_t* d_rows = (bf16_t*)ws; ws += (size_t)32 * h * 2;
  bf16_t* mid_rows = (bf16_t*)ws; ws += (size_t)32 * h * 2;
  p.act_f = ws;
  p.a_frag = a_frag; p.res_in = (const bf16_t*)res_in;
  p.res_mid = w_qkv_next ? mid_rows : (bf16_t*)res_out; p.res_out = (bf16_t*)res_out;
  p.d_rows = w_qkv_next ? d_rows : (bf16_t*)h_out;
  p.Wo = w_o; p.Wgu = w_gu; p.Wd = w_d; p.Wqkv = w_qkv_next;
  p.ln_post = (const bf16_t*)ln_post; p.ln_next = (const bf16_t*)ln_next;
  p.positions = positions; p.cos_sin = cos_sin; p.slots = slots;
  p.q_out = (bf16_t*)q_out; p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache; p.qkv_rows = (bf16_t*)qkv_rows_next;
  p.gen = (const unsigned*)gen; p.err = (unsigned*)err;
  p.eps = eps; p.M = M; p.h = h; p.qn = qn; p.I = I; p.qkv_n = qkv_n; p.nh = nh; p.nkv = nkv; p.hd = hd; p.bs = block_size; p.layer = layer;
  p.spin_budget = budget;
  hipLaunchKernelGGL(kern, dim3(TS_GRID), dim3(TS_THREADS), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

KT_DEFINE_SETTER(tree_segment)
