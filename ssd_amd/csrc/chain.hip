// One decoder layer's GEMV chain of a single-token (M = 1, one sequence) forward as ONE launch:
//
//   o_proj -> (+residual, RMSNorm) -> gate_up + SiLU*mul -> down_proj -> (+residual, RMSNorm) -> the NEXT layer's QKV + RoPE + KV store
//
// i.e. everything between two attention launches (reference ssd/models/llama3.py:128-199 LlamaDecoderLayer.forward minus the
// attention call; ssd/layers/linear.py:97-199, layernorm.py:64-88, activation.py:11-14, rotary_embedding.py:40-60,
// attention.py:10-41 store_kvcache).  The single-token chain of the draft (speculator_sync.py:25-69, the JIT chain of
// draft_runner.py:186-378) is a latency chain: four all-to-all edges per layer, each a kernel boundary + a cold first read of the
// producer's output ("the edge model": ~5.5 us per edge against 17.6 us of weight streaming per 1B layer).
//
// MI355X design.  256 workgroups x 8 waves stay resident for the whole chain; an all-to-all edge is an ALL-GATHER of the
// producing phase's finished bf16 vector through data-tagged 8-byte granules {2 x bf16, tag} -- one agent-scope (sc1) store per
// granule, every workgroup polls the granules themselves with agent-scope loads until all tags match: no flag, no fence, no
// cache invalidate (profiles/r04_gridbar2.txt: 2.8 us per 4 KB edge against 1.7 us boundary + cold re-read, and -- the point --
// the NEXT phase's first weight tiles are already in flight while the edge resolves).  Every workgroup therefore holds the whole
// residual stream (fp32, LDS) and repeats the add + RMSNorm itself: nothing but finished GEMV outputs ever crosses workgroups.
//   * o_proj / down_proj have only h / 16 row groups (128 for the 1B): each is split by ROW HALVES over two workgroups -- the
//     lanes of the other half do not load (the 128-byte lines of a fragment-major tile alternate between the halves) -- so that
//     256 workgroups each finish 8 whole rows over the full K: no split-K slabs, no partial sums to re-read.
//   * the tag is (forward generation << 8 | layer << 2 | phase): generation is a device word bumped by ssd_chain_tick once per
//     forward (a kernel argument would be frozen by hipGraph replay), so nothing has to be zeroed between launches.
//   * every wait is bounded; a give-up sets *err (the caller checks it) and lets the launch finish with wrong numbers.
// Same rounding points as the separate kernels (bf16 after every projection, fp32 residual add, bf16 residual, fp32 norm); the
// fp32 summation ORDER inside a projection differs from the split-K slab path it replaces (tolerance-tested, not bit-identical).
#include "common.h"
#include <type_traits>

typedef unsigned long long u64_t;
#define SEG_AGENT __HIP_MEMORY_SCOPE_AGENT
constexpr int SEG_WAVES = 8, SEG_THREADS = SEG_WAVES * 64, SEG_GRID = 256;

struct SegParams {
  const u32x4_t* a_frag;     // attention output of this layer, fragment-major [16][qn] (token row 0)
  const bf16_t* res_in;      // [h] residual entering the layer's attention add
  bf16_t* res_out;           // [h] residual handed on: after the MLP add (layers before the last) / after the attention add (last)
  bf16_t* h_out;             // last layer: down_proj output rows [h] (the final norm adds res_out itself); else null
  const u32x4_t* Wo;
  const u32x4_t* Wgu;
  const u32x4_t* Wd;
  const u32x4_t* Wqkv;       // NEXT layer's fused QKV (rows permuted for the RoPE epilogue, layout.hip mode 2); null on the last layer
  const bf16_t* ln_post;     // this layer's post_attention_layernorm [h]
  const bf16_t* ln_next;     // next layer's input_layernorm [h]
  const int64_t* positions;
  const float* cos_sin;
  const int32_t* slots;
  bf16_t* q_out;             // next layer's attention inputs
  bf16_t* k_cache;
  bf16_t* v_cache;
  u64_t* gr_o;               // granules: h / 2, I / 2, h / 2
  u64_t* gr_act;
  u64_t* gr_d;
  const unsigned* gen;
  unsigned* err;
  float eps;
  int h, qn, I, qkv_n, nh, nkv, hd, bs, layer;
  long spin_budget;
};

template <int NT, int U>
struct SegBuf {
  u32x4_t a[2][U][NT];
  u32x4_t x[2][U];
};

// One workgroup's share of a GEMV: NT adjacent 16-row groups (HALF: 8 rows of one group) over the full K, the k-tiles dealt to
// the 8 waves in groups of U, two groups in flight per wave.  XG: the B operand comes fragment-major from global memory (token
// row 0 only), else from the LDS image of x^ (16-byte chunk k8 at xlds[k8]).
template <int NT, int U, bool HALF, bool XG>
struct SegGemv {
  const u32x4_t* wp;
  const u32x4_t* xg;
  size_t wstride;
  int kt0, kstep, nmain;
  bool wact, xact;
  SegBuf<NT, U> b;

  __device__ __forceinline__ void init(const u32x4_t* W, int g0, int KT, int half, const u32x4_t* Xf, int wave, int lane) {
    wp = W + ((size_t)g0 * KT << 6) + lane;
    wstride = (size_t)KT << 6;
    xg = Xf + lane;
    kstep = SEG_WAVES * U;
    kt0 = wave * U;
    const int ngroups = KT / U;
    nmain = ngroups > wave ? (ngroups - wave + SEG_WAVES - 1) / SEG_WAVES : 0;
    wact = !HALF || (((lane & 15) >> 3) == half);
    xact = (lane & 15) == 0;
  }
  template <int BUF>
  __device__ __forceinline__ void load(int kt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (wact) v = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)(kt + u) << 6));
        b.a[BUF][u][nt] = v;
      }
      if (XG) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (xact) v = xg[(size_t)(kt + u) << 6];
        b.x[BUF][u] = v;
      }
    }
  }
  __device__ __forceinline__ void prefetch() {
    if (nmain > 0) load<0>(kt0);
    if (nmain > 1) load<1>(kt0 + kstep);
  }
  template <int BUF>
  __device__ __forceinline__ void stage(const u32x4_t* xlds, f32x4_t (&acc)[NT], int& kt, int it, int lane) {
    u32x4_t xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (XG) {
        xb[u] = b.x[BUF][u];
      } else {
        u32x4_t o = {0u, 0u, 0u, 0u};
        if (xact) o = xlds[(kt + u) * 4 + (lane >> 4)];
        xb[u] = o;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(b.a[BUF][u][nt], xb[u], acc[nt]);
    if (it + 2 < nmain) load<BUF>(kt + 2 * kstep);
    kt += kstep;
  }
  __device__ __forceinline__ void run(const u32x4_t* xlds, f32x4_t (&acc)[NT], int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int kt = kt0;
    for (int it = 0; it < nmain; it += 2) {
      stage<0>(xlds, acc, kt, it, lane);
      if (it + 1 < nmain) stage<1>(xlds, acc, kt, it + 1, lane);
    }
  }
};

__device__ __forceinline__ void seg_store_granule(u64_t* g, unsigned tag, unsigned value) {
  __hip_atomic_store(g, ((u64_t)tag << 32) | value, __ATOMIC_RELAXED, SEG_AGENT);
}

// Four consecutive granules (= one 8-element bf16 chunk) of a gathered vector: polled until all four carry `tag`.
__device__ __forceinline__ u32x4_t seg_gather_chunk(const u64_t* g, unsigned tag, const SegParams& p) {
  u64_t v[4];
  long spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = __hip_atomic_load(g + k, __ATOMIC_RELAXED, SEG_AGENT);
      ok = ok && (unsigned)(v[k] >> 32) == tag;
    }
    if (ok) break;
    if (++spins > p.spin_budget) { atomicExch(p.err, 1u); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  return u32x4_t{(unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]};
}

// The gathered projection output (bf16) + the residual -> fp32 stream in LDS, the new bf16 residual, the row's sum of squares
// (chunk sums, then lane-strided partials + xor tree: the order of gemm_fused.hip's prologue and of ssd_rmsnorm), and x^ =
// bf16((x * rs) * w) as the next GEMV's B operand.  Every workgroup does the whole row.
struct SegLds {
  f32x4_t* cred;     // [8][2][64] split-K combine area
  float* x32;        // [h]
  float* ssbuf;      // [h / 8] (+ 8 waves x 1 rs behind it)
  u32x4_t* resm;     // [h / 8] bf16 residual after the attention add
  u32x4_t* xlds;     // [h / 8] x^ chunks
  u32x4_t* xact;     // [I / 8] activation chunks
};

template <bool FROM_LDS_RES>
__device__ __forceinline__ void seg_add_norm(const SegParams& p, const SegLds& L, const u64_t* gr, unsigned tag, const bf16_t* res_g,
                                             const bf16_t* ln_w, bf16_t* res_store, bf16_t* h_store, bool do_norm, int wave, int lane) {
  const int K8 = p.h >> 3;
  for (int c = threadIdx.x; c < K8; c += SEG_THREADS) {
    u32x4_t rv;
    if (FROM_LDS_RES) rv = L.resm[c];
    else rv = *reinterpret_cast<const u32x4_t*>(res_g + c * 8);
    const u32x4_t gv = seg_gather_chunk(gr + c * 4, tag, p);
    float x[8], ss = 0.f;
    u32x4_t ro;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[2 * j] = bf2f(gv[j] & 0xffffu) + bf2f(rv[j] & 0xffffu);
      x[2 * j + 1] = bf2f(gv[j] >> 16) + bf2f(rv[j] >> 16);
      ss += x[2 * j] * x[2 * j]; ss += x[2 * j + 1] * x[2 * j + 1];
      ro[j] = pack_bf2(x[2 * j], x[2 * j + 1]);
    }
    reinterpret_cast<f32x4_t*>(L.x32)[c * 2] = f32x4_t{x[0], x[1], x[2], x[3]};
    reinterpret_cast<f32x4_t*>(L.x32)[c * 2 + 1] = f32x4_t{x[4], x[5], x[6], x[7]};
    L.ssbuf[c] = ss;
    if (!FROM_LDS_RES) L.resm[c] = ro;
    const bool mine = (c % SEG_GRID) == (int)blockIdx.x;        // the row's global copies: chunk c by workgroup c mod 256
    if (h_store) {                                              // last layer: down rows + the residual of the attention add stay apart
      if (mine) *reinterpret_cast<u32x4_t*>(h_store + c * 8) = gv;
    } else if (res_store && mine) {
      *reinterpret_cast<u32x4_t*>(res_store + c * 8) = ro;
    }
  }
  __syncthreads();
  if (!do_norm) return;
  float t = 0.f;
  for (int c = lane; c < K8; c += 64) t += L.ssbuf[c];
  t = wave_sum(t);
  const float rs = 1.0f / sqrtf(t / (float)p.h + p.eps);
  for (int c = threadIdx.x; c < K8; c += SEG_THREADS) {
    const u32x4_t wv = *reinterpret_cast<const u32x4_t*>(ln_w + c * 8);
    const f32x4_t a = reinterpret_cast<const f32x4_t*>(L.x32)[c * 2], bq = reinterpret_cast<const f32x4_t*>(L.x32)[c * 2 + 1];
    u32x4_t o;
    o[0] = pack_bf2((a[0] * rs) * bf2f(wv[0] & 0xffffu), (a[1] * rs) * bf2f(wv[0] >> 16));
    o[1] = pack_bf2((a[2] * rs) * bf2f(wv[1] & 0xffffu), (a[3] * rs) * bf2f(wv[1] >> 16));
    o[2] = pack_bf2((bq[0] * rs) * bf2f(wv[2] & 0xffffu), (bq[1] * rs) * bf2f(wv[2] >> 16));
    o[3] = pack_bf2((bq[2] * rs) * bf2f(wv[3] & 0xffffu), (bq[3] * rs) * bf2f(wv[3] >> 16));
    L.xlds[c] = o;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(SEG_THREADS) chain_segment_kernel(const SegParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int mcol = lane & 15, q4 = lane >> 4;
  const bool last = p.Wqkv == nullptr;
  SegLds L;
  {
    char* s = smem;
    L.cred = reinterpret_cast<f32x4_t*>(s); s += SEG_WAVES * 2 * 64 * sizeof(f32x4_t);
    L.x32 = reinterpret_cast<float*>(s); s += (size_t)p.h * 4;
    L.ssbuf = reinterpret_cast<float*>(s); s += (size_t)(p.h >> 3) * 4 + 64;
    L.resm = reinterpret_cast<u32x4_t*>(s); s += (size_t)p.h * 2;
    L.xlds = reinterpret_cast<u32x4_t*>(s); s += (size_t)p.h * 2;
    L.xact = reinterpret_cast<u32x4_t*>(s);
  }
  const int KTS = last ? 13 : 12;          // trace slots (profiling builds only)
  KTRACE(KTS, 0);
  const unsigned tag0 = ((*p.gen & 0xffffffu) << 8) | ((unsigned)p.layer << 2);
  const int KTq = p.qn >> 5, KTh = p.h >> 5, KTi = p.I >> 5;

  // ---------------- phase 1: o_proj, half row groups ----------------
  {
    const int units = (p.h >> 4) * 2;
    SegGemv<1, 4, true, true> g;
    for (int u = b; u < units; u += SEG_GRID) {
      const int grp = u >> 1, half = u & 1;
      g.init(p.Wo, grp, KTq, half, p.a_frag, wave, lane);
      g.prefetch();
      f32x4_t acc[1];
      g.run(nullptr, acc, lane);
      L.cred[wave * 64 + lane] = acc[0];
      __syncthreads();
      if (wave == 0 && mcol == 0 && (q4 >> 1) == half) {
        f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < SEG_WAVES; ++w) s += L.cred[w * 64 + lane];
        const int n = grp * 16 + q4 * 4;
        seg_store_granule(p.gr_o + (n >> 1), tag0 | 1u, pack_bf2(s[0], s[1]));
        seg_store_granule(p.gr_o + (n >> 1) + 1, tag0 | 1u, pack_bf2(s[2], s[3]));
      }
      __syncthreads();
    }
  }

  KTRACE(KTS, 1);
  // ---------------- phase 2: gate_up + SiLU * mul (first unit's weights fly during the edge) ----------------
  const int pairs = p.I >> 4;          // (gate, up) row-group pairs
  SegGemv<2, 4, false, false> g2;
  if (b < pairs) { g2.init(p.Wgu, 2 * b, KTh, 0, nullptr, wave, lane); g2.prefetch(); }
  seg_add_norm<false>(p, L, p.gr_o, tag0 | 1u, p.res_in, p.ln_post, last ? p.res_out : nullptr, nullptr, true, wave, lane);
  KTRACE(KTS, 2);
  for (int pr = b; pr < pairs; pr += SEG_GRID) {
    f32x4_t acc[2];
    g2.run(L.xlds, acc, lane);
    if (pr + SEG_GRID < pairs) { g2.init(p.Wgu, 2 * (pr + SEG_GRID), KTh, 0, nullptr, wave, lane); g2.prefetch(); }
    L.cred[(wave * 2) * 64 + lane] = acc[0];
    L.cred[(wave * 2 + 1) * 64 + lane] = acc[1];
    __syncthreads();
    if (wave == 0 && mcol == 0) {
      f32x4_t gs = f32x4_t{0.f, 0.f, 0.f, 0.f}, us = gs;
      for (int w = 0; w < SEG_WAVES; ++w) { gs += L.cred[(w * 2) * 64 + lane]; us += L.cred[(w * 2 + 1) * 64 + lane]; }
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gb = round_bf(gs[r]), ub = round_bf(us[r]);
        o[r] = (gb / (1.0f + __expf(-gb))) * ub;
      }
      const int n = pr * 16 + q4 * 4;
      seg_store_granule(p.gr_act + (n >> 1), tag0 | 2u, pack_bf2(o[0], o[1]));
      seg_store_granule(p.gr_act + (n >> 1) + 1, tag0 | 2u, pack_bf2(o[2], o[3]));
    }
    __syncthreads();
  }

  KTRACE(KTS, 3);
  // ---------------- phase 3: down_proj, half row groups (first tiles fly during the edge) ----------------
  {
    const int units = (p.h >> 4) * 2;
    SegGemv<1, 4, true, false> g;
    if (b < units) { g.init(p.Wd, b >> 1, KTi, b & 1, nullptr, wave, lane); g.prefetch(); }
    for (int c = threadIdx.x; c < (p.I >> 3); c += SEG_THREADS) L.xact[c] = seg_gather_chunk(p.gr_act + c * 4, tag0 | 2u, p);
    __syncthreads();
    KTRACE(KTS, 4);
    for (int u = b; u < units; u += SEG_GRID) {
      const int grp = u >> 1, half = u & 1;
      f32x4_t acc[1];
      g.run(L.xact, acc, lane);
      if (u + SEG_GRID < units) { g.init(p.Wd, (u + SEG_GRID) >> 1, KTi, (u + SEG_GRID) & 1, nullptr, wave, lane); g.prefetch(); }
      L.cred[wave * 64 + lane] = acc[0];
      __syncthreads();
      if (wave == 0 && mcol == 0 && (q4 >> 1) == half) {
        f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < SEG_WAVES; ++w) s += L.cred[w * 64 + lane];
        const int n = grp * 16 + q4 * 4;
        seg_store_granule(p.gr_d + (n >> 1), tag0 | 3u, pack_bf2(s[0], s[1]));
        seg_store_granule(p.gr_d + (n >> 1) + 1, tag0 | 3u, pack_bf2(s[2], s[3]));
      }
      __syncthreads();
    }
  }

  KTRACE(KTS, 5);
  // ---------------- the MLP add (+ next layer's norm), then phase 4: next layer's QKV + RoPE + KV store ----------------
  if (last) {
    seg_add_norm<true>(p, L, p.gr_d, tag0 | 3u, nullptr, nullptr, nullptr, p.h_out, false, wave, lane);
    KTRACE(KTS, 6);
    return;
  }
  const int groups = p.qkv_n >> 4;
  SegGemv<1, 4, false, false> g4;
  if (b < groups) { g4.init(p.Wqkv, b, KTh, 0, nullptr, wave, lane); g4.prefetch(); }
  seg_add_norm<true>(p, L, p.gr_d, tag0 | 3u, nullptr, p.ln_next, p.res_out, nullptr, true, wave, lane);
  KTRACE(KTS, 6);
  for (int grp = b; grp < groups; grp += SEG_GRID) {
    f32x4_t acc[1];
    g4.run(L.xlds, acc, lane);
    if (grp + SEG_GRID < groups) { g4.init(p.Wqkv, grp + SEG_GRID, KTh, 0, nullptr, wave, lane); g4.prefetch(); }
    L.cred[wave * 64 + lane] = acc[0];
    __syncthreads();
    if (wave == 0) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < SEG_WAVES; ++w) s += L.cred[w * 64 + lane];
      // the epilogue of gemm_fused.hip FEPI_QKV_ROPE for token row 0
      const int gph = p.hd >> 4, qk_groups = (p.nh + p.nkv) * gph, half = p.hd >> 1;
      float x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = round_bf(s[r]);
      if (grp < qk_groups) {
        const int head = grp / gph, j = grp % gph;
        float other[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(x[r], 32, 64);
        if (mcol == 0) {
          const int hi = q4 >> 1, d = j * 8 + (q4 & 1) * 4;
          const float* cs = p.cos_sin + (size_t)p.positions[0] * p.hd;
          float yv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float co = cs[d + r], si = cs[half + d + r];
            yv[r] = hi ? __fadd_rn(__fmul_rn(x[r], co), __fmul_rn(other[r], si))
                       : __fsub_rn(__fmul_rn(x[r], co), __fmul_rn(other[r], si));
          }
          const u32x2_t v = {pack_bf2(yv[0], yv[1]), pack_bf2(yv[2], yv[3])};
          const int dim = hi * half + d;
          if (head < p.nh) {
            *reinterpret_cast<u32x2_t*>(p.q_out + (size_t)head * p.hd + dim) = v;
          } else {
            const int slot = p.slots[0];
            if (slot >= 0) {
              const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + (head - p.nh)) * p.bs + (slot % p.bs);
              *reinterpret_cast<u32x2_t*>(p.k_cache + rowi * p.hd + dim) = v;
            }
          }
        }
      } else if (mcol == 0) {
        const int vg = grp - qk_groups;
        const int kvh = vg / gph, dim = (vg % gph) * 16 + q4 * 4;
        const int slot = p.slots[0];
        if (slot >= 0) {
          const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + kvh) * p.bs + (slot % p.bs);
          const u32x2_t v = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
          *reinterpret_cast<u32x2_t*>(p.v_cache + rowi * p.hd + dim) = v;
        }
      }
    }
    __syncthreads();
  }
  KTRACE(KTS, 7);
}

__global__ void chain_tick_kernel(unsigned* gen) { *gen = *gen + 1u; }

extern "C" int ssd_chain_tick(void* gen, void* stream) {
  if (!gen) return SSD_ERR_ARG;
  hipLaunchKernelGGL(chain_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)gen);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Bytes of the granule area (gr_o | gr_act | gr_d), to be zeroed ONCE at allocation.
extern "C" int ssd_chain_granule_bytes(int h, int I) { return (h / 2 + I / 2 + h / 2) * 8; }

extern "C" int ssd_chain_segment_ok(int h, int qn, int I, int qkv_n, int nh, int nkv, int hd) {
  if (h <= 0 || (h & 127) || (qn & 127) || (I & 127) || (qkv_n & 15) || h > 4096 || I > 16384) return SSD_ERR_SHAPE;
  if ((hd & 15) || qkv_n != (nh + 2 * nkv) * hd || qn != nh * hd) return SSD_ERR_SHAPE;
  return SSD_OK;
}

extern "C" int ssd_chain_segment(const void* a_frag, const void* res_in, void* res_out, void* h_out, const void* w_o, const void* w_gu,
                                 const void* w_d, const void* w_qkv_next, const void* ln_post, const void* ln_next, float eps,
                                 const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache,
                                 void* v_cache, int h, int qn, int I, int qkv_n, int nh, int nkv, int hd, int block_size, int layer,
                                 void* granules, const void* gen, void* err, void* stream) {
  if (int rc = ssd_chain_segment_ok(h, qn, I, qkv_n, nh, nkv, hd)) return rc;
  if (!a_frag || !res_in || !res_out || !w_o || !w_gu || !w_d || !ln_post || !granules || !gen || !err) return SSD_ERR_ARG;
  if (w_qkv_next ? (!ln_next || !positions || !cos_sin || !slots || !q_out || !k_cache || !v_cache || h_out) : !h_out) return SSD_ERR_ARG;
  if (layer < 0 || layer > 63) return SSD_ERR_ARG;
  if (res_out == res_in) return SSD_ERR_ARG;      // every workgroup reads the whole of res_in while chunk owners write res_out
  constexpr long budget = 200000L;         // polls of >= 64 clocks + one memory round trip each: >= 0.2 s before a wait gives up
  SegParams p;
  p.a_frag = (const u32x4_t*)a_frag; p.res_in = (const bf16_t*)res_in; p.res_out = (bf16_t*)res_out; p.h_out = (bf16_t*)h_out;
  p.Wo = (const u32x4_t*)w_o; p.Wgu = (const u32x4_t*)w_gu; p.Wd = (const u32x4_t*)w_d; p.Wqkv = (const u32x4_t*)w_qkv_next;
  p.ln_post = (const bf16_t*)ln_post; p.ln_next = (const bf16_t*)ln_next;
  p.positions = positions; p.cos_sin = cos_sin; p.slots = slots;
  p.q_out = (bf16_t*)q_out; p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache;
  p.gr_o = (u64_t*)granules; p.gr_act = p.gr_o + h / 2; p.gr_d = p.gr_act + I / 2;
  p.gen = (const unsigned*)gen; p.err = (unsigned*)err;
  p.eps = eps; p.h = h; p.qn = qn; p.I = I; p.qkv_n = qkv_n; p.nh = nh; p.nkv = nkv; p.hd = hd; p.bs = block_size; p.layer = layer;
  p.spin_budget = budget;
  const size_t lds = (size_t)SEG_WAVES * 2 * 64 * sizeof(f32x4_t) + (size_t)h * 4 + (size_t)(h / 8) * 4 + 64 + (size_t)h * 2 + (size_t)h * 2 +
                     (size_t)I * 2;
  if (lds > 96 * 1024) return SSD_ERR_SHAPE;
  // The 256 workgroups wait for each other: they must ALL be resident at once.  Asked of the runtime once (at the largest LDS image the
  // kernel can be given): occupancy x compute units >= the grid, else refuse -- a partitioned or smaller device would spin every
  // gather to its budget.
  static unsigned char resident[SSD_MAX_DEVICES];      // per device: 0 = not asked yet, 1 = resident, 2 = refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SSD_MAX_DEVICES) return SSD_ERR_LAUNCH;
  if (resident[dev] == 0) {
    int cus = 0, per_cu = 0;
    resident[dev] = (hipFuncSetAttribute(reinterpret_cast<const void*>(chain_segment_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess &&
                     hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                     hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(chain_segment_kernel), SEG_THREADS, 96 * 1024) == hipSuccess &&
                     (long)per_cu * cus >= SEG_GRID) ? 1 : 2;
  }
  if (resident[dev] != 1) return SSD_ERR_LAUNCH;
  hipLaunchKernelGGL(chain_segment_kernel, dim3(SEG_GRID), dim3(SEG_THREADS), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

KT_DEFINE_SETTER(chain)
