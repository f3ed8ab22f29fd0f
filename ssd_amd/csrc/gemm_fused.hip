// Fused decode-layer GEMMs for M <= 16 rows (single-token draft decode, K+1-token verify at TP = 1):
//
//   [residual add + RMSNorm]  ->  skinny GEMM  ->  [RoPE + paged KV store | SiLU*mul | rows]
//
// One launch replaces add_norm_forward + F.linear + rotary_emb + store_kvcache (reference
// ssd/layers/layernorm.py:64-88, ssd/layers/linear.py:97-98, ssd/layers/rotary_embedding.py:40-60,
// ssd/layers/attention.py:10-41) for the QKV projection, and add_norm_forward + F.linear + SiluAndMul
// (ssd/layers/activation.py:11-14) for gate_up.  A decode layer becomes five launches
// (norm+qkv+rope | attention | o | norm+gate_up+silu | down) instead of nine: at M = 1 the 1B draft is
// launch-latency bound, so removing launches is worth more than any bandwidth tuning.
//
// Prologue (XNORM): every workgroup recomputes rs[m] = rsqrt(mean((h+res)^2) + eps) for its M rows from the
// row-major bf16 h / res (<= 16 x K x 4 B, L2 resident) while its first weight tiles are in flight, then forms
// the MFMA B operand on the fly: x^[m][k] = bf16((h+res) * rs[m] * w[k]) -- identical arithmetic and
// rounding points to ssd_rmsnorm.  The new residual bf16(h+res) is written by the workgroups in disjoint
// column slices into a SECOND buffer (res_out != res_in: other workgroups still read res_in).
//
// RoPE epilogue: the QKV weight rows were permuted at load (ssd_rows_to_frag mode 2) so that each 16-row
// group of a q/k head holds dims [d0..d0+7] and [d0+hd/2..d0+hd/2+7]: the two halves of a rotation pair sit in
// lanes l and l^32 of the accumulator and are exchanged with one shuffle.
#include "common.h"
#include <type_traits>

enum { FEPI_ROWS = 0, FEPI_SILU_FRAG = 1, FEPI_QKV_ROPE = 3 };

struct FusedParams {
  const u32x4_t* Wf;
  const u32x4_t* Xf;        // fragment-major x (when h == nullptr)
  const bf16_t* h;          // row-major [M][K] (XNORM)
  const float* h_parts;     // XNORM alternative to h: fp32 [S][M][K] split-K partial sums of the producer GEMM (gemm_sk.hip
  int S;                    //   ssd_gemm_parts); h = bf16(sum over s, in s order) is formed here
  const bf16_t* res_in;     // row-major [M][K] or nullptr
  bf16_t* res_out;          // row-major [M][K] or nullptr
  const bf16_t* norm_w;     // [K]
  const bf16_t* bias;       // [N] (in the shuffled row order) or nullptr
  void* y;                  // FEPI_ROWS: bf16 rows [M][ldy]; FEPI_SILU_FRAG: frag [M][N/2]
  const int64_t* positions; // rope
  const float* cos_sin;
  const int32_t* slots;
  bf16_t* q_out;
  bf16_t* k_cache;
  bf16_t* v_cache;
  float eps;
  int M, N, K, ldy;
  int nh, nkv, hd, bs;
  int scratch_bytes;        // LDS bytes in front of the x^ image (split-K combine area, also the prologue scratch)
};

// 8-element chunks of (h + res) a thread may hold in registers during the prologue: template parameter MAXC (1 when M*K/8
// fits the workgroup's threads -- the single-token draft decode -- which keeps the kernel clear of the 128-VGPR ceiling)

// x32[0..7] = fp32(h) + fp32(res) for chunk k8 of row mm, where h is either the producer's bf16 rows or the bf16 rounding
// of the sum of its S fp32 partial slabs (summed in slab order: deterministic; the rounding mirrors the bf16 store of the
// reference's F.linear).  All loads of a chunk are issued before any arithmetic.
__device__ __forceinline__ void fused_load_x32(const FusedParams& p, int mm, int k8, float (&x)[8]) {
  u32x4_t rv = {0u, 0u, 0u, 0u};
  if (p.res_in) rv = *reinterpret_cast<const u32x4_t*>(p.res_in + (size_t)mm * p.K + k8 * 8);
  float hf[8];
  if (p.h_parts) {
    f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    for (int sidx = 0; sidx < p.S; ++sidx) {
      const float* src = p.h_parts + ((size_t)sidx * p.M + mm) * p.K + k8 * 8;
      lo += *reinterpret_cast<const f32x4_t*>(src);
      hi += *reinterpret_cast<const f32x4_t*>(src + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { hf[j] = round_bf(lo[j]); hf[4 + j] = round_bf(hi[j]); }
  } else {
    const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(p.h + (size_t)mm * p.K + k8 * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) { hf[2 * j] = bf2f(hv[j] & 0xffffu); hf[2 * j + 1] = bf2f(hv[j] >> 16); }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[2 * j] = hf[2 * j] + bf2f(rv[j] & 0xffffu);
    x[2 * j + 1] = hf[2 * j + 1] + bf2f(rv[j] >> 16);
  }
}

// (Round 6 measured a DEEP form of this kernel -- twice the k-tiles per group at 8 waves, as gemm.hip's -- for the 70B QKV launch: the
//  same 29.5 us in isolation and the same c4 step; 8 PLAIN waves cost the step 0.4 ms.  Not kept: profiles/r06_deep_probe.txt,
//  profiles/r06_c4_deep_ab_same_box.txt.)
template <int NT, int EPI, bool XNORM, int FUSED_MAXC>
__global__ void __launch_bounds__(1024) gemm_fused_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KTS = !XNORM ? 2 : (EPI == FEPI_QKV_ROPE ? 0 : 1);      // trace slot (profiling builds only)
  KTRACE(KTS, 0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int M = p.M, K = p.K;
  const int KT = K >> 5, K8 = K >> 3;
  const int tile0 = blockIdx.x * NT;
  const int mcol = lane & 15, q4 = lane >> 4;
  // LDS: [combine / prologue scratch: nw*NT KiB (>= M*K8*4 B)] [x^ image: M*K*2 B, chunk (k8, m) at (k8*M + m)*16 B]
  u32x4_t* xlds = reinterpret_cast<u32x4_t*>(smem + p.scratch_bytes);

  f32x4_t acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const u32x4_t* wp = p.Wf + ((size_t)tile0 * KT << 6) + lane;
  const size_t wstride = (size_t)KT << 6;

  constexpr int U = (NT <= 2) ? 4 : 2;
  u32x4_t wa[2][U][NT];
  u32x4_t xg[2][U];       // fragment-major x travels with the weights when there is no prologue
  auto loadw = [&](int buf, int kt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wa[buf][u][nt] = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)(kt + u) << 6));
      if (!XNORM) {     // padding token rows (>= M) are not loaded: halves the B-operand bytes at M = 7
        u32x4_t b = {0u, 0u, 0u, 0u};
        if (mcol < M) b = p.Xf[((size_t)(kt + u) << 6) + lane];
        xg[buf][u] = b;
      }
    }
  };
  // K is dealt to the waves in groups of U k-tiles, round-robin (wave w: groups w, w + nw, ...): the workgroup walks each
  // row group's K run linearly (see gemm.hip / profiles/micro/readpat.hip); the < U left-over k-tiles go to the last wave
  const int kstep = nw * U;
  const int kt0 = wave * U;
  const int ngroups = KT / U;
  const int nmain = ngroups > wave ? (ngroups - wave + nw - 1) / nw : 0;   // groups of this wave
  // the first TWO groups of weight tiles fly while the norm prologue runs (for the 1B draft that is the whole K range:
  // one HBM round trip per wave; a kernel this short is a latency chain)
  if (nmain > 0) loadw(0, kt0);
  if (nmain > 1) loadw(1, kt0 + kstep);
  // RoPE epilogue operands of the wave that will own row group `wave` (positions -> cos/sin rows -> slot: a chain of
  // dependent L2 round trips if left to the epilogue), fetched now, behind the weight stream
  float pre_cs[8];
  int pre_slot = -1;
  bool pre_ok = false;
  if (EPI == FEPI_QKV_ROPE && wave < NT && mcol < M) {
    const int grp = tile0 + wave;
    const int gph = p.hd >> 4;
    pre_slot = p.slots[mcol];
    if (grp < (p.nh + p.nkv) * gph) {
      const int d = (grp % gph) * 8 + (q4 & 1) * 4;
      const float* cs = p.cos_sin + (size_t)p.positions[mcol] * p.hd;
      const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(cs + d), s4 = *reinterpret_cast<const f32x4_t*>(cs + (p.hd >> 1) + d);
#pragma unroll
      for (int r = 0; r < 4; ++r) { pre_cs[r] = c4[r]; pre_cs[4 + r] = s4[r]; }
    }
    pre_ok = true;
  }
  KTRACE(KTS, 1);
  if (XNORM) {
    // ---- prologue: x32 = h + res kept in registers; per-chunk sums of squares -> per-row rs (fixed order) ->
    //      x^ = bf16(x32 * rs * w) into the LDS image; residual slice written to res_out ----
    float* ssbuf = reinterpret_cast<float*>(smem);      // [M*K8] chunk sums, then [nw][16] per-wave copies of rs
    constexpr int MC = FUSED_MAXC > 0 ? FUSED_MAXC : 1;     // MAXC = 0: no register cache, two passes over the L2-resident rows
    float x32[MC][8];
    u32x4_t wv[MC];
    const int total = M * K8;
    const bool cached = FUSED_MAXC > 0 && total <= FUSED_MAXC * (int)blockDim.x;     // block-uniform
    const int cpb = (K8 + gridDim.x - 1) / gridDim.x;              // residual: workgroup b owns chunk columns [b*cpb, (b+1)*cpb)
    if (!cached) {   // large M*K (or few waves): two passes over the L2-resident rows instead of registers
      for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int mm = c / K8, k8 = c % K8;
        float xx[8];
        fused_load_x32(p, mm, k8, xx);
        float ss = 0.f;
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ss += xx[2 * j] * xx[2 * j]; ss += xx[2 * j + 1] * xx[2 * j + 1];
          o[j] = pack_bf2(xx[2 * j], xx[2 * j + 1]);
        }
        ssbuf[c] = ss;
        if (p.res_out && k8 / cpb == (int)blockIdx.x) *reinterpret_cast<u32x4_t*>(p.res_out + (size_t)mm * K + k8 * 8) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) {
      const int c = threadIdx.x + i * blockDim.x;
      if (cached && c < total) {
        const int mm = c / K8, k8 = c % K8;
        fused_load_x32(p, mm, k8, x32[i]);
        wv[i] = *reinterpret_cast<const u32x4_t*>(p.norm_w + k8 * 8);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ss += x32[i][2 * j] * x32[i][2 * j]; ss += x32[i][2 * j + 1] * x32[i][2 * j + 1]; }
        ssbuf[c] = ss;
        if (p.res_out && k8 / cpb == (int)blockIdx.x) {
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = pack_bf2(x32[i][2 * j], x32[i][2 * j + 1]);
          *reinterpret_cast<u32x4_t*>(p.res_out + (size_t)mm * K + k8 * 8) = o;
        }
      }
    }
    __syncthreads();
    KTRACE(KTS, 2);
    // every wave reduces the chunk sums of all M rows itself (same order as ssd_rmsnorm: lane-strided partials, then the
    // xor tree) and keeps rs in its own LDS row: no second workgroup barrier
    float* rsbuf = ssbuf + total + wave * 16;           // behind the chunk sums
    for (int mm = 0; mm < M; ++mm) {
      float t = 0.f;
      for (int c = lane; c < K8; c += 64) t += ssbuf[mm * K8 + c];
      t = wave_sum(t);
      if (lane == 0) rsbuf[mm] = 1.0f / sqrtf(t / (float)K + p.eps);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!cached) {
      for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int mm = c / K8, k8 = c % K8;
        const float rs = rsbuf[mm];
        float xx[8];
        fused_load_x32(p, mm, k8, xx);
        const u32x4_t nwv = *reinterpret_cast<const u32x4_t*>(p.norm_w + k8 * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = pack_bf2((xx[2 * j] * rs) * bf2f(nwv[j] & 0xffffu), (xx[2 * j + 1] * rs) * bf2f(nwv[j] >> 16));
        xlds[k8 * M + mm] = o;
      }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) {
      const int c = threadIdx.x + i * blockDim.x;
      if (cached && c < total) {
        const int mm = c / K8, k8 = c % K8;
        const float rs = rsbuf[mm];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = pack_bf2((x32[i][2 * j] * rs) * bf2f(wv[i][j] & 0xffffu), (x32[i][2 * j + 1] * rs) * bf2f(wv[i][j] >> 16));
        xlds[k8 * M + mm] = o;
      }
    }
    __syncthreads();
    KTRACE(KTS, 3);
  }

  auto xfrag = [&](int buf, int u, int kt) -> u32x4_t {
    if (!XNORM) return xg[buf][u];
    u32x4_t o = {0u, 0u, 0u, 0u};
    if (mcol < M) o = xlds[(kt * 4 + q4) * M + mcol];
    return o;
  };

  // ---- main loop: two groups of weight tiles (and fragment-major x) are in flight from the start; the buffer just
  //      consumed is refilled with the group two steps ahead ----
  int kt = kt0;
  auto stage = [&](auto curc, int it) {          // curc: compile-time buffer index (runtime-indexed register
    constexpr int cur = decltype(curc)::value;   // arrays would go to scratch)
    u32x4_t xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xb[u] = xfrag(cur, u, kt + u);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wa[cur][u][nt], xb[u], acc[nt]);
    if (it + 2 < nmain) loadw(cur, kt + 2 * kstep);
    kt += kstep;
  };
  for (int it = 0; it < nmain; it += 2) {
    stage(std::integral_constant<int, 0>{}, it);
    if (it + 1 < nmain) stage(std::integral_constant<int, 1>{}, it + 1);
  }
  for (kt = (wave == nw - 1) ? ngroups * U : KT; kt < KT; ++kt) {
    u32x4_t xb;
    if (!XNORM) { xb = u32x4_t{0u, 0u, 0u, 0u}; if (mcol < M) xb = p.Xf[((size_t)kt << 6) + lane]; }
    else { xb = u32x4_t{0u, 0u, 0u, 0u}; if (mcol < M) xb = xlds[(kt * 4 + q4) * M + mcol]; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[nt] = mfma16(__builtin_nontemporal_load(wp + nt * wstride + ((size_t)kt << 6)), xb, acc[nt]);
  }

  // ---- split-K combine through LDS in wave order ----
  f32x4_t* cred = reinterpret_cast<f32x4_t*>(smem);   // [nw][NT][64]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) cred[(wave * NT + nt) * 64 + lane] = acc[nt];
  KTRACE(KTS, 4);
  __syncthreads();
  KTRACE(KTS, 5);
  const int nrow = q4 * 4;
  const int m = mcol;

  if (EPI == FEPI_SILU_FRAG) {
    constexpr int PAIRS = NT / 2;
    const int KT2 = (p.N >> 1) >> 5;
    u32x2_t* out = reinterpret_cast<u32x2_t*>(p.y);
    for (int pr = wave; pr < PAIRS; pr += nw) {
      f32x4_t g = f32x4_t{0.f, 0.f, 0.f, 0.f}, u = g;
      for (int w = 0; w < nw; ++w) {
        g += cred[(w * NT + 2 * pr) * 64 + lane];
        u += cred[(w * NT + 2 * pr + 1) * 64 + lane];
      }
      const int n = ((tile0 >> 1) + pr) * 16 + nrow;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gb = round_bf(g[r]), ub = round_bf(u[r]);
        o[r] = (gb / (1.0f + __expf(-gb))) * ub;
      }
      if (m < M) {
        const u32x2_t v = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
        out[frag_chunk(m, n >> 3, KT2) * 2 + ((n >> 2) & 1)] = v;
      }
    }
  } else {
    for (int nt = wave; nt < NT; nt += nw) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < nw; ++w) s += cred[(w * NT + nt) * 64 + lane];
      const int grp = tile0 + nt;
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += bf2f(p.bias[grp * 16 + nrow + r]);
      }
      if (EPI == FEPI_ROWS) {
        if (m < M) {
          const u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
          *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(p.y) + (size_t)m * p.ldy + grp * 16 + nrow) = v;
        }
      } else {  // FEPI_QKV_ROPE
        const int gph = p.hd >> 4;                 // 16-row groups per head
        const int qk_groups = (p.nh + p.nkv) * gph;
        const int half = p.hd >> 1;
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = round_bf(s[r]);   // the reference stores qkv as bf16 before RoPE
        if (grp < qk_groups) {
          const int head = grp / gph, j = grp % gph;
          float other[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(x[r], 32, 64);   // rotation partner: rows i <-> i + 8
          if (m < M) {
            const int hi = q4 >> 1;                       // 0: first half of the head dim (x1), 1: second half (x2)
            const int d = j * 8 + (q4 & 1) * 4;           // dim within the half
            float cs8[8];
            if (nt == wave && pre_ok) {
#pragma unroll
              for (int r = 0; r < 8; ++r) cs8[r] = pre_cs[r];
            } else {
              const float* cs = p.cos_sin + (size_t)p.positions[m] * p.hd;
#pragma unroll
              for (int r = 0; r < 4; ++r) { cs8[r] = cs[d + r]; cs8[4 + r] = cs[half + d + r]; }
            }
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float co = cs8[r], si = cs8[4 + r];
              yv[r] = hi ? __fadd_rn(__fmul_rn(x[r], co), __fmul_rn(other[r], si))
                         : __fsub_rn(__fmul_rn(x[r], co), __fmul_rn(other[r], si));
            }
            const u32x2_t v = {pack_bf2(yv[0], yv[1]), pack_bf2(yv[2], yv[3])};
            const int dim = hi * half + d;
            if (head < p.nh) {
              *reinterpret_cast<u32x2_t*>(p.q_out + ((size_t)m * p.nh + head) * p.hd + dim) = v;
            } else {
              const int slot = (nt == wave && pre_ok) ? pre_slot : p.slots[m];
              if (slot >= 0) {
                const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + (head - p.nh)) * p.bs + (slot % p.bs);
                *reinterpret_cast<u32x2_t*>(p.k_cache + rowi * p.hd + dim) = v;
              }
            }
          }
        } else if (m < M) {   // V: natural row order, straight to the paged cache
          const int vg = grp - qk_groups;
          const int kvh = vg / gph, dim = (vg % gph) * 16 + nrow;
          const int slot = (nt == wave && pre_ok) ? pre_slot : p.slots[m];
          if (slot >= 0) {
            const size_t rowi = ((size_t)(slot / p.bs) * p.nkv + kvh) * p.bs + (slot % p.bs);
            const u32x2_t v = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
            *reinterpret_cast<u32x2_t*>(p.v_cache + rowi * p.hd + dim) = v;
          }
        }
      }
    }
  }
  KTRACE(KTS, 6);
}

// ---------------------------------------------------------------------------------------------------------------------
// QKV GEMM + RoPE + paged KV store for 17..32 token rows (the 24-branch tree-decode step of asynchronous speculation,
// reference ssd/engine/draft_runner.py:713-812): the same epilogue as above over TWO 16-row token tiles, x in the
// fragment-major layout (no norm prologue at this size: the normalised x comes from ssd_rmsnorm / ssd_rmsnorm_parts).
// One launch instead of F.linear + rotary_emb + store_kvcache.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(1024) gemm_qkv_rope_m32_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  KTRACE(3, 0);
  constexpr int MT = 2, U = 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int M = p.M, KT = p.K >> 5;
  const int tile0 = blockIdx.x * NT;
  const int mcol = lane & 15, q4 = lane >> 4;
  const u32x4_t* wp = p.Wf + ((size_t)tile0 * KT << 6) + lane;
  const size_t wstride = (size_t)KT << 6, xstride = (size_t)KT << 6;
  f32x4_t acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t wa[2][U][NT], xg[2][U][MT];
  auto loadw = [&](int buf, int kt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wa[buf][u][nt] = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)(kt + u) << 6));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4_t b = {0u, 0u, 0u, 0u};
        if (mt * 16 + mcol < M) b = p.Xf[mt * xstride + ((size_t)(kt + u) << 6) + lane];
        xg[buf][u][mt] = b;
      }
    }
  };
  const int kstep = nw * U, kt0 = wave * U, ngroups = KT / U;
  const int nmain = ngroups > wave ? (ngroups - wave + nw - 1) / nw : 0;
  if (nmain > 0) loadw(0, kt0);
  if (nmain > 1) loadw(1, kt0 + kstep);
  int kt = kt0;
  auto stage = [&](auto curc, int it) {
    constexpr int cur = decltype(curc)::value;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(wa[cur][u][nt], xg[cur][u][mt], acc[nt][mt]);
    if (it + 2 < nmain) loadw(cur, kt + 2 * kstep);
    kt += kstep;
  };
  for (int it = 0; it < nmain; it += 2) {
    stage(std::integral_constant<int, 0>{}, it);
    if (it + 1 < nmain) stage(std::integral_constant<int, 1>{}, it + 1);
  }
  for (kt = (wave == nw - 1) ? ngroups * U : KT; kt < KT; ++kt) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      u32x4_t xb = {0u, 0u, 0u, 0u};
      if (mt * 16 + mcol < M) xb = p.Xf[mt * xstride + ((size_t)kt << 6) + lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt][mt] = mfma16(__builtin_nontemporal_load(wp + nt * wstride + ((size_t)kt << 6)), xb, acc[nt][mt]);
    }
  }
  f32x4_t* cred = reinterpret_cast<f32x4_t*>(smem);   // [nw][NT*MT][64]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cred[(wave * NT * MT + nt * MT + mt) * 64 + lane] = acc[nt][mt];
  KTRACE(3, 4);
  __syncthreads();
  KTRACE(3, 5);
  const int nrow = q4 * 4;
  const int gph = p.hd >> 4, qk_groups = (p.nh + p.nkv) * gph, half = p.hd >> 1;
  for (int item = wave; item < NT * MT; item += nw) {
    const int nt = item / MT, mt = item % MT;
    f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < nw; ++w) s += cred[(w * NT * MT + item) * 64 + lane];
    const int grp = tile0 + nt;
    const int m = mt * 16 + mcol;
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += bf2f(p.bias[grp * 16 + nrow + r]);
    }
    float x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = round_bf(s[r]);   // the reference stores qkv as bf16 before RoPE
    if (grp < qk_groups) {
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
        const u32x2_t v = {pack_bf2(yv[0], yv[1]), pack_bf2(yv[2], yv[3])};
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
        const u32x2_t v = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
        *reinterpret_cast<u32x2_t*>(p.v_cache + rowi * p.hd + dim) = v;
      }
    }
  }
  KTRACE(3, 6);
}

template <int NT>
static int launch_qkv_m32(const FusedParams& p, int waves, hipStream_t st) {
  const size_t lds = (size_t)waves * NT * 2 * 64 * sizeof(f32x4_t);
  auto kern = gemm_qkv_rope_m32_kernel<NT>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSD_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((p.N / 16) / NT), dim3(waves * 64), lds, st, p);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

template <int NT, int EPI, bool XNORM, int MAXC>
static int launch_fused_c(const FusedParams& p, int waves, hipStream_t st) {
  const int blocks = (p.N / 16) / NT;
  size_t lds = (size_t)waves * NT * 64 * sizeof(f32x4_t);
  FusedParams q = p;
  if (XNORM) {
    const size_t chunks = (size_t)p.M * (p.K / 8);
    const size_t need = chunks * 4 + (size_t)waves * 64;                      // chunk sums + one rs row per wave
    if (lds < need) lds = (need + 15) & ~(size_t)15;
    q.scratch_bytes = (int)lds;
    lds += chunks * 16;                                                       // x^ image
    if (lds > 160 * 1024) return SSD_ERR_SHAPE;
  } else {
    q.scratch_bytes = (int)lds;
  }
  auto kern = gemm_fused_kernel<NT, EPI, XNORM, MAXC>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSD_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, st, q);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

template <int NT, int EPI, bool XNORM>
static int launch_fused(const FusedParams& p, int waves, hipStream_t st) {
  if constexpr (XNORM) {
    if (p.M * (p.K / 8) <= waves * 64) return launch_fused_c<NT, EPI, true, 1>(p, waves, st);
    return launch_fused_c<NT, EPI, true, 0>(p, waves, st);
  } else {
    return launch_fused_c<NT, EPI, false, 1>(p, waves, st);
  }
}

template <int EPI, bool XNORM>
static int launch_fused_nt(const FusedParams& p, int nt, int waves, hipStream_t st) {
  if (nt == 1) {
    if constexpr (EPI == FEPI_SILU_FRAG) return SSD_ERR_ARG;
    else return launch_fused<1, EPI, XNORM>(p, waves, st);
  }
  if (nt == 2) return launch_fused<2, EPI, XNORM>(p, waves, st);
  if (nt == 4) return launch_fused<4, EPI, XNORM>(p, waves, st);
  return SSD_ERR_ARG;
}

static int fused_impl(const void* x_frag, const void* h_rows, const float* h_parts, int S, const void* res_in, void* res_out,
                      const void* norm_w, float eps, const void* w_frag, const void* bias, int M, int N, int K,
                      int epilogue, void* y, int ldy, const int64_t* positions, const float* cos_sin,
                      const int32_t* slots, void* q_out, void* k_cache, void* v_cache, int nh, int nkv, int hd,
                      int block_size, int nt, int waves, void* stream) {
  const bool m32 = M > 16 && M <= 32 && x_frag && !h_rows && !h_parts && epilogue == FEPI_QKV_ROPE;    // two token tiles
  if (M <= 0 || (M > 16 && !m32) || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  if ((h_rows != nullptr) + (x_frag != nullptr) + (h_parts != nullptr) != 1) return SSD_ERR_ARG;     // exactly one x source
  if ((h_rows || h_parts) && !norm_w) return SSD_ERR_ARG;
  if (h_parts && (S < 1 || S > 16)) return SSD_ERR_ARG;
  if (res_out && res_out == res_in) return SSD_ERR_ARG;                   // in-place residual would race
  const int groups = N / 16;
  if (nt <= 0 || waves <= 0) {
    int nt1, waves1, tpw1;
    ssd_pick_skinny_cfg(groups, K / 32, epilogue == FEPI_SILU_FRAG, &nt1, &waves1, &tpw1);
    if (nt <= 0) nt = nt1;
    if (waves <= 0) waves = waves1;
  }
  if (waves > 16 || groups % nt) return SSD_ERR_ARG;
  if (epilogue == FEPI_SILU_FRAG && (nt & 1)) return SSD_ERR_ARG;
  if (epilogue == FEPI_QKV_ROPE) {
    if (!positions || !cos_sin || !slots || !q_out || !k_cache || !v_cache) return SSD_ERR_ARG;
    if (hd != 64 && hd != 128 && hd != 256) return SSD_ERR_SHAPE;
    if (N != (nh + 2 * nkv) * hd) return SSD_ERR_SHAPE;
  }
  FusedParams p;
  p.Wf = (const u32x4_t*)w_frag; p.Xf = (const u32x4_t*)x_frag; p.h = (const bf16_t*)h_rows;
  p.h_parts = h_parts; p.S = S;
  p.res_in = (const bf16_t*)res_in; p.res_out = (bf16_t*)res_out; p.norm_w = (const bf16_t*)norm_w;
  p.bias = (const bf16_t*)bias; p.y = y; p.positions = positions; p.cos_sin = cos_sin; p.slots = slots;
  p.q_out = (bf16_t*)q_out; p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache;
  p.eps = eps; p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.nh = nh; p.nkv = nkv; p.hd = hd; p.bs = block_size;
  hipStream_t st = (hipStream_t)stream;
  if (m32) {
    if (nt == 4) nt = 2;                       // two m-tiles double the accumulators and the x operands
    return nt == 1 ? launch_qkv_m32<1>(p, waves, st) : launch_qkv_m32<2>(p, waves, st);
  }
  const bool xn = h_rows != nullptr || h_parts != nullptr;
#define FUSED_DISPATCH(E)                                                                          \
  return xn ? launch_fused_nt<E, true>(p, nt, waves, st) : launch_fused_nt<E, false>(p, nt, waves, st);
  switch (epilogue) {
    case FEPI_ROWS: FUSED_DISPATCH(FEPI_ROWS)
    case FEPI_SILU_FRAG: FUSED_DISPATCH(FEPI_SILU_FRAG)
    case FEPI_QKV_ROPE: FUSED_DISPATCH(FEPI_QKV_ROPE)
    default: return SSD_ERR_ARG;
  }
#undef FUSED_DISPATCH
}

extern "C" int ssd_gemm_fused(const void* x_frag, const void* h_rows, const void* res_in, void* res_out,
                              const void* norm_w, float eps, const void* w_frag, const void* bias, int M, int N, int K,
                              int epilogue, void* y, int ldy, const int64_t* positions, const float* cos_sin,
                              const int32_t* slots, void* q_out, void* k_cache, void* v_cache, int nh, int nkv, int hd,
                              int block_size, int nt, int waves, void* stream) {
  return fused_impl(x_frag, h_rows, nullptr, 0, res_in, res_out, norm_w, eps, w_frag, bias, M, N, K, epilogue, y, ldy, positions,
                    cos_sin, slots, q_out, k_cache, v_cache, nh, nkv, hd, block_size, nt, waves, stream);
}

// Same, with the producer GEMM's output given as `splits` fp32 partial slabs [splits][M][K] (ssd_gemm_parts) instead of
// bf16 rows: the prologue sums them in slab order, rounds to bf16 (the reference's F.linear store) and continues as above.
extern "C" int ssd_gemm_fused_parts(const void* h_parts, int splits, const void* res_in, void* res_out, const void* norm_w,
                                    float eps, const void* w_frag, const void* bias, int M, int N, int K, int epilogue, void* y,
                                    int ldy, const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out,
                                    void* k_cache, void* v_cache, int nh, int nkv, int hd, int block_size, int nt, int waves,
                                    void* stream) {
  return fused_impl(nullptr, nullptr, (const float*)h_parts, splits, res_in, res_out, norm_w, eps, w_frag, bias, M, N, K, epilogue,
                    y, ldy, positions, cos_sin, slots, q_out, k_cache, v_cache, nh, nkv, hd, block_size, nt, waves, stream);
}

KT_DEFINE_SETTER(gemm_fused)
