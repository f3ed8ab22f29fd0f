// Skinny bf16 GEMM  y[M,N] = x[M,K] . W[N,K]^T  for the draft/verify/tree forwards (M = B*(K+1) <= ~32,
// and chunked for prefill).  Replaces cuBLAS F.linear at reference ssd/layers/linear.py:65,98,196 and
// ssd/layers/embed_head.py:88,95,111.
//
// MI355X design (HBM-bound weight streaming):
//  * W is stored "fragment-major" (common.h): each 16x32 tile is 1 KiB in MFMA lane order, tiles of
//    one 16-row group contiguous along K.  One wave load = 1 KiB contiguous = one MFMA A operand.
//    No LDS round trip for weights (guide: "GEMV / M <= 16: load straight to VGPRs").
//  * x is stored in the same fragment-major layout by its producer kernel (rmsnorm / attention /
//    the SiLU epilogue below), so the B operand is also a coalesced 1 KiB wave load (L2 resident).
//  * One workgroup owns NT adjacent 16-row groups of W for the WHOLE K; its waves split K and
//    combine through LDS in a fixed order (deterministic, no atomics, no inter-workgroup traffic).
//  * Weight loads are non-temporal (each byte is read exactly once per forward).
#include "common.h"

enum { EPI_ROWS = 0, EPI_SILU_FRAG = 1, EPI_ROWS_F32 = 2, EPI_ROWS_ARGMAX = 3 };

// EPI_ROWS_ARGMAX (the LM head on the greedy path): bf16 rows as EPI_ROWS, plus every workgroup's own (max value, lowest
// index) of each token row over the features it produced -- compared on the bf16-ROUNDED values, i.e. exactly what an
// argmax over the stored logits sees -- written to part_val / part_idx [m * part_stride + blockIdx.x].  ssd_argmax_parts*
// (sample.hip) finishes the argmax from these few thousand candidates instead of re-reading M x V logits from one
// workgroup per row (13 us for one 128K-token row: a single CU's load bandwidth).
struct ArgPart { float v; int i; };
__device__ __forceinline__ ArgPart arg_better(ArgPart a, ArgPart b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

extern "C" int ssd_gemm_splitk(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                               int splits, int waves, void* workspace, void* counters, void* stream);

template <int MT, int NT>
struct Stage {
  u32x4_t a[NT];
  u32x4_t b[MT];
};

// DEEP (round 6): twice the k-tiles per stage -- twice the bytes in flight per wave -- for launches of <= 8 waves per workgroup, which
// may use 256 VGPRs (the 128-VGPR ceiling of the 16-wave form is what set the depth of the plain kernel; a weight-streaming wave's
// throughput is its bytes in flight over the memory latency).  Same rounding points; the k-tiles are dealt to the waves in groups
// twice as large, so the fp32 summation order differs from the plain form (tolerance-equal).  Measured at the 70B verify's shapes,
// same box, same run (profiles/r06_xsum_probe_v3_hwcvt_and_deep.txt): gate_up 149.3 -> 147.5 us (145.7 as 256 workgroups of 2 row
// groups x 7 tiles), o_proj 21.9 -> 21.5, down_proj 74.1 -> 72.9.
template <int MT, int NT, int EPI, bool DEEP = false>
__global__ void __launch_bounds__(DEEP ? 512 : 1024)
gemm_wf_kernel(const u32x4_t* __restrict__ Wf, const u32x4_t* __restrict__ Xf,
               const bf16_t* __restrict__ bias, void* __restrict__ Yv, int M, int N, int K, int ldy, int tpw,
               float* __restrict__ part_val, int* __restrict__ part_idx, int part_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KTS = EPI == EPI_SILU_FRAG ? 8 : 9;        // trace slot (profiling builds only)
  KTRACE(KTS, 0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int KT = K >> 5;
  // A workgroup owns `tpw` CONSECUTIVE tiles of NT row groups = one contiguous slab of W (long sequential DRAM runs, no
  // tail of half-empty workgroup rounds), and the first loads of the next tile are issued before the cross-wave
  // combine + epilogue of the current one, so the memory pipe never drains inside a workgroup.
  const int ntiles = (N / 16) / NT;
  const int t_begin = blockIdx.x * tpw, t_end = min(ntiles, t_begin + tpw);
  const u32x4_t* xp = Xf + lane;
  const size_t wstride = (size_t)KT << 6;  // chunks between adjacent row groups
  const size_t xstride = (size_t)KT << 6;
  constexpr int U = ((MT + NT <= 3) ? 4 : ((MT + NT <= 6) ? 2 : 1)) * (DEEP ? 2 : 1);
  Stage<MT, NT> cur[U], nxt[U];
  // K is dealt to the waves in groups of U k-tiles, round-robin: wave w takes groups w, w + nw, ...  The workgroup as a
  // whole therefore walks each row group's K run linearly (DRAM-friendly: measured 6.4-6.9 TB/s for this pattern against
  // 4.7-6.1 when every wave streams its own distant K slice, profiles/micro/readpat.hip).  The < U left-over k-tiles go
  // to the last wave.
  const int kstep = nw * U;
  const int kt0 = wave * U;
  const int kmain = (KT / U) * U;                         // first k-tile of the left-over
  const u32x4_t* wp = Wf + ((size_t)t_begin * NT * KT << 6) + lane;
  const int mt_last = (M - 1) >> 4;
  auto load = [&](Stage<MT, NT>(&s)[U], int kt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        s[u].a[nt] = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)(kt + u) << 6));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // token rows >= M of the last 16-row tile are padding: their lanes do not load (x traffic is per 16-byte lane;
        // at M = 7 this more than halves the L2 -> CU bytes of the B operand, at M = 1 it is 1/16)
        // (only where x is not amortised over several weight tiles: the predication costs the wider kernels more than
        // the saved bytes are worth)
        // m-tiles past the last one that holds a token (MT is rounded up to 1/2/4/8) re-read that last tile: x_frag only
        // has ceil(M/16) row groups, and their products are never stored
        u32x4_t b = {0u, 0u, 0u, 0u};
        if (NT > 1 || mt * 16 + (lane & 15) < M) b = xp[(mt < mt_last ? mt : mt_last) * xstride + ((size_t)(kt + u) << 6)];
        s[u].b[mt] = b;
      }
    }
  };
  if (t_begin < t_end && kt0 < kmain) load(cur, kt0);
  ArgPart run[MT];          // EPI_ROWS_ARGMAX: this wave's best candidate per token row so far (identical in the 4 lanes of a row)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) run[mt] = ArgPart{-INFINITY, 0x7fffffff};

  for (int tile = t_begin; tile < t_end; ++tile) {
  const int tile0 = tile * NT;
  f32x4_t acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](Stage<MT, NT>(&s)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(s[u].a[nt], s[u].b[mt], acc[nt][mt]);
  };

  int kt = kt0;
  if (kt < kmain) {
    for (; kt + kstep < kmain; kt += kstep) {
      load(nxt, kt + kstep);
      compute(cur);
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    compute(cur);
  }
  for (kt = (wave == nw - 1) ? kmain : KT; kt < KT; ++kt) {  // K remainder (< U tiles): last wave
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      u32x4_t a = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)kt << 6));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4_t b = {0u, 0u, 0u, 0u};
        if (mt * 16 + (lane & 15) < M) b = xp[mt * xstride + ((size_t)kt << 6)];
        acc[nt][mt] = mfma16(a, b, acc[nt][mt]);
      }
    }
  }

  // next tile: advance the weight pointer and put its first loads in flight before the combine
  wp += (size_t)NT * wstride;
  if (tile + 1 < t_end && kt0 < kmain) load(cur, kt0);

  // ---- cross-wave split-K combine through LDS, fixed order ----
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);  // [nw][NT*MT][64]
  constexpr int ITEMS = NT * MT;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[((wave * ITEMS) + nt * MT + mt) * 64 + lane] = acc[nt][mt];
  KTRACE(KTS, 4);      // (of the workgroup's LAST tile)
  __syncthreads();
  KTRACE(KTS, 5);

  const int mcol = lane & 15;   // D column j  -> token row m
  const int nrow = (lane >> 4) * 4;  // D rows i = nrow + r -> output feature n
  if (EPI == EPI_SILU_FRAG) {
    // W row groups come in (gate, up) pairs (interleaved at weight-shuffle time).  act = silu(g)*u with
    // both rounded to bf16 first (the reference's F.linear stores bf16, ssd/layers/activation.py:11-14
    // then computes in fp32 with one final rounding).  Output goes straight to the fragment-major
    // input buffer of down_proj (K' = N/2).
    constexpr int PAIRS = NT / 2;
    const int KT2 = (N >> 1) >> 5;
    u32x2_t* out = reinterpret_cast<u32x2_t*>(Yv);
    for (int item = wave; item < PAIRS * MT; item += nw) {
      const int pr = item / MT, mt = item % MT;
      f32x4_t g = f32x4_t{0.f, 0.f, 0.f, 0.f}, u = g;
      for (int w = 0; w < nw; ++w) {
        g += red[((w * ITEMS) + (2 * pr) * MT + mt) * 64 + lane];
        u += red[((w * ITEMS) + (2 * pr + 1) * MT + mt) * 64 + lane];
      }
      const int m = mt * 16 + mcol;
      const int n = ((tile0 >> 1) + pr) * 16 + nrow;  // feature index in [0, N/2)
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float gb = g[r], ub = u[r];
        if (bias) { gb += bf2f(bias[(tile0 + 2 * pr) * 16 + nrow + r]); ub += bf2f(bias[(tile0 + 2 * pr + 1) * 16 + nrow + r]); }
        gb = round_bf(gb); ub = round_bf(ub);
        o[r] = (gb / (1.0f + __expf(-gb))) * ub;
      }
      if (m < M) {
        u32x2_t v = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
        out[frag_chunk(m, n >> 3, KT2) * 2 + ((n >> 2) & 1)] = v;
      }
    }
  } else {
    for (int item = wave; item < ITEMS; item += nw) {
      const int nt = item / MT, mt = item % MT;
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < nw; ++w) s += red[((w * ITEMS) + item) * 64 + lane];
      const int m = mt * 16 + mcol;
      const int n = (tile0 + nt) * 16 + nrow;
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += bf2f(bias[n + r]);
      }
      if (m < M) {
        if (EPI == EPI_ROWS_F32) {
          *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(Yv) + (size_t)m * ldy + n) = s;
        } else {
          u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
          *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + n) = v;
        }
      }
      if (EPI == EPI_ROWS_ARGMAX) {
        ArgPart c = {-INFINITY, 0x7fffffff};
#pragma unroll
        for (int r = 0; r < 4; ++r) {              // ascending n: a strict > keeps the lowest index of equal values
          const float v = round_bf(s[r]);
          if (v > c.v) c = ArgPart{v, n + r};
        }
        c = arg_better(c, ArgPart{__shfl_xor(c.v, 16, 64), __shfl_xor(c.i, 16, 64)});
        c = arg_better(c, ArgPart{__shfl_xor(c.v, 32, 64), __shfl_xor(c.i, 32, 64)});
#pragma unroll
        for (int q = 0; q < MT; ++q)
          if (q == mt) run[q] = arg_better(run[q], c);
      }
    }
  }
  if (EPI == EPI_ROWS_ARGMAX && tile + 1 == t_end) {
    // the last tile's candidates are in: per-wave bests -> their own LDS area BEHIND the combine area, so that the barrier
    // below (which every tile needs anyway) is the only one between them and the merge -- a second barrier per workgroup
    // costs these ~5 us-lived workgroups a measurable share of their life
    ArgPart* lbw = reinterpret_cast<ArgPart*>(smem + (size_t)nw * NT * MT * 64 * sizeof(f32x4_t));      // [nw][MT*16]
    if (lane < 16) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) lbw[(wave * MT + mt) * 16 + lane] = run[mt];
    }
  }
  __syncthreads();   // the combine area is reused by the next tile
  }
  if (EPI == EPI_ROWS_ARGMAX) {
    // one (value, index) per token row and workgroup
    const ArgPart* lb = reinterpret_cast<const ArgPart*>(smem + (size_t)nw * NT * MT * 64 * sizeof(f32x4_t));
    for (int t = threadIdx.x; t < MT * 16; t += blockDim.x) {
      if (t >= M) break;
      ArgPart b = lb[t];
      for (int w = 1; w < nw; ++w) b = arg_better(b, lb[w * MT * 16 + t]);
      part_val[(size_t)t * part_stride + blockIdx.x] = b.v;
      part_idx[(size_t)t * part_stride + blockIdx.x] = b.i;
    }
  }
  KTRACE(KTS, 6);
}

// ---------------------------------------------------------------------------------------------
// Launch heuristics.  waves/block * blocks should put >= ~8-16 waves on each of the 256 CUs while
// each wave still streams a few KiB contiguously.
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int EPI, bool DEEP = false>
static int launch_t(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int ldy,
                    int waves, int tpw, hipStream_t st, float* part_val = nullptr, int* part_idx = nullptr, int part_stride = 0) {
  const int ntiles = (N / 16) / NT;
  if (tpw < 1) tpw = 1;
  const int blocks = (ntiles + tpw - 1) / tpw;
  if (EPI == EPI_ROWS_ARGMAX && (!part_val || !part_idx || part_stride < blocks)) return SSD_ERR_ARG;
  size_t lds = (size_t)waves * NT * MT * 64 * sizeof(f32x4_t);
  if (EPI == EPI_ROWS_ARGMAX) lds += (size_t)waves * MT * 16 * sizeof(ArgPart);      // per-wave argmax candidates behind the combine area
  if (DEEP && waves > 8) return SSD_ERR_ARG;
  auto kern = gemm_wf_kernel<MT, NT, EPI, DEEP>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SSD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, st, (const u32x4_t*)w, (const u32x4_t*)x,
                     (const bf16_t*)bias, y, M, N, K, ldy, tpw, part_val, part_idx, part_stride);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

template <int MT, int EPI>
static int launch_nt(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int ldy,
                     int nt, int waves, int tpw, hipStream_t st) {
  if (nt == 1) {
    if constexpr (EPI == EPI_SILU_FRAG) return SSD_ERR_ARG;
    else return launch_t<MT, 1, EPI>(x, w, bias, y, M, N, K, ldy, waves, tpw, st);
  }
  if (nt == 2) return launch_t<MT, 2, EPI>(x, w, bias, y, M, N, K, ldy, waves, tpw, st);
  if (nt == 4) {
    if constexpr (MT > 2) return SSD_ERR_ARG;
    else return launch_t<MT, 4, EPI>(x, w, bias, y, M, N, K, ldy, waves, tpw, st);
  }
  return SSD_ERR_ARG;
}

// `waves` may carry the tiles-per-workgroup count in bits 8..15 (0 = 1): a workgroup then streams that many consecutive
// tiles of nt row groups.
extern "C" int ssd_gemm_wf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N,
                               int K, int ldy, int epilogue, int nt, int waves, void* stream) {
  if (M <= 0 || M > 128 || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  const int tpw = (waves >> 8) & 0xff;
  waves &= 0xff;
  const bool deep = (nt >> 8) & 1;          // bit 8 of nt: the DEEP form (M <= 16, <= 8 waves, nt 2 / 4, rows or SiLU epilogue)
  nt &= 0xff;
  if (deep) {
    if (M > 16 || waves > 8 || (nt != 2 && nt != 4) || ((N / 16) % nt) != 0) return SSD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_ROWS) {
      if (nt == 2) return launch_t<1, 2, EPI_ROWS, true>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st);
      return launch_t<1, 4, EPI_ROWS, true>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st);
    }
    if (epilogue == EPI_SILU_FRAG) {
      if (nt == 2) return launch_t<1, 2, EPI_SILU_FRAG, true>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st);
      return launch_t<1, 4, EPI_SILU_FRAG, true>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st);
    }
    return SSD_ERR_ARG;
  }
  if (waves < 1 || waves > 16) return SSD_ERR_ARG;
  if (((N / 16) % nt) != 0) return SSD_ERR_ARG;
  if (epilogue == EPI_SILU_FRAG && (nt & 1)) return SSD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int mt = (M + 15) / 16;
#define DISPATCH_MT(MTV)                                                                                         \
  switch (epilogue) {                                                                                            \
    case EPI_ROWS: return launch_nt<MTV, EPI_ROWS>(x_frag, w_frag, bias, y, M, N, K, ldy, nt, waves, tpw, st);         \
    case EPI_SILU_FRAG: return launch_nt<MTV, EPI_SILU_FRAG>(x_frag, w_frag, bias, y, M, N, K, ldy, nt, waves, tpw, st); \
    case EPI_ROWS_F32: return launch_nt<MTV, EPI_ROWS_F32>(x_frag, w_frag, bias, y, M, N, K, ldy, nt, waves, tpw, st); \
    default: return SSD_ERR_ARG;                                                                                 \
  }
  if (mt == 1) { DISPATCH_MT(1) }
  if (mt == 2) { DISPATCH_MT(2) }
  if (mt <= 4) { DISPATCH_MT(4) }
  { DISPATCH_MT(8) }
#undef DISPATCH_MT
}

// A/B facility for measurements (ssd_hip_tune.h; bench.py --tune-deep): how ssd_gemm_wf's default dispatch treats the 70B-class
// matrices.  1 (default) = DEEP form at the tuned decompositions; 0 = the plain kernels of rounds 1-5.  Process-wide; set before any
// graph is captured.  (A third mode -- gate_up as exactly 256 workgroups of 2 row groups x 7 tiles, the fastest gate_up KERNEL in
// isolation, 145.7 vs 147.5 us -- made the c4 STEP 1.1 ms slower on the same box, profiles/r06_c4_deep_ab_same_box.txt: with the tuned
// 224 workgroups 32 CUs stay free during half of the verify, and the co-located draft's kernels run there.  Deleted.)
static int g_deep_mode = 1;
extern "C" int ssd_tune_deep(int mode) {
  if (mode < 0 || mode > 1) return SSD_ERR_ARG;
  g_deep_mode = mode;
  return SSD_OK;
}

// Default configuration: pick (row groups per workgroup, waves per workgroup) from the shape.
extern "C" int ssd_gemm_wf(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N,
                           int K, int ldy, int epilogue, void* stream) {
  if ((N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  const int groups = N / 16, KT = K / 32, mt = (M + 15) / 16;
  if (mt == 1) {       // decode / verify rows: the tuned table
    int nt1, waves1, tpw1;
    ssd_pick_skinny_cfg(groups, KT, epilogue == EPI_SILU_FRAG, &nt1, &waves1, &tpw1);
    // the 70B-class matrices (very long K, >= 512 row groups, 8 waves per workgroup): the DEEP form at the same decomposition, measured
    // 1-2 % faster on each kernel and -0.7 ms on the c4 step (round 6, profiles/r06_deep_probe.txt, r06_c4_deep_ab_same_box.txt)
    // (only the classes that were measured: N = 8192 rows matrices with K >= 8192 -- o_proj, down_proj -- and the gate_up whose row
    //  groups are a multiple of 512; tensor-parallel shards and the other models keep the plain form)
    const bool deep_rows = epilogue == EPI_ROWS && groups == 512 && nt1 == 2;
    const bool deep_silu = epilogue == EPI_SILU_FRAG && nt1 == 4 && groups % 512 == 0 && groups / 512 <= 8;
    if (g_deep_mode && waves1 == 8 && KT >= 256 && (deep_rows || deep_silu))
      return ssd_gemm_wf_cfg(x_frag, w_frag, bias, y, M, N, K, ldy, epilogue, nt1 | 256, waves1 | (tpw1 << 8), stream);
    // one row group per workgroup, bf16 rows: the single-buffered kernel of gemm_sk.hip (fewer registers -> more resident
    // workgroups) measured 5-17 % faster than the register double buffer below (profiles/micro/splitk_probe.py)
    if (nt1 == 1 && tpw1 == 1 && epilogue == EPI_ROWS)
      return ssd_gemm_splitk(x_frag, w_frag, bias, y, M, N, K, ldy, 1, waves1, nullptr, nullptr, stream);
    return ssd_gemm_wf_cfg(x_frag, w_frag, bias, y, M, N, K, ldy, epilogue, nt1, waves1 | (tpw1 << 8), stream);
  }
  int nt = 1;
  if (epilogue == EPI_SILU_FRAG) nt = 2;
  else if (mt >= 4 && groups % 2 == 0) nt = 2;          // amortise the larger x operand
  else if (groups >= 2048 && groups % 2 == 0) nt = 2;   // plenty of workgroups anyway
  // waves: each wave should stream >= 4 k-tiles per row group; 16 waves/CU wanted when blocks ~ CUs.
  int waves = 16;
  while (waves > 1 && KT / waves < 4) waves >>= 1;
  const int blocks = groups / nt;
  if (blocks >= 1024 && waves > 8) waves = 8;
  // two token tiles (the 24-branch tree step) on a mid-sized matrix: 4 waves x 16 k-tiles beat 16 x 4 (1B gate_up at M = 24:
  // 16.2 -> 13.4 us, profiles/r02_tune_1b.txt) -- more workgroups resident per CU, fewer partials to combine
  if (mt == 2 && blocks >= 512 && blocks < 2048 && waves > 4 && KT / 4 >= 4) waves = 4;
  if (mt >= 4 && waves > 8) waves = 8;  // LDS for the combine: waves*nt*mt KiB
  if (mt >= 8 && waves > 4) waves = 4;
  return ssd_gemm_wf_cfg(x_frag, w_frag, bias, y, M, N, K, ldy, epilogue, nt, waves, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// LM head on the greedy path: logits rows + per-workgroup argmax candidates in one launch (EPI_ROWS_ARGMAX above).
// The decomposition is the default one of ssd_gemm_wf for (M, N, K); ssd_gemm_wf_argmax_parts returns how many candidates
// per token row it writes (= workgroups), so the caller can size part_val / part_idx [M][part_stride >= that].
// M <= 32 (decode / verify / tree-step rows); replaces F.linear + logits.argmax(-1) (reference ssd/layers/embed_head.py:
// 88-116 + ssd/layers/sampler.py:15-20 / ssd/utils/verify.py:34).
// ---------------------------------------------------------------------------------------------------------------------
static void argmax_head_cfg(int M, int N, int K, int* nt, int* waves, int* tpw) {
  const int groups = N / 16, KT = K / 32;
  if (M <= 16) {
    ssd_pick_skinny_cfg(groups, KT, false, nt, waves, tpw);
    return;
  }
  // two token tiles: the generic choice of ssd_gemm_wf
  int n = (groups >= 2048 && groups % 2 == 0) ? 2 : 1;
  int w = 16;
  while (w > 1 && KT / w < 4) w >>= 1;
  const int blocks = groups / n;
  if (blocks >= 1024 && w > 8) w = 8;
  if (blocks >= 512 && blocks < 2048 && w > 4 && KT / 4 >= 4) w = 4;
  *nt = n; *waves = w; *tpw = 1;
}

extern "C" int ssd_gemm_wf_argmax_parts(int M, int N, int K) {
  if (M <= 0 || M > 32 || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  int nt, waves, tpw;
  argmax_head_cfg(M, N, K, &nt, &waves, &tpw);
  const int ntiles = (N / 16) / nt;
  return (ntiles + tpw - 1) / tpw;
}

extern "C" int ssd_gemm_wf_argmax(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K,
                                  int ldy, float* part_val, int32_t* part_idx, int part_stride, void* stream) {
  if (M <= 0 || M > 32 || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  if (!part_val || !part_idx) return SSD_ERR_ARG;
  int nt, waves, tpw;
  argmax_head_cfg(M, N, K, &nt, &waves, &tpw);
  if (((N / 16) % nt) != 0) return SSD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 16) {
    if (nt == 1) return launch_t<1, 1, EPI_ROWS_ARGMAX>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st, part_val, part_idx, part_stride);
    if (nt == 2) return launch_t<1, 2, EPI_ROWS_ARGMAX>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st, part_val, part_idx, part_stride);
    return launch_t<1, 4, EPI_ROWS_ARGMAX>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st, part_val, part_idx, part_stride);
  }
  if (nt == 1) return launch_t<2, 1, EPI_ROWS_ARGMAX>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st, part_val, part_idx, part_stride);
  return launch_t<2, 2, EPI_ROWS_ARGMAX>(x_frag, w_frag, bias, y, M, N, K, ldy, waves, tpw, st, part_val, part_idx, part_stride);
}

KT_DEFINE_SETTER(gemm)
