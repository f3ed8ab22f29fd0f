// Prefill GEMM  y[M,N] = x[M,K] . W[N,K]^T  for 32 < M <= 128 token rows (one prefill chunk).  Replaces the cuBLAS
// F.linear calls of the reference's eager prefill (ssd/layers/linear.py:65,98,196; model_runner.py:602).
//
// Still HBM-bound on MI355X (128 FLOP/B against a ridge of ~312), but unlike the skinny kernel (gemm.hip) the x
// operand is no longer negligible: a wave that owns NT row groups of W needs 128/(16 NT) bytes of x per byte of W.
// Measured with gemm_wf_kernel<8,2>: every CU pulls 5 bytes through L2 per weight byte and the GEMM runs at ~2.5 TB/s.
// Here:
//  * a workgroup = 4 waves that own 4*NT ADJACENT row groups (256 rows of W for NT = 4) over the same K range, so the
//    x tile of a k-step ([128 rows] x [32 k] = 8 KiB, fragment-major) is fetched from L2 ONCE per workgroup, staged in
//    LDS (double-buffered ring, one barrier per k-step) and read by all four waves as ready-made MFMA B operands:
//    0.5 bytes of x per byte of W;
//  * W streams HBM -> VGPRs as before (1 KiB fragment tiles, non-temporal, U k-steps in flight per wave);
//  * N alone gives too few workgroups (N = 8192 -> 32), so K is split across gridDim.y and every split writes its
//    fp32 partial tile to a workspace; gemm_pf_epilogue_kernel sums the splits in a fixed order (deterministic, no
//    atomics), adds the bias, rounds once to bf16 and applies the same epilogues as gemm.hip (rows | SiLU*mul ->
//    fragment-major).  The partials are ~6-25 % extra traffic and mostly live in the 256 MiB Infinity Cache.
#include "common.h"
#include <cstdlib>

enum { PF_EPI_ROWS = 0, PF_EPI_SILU_FRAG = 1, PF_EPI_PARTIALS = 2 };
// PF_EPI_PARTIALS: stop after the split-K GEMM -- the fp32 partials ws[z][m][n] (z < splits, slab stride M * N) are the
// output, summed by the CONSUMER (ssd_rmsnorm_parts: the add + RMSNorm that follows o_proj / down_proj sums the slabs in the
// same order, rounds once to bf16 like the epilogue kernel would, and goes on): one launch per GEMM less in a prefill.

constexpr int PF_WAVES_DEFAULT = 4;   // waves per workgroup; the 8-wave form (NT = 2, same 16 row groups per workgroup) halves each
                                      // wave's accumulators and doubles the weight tiles in flight per workgroup
constexpr int PF_U = 4;      // k-steps of W in flight per wave; every K split is a multiple of this many k-steps

// direct: 0 = write fp32 partials for the epilogue kernel; 1 / 2 = the K range is not split, finish in place
// (1: rows bf16 + bias, 2: SiLU(gate) * up -> fragment-major) and skip the workspace round trip.
// BPS = k-steps per workgroup barrier: the x ring holds two PHASES of BPS k-steps each (one being read, one being staged).
template <int MT, int NT, int UU, int DIRECT, int PF_WAVES, int BPS = 1, int BPRE = 0>
__global__ void __launch_bounds__(64 * PF_WAVES, PF_WAVES <= 4 ? 2 : 1)
gemm_pf_kernel(const u32x4_t* __restrict__ Wf, const u32x4_t* __restrict__ Xf, float* __restrict__ ws,
               int M, int N, int K, int kt_per_split, int mt_valid, const bf16_t* __restrict__ bias,
               void* __restrict__ Yv, int ldy) {
  __shared__ u32x4_t xs[2 * BPS][MT][64];                // ring of two phases of BPS k-steps, MT fragment tiles per k-step
  constexpr int KTS = 16 + DIRECT;                        // trace slot (profiling builds only)
  KTRACE(KTS, 0);
  constexpr int U = UU;                                   // W k-steps in flight per wave (nk is a multiple of it)
  static_assert(UU % (2 * BPS) == 0, "an even number of phases per unrolled pass");
  constexpr int FPW = (MT + PF_WAVES - 1) / PF_WAVES;     // x fragment tiles each wave fetches per k-step
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KT = K >> 5;
  const int g0 = (blockIdx.x * PF_WAVES + wave) * NT;     // first row group of this wave
  const int kz0 = blockIdx.y * kt_per_split;
  const int nk = kt_per_split;                            // multiple of U, >= U (host-checked)

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const u32x4_t* wp = Wf + (((size_t)g0 * KT + kz0) << 6) + lane;
  const size_t wstride = (size_t)KT << 6;
  // x fragment tiles this wave stages: mt = wave + PF_WAVES * f (clamped to a valid tile; extra rows are never stored)
  const u32x4_t* xp[FPW];
  int xmt[FPW];
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    xmt[f] = wave + PF_WAVES * f;
    const int src = min(xmt[f], mt_valid - 1);
    xp[f] = Xf + (((size_t)src * KT + kz0) << 6) + lane;
  }

  u32x4_t a[U][NT], xr[2][BPS][FPW];
  auto load_w = [&](u32x4_t (&d)[NT], int kt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_nontemporal_load(wp + nt * wstride + ((size_t)kt << 6));
  };
  auto load_x = [&](u32x4_t (&d)[FPW], int kt) {
#pragma unroll
    for (int f = 0; f < FPW; ++f) d[f] = xp[f][(size_t)kt << 6];
  };
  auto stage_x = [&](const u32x4_t (&d)[FPW], int slot) {
#pragma unroll
    for (int f = 0; f < FPW; ++f)
      if (xmt[f] < MT) xs[slot][xmt[f]][lane] = d[f];
  };
  auto compute = [&](const u32x4_t (&w)[NT], int slot) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const u32x4_t b = xs[slot][mt][lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mfma16(w[nt], b, acc[nt][mt]);
    }
  };

  // prologue: phase 0 of x -> LDS half 0; phases 1 and 2 in registers; W(0..U-1) in flight.
  // Every load in the loop is issued unconditionally (tail iterations re-read the last k-step instead of being
  // predicated off), so the number of loads in flight is static and the compiler can wait for exactly the one it
  // needs (s_waitcnt vmcnt(n > 0)) instead of draining the queue at every k-step.
  const int klast = nk - 1;
#pragma unroll
  for (int b = 0; b < BPS; ++b) load_x(xr[0][b], b);
#pragma unroll
  for (int u = 0; u < U; ++u) load_w(a[u], u);
#pragma unroll
  for (int b = 0; b < BPS; ++b) load_x(xr[1][b], BPS + b);          // nk >= U >= 2 * BPS
#pragma unroll
  for (int b = 0; b < BPS; ++b) stage_x(xr[0][b], b);
#pragma unroll
  for (int b = 0; b < BPS; ++b) load_x(xr[0][b], min(2 * BPS + b, klast));
  KTRACE(KTS, 1);
  __syncthreads();
  KTRACE(KTS, 2);
  for (int kt = 0; kt < nk; kt += U) {
#pragma unroll
    for (int u = 0; u < U; u += BPS) {
      const int k = kt + u;
      const int par = (u / BPS) & 1;          // ring half of this phase (compile-time after unrolling)
      // the registers of parity par^1 hold the NEXT phase (loaded two phases ago) -> the other ring half, whose last
      // readers finished before the previous barrier; then the freed registers fetch the phase three ahead
#pragma unroll
      for (int b = 0; b < BPS; ++b) stage_x(xr[par ^ 1][b], (par ^ 1) * BPS + b);
#pragma unroll
      for (int b = 0; b < BPS; ++b) load_x(xr[par ^ 1][b], min(k + 3 * BPS + b, klast));
      if constexpr (BPRE == 2 && BPS == 2) {
        // both k-steps' B operands are requested before the first MFMA: the second k-step's LDS latency hides behind the first's
        // 16 MFMAs (one exposed LDS latency per phase instead of one per k-step; + 32 VGPRs).  The plain form compiles to
        // ds_read x2 -> s_waitcnt lgkmcnt -> 2-4 MFMAs, eight times per k-step.  Same MFMA order per accumulator: bit-identical.
        // Measured on MI355X (profiles/r04_pf_probe_bpre.txt): 1-6 % on every 70B / 8B matrix at M = 100 / 128 (gate_up 155 -> 151 us,
        // qkv 47.5 -> 46.4, o 32.1 -> 30.6), TTFT 30.1 -> 29.1 ms; a one-k-step variant (BPRE = 1) was no better and is gone
        u32x4_t b0[MT], b1[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b0[mt] = xs[par * BPS][mt][lane];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b1[mt] = xs[par * BPS + 1][mt][lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mfma16(a[u][nt], b0[mt], acc[nt][mt]);
        load_w(a[u], min(k + U, klast));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mfma16(a[u + 1][nt], b1[mt], acc[nt][mt]);
        load_w(a[u + 1], min(k + 1 + U, klast));
      } else {
#pragma unroll
        for (int b = 0; b < BPS; ++b) {
          compute(a[u + b], par * BPS + b);
          load_w(a[u + b], min(k + b + U, klast));
        }
      }
      __syncthreads();
    }
  }

  KTRACE(KTS, 3);
  const int mcol = lane & 15, nrow = (lane >> 4) * 4;
  if constexpr (DIRECT == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (g0 + nt) * 16 + nrow;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + mcol;
        f32x4_t s = acc[nt][mt];
        if (bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r] += bf2f(bias[n + r]);
        }
        if (m < M) {
          const u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
          *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + n) = v;
        }
      }
    }
    KTRACE(KTS, 4);
    return;
  }
  if constexpr (DIRECT == 2) {
    const int KT2 = (N >> 1) >> 5;
    u32x2_t* out2 = reinterpret_cast<u32x2_t*>(Yv);
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      const int f = ((g0 >> 1) + pr) * 16 + nrow;             // feature index in [0, N/2)
      const int ng = (g0 + 2 * pr) * 16 + nrow, nu = ng + 16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + mcol;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gb = acc[2 * pr][mt][r], ub = acc[2 * pr + 1][mt][r];
          if (bias) { gb += bf2f(bias[ng + r]); ub += bf2f(bias[nu + r]); }
          gb = round_bf(gb); ub = round_bf(ub);
          o[r] = (gb / (1.0f + __expf(-gb))) * ub;
        }
        if (m < M) {
          const u32x2_t v = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
          out2[frag_chunk(m, f >> 3, KT2) * 2 + ((f >> 2) & 1)] = v;
        }
      }
    }
    KTRACE(KTS, 4);
    return;
  }
  // fp32 partial tile of this split: ws[z][m][n]
  float* out = ws + (size_t)blockIdx.y * M * N;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (g0 + nt) * 16 + nrow;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + mcol;
      if (m < M) *reinterpret_cast<f32x4_t*>(out + (size_t)m * N + n) = acc[nt][mt];
    }
  }
  KTRACE(KTS, 4);
}

// Sum the K-splits in order, + bias, one rounding to bf16, then the same epilogues as gemm.hip.
template <int EPI>
__global__ void __launch_bounds__(256)
gemm_pf_epilogue_kernel(const float* __restrict__ ws, const bf16_t* __restrict__ bias, void* __restrict__ Yv,
                        int M, int N, int splits, int ldy) {
  const size_t MN = (size_t)M * N;
  if (EPI == PF_EPI_ROWS) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= MN) return;
    const int m = (int)(i4 / N), n = (int)(i4 % N);
    f32x4_t s = *reinterpret_cast<const f32x4_t*>(ws + i4);
    for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4_t*>(ws + z * MN + i4);
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += bf2f(bias[n + r]);
    }
    const u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + n) = v;
  } else {
    // W row groups come in (gate, up) pairs; act = silu(bf16(g)) * bf16(u) in fp32, one rounding
    // (ssd/layers/activation.py:11-14 as compiled), written fragment-major as down_proj's input (K' = N/2)
    const int half = N >> 1;
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= (size_t)M * half) return;
    const int m = (int)(i4 / half), f = (int)(i4 % half);          // f .. f+3 inside one 16-feature group
    const int ng = ((f >> 4) * 2) * 16 + (f & 15), nu = ng + 16;
    f32x4_t g = *reinterpret_cast<const f32x4_t*>(ws + (size_t)m * N + ng);
    f32x4_t u = *reinterpret_cast<const f32x4_t*>(ws + (size_t)m * N + nu);
    for (int z = 1; z < splits; ++z) {
      g += *reinterpret_cast<const f32x4_t*>(ws + z * MN + (size_t)m * N + ng);
      u += *reinterpret_cast<const f32x4_t*>(ws + z * MN + (size_t)m * N + nu);
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float gb = g[r], ub = u[r];
      if (bias) { gb += bf2f(bias[ng + r]); ub += bf2f(bias[nu + r]); }
      gb = round_bf(gb); ub = round_bf(ub);
      o[r] = (gb / (1.0f + __expf(-gb))) * ub;
    }
    const u32x2_t v = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    reinterpret_cast<u32x2_t*>(Yv)[frag_chunk(m, f >> 3, half >> 5) * 2 + ((f >> 2) & 1)] = v;
  }
}

template <int MT, int NT, int DIRECT>
static int pf_launch_d(const void* x, const void* w, float* ws, int M, int N, int K, int splits, const void* bias, void* y,
                       int ldy, int waves, int uu, int bps, int bpre, hipStream_t st) {
  const int KT = K >> 5;
  dim3 grid(N / (16 * NT * waves), splits);
  const int nk = KT / splits;
  // W k-steps in flight per wave: the narrow tile (NT = 2) has the registers for 8; nk must be a multiple
  if (uu <= 0) uu = (NT <= 2 && nk % 8 == 0) ? 8 : PF_U;
  if (nk % uu != 0 || bps < 1) return SSD_ERR_ARG;
  bool launched = false;
#define PF_GO_B(UU, WV, B, PRE)                                                                                          \
  if (!launched && uu == UU && waves == WV && bps == B && bpre == PRE) {                                                 \
    launched = true;                                                                                                     \
    hipLaunchKernelGGL((gemm_pf_kernel<MT, NT, UU, DIRECT, WV, B, PRE>), grid, dim3(64 * WV), 0, st, (const u32x4_t*)w, \
                       (const u32x4_t*)x, ws, M, N, K, nk, (M + 15) / 16, (const bf16_t*)bias, y, ldy);                  \
  }
#define PF_GO(UU, WV, B) PF_GO_B(UU, WV, B, 0)
  PF_GO(4, 4, 1)
  if constexpr (NT == 1 && MT == 8 && DIRECT != 2) {     // 64-row (4 waves) / 128-row (8 waves) tiles: twice the workgroups per split
    PF_GO(8, 4, 1) PF_GO(8, 4, 2) PF_GO(8, 8, 1) PF_GO(8, 8, 2)
  }
  if constexpr (NT == 2) {
    PF_GO(8, 4, 1) PF_GO(4, 8, 1) PF_GO(8, 8, 1)
    if constexpr (MT == 8) {     // the full prefill chunk (65..128 rows): two k-steps per barrier, 3..7-wave workgroups (so that
      PF_GO(8, 4, 2) PF_GO(8, 8, 2)                                        // row groups x splits can land on a multiple of 256 CUs)
      PF_GO_B(8, 4, 2, 2) PF_GO_B(8, 8, 2, 2) PF_GO_B(8, 5, 2, 2)          // B operands of both k-steps of a phase read up front (BPRE = 2)
      PF_GO(8, 3, 1) PF_GO(8, 3, 2) PF_GO(8, 5, 1) PF_GO(8, 5, 2) PF_GO(8, 6, 2) PF_GO(8, 7, 1) PF_GO(8, 7, 2)
    }
  }
#undef PF_GO
#undef PF_GO_B
  if (!launched) return SSD_ERR_ARG;
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

template <int MT, int NT>
static int pf_launch(const void* x, const void* w, float* ws, int M, int N, int K, int splits, int direct, const void* bias,
                     void* y, int ldy, int waves, int uu, int bps, int bpre, hipStream_t st) {
  if (direct == 1) return pf_launch_d<MT, NT, 1>(x, w, ws, M, N, K, splits, bias, y, ldy, waves, uu, bps, bpre, st);
  if (direct == 2) return pf_launch_d<MT, NT, 2>(x, w, ws, M, N, K, splits, bias, y, ldy, waves, uu, bps, bpre, st);
  return pf_launch_d<MT, NT, 0>(x, w, ws, M, N, K, splits, bias, y, ldy, waves, uu, bps, bpre, st);
}

// Default decomposition, from profiles/r02_pf_probe.txt (MI355X, M = 128, us): the narrow tile (nt = 2) everywhere --
// 70B gate_up 57344x8192: nt4 176.8, nt2 185.1, nt2 with 8 waves 160.4 (unsplit); qkv 10240x8192: nt4 s4 54.2, nt2w8 s4 51.0;
// o 8192x8192: nt4 s8 44.7, nt2 s4 40.7; down 8192x28672: nt4 s8 100.0, nt2w8 s8 96.5; 8B gate_up: nt2 s1 58.4; 8B down: nt2 s8 37.1.
//  * K splits: none when N alone gives >= 160 workgroups; otherwise enough for >= 160 workgroups AND <= ~96 k-steps per split
//    (each split costs M*N*8 bytes of partial traffic, but a long serial K walk by few workgroups costs more);
//  * 8 waves per workgroup (same 16 row groups, half the accumulators per wave, twice the weight tiles in flight) whenever
//    that still leaves >= 160 workgroups.
static void pf_pick(int N, int K, int* nt_out, int* splits_out, int* waves_out = nullptr) {
  const int KT = K >> 5;
  int nt = 2, waves = 4, s = 1;
  if (N % (16 * 2 * 4) != 0) { *nt_out = 4; *splits_out = 1; if (waves_out) *waves_out = 4; return; }   // caller rejects the shape
  const int blocks4 = N / (16 * 2 * 4);
  if (blocks4 < 160) {
    while (s < 8 && (blocks4 * s < 160 || KT / s > 96)) s *= 2;
    while (s > 1 && KT % (s * PF_U) != 0) s /= 2;
  }
  if (N % (16 * 2 * 8) == 0 && (N / (16 * 2 * 8)) * s >= 160) waves = 8;
  *nt_out = nt; *splits_out = s;
  if (waves_out) *waves_out = waves;
}

// Refinements for the full chunk (65..128 rows), from profiles/r03_pf_probe.txt (all bit-identical to the plain form: the K order
// inside a split does not change): two k-steps per barrier whenever the split is a multiple of 8 k-steps (3-12 % everywhere), and
// 5-wave workgroups (160 rows) when that puts exactly a multiple of 256 workgroups on the 256 CUs -- 70B qkv: 10240 rows x 4 splits
// = 256 workgroups, 46.2 us against 49.9 us with 160 eight-wave workgroups.
static void pf_refine(int M, int N, int K, int splits, int* waves, int* bps) {
  *bps = 1;
  if (M <= 64) return;
  const int nk = (K >> 5) / splits;
  if (nk % 8 != 0) return;
  *bps = 2;
  if (N % (16 * 2 * 5) == 0 && ((N / (16 * 2 * 5)) * splits) % 256 == 0) *waves = 5;
}

extern "C" int ssd_gemm_pf_workspace_bytes(int M, int N, int K, int64_t* bytes) {
  if (!bytes || M <= 0 || N <= 0 || K <= 0 || N % (16 * 2 * PF_WAVES_DEFAULT) != 0) return SSD_ERR_ARG;
  int nt, splits;
  pf_pick(N, K, &nt, &splits);
  *bytes = (int64_t)splits * M * N * 4;
  return SSD_OK;
}

extern "C" int ssd_gemm_pf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K,
                               int ldy, int epilogue, void* workspace, int64_t workspace_bytes, int nt, int splits,
                               void* stream) {
  if (M <= 16 || M > 128 || N <= 0 || K <= 0 || (K % (32 * PF_U))) return SSD_ERR_SHAPE;
  if (epilogue != PF_EPI_ROWS && epilogue != PF_EPI_SILU_FRAG && epilogue != PF_EPI_PARTIALS) return SSD_ERR_ARG;
  if (epilogue == PF_EPI_PARTIALS && bias) return SSD_ERR_ARG;
  // nt may carry the launch shape above its low byte: waves per workgroup in bits 8..15 (0 = 4; 7 or 8 only with nt = 2),
  // W k-steps in flight per wave in bits 16..23 (0 = default), k-steps per barrier in bits 24..27 (0 = 1)
  int waves = (nt >> 8) & 0xff;
  const int uu = (nt >> 16) & 0xff;
  int bps = (nt >> 24) & 0xf;
  const int bpre = (nt >> 28) & 3;          // bits 28-29: 2 = B operands of a two-k-step phase read up front (compiled for nt = 2, two-k-step phases, 4 / 5 / 8 waves, M > 64)
  nt &= 0xff;
  if (waves == 0) waves = PF_WAVES_DEFAULT;
  if (bps == 0) bps = 1;
  if (nt != 1 && nt != 2 && nt != 4) return SSD_ERR_ARG;
  if (nt == 1 && (epilogue == PF_EPI_SILU_FRAG || M <= 64)) return SSD_ERR_ARG;
  if (waves != 4 && !(waves >= 3 && waves <= 8 && nt <= 2)) return SSD_ERR_ARG;
  if (N % (16 * nt * waves) != 0) return SSD_ERR_SHAPE;
  const int KT = K >> 5;
  if (splits <= 0) { int nt_d; pf_pick(N, K, &nt_d, &splits); if (nt_d != nt) splits = 1; }
  if (splits > 16) return SSD_ERR_ARG;
  if (KT % (splits * PF_U) != 0) return SSD_ERR_ARG;
  const int direct = (splits == 1 && epilogue != PF_EPI_PARTIALS) ? (epilogue == PF_EPI_ROWS ? 1 : 2) : 0;     // unsplit K: finish inside the GEMM kernel
  if (!direct && (!workspace || workspace_bytes < (int64_t)splits * M * N * 4)) return SSD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int mt = (M + 15) / 16;
  int rc;
#define PF_ARGS x_frag, w_frag, ws, M, N, K, splits, direct, bias, y, ldy, waves, uu, bps, bpre, st
  if (mt <= 4) rc = nt == 4 ? pf_launch<4, 4>(PF_ARGS) : pf_launch<4, 2>(PF_ARGS);
  else rc = nt == 4 ? pf_launch<8, 4>(PF_ARGS) : nt == 2 ? pf_launch<8, 2>(PF_ARGS) : pf_launch<8, 1>(PF_ARGS);
#undef PF_ARGS
  if (rc != SSD_OK || direct || epilogue == PF_EPI_PARTIALS) return rc;
  if (epilogue == PF_EPI_ROWS) {
    const size_t items = (size_t)M * N / 4;
    hipLaunchKernelGGL((gemm_pf_epilogue_kernel<PF_EPI_ROWS>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, ws,
                       (const bf16_t*)bias, y, M, N, splits, ldy);
  } else {
    const size_t items = (size_t)M * (N / 2) / 4;
    hipLaunchKernelGGL((gemm_pf_epilogue_kernel<PF_EPI_SILU_FRAG>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st,
                       ws, (const bf16_t*)bias, y, M, N, splits, ldy);
  }
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_gemm_pf(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K,
                           int ldy, int epilogue, void* workspace, int64_t workspace_bytes, int splits, void* stream) {
  if (N <= 0 || K <= 0 || N % (16 * 2 * PF_WAVES_DEFAULT) != 0) return SSD_ERR_SHAPE;
  int nt, s, waves, bps;
  pf_pick(N, K, &nt, &s, &waves);
  if (splits > 0) { s = splits; waves = 4; }
  if (s > 16 || (K >> 5) % s != 0) return SSD_ERR_ARG;
  pf_refine(M, N, K, s, &waves, &bps);
  // B operands of a two-k-step phase read from LDS up front (BPRE = 2) wherever that form exists (measured 1-6 % faster, bit-identical:
  // profiles/r04_pf_probe_bpre.txt; the A/B switch SSD_PF_BPRE is gone since round 6, ssd_gemm_pf_cfg's nt bits 28-29 still select)
  const int bpre = (bps == 2 && nt == 2 && M > 64 && (waves == 4 || waves == 5 || waves == 8) && ((K >> 5) / s) % 8 == 0) ? 2 : 0;
  return ssd_gemm_pf_cfg(x_frag, w_frag, bias, y, M, N, K, ldy, epilogue, workspace, workspace_bytes,
                         nt | (waves << 8) | (bps << 24) | (bpre << 28), s, stream);
}

KT_DEFINE_SETTER(gemm_pf)
