// Skinny GEMM with the K range split ACROSS workgroups, for matrices with too few 16-row groups to fill the chip
// (the 1B draft's o_proj / down_proj: N = 2048 -> 128 row groups on 256 CUs; a pure read of such a matrix by 128
// workgroups tops out at ~3 TB/s, profiles/micro/readpat.hip).  y[M,N] = x[M,K] . W[N,K]^T (+ bias), M <= 16, bf16 rows.
//
// grid = (row groups, S).  Workgroup (g, z) streams the z-th K range of row group g exactly like gemm_wf_kernel
// (waves deal k-tiles round-robin, fixed-order LDS combine), publishes its fp32 partial tile (1 KiB) with agent-scope
// stores and bumps the group's arrival counter; the LAST workgroup to arrive sums the S partials in z order -- so the
// result does not depend on the arrival order -- applies the epilogue and resets the counter (hipGraph-replay safe).
// No atomics on data, no second launch.
#include "common.h"

__device__ __forceinline__ void st_agent64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(1024)
gemm_sk_kernel(const u32x4_t* __restrict__ Wf, const u32x4_t* __restrict__ Xf, const bf16_t* __restrict__ bias,
               bf16_t* __restrict__ Y, int M, int N, int K, int ldy, unsigned long long* __restrict__ ws,
               unsigned int* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned int s_prev;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int KT = K >> 5;
  const int g = blockIdx.x, z = blockIdx.y, S = gridDim.y;
  const int kz0 = (int)(((long)KT * z) / S), kz1 = (int)(((long)KT * (z + 1)) / S);
  constexpr int U = 4;
  const u32x4_t* wp = Wf + ((size_t)g * KT << 6) + lane;
  const u32x4_t* xp = Xf + lane;
  const bool xrow = (lane & 15) < M;           // padding token rows are not loaded
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  u32x4_t a[U], b[U];
  const int ngroups = (kz1 - kz0) / U;
  for (int grp = wave; grp < ngroups; grp += nw) {
    const int kt = kz0 + grp * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = __builtin_nontemporal_load(wp + ((size_t)(kt + u) << 6));
      b[u] = u32x4_t{0u, 0u, 0u, 0u};
      if (xrow) b[u] = xp[(size_t)(kt + u) << 6];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = mfma16(a[u], b[u], acc);
  }
  if (wave == nw - 1)
    for (int kt = kz0 + ngroups * U; kt < kz1; ++kt) {
      u32x4_t bb = {0u, 0u, 0u, 0u};
      if (xrow) bb = xp[(size_t)kt << 6];
      acc = mfma16(__builtin_nontemporal_load(wp + ((size_t)kt << 6)), bb, acc);
    }
  // in-workgroup combine, wave order
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  for (int w = 0; w < nw; ++w) s += red[w * 64 + lane];
  if (S > 1) {
    unsigned long long* mine = ws + ((size_t)(g * S + z) * 64 + lane) * 2;
    st_agent64(mine, (unsigned long long)__float_as_uint(s[0]) | ((unsigned long long)__float_as_uint(s[1]) << 32));
    st_agent64(mine + 1, (unsigned long long)__float_as_uint(s[2]) | ((unsigned long long)__float_as_uint(s[3]) << 32));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) s_prev = __hip_atomic_fetch_add(counters + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const unsigned int prev = *(volatile unsigned int*)&s_prev;
    if (prev != (unsigned)(S - 1)) return;                  // not the last one: done
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < S; ++zz) {                         // fixed order, whoever arrived last
      const unsigned long long* pp = ws + ((size_t)(g * S + zz) * 64 + lane) * 2;
      const unsigned long long v0 = ld_agent64(pp), v1 = ld_agent64(pp + 1);
      s[0] += __uint_as_float((unsigned)v0); s[1] += __uint_as_float((unsigned)(v0 >> 32));
      s[2] += __uint_as_float((unsigned)v1); s[3] += __uint_as_float((unsigned)(v1 >> 32));
    }
    if (lane == 0) __hip_atomic_store(counters + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int m = lane & 15, n = g * 16 + (lane >> 4) * 4;
  if (bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] += bf2f(bias[n + r]);
  }
  if (m < M) {
    const u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
    *reinterpret_cast<u32x2_t*>(Y + (size_t)m * ldy + n) = v;
  }
}

// workspace: >= (N/16) * splits KiB of device memory; counters: >= N/16 uint32, zero-initialised once (the kernel leaves
// them at zero).  splits in 1..8; waves in 1..16.
extern "C" int ssd_gemm_splitk(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                               int splits, int waves, void* workspace, void* counters, void* stream) {
  if (M <= 0 || M > 16 || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  if (splits < 1 || splits > 8 || waves < 1 || waves > 16) return SSD_ERR_ARG;
  if (splits > 1 && (!workspace || !counters)) return SSD_ERR_ARG;
  if ((K >> 5) / splits < 1) return SSD_ERR_ARG;
  hipLaunchKernelGGL(gemm_sk_kernel, dim3(N / 16, splits), dim3(64 * waves), (size_t)waves * 64 * sizeof(f32x4_t),
                     (hipStream_t)stream, (const u32x4_t*)w_frag, (const u32x4_t*)x_frag, (const bf16_t*)bias, (bf16_t*)y, M, N,
                     K, ldy, (unsigned long long*)workspace, (unsigned int*)counters);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_sp_kernel: the latency-optimal form for the SMALL matrices of a decode layer (1B-class o_proj / down_proj: 8-34 MB,
// N = 2048 -> only 128 row groups).  Two measured facts shape it (profiles/r02_draft_probe.txt):
//   * one CU pulls ~25-28 GB/s from HBM whatever it runs, so 128 workgroups cap a launch at ~3 TB/s: the K range is split
//     over gridDim.y = S workgroups per row group (>= 256 workgroups in total);
//   * a launch this short is a latency chain, not a stream: every wave therefore puts ALL of its k-tiles in flight at
//     once (TPW 1-KiB weight loads per lane, one HBM round trip) instead of walking groups of 4.
// The S partial sums are NOT combined here (an in-launch cross-workgroup combine costs an agent-scope release + acquire,
// ~3.5 us, more than the kernel itself -- gemm_sk_kernel above measures that): they are written as fp32 slabs
// P[z][m][n], and the CONSUMER kernel (the next fused norm+GEMM's prologue, gemm_fused.hip, or ssd_rmsnorm) sums the S
// slabs in z order while it forms x = bf16(sum) + residual anyway.  The kernel boundary is the only synchronisation.
// With P == nullptr and S == 1 it writes bf16 rows (+ bias) like gemm_sk_kernel.
// ---------------------------------------------------------------------------------------------------------------------
template <int TPW, int MT>
__global__ void __launch_bounds__(1024)
gemm_sp_kernel(const u32x4_t* __restrict__ Wf, const u32x4_t* __restrict__ Xf, const bf16_t* __restrict__ bias,
               bf16_t* __restrict__ Y, float* __restrict__ P, int M, int N, int K, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KTS = K > N ? 5 : 4;        // trace slot (profiling builds only): down_proj / o_proj
  KTRACE(KTS, 0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int KT = K >> 5;
  const int g = blockIdx.x, z = blockIdx.y, S = gridDim.y;
  const int kz0 = (int)(((long)KT * z) / S), kz1 = (int)(((long)KT * (z + 1)) / S);
  const u32x4_t* wp = Wf + ((size_t)g * KT << 6) + lane;
  const u32x4_t* xp = Xf + lane;
  const size_t xstride = (size_t)KT << 6;      // chunks between the 16-row tiles of x
  u32x4_t a[TPW], b[TPW][MT];
  // tiles are dealt round-robin (wave w: kz0 + w, kz0 + w + nw, ...): the workgroup walks its K slice linearly
  // (every load is unconditional on a clamped tile index: a branch around a load makes hipcc drain vmcnt per element; a
  // slot past the slice end is zeroed after the fact)
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = min(kz0 + wave + i * nw, kz1 - 1);        // wave-uniform
    a[i] = __builtin_nontemporal_load(wp + ((size_t)kt << 6));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      b[i][mt] = u32x4_t{0u, 0u, 0u, 0u};
      if (mt * 16 + (lane & 15) < M) b[i][mt] = xp[mt * xstride + ((size_t)kt << 6)];    // padding token rows are not loaded
    }
  }
  KTRACE(KTS, 1);
  f32x4_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (kz0 + wave + i * nw >= kz1) a[i] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(a[i], b[i][mt], acc[mt]);
  }
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);        // [nw][MT][64]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
  KTRACE(KTS, 2);
  __syncthreads();
  KTRACE(KTS, 3);
  if (wave >= MT) return;                                  // wave mt finishes m-tile mt
  const int mt = wave;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  for (int w = 0; w < nw; ++w) s += red[(w * MT + mt) * 64 + lane];
  const int m = mt * 16 + (lane & 15), n = g * 16 + (lane >> 4) * 4;
  if (m >= M) return;
  if (P) {
    *reinterpret_cast<f32x4_t*>(P + ((size_t)z * M + m) * N + n) = s;
  } else {
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += bf2f(bias[n + r]);
    }
    const u32x2_t v = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
    *reinterpret_cast<u32x2_t*>(Y + (size_t)m * ldy + n) = v;
  }
  KTRACE(KTS, 4);
}

// parts: fp32 [splits][M][N] slabs (y must be null) or null (splits must be 1; bf16 rows into y, + bias).
// M <= 32; splits 1..16, waves 1..16 (>= 2 when M > 16), and ceil((K/32 / splits) / waves) <= 8 k-tiles per wave.
extern "C" int ssd_gemm_parts(const void* x_frag, const void* w_frag, const void* bias, void* y, void* parts, int M, int N,
                              int K, int ldy, int splits, int waves, void* stream) {
  if (M <= 0 || M > 32 || (N & 15) || (K & 31) || N <= 0 || K <= 0) return SSD_ERR_SHAPE;
  if (splits < 1 || splits > 16 || waves < 1 || waves > 16) return SSD_ERR_ARG;
  if ((parts == nullptr) == (y == nullptr)) return SSD_ERR_ARG;
  if (!parts && splits != 1) return SSD_ERR_ARG;
  const int KT = K >> 5;
  const int mt = (M + 15) / 16;
  if (KT < splits || waves < mt) return SSD_ERR_ARG;
  const int per_wg = (KT + splits - 1) / splits;
  const int tpw = (per_wg + waves - 1) / waves;
  if (tpw > 8) return SSD_ERR_ARG;
  const dim3 grid(N / 16, splits), block(64 * waves);
  const size_t lds = (size_t)waves * mt * 64 * sizeof(f32x4_t);
  hipStream_t st = (hipStream_t)stream;
#define SP_LAUNCH(T, MTV)                                                                                               \
  hipLaunchKernelGGL((gemm_sp_kernel<T, MTV>), grid, block, lds, st, (const u32x4_t*)w_frag, (const u32x4_t*)x_frag,     \
                     (const bf16_t*)bias, (bf16_t*)y, (float*)parts, M, N, K, ldy)
#define SP_TPW(MTV)                                                                                                     \
  if (tpw <= 1) SP_LAUNCH(1, MTV);                                                                                      \
  else if (tpw <= 2) SP_LAUNCH(2, MTV);                                                                                 \
  else if (tpw <= 4) SP_LAUNCH(4, MTV);                                                                                 \
  else SP_LAUNCH(8, MTV)
  if (mt == 1) { SP_TPW(1); } else { SP_TPW(2); }
#undef SP_TPW
#undef SP_LAUNCH
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

KT_DEFINE_SETTER(gemm_sk)
