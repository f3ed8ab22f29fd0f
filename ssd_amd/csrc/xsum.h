// "xsum" (round 6): the residual add + RMSNorm between two skinny GEMMs carried BY the GEMMs -- no norm launch, no
// cross-workgroup protocol; the only synchronisation is the kernel boundary that separates the two GEMMs anyway.
//
// Reference: RMSDNorm.forward (ssd/layers/layernorm.py:64-88, as compiled: x32 = fp32(x) + fp32(res); res_out = bf16(x32);
// y = bf16((x32 * rsqrt(mean(x32^2) + eps)) * w)) between a row-parallel linear (o_proj / down_proj, ssd/layers/linear.py:186-199)
// and the next column-parallel one (gate_up / next layer's qkv, linear.py:97-98) -- LlamaDecoderLayer.forward, ssd/models/llama3.py:185-199.
//
// Why (profiles/r06_c4_kernel_stats_start_of_round.txt): at the 70B verify's M = 8 rows the stand-alone norm is 160 launches of
// 4.9 us per verify (0.78 ms of 22.7) that move 0.4 MB each -- pure launch latency.  A norm INSIDE the consumer launch needs x^ of
// all h columns, i.e. every producer workgroup's output: round 4 built that as norm workgroups + flags + spinning GEMM workgroups
// (-3.4 us, a protocol in every layer; profiles/r04_nig_prototype.patch).  This form splits the norm at the one point where no
// cross-workgroup data is needed:
//   producer epilogue (gemm.hip EPI_ROWS_RES): the workgroup that finishes output columns [n, n + 16) of row m forms
//       x32 = fp32(bf16(acc)) + fp32(res_in[m][n..])          (the reference's bf16 store of F.linear, then the fp32 add)
//     writes res_out = bf16(x32), x32 itself (fp32, fragment-major: the consumer's B-operand layout at 32 B per lane), and ONE float
//     per (16-column group, row): the group's sum of squares;
//   consumer prologue (gemm.hip / gemm_fused.hip XS): while its first weight tiles are in flight every workgroup sums the h / 16
//     group sums of its M rows in a fixed order (each wave a contiguous block of groups, all loads issued before the first add, wave
//     totals through LDS in wave order) -> rs[m]; the norm weights go to LDS (2 bytes per column); the MFMA B operand is formed on
//     the fly per k-tile: x^ = bf16((x32 * rs) * w) -- the norm kernel's formula and rounding point.
// Same rounding points as ssd_rmsnorm; the fp32 ORDER of the sum of squares differs from rmsnorm_kernel's (chunk -> thread map), so
// rs can differ in its last ulp and an x^ element in one bf16 ulp, rarely: tolerance-tested against the oracle and against the
// separate launches (tests/test_hip_xsum.py), not bit-identical.
#pragma once
#include "common.h"

struct XsumIn {
  const float* x32f;      // fp32 fragment-major [16][K] (rows >= M never read)
  const float* ssp;       // [K / 16][16]: sum of squares of (16-column group g, row m) at g * 16 + m
  const bf16_t* norm_w;   // [K]
  float eps;
};

struct XsumOut {
  const bf16_t* res_in;   // [M][N] rows
  bf16_t* res_out;        // [M][N] rows (may alias res_in: every element is read and written by the same lane)
  float* x32f;            // fp32 fragment-major [16][N]
  float* ssp;             // [N / 16][16]
};

// LDS the consumer prologue needs behind the split-K combine area: the norm weights + one partial per (wave, row)
__host__ __device__ inline size_t xsum_lds_bytes(int K, int waves) { return (size_t)K * 2 + (size_t)waves * 16 * 4; }

// Consumer prologue in two halves (first measurement, profiles/r06_xsum_probe_v1.txt: with the small loads issued BEHIND the first
// weight tiles and a __syncthreads -- whose fence is `s_waitcnt vmcnt(0)` -- the prologue cost as much as the norm launch it replaces:
// a CU returns its loads in order, so the 2 KB of group sums came back after the first 128 KB of weights, and only then was the second
// stage of weights requested).  xsum_issue goes FIRST in the kernel -- the group sums and norm weights this thread needs, 24 small loads
// into registers -- then the caller puts its weight stages in flight, then xsum_finish consumes the small loads (the compiler's vmcnt
// counts only the younger weight loads: it does not wait for them), parks the norm weights in LDS, meets at an LDS-only barrier and
// returns rs of row (lane & 15).  Shape limits (host-checked, xsum_shape_ok): K / 16 <= 64 * waves, K / 8 <= 2 * threads.
struct XsumPre {
  float v[16];
  u32x4_t w[2];
};

__host__ __device__ inline bool xsum_shape_ok(int K, int waves) { return (K >> 4) <= 64 * waves && (K >> 3) <= 2 * waves * 64 && !(K & 31); }

__device__ __forceinline__ void xsum_issue(const XsumIn& xs, int K, int wave, int nw, int lane, XsumPre& pre) {
  const int mcol = lane & 15, q4 = lane >> 4;
  const int G = K >> 4;
  const int Gw = (G + nw - 1) / nw;
  const int g0 = wave * Gw, g1 = min(G, g0 + Gw);
  // every load unconditional (index clamped, value selected afterwards): a predicated load is a branch, and the compiler put the first
  // add -- and with it a full wait -- inside the first one
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int g = g0 + q4 + 4 * i;
    pre.v[i] = xs.ssp[min(g, G - 1) * 16 + mcol];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * blockDim.x;
    pre.w[i] = reinterpret_cast<const u32x4_t*>(xs.norm_w)[min(c, (K >> 3) - 1)];
  }
}

__device__ __forceinline__ float xsum_finish(const XsumIn& xs, const XsumPre& pre, int K, u32x4_t* wlds, float* part, int wave, int nw, int lane) {
  const int mcol = lane & 15, q4 = lane >> 4;
  const int Gw = ((K >> 4) + nw - 1) / nw;
  const int gq = wave * Gw + q4, g1 = min(K >> 4, wave * Gw + Gw);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += (gq + 4 * i < g1) ? pre.v[i] : 0.f;
  t += __shfl_xor(t, 16, 64);
  t += __shfl_xor(t, 32, 64);
  if (q4 == 0) part[wave * 16 + mcol] = t;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * blockDim.x;
    if (c < (K >> 3)) wlds[c] = pre.w[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // LDS only: the weight stages stay in flight
  float tot = 0.f;
  for (int w = 0; w < nw; ++w) tot += part[w * 16 + mcol];
  return 1.0f / sqrtf(tot / (float)K + xs.eps);
}

// x32 of this lane's B-operand chunk (8 consecutive k of row lane & 15): two 16-byte loads
__device__ __forceinline__ void xsum_load(const float* x32f, size_t chunk, bool live, f32x4_t (&xr)[2]) {
  xr[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  xr[1] = xr[0];
  if (live) {
    const f32x4_t* q = reinterpret_cast<const f32x4_t*>(x32f) + 2 * chunk;
    xr[0] = q[0];
    xr[1] = q[1];
  }
}

// x^ = bf16((x32 * rs) * w): rmsnorm_kernel's expression (norm.hip)
__device__ __forceinline__ u32x4_t xsum_bfrag(const f32x4_t (&xr)[2], float rs, u32x4_t wv) {
  u32x4_t o;
  // (v_cvt_pk_bf16_f32: bit-identical to the integer round-to-nearest-even of pack_bf2 on all 2^32 inputs -- ssd_selftest_bf16_cvt --
  //  and a third of its instructions: this conversion sits between a stage's arrival and the refill of its buffer)
  o[0] = pack_bf2_hw((xr[0][0] * rs) * bf2f(wv[0] & 0xffffu), (xr[0][1] * rs) * bf2f(wv[0] >> 16));
  o[1] = pack_bf2_hw((xr[0][2] * rs) * bf2f(wv[1] & 0xffffu), (xr[0][3] * rs) * bf2f(wv[1] >> 16));
  o[2] = pack_bf2_hw((xr[1][0] * rs) * bf2f(wv[2] & 0xffffu), (xr[1][1] * rs) * bf2f(wv[2] >> 16));
  o[3] = pack_bf2_hw((xr[1][2] * rs) * bf2f(wv[3] & 0xffffu), (xr[1][3] * rs) * bf2f(wv[3] >> 16));
  return o;
}

// Producer epilogue for the accumulator quad s of (row m, columns n .. n + 3), n = 16 * group + 4 * (lane >> 4): all 64 lanes of the
// wave call it (the group sum is a wave reduction over the four lane quarters); rows >= M store nothing.  `rv`: res_in[m][n .. n + 3],
// loaded by the caller ahead of the split-K combine.
__device__ __forceinline__ void xsum_epilogue(const XsumOut& xo, f32x4_t s, u32x2_t rv, int m, int n, int M, int N, int lane) {
  float x[4];
  x[0] = round_bf(s[0]) + bf2f(rv[0] & 0xffffu);
  x[1] = round_bf(s[1]) + bf2f(rv[0] >> 16);
  x[2] = round_bf(s[2]) + bf2f(rv[1] & 0xffffu);
  x[3] = round_bf(s[3]) + bf2f(rv[1] >> 16);
  float ss = x[0] * x[0];
  ss += x[1] * x[1]; ss += x[2] * x[2]; ss += x[3] * x[3];
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (m < M) {
    *reinterpret_cast<u32x2_t*>(xo.res_out + (size_t)m * N + n) = u32x2_t{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
    *reinterpret_cast<f32x4_t*>(xo.x32f + frag_chunk(m, n >> 3, N >> 5) * 8 + (n & 4)) = f32x4_t{x[0], x[1], x[2], x[3]};
    if ((lane >> 4) == 0) xo.ssp[(n >> 4) * 16 + m] = ss;
  }
}
