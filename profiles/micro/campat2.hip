// Round 6: which ingredient of the prefill GEMM's k loop keeps it from overlapping with its weight stream?  The access structure of
// csrc/gemm_pf.hip's o_proj launch ([8192 x 8192], 64 tiles x 4 K splits = 256 workgroups of 4 waves, 2 row groups per wave, U k-steps of
// weights in flight per wave in a register ring, every load unconditional) with the loop body built up step by step:
//   MODE 0  weights only (xor)
//   MODE 1  + 16 MFMA 16x16x32 per k-step, B operands constant registers
//   MODE 2  + the B operands read from LDS per k-step (8 ds_read_b128, LDS filled once)
//   MODE 3  + one __syncthreads per 2 k-steps
//   MODE 4  + the x tile loaded from global (L2) 3 phases ahead and staged into the LDS ring (the real loop's traffic, minus BPRE)
// `rot` de-phases the workgroups (each starts at its own k offset and wraps).
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/campat2 profiles/micro/campat2.hip && /tmp/campat2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int WAVES, int U, int MODE>
__global__ void __launch_bounds__(64 * WAVES, WAVES <= 4 ? 2 : 1)
rd(const u32x4* __restrict__ W, const u32x4* __restrict__ X, size_t rg_stride, int KT, int nk, int rot, float* out) {
  __shared__ u32x4 xs[4][8][64];          // two phases of two k-steps, 8 fragment tiles each
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t g0 = ((size_t)blockIdx.x * WAVES + wave) * 2;
  const int kz0 = blockIdx.y * nk;
  const u32x4* p0 = W + g0 * rg_stride + lane;
  const u32x4* p1 = p0 + rg_stride;
  const int start = rot ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nk) & ~7 : 0;
  auto kk = [&](int k) { k += start; if (k >= nk) k -= nk; return kz0 + k; };
  for (int i = threadIdx.x; i < 4 * 8 * 64; i += blockDim.x) (&xs[0][0][0])[i] = X[i];
  __syncthreads();
  f32x4 acc[2][8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 bc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) bc[m] = X[m * 64 + lane];
  u32x4 a[U][2];
  u32x4 xr[2][2][2];                      // [register set][k-step of the phase][tile of this wave]
  const int klast = nk - 1;
  auto ldw = [&](u32x4 (&d)[2], int k) {
    d[0] = __builtin_nontemporal_load(p0 + ((size_t)kk(k) << 6));
    d[1] = __builtin_nontemporal_load(p1 + ((size_t)kk(k) << 6));
  };
  auto ldx = [&](u32x4 (&d)[2][2], int ph) {      // phase ph = k-steps 2 ph, 2 ph + 1; this wave's tiles: wave, wave + WAVES (clamped)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int mt = min(wave + WAVES * f, 7);
        d[b][f] = X[((size_t)mt * KT + kk(min(2 * ph + b, klast))) * 64 + lane];
      }
  };
  auto stx = [&](const u32x4 (&d)[2][2], int half) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        if (wave + WAVES * f < 8) xs[half * 2 + b][wave + WAVES * f][lane] = d[b][f];
  };
  if (MODE >= 4) { ldx(xr[0], 1); }
#pragma unroll
  for (int u = 0; u < U; ++u) ldw(a[u], u);
  if (MODE >= 4) { ldx(xr[1], 2); }
  for (int k = 0; k < nk; k += U) {
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      const int ph = (k + u) >> 1, par = (u >> 1) & 1;
      if (MODE >= 4) {
        stx(xr[par], par ^ 1);              // next phase's tiles -> the other ring half
        ldx(xr[par], ph + 3);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (MODE == 0) {
          acc[0][0] += __builtin_bit_cast(f32x4, a[u + b][0]);
          acc[1][0] += __builtin_bit_cast(f32x4, a[u + b][1]);
        } else {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const u32x4 bv = MODE >= 2 ? xs[par * 2 + b][m][lane] : bc[m];
            acc[0][m] = mfma16(a[u + b][0], bv, acc[0][m]);
            acc[1][m] = mfma16(a[u + b][1], bv, acc[1][m]);
          }
        }
        ldw(a[u + b], min(k + u + b + U, klast));
      }
      if (MODE >= 3) __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) s += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
  if (s == 12345.678f) *out = s;
}

template <int WAVES, int U, int MODE>
static float run(const u32x4* W, const u32x4* X, size_t stride_kib, int tiles, int S, int KT, int nk, int rot, float* out, int copies, size_t copy_chunks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((rd<WAVES, U, MODE>), dim3(tiles, S), dim3(64 * WAVES), 0, 0, W + (i % copies) * copy_chunks, X, stride_kib * 64, KT, nk, rot, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((rd<WAVES, U, MODE>), dim3(tiles, S), dim3(64 * WAVES), 0, 0, W + (i % copies) * copy_chunks, X, stride_kib * 64, KT, nk, rot, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const size_t groups = 640, stride_kib = 256, copies = 4;
  const size_t copy_chunks = groups * stride_kib * 64;
  u32x4 *W, *X;
  float* out;
  if (hipMalloc(&W, copies * copy_chunks * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&X, (size_t)8 * 256 * 64 * 16);       // x: 8 fragment tiles x 256 k-tiles x 1 KiB = 2 MiB
  hipMalloc(&out, 4);
  hipMemset(W, 0, copies * copy_chunks * 16);
  hipMemset(X, 0, (size_t)8 * 256 * 64 * 16);
  const int KT = 256;
  const double mb = 134.2;
#define ROW(WV, UU, MODE, tiles, S, nk, name)                                                                         \
  for (int rot = 0; rot < 2; ++rot) {                                                                                 \
    float us = run<WV, UU, MODE>(W, X, stride_kib, tiles, S, KT, nk, rot, out, copies, copy_chunks);                  \
    printf("%-34s waves %d U %2d mode %d rot %d : %7.2f us  %5.2f TB/s\n", name, WV, UU, MODE, rot, us, mb / us);    \
  }
  ROW(4, 8, 0, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 8, 1, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 8, 2, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 8, 3, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 8, 4, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 16, 0, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 16, 1, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 16, 2, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(4, 16, 3, 64, 4, 64, "o_proj 64 tiles x 4 splits")
  ROW(8, 8, 0, 32, 8, 32, "o_proj 32 tiles x 8 splits")
  ROW(8, 8, 1, 32, 8, 32, "o_proj 32 tiles x 8 splits")
  ROW(8, 8, 3, 32, 8, 32, "o_proj 32 tiles x 8 splits")
  ROW(8, 8, 4, 32, 8, 32, "o_proj 32 tiles x 8 splits")
  return 0;
}
