// Embedding gather and fused (residual-add +) RMSNorm.
//  * ssd_embedding  replaces VocabParallelEmbedding.forward, reference ssd/layers/embed_head.py:49-57
//  * ssd_rmsnorm    replaces RMSDNorm.norm_forward / add_norm_forward, reference
//                   ssd/layers/layernorm.py:64-88 (as run under torch.compile: fp32 throughout, one
//                   rounding at each store: res_out = bf16(x32), y = bf16(x32 * rsqrt(mean(x32^2)+eps) * w32)).
// The normalised output can be written row-major and/or directly in the fragment-major layout that
// the next skinny GEMM consumes (common.h), so no separate re-layout pass exists on the hot path.
#include "common.h"
#include <type_traits>

__global__ void embedding_kernel(const int64_t* __restrict__ ids, const u32x4_t* __restrict__ table,
                                 u32x4_t* __restrict__ out, int H8, long vocab_start, long vocab_count) {
  const int t = blockIdx.x;
  const long id = ids[t] - vocab_start;
  const bool ok = id >= 0 && id < vocab_count;
  const u32x4_t* src = table + (size_t)(ok ? id : 0) * H8;
  for (int c = threadIdx.x; c < H8; c += blockDim.x) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (ok) v = src[c];
    out[(size_t)t * H8 + c] = v;
  }
}

extern "C" int ssd_embedding(const int64_t* ids, const void* table, void* out_rows, int T, int H,
                             long vocab_start, long vocab_count, void* stream) {
  if (T <= 0 || H <= 0 || (H & 7)) return SSD_ERR_SHAPE;
  int threads = H / 8; if (threads > 256) threads = 256; if (threads < 64) threads = 64;
  hipLaunchKernelGGL(embedding_kernel, dim3(T), dim3(threads), 0, (hipStream_t)stream, ids,
                     (const u32x4_t*)table, (u32x4_t*)out_rows, H / 8, vocab_start, vocab_count);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

constexpr int NORM_MAXH = 16384;   // chunks of 8 elements: 256 threads x 8 chunks, or 1024 x 2

template <int NORM_THREADS>
__global__ void __launch_bounds__(NORM_THREADS)
rmsnorm_kernel(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ res_in, u32x4_t* __restrict__ res_out,
               const u32x4_t* __restrict__ w, float eps, u32x4_t* __restrict__ out_rows,
               u32x4_t* __restrict__ out_frag, const int32_t* __restrict__ gather, int H,
               const float* __restrict__ parts, int S, int slab_rows) {
  __shared__ float red[NORM_THREADS / 64];
  const int KTS = parts ? 10 : 11;        // trace slot (profiling builds only)
  KTRACE(KTS, 0);
  const int row_out = blockIdx.x;
  const int row_in = gather ? gather[row_out] : row_out;
  const int H8 = H >> 3;
  const int KT = H >> 5;
  constexpr int NORM_MAXC = NORM_MAXH / 8 / NORM_THREADS;
  float v[NORM_MAXC][8];
  u32x4_t wreg[NORM_MAXC];      // the norm weight, fetched with the inputs (after the barrier it was one more dependent round trip)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < H8) {
      u32x4_t xv;
      wreg[i] = w[c];
      u32x4_t rv = {0u, 0u, 0u, 0u};
      if (res_in) rv = res_in[(size_t)row_in * H8 + c];
      if (parts) {      // x = bf16(sum of the producer GEMM's S fp32 partial slabs, in slab order) -- csrc/gemm_sk.hip ssd_gemm_parts
        // every slab's loads are issued before the first add (a rolled `for s < S` is load -> wait -> add per slab: S dependent
        // L2 round trips in a kernel that is nothing but a latency chain, 3.8 us at S = 4) -- the same additions in the same order
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = a;
        const float* src0 = parts + (size_t)row_in * H + c * 8;
        const size_t sstride = (size_t)slab_rows * H;
        // one straight-line path per slab count class (block-uniform switch): SN loads issued back to back, then SN adds.  (Round 4
        // tried "always 8 clamped loads" -- better at S = 4, worse at S = 2 -- and "conditional loads + opaque uses" -- worse
        // everywhere, the code tripled: profiles/r04_ktrace_1b_after_v1.txt / _after2.txt.)
        auto sum_slabs = [&](auto sn, int s0) {
          constexpr int SN = decltype(sn)::value;
          f32x4_t ta[SN], tb[SN];
#pragma unroll
          for (int j = 0; j < SN; ++j) {
            const float* src = src0 + (size_t)min(s0 + j, S - 1) * sstride;
            ta[j] = *reinterpret_cast<const f32x4_t*>(src);
            tb[j] = *reinterpret_cast<const f32x4_t*>(src + 4);
          }
#pragma unroll
          for (int j = 0; j < SN; ++j)
            if (s0 + j < S) { a += ta[j]; b += tb[j]; }
        };
        if (S <= 2) sum_slabs(std::integral_constant<int, 2>{}, 0);
        else if (S <= 4) sum_slabs(std::integral_constant<int, 4>{}, 0);
        else {
          sum_slabs(std::integral_constant<int, 8>{}, 0);
          if (S > 8) sum_slabs(std::integral_constant<int, 8>{}, 8);
        }
        xv = u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
      } else {
        xv = x[(size_t)row_in * H8 + c];
      }
      u32x4_t ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = bf2f(xv[j] & 0xffffu), hi = bf2f(xv[j] >> 16);
        if (res_in) { lo += bf2f(rv[j] & 0xffffu); hi += bf2f(rv[j] >> 16); }
        v[i][2 * j] = lo; v[i][2 * j + 1] = hi;
        ro[j] = pack_bf2(lo, hi);
        ss += lo * lo; ss += hi * hi;
      }
      if (res_out) res_out[(size_t)row_out * H8 + c] = ro;
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_THREADS / 64; ++i) tot += red[i];
  const float rs = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < H8) {
      const u32x4_t wv = wreg[i];
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = (v[i][2 * j] * rs) * bf2f(wv[j] & 0xffffu);
        const float hi = (v[i][2 * j + 1] * rs) * bf2f(wv[j] >> 16);
        o[j] = pack_bf2(lo, hi);
      }
      if (out_rows) out_rows[(size_t)row_out * H8 + c] = o;
      if (out_frag) out_frag[frag_chunk(row_out, c, KT)] = o;
    }
  }
  KTRACE(KTS, 6);
}

extern "C" int ssd_rmsnorm(const void* x_rows, const void* res_in, void* res_out, const void* weight, float eps,
                           void* out_rows, void* out_frag, const int32_t* gather_rows, int T, int H, void* stream) {
  if (T <= 0 || H <= 0 || (H & 31) || H > NORM_MAXH) return SSD_ERR_SHAPE;
  if (ssd_norm_threads(H) == 1024)
    hipLaunchKernelGGL(rmsnorm_kernel<1024>, dim3(T), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t*)x_rows,
                       (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, (u32x4_t*)out_rows,
                       (u32x4_t*)out_frag, gather_rows, H, (const float*)nullptr, 0, 0);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<256>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x_rows,
                       (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, (u32x4_t*)out_rows,
                       (u32x4_t*)out_frag, gather_rows, H, (const float*)nullptr, 0, 0);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Two independent RMSNorms written side by side into ONE [T][2H] fragment-major activation: the input of the EAGLE-3
// draft layer's QKV projection, cat([input_layernorm(token embeddings), conditioning_feature_ln(features)], -1)
// (ssd/models/eagle3_draft_llama3.py:148-150).  blockIdx.y selects the half; each half is exactly ssd_rmsnorm's
// norm_forward (fp32 math, one rounding at the store), so the concatenation costs no extra pass over the rows.
template <int NORM_THREADS>
__global__ void __launch_bounds__(NORM_THREADS)
rmsnorm_pair_kernel(const u32x4_t* __restrict__ x0, const u32x4_t* __restrict__ w0, const u32x4_t* __restrict__ x1,
                    const u32x4_t* __restrict__ w1, float eps, u32x4_t* __restrict__ out_frag, int H) {
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x, half = blockIdx.y;
  const u32x4_t* x = half ? x1 : x0;
  const u32x4_t* w = half ? w1 : w0;
  const int H8 = H >> 3;
  const int KT2 = H >> 4;                 // k-tiles of the 2H-wide output
  constexpr int NORM_MAXC = NORM_MAXH / 8 / NORM_THREADS;
  float v[NORM_MAXC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < H8) {
      const u32x4_t xv = x[(size_t)row * H8 + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = bf2f(xv[j] & 0xffffu), hi = bf2f(xv[j] >> 16);
        v[i][2 * j] = lo; v[i][2 * j + 1] = hi;
        ss += lo * lo; ss += hi * hi;
      }
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_THREADS / 64; ++i) tot += red[i];
  const float rs = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < H8) {
      const u32x4_t wv = w[c];
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = (v[i][2 * j] * rs) * bf2f(wv[j] & 0xffffu);
        const float hi = (v[i][2 * j + 1] * rs) * bf2f(wv[j] >> 16);
        o[j] = pack_bf2(lo, hi);
      }
      out_frag[frag_chunk(row, c + half * H8, KT2)] = o;
    }
  }
}

extern "C" int ssd_rmsnorm_pair(const void* x0_rows, const void* weight0, const void* x1_rows, const void* weight1, float eps,
                                void* out_frag, int T, int H, void* stream) {
  if (T <= 0 || H <= 0 || (H & 31) || H > NORM_MAXH) return SSD_ERR_SHAPE;
  if (!x0_rows || !x1_rows || !weight0 || !weight1 || !out_frag) return SSD_ERR_ARG;
  if (ssd_norm_threads(H) == 1024)
    hipLaunchKernelGGL(rmsnorm_pair_kernel<1024>, dim3(T, 2), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t*)x0_rows,
                       (const u32x4_t*)weight0, (const u32x4_t*)x1_rows, (const u32x4_t*)weight1, eps, (u32x4_t*)out_frag, H);
  else
    hipLaunchKernelGGL(rmsnorm_pair_kernel<256>, dim3(T, 2), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x0_rows,
                       (const u32x4_t*)weight0, (const u32x4_t*)x1_rows, (const u32x4_t*)weight1, eps, (u32x4_t*)out_frag, H);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Same with x given as `splits` fp32 partial slabs [splits][slab_rows][H] of the producing split-K GEMM (ssd_gemm_parts):
// x = bf16(sum over the slabs in order), then exactly ssd_rmsnorm.
extern "C" int ssd_rmsnorm_parts(const void* parts, int splits, int slab_rows, const void* res_in, void* res_out,
                                 const void* weight, float eps, void* out_rows, void* out_frag, int T, int H, void* stream) {
  if (T <= 0 || H <= 0 || (H & 31) || H > NORM_MAXH || splits < 1 || splits > 16 || slab_rows < T) return SSD_ERR_SHAPE;
  if (!parts) return SSD_ERR_ARG;
  if (ssd_norm_threads(H) == 1024)
    hipLaunchKernelGGL(rmsnorm_kernel<1024>, dim3(T), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t*)nullptr,
                       (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, (u32x4_t*)out_rows,
                       (u32x4_t*)out_frag, (const int32_t*)nullptr, H, (const float*)parts, splits, slab_rows);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<256>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)nullptr,
                       (const u32x4_t*)res_in, (u32x4_t*)res_out, (const u32x4_t*)weight, eps, (u32x4_t*)out_rows,
                       (u32x4_t*)out_frag, (const int32_t*)nullptr, H, (const float*)parts, splits, slab_rows);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone forms of two ops the hot path only runs fused (GEMM epilogue / RoPE kernel): the module-level binding points of the
// reference (VERDICT r4 "missing" 3).
// ---------------------------------------------------------------------------------------------------------------------
// RMSHeadNorm.forward -- ssd/layers/layernorm.py:16-40 (Qwen3 q_norm / k_norm, ssd/models/qwen3.py:96-104): x [T][heads][hd] normalised
// over hd per (token, head).  hd / 16 threads per head, each owning elements [8c, 8c + 8) and [hd/2 + 8c, hd/2 + 8c + 8): the arithmetic
// and the summation order of rope_store_kernel's fused norm (csrc/rope.hip), so the two are bit-identical.
__global__ void head_rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, float eps, bf16_t* __restrict__ y,
                                    long items, int hd) {
  const int c16 = hd >> 4, half = hd >> 1;
  const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = it < items;
  const long head = live ? it / c16 : 0;
  const int c = live ? (int)(it % c16) : 0;
  const bf16_t* src = x + head * hd;
  float x1[8], x2[8];
  {
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(src + c * 8), b = *reinterpret_cast<const u32x4_t*>(src + half + c * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x1[2 * j] = bf2f(a[j] & 0xffffu); x1[2 * j + 1] = bf2f(a[j] >> 16);
      x2[2 * j] = bf2f(b[j] & 0xffffu); x2[2 * j + 1] = bf2f(b[j] >> 16);
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ss += x1[j] * x1[j]; ss += x2[j] * x2[j]; }
  for (int o = 1; o < c16; o <<= 1) ss += __shfl_xor(ss, o, 64);          // (a head's threads are lane-aligned: c16 divides 64)
  const float rs = 1.0f / sqrtf(ss / (float)hd + eps);
  if (!live) return;
  u32x4_t o1, o2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o1[j] = pack_bf2((x1[2 * j] * rs) * bf2f(w[c * 8 + 2 * j]), (x1[2 * j + 1] * rs) * bf2f(w[c * 8 + 2 * j + 1]));
    o2[j] = pack_bf2((x2[2 * j] * rs) * bf2f(w[half + c * 8 + 2 * j]), (x2[2 * j + 1] * rs) * bf2f(w[half + c * 8 + 2 * j + 1]));
  }
  bf16_t* dst = y + head * hd;
  *reinterpret_cast<u32x4_t*>(dst + c * 8) = o1;
  *reinterpret_cast<u32x4_t*>(dst + half + c * 8) = o2;
}

extern "C" int ssd_head_rmsnorm(const void* x_rows, const void* weight, float eps, void* out_rows, int T, int heads, int hd, void* stream) {
  if (T <= 0 || heads <= 0 || (hd != 64 && hd != 128 && hd != 256)) return SSD_ERR_SHAPE;
  if (!x_rows || !weight || !out_rows) return SSD_ERR_ARG;
  const long items = (long)T * heads * (hd >> 4);
  hipLaunchKernelGGL(head_rmsnorm_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_rows,
                     (const bf16_t*)weight, eps, (bf16_t*)out_rows, items, hd);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// SiluAndMul.forward -- ssd/layers/activation.py:11-14: x [T][2 I] = [gate | up] -> silu(gate) * up, fp32 math on the bf16 inputs, one
// rounding: the arithmetic of the gate_up GEMM's fused epilogue (csrc/gemm.hip EPI_SILU_FRAG), as rows and / or fragment-major.
__global__ void silu_mul_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y_rows, u32x4_t* __restrict__ y_frag, int T, int I) {
  const int I8 = I >> 3;
  const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= (long)T * I8) return;
  const int t = (int)(it / I8), c = (int)(it % I8);
  const u32x4_t g = *reinterpret_cast<const u32x4_t*>(x + (size_t)t * 2 * I + c * 8);
  const u32x4_t u = *reinterpret_cast<const u32x4_t*>(x + (size_t)t * 2 * I + I + c * 8);
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g0 = bf2f(g[j] & 0xffffu), g1 = bf2f(g[j] >> 16), u0 = bf2f(u[j] & 0xffffu), u1 = bf2f(u[j] >> 16);
    o[j] = pack_bf2((g0 / (1.0f + __expf(-g0))) * u0, (g1 / (1.0f + __expf(-g1))) * u1);
  }
  if (y_rows) *reinterpret_cast<u32x4_t*>(y_rows + (size_t)t * I + c * 8) = o;
  if (y_frag) y_frag[frag_chunk(t, c, I >> 5)] = o;
}

extern "C" int ssd_silu_mul(const void* x_rows, void* out_rows, void* out_frag, int T, int I, void* stream) {
  if (T <= 0 || I <= 0 || (I & 31)) return SSD_ERR_SHAPE;
  if (!x_rows || (!out_rows && !out_frag)) return SSD_ERR_ARG;
  const long items = (long)T * (I >> 3);
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_rows,
                     (bf16_t*)out_rows, (u32x4_t*)out_frag, T, I);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

KT_DEFINE_SETTER(norm)
