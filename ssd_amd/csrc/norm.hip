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

KT_DEFINE_SETTER(norm)
