// Fused (optional per-head RMSNorm) + neox RoPE + paged KV-cache store.
// Replaces, in one launch:
//   * RMSHeadNorm q_norm/k_norm (Qwen3 only), reference ssd/models/qwen3.py:96-104, ssd/layers/layernorm.py:16-40
//   * RotaryEmbedding.forward, reference ssd/layers/rotary_embedding.py:6-60 (fp32 math from the fp32
//     cos||sin table gathered by `positions`, one rounding to bf16)
//   * the Triton store_kvcache kernel, reference ssd/layers/attention.py:10-41 (slot == -1 -> skip)
// KV cache layout here is [num_blocks][n_kv_heads][block_size][head_dim] per layer and per K/V
// ("HND"): one (page, kv-head) is a contiguous 64 KiB run, which is what the attention kernel streams.
// The reference's slot semantics are kept: slot = block_id * block_size + pos_in_block.
#include "common.h"

// PARTS: the QKV rows are not materialised -- the prefill GEMM (gemm_pf.hip, PF_EPI_PARTIALS) left S fp32 split-K slabs
// parts[s][t][n]; a row's value is bf16(slab 0 + slab 1 + ... in that order), exactly what gemm_pf_epilogue_kernel would have
// stored, so the result is bit-identical to "epilogue launch, then this kernel" with one launch less per layer.
template <bool PARTS>
__global__ void rope_store_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ parts, int S, size_t slab,
                                  const int64_t* __restrict__ positions,
                                  const float* __restrict__ cos_sin, const int32_t* __restrict__ slots,
                                  bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache,
                                  bf16_t* __restrict__ v_cache, const bf16_t* __restrict__ qn_w,
                                  const bf16_t* __restrict__ kn_w, float eps, int nh, int nkv, int hd, int bs, int perm) {
  const int t = blockIdx.x;
  const int c16 = hd >> 4;                 // threads per rotated head (each owns 8+8 elements)
  const int rot_items = (nh + nkv) * c16;  // q and k heads
  const int v_items = nkv * (hd >> 3);
  const int row_w = (nh + 2 * nkv) * hd;
  const bf16_t* row = qkv + (size_t)t * row_w;
  // 8 consecutive columns of row t as fp32 values of bf16 numbers
  auto load8 = [&](int col, float (&x)[8]) {
    if (PARTS) {
      // every slab's loads are issued before the first add (a rolled `for z < S` is load -> wait -> add per slab: S dependent round trips
      // in a kernel that is a latency chain) -- the same additions in the same order (norm.hip's sum_slabs, the same reasoning)
      const float* src = parts + (size_t)t * row_w + col;
      // (the sum STARTS from slab 0 -- not from +0.0 -- so that a row of -0.0 partials stays -0.0 like the epilogue kernel's)
      f32x4_t lo, hi;
      for (int z0 = 0; z0 < S; z0 += 4) {          // block-uniform
        f32x4_t ta[4], tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* sj = src + (size_t)min(z0 + j, S - 1) * slab;
          ta[j] = *reinterpret_cast<const f32x4_t*>(sj);
          tb[j] = *reinterpret_cast<const f32x4_t*>(sj + 4);
        }
        if (z0 == 0) { lo = ta[0]; hi = tb[0]; } else { lo += ta[0]; hi += tb[0]; }
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (z0 + j < S) { lo += ta[j]; hi += tb[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = round_bf(lo[j]); x[4 + j] = round_bf(hi[j]); }
    } else {
      const u32x4_t a = *reinterpret_cast<const u32x4_t*>(row + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[2 * j] = bf2f(a[j] & 0xffffu); x[2 * j + 1] = bf2f(a[j] >> 16); }
    }
  };
  const long pos = positions[t];
  const int slot = slots ? slots[t] : -1;
  const float* cs = cos_sin + (size_t)pos * hd;
  const int half = hd >> 1;
  long kv_base = -1;
  if (slot >= 0) kv_base = ((long)(slot / bs) * nkv) * bs + (slot % bs);  // + kvh*bs, then * hd

  // (gridDim.y workgroups share a token row: the single-chunk prefill's 128 rows alone fill half the chip)
  for (int it = blockIdx.y * blockDim.x + threadIdx.x; it < rot_items + v_items; it += blockDim.x * gridDim.y) {
    if (it < rot_items) {
      const int head = it / c16, c = it % c16;
      const bool is_q = head < nh;
      const int src = head * hd;                    // q heads then k heads are contiguous in qkv
      // perm: the QKV GEMM wrote q/k heads in the rotation-paired order (layout.hip): chunk c of the first half
      // and its partner chunk of the second half are adjacent
      float x1[8], x2[8];
      load8(src + (perm ? c * 16 : c * 8), x1);
      load8(src + (perm ? c * 16 + 8 : half + c * 8), x2);
      const bf16_t* nw = is_q ? qn_w : kn_w;
      if (nw) {  // per-head RMSNorm, rounded to bf16 before the rotation (separate kernels in the reference)
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ss += x1[j] * x1[j]; ss += x2[j] * x2[j]; }
        for (int o = 1; o < c16; o <<= 1) ss += __shfl_xor(ss, o, 64);
        const float rs = 1.0f / sqrtf(ss / (float)hd + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          x1[j] = round_bf((x1[j] * rs) * bf2f(nw[c * 8 + j]));
          x2[j] = round_bf((x2[j] * rs) * bf2f(nw[half + c * 8 + j]));
        }
      }
      float y1[8], y2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float co = cs[c * 8 + j], si = cs[half + c * 8 + j];
        y1[j] = __fsub_rn(__fmul_rn(x1[j], co), __fmul_rn(x2[j], si));
        y2[j] = __fadd_rn(__fmul_rn(x2[j], co), __fmul_rn(x1[j], si));
      }
      const u32x4_t o1 = {pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3]), pack_bf2(y1[4], y1[5]), pack_bf2(y1[6], y1[7])};
      const u32x4_t o2 = {pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3]), pack_bf2(y2[4], y2[5]), pack_bf2(y2[6], y2[7])};
      if (is_q) {
        bf16_t* dst = q_out + ((size_t)t * nh + head) * hd;
        *reinterpret_cast<u32x4_t*>(dst + c * 8) = o1;
        *reinterpret_cast<u32x4_t*>(dst + half + c * 8) = o2;
      } else if (kv_base >= 0) {
        bf16_t* dst = k_cache + (size_t)(kv_base + (long)(head - nh) * bs) * hd;
        *reinterpret_cast<u32x4_t*>(dst + c * 8) = o1;
        *reinterpret_cast<u32x4_t*>(dst + half + c * 8) = o2;
      }
    } else if (kv_base >= 0) {
      const int vi = it - rot_items;
      const int kvh = vi / (hd >> 3), c = vi % (hd >> 3);
      u32x4_t v;
      if (PARTS) {
        float xv[8];
        load8((nh + nkv + kvh) * hd + c * 8, xv);
        v = u32x4_t{pack_bf2(xv[0], xv[1]), pack_bf2(xv[2], xv[3]), pack_bf2(xv[4], xv[5]), pack_bf2(xv[6], xv[7])};
      } else {
        v = *reinterpret_cast<const u32x4_t*>(row + (size_t)(nh + nkv + kvh) * hd + c * 8);
      }
      *reinterpret_cast<u32x4_t*>(v_cache + (size_t)(kv_base + (long)kvh * bs) * hd + c * 8) = v;
    }
  }
}

static int rope_store_launch(const void* qkv_rows, const float* parts, int S, const int64_t* positions, const float* cos_sin,
                             const int32_t* slot_mapping, void* q_out_rows, void* k_cache, void* v_cache,
                             const void* q_norm_w, const void* k_norm_w, float eps, int T, int nh, int nkv,
                             int hd, int block_size, int qkv_perm, void* stream) {
  if (T <= 0 || nh <= 0 || nkv <= 0 || (hd != 64 && hd != 128 && hd != 256) || block_size <= 0) return SSD_ERR_SHAPE;
  const int items = (nh + nkv) * (hd / 16) + nkv * (hd / 8);
  int threads = ((items + 63) / 64) * 64;
  if (threads > 512) threads = 512;
  const size_t slab = (size_t)T * (nh + 2 * nkv) * hd;
  if (parts) {
    // prefill-sized launches: 256-thread workgroups, as many per token row as the row has items (70B: 704 items -> 3 x 128 rows = 384
    // workgroups instead of 128 of 512 threads; in situ 11.6 -> 9.2 us per 70B layer, profiles/r06_c4_prefill_timeline.txt).  The c16 threads of a head
    // (per-head norm shuffles) stay inside one wave: item runs of a head start at multiples of c16 <= 16.
    threads = 256;
    const int gy = (items + threads - 1) / threads;
    hipLaunchKernelGGL(rope_store_kernel<true>, dim3(T, gy), dim3(threads), 0, (hipStream_t)stream, (const bf16_t*)nullptr, parts, S,
                       slab, positions, cos_sin, slot_mapping, (bf16_t*)q_out_rows, (bf16_t*)k_cache, (bf16_t*)v_cache,
                       (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, eps, nh, nkv, hd, block_size, qkv_perm);
  } else
    hipLaunchKernelGGL(rope_store_kernel<false>, dim3(T), dim3(threads), 0, (hipStream_t)stream, (const bf16_t*)qkv_rows,
                       (const float*)nullptr, 0, slab, positions, cos_sin, slot_mapping, (bf16_t*)q_out_rows, (bf16_t*)k_cache,
                       (bf16_t*)v_cache, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, eps, nh, nkv, hd, block_size, qkv_perm);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_rope_store_kv(const void* qkv_rows, const int64_t* positions, const float* cos_sin,
                                 const int32_t* slot_mapping, void* q_out_rows, void* k_cache, void* v_cache,
                                 const void* q_norm_w, const void* k_norm_w, float eps, int T, int nh, int nkv,
                                 int hd, int block_size, int qkv_perm, void* stream) {
  if (!qkv_rows) return SSD_ERR_ARG;
  return rope_store_launch(qkv_rows, nullptr, 0, positions, cos_sin, slot_mapping, q_out_rows, k_cache, v_cache, q_norm_w, k_norm_w,
                           eps, T, nh, nkv, hd, block_size, qkv_perm, stream);
}

// Same, reading the QKV rows from `splits` fp32 split-K slabs [splits][T][(nh + 2 nkv) * hd] (ssd_gemm_pf with PF_EPI_PARTIALS).
extern "C" int ssd_rope_store_kv_parts(const float* parts, int splits, const int64_t* positions, const float* cos_sin,
                                       const int32_t* slot_mapping, void* q_out_rows, void* k_cache, void* v_cache,
                                       const void* q_norm_w, const void* k_norm_w, float eps, int T, int nh, int nkv,
                                       int hd, int block_size, int qkv_perm, void* stream) {
  if (!parts || splits < 1 || splits > 16) return SSD_ERR_ARG;
  return rope_store_launch(nullptr, parts, splits, positions, cos_sin, slot_mapping, q_out_rows, k_cache, v_cache, q_norm_w,
                           k_norm_w, eps, T, nh, nkv, hd, block_size, qkv_perm, stream);
}
