// Row-major <-> fragment-major layout conversion (one-time weight pre-shuffle at load, and test/debug
// helpers for activations).  The fragment-major layout is defined in common.h (frag_chunk).
// The reference keeps nn.Linear weights [out,in] row-major (ssd/layers/linear.py:58-62,84-89) and lets
// cuBLAS pick its own tiling; on CDNA4 we pre-tile once so that every decode-time wave load is a
// contiguous 1 KiB MFMA operand.
#include "common.h"

// mode 0: identity row order.
// mode 1: "interleave halves": the source is [gate(I rows) ; up(I rows)] (MergedColumnParallelLinear,
//         reference ssd/layers/linear.py:101-122) and destination 16-row groups alternate
//         gate-group, up-group, ... so one workgroup sees both operands of SiLU(g)*u.
__global__ void shuffle_rows_to_frag_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst,
                                            int R, int K, int mode, long total_chunks) {
  const int KT = K >> 5;
  for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < total_chunks;
       c += (long)gridDim.x * blockDim.x) {
    const long tile = c >> 6;
    const int lane = (int)(c & 63);
    const int g = (int)(tile / KT), kt = (int)(tile % KT);
    int r;
    if (mode == 1) {
      const int half = R >> 1;
      r = (g & 1) * half + (g >> 1) * 16 + (lane & 15);
    } else {
      r = g * 16 + (lane & 15);
    }
    const int k8 = kt * 4 + (lane >> 4);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (r < R) v = src[(size_t)r * (K >> 3) + k8];
    dst[c] = v;
  }
}

__global__ void frag_to_rows_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int R, int K,
                                    long total_chunks) {
  const int KT = K >> 5;
  for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < total_chunks;
       c += (long)gridDim.x * blockDim.x) {
    const int r = (int)(c / (K >> 3)), k8 = (int)(c % (K >> 3));
    dst[c] = src[frag_chunk(r, k8, KT)];
  }
}

// QKV "rotation-paired" row order (used by the fused RoPE epilogue and by rope_store_kernel with qkv_perm = 1):
// inside every q / k head, 16-row group j holds dims [8j .. 8j+7] followed by [hd/2 + 8j .. hd/2 + 8j + 7], so the two
// members of a neox rotation pair (d, d + hd/2) live in one MFMA accumulator tile (rows i and i + 8).  V rows keep
// their natural order.  src is the reference layout [q heads | k heads | v heads] x hd rows (ssd/layers/linear.py:125-162).
__global__ void shuffle_qkv_to_frag_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int nh, int nkv,
                                           int hd, int K, long total_chunks) {
  const int KT = K >> 5;
  const int gph = hd >> 4, half = hd >> 1;
  const int qk_groups = (nh + nkv) * gph;
  for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < total_chunks;
       c += (long)gridDim.x * blockDim.x) {
    const long tile = c >> 6;
    const int lane = (int)(c & 63);
    const int g = (int)(tile / KT), kt = (int)(tile % KT);
    const int i = lane & 15;
    int r;
    if (g < qk_groups) {
      const int head = g / gph, j = g % gph;
      r = head * hd + (i < 8 ? 8 * j + i : half + 8 * j + (i - 8));
    } else {
      r = g * 16 + i;
    }
    const int k8 = kt * 4 + (lane >> 4);
    dst[c] = src[(size_t)r * (K >> 3) + k8];
  }
}

extern "C" int ssd_rows_to_frag_qkv(const void* src_rows, void* dst_frag, int nh, int nkv, int hd, int K, void* stream) {
  if (nh <= 0 || nkv <= 0 || (hd & 15) || K <= 0 || (K & 31)) return SSD_ERR_SHAPE;
  const long groups = (long)(nh + 2 * nkv) * (hd >> 4);
  const long total = groups * (K >> 5) * 64;
  const int threads = 256;
  long blocks = (total + threads - 1) / threads;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(shuffle_qkv_to_frag_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream,
                     (const u32x4_t*)src_rows, (u32x4_t*)dst_frag, nh, nkv, hd, K, total);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_rows_to_frag(const void* src_rows, void* dst_frag, int R, int K, int mode, void* stream) {
  if (R <= 0 || K <= 0 || (K & 31)) return SSD_ERR_SHAPE;
  if (mode == 1 && ((R & 31) != 0)) return SSD_ERR_SHAPE;
  const long groups = (R + 15) / 16;
  const long total = groups * (K >> 5) * 64;
  const int threads = 256;
  long blocks = (total + threads - 1) / threads;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(shuffle_rows_to_frag_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream,
                     (const u32x4_t*)src_rows, (u32x4_t*)dst_frag, R, K, mode, total);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_frag_to_rows(const void* src_frag, void* dst_rows, int R, int K, void* stream) {
  if (R <= 0 || K <= 0 || (K & 31)) return SSD_ERR_SHAPE;
  const long total = (long)R * (K >> 3);
  const int threads = 256;
  long blocks = (total + threads - 1) / threads;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(frag_to_rows_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream,
                     (const u32x4_t*)src_frag, (u32x4_t*)dst_rows, R, K, total);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
