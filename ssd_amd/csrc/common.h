// Shared device helpers for the gfx950 (CDNA4) kernels of libssdhip.
// Everything here is wave64 / MFMA-16x16x32 specific; there is no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits (row-major tensors are passed as plain pointers)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
constexpr int SSD_MAX_DEVICES = 16;  // per-device host-side caches (function attributes, residency checks): indexed by the HIP device ordinal

// The public C ABI: every translation unit sees the declarations its extern "C" definitions must match (a mismatch is a compile
// error: conflicting types), and the error codes / ABI version exist in ONE place.
#include "ssd_hip.h"
#include "ssd_hip_tune.h"

__device__ __forceinline__ float bf2f(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// fp32 -> bf16, round-to-nearest-even (the rounding every torch bf16 store performs).
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

// The same conversion on gfx950's converter (v_cvt_pk_bf16_f32: one instruction for two values instead of ~12): round-to-nearest-even,
// bit-identical to f2bf on every non-NaN input (checked exhaustively over all 2^32 patterns on the MI355X by ssd_selftest_bf16_cvt,
// tests/test_hip_ops.py); used where a kernel converts tens of values per thread (csrc/tree_segment.hip).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2_hw(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ float round_bf_hw(float f) { return bf2f(pack_bf2_hw(f, 0.f) & 0xffffu); }

// Round an fp32 value through bf16 (models "store bf16, reload" between two reference kernels).
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// "Fragment-major" (XF / WF) layout shared by activations and weights.
// A logical [R][K] bf16 matrix (K % 32 == 0) is cut into 16-row x 32-col tiles of 1 KiB; inside a
// tile the 64 16-byte chunks are stored in MFMA-16x16x32 lane order:
//   lane l = (r & 15) + 16 * ((k & 31) >> 3) holds row r, columns (k & ~7) .. +7.
// Tiles are ordered [row_tile][k_tile], so one wave streaming along K for a fixed row tile reads
// contiguous memory with fully coalesced 1 KiB wave loads that ARE the MFMA operand.
// Returns the chunk (16-byte unit) index of (row r, 8-column group k8 = k / 8).
__device__ __forceinline__ size_t frag_chunk(int r, int k8, int KT) {
  return ((size_t)((r >> 4) * KT + (k8 >> 2)) << 6) + (size_t)((r & 15) + ((k8 & 3) << 4));
}

__device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Threads per row of the RMSNorm kernels (norm.hip rmsnorm_kernel and comm.hip allreduce_add_rmsnorm_kernel MUST agree: the
// chunk -> thread mapping fixes the fp32 summation order, and the fused all-reduce + norm is bit-identical to the unfused
// pair).  A row is a latency chain (load, reduce, scale, store): hidden sizes >= 4096 get 1024 threads, i.e. at most two
// 8-element chunks per thread instead of four to eight (profiles/r02_norm_probe.txt).
static inline int ssd_norm_threads(int H) { return H >= 4096 ? 1024 : 256; }

// Default decomposition of the skinny (M <= 16 token rows) weight-streaming GEMMs, from the MI355X sweep in
// profiles/r01_gemm_tune_sweep.txt (python profiles/tune_gemm.py): groups = N / 16 row groups, KT = K / 32 k-tiles.
//   nt    row groups per workgroup (x operand amortised over nt weight tiles; fewer, fatter workgroups)
//   waves K-split inside the workgroup
//   tpw   consecutive tiles per workgroup (gemm.hip only)
// Consecutive tiles per workgroup for `ntiles` tiles: the candidate that wastes the least of the last round of 256
// workgroups (a launch whose workgroup count lands just above a multiple of the CU count pays a nearly empty round:
// Qwen3-32B gate_up at 3 tiles -> 267 workgroups ran 117 us, at 4 -> 200 workgroups 85 us; profiles/r02_tune_qwen.txt),
// larger runs winning ties.
static inline int ssd_pick_tpw(int ntiles, const int* cand, int ncand) {
  int best = 1;
  double best_eff = -1.0;
  for (int i = 0; i < ncand; ++i) {
    const int t = cand[i];
    const int blocks = (ntiles + t - 1) / t;
    if (blocks < 192 && t > 1) continue;                     // never starve the chip for the sake of a full round
    const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = t; }
  }
  return best;
}

static inline void ssd_pick_skinny_cfg(int groups, int KT, bool silu_pairs, int* nt, int* waves, int* tpw) {
  int n = 1, w = 8, t = 1;
  if (silu_pairs) {                       // gate_up: row groups come in (gate, up) pairs -> nt even
    n = (groups >= 1792 && groups % 4 == 0) ? 4 : 2;
    if (n == 4 && KT > 128) {             // consecutive tiles per workgroup
      const int cand[] = {4, 3, 2, 1};
      t = ssd_pick_tpw(groups / 4, cand, 4);
    }
  } else if (groups >= 4096) {            // LM heads
    n = (groups % 2 == 0) ? 2 : 1;
    if (KT >= 128) {
      const int cand[] = {8, 4, 2, 1};
      t = ssd_pick_tpw(groups / n, cand, 4);
      w = t > 1 ? 16 : 8;
    }
  } else if (groups >= 1000 && KT >= 256 && groups % 2 == 0) {   // vocabulary shards
    n = 2; w = 16; t = 2;
  } else if (groups >= 640 && KT >= 256 && groups % 4 == 0) {    // 70B-class qkv
    n = 4;
  } else if (groups >= 512 && KT >= 512 && groups % 2 == 0) {    // 70B-class down_proj: few row groups, very long K
    n = 2; w = 8;
  } else if (groups >= 512 && KT >= 256 && groups % 2 == 0) {    // 70B-class o_proj
    n = 2; w = 8;
  } else if (groups > 256 && groups < 512 && KT >= 256) {
    // between one and two workgroups per CU over a long K (Qwen3-32B o_proj / down_proj: 320 row groups): 4-wave workgroups of the
    // single-buffered kernel -- measured at M = 8 (profiles/r06_qwen32b_tune.txt): down_proj [5120 x 25600] 49.3 us against 53-57 with
    // 8 waves, o_proj [5120 x 8192] 17.8 against 19.5
    n = 1; w = 4;
  }
  // fewer workgroups than CUs (tensor-parallel shards of qkv, the drafts' qkv): a CU's pull from HBM is bounded by the bytes
  // it has in flight (8 waves x 8 KiB ~ 30 GB/s), so an under-filled launch gets 16 waves per workgroup
  // (profiles/r03_tp_shard_per_kind.txt: 70B qkv at TP = 4, 160 workgroups)
  if (!silu_pairs && groups / n < 256 && KT >= 64) w = 16;
  while (w > 1 && KT / w < 2) w >>= 1;    // every wave needs a couple of k-tiles
  *nt = n; *waves = w; *tpw = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// In-kernel timeline (profiling builds only: `make trace` -> _lib/libssdhip_trace.so with -DSSD_KTRACE; the product library
// contains none of it).  KTRACE(slot, i): thread 0 of every workgroup stores the chip-wide 100 MHz clock (s_memrealtime) into
// buf[slot][workgroup][i]; a later launch of the same slot overwrites an earlier one (all layers are alike: the last layer's
// launch is what a probe reads).  profiles/ktrace_probe.py turns the marks into a per-kernel ramp profile: dispatch skew,
// time to the first weight tile, time after the last, boundary to the next kernel.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef SSD_KTRACE
#define KT_MARKS 12
#define KT_MAXWG 4096
#define KT_SLOTS 32
static __device__ unsigned long long* kt_dev_buf;       // one copy per translation unit (no -fgpu-rdc): ssd_ktrace_set_<tu>
__device__ __forceinline__ void kt_mark(int slot, int i) {
  if (threadIdx.x == 0 && kt_dev_buf) {
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (wg < KT_MAXWG) kt_dev_buf[((size_t)slot * KT_MAXWG + wg) * KT_MARKS + i] = wall_clock64();
  }
}
#define KTRACE(slot, i) kt_mark((slot), (i))
#define KT_DEFINE_SETTER(tu)                                                                                         \
  extern "C" int ssd_ktrace_set_##tu(void* p) {                                                                      \
    return hipMemcpyToSymbol(HIP_SYMBOL(kt_dev_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;                          \
  }
#else
#define KTRACE(slot, i) ((void)0)
#define KT_DEFINE_SETTER(tu)
#endif
