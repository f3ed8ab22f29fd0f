// Greedy sampling, accept/reject and tree-fork selection kernels.
//  * ssd_argmax_rows    replaces Sampler.forward at temperature 0 (reference ssd/layers/sampler.py:15-20)
//                       and `logits_p.argmax(dim=-1)` (reference ssd/utils/verify.py:34).  Ties resolve to
//                       the lowest index (what torch's CPU argmax returns).
//  * ssd_verify_greedy  replaces the greedy branch of verify() (reference ssd/utils/verify.py:28-48):
//                       first mismatch between draft tokens and target argmax via wave ballot +
//                       count-trailing-zeros, recovery token gathered at that position.
//  * ssd_fork_topf      replaces get_forked_recovery_tokens_from_logits (reference
//                       ssd/utils/async_helpers/async_spec_helpers.py:26-78): top-F per glue position with
//                       the draft's own next token excluded for positions 0..K-1.
#include "common.h"

struct ArgBest { float v; int i; };

__device__ __forceinline__ ArgBest better(ArgBest a, ArgBest b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

// Block-wide argmax of one bf16 row with up to NEX excluded indices.
template <int THREADS>
__device__ ArgBest block_row_argmax(const bf16_t* __restrict__ row, int V, const int* excl, int nex, ArgBest* sm) {
  ArgBest best = {-INFINITY, 0x7fffffff};
  const int V8 = V >> 3;
  for (int c = threadIdx.x; c < V8; c += THREADS) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = bf2f(v[j] & 0xffffu), hi = bf2f(v[j] >> 16);
      const int i0 = c * 8 + 2 * j;
      bool e0 = false, e1 = false;
      for (int e = 0; e < nex; ++e) { e0 |= (excl[e] == i0); e1 |= (excl[e] == i0 + 1); }
      if (!e0 && (lo > best.v || (lo == best.v && i0 < best.i))) best = {lo, i0};
      if (!e1 && (hi > best.v || (hi == best.v && i0 + 1 < best.i))) best = {hi, i0 + 1};
    }
  }
  for (int i = V8 * 8 + threadIdx.x; i < V; i += THREADS) {  // tail (V % 8)
    bool ex = false;
    for (int e = 0; e < nex; ++e) ex |= (excl[e] == i);
    const float x = bf2f(row[i]);
    if (!ex && (x > best.v || (x == best.v && i < best.i))) best = {x, i};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgBest other = {__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
    best = better(best, other);
  }
  __syncthreads();  // protect sm reuse across calls
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
  __syncthreads();
  ArgBest r = sm[0];
  for (int w = 1; w < THREADS / 64; ++w) r = better(r, sm[w]);
  if (r.i == 0x7fffffff) r.i = 0;   // nothing comparable in the row (all NaN / -inf / excluded): never emit the sentinel as a token id
  return r;
}

constexpr int ARG_THREADS = 1024;

__global__ void __launch_bounds__(ARG_THREADS)
argmax_rows_kernel(const bf16_t* __restrict__ logits, long ld, int V, int64_t* __restrict__ out, int64_t* __restrict__ out2,
                   float* __restrict__ out_val, long idx_offset) {
  __shared__ ArgBest sm[ARG_THREADS / 64];
  const ArgBest r = block_row_argmax<ARG_THREADS>(logits + (size_t)blockIdx.x * ld, V, nullptr, 0, sm);
  if (threadIdx.x == 0) {
    out[blockIdx.x] = r.i + idx_offset;
    if (out2) out2[blockIdx.x] = r.i + idx_offset;
    if (out_val) out_val[blockIdx.x] = r.v;
  }
}

// Vocab-parallel argmax merge: vals [tp][stride >= T] floats, idxs [tp][stride_idx >= T] int64 (per-rank local maxima with GLOBAL indices) -> out[T].
// Equivalent to argmax over the concatenated logits (ParallelLMHead gather + cat, reference
// ssd/layers/embed_head.py:88-92): larger value wins, lowest global index on ties.
__global__ void argmax_merge_kernel(const float* __restrict__ vals, const int64_t* __restrict__ idxs, int tp, int T,
                                    long stride, long stride_idx, int64_t* __restrict__ out, int64_t* __restrict__ out2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float bv = vals[t];
  int64_t bi = idxs[t];
  for (int r = 1; r < tp; ++r) {
    const float v = vals[(size_t)r * stride + t];
    const int64_t i = idxs[(size_t)r * stride_idx + t];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  out[t] = bi;
  if (out2) out2[t] = bi;
}

extern "C" int ssd_argmax_merge(const float* vals, const int64_t* idxs, int tp, int T, long stride, long stride_idx,
                                int64_t* out, int64_t* out2, void* stream) {
  if (tp <= 0 || T <= 0 || stride < T || stride_idx < T) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_merge_kernel, dim3((T + 63) / 64), dim3(64), 0, (hipStream_t)stream, vals, idxs, tp, T, stride,
                     stride_idx, out, out2);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_argmax_rows(const void* logits, long ld, int T, int V, int64_t* out, int64_t* out2, void* stream) {
  if (T <= 0 || V <= 0 || (ld & 7)) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(T), dim3(ARG_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                     out, out2, (float*)nullptr, 0L);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_argmax_rows_val(const void* logits, long ld, int T, int V, long idx_offset, int64_t* out_idx,
                                   float* out_val, void* stream) {
  if (T <= 0 || V <= 0 || (ld & 7)) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(T), dim3(ARG_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                     out_idx, (int64_t*)nullptr, out_val, idx_offset);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// One wave per sequence.  speculations[b] = [recovery, x_1 .. x_K]; preds[b][i] = argmax of target row i.
// accept_len[b] = number of accepted draft tokens n (0..K); recovery[b] = preds[b][n];
// out_suffix[b] (optional, [K+2] per seq) = [n+1, recovery_prev, x_1..x_n ...] packed for one D2H copy.
__global__ void verify_greedy_kernel(const int64_t* __restrict__ preds, const int64_t* __restrict__ spec, int K,
                                     int32_t* __restrict__ accept_len, int64_t* __restrict__ recovery,
                                     int64_t* __restrict__ packed) {
  const int b = blockIdx.x, lane = threadIdx.x;
  bool mismatch = false;
  if (lane < K) mismatch = spec[(size_t)b * (K + 1) + lane + 1] != preds[(size_t)b * (K + 1) + lane];
  const unsigned long long mask = __ballot(mismatch);
  const int n = mask ? (int)__builtin_ctzll(mask) : K;
  if (lane == 0) {
    accept_len[b] = n;
    recovery[b] = preds[(size_t)b * (K + 1) + n];
  }
  if (packed) {  // one row per sequence for a single D2H copy: [n, recovery, spec_0 .. spec_K]
    int64_t* row = packed + (size_t)b * (K + 3);
    if (lane == 0) { row[0] = n; row[1] = preds[(size_t)b * (K + 1) + n]; }
    if (lane <= K) row[2 + lane] = spec[(size_t)b * (K + 1) + lane];
  }
}

extern "C" int ssd_verify_greedy(const int64_t* preds, const int64_t* speculations, int B, int K, int32_t* accept_len,
                                 int64_t* recovery, int64_t* packed, void* stream) {
  if (B <= 0 || K < 0 || K > 62) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(verify_greedy_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, preds, speculations, K, accept_len,
                     recovery, packed);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// logits [B*(K+1)][ld]; returned [B][K+1] = (rec, x_1..x_K); counts [B][K+1] fan-out per position
// (hit or miss list already selected per sequence); out [B][mq] flattened in position order.
constexpr int FORK_THREADS = 1024;
constexpr int FORK_MAXF = 16;
__global__ void __launch_bounds__(FORK_THREADS)
fork_topf_kernel(const bf16_t* __restrict__ logits, long ld, int V, const int64_t* __restrict__ returned,
                 const int32_t* __restrict__ counts, const int32_t* __restrict__ offsets, int K, int mq,
                 int64_t* __restrict__ out) {
  __shared__ ArgBest sm[FORK_THREADS / 64];
  __shared__ int excl[FORK_MAXF + 1];
  const int b = blockIdx.x / (K + 1), j = blockIdx.x % (K + 1);
  const int cnt = min(counts[b * (K + 1) + j], FORK_MAXF - 1);   // excl[] holds x_{j+1} + the picks so far (Config validates)
  int nex = 0;
  if (threadIdx.x == 0 && j < K) excl[0] = (int)returned[(size_t)b * (K + 1) + j + 1];
  if (j < K) nex = 1;
  __syncthreads();
  for (int f = 0; f < cnt; ++f) {
    const ArgBest r = block_row_argmax<FORK_THREADS>(logits + (size_t)blockIdx.x * ld, V, excl, nex, sm);
    if (threadIdx.x == 0) {
      out[(size_t)b * mq + offsets[b * (K + 1) + j] + f] = r.i;
      excl[nex] = r.i;
    }
    ++nex;
    __syncthreads();
  }
}

extern "C" int ssd_fork_topf(const void* logits, long ld, int V, const int64_t* returned_tokens, const int32_t* counts,
                             const int32_t* offsets, int B, int K, int mq, int64_t* out, void* stream) {
  if (B <= 0 || K < 0 || V <= 0 || (ld & 7)) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(fork_topf_kernel, dim3(B * (K + 1)), dim3(FORK_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)logits, ld, V, returned_tokens, counts, offsets, K, mq, out);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// The same fork, spread over the chip (round 5).  ssd_fork_topf puts ONE workgroup on each of the B * (K + 1) glue rows and walks the
// 128 K-logit row F times: 8 workgroups, 84.7 us per launch on the draft's critical path (profiles/r04_c4_kernel_stats.csv).  Here a row
// is cut into S slices of <= 4096 logits; workgroup (slice, row) loads its slice ONCE (16 values per thread, one round trip), picks the
// slice's top-F with the same comparison (larger value, then lower index; x_{j+1} excluded for rows j < K) and leaves F candidates;
// a second, one-wave-per-row launch picks the row's top-F from the S * F candidates.  The top-F of a union is the top-F of the
// per-slice top-Fs, and ties resolve by index in both stages: bit-equal to ssd_fork_topf.  V % 8 == 0.
constexpr int FS_THREADS = 256;
constexpr int FS_PER = 2;                        // 16-byte chunks per thread
constexpr int FS_SLICE = FS_THREADS * FS_PER * 8;

__global__ void __launch_bounds__(FS_THREADS)
fork_slice_kernel(const bf16_t* __restrict__ logits, long ld, int V, const int64_t* __restrict__ returned, const int32_t* __restrict__ counts,
                  int K, int S, int chunks_per_slice, float* __restrict__ cand_val, int* __restrict__ cand_idx) {
  __shared__ ArgBest sm[FS_THREADS / 64];
  const int row = blockIdx.y, sl = blockIdx.x, b = row / (K + 1), j = row % (K + 1);
  const int cnt = min(counts[row], FORK_MAXF - 1);
  const int excl = j < K ? (int)returned[(size_t)b * (K + 1) + j + 1] : -1;
  const int V8 = V >> 3, c_end = min(V8, (sl + 1) * chunks_per_slice);
  float v[FS_PER * 8];
  int base[FS_PER];
#pragma unroll
  for (int u = 0; u < FS_PER; ++u) {
    const int c = sl * chunks_per_slice + u * FS_THREADS + (int)threadIdx.x;
    base[u] = c * 8;
    u32x4_t q = {0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};      // -inf: a chunk past the slice never wins
    if (c < c_end) q = *reinterpret_cast<const u32x4_t*>(logits + (size_t)row * ld + (size_t)c * 8);
    else base[u] = 0x7fffff00;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[u * 8 + 2 * e] = bf2f(q[e] & 0xffffu); v[u * 8 + 2 * e + 1] = bf2f(q[e] >> 16); }
  }
  unsigned taken = 0;                             // elements of this thread that are excluded or already picked
#pragma unroll
  for (int u = 0; u < FS_PER; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (base[u] + e == excl || base[u] == 0x7fffff00) taken |= 1u << (u * 8 + e);
  for (int f = 0; f < cnt; ++f) {
    ArgBest best = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int u = 0; u < FS_PER; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (!((taken >> (u * 8 + e)) & 1u)) best = better(best, ArgBest{v[u * 8 + e], base[u] + e});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = better(best, ArgBest{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)});
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
    __syncthreads();
    ArgBest r = sm[0];
#pragma unroll
    for (int w = 1; w < FS_THREADS / 64; ++w) r = better(r, sm[w]);
#pragma unroll
    for (int u = 0; u < FS_PER; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (base[u] + e == r.i) taken |= 1u << (u * 8 + e);
    if (threadIdx.x == 0) {
      const size_t o = ((size_t)row * S + sl) * FORK_MAXF + f;
      cand_val[o] = r.v;
      cand_idx[o] = r.i;
    }
  }
}

__global__ void __launch_bounds__(64)
fork_merge_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx, const int32_t* __restrict__ counts,
                  const int32_t* __restrict__ offsets, int K, int S, int mq, int64_t* __restrict__ out) {
  const int row = blockIdx.x, b = row / (K + 1), lane = threadIdx.x;
  const int cnt = min(counts[row], FORK_MAXF - 1);
  const int n = S * cnt;                          // candidate (slice s, rank f) at (row * S + s) * FORK_MAXF + f
  constexpr int PER = 12;                         // 64 lanes x 12 >= 48 slices x 15
  ArgBest c[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = u * 64 + lane;
    c[u] = ArgBest{-INFINITY, 0x7fffffff};
    if (q < n) {
      const size_t o = ((size_t)row * S + q / cnt) * FORK_MAXF + q % cnt;
      c[u] = ArgBest{cand_val[o], cand_idx[o]};
    }
  }
  unsigned taken = 0;
  for (int f = 0; f < cnt; ++f) {
    ArgBest best = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (!((taken >> u) & 1u)) best = better(best, c[u]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = better(best, ArgBest{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)});
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (c[u].i == best.i) taken |= 1u << u;
    if (lane == 0) out[(size_t)b * mq + offsets[row] + f] = best.i == 0x7fffffff ? 0 : best.i;   // (as ssd_fork_topf: never the sentinel)
  }
}

// Also the library's predicate for "ssd_fork_topf_split takes this vocabulary" (ADVICE r5): SSD_ERR_SHAPE where the split form would
// refuse (V not a multiple of 8, or more than 48 slices = V > 196608) -- the caller then stays on ssd_fork_topf.
extern "C" int ssd_fork_topf_workspace_bytes(int V, int B, int K) {
  if (V <= 0 || B <= 0 || K < 0 || (V & 7)) return SSD_ERR_SHAPE;
  const int S = (V + FS_SLICE - 1) / FS_SLICE;
  if (S > 48) return SSD_ERR_SHAPE;
  return B * (K + 1) * S * FORK_MAXF * 8;
}

extern "C" int ssd_fork_topf_split(const void* logits, long ld, int V, const int64_t* returned_tokens, const int32_t* counts,
                                   const int32_t* offsets, int B, int K, int mq, void* workspace, int64_t* out, void* stream) {
  if (B <= 0 || K < 0 || V <= 0 || (V & 7) || (ld & 7)) return SSD_ERR_SHAPE;
  if (!logits || !returned_tokens || !counts || !offsets || !workspace || !out) return SSD_ERR_ARG;
  const int S = (V + FS_SLICE - 1) / FS_SLICE;
  if (S > 48) return SSD_ERR_SHAPE;              // fork_merge_kernel keeps S * F <= 768 candidates in one wave's registers
  const int rows = B * (K + 1);
  const int chunks = ((V >> 3) + S - 1) / S;      // <= FS_THREADS * FS_PER
  float* cv = (float*)workspace;
  int* ci = (int*)(cv + (size_t)rows * S * FORK_MAXF);
  hipLaunchKernelGGL(fork_slice_kernel, dim3(S, rows), dim3(FS_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                     returned_tokens, counts, K, S, chunks, cv, ci);
  hipLaunchKernelGGL(fork_merge_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, cv, ci, counts, offsets, K, S, mq, out);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Argmax from the LM-head GEMM's per-workgroup candidates (gemm.hip EPI_ROWS_ARGMAX: part_val / part_idx
// [row * part_stride + p], p < nparts) -- and, in the same launch, what the loop does with the token next:
//   plain    out[row] (+ out2[row], + out3[row * out3_stride]: the tree step's next input ids and its [T][K] token table)
//   verify   ssd_verify_greedy's accept / recovery / packed row of each sequence (reference ssd/utils/verify.py:28-48)
//   advance  ssd_draft_advance's "append token, bump position, recompute slot" of the chained draft forwards
// One wave per token row; (max value, lowest index) -- identical to an argmax over the stored bf16 logits.
// ---------------------------------------------------------------------------------------------------------------------
// All 512 threads of a workgroup scan the candidates of up to AP_ROWS rows: every thread issues ALL its loads of all rows
// before the first compare (a few thousand candidates per row are ONE memory round trip for the workgroup; a single wave
// walking them took 63 dependent round trips = 40 us), then wave shuffles + one LDS stage.  res[r] = (max, lowest index).
constexpr int AP_THREADS = 512;     // (256 VGPRs per lane: AP_ROWS x AP_UN candidates stay in registers)
constexpr int AP_ROWS = 4;          // rows per pass (registers: AP_ROWS x AP_UN candidates in flight per thread)
constexpr int AP_MAX_SEQ_ROWS = 16; // rows of one sequence in the verify tail (K + 1)
constexpr int AP_UN = 8;        // candidates per thread and row in flight (covers 4096 candidates per pass)

__device__ void block_parts_argmax(const float* __restrict__ part_val, const int* __restrict__ part_idx, int nparts,
                                   long part_stride, int row0, int nrows, ArgBest* res /* shared [AP_ROWS] */,
                                   ArgBest* sm /* shared [AP_THREADS / 64][AP_ROWS] */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  ArgBest best[AP_ROWS];
#pragma unroll
  for (int r = 0; r < AP_ROWS; ++r) best[r] = ArgBest{-INFINITY, 0x7fffffff};
  for (int base = 0; base < nparts; base += AP_THREADS * AP_UN) {
    float v[AP_ROWS][AP_UN];
    int ix[AP_ROWS][AP_UN];
#pragma unroll
    for (int r = 0; r < AP_ROWS; ++r) {
      if (r < nrows) {           // block-uniform
        const float* pv = part_val + (size_t)(row0 + r) * part_stride;
        const int* pi = part_idx + (size_t)(row0 + r) * part_stride;
#pragma unroll
        for (int u = 0; u < AP_UN; ++u) {
          const int p = base + u * AP_THREADS + tid;
          const int pc = p < nparts ? p : nparts - 1;          // clamped: an unconditional load (a duplicate never changes the result)
          v[r][u] = pv[pc];
          ix[r][u] = pi[pc];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < AP_ROWS; ++r) {
      if (r < nrows) {
#pragma unroll
        for (int u = 0; u < AP_UN; ++u) best[r] = better(best[r], ArgBest{v[r][u], ix[r][u]});
      }
    }
  }
#pragma unroll
  for (int r = 0; r < AP_ROWS; ++r) {
    if (r < nrows) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) best[r] = better(best[r], ArgBest{__shfl_xor(best[r].v, o, 64), __shfl_xor(best[r].i, o, 64)});
      if (lane == 0) sm[w * AP_ROWS + r] = best[r];
    }
  }
  __syncthreads();
  if (tid < nrows) {
    ArgBest b = sm[tid];
    for (int ww = 1; ww < AP_THREADS / 64; ++ww) b = better(b, sm[ww * AP_ROWS + tid]);
    if (b.i == 0x7fffffff) b.i = 0;       // nothing comparable in the row (all NaN / -inf): never emit the sentinel
    res[tid] = b;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(AP_THREADS)
argmax_parts_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int nparts, long part_stride, int T,
                    int rows_per_block, long idx_offset, int64_t* __restrict__ out, int64_t* __restrict__ out2,
                    int64_t* __restrict__ out3, long out3_stride, float* __restrict__ out_val) {
  __shared__ ArgBest res[AP_ROWS], sm[(AP_THREADS / 64) * AP_ROWS];
  const int row0 = blockIdx.x * rows_per_block;
  const int nrows = min(rows_per_block, T - row0);
  block_parts_argmax(part_val, part_idx, nparts, part_stride, row0, nrows, res, sm);
  if ((int)threadIdx.x < nrows) {
    const int row = row0 + threadIdx.x;
    const int64_t tok = (int64_t)res[threadIdx.x].i + idx_offset;
    if (out) out[row] = tok;
    if (out2) out2[row] = tok;
    if (out3) out3[(size_t)row * out3_stride] = tok;
    if (out_val) out_val[row] = res[threadIdx.x].v;
  }
}

extern "C" int ssd_argmax_parts(const float* part_val, const int32_t* part_idx, int nparts, long part_stride, int T, long idx_offset,
                                int64_t* out, int64_t* out2, int64_t* out3, long out3_stride, float* out_val, void* stream) {
  if (T <= 0 || nparts <= 0 || part_stride < nparts || !part_val || !part_idx) return SSD_ERR_SHAPE;
  const int rpb = T <= 4 ? T : (T <= 64 ? 4 : AP_ROWS);     // a few rows per workgroup: more workgroups = more loads in flight
  hipLaunchKernelGGL(argmax_parts_kernel, dim3((T + rpb - 1) / rpb), dim3(AP_THREADS), 0, (hipStream_t)stream, part_val, part_idx,
                     nparts, part_stride, T, rpb, idx_offset, out, out2, out3, out3_stride, out_val);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// one workgroup per sequence: its K + 1 <= 16 rows
__global__ void __launch_bounds__(AP_THREADS)
argmax_parts_verify_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int nparts, long part_stride,
                           const int64_t* __restrict__ spec, int K, int64_t* __restrict__ preds, int32_t* __restrict__ accept_len,
                           int64_t* __restrict__ recovery, int64_t* __restrict__ packed) {
  __shared__ ArgBest res[AP_MAX_SEQ_ROWS], sm[(AP_THREADS / 64) * AP_ROWS];
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int r0 = 0; r0 <= K; r0 += AP_ROWS)
    block_parts_argmax(part_val, part_idx, nparts, part_stride, b * (K + 1) + r0, min(AP_ROWS, K + 1 - r0), res + r0, sm);
  if (threadIdx.x >= 64) return;
  if (preds && lane <= K) preds[(size_t)b * (K + 1) + lane] = res[lane].i;
  bool mismatch = false;
  if (lane < K) mismatch = spec[(size_t)b * (K + 1) + lane + 1] != (int64_t)res[lane].i;
  const unsigned long long mask = __ballot(mismatch);
  const int n = mask ? (int)__builtin_ctzll(mask) : K;
  if (lane == 0) {
    accept_len[b] = n;
    recovery[b] = res[n].i;
  }
  if (packed) {
    int64_t* prow = packed + (size_t)b * (K + 3);
    if (lane == 0) { prow[0] = n; prow[1] = res[n].i; }
    if (lane <= K) prow[2 + lane] = spec[(size_t)b * (K + 1) + lane];
  }
}

extern "C" int ssd_argmax_parts_verify(const float* part_val, const int32_t* part_idx, int nparts, long part_stride,
                                       const int64_t* speculations, int B, int K, int64_t* preds, int32_t* accept_len,
                                       int64_t* recovery, int64_t* packed, void* stream) {
  if (B <= 0 || K < 0 || K > 15 || nparts <= 0 || part_stride < nparts || !part_val || !part_idx) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_parts_verify_kernel, dim3(B), dim3(AP_THREADS), 0, (hipStream_t)stream, part_val, part_idx, nparts,
                     part_stride, speculations, K, preds, accept_len, recovery, packed);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// ONE workgroup (the step counter is shared by all sequences) walks the B rows, AP_ROWS at a time
__global__ void __launch_bounds__(AP_THREADS)
argmax_parts_advance_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int nparts, long part_stride,
                            int64_t* __restrict__ next, int64_t* __restrict__ input_ids, int64_t* __restrict__ positions,
                            int32_t* __restrict__ slots, int32_t* __restrict__ ctx, const int32_t* __restrict__ block_tables,
                            int max_blocks, int bs, int64_t* __restrict__ spec, int K, int32_t* step, int B) {
  __shared__ ArgBest res[AP_ROWS], sm[(AP_THREADS / 64) * AP_ROWS];
  const int s = *step;
  for (int row0 = 0; row0 < B; row0 += AP_ROWS) {
    const int nrows = min(AP_ROWS, B - row0);
    block_parts_argmax(part_val, part_idx, nparts, part_stride, row0, nrows, res, sm);
    if ((int)threadIdx.x < nrows) {
      const int b = row0 + threadIdx.x;
      const int64_t tok = res[threadIdx.x].i;
      next[b] = tok;
      if (s + 1 <= K) spec[(size_t)b * (K + 1) + s + 1] = tok;
      input_ids[b] = tok;
      const long pos = positions[b] + 1;
      positions[b] = pos;
      ctx[b] += 1;
      const int blk = block_tables[(size_t)b * max_blocks + (int)(pos / bs)];
      slots[b] = blk >= 0 ? blk * bs + (int)(pos % bs) : -1;
    }
    __syncthreads();      // res / sm are reused by the next group of rows
  }
  if (threadIdx.x == 0) *step = s + 1;
}

extern "C" int ssd_argmax_parts_advance(const float* part_val, const int32_t* part_idx, int nparts, long part_stride, int64_t* next,
                                        int64_t* input_ids, int64_t* positions, int32_t* slots, int32_t* context_lens,
                                        const int32_t* block_tables, int max_blocks, int block_size, int64_t* spec, int K,
                                        int32_t* step, int B, void* stream) {
  if (B <= 0 || B > 1024 || nparts <= 0 || part_stride < nparts || !part_val || !part_idx) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_parts_advance_kernel, dim3(1), dim3(AP_THREADS), 0, (hipStream_t)stream, part_val, part_idx, nparts,
                     part_stride, next, input_ids, positions, slots, context_lens, block_tables, max_blocks, block_size, spec, K,
                     step, B);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
