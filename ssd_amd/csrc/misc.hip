// Small device-side state kernels that keep the draft->verify loop free of host round trips.
#include "common.h"


extern "C" int ssd_abi_version(void) { return SSD_HIP_ABI_VERSION; }

// Diagnostic: the hardware fp32 -> bf16 conversion (common.h pack_bf2_hw) against the integer form (f2bf) over ALL 2^32 fp32 bit
// patterns.  counts[0] = mismatches on non-NaN inputs (must be 0), counts[1] = NaN inputs whose result is not a NaN (must be 0).
__global__ void bf16_cvt_selftest_kernel(unsigned long long* counts) {
  unsigned long long bad = 0, badnan = 0;
  const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) << 12;
  for (uint32_t i = 0; i < 4096u; ++i) {
    const uint32_t u = base + i;
    const float f = __uint_as_float(u);
    const uint32_t sw = f2bf(f), hw = pack_bf2_hw(f, 0.f) & 0xffffu, hw2 = pack_bf2_hw(0.f, f) >> 16;
    if ((u & 0x7fffffffu) > 0x7f800000u) badnan += ((hw & 0x7fffu) <= 0x7f80u) || ((hw2 & 0x7fffu) <= 0x7f80u);
    else bad += (sw != hw) || (sw != hw2);
  }
  if (bad) atomicAdd(counts, bad);
  if (badnan) atomicAdd(counts + 1, badnan);
}
extern "C" int ssd_selftest_bf16_cvt(void* counts2, void* stream) {
  if (!counts2) return SSD_ERR_ARG;
  hipLaunchKernelGGL(bf16_cvt_selftest_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)counts2);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// After one single-token draft forward + argmax (`next[b]`): record the token as speculation step+1 and
// turn the static decode inputs into the inputs of the next draft step -- what the reference does on the
// host between graph replays (ssd/engine/speculator_sync.py:47-66 + ssd/engine/helpers/runner_helpers.py:59-75,
// each iteration a `.tolist()` sync).  Here the K+1 draft forwards are enqueued back to back.
__global__ void draft_advance_kernel(const int64_t* __restrict__ next, int64_t* __restrict__ input_ids,
                                     int64_t* __restrict__ positions, int32_t* __restrict__ slots,
                                     int32_t* __restrict__ ctx, const int32_t* __restrict__ block_tables,
                                     int max_blocks, int bs, int64_t* __restrict__ spec, int K, int32_t* step, int B) {
  const int b = threadIdx.x;
  const int s = *step;
  __syncthreads();
  if (b < B) {
    const int64_t tok = next[b];
    if (s + 1 <= K) spec[(size_t)b * (K + 1) + s + 1] = tok;
    input_ids[b] = tok;
    const long pos = positions[b] + 1;
    positions[b] = pos;
    ctx[b] += 1;
    const int blk = block_tables[(size_t)b * max_blocks + (int)(pos / bs)];
    slots[b] = blk >= 0 ? blk * bs + (int)(pos % bs) : -1;
  }
  if (b == 0) *step = s + 1;
}

extern "C" int ssd_draft_advance(const int64_t* next, int64_t* input_ids, int64_t* positions, int32_t* slots,
                                 int32_t* context_lens, const int32_t* block_tables, int max_blocks, int block_size,
                                 int64_t* spec, int K, int32_t* step, int B, void* stream) {
  if (B <= 0 || B > 1024) return SSD_ERR_SHAPE;
  const int threads = ((B + 63) / 64) * 64;
  hipLaunchKernelGGL(draft_advance_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, next, input_ids, positions,
                     slots, context_lens, block_tables, max_blocks, block_size, spec, K, step, B);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Speculation-cache lookup on the draft device (reference ssd/engine/draft_runner.py:215-252: request keys compared
// against the tensor-backed tree cache).  The cache of a round is indexed by (sequence id of row b, glue position j of branch
// i, fork token of branch i): entry c = b * W + i.  out_idx[r] = the FIRST entry equal to request key r = (seq, j, token), or
// -1.  One workgroup per request; nothing but this [B] int32 vector crosses to the host to decide reply-from-cache vs. JIT.
__global__ void cache_lookup_kernel(const int64_t* __restrict__ req, const int64_t* __restrict__ cache_seq,
                                    const int32_t* __restrict__ cache_j, const int64_t* __restrict__ cache_forks, int Bc, int W,
                                    int32_t* __restrict__ out_idx) {
  __shared__ int best;
  const int r = blockIdx.x;
  if (threadIdx.x == 0) best = 0x7fffffff;
  __syncthreads();
  const int64_t k_seq = req[r * 3], k_j = req[r * 3 + 1], k_tok = req[r * 3 + 2];
  for (int c = threadIdx.x; c < Bc * W; c += blockDim.x)
    if (cache_seq[c / W] == k_seq && (int64_t)cache_j[c] == k_j && cache_forks[c] == k_tok) atomicMin(&best, c);
  __syncthreads();
  if (threadIdx.x == 0) out_idx[r] = best == 0x7fffffff ? -1 : best;
}

extern "C" int ssd_cache_lookup(const int64_t* req_keys, const int64_t* cache_seq, const int32_t* cache_j,
                                const int64_t* cache_forks, int B, int Bc, int W, int32_t* out_idx, void* stream) {
  if (B <= 0 || Bc <= 0 || W <= 0) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(cache_lookup_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, req_keys, cache_seq, cache_j, cache_forks, Bc,
                     W, out_idx);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Keep the draft's logits of chain step `*step` as logits_q[b][step] (SpeculatorSync collects them with torch.stack,
// reference ssd/engine/speculator_sync.py:58,67); the step index lives on the device so the launch is graph-replayable.
__global__ void store_step_rows_kernel(const u32x4_t* __restrict__ src, long src_ld8, u32x4_t* __restrict__ dst, int V8, int K,
                                       const int32_t* __restrict__ step) {
  const int b = blockIdx.x, s = *step;
  if (s >= K) return;
  for (int c = threadIdx.x; c < V8; c += blockDim.x) dst[((size_t)b * K + s) * V8 + c] = src[(size_t)b * src_ld8 + c];
}

extern "C" int ssd_store_step_rows(const void* src_rows, long src_ld, void* dst, int B, int V, int K, const int32_t* step,
                                   void* stream) {
  if (B <= 0 || V <= 0 || (V & 7) || (src_ld & 7) || K <= 0) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(store_step_rows_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t*)src_rows, src_ld / 8,
                     (u32x4_t*)dst, V / 8, K, step);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}


// hipGraph capture for a host that is not PyTorch (the reference captures its decode / verify / glue / tree steps with torch.cuda.CUDAGraph,
// ssd/engine/helpers/cudagraph_helpers.py:20-120; this repo's engine does the same through torch on ROCm, where that IS hipGraph): every
// entry point of this library only enqueues on `stream`, so whatever sequence of calls lies between begin and end becomes one replayable
// graph.  ssd_graph_end returns an executable graph handle; launch it as often as wanted, destroy it once.
extern "C" int ssd_graph_begin(void* stream) {
  if (!stream) return SSD_ERR_ARG;                          // the legacy null stream cannot be captured
  return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
extern "C" int ssd_graph_end(void* stream, void** out_exec) {
  if (!stream || !out_exec) return SSD_ERR_ARG;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return SSD_ERR_LAUNCH;
  hipGraphExec_t e = nullptr;
  const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (rc != hipSuccess) return SSD_ERR_LAUNCH;
  *out_exec = (void*)e;
  return SSD_OK;
}
extern "C" int ssd_graph_launch(void* exec, void* stream) {
  if (!exec) return SSD_ERR_ARG;
  return hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
extern "C" int ssd_graph_destroy(void* exec) {
  if (!exec) return SSD_ERR_ARG;
  return hipGraphExecDestroy((hipGraphExec_t)exec) == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
