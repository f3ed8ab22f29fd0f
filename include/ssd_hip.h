/*
 * libssdhip -- C ABI of the MI355X (gfx950) draft->verify hot path.
 *
 * The reference (tanishqkumar/ssd) has no FFI of its own: its hot ops are Python call sites into
 * third-party CUDA wheels (torch/cuBLAS, Inductor, Triton, sgl-kernel FA3, flashinfer).  Each entry point
 * below replaces one such call site; the citation names the reference file:line it stands in for.
 * INTEGRATION.md shows the ctypes stub a maintainer would add at that call site.
 *
 * Conventions
 *   - every function returns 0 (SSD_OK) or a negative error code; nothing throws, allocates, or
 *     synchronises; all work is enqueued on `stream` (a hipStream_t passed as void*), so every call is
 *     hipGraph-capturable.
 *   - all pointers are DEVICE pointers owned by the caller; bf16 tensors are raw 16-bit words.
 *   - "rows"  = row-major [M][K] bf16.
 *     "frag"  = fragment-major [ceil(M/16)][K/32][64 lanes][8] bf16: 16x32 tiles of 1 KiB stored in
 *               MFMA-16x16x32 lane order, lane = (row & 15) + 16 * ((col & 31) >> 3).  Weights are
 *               pre-shuffled into this layout once at load (ssd_rows_to_frag); activations are
 *               produced in it directly by the kernel that feeds a GEMM.
 *   - token ids / positions are int64, every other index is int32 (as the reference,
 *     ssd/engine/helpers/runner_helpers.py:99-106).
 *   - KV cache layout per layer and per K/V: [num_blocks][n_kv_heads][block_size][head_dim] bf16;
 *     slot = block_id * block_size + pos_in_block, slot -1 = do not store (reference semantics,
 *     ssd/layers/attention.py:23-25).
 */
#ifndef SSD_HIP_H
#define SSD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a signature or a semantic of this header changes; ssd_abi_version() returns the value the library was built
 * with, so a consumer compiled against another header can tell (tests/abi_consumer.c does). */
#define SSD_HIP_ABI_VERSION 2

#define SSD_OK 0
#define SSD_ERR_SHAPE (-1)
#define SSD_ERR_LAUNCH (-2)
#define SSD_ERR_ARG (-3)

/* GEMM epilogues */
#define SSD_EPI_ROWS 0      /* y rows bf16 [M][ldy]                                             */
#define SSD_EPI_SILU_FRAG 1 /* W row groups alternate gate/up; y = frag bf16 [M][N/2] = silu(g)*u */
#define SSD_EPI_ROWS_F32 2  /* y rows fp32 [M][ldy] (diagnostics: logits before the bf16 rounding) */

int ssd_abi_version(void);

/* Layout conversion.  mode 0: identity; mode 1: source rows are [gate ; up] halves and destination
 * 16-row groups alternate gate/up (MergedColumnParallelLinear, ssd/layers/linear.py:101-122). */
int ssd_rows_to_frag(const void* src_rows, void* dst_frag, int R, int K, int mode, void* stream);
int ssd_frag_to_rows(const void* src_frag, void* dst_rows, int R, int K, void* stream);
/* QKV weights [q heads | k heads | v heads] (QKVParallelLinear, ssd/layers/linear.py:125-162) -> fragment-major with the
 * "rotation-paired" row order: in every q/k head, 16-row group j = dims [8j..8j+7] ++ [hd/2+8j..hd/2+8j+7], so both members of
 * a neox RoPE pair share an MFMA accumulator tile.  GEMMs on such weights emit q/k in that order (qkv_perm = 1 below). */
int ssd_rows_to_frag_qkv(const void* src_rows, void* dst_frag, int nh, int nkv, int hd, int K, void* stream);

/* VocabParallelEmbedding.forward -- ssd/layers/embed_head.py:49-57 (rows outside
 * [vocab_start, vocab_start+vocab_count) produce zeros, the TP-masked form). */
int ssd_embedding(const int64_t* ids, const void* table_rows, void* out_rows, int T, int H, long vocab_start,
                  long vocab_count, void* stream);

/* RMSDNorm.norm_forward / add_norm_forward -- ssd/layers/layernorm.py:64-88 (as compiled: fp32 math,
 * res_out = bf16(x+res), y = bf16((x+res) * rsqrt(mean^2 + eps) * w)).  res_in/res_out/out_rows/out_frag
 * may be NULL; gather_rows (int32[T], optional) selects input rows (prefill last-token logits,
 * ssd/layers/embed_head.py:81-84). */
int ssd_rmsnorm(const void* x_rows, const void* res_in, void* res_out, const void* weight, float eps,
                void* out_rows, void* out_frag, const int32_t* gather_rows, int T, int H, void* stream);

/* ssd_rmsnorm whose x is the bf16 rounding of the sum of `splits` fp32 partial slabs [splits][slab_rows][H] written by
 * ssd_gemm_parts (the row-parallel GEMM that precedes every add_norm_forward, ssd/models/llama3.py:185-199). */
int ssd_rmsnorm_parts(const void* parts, int splits, int slab_rows, const void* res_in, void* res_out, const void* weight,
                      float eps, void* out_rows, void* out_frag, int T, int H, void* stream);

/* RMSHeadNorm.forward -- ssd/layers/layernorm.py:16-40 (Qwen3's q_norm / k_norm, ssd/models/qwen3.py:96-104) as a call of its own:
 * x_rows [T][heads][hd] normalised over hd per (token, head), weight bf16[hd].  The hot path runs it inside ssd_rope_store_kv
 * (q_norm_w / k_norm_w); this entry point is bit-identical to that fused form and exists for a binding at the module boundary. */
int ssd_head_rmsnorm(const void* x_rows, const void* weight, float eps, void* out_rows, int T, int heads, int hd, void* stream);

/* SiluAndMul.forward -- ssd/layers/activation.py:11-14 as a call of its own: x_rows [T][2 I] = [gate | up] ->
 * bf16(silu(gate) * up) as rows [T][I] and / or fragment-major (either may be NULL).  The hot path runs it as the gate_up GEMM's
 * epilogue (SSD_EPI_SILU_FRAG), same arithmetic. */
int ssd_silu_mul(const void* x_rows, void* out_rows, void* out_frag, int T, int I, void* stream);

/* The EAGLE-3 draft layer's QKV input, torch.cat([input_layernorm(token_embeddings), conditioning_feature_ln(features)], -1)
 * -- ssd/models/eagle3_draft_llama3.py:148-150 -- written directly as ONE fragment-major [T][2H] activation (each half is
 * RMSDNorm.norm_forward as in ssd_rmsnorm). */
int ssd_rmsnorm_pair(const void* x0_rows, const void* weight0, const void* x1_rows, const void* weight1, float eps,
                     void* out_frag, int T, int H, void* stream);

/* F.linear(x, W, b) -- ssd/layers/linear.py:65,98,196; ssd/layers/embed_head.py:88,95,111.
 * x_frag [M][K] frag, w_frag [N][K] frag, bias bf16[N] or NULL.  M <= 128 per call.
 * SSD_EPI_SILU_FRAG additionally fuses SiluAndMul.forward -- ssd/layers/activation.py:11-14. */
int ssd_gemm_wf(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                int epilogue, void* stream);
/* The same F.linear for matrices with too few 16-row groups to fill the chip (csrc/gemm_sk.hip; the 1B draft's o_proj /
 * down_proj): K is split across `splits` workgroups per row group, the last one to arrive reduces the fp32 partials in a
 * fixed order.  workspace >= (N/16)*splits KiB; counters >= N/16 uint32, zeroed once.  M <= 16, bf16 rows. */
int ssd_gemm_splitk(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                    int splits, int waves, void* workspace, void* counters, void* stream);

/* The same F.linear in its latency-optimal form for small matrices at M <= 16 (csrc/gemm_sk.hip gemm_sp_kernel): K is
 * split over `splits` workgroups per 16-row group, every wave keeps all its k-tiles in flight, and the partial sums are
 * NOT combined here: `parts` receives fp32 slabs [splits][M][N] which the consumer (ssd_gemm_fused_parts /
 * ssd_rmsnorm_parts) sums in slab order while forming x = bf16(sum) + residual -- the kernel boundary is the only
 * synchronisation.  parts == NULL: splits must be 1 and bf16 rows (+ bias) go to y.  ceil(K/32/splits/waves) <= 8. */
int ssd_gemm_parts(const void* x_frag, const void* w_frag, const void* bias, void* y, void* parts, int M, int N, int K,
                   int ldy, int splits, int waves, void* stream);

/* Prefill-chunk GEMM, 16 < M <= 128 (csrc/gemm_pf.hip): the reference's eager prefill F.linear calls
 * (ssd/engine/model_runner.py:602 -> ssd/layers/linear.py:65,98,196).  Same operands and epilogues (SSD_EPI_ROWS,
 * SSD_EPI_SILU_FRAG) as ssd_gemm_wf; the x tile of a k-step is shared by a workgroup through LDS and K is split
 * across workgroups into fp32 partials in `workspace` (>= ssd_gemm_pf_workspace_bytes), summed in a fixed order.
 * N % 128 == 0, K % 128 == 0; splits <= 0 picks the default.  epilogue 2 (partials only, y may be NULL, no bias): the fp32
 * partials [splits][M][N] in `workspace` ARE the output (splits = ssd_gemm_pf_workspace_bytes(M, N, K) / (4 M N)), to be
 * summed by ssd_rmsnorm_parts -- the add + RMSNorm that follows o_proj / down_proj -- instead of an epilogue launch. */
int ssd_gemm_pf_workspace_bytes(int M, int N, int K, int64_t* bytes);
int ssd_gemm_pf(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                int epilogue, void* workspace, int64_t workspace_bytes, int splits, void* stream);
/* Fused decode-layer GEMM for M <= 16 (csrc/gemm_fused.hip): [residual add + RMSNorm] -> F.linear ->
 * [RoPE + paged KV store | SiLU*mul | rows] in ONE launch; replaces add_norm_forward (layernorm.py:76-88) + F.linear
 * (linear.py:97-98) + RotaryEmbedding.forward (rotary_embedding.py:40-60) + store_kvcache (attention.py:10-41), or
 * + SiluAndMul (activation.py:11-14).  x comes either fragment-major (x_frag) or as row-major h (+ res_in) normalised on
 * the fly with norm_w / eps; then res_out (!= res_in) receives bf16(h + res_in).  epilogue: 0 rows, 1 SiLU*mul -> frag,
 * 3 RoPE + KV store (weights from ssd_rows_to_frag_qkv; q_out rows [M][nh*hd], K/V into the paged cache at `slots`).
 * nt / waves <= 0 pick the default decomposition. */
int ssd_gemm_fused(const void* x_frag, const void* h_rows, const void* res_in, void* res_out, const void* norm_w, float eps,
                   const void* w_frag, const void* bias, int M, int N, int K, int epilogue, void* y, int ldy,
                   const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache,
                   void* v_cache, int nh, int nkv, int hd, int block_size, int nt, int waves, void* stream);

/* ssd_gemm_fused with the norm prologue fed by the producer GEMM's fp32 partial slabs [splits][M][K] (ssd_gemm_parts)
 * instead of bf16 rows: h = bf16(sum over the slabs, in order), the value the reference's F.linear would have stored. */
int ssd_gemm_fused_parts(const void* h_parts, int splits, const void* res_in, void* res_out, const void* norm_w, float eps,
                         const void* w_frag, const void* bias, int M, int N, int K, int epilogue, void* y, int ldy,
                         const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache,
                         void* v_cache, int nh, int nkv, int hd, int block_size, int nt, int waves, void* stream);

/* (RMSHeadNorm q/k, Qwen3: ssd/models/qwen3.py:96-104) + RotaryEmbedding.forward
 * (ssd/layers/rotary_embedding.py:40-60) + store_kvcache (ssd/layers/attention.py:10-41).
 * qkv_rows [T][(nh+2nkv)*hd]; cos_sin fp32 [max_pos][hd] (cos || sin); q_norm_w/k_norm_w bf16[hd] or NULL. */
int ssd_rope_store_kv(const void* qkv_rows, const int64_t* positions, const float* cos_sin,
                      const int32_t* slot_mapping, void* q_out_rows, void* k_cache, void* v_cache,
                      const void* q_norm_w, const void* k_norm_w, float eps, int T, int nh, int nkv, int hd,
                      int block_size, int qkv_perm, void* stream);
/* Same, with the QKV rows given as `splits` fp32 split-K slabs parts[s][T][(nh+2nkv)*hd] of the prefill GEMM (ssd_gemm_pf with
 * epilogue 2): a value is bf16(slab 0 + slab 1 + ...), what the GEMM's own epilogue (the F.linear store, ssd/layers/linear.py:98)
 * would have written -- bit-identical to ssd_gemm_pf rows + ssd_rope_store_kv, one launch less per prefill layer. */
int ssd_rope_store_kv_parts(const float* parts, int splits, const int64_t* positions, const float* cos_sin,
                            const int32_t* slot_mapping, void* q_out_rows, void* k_cache, void* v_cache,
                            const void* q_norm_w, const void* k_norm_w, float eps, int T, int nh, int nkv, int hd,
                            int block_size, int qkv_perm, void* stream);

/* Attention.forward, all branches -- ssd/layers/attention.py:73-134.
 *   mode 0: causal, bottom-right aligned over context_lens (prefill :90-93, verify/glue :105-111,
 *           single-query decode :126-131).  cu_q int32[B+1] or NULL (then q_per_seq queries per sequence).
 *   mode 1: draft-tree decode (:113-125): tree_mq queries per sequence, structural mask of
 *           ssd/engine/helpers/mask_helpers.py:12-21 at tree step `tree_step`; tree_jidx int32[B][tree_mq]
 *           gives each branch's glue position (NULL -> branch / tree_F).
 * splits = key-range splits per (sequence, kv head) (static per launch; ranges derive from context_lens on
 * device).  ws_o fp32[T*nh*splits*hd], ws_ml fp32[T*nh*splits*2] needed when splits > 1.
 * flags bit0: use scalar LDS gathers instead of ds_read_b64_tr_b16 (diagnostic).
 * flags bit1: single-bf16 probabilities in P.V (FlashAttention's rounding) instead of the default hi+lo split.
 * flags bit2: one 16-row tile per workgroup also for query blocks of more than 8 row tiles per kv head (default: two) -- bit-identical,
 *             twice the workgroups; the host takes it for prefill chunks (ssd_amd/model.py _attn_flags).
 * flags bits 8..11: waves per workgroup (1..8, 0 = 1) that split the key range and merge in LDS. */
int ssd_attn_paged(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                   int max_blocks, const int32_t* context_lens, const int32_t* cu_q, int q_per_seq, int B, int T,
                   int max_q, int nh, int nkv, int hd, int block_size, float scale, int mode, int tree_K,
                   int tree_mq, int tree_step, int tree_F, const int32_t* tree_jidx, int splits, int flags,
                   void* ws_o, void* ws_ml, void* out_rows, void* out_frag, void* stream);

/* The same kernel under the names of the reference's two other attention call sites (thin forms: no split workspaces, one key range):
 *   ssd_attn_prefill_varlen  flash_attn_varlen_func, ssd/layers/attention.py:90-93 -- B packed causal sequences (cu_q int32[B + 1]) over the
 *                            paged cache this forward has just filled (context_lens = total lengths; T = cu_q[B] rows, max_q the longest)
 *   ssd_attn_tree            the draft tree's custom-mask prefill, ssd/layers/attention.py:113-125 + ssd/engine/helpers/mask_helpers.py:12-21 --
 *                            tree_mq branch rows per sequence at step tree_step (T = B * tree_mq), mask computed from (branch, step) */
int ssd_attn_prefill_varlen(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables, int max_blocks,
                            const int32_t* context_lens, const int32_t* cu_q, int B, int T, int max_q, int nh, int nkv, int hd,
                            int block_size, float scale, void* out_rows, void* out_frag, void* stream);
int ssd_attn_tree(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables, int max_blocks,
                  const int32_t* context_lens, int B, int tree_K, int tree_mq, int tree_step, int tree_F, const int32_t* tree_jidx,
                  int nh, int nkv, int hd, int block_size, float scale, void* out_rows, void* out_frag, void* stream);

/* Attention + o_proj in ONE launch for the single-GPU drafts' decode / glue forwards: flash_attn_with_kvcache
 * (ssd/layers/attention.py:105-111,126-131) followed by RowParallelLinear o_proj (ssd/layers/linear.py:186-199, no all-reduce
 * at tp = 1).  One sequence, T causal (bottom-right aligned) query rows; parts = fp32 slabs [nkv][T][N], slab h = o_proj
 * restricted to the columns of kv head h's q heads: their sum over h is o_proj(attention output), consumed exactly like
 * ssd_gemm_parts' slabs (ssd_gemm_fused_parts / ssd_rmsnorm_parts with splits = nkv).  T * (nh / nkv) <= 32 (hd 64) or 16
 * (hd 128), (nh / nkv) * hd <= 256, N % 128 == 0, nkv <= 16; the whole context is scanned by the 8 waves of a workgroup (use
 * it up to ~1 K keys). */
int ssd_attn_oproj_parts(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                         int max_blocks, const int32_t* context_lens, int T, int nh, int nkv, int hd, int block_size,
                         float scale, const void* w_o_frag, int N, void* parts, void* stream);

/* LM head on the greedy path: F.linear + logits.argmax(-1) -- ssd/layers/embed_head.py:88-116 followed by
 * ssd/layers/sampler.py:15-20 / ssd/utils/verify.py:34.  ssd_gemm_wf_argmax writes the bf16 logits rows like ssd_gemm_wf
 * (M <= 32) AND, per workgroup, the (max value, lowest index) of each token row over the features that workgroup produced
 * (compared on the bf16-rounded logits): part_val / part_idx [m * part_stride + p], p < ssd_gemm_wf_argmax_parts(M, N, K)
 * (a host-side query; returns the count, or a negative error).  ssd_argmax_parts* finish the argmax from those candidates
 * (larger value, then lower index == argmax over the stored logits) and, in the same launch, do what the loop does with
 * the tokens next:
 *   ssd_argmax_parts          out / out2 / out3[row * out3_stride] (any may be NULL) = idx + idx_offset; out_val (optional)
 *                             = the row maximum (vocab-parallel merge input, see ssd_argmax_merge)
 *   ssd_argmax_parts_verify   + ssd_verify_greedy (K + 1 <= 16 rows per sequence; preds optional [B*(K+1)])
 *   ssd_argmax_parts_advance  + ssd_draft_advance */
int ssd_gemm_wf_argmax_parts(int M, int N, int K);
int ssd_gemm_wf_argmax(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                       float* part_val, int32_t* part_idx, int part_stride, void* stream);

/* The single-token decode chain (one sequence, M = 1: SpeculatorSync's draft chain, ssd/engine/speculator_sync.py:25-69, and the
 * JIT chain of ssd/engine/draft_runner.py:186-378): everything of a decoder layer between two attention launches in ONE launch --
 * LlamaDecoderLayer.forward (ssd/models/llama3.py:128-199) minus the attention call: o_proj, residual add + post-attention
 * RMSNorm, gate_up + SiluAndMul, down_proj, residual add, and then EITHER the next layer's input RMSNorm + QKV projection + RoPE +
 * store_kvcache (w_qkv_next != NULL; q_out / k_cache / v_cache / positions / slots / cos_sin are the NEXT layer's) OR, on the last
 * layer (h_out != NULL), the hand-over to the final norm: h_out = down_proj rows, res_out = the residual they are added to.
 * 256 resident workgroups; the three all-to-all edges inside are all-gathers of finished bf16 vectors through data-tagged
 * 8-byte granules (csrc/chain.hip).  Same rounding points as the separate launches; fp32 summation order differs (tolerance).
 *   granules  ssd_chain_granule_bytes(h, I) bytes, zeroed ONCE at allocation
 *   gen       uint32 device word; ssd_chain_tick(gen, stream) once per forward, before its first segment
 *   err       uint32 device word, set to 1 if a bounded wait gave up (the forward's results are then invalid)
 * Shapes: ssd_chain_segment_ok(...) == SSD_OK (h, qn, I multiples of 128; h <= 4096; I <= 16384; no biases, no q/k norm). */
int ssd_chain_tick(void* gen, void* stream);
int ssd_chain_granule_bytes(int h, int I);
int ssd_chain_segment_ok(int h, int qn, int I, int qkv_n, int nh, int nkv, int hd);
int ssd_chain_segment(const void* a_frag, const void* res_in, void* res_out, void* h_out, const void* w_o, const void* w_gu,
                      const void* w_d, const void* w_qkv_next, const void* ln_post, const void* ln_next, float eps,
                      const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache, void* v_cache,
                      int h, int qn, int I, int qkv_n, int nh, int nkv, int hd, int block_size, int layer, void* granules,
                      const void* gen, void* err, void* stream);

/* The same layer segment for 2..30 token rows (csrc/tree_segment.hip): the MQ_LEN-branch tree-decode step of asynchronous speculation
 * (DraftRunner._decode_tree, ssd/engine/draft_runner.py:713-812: K forwards of (K+1)*F rows through LlamaDecoderLayer.forward,
 * ssd/models/llama3.py:128-199) and the K+1-row glue decode (draft_runner.py:560-640).  Operands as ssd_chain_segment with M rows:
 * a_frag frag [M][qn], res_in / res_out / h_out rows [M][h] (res_out != res_in), positions / slots [M], q_out rows [M][nh*hd].
 * 256 resident workgroups; the all-to-all edges are 16-byte write-through stores + one flag word per producer workgroup, payload
 * read with sc1 loads; x^ of all M rows lives in LDS.  h in {1024, 2048}; no biases.
 * qkv_rows_next (models with a per-head q / k RMSNorm, ssd/models/qwen3.py:96-104: the norm needs a whole head, i.e. eight of this
 * kernel's workgroups): instead of the RoPE + KV-store epilogue the NEXT layer's raw QKV projection rows [M][qkv_n] are written (in the
 * rotation-paired order of its weights, qkv_perm = 1) for ssd_rope_store_kv; positions / cos_sin / slots / q_out /
 * k_cache / v_cache are then NULL.
 *   workspace  ssd_tree_segment_workspace_bytes(h, I) bytes, zeroed ONCE at allocation (flags + the three hand-off buffers)
 *   gen / err  as ssd_chain_segment (one ssd_chain_tick per forward)
 * Returns SSD_ERR_LAUNCH when the device cannot keep 256 of its workgroups resident at once (they wait for each other). */
int ssd_tree_segment_workspace_bytes(int h, int I);
int ssd_tree_segment_ok(int M, int h, int qn, int I, int qkv_n, int nh, int nkv, int hd);
int ssd_tree_segment(const void* a_frag, const void* res_in, void* res_out, void* h_out, const void* w_o, const void* w_gu,
                     const void* w_d, const void* w_qkv_next, const void* ln_post, const void* ln_next, float eps,
                     const int64_t* positions, const float* cos_sin, const int32_t* slots, void* q_out, void* k_cache, void* v_cache,
                     void* qkv_rows_next, int M, int h, int qn, int I, int qkv_n, int nh, int nkv, int hd, int block_size, int layer,
                     void* workspace, const void* gen, void* err, void* stream);
int ssd_argmax_parts(const float* part_val, const int32_t* part_idx, int nparts, long part_stride, int T, long idx_offset,
                     int64_t* out, int64_t* out2, int64_t* out3, long out3_stride, float* out_val, void* stream);
int ssd_argmax_parts_verify(const float* part_val, const int32_t* part_idx, int nparts, long part_stride,
                            const int64_t* speculations, int B, int K, int64_t* preds, int32_t* accept_len,
                            int64_t* recovery, int64_t* packed, void* stream);
int ssd_argmax_parts_advance(const float* part_val, const int32_t* part_idx, int nparts, long part_stride, int64_t* next,
                             int64_t* input_ids, int64_t* positions, int32_t* slots, int32_t* context_lens,
                             const int32_t* block_tables, int max_blocks, int block_size, int64_t* spec, int K,
                             int32_t* step, int B, void* stream);

/* Sampler.forward at temperature 0 -- ssd/layers/sampler.py:15-20; verify.py:34.  out2 optional copy. */
int ssd_argmax_rows(const void* logits_rows, long ld, int T, int V, int64_t* out, int64_t* out2, void* stream);
/* Vocab-parallel form of the same argmax (ParallelLMHead gather + cat, ssd/layers/embed_head.py:88-92, followed by
 * argmax): each rank reduces its shard to (max value, global index = local + idx_offset); the per-rank rows (strides in elements of each array) are
 * all-gathered (16 bytes per row instead of the logits) and merged: larger value, then lower index. */
int ssd_argmax_rows_val(const void* logits_rows, long ld, int T, int V, long idx_offset, int64_t* out_idx,
                        float* out_val, void* stream);
int ssd_argmax_merge(const float* vals, const int64_t* idxs, int tp, int T, long stride, long stride_idx, int64_t* out,
                     int64_t* out2, void* stream);

/* verify(), greedy branch -- ssd/utils/verify.py:28-48.  preds/speculations int64 [B][K+1].
 * packed (optional) int64 [B][K+3] = (accept_len, recovery, spec_0..spec_K): the step's whole result in one D2H
 * copy instead of the reference's four .tolist() syncs (verify.py:173-181). */
int ssd_verify_greedy(const int64_t* preds, const int64_t* speculations, int B, int K, int32_t* accept_len,
                      int64_t* recovery, int64_t* packed, void* stream);

/* Temperature > 0 (csrc/stochastic.hip).  rng_state: uint64 device word (seed); advance it with ssd_rng_advance after each
 * use so hipGraph replays draw fresh numbers.  temps float[T / rows_per_temp] (one temperature per sequence).
 * ssd_sample_rows: Sampler.forward -- ssd/layers/sampler.py:15-36 (T == 0 -> argmax; else Gumbel-max == the reference's
 * argmax(softmax / Exp(1))).  ssd_row_lse: log-sum-exp of logits / T per row (softmax normaliser of verify.py:76-99).
 * ssd_verify_ratio: verify() with ratio acceptance and residual resampling -- ssd/utils/verify.py:50-167.  ratio_rows
 * int32[B]: 1 where the draft tokens were really drawn from q (cache hit or JIT), else greedy acceptance + recovery ~ p.
 * accept_prob (optional) float[B][K] receives min(1, p/q) per position.
 * sampler_x (apply_sampler_x_rescaling -- ssd/utils/async_helpers/async_spec_helpers.py:79-105; sampler.py:29-31;
 * verify.py:101-105): boost_idx int32[rows][boost_k] = the F+1 most probable tokens of each row (ssd_topk_rows),
 * boost_x = sampler_x; NULL disables it.  ssd_row_lse and ssd_verify_ratio must see the same boost rows for q. */
int ssd_topk_rows(const void* logits_rows, long ld, int T, int V, int k, int32_t* out_idx, void* stream);
int ssd_sample_rows(const void* logits_rows, long ld, int T, int V, const float* temps, int rows_per_temp,
                    const void* rng_state, unsigned salt, int64_t* out, int64_t* out2, const int32_t* boost_idx,
                    int boost_k, float boost_x, void* stream);
int ssd_rng_advance(void* rng_state, void* stream);
int ssd_row_lse(const void* logits_rows, long ld, int T, int V, const float* temps, int rows_per_temp, float* lse,
                const int32_t* boost_idx, int boost_k, float boost_x, void* stream);
int ssd_verify_ratio(const void* logits_p, long ld_p, const void* logits_q, long ld_q, int V, int B, int K,
                     const int64_t* speculations, const int64_t* preds_p, const float* lse_p, const float* lse_q,
                     const float* temps_t, const float* temps_q, const int32_t* ratio_rows, const void* rng_state,
                     unsigned salt, int32_t* accept_len, int64_t* recovery, int64_t* packed, float* accept_prob,
                     const int32_t* boost_idx_q, int boost_k, float boost_x, void* stream);

/* get_forked_recovery_tokens_from_logits -- ssd/utils/async_helpers/async_spec_helpers.py:26-78.
 * counts/offsets int32 [B][K+1]: fan-out and output offset of each glue position. */
int ssd_fork_topf(const void* logits_rows, long ld, int V, const int64_t* returned_tokens, const int32_t* counts,
                  const int32_t* offsets, int B, int K, int mq, int64_t* out, void* stream);
/* The same selection spread over the chip (every row cut into slices of <= 4096 logits, per-slice top-F candidates in `workspace`
 * -- ssd_fork_topf_workspace_bytes(V, B, K) bytes -- then one wave per row merges them): bit-equal to ssd_fork_topf, two short
 * launches instead of B*(K+1) workgroups walking whole vocabulary rows F times (84.7 -> ~8 us at V = 128256).  V % 8 == 0 and
 * V <= 196608 (48 slices): ssd_fork_topf_workspace_bytes returns SSD_ERR_SHAPE (< 0) for a vocabulary the split form refuses --
 * the predicate a caller gates on before it captures the call in a graph. */
int ssd_fork_topf_workspace_bytes(int V, int B, int K);
int ssd_fork_topf_split(const void* logits_rows, long ld, int V, const int64_t* returned_tokens, const int32_t* counts,
                        const int32_t* offsets, int B, int K, int mq, void* workspace, int64_t* out, void* stream);

/* Speculation-cache lookup -- the tensor compare of DraftRunner.hit_cache_and_respond, ssd/engine/draft_runner.py:215-252.
 * The cache of a round has Bc * W entries: entry c = b * W + i has key (cache_seq[b], cache_j[c], cache_forks[c]) =
 * (sequence id, glue position of branch i, fork token of branch i).  out_idx[r] (int32[B]) = first entry equal to request key
 * r = req_keys[r][0..2] (int64 [B][3]: seq id, accepted length - 1, recovery token), or -1. */
int ssd_cache_lookup(const int64_t* req_keys, const int64_t* cache_seq, const int32_t* cache_j, const int64_t* cache_forks,
                     int B, int Bc, int W, int32_t* out_idx, void* stream);

/* Device-side replacement for the host loop body of SpeculatorSync.speculate --
 * ssd/engine/speculator_sync.py:47-66 (+ runner_helpers.py:59-75): append the sampled token, bump
 * position / context length, recompute the KV slot from the block table. */
int ssd_draft_advance(const int64_t* next, int64_t* input_ids, int64_t* positions, int32_t* slots,
                      int32_t* context_lens, const int32_t* block_tables, int max_blocks, int block_size,
                      int64_t* spec, int K, int32_t* step, int B, void* stream);

/* logits_q[b][*step] <- src row b (bf16 [B][K][V]); collects the draft logits of a device-side chain step
 * (torch.stack(logits_q), ssd/engine/speculator_sync.py:58,67). */
int ssd_store_step_rows(const void* src_rows, long src_ld, void* dst, int B, int V, int K, const int32_t* step, void* stream);

/* One-shot full-mesh all-reduce (sum, bf16 in/out, fp32 accumulate in rank order) over hipIpc-shared fine-grained
 * buffers -- replaces dist.all_reduce after o_proj / down_proj / the embedding (ssd/layers/linear.py:195-199,
 * ssd/layers/embed_head.py:53-56) for the small decode-time messages; csrc/comm.hip describes the protocol.
 * Setup (not hot path): ssd_comm_alloc (fine-grained device memory, zeroed) / ssd_comm_ipc_export (64-byte handle) /
 * ssd_comm_ipc_open (map a peer's allocation) / ssd_comm_ipc_close / ssd_comm_free.
 * ssd_allreduce_bf16: slots[r] / flags[r] = rank r's staging area (2 * slot_elems bf16) and flag array (8*8 uint32);
 * counters uint32[8] and err uint32[1] are local device words (zeroed once); err becomes 1 if a peer did not arrive
 * within spin_budget polls (the call then leaves `out` undefined and the caller must fall back). */
int ssd_comm_alloc(void** out, long bytes);
int ssd_comm_free(void* p);
int ssd_comm_ipc_export(void* p, void* handle64);
int ssd_comm_ipc_open(const void* handle64, void** out);
int ssd_comm_ipc_close(void* p);
int ssd_allreduce_bf16(const void* in, void* out, long n, int rank, int world, void* const* slots, void* const* flags,
                       long slot_elems, void* counters, void* err, long spin_budget, void* stream);
/* Same transport as an all-gather of n8 opaque 8-byte words per rank: out [world][n8] (replaces the logits gather of
 * ParallelLMHead, ssd/layers/embed_head.py:88-92, by a (value, index) exchange of 12 bytes per row). */
int ssd_allgather_u64(const void* in, void* out, long n8, int rank, int world, void* const* slots, void* const* flags,
                      long slot_elems, void* counters, void* err, long spin_budget, void* stream);

/* RowParallelLinear's all_reduce (ssd/layers/linear.py:195-199) fused with the add + RMSNorm that always follows it
 * (ssd/layers/layernorm.py:76-88; ssd/models/llama3.py:185-199): res_out = bf16(allreduce(in) + res_in),
 * out = bf16(x32 * rsqrt(mean(x32^2) + eps) * weight) row-major and/or fragment-major.  Bit-identical to
 * ssd_allreduce_bf16 followed by ssd_rmsnorm; same slots / flags / counters.  T*H <= slot_elems. */
int ssd_allreduce_add_rmsnorm_bf16(const void* in, const void* res_in, void* res_out, const void* weight, float eps,
                                   void* out_rows, void* out_frag, int T, int H, int rank, int world, void* const* slots,
                                   void* const* flags, long slot_elems, void* counters, void* err, long spin_budget,
                                   void* stream);

/* The same two collectives over DATA-TAGGED GRANULES (round 4; csrc/comm.hip): every rank pushes its values into each peer's
 * inbox as 8-byte {2 x bf16, epoch} words -- one write-through store each, single-copy atomic -- and polls its OWN inbox: one
 * hop, no drain / flag / remote read, no ordering between separate payload and flag stores to rely on.  2x the wire bytes, so
 * for messages up to 64 Ki elements (every decode / verify sum of the tensor-parallel target); larger ones use the calls above.
 * inboxes[r] = rank r's inbox (2 * 8 * gr_cap granules, zeroed); counters / err are the SAME as in the calls above -- the two
 * protocols share one epoch and may be interleaved.  Replace ssd/layers/linear.py:195-199 (+ layernorm.py:76-88) like them. */
int ssd_allreduce_gr_bf16(const void* in, void* out, long n, int rank, int world, void* const* inboxes, long gr_cap,
                          void* counters, void* err, long spin_budget, void* stream);
int ssd_allreduce_add_rmsnorm_gr_bf16(const void* in, const void* res_in, void* res_out, const void* weight, float eps,
                                      void* out_rows, void* out_frag, int T, int H, int rank, int world, void* const* inboxes,
                                      long gr_cap, void* counters, void* err, long spin_budget, void* stream);

/* hipGraph capture for a host that is not PyTorch -- the reference captures its decode / verify / glue / tree steps with
 * torch.cuda.CUDAGraph (ssd/engine/helpers/cudagraph_helpers.py:20-120); every entry point above only enqueues on `stream`, so the calls
 * between ssd_graph_begin and ssd_graph_end become ONE replayable graph.  stream: a non-null hipStream_t; *out_exec: the executable graph. */
int ssd_graph_begin(void* stream);
int ssd_graph_end(void* stream, void** out_exec);
int ssd_graph_launch(void* exec, void* stream);
int ssd_graph_destroy(void* exec);

#ifdef __cplusplus
}
#endif
#endif
