/*
 * libssdhip -- tuning and experimental entry points, kept OUT of the call-site-replacement header (ssd_hip.h).
 *
 * Nothing here stands in for a reference call site by itself: the *_cfg forms are the production GEMMs with their decomposition
 * chosen by the caller (profiles/tune_gemm.py sweeps them; the defaults in ssd_hip.h's forms come from those sweeps), and the rest
 * are paths that were built, measured SLOWER than the default on MI355X and kept only as A/B switches with their measurements
 * (DESIGN.md 8c).  A maintainer integrating libssdhip binds ssd_hip.h only.
 */
#ifndef SSD_HIP_TUNE_H
#define SSD_HIP_TUNE_H
#include "ssd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ssd_gemm_wf with an explicit decomposition: nt = 16-row groups per workgroup (1,2,4), waves = K-split (1..16). */
int ssd_gemm_wf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                    int epilogue, int nt, int waves, void* stream);


/* ssd_gemm_pf with an explicit decomposition: nt = 16-row groups per wave (2 or 4; a workgroup owns 4*nt), splits of K. */
int ssd_gemm_pf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                    int epilogue, void* workspace, int64_t workspace_bytes, int nt, int splits, void* stream);


/* EXPERIMENTAL, default off (SSD_FUSE_QKV_ATTN=1), measured 3 % slower on Qwen3-32B + 0.6B (profiles/r04_bench_c5t_qkvattn_ab.txt).
ssd_rope_store_kv + ssd_attn_paged in ONE launch for the decode-side shapes (round 4): q_per_seq <= 32 new tokens per sequence
 * (single-token decode, K+1-row verify / glue, the MQ_LEN-branch tree step), context within one workgroup scan (buckets <= 1024).
 * qkv_rows: the QKV projection's rows [T][(nh + 2 nkv) * hd]; every workgroup norms (q_norm_w / k_norm_w: Qwen3's per-head
 * RMSNorm, or NULL), rotates and stores the new K / V rows of its (sequence, kv head), then builds its Q fragments from the raw
 * rows with the same arithmetic: bit-identical to the two calls.  Replaces ssd/models/qwen3.py:96-104 + ssd/layers/
 * rotary_embedding.py:40-60 + ssd/layers/attention.py:10-41 + :105-131 wherever RoPE cannot ride the QKV GEMM's epilogue. */
int ssd_attn_paged_qkv(const void* qkv_rows, const int64_t* positions, const float* cos_sin, const int32_t* slot_mapping,
                       const void* q_norm_w, const void* k_norm_w, float eps, int qkv_perm, void* k_cache, void* v_cache,
                       const int32_t* block_tables, int max_blocks, const int32_t* context_lens, int q_per_seq, int B, int T,
                       int nh, int nkv, int hd, int block_size, float scale, int mode, int tree_K, int tree_mq, int tree_step,
                       int tree_F, const int32_t* tree_jidx, int flags, void* out_rows, void* out_frag, void* stream);


/* Diagnostic (tests only): gfx950's v_cvt_pk_bf16_f32 against the integer round-to-nearest-even used everywhere else, over all 2^32
 * fp32 patterns.  counts2: two uint64 device words, zeroed by the caller: [0] mismatches on non-NaN inputs, [1] NaN inputs that did
 * not stay NaN. */
int ssd_selftest_bf16_cvt(void* counts2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
