/*
 * libssdhip -- tuning and experimental entry points, kept OUT of the call-site-replacement header (ssd_hip.h).
 *
 * Nothing here stands in for a reference call site by itself: the *_cfg forms are the production GEMMs with their decomposition
 * chosen by the caller (profiles/tune_gemm.py sweeps them; the defaults in ssd_hip.h's forms come from those sweeps), plus one
 * diagnostic.  (Round 6: the default-off experiment ssd_attn_paged_qkv -- measured 3 % slower, DESIGN.md 8c item 5 -- was deleted with
 * its switch.)  A maintainer integrating libssdhip binds ssd_hip.h only.
 */
#ifndef SSD_HIP_TUNE_H
#define SSD_HIP_TUNE_H
#include "ssd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ssd_gemm_wf with an explicit decomposition: nt = 16-row groups per workgroup (1,2,4), waves = K-split (1..16), bits 8..15 of waves =
 * consecutive tiles per workgroup; bit 8 of nt = the DEEP form (twice the k-tiles per stage; M <= 16, <= 8 waves, nt 2 / 4, epilogue 0 / 1). */
int ssd_gemm_wf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                    int epilogue, int nt, int waves, void* stream);


/* ssd_gemm_pf with an explicit decomposition: nt = 16-row groups per wave (2 or 4; a workgroup owns 4*nt), splits of K. */
int ssd_gemm_pf_cfg(const void* x_frag, const void* w_frag, const void* bias, void* y, int M, int N, int K, int ldy,
                    int epilogue, void* workspace, int64_t workspace_bytes, int nt, int splits, void* stream);


/* A/B facility (bench.py --tune-deep, profiles/): ssd_gemm_wf's default dispatch for the 70B-class matrices -- 1 (default) the DEEP
 * form, 0 the plain kernels of rounds 1-5.
 * Process-wide; call before any hipGraph is captured. */
int ssd_tune_deep(int mode);

/* Diagnostic (tests only): gfx950's v_cvt_pk_bf16_f32 against the integer round-to-nearest-even used everywhere else, over all 2^32
 * fp32 patterns.  counts2: two uint64 device words, zeroed by the caller: [0] mismatches on non-NaN inputs, [1] NaN inputs that did
 * not stay NaN. */
int ssd_selftest_bf16_cvt(void* counts2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
