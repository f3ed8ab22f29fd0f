"""Import the REAL reference (``/root/reference``) on CPU so it can pin the oracle.

Only usable in the build container (the GPU box has no /root/reference); used by
``tests/golden/make_golden.py`` to generate the committed golden vectors and by
``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent).

What is stubbed (the un-vendored CUDA wheels):
  * ``flashinfer.BatchPrefillWithPagedKVCacheWrapper``          (ssd/engine/model_runner.py:10)
  * ``sgl_kernel.flash_attn.flash_attn_varlen_func`` / ``flash_attn_with_kvcache`` (ssd/layers/attention.py:6)
  * ``ssd.layers.attention.store_kvcache`` (Triton, GPU-only)  (ssd/layers/attention.py:34-41)
They are replaced by the fp32 restatement in ``oracle.ops`` (softmax(q k^T scale + mask) v), which is the
published algorithm of those wheels; everything else executed is the reference's own code.
"""
from __future__ import annotations

import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("SSD_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ssd"))


_loaded = None


def load_reference():
    """Returns the imported reference ``ssd`` package (with CPU shims installed)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    os.environ.setdefault("SSD_HF_CACHE", "/tmp/ssd_ref_cache")
    os.environ.setdefault("SSD_DATASET_DIR", "/tmp/ssd_ref_cache")
    # our repo also ships a drop-in `ssd` alias package; make sure the reference wins in this process
    for name in [m for m in sys.modules if m == "ssd" or m.startswith("ssd.")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)

    from oracle import ops as O

    fi = types.ModuleType("flashinfer")
    fi.BatchPrefillWithPagedKVCacheWrapper = object
    sys.modules["flashinfer"] = fi
    sgl = types.ModuleType("sgl_kernel")
    fa = types.ModuleType("sgl_kernel.flash_attn")

    def flash_attn_varlen_func(q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k, softmax_scale, causal):
        assert causal
        return O.attn_prefill_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale)

    def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, page_table, softmax_scale, causal,
                                cu_seqlens_q=None, max_seqlen_q=None):
        assert causal
        if cu_seqlens_q is None:  # q [B, 1, nh, hd]
            o = O.attn_paged(q.squeeze(1), k_cache, v_cache, cache_seqlens, page_table, softmax_scale)
            return o.unsqueeze(1)
        return O.attn_paged(q, k_cache, v_cache, cache_seqlens, page_table, softmax_scale, cu_q=cu_seqlens_q)

    fa.flash_attn_varlen_func = flash_attn_varlen_func
    fa.flash_attn_with_kvcache = flash_attn_with_kvcache
    sgl.flash_attn = fa
    sys.modules["sgl_kernel"] = sgl
    sys.modules["sgl_kernel.flash_attn"] = fa

    import ssd as ref  # noqa: E402  (the reference package)
    import ssd.layers.attention as A

    A.store_kvcache = O.store_kv
    _loaded = ref
    return ref


class TreeShim:
    """Stands in for flashinfer's prefill wrapper (reference ssd/layers/attention.py:114-125); the mask comes
    from the reference's own get_custom_mask (ssd/engine/helpers/mask_helpers.py)."""

    def __init__(self, cfg, K, F, get_context, get_custom_mask):
        self.cfg, self.K, self.F = cfg, K, F
        self.get_context, self.get_custom_mask = get_context, get_custom_mask
        self.step = 0
        self.cache_hits = None

    def run(self, q, kv):
        from oracle import ops as O
        ctx = self.get_context()
        B = ctx.context_lens.shape[0]
        mq = q.shape[0] // B
        k_cache, v_cache = kv
        mask = self.get_custom_mask(self.cfg, ctx.context_lens, self.step, self.K, self.F, B, q.device, self.cache_hits)
        outs, off = [], 0
        scale = q.shape[-1] ** -0.5
        for b in range(B):
            L = int(ctx.context_lens[b])
            mb = mask[off:off + mq * L].view(mq, L)
            off += mq * L
            ks = O.gather_paged(k_cache, ctx.block_tables[b], L)
            vs = O.gather_paged(v_cache, ctx.block_tables[b], L)
            outs.append(O._sdpa(q[b * mq:(b + 1) * mq], ks, vs, mb, scale))
        return torch.cat(outs, 0)
