"""The EAGLE-3 draft on MI355X: one decoder layer whose QKV projection reads the concatenation of the normalised token
embedding and a normalised CONDITIONING row -- the target's tapped activations through ``fc`` for tokens the target has
already processed, the draft's own previous output ("prenorm") for tokens it is speculating about.

Functionally Eagle3DraftForCausalLM of the reference (ssd/models/eagle3_draft_llama3.py: Eagle3Attention.forward :91-99,
Eagle3DecoderLayer.forward :142-156, Eagle3DraftForCausalLM.forward :262-281, compute_logits :305-352), laid out as a
fixed launch list over pre-allocated buffers like ssd_amd.model.HipDecoder (whose GEMM / attention / sampling plumbing
it inherits):
  embedding -> ssd_rmsnorm_pair (both input norms, written as ONE [T][2h] fragment activation: no torch.cat) ->
  QKV GEMM over K = 2h with RoPE + KV store in its epilogue -> paged attention -> O GEMM ->
  add(conditioning) + RMSNorm -> gate_up GEMM + SiLU*mul -> down GEMM -> + residual = prenorm.
compute_logits runs the draft-vocabulary head and drops its columns at their target-vocabulary positions of a logits
buffer that is -inf everywhere else (filled once: d2t never changes), so argmax / top-F fork run unchanged.
"""
from __future__ import annotations

import torch

from ssd_amd.hip import ops as H
from ssd_amd.model import HipDecoder, AttnMeta, BF16


class HipEagleDraft(HipDecoder):
    def __init__(self, cfg, **kw):
        assert cfg.family == "eagle3" and cfg.num_layers == 1 and not cfg.qk_norm
        assert kw.get("tp_size", 1) == 1, "the draft model is not tensor-parallel"
        super().__init__(cfg, **kw)
        self.use_parts = False
        self.argmax_fused = False        # the draft-vocabulary head is scattered into target-vocabulary columns: argmax reads the logits
        dev, T, h = self.device, self.max_tokens, self.h
        self.A = cfg.eagle_taps * cfg.d_model_target        # width of the target activations fc projects
        self.Vd = cfg.draft_vocab_size
        assert self.Vd % 16 == 0 and self.A % 32 == 0

        def z(*shape, dtype=BF16):
            return torch.zeros(*shape, dtype=dtype, device=dev)

        self.buf_emb = z(T, h)
        self.buf_cond = z(T, h)         # conditioning rows of the next forward: fc(target activations) or previous prenorms
        self.buf_pre = z(T, h)          # output of the last forward
        self.buf_xf2 = z(H.frag_numel(T, 2 * h))
        self.buf_actsf = z(H.frag_numel(T, self.A))
        self.logits_d = z(self.max_logit_rows, self.Vd)
        self.logits.fill_(float("-inf"))
        self.target_index: torch.Tensor | None = None

    def load_weights(self, weight_iter) -> None:
        super().load_weights(weight_iter)       # qkv_proj -> rotation-paired fragment layout over K = 2h; fc / lm_head -> fragment layout
        d2t = self.w.pop("d2t").to(torch.int64)
        self.target_index = (torch.arange(self.Vd, dtype=torch.int64, device=self.device) + d2t).contiguous()
        assert int(self.target_index.max()) < self.cfg.vocab_size and int(self.target_index.min()) >= 0

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for n, t in self.w.items() if n != "model.embed_tokens.weight")

    def project(self, acts_rows: torch.Tensor, n: int, out_rows: torch.Tensor) -> None:
        """fc (:250,275): out_rows[:n] = acts_rows[:n] @ fc.weight^T; acts_rows bf16 [n, taps * d_model_target] contiguous."""
        assert acts_rows.is_contiguous() and acts_rows.shape[-1] == self.A and acts_rows.dtype == BF16
        H.rows_to_frag(acts_rows, self.buf_actsf, n, self.A)
        self._gemm(self.buf_actsf, self.A, self.w["fc.weight"], self.h, out_rows, n, self.h)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, T: int, meta: AttnMeta) -> None:
        """One draft forward over T rows whose conditioning rows are in buf_cond[:T]; leaves the prenorm in buf_pre[:T]."""
        cfg, w, h = self.cfg, self.w, self.h
        p = "model.layer."
        H.embedding(input_ids, w["model.embed_tokens.weight"], self.buf_emb, T, h, vocab_start=0, vocab_count=cfg.vocab_size)
        H.rmsnorm_pair(self.buf_emb, w[p + "input_layernorm.weight"], self.buf_cond, w[p + "conditioning_feature_ln.weight"],
                       cfg.rms_norm_eps, T, h, out_frag=self.buf_xf2)
        kc, vc = self.kv_cache[0, 0], self.kv_cache[0, 1]
        if T <= 32:         # RoPE + KV store in the QKV epilogue (csrc/gemm_fused.hip), one or two token tiles
            H.gemm_fused(w[p + "self_attn.qkv_proj.weight"], T, self.qkv_n, 2 * h, H.FEPI_QKV_ROPE, x_frag=self.buf_xf2,
                         positions=positions, cos_sin=self.cos_sin, slots=meta.slot_mapping, q_out=self.buf_q, k_cache=kc,
                         v_cache=vc, nh=self.nh, nkv=self.nkv, hd=self.hd, block_size=self.block_size)
        else:
            self._gemm(self.buf_xf2, 2 * h, w[p + "self_attn.qkv_proj.weight"], self.qkv_n, self.buf_qkv, T, self.qkv_n)
            H.rope_store_kv(self.buf_qkv, positions, self.cos_sin, meta.slot_mapping, self.buf_q, kc, vc, T, self.nh, self.nkv,
                            self.hd, self.block_size, eps=cfg.rms_norm_eps, qkv_perm=1)
        splits, attn_waves = self._attn_cfg(T, meta)
        H.attn_paged(self.buf_q, kc, vc, meta.block_tables, self.max_blocks, meta.context_lens, meta.B, T, meta.max_q, self.nh,
                     self.nkv, self.hd, self.block_size, self.hd ** -0.5, cu_q=meta.cu_q, q_per_seq=meta.q_per_seq, mode=meta.mode,
                     tree_K=meta.tree_K, tree_mq=meta.tree_mq, tree_step=meta.tree_step, tree_F=meta.tree_F, tree_jidx=meta.tree_jidx,
                     splits=splits, ws_o=self.ws_o, ws_ml=self.ws_ml, out_frag=self.buf_af, waves=attn_waves)
        self._gemm(self.buf_af, self.qn, w[p + "self_attn.o_proj.weight"], h, self.buf_h, T, h)
        # the CONDITIONING features, not the token embeddings, are the residual stream (:151-155)
        H.rmsnorm(self.buf_h, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, T, h, res_in=self.buf_cond,
                  res_out=self.buf_res, out_frag=self.buf_xf)
        self._gemm(self.buf_xf, h, w[p + "mlp.gate_up_proj.weight"], 2 * self.I, self.buf_actf, T, 0, epi=H.EPI_SILU_FRAG)
        self._gemm(self.buf_actf, self.I, w[p + "mlp.down_proj.weight"], h, self.buf_h, T, h)
        torch.add(self.buf_h[:T], self.buf_res[:T], out=self.buf_pre[:T])          # eager bf16 add of the reference (:155)
        self._last_parts = False

    def compute_logits(self, T: int, gather: torch.Tensor | None = None, rows: int | None = None) -> int:
        """final_norm + draft-vocabulary head, scattered to target-vocabulary columns of self.logits[:n] (:305-352)."""
        n = T if gather is None else rows
        assert n <= self.max_logit_rows
        H.rmsnorm(self.buf_pre, self.w["final_norm.weight"], self.cfg.rms_norm_eps, n, self.h, out_frag=self.buf_lastf, gather=gather)
        self._gemm(self.buf_lastf, self.h, self.w["lm_head.weight"], self.Vd, self.logits_d, n, self.Vd)
        self.logits[:n].index_copy_(1, self.target_index, self.logits_d[:n])
        return n
