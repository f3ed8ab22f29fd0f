"""CPU oracle of the EAGLE-3 draft (TEST INFRASTRUCTURE ONLY): Eagle3DraftForCausalLM restated with oracle.ops.

Follows ssd/models/eagle3_draft_llama3.py: Eagle3Attention.forward :91-99 (QKV over the 2h-wide concatenation),
Eagle3DecoderLayer.forward :142-156 (two input norms, the CONDITIONING features are the residual stream),
Eagle3DraftForCausalLM.forward :262-281 (fc only when the features are target activations) and .compute_logits
:305-352 (draft-vocabulary head scattered into the target vocabulary through d2t, -inf elsewhere).  Parameter names are
the reference's (fc.weight, final_norm.weight, model.embed_tokens.weight, model.layer.*, lm_head.weight, d2t), so a
reference ``state_dict()`` + its ``d2t_tensor`` load unchanged.  Pinned to tests/golden/tiny_eagle3.npz.
"""
from __future__ import annotations

import torch

from oracle import ops as O
from oracle.model import OracleModel, Ctx  # noqa: F401  (Ctx re-exported for the callers)


class OracleEagleDraft(OracleModel):
    def __init__(self, cfg, weights: dict, num_blocks: int, block_size: int):
        assert cfg.family == "eagle3" and cfg.num_layers == 1
        super().__init__(cfg, weights, num_blocks, block_size)
        self.target_index = torch.arange(cfg.draft_vocab_size, dtype=torch.int64) + weights["d2t"].to(torch.int64)
        self.act_dim = cfg.eagle_taps * cfg.d_model_target

    def project(self, target_acts: torch.Tensor) -> torch.Tensor:
        """fc (eagle3_draft_llama3.py:250,275): [T, taps * d_model_target] target activations -> [T, h] conditioning rows."""
        w = self.w["fc.weight"]
        return O.linear(target_acts.to(w.dtype), w)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, cond: torch.Tensor, ctx: Ctx) -> torch.Tensor:
        """-> prenorm [T, h]: the draft's own conditioning vector for the next step AND the input of compute_logits."""
        cfg, w = self.cfg, self.w
        if cond.shape[-1] == self.act_dim:          # :268-275: target activations are projected, draft prenorms pass through
            cond = self.project(cond)
        p = "model.layer."
        emb = O.embedding(input_ids, w["model.embed_tokens.weight"])
        x = torch.cat([O.rmsnorm(emb, w[p + "input_layernorm.weight"], cfg.rms_norm_eps),
                       O.rmsnorm(cond, w[p + "conditioning_feature_ln.weight"], cfg.rms_norm_eps)], dim=-1)
        qs, kvs = self.nh * cfg.head_dim, self.nkv * cfg.head_dim
        q, k, v = O.linear(x, w[p + "self_attn.qkv_proj.weight"]).split([qs, kvs, kvs], dim=-1)
        q, k = O.rope(positions, q.contiguous(), k.contiguous(), self.cos_sin, cfg.head_dim)
        o = self._attention(0, q, k, v.contiguous(), ctx)
        h = O.linear(o, w[p + "self_attn.o_proj.weight"])
        x, residual = O.rmsnorm(h, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, cond)
        a = O.silu_mul(O.linear(x, w[p + "mlp.gate_up_proj.weight"]))
        return O.linear(a, w[p + "mlp.down_proj.weight"]) + residual          # eager bf16 add (:155)

    def compute_logits(self, prenorm: torch.Tensor) -> torch.Tensor:
        """[n, vocab_size]: the draft head's rows at their target-vocabulary positions, -inf everywhere else."""
        h = O.rmsnorm(prenorm, self.w["final_norm.weight"], self.cfg.rms_norm_eps)
        lg = O.linear(h, self.w["lm_head.weight"])
        full = lg.new_full((lg.shape[0], self.cfg.vocab_size), float("-inf"))
        full[:, self.target_index] = lg
        return full
