"""CPU oracle: the reference's decoder forward restated with oracle.ops (TEST INFRASTRUCTURE ONLY).

Follows LlamaForCausalLM / Qwen3ForCausalLM: ssd/models/llama3.py:89-99,185-199,248-273 and
ssd/models/qwen3.py:90-108; parameter names are the reference's packed names (qkv_proj, gate_up_proj --
ssd/models/llama3.py:277-283) so a reference ``state_dict()`` loads unchanged.  The attention branch is chosen
exactly as ssd/layers/attention.py:73-134 does from its global Context.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.distributed as dist

from ssd_amd.model_config import ModelConfig
from oracle import ops as O


@dataclass
class Ctx:
    """Explicit form of the reference's process-global Context (ssd/utils/context.py:5-15)."""
    mode: str                                  # "prefill" | "decode" | "verify" | "tree"
    slot_mapping: torch.Tensor | None = None   # int32 [T]
    context_lens: torch.Tensor | None = None   # int32 [B]
    block_tables: torch.Tensor | None = None   # int32 [B, max_blocks]
    cu_q: torch.Tensor | None = None           # int32 [B+1]
    cu_k: torch.Tensor | None = None
    tree_step: int = 0
    tree_K: int = 0
    tree_jidx: list = field(default_factory=list)   # per sequence list[int] of length MQ_LEN


def shard_weights(cfg: ModelConfig, full: dict, rank: int, tp: int) -> dict:
    """Per-rank shards exactly as the reference weight loaders cut them (ssd/layers/linear.py:90-95,116-122,
    148-162,188-193; ssd/layers/embed_head.py:41-47)."""
    if tp == 1:
        return full
    out = {}
    hd, nh, nkv, I = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size
    for name, w in full.items():
        if name.endswith("qkv_proj.weight") or name.endswith("qkv_proj.bias"):
            q, k, v = w.split([nh * hd, nkv * hd, nkv * hd], dim=0)
            out[name] = torch.cat([q.chunk(tp, 0)[rank], k.chunk(tp, 0)[rank], v.chunk(tp, 0)[rank]], 0).contiguous()
        elif name.endswith("gate_up_proj.weight"):
            g, u = w.split([I, I], dim=0)
            out[name] = torch.cat([g.chunk(tp, 0)[rank], u.chunk(tp, 0)[rank]], 0).contiguous()
        elif name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
            out[name] = w.chunk(tp, 1)[rank].contiguous()
        elif name.endswith("embed_tokens.weight") or name.endswith("lm_head.weight"):
            out[name] = w.chunk(tp, 0)[rank].contiguous()
        else:
            out[name] = w
    return out


class OracleModel:
    def __init__(self, cfg: ModelConfig, weights: dict, num_blocks: int, block_size: int,
                 tp_rank: int = 0, tp_size: int = 1, tp_group=None):
        self.cfg, self.w = cfg, weights
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        self.nh, self.nkv = cfg.num_heads // tp_size, cfg.num_kv_heads // tp_size
        self.block_size = block_size
        dt = weights["model.embed_tokens.weight"].dtype
        # reference layout [2, L, blocks, block_size, nkv, hd] (ssd/engine/model_runner.py:484-491)
        self.kv_cache = torch.zeros(2, cfg.num_layers, num_blocks, block_size, self.nkv, cfg.head_dim, dtype=dt)
        self.cos_sin = O.make_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        self.vocab_per_rank = cfg.vocab_size // tp_size

    def _allreduce(self, y):
        if self.tp_size > 1:
            dist.all_reduce(y, group=self.tp_group)
        return y

    def _attention(self, li: int, q, k, v, ctx: Ctx):
        cfg = self.cfg
        T = q.shape[0]
        q = q.view(T, self.nh, cfg.head_dim)
        k = k.view(T, self.nkv, cfg.head_dim)
        v = v.view(T, self.nkv, cfg.head_dim)
        kc, vc = self.kv_cache[0, li], self.kv_cache[1, li]
        if ctx.slot_mapping is not None:
            O.store_kv(k, v, kc, vc, ctx.slot_mapping)
        scale = cfg.head_dim ** -0.5
        if ctx.mode == "prefill":
            if ctx.block_tables is not None:   # prefix-cache hit: keys come from the paged cache
                B = ctx.cu_q.numel() - 1
                lens = (ctx.cu_k[1:] - ctx.cu_k[:-1]).to(torch.int32)
                o = O.attn_paged(q, kc, vc, lens, ctx.block_tables, scale, cu_q=ctx.cu_q)
            else:
                o = O.attn_prefill_varlen(q, k, v, ctx.cu_q, ctx.cu_k, scale)
        elif ctx.mode == "verify":
            o = O.attn_paged(q, kc, vc, ctx.context_lens, ctx.block_tables, scale, cu_q=ctx.cu_q)
        elif ctx.mode == "tree":
            o = O.attn_tree(q, kc, vc, ctx.context_lens, ctx.block_tables, scale, ctx.tree_step, ctx.tree_K, ctx.tree_jidx)
        else:
            o = O.attn_paged(q, kc, vc, ctx.context_lens, ctx.block_tables, scale)
        return o.reshape(T, self.nh * cfg.head_dim)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, ctx: Ctx, taps=None):
        """taps (EAGLE-3 target, llama3.py:256-271): layer indices whose INPUT residual stream (hidden + residual, bf16) is
        collected; returns (hidden, acts [T, len(set(taps)) * h]) in ascending layer order, as the reference's loop does."""
        cfg, w = self.cfg, self.w
        collected = []
        h = O.embedding(input_ids, w["model.embed_tokens.weight"], self.vocab_per_rank * self.tp_rank if self.tp_size > 1 else 0)
        h = self._allreduce(h)
        residual = None
        qs, kvs = self.nh * cfg.head_dim, self.nkv * cfg.head_dim
        for li in range(cfg.num_layers):
            p = f"model.layers.{li}."
            if taps is not None and li in taps:
                collected.append(h if residual is None else h + residual)
            if residual is None:
                x, residual = O.rmsnorm(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps), h
            else:
                x, residual = O.rmsnorm(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps, residual)
            qkv = O.linear(x, w[p + "self_attn.qkv_proj.weight"], w.get(p + "self_attn.qkv_proj.bias"))
            q, k, v = qkv.split([qs, kvs, kvs], dim=-1)
            if cfg.qk_norm:
                q = O.rmsnorm(q.reshape(-1, cfg.head_dim), w[p + "self_attn.q_norm.weight"], cfg.rms_norm_eps).reshape(q.shape)
                k = O.rmsnorm(k.reshape(-1, cfg.head_dim), w[p + "self_attn.k_norm.weight"], cfg.rms_norm_eps).reshape(k.shape)
            q, k = O.rope(positions, q, k, self.cos_sin, cfg.head_dim)
            o = self._attention(li, q, k, v.contiguous(), ctx)
            h = self._allreduce(O.linear(o, w[p + "self_attn.o_proj.weight"]))
            x, residual = O.rmsnorm(h, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, residual)
            a = O.silu_mul(O.linear(x, w[p + "mlp.gate_up_proj.weight"]))
            h = self._allreduce(O.linear(a, w[p + "mlp.down_proj.weight"]))
        h, _ = O.rmsnorm(h, w["model.norm.weight"], cfg.rms_norm_eps, residual)
        if taps is not None:
            return h, torch.cat(collected, dim=-1)
        return h

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor | None:
        """ParallelLMHead.forward -- ssd/layers/embed_head.py:78-116 (gather to rank 0 + cat under TP)."""
        name = "model.embed_tokens.weight" if self.cfg.tie_word_embeddings else "lm_head.weight"
        logits = O.linear(hidden, self.w[name])
        if self.tp_size > 1:
            parts = [torch.empty_like(logits) for _ in range(self.tp_size)] if self.tp_rank == 0 else None
            dist.gather(logits, parts, dst=dist.get_global_rank(self.tp_group, 0) if self.tp_group is not None else 0,
                        group=self.tp_group)
            return torch.cat(parts, dim=-1) if self.tp_rank == 0 else None
        return logits
