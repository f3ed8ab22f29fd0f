"""Draft-server operations of an EAGLE-3 draft on one MI355X (the `runner` behind engine/draft_runner.py DraftServer when
config.use_eagle): the same four operations as ModelRunner's -- draft_prefill, draft_jit, draft_glue_fork, draft_tree --
each with the CONDITIONING rows of ssd_amd.eagle.HipEagleDraft staged around the forward.

Reference: the use_eagle branches of ssd/engine/draft_runner.py -- :72-101 prefill from target activations, :133-177 JIT
chain (fc of the recovery activation, then self-conditioning), :538-612 the variable-length glue over
[extend | recovery | spec] rows, :660-676 tree hidden states from the glue prenorms, :734-750 per-step prenorm hand-over
and the cached branch activations.  The JIT chain and the K tree steps are single hipGraphs like their non-EAGLE
counterparts, and so is the glue: its data-dependent row count is padded to B*(2K+1) rows with ghost rows (round 3).
"""
from __future__ import annotations

import torch

from ssd_amd.engine.model_runner import ModelRunner
from ssd_amd.hip import ops as H


def glue_layout(num_tokens, counts, K: int) -> dict:
    """Row layout of the EAGLE glue forward (reference draft_runner.py:548-612 builds it with masks over a packed tensor;
    prepare_glue_decode_ctxt_eagle :450-493 the positions).  num_tokens are the ALREADY SHIFTED counts (sequence length - 1);
    sequence b contributes counts[b] extend rows, the recovery row and K spec rows at positions n-1-counts[b] .. n-1+K.
      pos     position of every packed row            cu      row offsets per sequence
      tc_src  rows of the request's [B, K+1, A] activation tensor (extend rows, then the recovery row K) ...
      tc_dst  ... and the packed rows they condition  sp_dst  packed rows conditioned on the previous round's prenorms
      kp1     packed rows [recovery | spec] whose logits / prenorms feed the fork and the tree"""
    pos, cu, tc_src, tc_dst, sp_dst, kp1 = [], [0], [], [], [], []
    for b, n in enumerate(num_tokens):
        ne, base = counts[b], cu[-1]
        pos.extend(range(n - 1 - ne, n + K))
        tc_src.extend([b * (K + 1) + j for j in range(ne)] + [b * (K + 1) + K])
        tc_dst.extend(range(base, base + ne + 1))
        sp_dst.extend(range(base + ne + 1, base + ne + 1 + K))
        kp1.extend(range(base + ne, base + ne + K + 1))
        cu.append(base + ne + K + 1)
    return dict(pos=pos, cu=cu, tc_src=tc_src, tc_dst=tc_dst, sp_dst=sp_dst, kp1=kp1)


class EagleDraftRunner(ModelRunner):
    def __init__(self, config, model_cfg, **kw):
        assert kw.get("is_draft", True) and model_cfg.family == "eagle3"
        K = config.speculate_k
        kw.setdefault("max_decode_tokens", config.max_num_seqs * max(K + 1, config.MQ_LEN, 2 * K + 1))
        super().__init__(config, model_cfg, **kw)
        m, B, dev = self.model, self.max_bs, dict(device=self.device)
        self.mq = config.MQ_LEN
        rows = B * self.mq
        self.d_acts_in = torch.zeros(B * (K + 1), m.A, dtype=torch.bfloat16, **dev)       # request activations [B][K extend | recovery]
        self.d_tc = torch.zeros(B * (K + 1), m.h, dtype=torch.bfloat16, **dev)            # fc of the rows gathered from d_acts_in
        self.d_jit_acts = torch.zeros(B, K, m.h, dtype=torch.bfloat16, **dev)
        self.d_glue_pre = torch.zeros(B * (K + 1), m.h, dtype=torch.bfloat16, **dev)      # prenorms of the [recovery | spec] glue rows
        self.d_tree_acts = torch.zeros(rows, K, m.h, dtype=torch.bfloat16, **dev)
        self.d_cond_idx = torch.zeros(rows, dtype=torch.int64, **dev)                     # row of d_glue_pre that seeds each branch
        self.d_idx = {n: torch.zeros(self.max_decode_tokens, dtype=torch.int64, **dev) for n in ("tc_src", "tc_dst", "sp_dst", "kp1")}
        self.d_gather32 = torch.zeros(B * (K + 1), dtype=torch.int32, **dev)
        # static inputs of the glue hipGraph
        self.d_glue_ids = torch.zeros(B, K + 1, dtype=torch.int64, **dev)
        self.d_prev_acts = torch.zeros(B * K, m.h, dtype=torch.bfloat16, **dev)
        self.d_gathered = torch.zeros(B * (K + 1), m.A, dtype=torch.bfloat16, **dev)

    # ---- prefill: token j (of the already shifted prompt) is conditioned on fc(target activation of position j-1) ----
    @torch.inference_mode()
    def draft_prefill(self, token_lists, tables, acts: torch.Tensor) -> None:
        assert len(token_lists) <= self.seq_cap
        B, T, max_q = self._stage_draft_prefill(token_lists, tables)
        assert acts.shape[0] == T
        self.model.project(acts.to(self.device).contiguous(), T, self.model.buf_cond)
        self.model.forward(self.d_ids, self.d_pos, T, self._meta("prefill", B, max_q))

    # ---- cache miss: K chained single-token decodes, conditioned on fc(recovery activation) and then on themselves ----
    @torch.inference_mode()
    def draft_jit(self, rec, num_tokens, tables, temps=None, cond: torch.Tensor | None = None) -> torch.Tensor:
        assert temps is None or not any(t > 0 for t in temps), "EAGLE-3 drafts are greedy-only (reference bench.py:86)"
        B, K, m = len(rec), self.K, self.model
        self._note_ctx(max(num_tokens) + self._async_lookahead())

        def stage():
            pos = [n - 1 for n in num_tokens]
            self._upload(self.d_ids, list(rec), torch.int64)
            self._upload(self.d_pos, pos, torch.int64)
            self._upload(self.d_slots, [self._slot(tb, p) for tb, p in zip(tables, pos)], torch.int32)
            self._upload(self.d_ctx, list(num_tokens), torch.int32)
            self._upload_tables(tables)
            self.d_step.zero_()
            self.d_acts_in[:B].copy_(cond)

        def chain():
            m.project(self.d_acts_in, B, m.buf_cond)
            for k in range(K):
                m.forward(self.d_ids, self.d_pos, B, self._meta("decode", B))
                m.compute_logits(B)
                m.argmax(B, self.d_next)
                self.d_jit_acts[:B, k].copy_(m.buf_pre[:B])
                m.buf_cond[:B].copy_(m.buf_pre[:B])
                H.draft_advance(self.d_next, self.d_ids, self.d_pos, self.d_slots, self.d_ctx, self.d_bt, self.max_blocks,
                                self.block_size, self.d_spec, self.K, self.d_step, B)

        stage()
        if self._launch(("eagle_chain", B), chain) == "captured":
            stage()
            self.graphs[("eagle_chain", B, self._ctx_hint)].replay()
        return self.d_spec[:B, 1:].clone()

    def jit_acts(self, B: int) -> torch.Tensor:
        return self.d_jit_acts[:B].clone()

    def tree_acts(self, T: int) -> torch.Tensor:
        return self.d_tree_acts[:T]

    # ---- glue: [extend rows | recovery | K spec rows] per sequence, packed without padding ----
    @torch.inference_mode()
    def draft_glue_fork(self, glue_ids: torch.Tensor, num_tokens, tables, fan_lists, eagle: dict | None = None) -> torch.Tensor:
        self._ensure_tree_buffers()
        B, K, m = glue_ids.shape[0], self.K, self.model
        counts, ext_ids = eagle["ext_counts"], eagle["ext_ids"]
        self._note_ctx(max(num_tokens) + self._async_lookahead())
        lay = glue_layout(num_tokens, counts, K)
        pos, cu, tc_src, tc_dst, sp_dst, kp1 = lay["pos"], lay["cu"], lay["tc_src"], lay["tc_dst"], lay["sp_dst"], lay["kp1"]
        ids, slots = [], []
        for b, tb in enumerate(tables):
            ids.extend(list(ext_ids[b][:counts[b]]) + [0] * (K + 1))
            slots.extend(self._slot(tb, p) for p in pos[cu[b]:cu[b + 1]])
        T, n_tc = cu[-1], len(tc_src)
        # The row count is data dependent (B*(K+1) .. B*(2K+1)); the launch geometry is not: the forward always runs over
        # T_pad = B*(2K+1) rows -- ghost rows behind the packed ones carry token 0, slot -1 (nothing stored) and belong to no
        # sequence of cu_q (attention skips them) -- and every index list is padded to its maximum with entries that move row 0
        # into a scratch row.  The whole glue (conditioning gather + fc, one layer, head, prenorm gather, fork) then replays as
        # ONE hipGraph like the other draft graphs (the reference captures it too, cudagraph_helpers.py:636-774).
        T_pad, NK = B * (2 * K + 1), B * (K + 1)
        dump = T_pad                                   # scratch row of buf_cond behind the padded rows
        assert T <= T_pad < m.buf_cond.shape[0] and T_pad <= self.max_decode_tokens
        self._upload(self.d_ids, ids + [0] * (T_pad - T), torch.int64)
        self._upload(self.d_pos, pos + [0] * (T_pad - T), torch.int64)
        self._upload(self.d_slots, slots + [-1] * (T_pad - T), torch.int32)
        self._upload(self.d_ctx, [n + K for n in num_tokens], torch.int32)
        self._upload(self.d_cu_q, cu, torch.int32)
        self._upload_tables(tables)
        pad = NK - n_tc
        for name, vals in (("tc_src", tc_src + [0] * pad), ("tc_dst", tc_dst + [dump] * pad), ("sp_dst", sp_dst), ("kp1", kp1)):
            self._upload(self.d_idx[name], vals, torch.int64)
        self._upload(self.d_gather32, kp1, torch.int32)
        fl = [list(f) for f in fan_lists]
        self._upload(self.d_fan, fl, torch.int32)
        self._upload(self.d_fan_off, [[sum(f[:j]) for j in range(len(f))] for f in fl], torch.int32)
        # device-resident inputs of the round, copied into the graph's static buffers
        self.d_glue_ids[:B].copy_(glue_ids)
        self.d_acts_in[:NK].copy_(eagle["acts"].reshape(NK, -1))
        self.d_prev_acts[:B * K].copy_(eagle["prev_acts"].reshape(B * K, -1))

        def body():
            kp1_d = self.d_idx["kp1"][:NK]
            # token ids: extend tokens came from the host, [recovery | spec] from the reply that is still on the device
            self.d_ids.index_copy_(0, kp1_d, self.d_glue_ids[:B].reshape(-1))
            # conditioning rows: ONE fc over every target-conditioned row (draft_runner.py:586-587), previous prenorms on the spec rows
            torch.index_select(self.d_acts_in, 0, self.d_idx["tc_src"][:NK], out=self.d_gathered[:NK])
            m.project(self.d_gathered, NK, self.d_tc)
            m.buf_cond.index_copy_(0, self.d_idx["tc_dst"][:NK], self.d_tc[:NK])
            m.buf_cond.index_copy_(0, self.d_idx["sp_dst"][:B * K], self.d_prev_acts[:B * K])
            m.forward(self.d_ids, self.d_pos, T_pad, self._meta("prefill", B, 2 * K + 1))
            m.compute_logits(T_pad, gather=self.d_gather32, rows=NK)            # only the K+1 [recovery | spec] rows feed the fork
            torch.index_select(m.buf_pre, 0, kp1_d, out=self.d_glue_pre[:NK])
            H.fork_topf(m.logits, m.V, m.V, self.d_glue_ids[:B], self.d_fan, self.d_fan_off, B, K, self.mq, self.d_forks)

        self._launch(("eagle_glue", B), body)       # "captured": the eager warm-up run already produced this call's result
        return self.d_forks[:B].clone()

    # ---- tree: branch i starts from the glue prenorm of its position j_i, then conditions on its own previous step ----
    def _body_tree(self, B: int, d: int, sample: bool = False, mq: int | None = None) -> None:
        m = self.model
        T = B * (mq or self.mq)
        if d == 0:
            torch.index_select(self.d_glue_pre, 0, self.d_cond_idx[:T], out=m.buf_cond[:T])
        super()._body_tree(B, d, sample, mq)
        self.d_tree_acts[:T, d].copy_(m.buf_pre[:T])
        m.buf_cond[:T].copy_(m.buf_pre[:T])

    @torch.inference_mode()
    def draft_tree(self, forks: torch.Tensor, num_tokens, tables, jlists, temps=None, eagle: bool = True) -> torch.Tensor:
        assert temps is None or not any(t > 0 for t in temps), "EAGLE-3 drafts are greedy-only (reference bench.py:86)"
        K = self.K
        self._upload(self.d_cond_idx, [b * (K + 1) + j for b, jl in enumerate(jlists) for j in jl], torch.int64)
        return super().draft_tree(forks, num_tokens, tables, jlists, temps)
