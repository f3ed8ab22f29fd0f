"""Target side of asynchronous speculation: ask the draft GPU for K tokens per sequence, keyed by the outcome of
the previous verification (reference SpeculatorAsync, ssd/engine/speculator_async.py:12-187).

Per step the head rank sends ONE request (header + fused int64 payload) and receives ONE reply
(cache_hits | tokens); the other tensor-parallel ranks get the reply by an RCCL broadcast inside the TP group
(the reference ships the updated sequences to its TP workers through shared-memory pickles instead,
model_runner.py:404-428).  The request is built from Python ints in one go -- the reference writes every field
with its own device scalar store (speculator_async.py:136-146).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from ssd_amd.engine import async_proto as P
from ssd_amd.engine.speculate_types import SpeculateResult, SpeculatorBase, VerifyResult


class AsyncLink:
    """The head rank's connection to the draft server (+ fan-out of replies to the other TP ranks)."""

    def __init__(self, config, topo, transport=None):
        self.config, self.topo = config, topo
        self.K = config.speculate_k
        self.max_blocks = config.max_blocks
        self.is_head = topo.tp_rank == 0
        if transport is None and self.is_head:
            transport = P.DistTransport(topo.async_group, topo.draft_rank, topo.device)
        self.tx = transport
        self._closed = False

    def _bcast_ints(self, values: list[int] | None, n: int) -> list[int]:
        if self.topo.tp_size == 1:
            return values
        t = torch.tensor(values if self.is_head else [0] * n, dtype=torch.int64, device=self.topo.device)
        dist.broadcast(t, src=dist.get_global_rank(self.topo.tp_group, 0), group=self.topo.tp_group)
        return t.tolist()

    def draft_num_blocks(self) -> int:
        v = None
        if self.is_head:
            self.tx.send_ints([P.CMD_HELLO, 0, 0, 0])
            v = self.tx.recv_ints(1)
        return self._bcast_ints(v, 1)[0]

    def prefill(self, token_lists, block_tables, eagle_acts=None) -> None:
        if not self.is_head:
            return
        payload = P.pack_prefill(token_lists, block_tables, self.max_blocks)
        self.tx.send_ints([P.CMD_PREFILL, len(token_lists), len(payload), P.FLAG_EAGLE if eagle_acts is not None else 0])
        self.tx.send_ints(payload)
        if eagle_acts is not None:
            self.tx.send_tensor(eagle_acts)
        # co-located draft server (loopback transport): let it take the command NOW, so that its prefill is enqueued on the
        # draft stream before the target's prefill starts (step.py:75-79: "draft and target prefill overlap") instead of
        # at the first speculation request
        pump = getattr(self.tx, "pump", None)
        if pump is not None and not (eagle_acts is None and getattr(self.tx, "defer_prefill", False)):
            pump()      # (co-located on a GPU: deferred to the first speculation request, engine/llm_engine.py)

    def speculate(self, keys, num_tokens, block_tables, temps, want_logits: bool = False, eagle=None):
        """-> (hits, tokens, logits_q or None).  logits_q bf16 [B, K, V] is requested only when some temperature is
        > 0 (FLAG_WANT_LOGITS) and reaches every TP rank: each rank runs the ratio test on its own device."""
        B, K = len(keys), self.K
        resp, lq = None, None
        V = self.config.hf_config.vocab_size
        if self.is_head:
            flags = P.FLAG_WANT_LOGITS if want_logits else 0
            if eagle is not None:       # (extend_counts [B], extend_ids [B][K], acts [B, K+1, A])
                payload = P.pack_speculate(keys, num_tokens, block_tables, temps, self.max_blocks, eagle[0], eagle[1])
                flags |= P.FLAG_EAGLE
            else:
                payload = P.pack_speculate(keys, num_tokens, block_tables, temps, self.max_blocks)
            self.tx.send_ints([P.CMD_SPECULATE, B, len(payload), flags])
            self.tx.send_ints(payload)
            if eagle is not None:
                self.tx.send_tensor(eagle[2])
            resp = self.tx.recv_tensor((B + B * K,), torch.int64).tolist()
            if want_logits:
                lq = self.tx.recv_tensor((B, K, V), torch.bfloat16).to(self.topo.device)
        resp = self._bcast_ints(resp, B + B * K)
        if want_logits and self.topo.tp_size > 1:
            if lq is None:
                lq = torch.empty(B, K, V, dtype=torch.bfloat16, device=self.topo.device)
            dist.broadcast(lq, src=dist.get_global_rank(self.topo.tp_group, 0), group=self.topo.tp_group)
        hits = resp[:B]
        tokens = [resp[B + b * K: B + (b + 1) * K] for b in range(B)]
        return hits, tokens, lq

    def shutdown(self) -> None:
        if self.is_head and not self._closed:
            self._closed = True
            self.tx.send_ints([P.CMD_EXIT, 0, 0, 0])


class SpeculatorAsync(SpeculatorBase):
    def __init__(self, lookahead: int, device, link: AsyncLink, config):
        super().__init__(lookahead, device)
        self.link, self.config = link, config

    def prefill(self, seqs, verify_result: VerifyResult) -> SpeculateResult:
        token_lists = [list(s.token_ids) for s in seqs]
        acts = verify_result.eagle_acts
        if acts is not None:
            # EAGLE token-conditioning shift (speculator_async.py:66-77): token j is conditioned on the target activation of
            # position j-1 -- drop every sequence's first token and its last activation
            rows, off = [], 0
            for ids in token_lists:
                rows.append(acts[off:off + len(ids) - 1])
                off += len(ids)
            acts = torch.cat(rows, dim=0)
            token_lists = [ids[1:] for ids in token_lists]
        self.link.prefill(token_lists, [list(s.draft_block_table) for s in seqs], eagle_acts=acts)
        return SpeculateResult([], [])

    def speculate(self, seqs, verify_result: VerifyResult) -> SpeculateResult:
        K = self.lookahead
        keys, nts, tables, temps = [], [], [], []
        for seq in seqs:
            assert seq.recovery_token_id is not None
            seq.append_token(seq.recovery_token_id)
            # (seq_id, accepted draft tokens of the previous step, recovery token); -2 on the first step => miss
            keys.append((seq.seq_id, seq.last_spec_step_accepted_len - 1, seq.recovery_token_id))
            nts.append(seq.num_tokens)
            tables.append(list(seq.draft_block_table))
            temps.append(seq.draft_temperature if seq.draft_temperature is not None else seq.temperature)
        # the target's temperature matters too: a greedy draft under a sampling target still takes the ratio path
        want = any(t > 0 for t in temps) or any(s.temperature > 0 for s in seqs)
        eagle = None
        if verify_result.eagle_acts is not None:       # speculator_async.py:158-179
            first = seqs[0].last_target_hidden_state
            acts = first.new_zeros(len(seqs), K + 1, first.shape[-1])
            counts, ext_ids = [], []
            for i, seq in enumerate(seqs):
                n = seq.extend_count if seq.extend_eagle_acts is not None else 0
                counts.append(n)
                ext_ids.append((list(seq.extend_token_ids[:n]) if n else []) + [0] * (K - n))
                if n:
                    acts[i, :n] = seq.extend_eagle_acts[:n]
                acts[i, K] = seq.last_target_hidden_state
            eagle = (counts, ext_ids, acts)
        hits, tokens, logits_q = self.link.speculate(keys, nts, tables, temps, want_logits=want, eagle=eagle)
        rows = []
        for seq, toks in zip(seqs, tokens):
            rows.append([seq.recovery_token_id] + toks)
            seq.token_ids.extend(toks)
            seq.num_tokens = len(seq.token_ids)
            seq.last_token = seq.token_ids[-1]
            seq.num_draft_cached_tokens += K + 1
        speculations = torch.tensor(rows, dtype=torch.int64).to(self.device)
        return SpeculateResult(speculations, logits_q, hits)
