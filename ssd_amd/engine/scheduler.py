"""Admission, lookahead reservation, preemption and post-step bookkeeping (host side of the hot loop).

Behavioural restatement of the reference Scheduler (ssd/engine/scheduler.py:12-327): prefill has priority over
decode; decode reserves K+1 target positions and K+1 (sync) or K+1+K*MQ_LEN (async) draft positions per
sequence (scheduler.py:97-126, async_spec_helpers.py:6-7); KV pressure preempts the youngest running sequence
and re-queues it as a fresh prompt; postprocess_speculate truncates the accepted suffix at EOS /
max_new_tokens / max_model_len, returns over-reserved blocks, and hashes blocks that became full.
"""
from __future__ import annotations

from collections import deque

from ssd_amd.engine.block_manager import BlockManager
from ssd_amd.engine.sequence import Sequence, SequenceStatus


def megaspec_lookahead(mq_len: int, k: int) -> int:
    """Draft positions beyond the sequence used by one async round: glue (K+1) + K tree steps of MQ_LEN
    (reference ssd/utils/async_helpers/async_spec_helpers.py:6-7)."""
    return k + 1 + k * mq_len


class Scheduler:
    def __init__(self, config, draft_num_blocks: int | None = None):
        self.config = config
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.max_model_len = config.max_model_len
        self.eos = config.eos
        self.speculate = config.speculate
        self.draft_async = config.draft_async
        self.K = config.speculate_k
        self.block_size = config.kvcache_block_size
        self.MQ_LEN = sum(config.fan_out_list) if (config.speculate and config.draft_async) else 0
        pc = not getattr(config, "use_eagle", False)
        self.block_manager = BlockManager(config.num_kvcache_blocks, self.block_size, is_draft=False,
                                          max_model_len=self.max_model_len, prefix_cache=pc)
        self.draft_block_manager = None
        if self.speculate:
            assert draft_num_blocks is not None and draft_num_blocks > 0
            self.draft_block_manager = BlockManager(draft_num_blocks, self.block_size, is_draft=True,
                                                    speculate_k=self.K, max_model_len=self.max_model_len, prefix_cache=pc)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()
        self._warned_cap = False
        self.capped: list[Sequence] = []        # finished by schedule() itself: see _length_capped

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    # ---- lookahead lengths (reference scheduler.py:97-107) ----
    def lookaheads(self) -> tuple[int, int | None]:
        if not self.speculate:
            return 1, None
        if self.draft_async:
            return self.K + 1, megaspec_lookahead(self.MQ_LEN, self.K)
        return self.K + 1, self.K + 1

    def _can_append(self, seq, tgt: int, dft: int | None) -> bool:
        ok = self.block_manager.can_append(seq, tgt)
        if self.speculate:
            ok = ok and self.draft_block_manager.can_append(seq, dft)
        return ok

    def _can_allocate(self, seq) -> bool:
        return self.block_manager.can_allocate(seq) and (not self.speculate or self.draft_block_manager.can_allocate(seq))

    # ---- one scheduling decision ----
    def schedule(self) -> tuple[list[Sequence], bool]:
        picked: list[Sequence] = []
        budget = 0
        while self.waiting:
            seq = self.waiting[0]
            fresh = len(seq) - seq.num_cached_tokens
            if budget + fresh > self.max_num_batched_tokens or not self._can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            if self.speculate:
                self.draft_block_manager.allocate(seq)
            budget += fresh
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            picked.append(seq)
        if picked:
            return picked, True

        tgt, dft = self.lookaheads()
        while self.running and len(picked) < self.max_num_seqs:
            seq = self.running.popleft()
            if self._length_capped(seq, tgt, dft):
                continue
            admitted = True
            while not self._can_append(seq, tgt, dft):
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    admitted = False
                    break
            if admitted:
                self.block_manager.may_append(seq, tgt)
                if self.speculate:
                    self.draft_block_manager.may_append(seq, dft)
                picked.append(seq)
        self.running.extendleft(reversed(picked))
        return picked, False

    def _length_capped(self, seq: Sequence, tgt: int, dft: int | None) -> bool:
        """A running sequence whose next step would reserve positions past max_model_len can never be scheduled again, however
        many blocks are free: it is finished here, at the length it has.  (The reference has no such exit: can_append refuses
        on the model-length guard (block_manager.py:137-147), the sequence preempts itself, and the step runs an empty batch
        -- `max() arg is an empty sequence` in prepare_block_tables_from_seqs.)  In speculative modes the cap therefore sits
        one lookahead (K+1 target positions; K+1+K*MQ_LEN draft positions under async speculation) below max_model_len."""
        need = max(tgt, dft or 0)
        if seq.num_tokens + need <= self.max_model_len:
            return False
        seq.status = SequenceStatus.FINISHED
        seq.finish_reason = "max_model_len"          # not max_new_tokens / EOS: the caller can tell a capped request from a complete one
        if not self._warned_cap:
            self._warned_cap = True
            import sys
            print(f"[ssd_amd] a request was finished at {seq.num_tokens} tokens: its next step would reserve {need} positions past "
                  f"max_model_len = {self.max_model_len} (speculation lookahead); raise max_model_len to generate further",
                  file=sys.stderr, flush=True)
        self.block_manager.deallocate(seq)
        if self.speculate:
            self.draft_block_manager.deallocate(seq)
        self.capped.append(seq)
        return True

    def pop_capped(self) -> list[Sequence]:
        out, self.capped = self.capped, []
        return out

    def preempt(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        seq.recovery_token_id = None
        self.block_manager.deallocate(seq)
        if self.speculate:
            self.draft_block_manager.deallocate(seq)
        self.waiting.appendleft(seq)
        seq.num_prompt_tokens = seq.num_tokens      # completions so far are re-prefilled as prompt
        seq.last_spec_step_accepted_len = -1
        seq.extend_count, seq.extend_eagle_acts, seq.extend_token_ids = 0, None, None     # scheduler.py:143-146

    # ---- autoregressive post-step (reference scheduler.py:149-170) ----
    def postprocess(self, seqs: list[Sequence], token_ids: list[int], is_prefill: bool) -> None:
        bm = self.block_manager
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            if is_prefill:
                seq.num_cached_tokens = seq.num_prompt_tokens
            else:
                seq.num_cached_tokens += 1
            done = (not seq.ignore_eos and tok == self.eos) or seq.num_completion_tokens == seq.max_new_tokens
            if done:
                seq.status = SequenceStatus.FINISHED
                bm.deallocate(seq)
                self.running.remove(seq)
            elif seq.last_block_num_tokens == self.block_size:
                bm.finalize_block(seq, seq.block_table, seq.num_blocks - 1)

    # ---- speculative post-step (reference scheduler.py:172-327) ----
    def _clip_suffix(self, seq: Sequence, suffix: list[int]) -> tuple[list[int], bool]:
        if not seq.ignore_eos and self.eos in suffix:
            suffix = suffix[:suffix.index(self.eos) + 1]
        if seq.num_completion_tokens + len(suffix) >= seq.max_new_tokens:
            suffix = suffix[:seq.max_new_tokens - seq.num_completion_tokens]
        if seq.num_tokens + len(suffix) > self.max_model_len:
            suffix = suffix[:max(0, self.max_model_len - seq.num_tokens)]
        n = len(suffix)
        finished = ((not seq.ignore_eos and self.eos in suffix)
                    or seq.num_completion_tokens + n == seq.max_new_tokens
                    or seq.num_tokens + n >= self.max_model_len)
        return suffix, finished

    def _return_excess_blocks(self, seq: Sequence, suffix_len: int) -> None:
        needed = -(-(seq.num_tokens + suffix_len) // self.block_size)
        for bm, table_name in ((self.block_manager, "block_table"), (self.draft_block_manager, "draft_block_table")):
            table = getattr(seq, table_name)
            extra = len(table) - needed
            if extra > 0:
                for block_id in table[-extra:]:
                    bm.unref(block_id)
                setattr(seq, table_name, table[:-extra])

    def _commit_suffix(self, seq: Sequence, suffix: list[int], recovery: int) -> None:
        n = len(suffix)
        assert n >= 1
        seq.token_ids.extend(suffix)
        seq.num_tokens += n
        seq.last_token = suffix[-1]
        seq.num_cached_tokens += n
        seq.num_draft_cached_tokens += n
        seq.last_spec_step_accepted_len = n
        seq.recovery_token_id = recovery
        assert seq.block_table and seq.draft_block_table
        # synchronous speculation defers the draft's KV deposit of x_K (engine/speculator_sync.py): after a fully
        # accepted round the draft KV row of position num_tokens-1 does not exist yet, so a draft block ending exactly
        # there must not become a prefix-cache entry now (it is hashed by a later commit, once the row is written)
        draft_valid = seq.num_tokens - (1 if (not self.draft_async and n == self.K + 1) else 0)
        for idx in range(len(seq.block_table)):
            if (idx + 1) * self.block_size <= seq.num_tokens:
                if self.block_manager.blocks[seq.block_table[idx]].hash == -1:
                    self.block_manager.finalize_block(seq, seq.block_table, idx)
                if (idx + 1) * self.block_size <= draft_valid and self.draft_block_manager.blocks[seq.draft_block_table[idx]].hash == -1:
                    self.draft_block_manager.finalize_block(seq, seq.draft_block_table, idx)

    def postprocess_speculate(self, seqs: list[Sequence], new_suffixes: list[list[int]], next_recovery_tokens: list[int],
                              eagle_acts=None) -> None:
        for i, (seq, suffix, rec) in enumerate(zip(seqs, new_suffixes, next_recovery_tokens)):
            suffix, finished = self._clip_suffix(seq, suffix)
            self._return_excess_blocks(seq, len(suffix))
            self._commit_suffix(seq, suffix, rec)
            if eagle_acts is not None:          # scheduler.py:303-320; eagle_acts [B, K+1, taps * h_target]
                n = len(suffix)                 # suffix = [recovery, accepted draft tokens...]
                seq.last_target_hidden_state = eagle_acts[i, min(n - 1, eagle_acts.shape[1] - 1)]
                n_ext = min(n - 1, self.K)
                seq.extend_count = n_ext
                seq.extend_eagle_acts = eagle_acts[i, :n_ext].clone() if n_ext > 0 else None
                seq.extend_token_ids = list(suffix[1:1 + n_ext]) if n_ext > 0 else None
            if finished:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.draft_block_manager.deallocate(seq)
                self.running.remove(seq)
