"""Paged-KV block accounting with xxh64 prefix caching.

Behavioural restatement of the reference BlockManager (ssd/engine/block_manager.py:25-176): same hash chain
(xxh64 over the previous block's hash, little-endian 8 bytes, then the block's token ids as int64 bytes), same
allocation order (free list FIFO), same lookahead reservation rule, so block tables -- and therefore KV slot
numbers -- come out identical for identical request streams.
"""
from __future__ import annotations

from collections import deque

import numpy as np
import xxhash


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = -1
        self.token_ids: list[int] = []

    def update(self, h: int, token_ids: list[int]) -> None:
        self.hash, self.token_ids = h, token_ids

    def reset(self) -> None:
        self.ref_count, self.hash, self.token_ids = 1, -1, []


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int, is_draft: bool = False, speculate_k: int = -1,
                 max_model_len: int = -1, verbose: bool = False, prefix_cache: bool = True):
        """prefix_cache=False: every allocation takes fresh blocks (EAGLE-3: the draft's KV row p belongs to token p+1 and
        is conditioned on the target activation of position p, which a prefix hit would neither compute nor align)."""
        assert num_blocks > 0
        self.prefix_cache = prefix_cache
        self.block_size = block_size
        self.is_draft = is_draft
        self.speculate_k = speculate_k
        self.max_model_len = max_model_len
        self.verbose = verbose
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self.used_block_ids: set[int] = set()
        self.hash_to_block_id: dict[int, int] = {}

    # ---- hashing (reference block_manager.py:48-54) ----
    @staticmethod
    def compute_hash(token_ids: list[int], prefix: int = -1) -> int:
        h = xxhash.xxh64()
        if prefix != -1:
            h.update(prefix.to_bytes(8, "little"))
        h.update(np.array(token_ids).tobytes())
        return h.intdigest()

    # ---- helpers ----
    def _table(self, seq) -> list[int]:
        return seq.draft_block_table if self.is_draft else seq.block_table

    def _take(self, block_id: int) -> Block:
        blk = self.blocks[block_id]
        assert blk.ref_count == 0
        blk.reset()
        self.free_block_ids.remove(block_id)
        self.used_block_ids.add(block_id)
        return blk

    def _release(self, block_id: int) -> None:
        assert self.blocks[block_id].ref_count == 0
        self.used_block_ids.remove(block_id)
        self.free_block_ids.append(block_id)

    def unref(self, block_id: int) -> None:
        blk = self.blocks[block_id]
        blk.ref_count -= 1
        if blk.ref_count == 0:
            self._release(block_id)

    # ---- whole-sequence allocation with prefix reuse (reference :99-127) ----
    def can_allocate(self, seq) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq) -> None:
        table = self._table(seq)
        assert not table
        h, missed = -1, not self.prefix_cache
        for i in range(seq.num_blocks):
            toks = seq.block(i)
            h = self.compute_hash(toks, h) if len(toks) == self.block_size else -1
            hit = self.hash_to_block_id.get(h, -1)
            # a sequence whose EVERY block is cached (a preempted sequence of exactly n * block_size tokens coming back) would
            # leave nothing to prefill and no logits to sample from -- the reference schedules that empty prefill
            # (block_manager.py:99-127, scheduler.py:69-86); here the last full block is recomputed instead (same KV values)
            whole_seq_cached = i == seq.num_blocks - 1 and len(toks) == self.block_size and not missed      # every block before it hit
            if hit == -1 or self.blocks[hit].token_ids != toks or whole_seq_cached:
                missed = True
            if missed:
                blk = self._take(self.free_block_ids[0])
                hit = blk.block_id
            else:
                if self.is_draft:
                    seq.num_draft_cached_tokens += self.block_size
                else:
                    seq.num_cached_tokens += self.block_size
                if hit in self.used_block_ids:
                    blk = self.blocks[hit]
                    blk.ref_count += 1
                else:
                    blk = self._take(hit)
            if h != -1:
                blk.update(h, toks)
                self.hash_to_block_id[h] = hit
            table.append(hit)

    def deallocate(self, seq) -> None:
        table = self._table(seq)
        for block_id in reversed(table):
            self.unref(block_id)
        if self.is_draft:
            seq.num_draft_cached_tokens = 0
        else:
            seq.num_cached_tokens = 0
        table.clear()

    # ---- lookahead reservation for decode / speculation (reference :137-176) ----
    def _blocks_for(self, seq, lookahead: int) -> int:
        return -(-(seq.num_tokens + lookahead) // self.block_size)

    def can_append(self, seq, lookahead_num_tokens: int = 1) -> bool:
        if seq.num_tokens + lookahead_num_tokens > self.max_model_len:
            return False
        need = self._blocks_for(seq, lookahead_num_tokens) - len(self._table(seq))
        return need <= 0 or len(self.free_block_ids) >= need

    def may_append(self, seq, lookahead_num_tokens: int = 1) -> None:
        table = self._table(seq)
        need = self._blocks_for(seq, lookahead_num_tokens) - len(table)
        if need <= 0:
            return
        if len(self.free_block_ids) < need:
            raise RuntimeError(f"Insufficient free blocks: need {need}, have {len(self.free_block_ids)}")
        for _ in range(need):
            block_id = self.free_block_ids.popleft()
            blk = self.blocks[block_id]
            assert blk.ref_count == 0
            blk.reset()
            self.used_block_ids.add(block_id)
            table.append(block_id)

    def finalize_block(self, seq, table: list[int], block_index: int) -> None:
        """Hash a block that just became full (reference scheduler.py:234-241: the prefix is taken from the
        second-to-last table entry, the hash lands on the last one)."""
        toks = seq.block(block_index)
        if block_index == len(table) - 1:
            prefix = self.blocks[table[-2]].hash if len(table) > 1 else -1
            last = self.blocks[table[-1]]
        else:
            # a block whose hashing was postponed (deferred draft KV deposit, scheduler._commit_suffix) is no longer the
            # last one: chain from ITS predecessor and label IT
            prefix = self.blocks[table[block_index - 1]].hash if block_index > 0 else -1
            last = self.blocks[table[block_index]]
        h = self.compute_hash(toks, prefix)
        last.update(h, toks)
        self.hash_to_block_id[h] = last.block_id
