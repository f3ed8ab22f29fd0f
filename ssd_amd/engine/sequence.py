"""Per-request state.  Field names and meanings follow the reference (ssd/engine/sequence.py:14-120) because the
scheduler / speculator / verifier contract is expressed in them: ``num_cached_tokens`` and
``num_draft_cached_tokens`` count KV positions valid in the target / draft cache, ``recovery_token_id`` is the
token sampled by the last verify (or prefill) that is NOT yet in ``token_ids``, and
``last_spec_step_accepted_len`` (-1 before the first step) feeds the async speculation-cache key."""
from __future__ import annotations

from enum import Enum, auto
from itertools import count

from ssd_amd.sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    counter = count()       # process-global, monotonically increasing seq ids (reference sequence.py:15,28)
    block_size = 256        # set by the engine from Config.kvcache_block_size

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams | None = None):
        sp = sampling_params or SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = self.num_tokens
        self.num_cached_tokens = 0
        self.num_draft_cached_tokens = 0
        self.block_table: list[int] = []
        self.draft_block_table: list[int] = []
        self.last_spec_step_accepted_len = -1
        self.recovery_token_id: int | None = None
        # the prefill's token once LLMEngine.generate handed it to the stream ahead of the first speculation round; consumed
        # (cleared) by the append that puts it into the sequence -- a LATER preemption must not resurrect it
        self.first_token_streamed: int | None = None
        self.temperature = sp.temperature
        self.draft_temperature = sp.draft_temperature
        self.max_new_tokens = sp.max_new_tokens
        self.ignore_eos = sp.ignore_eos
        # EAGLE-3 (reference sequence.py:23-24,47-51): the target activation that conditions the next recovery token, and
        # the accepted draft tokens + their target activations that the draft re-deposits ("extends") at the next glue
        self.last_target_hidden_state = None
        self.extend_eagle_acts = None
        self.extend_token_ids: list[int] | None = None
        self.extend_count = 0

    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def num_cached_blocks(self) -> int:
        return -(-self.num_cached_tokens // self.block_size)

    @property
    def num_draft_cached_blocks(self) -> int:
        return -(-self.num_draft_cached_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_cached_blocks - 1) * self.block_size

    @property
    def last_block_num_tokens_draft(self) -> int:
        return self.num_tokens - (self.num_draft_cached_blocks - 1) * self.block_size

    def block(self, i: int) -> list[int]:
        assert 0 <= i < self.num_blocks
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1
        self.first_token_streamed = None

    # -- speculation bookkeeping: the (K+1)-token lookahead is applied and rolled back around one
    #    speculate+verify round exactly as SpecDecodeStep.decode does (ssd/engine/step.py:97-145) --
    def snapshot(self):
        return (len(self.token_ids), self.num_tokens, self.last_token, self.num_draft_cached_tokens, self.num_cached_tokens)

    def restore(self, snap) -> None:
        n, self.num_tokens, self.last_token, self.num_draft_cached_tokens, self.num_cached_tokens = snap
        del self.token_ids[n:]
