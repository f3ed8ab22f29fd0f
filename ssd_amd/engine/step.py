"""Inference steps: autoregressive and speculate->verify (reference ssd/engine/step.py:15-163)."""
from __future__ import annotations

from abc import ABC, abstractmethod

from ssd_amd.engine.speculate_types import SpeculatorBase, VerifierBase, VerifyResult


class InferenceStep(ABC):
    def __init__(self, scheduler):
        self.scheduler = scheduler

    @abstractmethod
    def decode(self, seqs) -> int: ...

    @abstractmethod
    def prefill(self, seqs) -> int: ...


class AutoRegressiveStep(InferenceStep):
    def __init__(self, scheduler, model_runner, tokenizer=None):
        super().__init__(scheduler)
        self.model_runner = model_runner
        self.tokenizer = tokenizer

    def _step(self, seqs, is_prefill: bool) -> int:
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        self.scheduler.postprocess(seqs, token_ids, is_prefill)
        return sum(len(s) for s in seqs) if is_prefill else len(seqs)

    def prefill(self, seqs) -> int:
        return self._step(seqs, True)

    def decode(self, seqs) -> int:
        return self._step(seqs, False)


class SpecDecodeStep(InferenceStep):
    def __init__(self, scheduler, speculator: SpeculatorBase, verifier: VerifierBase, eagle: bool = False,
                 tokenizer=None, async_spec: bool = False):
        super().__init__(scheduler)
        self.speculator, self.verifier = speculator, verifier
        self.eagle, self.tokenizer, self.async_spec = eagle, tokenizer, async_spec

    def prefill(self, seqs) -> int:
        if self.async_spec:       # draft (other GPU) and target prefill overlap (step.py:75-79)
            self.speculator.prefill(seqs, VerifyResult([], [], None))
            self.verifier.prefill(seqs)
        else:
            vr = self.verifier.prefill(seqs)
            self.speculator.prefill(seqs, vr)
        for seq in seqs:
            assert seq.recovery_token_id is not None
            seq.num_cached_tokens = seq.num_prompt_tokens
            seq.num_draft_cached_tokens = seq.num_prompt_tokens
            seq.last_spec_step_accepted_len = -1        # a (re)prefill leaves no deferred draft KV deposit behind
        return sum(len(s) for s in seqs)

    def decode(self, seqs) -> int:
        saved = [seq.snapshot() for seq in seqs]
        spec = self.speculator.speculate(seqs, VerifyResult([], [], None))
        out = self.verifier.verify(seqs, spec)
        for seq, snap in zip(seqs, saved):      # undo the lookahead applied by speculate + verify
            seq.restore(snap)
        self.scheduler.postprocess_speculate(seqs, out.new_suffixes, out.recovery_tokens)
        return sum(len(s) for s in out.new_suffixes)
