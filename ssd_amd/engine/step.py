"""Inference steps: autoregressive and speculate->verify (reference ssd/engine/step.py:15-163)."""
from __future__ import annotations

from abc import ABC, abstractmethod

from ssd_amd.engine.speculate_types import SpeculatorBase, VerifierBase, VerifyResult
from ssd_amd.utils import profiling as prof


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
        if self.async_spec and not self.eagle:       # draft (other GPU) and target prefill overlap (step.py:75-79)
            self.speculator.prefill(seqs, VerifyResult([], [], None))
            self.verifier.prefill(seqs)
        else:                      # an EAGLE draft prefills from the target's activations: target first (step.py:80-82)
            vr = self.verifier.prefill(seqs, eagle=self.eagle)
            self.speculator.prefill(seqs, vr)
        for seq in seqs:
            assert seq.recovery_token_id is not None
            seq.num_cached_tokens = seq.num_prompt_tokens
            seq.num_draft_cached_tokens = seq.num_prompt_tokens
            seq.last_spec_step_accepted_len = -1        # a (re)prefill leaves no deferred draft KV deposit behind
        return sum(len(s) for s in seqs)

    def decode(self, seqs) -> int:
        trace = prof.enabled("SSD_PROFILE")           # reference step.py:92-161
        t0 = prof.sync_now() if trace else 0.0
        saved = [seq.snapshot() for seq in seqs]
        spec = self.speculator.speculate(seqs, VerifyResult([], [], True if self.eagle else None))
        t1 = prof.sync_now() if trace else 0.0
        out = self.verifier.verify(seqs, spec, eagle=self.eagle)
        # (the verify's read-back synchronised the stream the draft chain ran on: a resident segment that gave up a wait while
        #  producing THIS round's speculations is reported now, before the accepted tokens are committed)
        check = getattr(self.speculator, "check_round", None)
        if check is not None:
            check()
        t2 = prof.sync_now() if trace else 0.0
        for seq, snap in zip(seqs, saved):      # undo the lookahead applied by speculate + verify
            seq.restore(snap)
        self.scheduler.postprocess_speculate(seqs, out.new_suffixes, out.recovery_tokens,
                                             eagle_acts=out.eagle_acts if self.eagle else None)
        toks = sum(len(s) for s in out.new_suffixes)
        if trace:
            t3 = prof.sync_now()
            hits = spec.cache_hits
            hl = None if hits is None else (hits if isinstance(hits, list) else hits.tolist())
            hs = f"hits={sum(hl)}/{len(hl)} " if hl is not None else ""
            print(f"[PROFILE target] handshake={(t1 - t0) * 1e3:.2f}ms verify={(t2 - t1) * 1e3:.2f}ms "
                  f"postprocess={(t3 - t2) * 1e3:.2f}ms total={(t3 - t0) * 1e3:.2f}ms {hs}toks={toks}", flush=True)
        return toks
