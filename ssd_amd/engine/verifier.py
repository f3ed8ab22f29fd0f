"""Target-side verification (reference Verifier, ssd/engine/verifier.py:12-153, and verify(),
ssd/utils/verify.py:5-181, greedy branch): one K+1-query target forward per step; accept/reject runs on the
device (wave ballot + prefix count) and the step's whole result comes back in one packed copy."""
from __future__ import annotations

from time import perf_counter

from ssd_amd.engine.speculate_types import SpeculateResult, VerifierBase, VerifyResult
from ssd_amd.utils import profiling as prof


class Verifier(VerifierBase):
    def __init__(self, lookahead: int, device, target_model_runner, sampler_x=None, async_fan_out=None,
                 jit_speculate: bool = False, tokenizer=None, metrics: dict | None = None):
        super().__init__(lookahead, device)
        self.target_model_runner = target_model_runner
        self.sampler_x, self.async_fan_out, self.jit_speculate = sampler_x, async_fan_out, jit_speculate
        self.tokenizer = tokenizer
        self.metrics = metrics if metrics is not None else {}

    def prefill(self, seqs, eagle: bool = False) -> VerifyResult:
        token_ids = self.target_model_runner.call("run", seqs, True)
        acts = None
        if eagle:       # verifier.py:34-46: the tapped activations of every prompt token; the last one conditions the recovery token
            for s in seqs:
                assert s.num_cached_tokens == 0, "EAGLE-3 needs the activations of the whole prompt (no prefix-cache hits)"
            acts = self.target_model_runner.eagle_acts(sum(len(s) for s in seqs)).clone()
        off = 0
        for seq, tok in zip(seqs, token_ids):
            # already handed to the stream (LLMEngine.generate) and not yet appended: a re-prefill (preemption before the
            # first round) must not flip it.  The pin is cleared by Sequence.append_token, so a sequence preempted AFTER its
            # first round -- whose completion count restarts at 0 -- takes the freshly computed token
            pinned = seq.first_token_streamed
            seq.recovery_token_id = tok if pinned is None else pinned
            if eagle:
                off += len(seq)
                seq.last_target_hidden_state = acts[off - 1]
        return VerifyResult([], [seq.recovery_token_id for seq in seqs], acts)

    def verify(self, seqs, speculate_result: SpeculateResult, eagle: bool = False) -> VerifyResult:
        temps_t = [float(s.temperature) for s in seqs]
        temps_q = [float(s.draft_temperature) if s.draft_temperature is not None else float(s.temperature) for s in seqs]
        t0 = perf_counter()
        if any(t > 0 for t in temps_t + temps_q):
            # rows whose draft tokens really came from q: JIT speculation, or a speculation-cache hit (verify.py:57-62)
            hits = speculate_result.cache_hits
            hl = None if hits is None else (hits if isinstance(hits, list) else hits.tolist())
            ratio = [1 if (self.jit_speculate or (hl is not None and hl[b] == 1)) else 0 for b in range(len(seqs))]
            new_suffixes, recovery = self.target_model_runner.verify_chain(seqs, speculate_result.speculations,
                                                                           logits_q=speculate_result.logits_q,
                                                                           temps_q=temps_q, ratio_rows=ratio)
        else:
            new_suffixes, recovery = self.target_model_runner.verify_chain(seqs, speculate_result.speculations)
        for seq in seqs:
            seq.num_cached_tokens += self.lookahead + 1
        if prof.enabled("SSD_PROFILE_TARGET") or prof.enabled("SSD_PROFILE"):     # reference verifier.py:63-74,111-113
            # forward + accept/reject are ONE device graph here and its result read is the step's only host sync, so
            # the reference's separate target_fwd / verify_compute spans collapse into a single span
            print(f"[PROFILE verifier] target_call={(prof.sync_now() - t0) * 1e3:.2f}ms eagle={eagle} bs={len(seqs)}", flush=True)
        self.metrics.setdefault("target_verify_times", []).append(perf_counter() - t0)
        self.metrics.setdefault("accepted_suffix_lens_with_recovery", []).extend(len(s) for s in new_suffixes)
        hits = speculate_result.cache_hits
        if hits is not None:
            hl = hits if isinstance(hits, list) else hits.tolist()
            self.metrics.setdefault("cache_hits", []).append(sum(hl) / max(1, len(hl)))
            for h, s in zip(hl, new_suffixes):
                self.metrics.setdefault("accepted_suffix_lens_on_hit" if h == 1 else "accepted_suffix_lens_on_miss", []).append(len(s))
        acts = None
        if eagle:       # verifier.py:145-147
            B = len(seqs)
            acts = self.target_model_runner.eagle_acts(B * (self.lookahead + 1)).clone().view(B, self.lookahead + 1, -1)
        return VerifyResult(new_suffixes, recovery, acts)
