"""Synchronous speculation: the draft model proposes K tokens, then the target verifies.

Same plug-in contract as the reference SpeculatorSync (ssd/engine/speculator_sync.py:8-69): ``speculate`` appends
the recovery token plus K lookahead positions to every sequence (rolled back by SpecDecodeStep), bumps
``num_draft_cached_tokens`` by K+1 and returns SpeculateResult(speculations [B,K+1], logits_q).

What differs is where the loop runs: the reference does K+1 graph replays with a ``.tolist()`` host sync after
each one; here ``draft_runner.speculate_chain`` enqueues the forwards back to back with the sampled token
fed forward on the device, so ``speculations`` comes back as a device tensor that nobody has read yet.  The K
not-yet-known draft tokens are represented in ``seq.token_ids`` by the placeholder -1 until verification
returns them (the state is restored before it could be observed).

The reference's (K+1)-th forward only deposits the KV of x_K (speculator_sync.py:55-56), which a later step reads
only if ALL K draft tokens were accepted.  It is therefore deferred: the chain runs K forwards, and the next
``speculate`` first runs that deposit for exactly the sequences whose previous round accepted everything
(``deposit_pending``).  Same KV values, same tokens; one seventh of the draft work gone whenever a rejection occurs.
"""
from __future__ import annotations

from ssd_amd.engine.speculate_types import SpeculateResult, SpeculatorBase, VerifyResult

PLACEHOLDER = -1


class SpeculatorSync(SpeculatorBase):
    def __init__(self, lookahead: int, device, draft_model_runner):
        super().__init__(lookahead, device)
        self.draft_model_runner = draft_model_runner

    def prefill(self, seqs, verify_result: VerifyResult) -> SpeculateResult:
        self.draft_model_runner.call("run", seqs, True)     # fills the draft KV; its sampled token is unused
        return SpeculateResult([], [])

    def speculate(self, seqs, verify_result: VerifyResult) -> SpeculateResult:
        K = self.lookahead
        recovery = []
        for i, seq in enumerate(seqs):
            if seq.recovery_token_id is None:
                raise ValueError(f"recovery_token_id is None for seq {i}")
            recovery.append(seq.recovery_token_id)
            seq.append_token(seq.recovery_token_id)
        # suffix = [recovery] + accepted tokens: K+1 committed tokens <=> x_K was accepted and its KV is still missing
        pending = [s for s in seqs if s.last_spec_step_accepted_len == K + 1]
        if pending:
            self.draft_model_runner.deposit_pending(pending)
        speculations = self.draft_model_runner.speculate_chain(seqs, recovery)
        sampled = any((s.draft_temperature if s.draft_temperature is not None else s.temperature) > 0 for s in seqs)
        for seq in seqs:
            for _ in range(K):
                seq.append_token(PLACEHOLDER)
            seq.num_draft_cached_tokens += K + 1
        # logits_q is only read on the temperature > 0 ratio path (ssd/utils/verify.py:50-64)
        return SpeculateResult(speculations, self.draft_model_runner.logits_q(len(seqs)) if sampled else None)

    def check_round(self) -> None:
        """Called by the step after the verify's host synchronisation: the draft chain of THIS round ran on the same stream, so the
        mirror of its segments' error word is current (model_runner.check_segments)."""
        check = getattr(self.draft_model_runner, "check_segments", None)
        if check is not None:
            check(False)
