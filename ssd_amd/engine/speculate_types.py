"""The reference's own plug-in interface for the hot loop, kept name-for-name
(ssd/engine/helpers/speculate_types.py:7-46): SpecDecodeStep talks to a SpeculatorBase and a VerifierBase
through SpeculateResult / VerifyResult."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any


@dataclass
class SpeculateResult:
    speculations: Any            # int64 [B, K+1] (column 0 = recovery token); a DEVICE tensor on the HIP path
    logits_q: Any                # [B, K, V] draft logits, or None at temperature 0 (never read by verify then)
    cache_hits: Any = None       # int64 [B] (async only)


@dataclass
class VerifyResult:
    new_suffixes: list           # per sequence: [recovery] + accepted draft tokens
    recovery_tokens: list        # per sequence: the next recovery token
    eagle_acts: Any = None


class SpeculatorBase(ABC):
    def __init__(self, lookahead: int, device):
        self.lookahead = lookahead
        self.device = device

    @abstractmethod
    def prefill(self, seqs, verify_result: VerifyResult) -> SpeculateResult: ...

    @abstractmethod
    def speculate(self, seqs, verify_result: VerifyResult) -> SpeculateResult: ...


class VerifierBase(ABC):
    def __init__(self, lookahead: int, device):
        self.lookahead = lookahead
        self.device = device

    @abstractmethod
    def prefill(self, seqs, eagle: bool = False) -> VerifyResult: ...

    @abstractmethod
    def verify(self, seqs, speculate_result: SpeculateResult, eagle: bool = False) -> VerifyResult: ...
