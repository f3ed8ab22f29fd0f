"""SamplingParams -- same fields and defaults as the reference (ssd/sampling_params.py:4-9)."""
from dataclasses import dataclass


@dataclass
class SamplingParams:
    temperature: float = 1.0
    draft_temperature: float | None = None
    max_new_tokens: int = 256
    ignore_eos: bool = False
