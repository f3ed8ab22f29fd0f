"""The reference's tracing switches, same names and line formats:
  SSD_PROFILE=1         synchronise + wall-clock around handshake / verify / postprocess of every speculation step
                        (reference ssd/engine/step.py:92-161 "[PROFILE target] ..."; verifier.py:56-113 "[PROFILE verify] ...")
  SSD_PROFILE_TARGET=1  the target call of a verification (reference verifier.py:63-74 "[PROFILE verifier] ...")
  SSD_PROFILE_DRAFT=1   per-round device-event log of the draft server, flushed once per round so that the round itself
                        carries no extra synchronisation (reference draft_runner.py:880-915, cudagraph_helpers.py:129-148)
They cost nothing when unset (one dict lookup per step).  Kernel-level evidence comes from rocprofv3 (profiles/collect_r02.sh)."""
from __future__ import annotations

import os
from time import perf_counter

import torch


def enabled(flag: str) -> bool:
    return os.environ.get(flag, "0") == "1"


def sync_now() -> float:
    """Device-synchronised wall clock (what the reference's torch.cuda.synchronize(); perf_counter() pairs measure)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return perf_counter()


class EventLog:
    """Named spans of one draft round on the current stream: HIP events on the GPU (no host sync when recorded),
    wall clock on CPU runners; `flush` turns them into one line."""

    def __init__(self, tag: str):
        self.tag = tag
        self.spans: list[tuple[str, object, object]] = []
        self.gpu = torch.cuda.is_available()

    def mark(self):
        if self.gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return perf_counter()

    def span(self, name: str, start, end) -> None:
        self.spans.append((name, start, end))

    def flush(self, extra: str = "") -> str | None:
        if not self.spans:
            return None
        parts = []
        for name, a, b in self.spans:
            if self.gpu:
                b.synchronize()
                ms = a.elapsed_time(b)
            else:
                ms = (b - a) * 1e3
            parts.append(f"{name}={ms:.2f}ms")
        self.spans = []
        line = f"[PROFILE {self.tag}] " + " ".join(parts) + (" " + extra if extra else "")
        print(line, flush=True)
        return line
