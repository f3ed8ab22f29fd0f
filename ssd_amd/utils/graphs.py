"""hipGraph capture with the Python garbage collector held off.

torch.cuda.graph() no longer sweeps cyclic garbage before a capture (torch >= 2.9: only with force_cudagraph_gc), so a
generation-2 collection can start in the middle of one and run the destructors of hipGraphs, streams or RCCL communicators
that earlier, already dropped engines still own through reference cycles -- calls HIP refuses while a stream is capturing,
from a destructor, i.e. std::terminate.  Collect first, then keep the collector off until the capture has ended."""
from __future__ import annotations

import gc
from contextlib import contextmanager

import torch


@contextmanager
def capture(graph: "torch.cuda.CUDAGraph", **kw):
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()
