"""Single-process entry for multi-GPU engines: ``LLM(model, num_gpus=N)`` called from ONE Python process, like the
reference (ssd/engine/llm_engine.py:61-127 spawns its tensor-parallel workers and the draft process itself).

The engine proper is SPMD -- one process per GPU, every tensor-parallel rank runs the same ``generate`` in lock step
(see llm_engine.py).  When the caller is not already inside such a launch (no WORLD_SIZE in the environment), rank 0
is the calling process and this module spawns ranks 1..N-1:
  * tensor-parallel followers rebuild the same engine and then wait on a gloo control group for the leader's
    ("generate", prompts, sampling_params) commands, which they replay verbatim (outputs are discarded);
  * the draft rank of asynchronous speculation serves the wire protocol until the leader sends CMD_EXIT.
The reference ships every runner call through a shared-memory pickle RPC (model_runner.py:404-428); here only the
user's request crosses the process boundary, once per ``generate``.
"""
from __future__ import annotations

import importlib
import os
import socket

import torch.multiprocessing as mp


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def resolve_factory(spec):
    """A runner factory may be given as "module:function" (called with no arguments to obtain the factory) so that it
    can cross the process boundary; callables are passed through (single-process use)."""
    if isinstance(spec, str):
        mod, fn = spec.split(":")
        return getattr(importlib.import_module(mod), fn)()
    return spec


def _worker_main(rank: int, world: int, port: int, model: str, kwargs: dict, env: dict) -> None:
    os.environ.update(env)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ssd_amd.engine.llm_engine import LLMEngine
    eng = LLMEngine(model, **kwargs)
    eng.follow()


def spawn_followers(model: str, kwargs: dict, world: int) -> list:
    """Make this process rank 0 of a `world`-rank launch and start the other ranks.  Returns the processes."""
    if "runner_factory" in kwargs and callable(kwargs["runner_factory"]):
        raise ValueError("with num_gpus > 1 from a single process, pass runner_factory as 'module:function'")
    port = free_port()
    env = {k: os.environ[k] for k in ("SSD_DIST_BACKEND", "SSD_LOCAL_DEVICE", "SSD_CUSTOM_AR", "HSA_ENABLE_IPC_MODE_LEGACY",
                                      "PYTHONPATH") if k in os.environ}
    ctx = mp.get_context("spawn")
    procs = []
    for rank in range(1, world):
        p = ctx.Process(target=_worker_main, args=(rank, world, port, model, kwargs, env), daemon=True)
        p.start()
        procs.append(p)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return procs
