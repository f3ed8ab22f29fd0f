"""Which GPU / rank / role this process has.

One process per GPU.  Roles as in the reference (ssd/engine/llm_engine.py:61-93): ranks 0..TP-1 hold the
tensor-parallel target; with draft_async the LAST rank is the dedicated draft GPU.  The group layout mirrors
ssd/engine/model_runner.py:98-107 (tp group + a 2-rank async group), but the processes are launched SPMD
(``torchrun`` / ``python -m torch.distributed.run``) and rendezvous over 127.0.0.1.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, replace

import torch
import torch.distributed as dist


@dataclass
class Topology:
    rank: int
    world_size: int
    device: torch.device
    role: str                    # "target" | "draft"
    tp_rank: int
    tp_size: int
    tp_group: object = None
    async_group: object = None   # [rank 0, draft rank]
    draft_rank: int = -1
    ctl_group: object = None     # gloo group of the target ranks: control messages of a single-process launch
    draft_group: object = None   # draft data-parallel group (draft ranks only); dp_rank / dp_size = position in it
    dp_rank: int = 0
    dp_size: int = 1

    def single(self) -> "Topology":
        """Same device, no tensor parallelism (a sync-speculation draft is replicated on every rank)."""
        return replace(self, tp_rank=0, tp_size=1, tp_group=None)


def init_process_group_if_needed(backend: str | None = None) -> tuple[int, int]:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # SSD_DIST_BACKEND=gloo lets two ranks share ONE GPU in tests (RCCL refuses duplicate devices)
            backend = os.environ.get("SSD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(_local_device(rank))
        # every collective of the control plane gives up after SSD_PG_TIMEOUT_S (default 10 min: the slowest rank generates /
        # loads its weight shard before it reaches the first one) instead of the backend's 30 min -- a dead peer must surface
        from datetime import timedelta
        from ssd_amd.utils import watchdog
        watchdog.stage("process_group_init", watchdog.env_float("SSD_PG_TIMEOUT_S", 600.0) + 30.0)
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=timedelta(seconds=watchdog.env_float("SSD_PG_TIMEOUT_S", 600.0)))
    return rank, world


def _local_device(rank: int) -> int:
    """GPU index of this rank: LOCAL_RANK, unless SSD_LOCAL_DEVICE pins every rank to one GPU (single-GPU tests of
    the multi-rank paths, together with SSD_DIST_BACKEND=gloo)."""
    if "SSD_LOCAL_DEVICE" in os.environ:
        return int(os.environ["SSD_LOCAL_DEVICE"])
    return int(os.environ.get("LOCAL_RANK", str(rank)))


def resolve_topology(config, colocated_draft: bool = False) -> Topology:
    """colocated_draft: asynchronous speculation WITHOUT a dedicated draft GPU -- all ranks are tensor-parallel target
    ranks and TP rank 0 additionally hosts the draft server on a side stream (LLMEngine(inprocess_draft=True))."""
    rank, world = init_process_group_if_needed()
    local = _local_device(rank)
    device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    if world == 1:
        assert config.num_gpus == 1, (f"num_gpus={config.num_gpus} needs one process per GPU: launch with "
                                      f"`python -m torch.distributed.run --nproc-per-node {config.num_gpus} ...`")
        return Topology(0, 1, device, "target", 0, 1)
    assert world == config.num_gpus, f"WORLD_SIZE={world} but num_gpus={config.num_gpus}"
    from ssd_amd.utils import watchdog
    watchdog.stage("subgroup_creation (tp / async p2p / control / draft groups)")
    if config.speculate and config.draft_async and not colocated_draft:
        D = getattr(config, "num_draft_gpus", 1)
        tp = world - D
        assert tp >= 1, f"num_gpus={world} leaves no target rank beside {D} draft ranks"
        tp_group = dist.new_group(list(range(tp)))
        async_group = dist.new_group([0, tp])                      # head <-> draft leader
        ctl = dist.new_group(list(range(tp)), backend="gloo")
        draft_group = dist.new_group(list(range(tp, world))) if D > 1 else None
        if rank >= tp:
            return Topology(rank, world, device, "draft", 0, 1, None, async_group if rank == tp else None, tp,
                            draft_group=draft_group, dp_rank=rank - tp, dp_size=D)
        return Topology(rank, world, device, "target", rank, tp, tp_group, async_group if rank == 0 else None, tp, ctl)
    tp_group = dist.new_group(list(range(world)))
    ctl = dist.new_group(list(range(world)), backend="gloo")
    return Topology(rank, world, device, "target", rank, world, tp_group, ctl_group=ctl)
