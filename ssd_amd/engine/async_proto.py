"""Wire protocol between the target (TP rank 0) and the dedicated draft GPU of asynchronous speculation (SSD).

Same information as the reference exchanges (ssd/engine/speculator_async.py:130-187,
ssd/engine/draft_runner.py:51-101,288-378), re-framed so that each request is two messages instead of
three-plus-logits:

  header  int64[4]   = (cmd, B, payload_len, flags)          cmd: 0 speculate, 1 prefill, 2 exit, 3 hello
  payload int64[payload_len]
     speculate: keys[B,3] (seq_id, accepted_draft_tokens j, recovery token) | num_tokens[B] |
                draft block tables[B, max_blocks] (-1 padded) | temperature bits[B]
     prefill:   token ids (all sequences, concatenated) | num_tokens[B] | draft block tables[B, max_blocks]
  reply (speculate only) int64[B + B*K] = cache_hits[B] | tokens[B,K]
     + bf16 logits_q[B,K,V] only when flags & FLAG_WANT_LOGITS (some temperature > 0): at temperature 0 verify()
       never reads logits_q (ssd/utils/verify.py:50-64,130), so the reference's 1.8 MB per-step transfer is skipped.
  hello reply int64[1] = number of draft KV blocks (the reference passes this through an mp.Queue,
       llm_engine.py:96-105).
  EAGLE-3 (flags & FLAG_EAGLE; reference speculator_async.py:66-89,158-179, draft_runner.py:72-78,312-331): the target's
     tapped activations ride ONE bf16 tensor after the payload instead of the reference's three messages --
     prefill:   acts [sum of tokens, A]   (token j of the shifted prompt with the activation of position j-1)
     speculate: payload += extend_counts[B] | extend_token_ids[B, K];  acts [B, K+1, A] = K extend rows (zero padded)
                followed by the recovery token's activation

Transports: torch.distributed p2p on the 2-rank async group (RCCL over one xGMI hop on the GPU path, gloo in the
CPU tests), or an in-process loopback used by single-GPU tests.
"""
from __future__ import annotations

import struct
from collections import deque

import torch
import torch.distributed as dist

CMD_SPECULATE, CMD_PREFILL, CMD_EXIT, CMD_HELLO = 0, 1, 2, 3
FLAG_WANT_LOGITS = 1
FLAG_EAGLE = 2
HEADER_LEN = 4


def temp_bits(t: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(t)))[0]


def bits_temp(b: int) -> float:
    return struct.unpack("<f", struct.pack("<i", int(b)))[0]


def pack_speculate(keys: list[tuple[int, int, int]], num_tokens: list[int], block_tables: list[list[int]],
                   temps: list[float], max_blocks: int, extend_counts: list[int] | None = None,
                   extend_ids: list[list[int]] | None = None) -> list[int]:
    out: list[int] = []
    for k in keys:
        out.extend(k)
    out.extend(num_tokens)
    for bt in block_tables:
        assert len(bt) <= max_blocks
        out.extend(bt + [-1] * (max_blocks - len(bt)))
    out.extend(temp_bits(t) for t in temps)
    if extend_counts is not None:
        out.extend(extend_counts)
        for row in extend_ids:
            out.extend(row)
    return out


def unpack_speculate(payload: list[int], B: int, max_blocks: int, K: int | None = None):
    """K given: the payload carries the EAGLE extend block; returns two more values (counts [B], ids [B][K])."""
    off = 0
    keys = [tuple(payload[off + 3 * b: off + 3 * b + 3]) for b in range(B)]
    off += 3 * B
    num_tokens = payload[off:off + B]
    off += B
    tables = [payload[off + b * max_blocks: off + (b + 1) * max_blocks] for b in range(B)]
    off += B * max_blocks
    temps = [bits_temp(x) for x in payload[off:off + B]]
    off += B
    if K is not None:
        counts = payload[off:off + B]
        off += B
        ids = [payload[off + b * K: off + (b + 1) * K] for b in range(B)]
        off += B * K
        assert off == len(payload)
        return keys, num_tokens, tables, temps, counts, ids
    assert off == len(payload)
    return keys, num_tokens, tables, temps


def pack_prefill(token_lists: list[list[int]], block_tables: list[list[int]], max_blocks: int) -> list[int]:
    out: list[int] = []
    for t in token_lists:
        out.extend(t)
    out.extend(len(t) for t in token_lists)
    for bt in block_tables:
        out.extend(bt + [-1] * (max_blocks - len(bt)))
    return out


def unpack_prefill(payload: list[int], B: int, max_blocks: int):
    n_tail = B + B * max_blocks
    total = len(payload) - n_tail
    lens = payload[total:total + B]
    assert sum(lens) == total
    toks, off = [], 0
    for n in lens:
        toks.append(payload[off:off + n])
        off += n
    tables = [payload[total + B + b * max_blocks: total + B + (b + 1) * max_blocks] for b in range(B)]
    return toks, tables


class DistTransport:
    """p2p over a torch.distributed group (device tensors for nccl/RCCL, CPU tensors for gloo)."""

    def __init__(self, group, peer: int, device: torch.device):
        self.group, self.peer, self.device = group, peer, device
        # gloo moves host memory only: when two ranks share one GPU in tests (SSD_DIST_BACKEND=gloo) the messages are
        # staged through the CPU; on the deployment path (nccl = RCCL) device tensors go over xGMI directly
        self.wire = torch.device("cpu") if dist.get_backend(group) == "gloo" else device

    def send_ints(self, values: list[int]) -> None:
        dist.send(torch.tensor(values, dtype=torch.int64).to(self.wire), dst=self.peer, group=self.group)

    def recv_ints(self, n: int) -> list[int]:
        t = torch.empty(n, dtype=torch.int64, device=self.wire)
        dist.recv(t, src=self.peer, group=self.group)
        return t.tolist()

    def send_tensor(self, t: torch.Tensor) -> None:
        dist.send(t.contiguous().to(self.wire), dst=self.peer, group=self.group)

    def recv_tensor(self, shape, dtype) -> torch.Tensor:
        t = torch.empty(shape, dtype=dtype, device=self.wire)
        dist.recv(t, src=self.peer, group=self.group)
        return t.to(self.device)


def to_device(values, dtype, device) -> torch.Tensor:
    """Small host list -> device tensor WITHOUT blocking the host on the device stream.  `torch.tensor(list, device="cuda")`
    is a pageable H2D copy: the host waits until everything queued on the current stream -- e.g. a whole glue + tree round
    -- has finished.  A pinned source (torch's caching host allocator keeps the block alive until the copy's stream has
    passed it) + non_blocking copy returns at once."""
    t = torch.tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class LoopbackTransport:
    """Both endpoints in one process: messages are queued; a receive on the target side first lets the draft
    server consume everything pending (``pump``).  Used to exercise the draft server on a single GPU."""

    def __init__(self):
        self.to_peer: deque = deque()
        self.from_peer: deque = deque()
        self.pump = None
        self.peer_endpoint: "LoopbackTransport | None" = None

    @staticmethod
    def pair():
        a, b = LoopbackTransport(), LoopbackTransport()
        q_ab, q_ba = deque(), deque()
        a.to_peer, a.from_peer = q_ab, q_ba
        b.to_peer, b.from_peer = q_ba, q_ab
        a.peer_endpoint, b.peer_endpoint = b, a
        return a, b

    def send_ints(self, values):
        self.to_peer.append(list(values))

    def recv_ints(self, n):
        if not self.from_peer and self.pump is not None:
            self.pump()
        v = self.from_peer.popleft()
        assert len(v) == n, (len(v), n)
        return v

    def send_tensor(self, t):
        t = t.clone()
        if t.is_cuda:       # the endpoints may issue work on different streams: hand over finished data only
            torch.cuda.current_stream(t.device).synchronize()
        self.to_peer.append(t)

    def recv_tensor(self, shape, dtype):
        if not self.from_peer and self.pump is not None:
            self.pump()
        t = self.from_peer.popleft()
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype
        return t
