"""The draft side of asynchronous speculation ("SSD"): a server that answers speculation requests from a
speculation cache and, while the target verifies, pre-computes the next cache by decoding a TREE of
continuations -- one branch per (accepted-length j, recovery token) outcome the verifier could return.

Restates the reference DraftRunner loop (ssd/engine/draft_runner.py:186-286 hit_cache_and_respond, :288-378
_service_spec_request, :530-711 _build_tree_batch, :713-812 _decode_tree, :814-830 _populate_tree_cache,
:859-928 draft_loop) over a backend-neutral draft runner (HIP on the GPU, the oracle in CPU tests) with four
operations: draft_prefill, draft_jit, draft_glue_fork, draft_tree.

Geometry, with P = num_tokens - 1 = position of the recovery token:
  JIT (cache miss)  K single-token decodes at positions P .. P+K-1
  glue              K+1 tokens [rec, x_1..x_K] at P .. P+K (re-deposits the trunk KV, yields the fork logits)
  fork              top-F_j tokens per glue position j, excluding x_{j+1} for j < K  -> MQ_LEN branches
  tree step d       branch i feeds its token at RoPE position P+j_i+1+d, KV slot P+K+1+d*MQ_LEN+i
  cache             key (seq_id, j_i, fork token_i) -> K continuation tokens of branch i
The speculation cache lives on the draft device -- tokens AND keys (fork tokens, sequence ids, glue positions); a request is
looked up there (csrc/misc.hip ssd_cache_lookup) and answered before the glue/tree work of the new round starts, so that work
overlaps the target's verify.

EAGLE-3 drafts (config.use_eagle; reference branches at draft_runner.py:133-177,230-285,312-331,530-612,660-711):
every draft position is shifted by -1 (KV row p holds token p+1, conditioned on the target's activation of position
p), the request carries the recovery token's target activation plus the accepted draft tokens of the last round with
THEIR target activations ("extend" rows, re-deposited by the glue in place of the self-conditioned rows the tree
wrote), and the cache keeps each branch's K prenorm vectors next to its K tokens: they condition the spec rows of the
next glue.
"""
from __future__ import annotations

import torch

from ssd_amd.engine import async_proto as P
from ssd_amd.utils import profiling as prof


def branch_positions(fan_out: list[int]) -> list[int]:
    """Glue position j of each branch: [0]*F_0 + [1]*F_1 + ... (repeat_interleave, draft_runner.py:118-119)."""
    return [j for j, f in enumerate(fan_out) for _ in range(f)]


class DraftGroup:
    """Draft data-parallelism (BASELINE.json configs[4]: "draft x4 data-parallel"; the reference has it on its roadmap
    only, README.md:129-130).  The D draft ranks all keep the trunk KV (prefill + glue are replicated: one K+1-token
    forward), but the MQ_LEN branches of the speculation tree -- K forwards of MQ_LEN rows, the bulk of a round -- and
    the speculation cache built from them are SHARDED: member d owns a contiguous slice of the branches.  Only the
    leader (member 0) talks to the target; per request it broadcasts the payload to the group, every member looks the
    key up in its own cache shard, the members all-gather (hit, tokens) -- a few dozen int64 -- and the owner's answer
    goes back through the leader.  Every member then knows the answered tokens, which seed the next round's glue."""

    def __init__(self, group, dp_rank: int, dp_size: int, device):
        import torch.distributed as dist
        self.group, self.rank, self.size = group, dp_rank, dp_size
        self.wire = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
        self.leader = dist.get_global_rank(group, 0)
        self.dist = dist

    def bcast_ints(self, values, n: int) -> list[int]:
        t = torch.tensor(values if self.rank == 0 else [0] * n, dtype=torch.int64).to(self.wire)
        self.dist.broadcast(t, src=self.leader, group=self.group)
        return t.tolist()

    def allgather_ints(self, values: list[int]) -> list[list[int]]:
        t = torch.tensor(values, dtype=torch.int64).to(self.wire)
        out = [torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.tolist() for o in out]

    def bcast_tensor(self, t: torch.Tensor, src_member: int) -> torch.Tensor:
        dev = t.device
        w = t.to(self.wire).contiguous()
        self.dist.broadcast(w, src=self.dist.get_global_rank(self.group, src_member), group=self.group)
        return w.to(dev)

    def shard(self, n: int) -> tuple[int, int]:
        """Contiguous slice [lo, hi) of n branches owned by this member."""
        per = -(-n // self.size)
        return min(n, self.rank * per), min(n, (self.rank + 1) * per)


class DraftServer:
    def __init__(self, config, runner, transport, stream=None, deferred: bool = False, dp: DraftGroup | None = None):
        """stream / deferred: co-located mode (the server shares the target's process and GPU).  All draft work is
        issued on `stream` so that it overlaps the target's verify on the device, and the work that follows the reply
        (glue + fork + tree decode) is parked until the target has launched its verify (`run_deferred`, called from
        ModelRunner.verify_chain just before it blocks), instead of delaying that launch."""
        self.config, self.runner, self.tx = config, runner, transport
        self.stream, self.deferred = stream, deferred
        self.dp = dp                        # draft data-parallel group (None: a single draft GPU); transport is None off the leader
        self._parked = None
        self.K = config.speculate_k
        self.mq = config.MQ_LEN
        self.max_blocks = config.max_blocks
        self.j_hit = branch_positions(config.fan_out_list)
        self.j_miss = branch_positions(config.fan_out_list_miss)
        self._cache_index = None            # device (sequence ids [B], glue positions [B, W]) of pending_forks, built on first use
        self.cache_tokens = None            # device tensor [N, K]
        self.cache_logits = None            # device tensor [N, K, V] when the last tree was sampled (temperature > 0)
        self.eagle = bool(getattr(config, "use_eagle", False))
        self.cache_acts = None              # EAGLE: device tensor [N, K, h] of the branches' prenorm vectors
        assert not (self.eagle and dp is not None), "EAGLE-3 drafts are not data-parallel"
        self.pending_forks = None           # device tensor [B, MQ] of the round whose keys are not mirrored yet
        self.pending_meta = None
        self.stats = {"requests": 0, "hits": 0, "rounds": 0}
        # SSD_PROFILE_DRAFT=1: device events around the reply and the two halves of the next round, printed at the NEXT
        # request (reference draft_runner.py:880-915): the round itself gets no extra synchronisation
        self._trace = prof.EventLog("draft") if prof.enabled("SSD_PROFILE_DRAFT") else None

    # ---- one command ----
    def run_deferred(self) -> None:
        work, self._parked = self._parked, None
        if work is not None:
            self._on_stream(work)

    def reset(self) -> None:
        """Forget the parked round and the speculation cache (LLMEngine.abort_all: their sequences are gone)."""
        self._parked = None
        self.pending_forks = self.pending_meta = self._cache_index = None
        self.cache_tokens = self.cache_logits = self.cache_acts = None

    def _on_stream(self, fn):
        if self.stream is None:
            return fn()
        with torch.cuda.stream(self.stream):
            return fn()

    def handle_one(self) -> bool:
        self.run_deferred()          # nothing may overtake a parked round
        return self._on_stream(self._handle_one)

    def _handle_one(self) -> bool:
        dp = self.dp
        lead = dp is None or dp.rank == 0
        hdr = self.tx.recv_ints(P.HEADER_LEN) if lead else None
        if dp is not None:
            hdr = dp.bcast_ints(hdr, P.HEADER_LEN)
        cmd, B, n, flags = hdr
        if cmd == P.CMD_EXIT:
            return False
        if cmd == P.CMD_HELLO:
            blocks = self.runner.num_kvcache_blocks
            if dp is not None:          # every member allocates its own cache: the scheduler may use what the smallest has
                blocks = min(v[0] for v in dp.allgather_ints([blocks]))
            if lead:
                self.tx.send_ints([blocks])
            return True
        payload = (self.tx.recv_ints(n) if n else []) if lead else None
        if dp is not None and n:
            payload = dp.bcast_ints(payload, n)
        if cmd == P.CMD_PREFILL:
            toks, tables = P.unpack_prefill(payload, B, self.max_blocks)
            if flags & P.FLAG_EAGLE:        # draft_runner.py:72-78: one activation row per (shifted) prompt token
                acts = self.tx.recv_tensor((sum(len(t) for t in toks), self.runner.cfg.eagle_taps * self.runner.cfg.d_model_target),
                                           torch.bfloat16)
                self.runner.draft_prefill(toks, tables, acts)
            else:
                self.runner.draft_prefill(toks, tables)
            return True
        if cmd == P.CMD_SPECULATE:
            self._speculate(B, payload, flags)
            return True
        raise RuntimeError(f"draft server: unknown command {cmd}")

    def serve_forever(self) -> None:
        while self.handle_one():
            pass

    # ---- speculation round ----
    def _check_segments(self, sync: bool) -> None:
        check = getattr(self.runner, "check_segments", None)      # (oracle / EAGLE runners have no resident segments)
        if check is not None:
            check(sync)

    def _lookup(self, keys) -> tuple[list[int], torch.Tensor | None]:
        """Request keys against the cache of the finished round, ON THE DRAFT DEVICE (reference draft_runner.py:215-252): the
        fork tokens never leave it; what comes back is one int32 per request -- the entry index or -1 -- which the host needs
        anyway to choose between replying from the cache and running the JIT chain (the reference syncs on `cache_hits.all()`
        at the same point, :246)."""
        if self.pending_forks is None:
            return [-1] * len(keys), None
        if self._cache_index is None:
            self._cache_index = self.runner.cache_index(*self.pending_meta)
        idx_dev = self.runner.cache_lookup(keys, self._cache_index[0], self._cache_index[1], self.pending_forks)
        return idx_dev.tolist(), idx_dev

    def _speculate(self, B: int, payload: list[int], flags: int) -> None:
        K = self.K
        dp = self.dp
        lead = dp is None or dp.rank == 0
        eagle = None
        if flags & P.FLAG_EAGLE:
            keys, num_tokens, tables, temps, ext_counts, ext_ids = P.unpack_speculate(payload, B, self.max_blocks, K)
            cfg = self.runner.cfg
            acts = self.tx.recv_tensor((B, K + 1, cfg.eagle_taps * cfg.d_model_target), torch.bfloat16)
            # every draft position is one behind the sequence (pos_offset = -1, draft_runner.py:133-135,409,497): the
            # runner's geometry P = num_tokens - 1 becomes num_tokens - 2 by handing it num_tokens - 1
            num_tokens = [n - 1 for n in num_tokens]
            eagle = dict(acts=acts, ext_counts=ext_counts, ext_ids=ext_ids)
        else:
            keys, num_tokens, tables, temps = P.unpack_speculate(payload, B, self.max_blocks)
        if self._trace is not None:
            self._trace.flush(f"round={self.stats['rounds']} hits={self.stats['hits']}/{self.stats['requests']}")
            t_req = self._trace.mark()
        want_logits = bool(flags & P.FLAG_WANT_LOGITS)
        sample = any(t > 0 for t in temps)
        idx, idx_dev = self._lookup(keys)                                # rows of MY cache shard (-1: miss)
        # (the lookup's read-back synchronised the draft's stream: the glue + tree round whose cache is about to be served has
        #  finished -- a resident segment that gave up a wait while building it is reported before any of its tokens is used)
        self._check_segments(False)
        hits = [1 if i >= 0 else 0 for i in idx]
        owner = [0 if h else -1 for h in hits]                           # group member that holds each row's branch
        if dp is not None:
            # (hit, K tokens) per sequence from every member: the first member with a hit owns the row (keys are unique
            # across shards -- top-F picks distinct tokens per position -- so there is at most one)
            mine = []
            rows = self.cache_tokens[torch.tensor([i if i >= 0 else 0 for i in idx], dtype=torch.int64, device=self.cache_tokens.device)].tolist() \
                if (self.cache_tokens is not None and any(hits)) else [[0] * K] * B
            for b in range(B):
                mine.extend([hits[b]] + (list(rows[b]) if hits[b] else [0] * K))
            everyone = dp.allgather_ints(mine)
            owner, merged = [], []
            for b in range(B):
                o = next((d for d in range(dp.size) if everyone[d][b * (K + 1)] == 1), -1)
                owner.append(o)
                merged.append(everyone[o][b * (K + 1) + 1:(b + 1) * (K + 1)] if o >= 0 else [0] * K)
            hits = [1 if o >= 0 else 0 for o in owner]
        self.stats["requests"] += B
        self.stats["hits"] += sum(hits)
        rec = [k[2] for k in keys]
        logits_q = None
        jit = self.config.jit_speculate
        serve_from_cache = (any(hits) and not jit) or (all(hits) and jit)
        dev = self.runner.zeros_tokens(1, 1).device
        if serve_from_cache:
            if dp is None:
                rows = idx_dev.clamp_min(0).to(torch.int64)         # stays on the device
                tokens = self.cache_tokens[rows]
                if not all(hits):       # "fast" backup: miss rows carry filler tokens (the reference uses random ones)
                    tokens = tokens * P.to_device(hits, torch.int64, tokens.device).unsqueeze(1)
                if want_logits and self.cache_logits is not None:
                    logits_q = self.cache_logits[rows]          # [B, K, V]: the q the hit branch was sampled from
                if eagle is not None:
                    eagle["prev_acts"] = self.cache_acts[rows]  # [B, K, h]: the hit branches' prenorms (draft_runner.py:248-249)
            else:
                tokens = torch.tensor(merged, dtype=torch.int64, device=dev)
                if want_logits and sample:      # the owner's branch logits travel to the leader (and on to the target)
                    V = self.runner.cfg.vocab_size
                    logits_q = torch.zeros(B, K, V, dtype=torch.bfloat16, device=dev)
                    for b in range(B):
                        if owner[b] < 0:
                            continue
                        src = self.cache_logits[idx[b]] if (owner[b] == dp.rank and self.cache_logits is not None) else logits_q[b]
                        logits_q[b] = dp.bcast_tensor(src.contiguous(), owner[b])
        elif jit:
            if eagle is not None:       # the chain starts from fc(recovery activation) and conditions on itself (:133-177)
                tokens = self.runner.draft_jit(rec, num_tokens, tables, temps, cond=eagle["acts"][:, K])
                eagle["prev_acts"] = self.runner.jit_acts(B)
            elif lead:
                tokens = self.runner.draft_jit(rec, num_tokens, tables, temps)  # [B, K] on the draft device
                self._check_segments(True)      # the chain segment: its tokens leave with the reply below (misses only: one word read back)
                if want_logits and sample:
                    logits_q = self.runner.logits_q(B)
            else:
                tokens = self.runner.zeros_tokens(B, K)
            if dp is not None:          # the chain ran on the leader only; its tokens seed every member's glue
                tokens = dp.bcast_tensor(tokens, 0)
        else:
            tokens = self.runner.zeros_tokens(B, K)
        if lead:
            resp = torch.cat([P.to_device(hits, torch.int64, tokens.device), tokens.reshape(-1)])
            self.tx.send_tensor(resp)
            if want_logits:
                # rows that are neither hits nor JIT-drafted never take the ratio path (verify.py:57-62): zeros will do
                if logits_q is None:
                    logits_q = torch.zeros(B, K, self.runner.cfg.vocab_size, dtype=torch.bfloat16, device=tokens.device)
                self.tx.send_tensor(logits_q)
        if self._trace is not None:
            self._trace.span("lookup_reply" if serve_from_cache or not jit else "jit_reply", t_req, self._trace.mark())
        # ---- from here on the target is verifying; pre-compute the next round's cache ----
        def next_round():
            t_a = self._trace.mark() if self._trace is not None else None
            fan = [self.config.fan_out_list if h else self.config.fan_out_list_miss for h in hits]
            jl = [self.j_hit if h else self.j_miss for h in hits]
            glue_ids = torch.cat([P.to_device(rec, torch.int64, tokens.device).unsqueeze(1), tokens], dim=1)
            if eagle is not None:
                forks = self.runner.draft_glue_fork(glue_ids, num_tokens, tables, fan, eagle=eagle)
            else:
                forks = self.runner.draft_glue_fork(glue_ids, num_tokens, tables, fan)      # [B, MQ] (replicated under DP)
            t_b = self._trace.mark() if self._trace is not None else None
            if dp is not None:          # my slice of the branches: K tree steps of MQ/D rows instead of MQ
                lo, hi = dp.shard(self.mq)
                forks = forks[:, lo:hi].contiguous()
                jl = [j[lo:hi] for j in jl]
            if eagle is not None:
                self.cache_tokens = self.runner.draft_tree(forks, num_tokens, tables, jl, temps, eagle=True)
                self.cache_acts = self.runner.tree_acts(forks.numel())
            else:
                self.cache_tokens = self.runner.draft_tree(forks, num_tokens, tables, jl, temps)  # [B*mq', K]
            self.cache_logits = self.runner.tree_logits(forks.numel()) if sample else None
            self.pending_forks = forks
            self.pending_meta = ([k[0] for k in keys], jl)
            self._cache_index = self.runner.cache_index(*self.pending_meta)      # off the critical path: the target is verifying
            self.stats["rounds"] += 1
            if self._trace is not None:
                self._trace.span("glue_fork", t_a, t_b)
                self._trace.span(f"tree[{forks.shape[1]}x{self.K}]", t_b, self._trace.mark())
        self.pending_forks = None           # the finished round's cache is consumed by this request
        self._cache_index = None
        self.cache_tokens = None
        self.cache_acts = None
        if self.deferred:
            self._parked = next_round
        else:
            next_round()
