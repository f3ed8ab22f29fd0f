"""One model on one MI355X: weights, paged KV cache, static input buffers, hipGraph capture / replay.

Counterpart of the reference ModelRunner (ssd/engine/model_runner.py:37-680) and of the tensor preparation in
ssd/engine/helpers/runner_helpers.py:50-180 and the graph helpers in ssd/engine/helpers/cudagraph_helpers.py.
Re-designed for this hardware rather than translated:
  * SPMD tensor parallelism: every rank runs the same Python step on identical small metadata, so there is no
    shared-memory RPC (reference model_runner.py:404-428); ranks only meet in RCCL collectives.
  * the K+1 single-token draft forwards of synchronous speculation are chained ON THE DEVICE (argmax ->
    ssd_draft_advance -> next replay), and verification ends in ssd_verify_greedy writing one packed row per
    sequence: one host sync per speculation step instead of K+5.
  * decode / verify / tree forwards are captured once per batch size as hipGraphs over static buffers
    (torch.cuda.CUDAGraph == hipGraph); a prefill shape that keeps coming back replays a graph too (the reference
    prefills eagerly, model_runner.py:602).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from ssd_amd.hip import ops as H
from ssd_amd.hip.lib import load_library
from ssd_amd.model import HipDecoder, AttnMeta
from ssd_amd.model_config import ModelConfig
from ssd_amd import weights as W
from ssd_amd.utils.graphs import capture
from ssd_amd.engine.async_proto import to_device


class ModelRunner:
    PREFILL_GRAPHS = 4              # prefill shapes kept as hipGraphs
    PREFILL_GRAPH_MAX_T = 1024      # longer prefills are GPU-bound by far: launch overhead does not matter

    def __init__(self, config, model_cfg: ModelConfig, *, is_draft: bool, device: torch.device, tp_rank: int = 0,
                 tp_size: int = 1, tp_group=None, model_path: str | None = None, weights_seed: int = 0,
                 gen_device: str | None = None, num_kvcache_blocks: int = -1, memory_utilization: float | None = None,
                 max_decode_tokens: int | None = None, weight_source=None, force_collectives: bool = False,
                 custom_ar: bool | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("ssd_amd.ModelRunner needs an MI355X; there is no CPU fallback on the product path")
        load_library()  # fail loudly before allocating anything
        self.config, self.cfg = config, model_cfg
        self.is_draft, self.device = is_draft, device
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        self.block_size = config.kvcache_block_size
        self.max_blocks = config.max_blocks
        self.K = config.speculate_k if config.speculate else 0
        self.max_bs = config.max_num_seqs
        # prefill admits by token budget, not by max_num_seqs (reference scheduler.py:69-86): per-sequence buffers hold
        # seq_cap sequences and larger prefill batches are run in slices
        self.seq_cap = max(self.max_bs, 16)
        torch.cuda.set_device(device)

        mq = config.MQ_LEN if (config.speculate and config.draft_async) else 0
        dec_tokens = self.max_bs * max(1, self.K + 1, mq)
        self.max_decode_tokens = max_decode_tokens or dec_tokens
        max_tokens = max(config.max_num_batched_tokens, self.max_decode_tokens)
        eagle = bool(getattr(config, "use_eagle", False))
        model_cls = HipDecoder
        if model_cfg.family == "eagle3":
            from ssd_amd.eagle import HipEagleDraft as model_cls
        # EAGLE-3 target: tap the residual stream entering config.eagle_layers on every forward (reference llama3.py:256-271)
        taps = list(config.eagle_layers) if (eagle and not is_draft) else None
        self.model = model_cls(model_cfg, max_tokens=max_tokens, max_seqs=self.max_bs, max_blocks=self.max_blocks,
                               block_size=self.block_size, max_model_len=config.max_model_len, device=device,
                               tp_rank=tp_rank, tp_size=tp_size, tp_group=tp_group,
                               max_logit_rows=max(self.max_decode_tokens, self.seq_cap),
                               max_split_tokens=max(256, self.max_decode_tokens), force_collectives=force_collectives, taps=taps)
        if weight_source is not None:
            src = weight_source
        elif model_path is not None and W.has_safetensors(model_path) and model_cfg.family == "eagle3":
            import os
            src = W.load_eagle_safetensors(model_cfg, model_path, target_dir=config.model if os.path.isdir(config.model) else None,
                                           out_device=str(device))
        elif model_path is not None and W.has_safetensors(model_path):
            src = W.load_safetensors(model_cfg, model_path, tp_rank, tp_size, out_device=str(device))
        else:
            gd = gen_device or ("cuda" if model_cfg.hidden_size >= 1024 else "cpu")
            src = W.synthetic_weights(model_cfg, weights_seed, config.weights_std, tp_rank, tp_size, gen_device=gd,
                                      out_device=str(device), recipe=getattr(config, "weights_recipe", None))
        from ssd_amd.utils import watchdog
        watchdog.stage(f"weights ({'draft' if is_draft else 'target'})")
        self.model.load_weights(src)

        # ---- KV cache: free * utilisation // block_bytes (reference model_runner.py:446-492) ----
        if num_kvcache_blocks <= 0:
            free, _ = torch.cuda.mem_get_info(device)
            util = memory_utilization if memory_utilization is not None else config.gpu_memory_utilization
            num_kvcache_blocks = int(free * util) // self.model.kv_block_bytes()
            if tp_size > 1:   # all ranks must agree (SPMD schedulers)
                watchdog.stage("kv_blocks_agree (first collective on the tp group)")
                t = torch.tensor([num_kvcache_blocks], dtype=torch.int64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=tp_group)
                num_kvcache_blocks = int(t.item())
        assert num_kvcache_blocks > 0, "KV cache does not fit"
        self.num_kvcache_blocks = num_kvcache_blocks
        self.model.alloc_kv(num_kvcache_blocks)

        # ---- static inputs (device) + pinned host staging ----
        T, B = max_tokens, self.seq_cap
        dev = dict(device=device)
        self.d_ids = torch.zeros(T, dtype=torch.int64, **dev)
        self.d_pos = torch.zeros(T, dtype=torch.int64, **dev)
        self.d_slots = torch.zeros(T, dtype=torch.int32, **dev)
        self.d_ctx = torch.zeros(B, dtype=torch.int32, **dev)
        self.d_cu_q = torch.zeros(B + 1, dtype=torch.int32, **dev)
        self.d_gather = torch.zeros(B, dtype=torch.int32, **dev)
        self.d_bt = torch.zeros(B, self.max_blocks, dtype=torch.int32, **dev)
        self.d_next = torch.zeros(max(self.max_decode_tokens, B), dtype=torch.int64, **dev)
        self.d_spec = torch.zeros(B, self.K + 1, dtype=torch.int64, **dev)
        self.d_step = torch.zeros(1, dtype=torch.int32, **dev)
        self.d_accept = torch.zeros(B, dtype=torch.int32, **dev)
        self.d_recovery = torch.zeros(B, dtype=torch.int64, **dev)
        self.d_packed = torch.zeros(B, self.K + 3, dtype=torch.int64, **dev)
        self.h_packed = torch.zeros(B, self.K + 3, dtype=torch.int64).pin_memory()
        self.h_next = torch.zeros(max(self.max_decode_tokens, B), dtype=torch.int64).pin_memory()
        self._chain_err_host = torch.zeros(1, dtype=torch.int32).pin_memory()     # mirror of model.chain_err (_post_chain_err)
        self._setup_custom_ar(custom_ar)
        watchdog.stage("engine_init")           # (whatever the set-up's outcome: its stage limits end here)
        self._stage: dict = {}
        self._stage_set = 0
        self._ctx_hint = 4096
        self.graphs: dict = {}
        self._prefill_seen: dict = {}      # prefill shapes met once (run eagerly); a shape is captured when it comes back
        self.margin_log = None      # tests: set to {} to record (seq_id, position) -> top-2 margin of every greedy decision (TP = 1)
        self.graph_pool = None
        self.stream = torch.cuda.Stream(device)

    def _setup_custom_ar(self, want: bool | None) -> None:
        """Enable the one-shot all-reduce for the small TP sums when (a) tensor parallelism is on, (b) it is not
        disabled (SSD_CUSTOM_AR=0), and (c) EVERY rank's throw-away validation helper succeeded -- a fault in the IPC /
        peer-access path can then only kill a helper, never this process.  Otherwise RCCL carries all collectives."""
        import os
        m = self.model
        # why the one-shot collective is (not) in use: reported by bench.py's `collective` object -- a fall-back to RCCL must
        # be visible in the numbers' provenance, never silent
        self.custom_ar_status = "not applicable (no tensor-parallel collectives on this runner)"
        if not m.use_coll or self.is_draft:
            return
        if want is None:
            want = os.environ.get("SSD_CUSTOM_AR", "1") != "0"
        if not want:
            self.custom_ar_status = "disabled (SSD_CUSTOM_AR=0 / custom_ar=False): RCCL"
            return
        from ssd_amd.utils import custom_ar as CA
        world = dist.get_world_size(self.tp_group)
        from ssd_amd.utils import watchdog
        try:
            if world > 1:
                port = int(os.environ.get("MASTER_PORT", "29531")) + 17
                watchdog.stage("one_shot_allreduce_validation (helper processes)", 260.0)
                ok = CA.validate_in_subprocess(self.tp_rank, world, self.device.index or 0, port)
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.tp_group)
                if int(flag.item()) != 1:
                    self.custom_ar_status = ("self-validation FAILED on " + ("this rank" if not ok else "another rank")
                                             + ": fell back to RCCL")
                    return
            watchdog.stage("one_shot_allreduce_ipc_exchange (hipIpc handles)", 120.0)
            m.custom_ar = CA.OneShotAllReduce(self.tp_group, self.device)
            watchdog.stage("engine_init")
            arch = getattr(torch.cuda.get_device_properties(self.device), "gcnArchName", "?")
            proto = "granules <= 64 Ki elements, flag protocol above" if m.custom_ar.granules else "flag protocol"
            self.custom_ar_status = (f"validated on every rank of this node ({arch}; {proto})" if world > 1
                                     else f"single rank (nothing to validate; {arch}; {proto})")
        except Exception as e:
            m.custom_ar = None
            self.custom_ar_status = f"set-up raised {type(e).__name__}: {e}: fell back to RCCL"

    def _ensure_stochastic(self) -> None:
        """Buffers of the temperature > 0 path (allocated on first use: the benchmark configs are greedy)."""
        if hasattr(self, "d_rng"):
            return
        B, K, V = self.seq_cap, max(self.K, 1), self.cfg.vocab_size
        dev = dict(device=self.device)
        self.d_rng = torch.tensor([0x5D5D0000 + (7 if self.is_draft else 3) + 1000 * self.config.weights_seed], dtype=torch.int64, **dev)
        self.d_temps = torch.zeros(B, dtype=torch.float32, **dev)       # temperatures this model samples with
        self.d_temps_q = torch.zeros(B, dtype=torch.float32, **dev)     # (target) the draft's temperatures
        self.d_ratio = torch.zeros(B, dtype=torch.int32, **dev)
        self.d_lse_p = torch.zeros(B * (K + 1), dtype=torch.float32, **dev)
        self.d_lse_q = torch.zeros(B * K, dtype=torch.float32, **dev)
        if self.is_draft:
            self.d_logits_q = torch.zeros(B, K, V, dtype=torch.bfloat16, **dev)
        # sampler_x (config.py): the F+1 most probable draft tokens of a row are re-weighted (async_spec_helpers.py:79-105)
        self.sx = self.config.sampler_x
        self.sx_k = (self.config.async_fan_out + 1) if self.sx is not None else 0
        if self.sx is not None:
            rows = max(B * K, self.max_bs * getattr(self.config, "MQ_LEN", 1))
            self.d_boost = torch.zeros(rows, self.sx_k, dtype=torch.int32, **dev)

    def _check_collectives(self) -> None:
        ar = self.model.custom_ar
        if ar is not None and ar.failed():
            raise RuntimeError("one-shot all-reduce timed out waiting for a peer; results of this step are invalid "
                               "(rerun with SSD_CUSTOM_AR=0 to use RCCL)")
        if self._has_chain_err() and (int(self._chain_err_host[0]) or int(self.model.chain_err.item())):
            self._raise_chain_err()

    # The resident layer segments (csrc/chain.hip, csrc/tree_segment.hip) report a given-up bounded wait in a device word.  The
    # draft-side paths never read tokens on the host themselves (speculate_chain / draft_jit / draft_glue_fork / draft_tree /
    # deposit_pending return device tensors), so each of them mirrors the word into pinned host memory behind its launches
    # (_post_chain_err: one 4-byte async copy), and the word is looked at IN THE ROUND IT BELONGS TO (round 6; until round 5: on
    # entry of the next call, one round late) -- by `check_segments` at the first host synchronisation that follows the launches and
    # precedes any use of their tokens: after the verify's read-back (synchronous speculation, step.py), after the speculation-cache
    # lookup's read-back (the tree round whose cache is about to be served, draft_runner.py), and with a read of the device word
    # itself before a JIT chain's reply leaves the draft (the one path without a host sync of its own; misses only).
    def _has_chain_err(self) -> bool:
        return getattr(self.model, "chain_seg", False) or getattr(self.model, "tree_seg", False)

    def _raise_chain_err(self):
        raise RuntimeError("a bounded wait inside a resident layer segment (csrc/chain.hip / csrc/tree_segment.hip) gave up; the draft "
                           "forwards since the last check are invalid (rerun with SSD_CHAIN_SEG=0 SSD_TREE_SEG=0 for the separate launches)")

    def _post_chain_err(self) -> None:
        if self._has_chain_err():
            self._chain_err_host.copy_(self.model.chain_err, non_blocking=True)

    def _check_chain_host(self) -> None:
        if self._has_chain_err() and int(self._chain_err_host[0]):
            self._raise_chain_err()

    def check_segments(self, sync: bool = False) -> None:
        """Raise if a resident segment launched by this runner gave up a wait.  sync=False: the caller has synchronised the stream
        since the launches (the pinned mirror is current); sync=True: read the device word (a stream sync)."""
        if not self._has_chain_err():
            return
        if int(self._chain_err_host[0]) or (sync and int(self.model.chain_err.item())):
            self._raise_chain_err()

    # ---------------------------------------------------------------------------------------------
    # host -> device input staging
    # ---------------------------------------------------------------------------------------------
    def _table(self, seq):
        return seq.draft_block_table if self.is_draft else seq.block_table

    def _slot(self, table, pos: int) -> int:
        return table[pos // self.block_size] * self.block_size + pos % self.block_size

    def _upload(self, dst: torch.Tensor, values, dtype) -> None:
        """Host list -> persistent pinned staging -> async H2D.  Every step ends in a stream sync before the
        next one stages, so a staging buffer is never rewritten while its copy is in flight -- except for work that
        is queued WITHOUT a sync before the next staging (deposit_pending): such callers switch `_stage_set`, which
        selects a second set of pinned buffers."""
        n = len(values)
        if n:
            key = (id(dst), self._stage_set)
            stage = self._stage.get(key)
            if stage is None:
                stage = self._stage[key] = torch.zeros(dst.shape, dtype=dst.dtype).pin_memory()
            stage[:n] = torch.tensor(values, dtype=dtype)
            dst[:n].copy_(stage[:n], non_blocking=True)

    def _upload_block_tables(self, seqs) -> None:
        rows = []
        for s in seqs:
            t = self._table(s)
            assert len(t) <= self.max_blocks
            rows.append(t + [-1] * (self.max_blocks - len(t)))
        self._upload(self.d_bt, rows, torch.int32)

    def _prepare_prefill(self, seqs) -> tuple[int, int]:
        """prepare_prefill_tensors_from_seqs (runner_helpers.py:123-180): new tokens of each sequence, their
        positions and slots; context length = whole sequence (keys already cached by a prefix hit included)."""
        ids, pos, slots, ctx, cu, gather = [], [], [], [], [0], []
        max_q = 0
        for s in seqs:
            cached = s.num_draft_cached_tokens if self.is_draft else s.num_cached_tokens
            n = len(s)
            assert cached < n, "fully cached prompt: nothing to prefill"
            table = self._table(s)
            ids.extend(s[cached:])
            pos.extend(range(cached, n))
            slots.extend(self._slot(table, p) for p in range(cached, n))
            ctx.append(n)
            cu.append(cu[-1] + n - cached)
            gather.append(cu[-1] - 1)
            max_q = max(max_q, n - cached)
        T = len(ids)
        assert T <= self.d_ids.numel(), "prefill exceeds max_num_batched_tokens"
        self._note_ctx(max(ctx))
        self._upload(self.d_ids, ids, torch.int64)
        self._upload(self.d_pos, pos, torch.int64)
        self._upload(self.d_slots, slots, torch.int32)
        self._upload(self.d_ctx, ctx, torch.int32)
        self._upload(self.d_cu_q, cu, torch.int32)
        self._upload(self.d_gather, gather, torch.int32)
        self._upload_block_tables(seqs)
        return T, max_q

    def _prepare_decode(self, seqs, back: int = 0) -> int:
        """Single-token decode inputs (runner_helpers.py:59-75).  back = 1 feeds the token BEFORE the last one (the
        deferred KV deposit of a fully accepted draft chain, see deposit_pending)."""
        ids, pos, slots, ctx = [], [], [], []
        for s in seqs:
            cached = s.num_draft_cached_tokens if self.is_draft else s.num_cached_tokens
            assert cached == len(s) - 1, "decode expects exactly one uncached token"
            p = len(s) - 1 - back
            ids.append(s.last_token if back == 0 else s[p])
            pos.append(p)
            ctx.append(p + 1)
            slots.append(self._slot(self._table(s), p))
        self._note_ctx(max(ctx) + self.K + 1)
        self._upload(self.d_ids, ids, torch.int64)
        self._upload(self.d_pos, pos, torch.int64)
        self._upload(self.d_slots, slots, torch.int32)
        self._upload(self.d_ctx, ctx, torch.int32)
        self._upload_block_tables(seqs)
        return len(seqs)

    def _prepare_verify(self, seqs, ids_from_seq: bool) -> int:
        """K+1 query tokens per sequence at positions pos0 .. pos0+K (runner_helpers.py:77-96)."""
        K = self.K
        ids, pos, slots, ctx = [], [], [], []
        for s in seqs:
            pos0 = s.num_tokens - (K + 1)
            cached = s.num_draft_cached_tokens if self.is_draft else s.num_cached_tokens
            assert cached == pos0, f"num_cached_tokens={cached} != pos0={pos0}"
            table = self._table(s)
            if ids_from_seq:
                ids.extend(s[pos0:])
            pos.extend(range(pos0, pos0 + K + 1))
            slots.extend(self._slot(table, p) for p in range(pos0, pos0 + K + 1))
            ctx.append(len(s))
        self._note_ctx(max(ctx))
        if ids_from_seq:
            self._upload(self.d_ids, ids, torch.int64)
        self._upload(self.d_pos, pos, torch.int64)
        self._upload(self.d_slots, slots, torch.int32)
        self._upload(self.d_ctx, ctx, torch.int32)
        self._upload_block_tables(seqs)
        return len(seqs) * (K + 1)

    # ---------------------------------------------------------------------------------------------
    # graph bodies (all inputs already in the static buffers)
    # ---------------------------------------------------------------------------------------------
    def _meta(self, kind: str, B: int, max_q: int = 1, tree_step: int = 0) -> AttnMeta:
        hint = self._ctx_hint
        if kind == "prefill":
            return AttnMeta(H.MODE_CAUSAL, B, max_q, self.d_slots, self.d_ctx, self.d_bt, cu_q=self.d_cu_q, ctx_hint=hint)
        if kind == "decode":
            return AttnMeta(H.MODE_CAUSAL, B, 1, self.d_slots, self.d_ctx, self.d_bt, q_per_seq=1, ctx_hint=hint)
        if kind == "verify":
            return AttnMeta(H.MODE_CAUSAL, B, self.K + 1, self.d_slots, self.d_ctx, self.d_bt, q_per_seq=self.K + 1, ctx_hint=hint)
        raise ValueError(kind)

    def _note_ctx(self, max_ctx: int) -> None:
        """Host-side bound on the context lengths of the batch being staged (+ this step's lookahead); the graphs
        are keyed by its power-of-two bucket because the attention decomposition is fixed at capture."""
        self._ctx_hint = HipDecoder.ctx_bucket(max_ctx)

    def _body_decode(self, B: int, chain: bool, head: bool = True, sample: bool = False) -> None:
        self.model.forward(self.d_ids, self.d_pos, B, self._meta("decode", B))
        if not head:        # KV deposit only (the (K+1)-th draft forward, speculator_sync.py:55-56): no LM head / sampling
            return
        self.model.compute_logits(B)
        if sample:          # temperature > 0: keep logits_q for the ratio test and draw with the Gumbel-max sampler
            lg = self.model.full_logits(B)
            V = self.cfg.vocab_size
            if chain:
                H.store_step_rows(lg, V, self.d_logits_q, B, V, self.K, self.d_step)
            if chain and self.is_draft and self.sx is not None:
                # the async draft's JIT chain samples with is_tree=True (reference draft_runner.py:172 -> sampler.py:29-31):
                # under sampler_x its tokens come from the rescaled distribution, the q that verify() divides by
                H.topk_rows(lg, V, B, V, self.sx_k, self.d_boost)
                H.sample_rows(lg, V, B, V, self.d_temps, 1, self.d_rng, 1, self.d_next, boost_idx=self.d_boost, boost_k=self.sx_k,
                              boost_x=self.sx)
            else:
                H.sample_rows(lg, V, B, V, self.d_temps, 1, self.d_rng, 1, self.d_next)
            H.rng_advance(self.d_rng)
        elif chain and self.model.has_argmax_parts(B):
            # argmax from the LM head's candidates + the chain advance: one launch
            self.model.argmax_advance(B, self.d_next, self.d_ids, self.d_pos, self.d_slots, self.d_ctx, self.d_bt, self.max_blocks,
                                      self.block_size, self.d_spec, self.K, self.d_step)
            return
        else:
            self.model.argmax(B, self.d_next)
        if chain:
            H.draft_advance(self.d_next, self.d_ids, self.d_pos, self.d_slots, self.d_ctx, self.d_bt, self.max_blocks,
                            self.block_size, self.d_spec, self.K, self.d_step, B)

    def _body_verify(self, B: int, greedy_tail: bool, logits_q=None) -> None:
        T = B * (self.K + 1)
        self.model.forward(self.d_ids, self.d_pos, T, self._meta("verify", B))
        self.model.compute_logits(T)
        if logits_q is not None:     # temperature > 0: ratio acceptance + residual resampling (utils/verify.py:50-167)
            K, V = self.K, self.cfg.vocab_size
            lg = self.model.full_logits(T)
            self.model.argmax(T, self.d_next)
            H.row_lse(lg, V, T, V, self.d_temps, K + 1, self.d_lse_p)
            boost = {}
            if self.sx is not None:          # verify.py:101-105: q is the sampler_x-rescaled draft distribution
                H.topk_rows(logits_q, V, B * K, V, self.sx_k, self.d_boost)
                boost = dict(boost_k=self.sx_k, boost_x=self.sx)
            H.row_lse(logits_q, V, B * K, V, self.d_temps_q, K, self.d_lse_q, boost_idx=self.d_boost if boost else None, **boost)
            H.verify_ratio(lg, V, logits_q, V, V, B, K, self.d_ids, self.d_next, self.d_lse_p, self.d_lse_q, self.d_temps,
                           self.d_temps_q, self.d_ratio, self.d_rng, 2, self.d_accept, self.d_recovery, self.d_packed,
                           boost_idx_q=self.d_boost if boost else None, **boost)
            H.rng_advance(self.d_rng)
            return
        if greedy_tail and self.K + 1 <= 16 and self.model.has_argmax_parts(T):
            # LM head -> [argmax from its candidates + accept / reject] : the verify graph's tail is two launches
            self.model.argmax_verify(B, self.K, self.d_ids, self.d_next, self.d_accept, self.d_recovery, self.d_packed)
        elif greedy_tail:
            self.model.argmax(T, self.d_next)
            H.verify_greedy(self.d_next, self.d_ids, B, self.K, self.d_accept, self.d_recovery, self.d_packed)

    def _launch(self, key, body) -> None:
        """Replay the captured hipGraph for `key`, capturing it on first use (after one eager warm-up run, which
        also performs every lazy initialisation a capture must not contain)."""
        if self.config.enforce_eager:
            body()
            return
        key = (*key, self._ctx_hint)
        g = self.graphs.get(key)
        if g is None:
            body()                                   # eager warm-up on the current stream
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with capture(g, pool=self.graph_pool, stream=self.stream):
                body()
            if self.graph_pool is None:
                self.graph_pool = g.pool()
            self.graphs[key] = g
            # the warm-up advanced device-side state (draft chain); callers re-upload inputs before replay
            return "captured"
        g.replay()
        return None

    # ---------------------------------------------------------------------------------------------
    # public API
    # ---------------------------------------------------------------------------------------------
    def call(self, method: str, *args):
        """Reference-compatible entry (ssd/engine/model_runner.py:422-428); no RPC is needed under SPMD."""
        return getattr(self, method)(*args)

    @torch.inference_mode()
    def run(self, seqs, is_prefill: bool, last_only: bool = True, draft_return_logits: bool = False):
        """ModelRunner.run (model_runner.py:634-680).  Greedy only on this path (temperature 0): returns token
        ids (last_only) or the flat logits [B*(K+1), V] (verify with last_only=False)."""
        B = len(seqs)
        if is_prefill and B > self.seq_cap:
            assert not draft_return_logits
            toks, acts = [], []
            for i in range(0, B, self.seq_cap):
                part = seqs[i:i + self.seq_cap]
                toks.extend(self.run(part, True, last_only))
                if self.model.acts is not None:         # EAGLE-3 taps: the static buffer only holds the last slice
                    acts.append(self.model.acts[:sum(len(s) - s.num_cached_tokens for s in part)].clone())
            self._acts_sliced = torch.cat(acts, dim=0) if acts else None
            return toks
        self._acts_sliced = None
        if is_prefill:
            T, max_q = self._prepare_prefill(seqs)

            def body():
                self.model.forward(self.d_ids, self.d_pos, T, self._meta("prefill", B, max_q))
                self.model.compute_logits(T, gather=self.d_gather, rows=B)
            # the reference prefills eagerly (model_runner.py:602); here a prefill SHAPE that comes back (fixed-length
            # prompts: the benchmark protocol, batch jobs) replays one hipGraph -- every length-dependent quantity lives
            # in the static device buffers, only (B, T, longest query, context bucket) fix the launch geometry.  The FIRST
            # occurrence of a shape runs purely eagerly (no sync, no capture: with free-form prompt lengths almost every
            # prefill is a new shape); the second occurrence captures (its eager warm-up run is this call's result), later
            # ones replay.  A few graphs are kept, the oldest is dropped.
            key = ("prefill", B, T, max_q)
            full = (*key, self._ctx_hint)
            if self.config.enforce_eager or T > self.PREFILL_GRAPH_MAX_T:
                body()
            elif full not in self.graphs and self._prefill_seen.get(full, 0) < 1:
                if len(self._prefill_seen) >= 4096:
                    self._prefill_seen.clear()
                self._prefill_seen[full] = 1
                body()
            else:
                if full not in self.graphs and sum(1 for k in self.graphs if k[0] == "prefill") >= self.PREFILL_GRAPHS:
                    for k in [k for k in self.graphs if k[0] == "prefill"][:1]:
                        del self.graphs[k]
                self._launch(key, body)             # "captured": the eager warm-up run already produced this call's result
            self._log_margins(B, [(s.seq_id, len(s)) for s in seqs])
            self._sample_or_argmax(seqs, B)
            toks = self._read_tokens(B)
            return (toks, self.model.full_logits(B)) if draft_return_logits else toks
        if not last_only:
            T = self._prepare_verify(seqs, ids_from_seq=True)
            if self._launch(("verify_logits", B), lambda: self._body_verify(B, False)) == "captured":
                self._prepare_verify(seqs, ids_from_seq=True)
                self.graphs[("verify_logits", B, self._ctx_hint)].replay()
            return self.model.full_logits(T)
        self._prepare_decode(seqs)
        temps = self._seq_temps(seqs)
        if any(t > 0 for t in temps):       # autoregressive sampling (AutoRegressiveStep at temperature > 0)
            self._ensure_stochastic()
            self._upload(self.d_temps, temps, torch.float32)
            if self._launch(("decode_s", B), lambda: self._body_decode(B, False, sample=True)) == "captured":
                self._prepare_decode(seqs)
                self.graphs[("decode_s", B, self._ctx_hint)].replay()
        elif self._launch(("decode", B), lambda: self._body_decode(B, False)) == "captured":
            self._prepare_decode(seqs)
            self.graphs[("decode", B, self._ctx_hint)].replay()
        self._log_margins(B, [(s.seq_id, len(s)) for s in seqs])
        toks = self._read_tokens(B)
        return (toks, self.model.full_logits(B)) if draft_return_logits else toks

    def eagle_acts(self, n: int) -> torch.Tensor:
        """[n, taps * h] tapped activations of the last prefill / verify forward (reference model_runner.py:613-616); a view
        of a static buffer -- callers clone what they keep."""
        assert self.model.acts is not None, "activation taps are only collected under use_eagle"
        sliced = getattr(self, "_acts_sliced", None)
        if sliced is not None:              # a prefill batch that was run in slices of seq_cap sequences
            assert sliced.shape[0] == n
            return sliced
        return self.model.acts[:n]

    def _log_margins(self, rows: int, keys) -> None:
        """Debug aid of the parity tests (off unless margin_log is a dict): top-2 margin of logits[:rows]."""
        if self.margin_log is None or self.model.use_coll:
            return
        top = self.model.logits[:rows].float().topk(2, dim=-1).values
        for k, m in zip(keys, (top[:, 0] - top[:, 1]).tolist()):
            self.margin_log[k] = m

    def _seq_temps(self, seqs) -> list[float]:
        """prepare_sample (model_runner.py:542-550): the draft uses draft_temperature when given."""
        return [float(s.draft_temperature) if (self.is_draft and s.draft_temperature is not None) else float(s.temperature)
                for s in seqs]

    def _sample_or_argmax(self, seqs, B: int) -> None:
        temps = self._seq_temps(seqs)
        if any(t > 0 for t in temps):
            self._ensure_stochastic()
            self._upload(self.d_temps, temps, torch.float32)
            V = self.cfg.vocab_size
            H.sample_rows(self.model.full_logits(B), V, B, V, self.d_temps, 1, self.d_rng, 3, self.d_next)
            H.rng_advance(self.d_rng)
        else:
            self.model.argmax(B, self.d_next)

    def _read_tokens(self, n: int) -> list[int]:
        self.h_next[:n].copy_(self.d_next[:n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self._check_collectives()
        return self.h_next[:n].tolist()

    # ---- fast path of synchronous speculation: no host sync until the verify result ----
    @torch.inference_mode()
    def speculate_chain(self, seqs, recovery_tokens: list[int]) -> torch.Tensor:
        """K chained single-token draft forwards (SpeculatorSync.speculate, speculator_sync.py:47-66) starting from the
        recovery token at position N = len(seq) - 1 (the caller has appended it).  Returns the device tensor
        speculations [B, K+1] = (recovery, x_1..x_K); nothing is read back.  The reference's (K+1)-th forward, which
        only deposits x_K's KV, is deferred to `deposit_pending`: that KV is needed only if x_K gets accepted."""
        B, K = len(seqs), self.K
        self._check_chain_host()
        temps = self._seq_temps(seqs)
        sample = any(t > 0 for t in temps)
        key = ("decode_chain_s" if sample else "decode_chain", B)
        if sample:
            self._ensure_stochastic()

        def stage():
            self._prepare_decode(seqs)
            self.d_step.zero_()
            self.d_spec[:B, 0].copy_(self.d_ids[:B])
            if sample:
                self._upload(self.d_temps, temps, torch.float32)

        def chain():                     # all K forwards in ONE hipGraph: no host gap between them
            for _ in range(K):
                self._body_decode(B, True, sample=sample)

        stage()
        if self._launch(key, chain) == "captured":
            stage()                      # the warm-up + capture runs disturbed the chained state
            self.graphs[(*key, self._ctx_hint)].replay()
        self._post_chain_err()
        return self.d_spec[:B]

    @torch.inference_mode()
    def deposit_pending(self, seqs) -> None:
        """The deferred (K+1)-th draft forward of the previous round (speculator_sync.py:55-56), for sequences whose K
        draft tokens were all accepted: feed x_K (now the second-to-last token; the new recovery token has been
        appended) at its position so that its KV exists before the next chain reads it.  No LM head, no sampling."""
        B = len(seqs)
        self._stage_set = 1         # no host sync follows: the chain's staging must not reuse these pinned buffers
        try:
            self._prepare_decode(seqs, back=1)
        finally:
            self._stage_set = 0
        if self._launch(("decode_deposit", B), lambda: self._body_decode(B, False, head=False)) == "captured":
            self.graphs[("decode_deposit", B, self._ctx_hint)].replay()     # idempotent: same token, same slot
        self._post_chain_err()

    def logits_q(self, B: int) -> torch.Tensor:
        """[B, K, V] draft logits of the last sampled chain (temperature > 0 only)."""
        return self.d_logits_q[:B]

    @torch.inference_mode()
    def verify_chain(self, seqs, speculations: torch.Tensor, logits_q=None, temps_q=None, ratio_rows=None):
        """Target forward over the K+1 speculated tokens + greedy accept/reject on the device
        (Verifier.verify + verify(), verifier.py:54-153, utils/verify.py:5-48).  One packed D2H copy.
        Returns (new_suffixes, recovery_tokens)."""
        B, K = len(seqs), self.K
        self._acts_sliced = None
        stochastic = temps_q is not None
        key = ("verify_s" if stochastic else "verify", B)
        if stochastic:
            self._ensure_stochastic()
            if logits_q is None:        # greedy draft under a sampling target: q is one-hot, its logits are never read
                if not hasattr(self, "_lq_dummy"):
                    self._lq_dummy = torch.zeros(self.seq_cap, K, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device)
                logits_q = self._lq_dummy[:B]
            assert logits_q.is_contiguous() and tuple(logits_q.shape) == (B, K, self.cfg.vocab_size)
            if getattr(self, "_lq_ptr", None) not in (None, logits_q.data_ptr()):
                self.graphs = {k: v for k, v in self.graphs.items() if k[0] != "verify_s"}   # pointer is baked in the graph
            self._lq_ptr = logits_q.data_ptr()

        def stage():
            self._prepare_verify(seqs, ids_from_seq=False)
            self.d_ids[:B * (K + 1)].copy_(speculations.reshape(-1))
            if stochastic:
                self._upload(self.d_temps, self._seq_temps(seqs), torch.float32)
                self._upload(self.d_temps_q, [float(t) for t in temps_q], torch.float32)
                self._upload(self.d_ratio, [int(r) for r in ratio_rows], torch.int32)

        stage()
        if self._launch(key, lambda: self._body_verify(B, True, logits_q=logits_q)) == "captured":
            stage()
            self.graphs[(*key, self._ctx_hint)].replay()
        self._log_margins(B * (K + 1), [(s.seq_id, s.num_tokens - (K + 1) + j + 1) for s in seqs for j in range(K + 1)])
        self.h_packed[:B].copy_(self.d_packed[:B], non_blocking=True)
        # the verify is in flight and the host is about to block on it: the moment for a co-located draft server to
        # enqueue its next round on its own stream (engine/draft_runner.py run_deferred)
        hook = getattr(self, "overlap_hook", None)
        if hook is not None:
            hook()
        torch.cuda.current_stream().synchronize()
        self._check_collectives()
        rows = self.h_packed[:B].tolist()
        suffixes = [r[2:3 + r[0]] for r in rows]
        recovery = [r[1] for r in rows]
        return suffixes, recovery

    # ---------------------------------------------------------------------------------------------
    # draft-server operations of asynchronous speculation (explicit arrays instead of Sequence objects;
    # reference ssd/engine/draft_runner.py:51-101,124-184,380-450,620-812)
    # ---------------------------------------------------------------------------------------------
    def _async_lookahead(self) -> int:
        return self.K + 1 + self.K * self.config.MQ_LEN

    def cache_index(self, seq_ids, jlists):
        """Device copy of a round's cache keys besides the fork tokens: sequence id per row, glue position per branch."""
        return (to_device(list(seq_ids), torch.int64, self.device),
                to_device([list(j) for j in jlists], torch.int32, self.device).contiguous())

    @torch.inference_mode()
    def cache_lookup(self, keys, cache_seq: torch.Tensor, cache_j: torch.Tensor, forks: torch.Tensor) -> torch.Tensor:
        """int32 [B] device: index b * W + i of the cache entry (seq id, j, fork token) equal to each request key, or -1
        (csrc/misc.hip ssd_cache_lookup; reference draft_runner.py:215-252)."""
        B = len(keys)
        req = to_device([list(k) for k in keys], torch.int64, self.device)
        out = torch.empty(B, dtype=torch.int32, device=self.device)
        H.cache_lookup(req, cache_seq, cache_j, forks.contiguous(), B, forks.shape[0], forks.shape[1], out)
        return out

    def zeros_tokens(self, B: int, K: int) -> torch.Tensor:
        return torch.zeros(B, K, dtype=torch.int64, device=self.device)

    def _upload_tables(self, tables) -> None:
        self._upload(self.d_bt, [list(t) + [-1] * (self.max_blocks - len(t)) for t in tables], torch.int32)

    @torch.inference_mode()
    def draft_prefill(self, token_lists, tables) -> None:
        if len(token_lists) > self.seq_cap:
            for i in range(0, len(token_lists), self.seq_cap):
                self.draft_prefill(token_lists[i:i + self.seq_cap], tables[i:i + self.seq_cap])
            return
        B, T, max_q = self._stage_draft_prefill(token_lists, tables)
        self.model.forward(self.d_ids, self.d_pos, T, self._meta("prefill", B, max_q))

    def _stage_draft_prefill(self, token_lists, tables) -> tuple[int, int, int]:
        ids, pos, slots, ctx, cu, gather = [], [], [], [], [0], []
        max_q = 0
        for toks, tb in zip(token_lists, tables):
            n = len(toks)
            ids.extend(toks)
            pos.extend(range(n))
            slots.extend(self._slot(tb, p) for p in range(n))
            ctx.append(n)
            cu.append(cu[-1] + n)
            gather.append(cu[-1] - 1)
            max_q = max(max_q, n)
        B, T = len(token_lists), len(ids)
        assert T <= self.d_ids.numel()
        self._note_ctx(max(ctx))
        self._upload(self.d_ids, ids, torch.int64)
        self._upload(self.d_pos, pos, torch.int64)
        self._upload(self.d_slots, slots, torch.int32)
        self._upload(self.d_ctx, ctx, torch.int32)
        self._upload(self.d_cu_q, cu, torch.int32)
        self._upload_tables(tables)
        return B, T, max_q

    @torch.inference_mode()
    def draft_jit(self, rec, num_tokens, tables, temps=None) -> torch.Tensor:
        """K chained single-token decodes from the recovery token at P = n - 1 (no host sync).  With some
        temperature > 0 the tokens are sampled and the K rows of draft logits are kept for `logits_q`."""
        B, K = len(rec), self.K
        self._check_chain_host()
        sample = temps is not None and any(t > 0 for t in temps)
        key = ("decode_chain_s" if sample else "decode_chain", B)
        self._note_ctx(max(num_tokens) + self._async_lookahead())
        if sample:
            self._ensure_stochastic()

        def stage():
            pos = [n - 1 for n in num_tokens]
            self._upload(self.d_ids, list(rec), torch.int64)
            self._upload(self.d_pos, pos, torch.int64)
            self._upload(self.d_slots, [self._slot(tb, p) for tb, p in zip(tables, pos)], torch.int32)
            self._upload(self.d_ctx, list(num_tokens), torch.int32)
            self._upload_tables(tables)
            self.d_step.zero_()
            if sample:
                self._upload(self.d_temps, list(temps), torch.float32)

        def chain():
            for _ in range(K):
                self._body_decode(B, True, sample=sample)

        stage()
        if self._launch(key, chain) == "captured":
            stage()
            self.graphs[(*key, self._ctx_hint)].replay()
        self._post_chain_err()
        return self.d_spec[:B, 1:].clone()

    def _body_glue_fork(self, B: int) -> None:
        K, T = self.K, B * (self.K + 1)
        self.model.forward(self.d_ids, self.d_pos, T, self._meta("verify", B))
        self.model.compute_logits(T)
        if self.fork_split:             # per-slice candidates + one merge launch (csrc/sample.hip): bit-equal, 84.7 -> ~8 us per round
            H.fork_topf_split(self.model.logits, self.model.V, self.model.V, self.d_ids, self.d_fan, self.d_fan_off, B, K, self.mq,
                              self.d_fork_ws, self.d_forks)
        else:
            H.fork_topf(self.model.logits, self.model.V, self.model.V, self.d_ids, self.d_fan, self.d_fan_off, B, K, self.mq, self.d_forks)

    def _ensure_tree_buffers(self) -> None:
        if hasattr(self, "d_forks"):
            return
        B, K = self.max_bs, self.K
        self.mq = self.config.MQ_LEN
        T = B * self.mq
        dev = dict(device=self.device)
        self.d_fan = torch.zeros(B, K + 1, dtype=torch.int32, **dev)
        self.d_fan_off = torch.zeros(B, K + 1, dtype=torch.int32, **dev)
        self.d_forks = torch.zeros(B, self.mq, dtype=torch.int64, **dev)
        ws = H.fork_topf_workspace_bytes(self.model.V, B, K)       # 0: the split fork refuses this vocabulary (V % 8, V > 196608)
        self.fork_split = ws > 0
        self.d_fork_ws = torch.zeros(max(8, ws) // 8, dtype=torch.int64, **dev)
        self.d_jidx = torch.zeros(B, self.mq, dtype=torch.int32, **dev)
        self.d_jidx_flat = self.d_jidx.view(-1)      # packed [B][tree width] (the width may be a slice of MQ_LEN)
        self.d_tree_pos = torch.zeros(K, T, dtype=torch.int64, **dev)
        self.d_tree_slots = torch.zeros(K, T, dtype=torch.int32, **dev)
        self.d_tree_ctx = torch.zeros(K, B, dtype=torch.int32, **dev)
        self.d_tree_tokens = torch.zeros(T, K, dtype=torch.int64, **dev)
        self.d_steps_const = torch.arange(K, dtype=torch.int32, **dev)       # device-resident step indices
        self.d_tree_logits = None       # [T, K, V] bf16, allocated by the first sampled tree (43 MB per sequence at K=7 F=3)
        assert self.model.tp_size == 1, "the draft model is not tensor-parallel"

    @torch.inference_mode()
    def draft_glue_fork(self, glue_ids: torch.Tensor, num_tokens, tables, fan_lists) -> torch.Tensor:
        """Glue decode over [rec, x_1..x_K] at P..P+K, then top-F fork per position on the device."""
        self._ensure_tree_buffers()
        self._check_chain_host()
        B, K = glue_ids.shape[0], self.K
        key = ("glue_fork", B)
        self._note_ctx(max(num_tokens) + self._async_lookahead())

        def stage():
            pos, slots = [], []
            for n, tb in zip(num_tokens, tables):
                for p in range(n - 1, n + K):
                    pos.append(p)
                    slots.append(self._slot(tb, p))
            self.d_ids[:B * (K + 1)].copy_(glue_ids.reshape(-1))
            self._upload(self.d_pos, pos, torch.int64)
            self._upload(self.d_slots, slots, torch.int32)
            self._upload(self.d_ctx, [n + K for n in num_tokens], torch.int32)
            self._upload_tables(tables)
            fl = [list(f) for f in fan_lists]
            self._upload(self.d_fan, fl, torch.int32)
            self._upload(self.d_fan_off, [[sum(f[:j]) for j in range(len(f))] for f in fl], torch.int32)

        stage()
        if self._launch(key, lambda: self._body_glue_fork(B)) == "captured":
            stage()
            self.graphs[(*key, self._ctx_hint)].replay()
        self._post_chain_err()
        return self.d_forks[:B].clone()

    def _body_tree(self, B: int, d: int, sample: bool = False, mq: int | None = None) -> None:
        mq = mq or self.mq          # tree width: MQ_LEN, or one member's slice of it under draft data-parallelism
        T = B * mq
        meta = AttnMeta(H.MODE_TREE, B, mq, self.d_tree_slots[d], self.d_tree_ctx[d], self.d_bt, q_per_seq=mq,
                        tree_K=self.K, tree_mq=mq, tree_step=d, tree_F=1, tree_jidx=self.d_jidx,
                        ctx_hint=self._ctx_hint)
        self.model.forward(self.d_ids, self.d_tree_pos[d], T, meta)
        self.model.compute_logits(T)
        if sample:          # one temperature per sequence = per MQ_LEN branch rows; keep every branch's logits for logits_q
            V = self.cfg.vocab_size
            lg = self.model.full_logits(T)
            H.store_step_rows(lg, V, self.d_tree_logits, T, V, self.K, self.d_steps_const[d:])
            if self.sx is not None:          # Sampler(is_tree=True) with sampler_x (sampler.py:29-31)
                H.topk_rows(lg, V, T, V, self.sx_k, self.d_boost)
                H.sample_rows(lg, V, T, V, self.d_temps, mq, self.d_rng, 4, self.d_next, boost_idx=self.d_boost,
                              boost_k=self.sx_k, boost_x=self.sx)
            else:
                H.sample_rows(lg, V, T, V, self.d_temps, mq, self.d_rng, 4, self.d_next)
            H.rng_advance(self.d_rng)
        elif self.model.has_argmax_parts(T):
            # one launch writes the tokens, the next step's input ids and column d of the [T][K] token table
            self.model.argmax(T, self.d_next, self.d_ids, self.d_tree_tokens.view(-1)[d:], self.K)
            return
        else:
            self.model.argmax(T, self.d_next)
        self.d_ids[:T].copy_(self.d_next[:T])
        self.d_tree_tokens[:T, d].copy_(self.d_next[:T])

    @torch.inference_mode()
    def draft_tree(self, forks: torch.Tensor, num_tokens, tables, jlists, temps=None) -> torch.Tensor:
        """K tree-decode steps with the structural branch mask; tokens (greedy, or sampled when some temperature is
        > 0) are chained on the device.  Returns tokens [B*MQ, K]; the branches' logits stay in `tree_logits`."""
        self._ensure_tree_buffers()
        B, K, mq = forks.shape[0], self.K, forks.shape[1]      # mq < MQ_LEN: one member's branch slice (draft data-parallelism)
        assert mq <= self.mq
        T = B * mq
        self._note_ctx(max(num_tokens) + self._async_lookahead())
        sample = temps is not None and any(t > 0 for t in temps)
        if sample:
            self._ensure_stochastic()
            if self.d_tree_logits is None:
                self.d_tree_logits = torch.zeros(self.max_bs * self.mq, K, self.cfg.vocab_size, dtype=torch.bfloat16, device=self.device)

        def stage():
            pos = [[0] * T for _ in range(K)]
            slots = [[0] * T for _ in range(K)]
            ctx = [[0] * B for _ in range(K)]
            for d in range(K):
                for b, (n, tb) in enumerate(zip(num_tokens, tables)):
                    Pb = n - 1
                    for i in range(mq):
                        pos[d][b * mq + i] = Pb + jlists[b][i] + 1 + d
                        slots[d][b * mq + i] = self._slot(tb, Pb + K + 1 + d * mq + i)
                    ctx[d][b] = Pb + K + 1 + (d + 1) * mq
            # tables are [K][Tmax]-strided on the device: upload row by row into the leading T / B columns
            for d in range(K):
                self._stage_row(self.d_tree_pos, d, pos[d], torch.int64)
                self._stage_row(self.d_tree_slots, d, slots[d], torch.int32)
                self._stage_row(self.d_tree_ctx, d, ctx[d], torch.int32)
            self._upload(self.d_jidx_flat, [v for j in jlists for v in j], torch.int32)     # packed [B][mq]
            self._upload_tables(tables)
            self.d_ids[:T].copy_(forks.reshape(-1))
            if sample:
                self._upload(self.d_temps, list(temps), torch.float32)

        def all_steps():                 # the K tree steps in ONE hipGraph (each step has its own static metadata rows)
            for d in range(K):
                self._body_tree(B, d, sample, mq)

        stage()
        key = ("tree_s" if sample else "tree", B, mq)
        if self._launch(key, all_steps) == "captured":
            self.d_ids[:T].copy_(forks.reshape(-1))      # the eager warm-up consumed the inputs
            self.graphs[(*key, self._ctx_hint)].replay()
        self._post_chain_err()
        return self.d_tree_tokens[:T].clone()

    def tree_logits(self, T: int) -> torch.Tensor:
        """[T, K, V] logits of the last sampled tree (row = sequence * MQ_LEN + branch)."""
        return self.d_tree_logits[:T]

    def _stage_row(self, dst2d: torch.Tensor, row: int, values, dtype) -> None:
        key = (id(dst2d), row)
        stage = self._stage.get(key)
        if stage is None:
            stage = self._stage[key] = torch.zeros(dst2d.shape[1], dtype=dst2d.dtype).pin_memory()
        n = len(values)
        stage[:n] = torch.tensor(values, dtype=dtype)
        dst2d[row, :n].copy_(stage[:n], non_blocking=True)

    def exit(self, *a):
        self.graphs.clear()
