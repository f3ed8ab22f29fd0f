"""LLMEngine: request loop, metrics, step wiring (reference ssd/engine/llm_engine.py:39-381).

Public behaviour kept: ``LLM(model, **Config fields)``; ``generate(prompts, sampling_params, use_tqdm=True,
stream_callback=None) -> (outputs, METRICS)`` with ``outputs[i] = {"text", "token_ids"}``; the METRICS keys.

Process model (differs by design): one process per GPU, SPMD.  Under ``torchrun`` every rank constructs the same
engine and calls ``generate`` with the same requests; tensor-parallel ranks stay in lock step through the RCCL
collectives inside the forward and never exchange control messages (the reference pickles every call into a
shared-memory RPC, model_runner.py:404-428).
"""
from __future__ import annotations

import os
from dataclasses import fields
from time import perf_counter

import torch

from ssd_amd.config import Config
from ssd_amd.sampling_params import SamplingParams
from ssd_amd.engine.sequence import Sequence
from ssd_amd.engine.scheduler import Scheduler
from ssd_amd.engine.step import AutoRegressiveStep, SpecDecodeStep, InferenceStep
from ssd_amd.engine.speculator_sync import SpeculatorSync
from ssd_amd.engine.verifier import Verifier


def _fresh_metrics() -> dict:
    return {
        "cache_hits": [],
        "accepted_suffix_lens_with_recovery": [],
        "accepted_suffix_lens_on_hit": [],
        "accepted_suffix_lens_on_miss": [],
        "prefill_total_time": 0,
        "decode_total_time": 0,
        "prefill_total_tokens": 0,
        "decode_total_tokens": 0,
        "target_step_times": [],
        "target_verify_times": [],
    }


METRICS = _fresh_metrics()


def _load_tokenizer(path: str):
    if not os.path.isdir(path):
        return None
    if not any(os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "tokenizer_config.json", "tokenizer.model")):
        return None
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, use_fast=True)


def hip_runner_factory(config, model_cfg, *, is_draft: bool, topo, **kw):
    from ssd_amd.engine.model_runner import ModelRunner
    if model_cfg.family == "eagle3":
        from ssd_amd.engine.eagle_runner import EagleDraftRunner as ModelRunner
    path = config.draft if is_draft else config.model
    return ModelRunner(config, model_cfg, is_draft=is_draft, device=topo.device, tp_rank=topo.tp_rank,
                       tp_size=topo.tp_size, tp_group=topo.tp_group, model_path=path if os.path.isdir(path or "") else None,
                       weights_seed=config.draft_weights_seed if is_draft else config.weights_seed, **kw)


class LLMEngine:
    def __init__(self, model: str, runner_factory=None, topology=None, inprocess_draft: bool = False, **kwargs):
        """inprocess_draft: run the async draft server inside this process over a loopback transport (same
        protocol and server code, no second GPU) -- for single-GPU functional tests, not a deployment mode."""
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.config = config
        Sequence.block_size = config.kvcache_block_size
        # LLM(model, num_gpus=N) from a single process (reference llm_engine.py:61-127): become rank 0 and spawn the rest
        self._followers = []
        if config.num_gpus > 1 and "WORLD_SIZE" not in os.environ and topology is None and not inprocess_draft:
            from ssd_amd.engine.launcher import spawn_followers
            self._followers = spawn_followers(model, dict(kwargs, runner_factory=runner_factory) if runner_factory else dict(kwargs),
                                              config.num_gpus)
            import atexit
            atexit.register(self.exit)
        from ssd_amd.engine.launcher import resolve_factory
        runner_factory = resolve_factory(runner_factory)
        assert config.num_gpus > 1 or not config.draft_async or inprocess_draft, "draft_async requires at least 2 gpus"

        from ssd_amd.utils.topology import resolve_topology, Topology
        if topology is None and inprocess_draft:
            if int(os.environ.get("WORLD_SIZE", "1")) > 1:      # tensor-parallel target, draft co-located on TP rank 0
                topology = resolve_topology(config, colocated_draft=True)
            else:
                dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
                topology = Topology(0, 1, dev, "target", 0, 1)
        self.topo = topology or resolve_topology(config)
        from ssd_amd.utils import watchdog
        watchdog.stage("runner_init (weights, KV cache, buffers)")
        factory = runner_factory or hip_runner_factory

        self.draft_runner = None
        self._capped_ids: set[int] = set()
        self.async_link = None
        self.draft_server = None
        if self.topo.role == "draft":               # dedicated draft GPU of async speculation
            from ssd_amd.engine.draft_runner import DraftServer
            from ssd_amd.engine.async_proto import DistTransport
            runner = factory(config, config.draft_hf_config, is_draft=True, topo=self.topo, memory_utilization=0.8,
                             num_kvcache_blocks=config.num_draft_kvcache_blocks)
            dp = None
            if self.topo.dp_size > 1:
                from ssd_amd.engine.draft_runner import DraftGroup
                dp = DraftGroup(self.topo.draft_group, self.topo.dp_rank, self.topo.dp_size, self.topo.device)
            tx = DistTransport(self.topo.async_group, 0, self.topo.device) if self.topo.dp_rank == 0 else None
            self.draft_server = DraftServer(config, runner, tx, dp=dp)
            return
        self.model_runner = factory(config, config.hf_config, is_draft=False, topo=self.topo,
                                    num_kvcache_blocks=config.num_kvcache_blocks)
        config.num_kvcache_blocks = self.model_runner.num_kvcache_blocks
        draft_blocks = None
        if config.speculate and not config.draft_async:
            self.draft_runner = factory(config, config.draft_hf_config, is_draft=True, topo=self.topo.single(),
                                        memory_utilization=0.75, num_kvcache_blocks=config.num_draft_kvcache_blocks)
            draft_blocks = self.draft_runner.num_kvcache_blocks
            if self.topo.tp_size > 1:
                # the replicated draft sized its KV cache from THIS rank's free memory; every rank runs its own
                # Scheduler and must take identical admission / preemption decisions -> agree on the minimum
                import torch.distributed as dist
                t = torch.tensor([draft_blocks], dtype=torch.int64, device=self.topo.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.topo.tp_group)
                draft_blocks = int(t.item())
        elif config.speculate:
            from ssd_amd.engine.speculator_async import AsyncLink
            transport = None
            if inprocess_draft and self.topo.tp_rank == 0:
                from ssd_amd.engine.draft_runner import DraftServer
                from ssd_amd.engine.async_proto import LoopbackTransport
                transport, server_end = LoopbackTransport.pair()
                self.draft_runner = factory(config, config.draft_hf_config, is_draft=True, topo=self.topo.single(),
                                            memory_utilization=0.75, num_kvcache_blocks=config.num_draft_kvcache_blocks)
                # co-located draft: its own stream, next-round work parked until the target's verify is in flight
                on_gpu = self.topo.device.type == "cuda"
                # (retired in round 5, both measured on MI355X and not worth a switch: a high-priority draft stream -- 29.0 vs 28.4 ms per
                # step, profiles/r02_colocated_overlap.txt -- and a CU-mask partition of the chip between draft and target -- 36-38 vs
                # 30.5 ms: the target's launches are sized for 256 CUs, profiles/r02_cu_partition.txt)
                side = torch.cuda.Stream(self.topo.device) if on_gpu else None
                # the M-row resident segment (csrc/tree_segment.hip) keeps 256 workgroups waiting for each other: beside a verify that owns
                # the CUs they come up one by one and spin -- measured on c4 (profiles/r05_c4_kernel_stats_with_glue_segment.txt): 85.6 us
                # per glue launch against 27 us alone, and the CUs they hold are taken from the target.  "auto" keeps it for draft
                # servers that have the GPU to themselves (dedicated placement); the single-token chain segment only runs on misses,
                # with the target waiting for its reply, and stays on.
                if side is not None and on_gpu and os.environ.get("SSD_TREE_SEG", "auto") == "auto" and hasattr(self.draft_runner, "model"):
                    self.draft_runner.model.tree_seg_colocated = True
                self.draft_server = DraftServer(config, self.draft_runner, server_end, stream=side, deferred=side is not None)
                if side is not None:
                    self.model_runner.overlap_hook = self.draft_server.run_deferred

                def pump(server=self.draft_server, end=server_end):
                    while end.from_peer:
                        server.handle_one()
                transport.pump = pump
                if side is not None or os.environ.get("SSD_DEFER_DRAFT_PREFILL") == "1":        # (the switch: CPU tests of the order)
                    # One GPU serves both models, so the draft's prefill cannot overlap the target's for free as it does on the
                    # reference's dedicated draft GPU: enqueued first (as until round 2) its ~160 eager launches and ~2.5 ms of
                    # kernels sit in front of / between the target's prefill kernels and delay the FIRST token by ~3 ms
                    # (profiles/r03_c4_prefill_timeline.txt).  The command is left in the loop-back queue instead: the first
                    # speculation request's receive pumps it (commands are served in order), i.e. it runs after the first token
                    # has been streamed and before the first draft round needs the draft's KV.
                    transport.defer_prefill = True
            self.async_link = AsyncLink(config, self.topo, transport=transport)
            from ssd_amd.utils import watchdog
            watchdog.stage("draft_hello (first send / recv on the draft p2p group)", 600.0)     # the draft rank answers once its weights are up
            draft_blocks = self.async_link.draft_num_blocks()
            watchdog.stage("engine_init")

        self.tokenizer = _load_tokenizer(config.tokenizer_path or config.model)
        if self.tokenizer is not None and self.tokenizer.eos_token_id is not None:
            config.eos = self.tokenizer.eos_token_id
        self.scheduler = Scheduler(config, draft_num_blocks=draft_blocks)

    # ---------------------------------------------------------------------------------------------
    def add_request(self, prompt, sampling_params: SamplingParams) -> None:
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("string prompts need a tokenizer in the model directory; pass token-id lists")
            prompt = self.tokenizer.encode(prompt)
        self.scheduler.add(Sequence(prompt, sampling_params))

    def step(self, step: InferenceStep):
        t = perf_counter()
        seqs, is_prefill = self.scheduler.schedule()
        capped = []
        for s in self.scheduler.pop_capped():          # stopped at max_model_len (Scheduler._length_capped)
            toks = s.completion_token_ids
            if s.first_token_streamed is not None:      # streamed at the prefill, capped before the round that appends it:
                toks = toks + [s.first_token_streamed]  # the result must carry what the stream already delivered
            capped.append((s.seq_id, toks))
            self._capped_ids.add(s.seq_id)
        if not seqs:            # every runnable sequence was one of those
            return capped
        n = step.prefill(seqs) if is_prefill else step.decode(seqs)
        dt = perf_counter() - t
        if is_prefill:
            METRICS["prefill_total_time"] += dt
            METRICS["prefill_total_tokens"] += n
        else:
            METRICS["decode_total_time"] += dt
            METRICS["decode_total_tokens"] += n
        return capped + [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]

    def is_finished(self) -> bool:
        return self.scheduler.is_finished()

    def abort_all(self) -> None:
        """Drop every queued / running request and return its KV blocks (benchmarks that stop a request mid-generation)."""
        sch = self.scheduler
        self._drain_loopback()      # queued draft commands (a deferred prefill) run while their blocks still have their owners
        for seq in list(sch.running):
            sch.block_manager.deallocate(seq)
            if sch.draft_block_manager is not None:
                sch.draft_block_manager.deallocate(seq)
        sch.running.clear()
        sch.waiting.clear()
        srv = getattr(self, "draft_server", None)
        if srv is not None:         # co-located draft: its parked round and its speculation cache belong to the aborted sequences
            srv.reset()             # (the parked glue / tree work would write draft KV into blocks that may already have new owners)

    def _drain_loopback(self) -> None:
        """Co-located draft server: serve every command still queued in the loop-back transport (a draft prefill deferred to
        the first speculation request that never came: generate() cut short by max_steps, abort_all).  Left queued, it would run
        at the NEXT request against draft blocks that may have new owners by then."""
        link = getattr(self, "async_link", None)
        pump = getattr(getattr(link, "tx", None), "pump", None) if link is not None else None
        if pump is not None:
            pump()

    def create_inference_step(self, config: Config) -> InferenceStep:
        if not config.speculate:
            return AutoRegressiveStep(self.scheduler, self.model_runner, self.tokenizer)
        if config.draft_async:
            from ssd_amd.engine.speculator_async import SpeculatorAsync
            speculator = SpeculatorAsync(config.speculate_k, self.topo.device, self.async_link, config)
        else:
            speculator = SpeculatorSync(config.speculate_k, self.topo.device, self.draft_runner)
        verifier = Verifier(config.speculate_k, self.topo.device, self.model_runner, sampler_x=config.sampler_x,
                            async_fan_out=config.async_fan_out, jit_speculate=config.jit_speculate,
                            tokenizer=self.tokenizer, metrics=METRICS)
        return SpecDecodeStep(self.scheduler, speculator, verifier, eagle=config.use_eagle, tokenizer=self.tokenizer,
                              async_spec=config.draft_async)

    def log_metrics(self) -> None:
        if METRICS["prefill_total_time"]:
            print(f"Final Prefill Throughput: {int(METRICS['prefill_total_tokens'] / METRICS['prefill_total_time'])}tok/s", flush=True)
        if METRICS["decode_total_time"]:
            print(f"Final Decode Throughput: {int(METRICS['decode_total_tokens'] / METRICS['decode_total_time'])}tok/s", flush=True)
        lens = METRICS["accepted_suffix_lens_with_recovery"]
        if self.config.speculate and lens:
            print(f"[metrics] Avg Tokens per step (incl recovery): {sum(lens) / len(lens):.2f}", flush=True)
            print(f"[metrics] Avg Fraction of Speculated Tokens Accepted: "
                  f"{((sum(lens) - len(lens)) / len(lens)) / self.config.speculate_k:.2f}", flush=True)
            if METRICS["cache_hits"]:
                print(f"[metrics] Avg Cache Hits: {sum(METRICS['cache_hits']) / len(METRICS['cache_hits']):.2f}", flush=True)

    @property
    def is_draft_process(self) -> bool:
        return self.topo.role == "draft"

    def serve(self) -> None:
        """Draft rank: answer speculation requests until the target sends EXIT."""
        assert self.is_draft_process
        self.draft_server.serve_forever()

    def follow(self) -> None:
        """A rank spawned by the leader process (engine/launcher.py): the draft rank serves the wire protocol; a
        tensor-parallel follower replays the leader's generate() calls until told to exit."""
        import torch.distributed as dist
        if self.is_draft_process:
            self.serve()
            return
        while True:
            box = [None]
            dist.broadcast_object_list(box, src=0, group=self.topo.ctl_group)
            cmd = box[0]
            if cmd[0] == "exit":
                return
            self.generate(cmd[1], cmd[2], use_tqdm=False)

    def generate(self, prompts, sampling_params, use_tqdm: bool = True, stream_callback=None):
        if self.is_draft_process:           # SPMD launch: the draft rank serves instead of generating
            self.serve()
            return [], METRICS
        if self._followers and self.topo.tp_size > 1:      # single-process launch: hand the request to the TP followers
            import torch.distributed as dist
            dist.broadcast_object_list([("generate", prompts, sampling_params)], src=0, group=self.topo.ctl_group)
        for k, v in _fresh_metrics().items():
            METRICS[k] = v
        self._capped_ids.clear()            # finish reasons are per generate() call
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        outputs: dict[int, list[int]] = {}
        step = self.create_inference_step(self.config)
        max_steps = self.config.max_steps if self.config.max_steps is not None else float("inf")
        streamed: dict[int, int] = {}
        i = 0
        while not self.is_finished() and i < max_steps:
            i += 1
            t = perf_counter()
            finished = self.step(step)
            METRICS["target_step_times"].append(perf_counter() - t)
            if stream_callback:
                for seq in self.scheduler.running:
                    cur, prev = seq.num_completion_tokens, streamed.get(seq.seq_id, 0)
                    if cur > prev:
                        stream_callback(seq.seq_id, seq.completion_token_ids[prev:cur])
                        streamed[seq.seq_id] = cur
                    elif (cur == 0 and prev == 0 and self.config.speculate and seq.recovery_token_id is not None
                          and seq.temperature == 0):
                        # Speculative modes keep the token a prefill produced as the sequence's "recovery token" and append it
                        # at the START of the next speculation round (reference speculator_sync.py:33-36 / speculator_async.py:
                        # 118-124) -- unconditionally, so it IS the first output token.  Hand it to the stream now instead of one
                        # whole draft + verify round later (the reference's loop streams it then: its TTFT = prefill + first
                        # round).  The token is pinned: should the sequence be preempted and prefilled again before that
                        # round, Verifier.prefill keeps it (a re-prefill may flip a near-tie).
                        seq.first_token_streamed = seq.recovery_token_id
                        stream_callback(seq.seq_id, [seq.recovery_token_id])
                        streamed[seq.seq_id] = 1
            for seq_id, token_ids in finished:
                if stream_callback:
                    prev = streamed.get(seq_id, 0)
                    if len(token_ids) > prev:
                        stream_callback(seq_id, token_ids[prev:])
                outputs[seq_id] = token_ids
                if pbar:
                    pbar.update(1)
        if pbar:
            pbar.close()
        self._drain_loopback()
        result = []
        for seq_id in sorted(outputs):
            toks = outputs[seq_id]
            text = self.tokenizer.decode(toks) if self.tokenizer is not None else ""
            # finish_reason (beyond the reference's two keys): "max_model_len" marks a request the scheduler had to stop before
            # max_new_tokens / EOS because its next speculation step would not fit the model length
            result.append({"text": text, "token_ids": toks,
                           "finish_reason": "max_model_len" if seq_id in self._capped_ids else "stop"})
        if not stream_callback and self.config.verbose:
            self.log_metrics()
        # a snapshot, not the module-level dict itself (the reference returns the global: two engines run one after the other in
        # one process then "return" the same object, and comparing their metrics compares the second run with itself -- a round-3
        # test did exactly that)
        return result, {k: (list(v) if isinstance(v, list) else v) for k, v in METRICS.items()}

    def exit(self, hard: bool = False) -> None:
        if self._followers:
            import torch.distributed as dist
            if self.topo.tp_size > 1:
                dist.broadcast_object_list([("exit",)], src=0, group=self.topo.ctl_group)
        if self.async_link is not None:
            self.async_link.shutdown()
        if self._followers:
            import torch.distributed as dist
            for p in self._followers:
                p.join(timeout=60)
            self._followers = []
            if dist.is_initialized():
                dist.destroy_process_group()
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
                os.environ.pop(k, None)
