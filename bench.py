#!/usr/bin/env python
"""Headline benchmark: speculative decoding output tokens/s on MI355X, the reference's metric
(bench/bench.py:351-361: output tokens / wall clock; decode-only variant llm_engine.py:215-223).

Contract: ``python bench.py --gpus N --steps K --warmup W`` (one rank per GPU; under torchrun for N > 1) prints
ONE JSON line on rank 0.  A *step* is one full speculation round of the real engine -- K_spec+1 chained draft
forwards, the K_spec+1-query target verify forward, the on-device accept/reject, and the scheduler
post-processing -- for one sequence (b = 1) of synthetic token ids, exactly the objects ``LLM.generate`` drives.

Workload (``--workload``):
  c3   (default) Llama-3.1-70B shapes as target, tensor-parallel over the N GPUs (TP = N; 139 GB of bf16
       weights fit one 288 GB MI355X, so N = 1 is the same model), Llama-3.2-1B shapes as draft replicated on
       every rank, synchronous SD k = 6, b = 1, temp 0, 128-token prompt (BASELINE.json configs[2] shape;
       the metric's model).  Same total work at every N -> "strong" scaling.
  c2   Llama-3.1-8B target + 1B draft, sync SD k = 6 on one GPU (BASELINE.json configs[1]).
  tiny 2-layer toy shapes (plumbing check).
Weights are synthetic (seeded N(0, 0.02)); no checkpoints exist offline.  With unrelated random draft and target
the acceptance is ~0, so ``value`` (tokens/s at the MEASURED acceptance) is essentially 1 / step latency; the
line also carries ms_per_step, mean_accepted_len and the step's HBM roofline fraction, which are
acceptance-independent because every shape is fixed.

Extra objects on the line: ``roofline`` (dominant kernel = the skinny weight-streaming GEMM family
gemm_wf_kernel, timed with HIP events on the launch stream while rotating through all layers' weights so
nothing is cache-resident) and ``cpu_baseline`` (the oracle engine -- the reference's own modules restated --
timed on the host cores on a bounded sample; baseline only).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X spec (guide: 6.29e12 measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "tiny"])
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--input-len", type=int, default=128)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--ttft-samples", type=int, default=3)
    return ap.parse_args()


def workload_models(name):
    from ssd_amd.model_config import PRESETS, ModelConfig
    if name == "c3":
        return "llama-3.1-70b", PRESETS["llama-3.1-70b"], "llama-3.2-1b", PRESETS["llama-3.2-1b"]
    if name == "c2":
        return "llama-3.1-8b", PRESETS["llama-3.1-8b"], "llama-3.2-1b", PRESETS["llama-3.2-1b"]
    t = ModelConfig("llama", 512, 2, 8, 8, 64, 1024, 4096, 1e-5, 5e5, 8192, False)
    d = ModelConfig("llama", 256, 1, 8, 8, 32 * 2, 512, 4096, 1e-5, 5e5, 8192, False)
    return "tiny-target", t, "tiny-draft", d


@torch.inference_mode()
def gemm_roofline(engine, steps_k, draft_fwd_per_step=None):
    """Time every skinny-GEMM launch shape of one speculation step with HIP events on the launch stream.
    For each (matrix kind, M) all L layers' matrices are launched back to back (L x tens of MB >> the 256 MiB
    Infinity Cache), so each launch streams its weights from HBM as in the real forward."""
    from ssd_amd.hip import ops as H
    tot_bytes = tot_time = 0.0
    tot_launch = 0
    per_kind = {}
    if draft_fwd_per_step is None:
        draft_fwd_per_step = steps_k
    for runner, M, fwd_per_step in ((engine.model_runner, steps_k + 1, 1), (engine.draft_runner, 1, draft_fwd_per_step)):
        if runner is None:
            continue
        m = runner.model
        L = m.cfg.num_layers
        # the GEMM kernels are exactly the ones the forward issues for M tokens (HipDecoder.launch_*, fused norm /
        # RoPE / SiLU variants included); separate add+RMSNorm / RoPE launches of the unfused variants are left out
        kinds = [("qkv", m.qkv_n * m.h * 2, lambda li: m.launch_qkv(li, M, runner.d_pos, runner.d_slots, gemm_only=True)),
                 ("o", m.h * m.qn * 2, lambda li: m.launch_o(li, M)),
                 ("gate_up", 2 * m.I * m.h * 2, lambda li: m.launch_gate_up(li, M, gemm_only=True)),
                 ("down", m.h * m.I * 2, lambda li: m.launch_down(li, M))]
        runner.d_slots[:M].fill_(-1)          # timing only: do not touch the KV cache
        for kind, b, launch in kinds:
            for li in range(min(2, L)):
                launch(li)
            reps = max(2, 128 // L)
            # the L launches are captured in a hipGraph and the replay is timed: the real forward is a graph
            # replay too, and eager launches through ctypes are host-bound (~8 us each) for the small shapes
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for li in range(L):
                    launch(li)
            graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3 / (reps * L)
            tag = ("draft." if runner.is_draft else "target.") + kind
            per_kind[tag] = {"us": round(dt * 1e6, 2), "GBps": round(b / dt / 1e9, 1), "MB": round(b / 1e6, 1)}
            tot_bytes += b * L * fwd_per_step
            tot_time += dt * L * fwd_per_step
            tot_launch += L * fwd_per_step
    achieved = tot_bytes / tot_time
    # HBM traffic per launch: PMC counters cannot be collected from inside this process; the committed separate
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (profiles/r01_c2_pmc_traffic.csv, FETCH_SIZE doubled
    # as the gfx950 guide prescribes) measured read+write bytes = ratio x algorithmic bytes for this kernel family
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            traffic = int(tot_bytes / tot_launch * json.load(f)["gemm_traffic_over_algorithmic"])
    except Exception:
        pass
    return {"bound": "hbm", "kernel": "gemm_wf_kernel + gemm_sk_kernel + gemm_fused_kernel (skinny weight-streaming GEMM family)",
            "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
            "traffic_source": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r01_c2_pmc_traffic.csv (read bytes = x1.012 of algorithmic over the GEMM family)",
            "launches_per_step": round(tot_launch, 1), "avg_launch_us": round(tot_time / tot_launch * 1e6, 2),
            "bytes_per_launch_avg": int(tot_bytes / tot_launch), "per_kind": per_kind}


def cpu_baseline(k_spec):
    """The oracle engine (reference modules restated on CPU) on a bounded sample of the workload:
    (a) Llama-3.2-1B shapes, greedy AR decode, b=1 (BASELINE.json configs[0]) -> tokens/s measured end to end;
    the 70B target does not fit host RAM, so no extrapolation is made -- this is a baseline, not a target."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import PRESETS
    from ssd_amd.sampling_params import SamplingParams
    # GEMV-shaped bf16 matmuls stop scaling (and then collapse) beyond a few tens of threads: on the 256-core
    # GPU-box host, 256 threads ran ~1000x slower than 16.  Use at most 16 and report that number as `cores`.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = PRESETS["llama-3.2-1b"]
    eng = LLMEngine("llama-3.2-1b", hf_config=cfg, runner_factory=oracle_runner_factory(), max_model_len=1024,
                    max_num_batched_tokens=1024, kvcache_block_size=256, num_kvcache_blocks=4)
    random.seed(0)
    prompt = [random.randint(0, 10000) for _ in range(32)]
    # time-bounded sample: decode steps until ~12 s of CPU work have been spent (at least 4, at most 512 tokens), so a
    # slow or oversubscribed host cannot stall the benchmark
    eng.add_request(prompt, SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=512))
    step = eng.create_inference_step(eng.config)
    t0 = time.perf_counter()
    eng.step(step)                                  # prefill
    t1 = time.perf_counter()
    n = 0
    while not eng.is_finished() and (n < 4 or time.perf_counter() - t1 < 12.0):
        eng.step(step)
        n += 1
    wall = time.perf_counter() - t0
    dec = n / (time.perf_counter() - t1)
    return {"value": round(dec, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle engine (CPU restatement of the reference), Llama-3.2-1B shapes (the workload's draft model; "
                      f"BASELINE configs[0]), greedy AR b=1, 32-token prompt, {n} output tokens (time-bounded sample, ~12 s), bf16 weights; "
                      f"end-to-end wall {wall:.1f}s incl. prefill; decode-only rate reported"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    from ssd_amd.engine.llm_engine import LLMEngine, METRICS
    from ssd_amd.sampling_params import SamplingParams

    tname, tcfg, dname, dcfg = workload_models(args.workload)
    K = args.k
    max_len = 2048
    blocks = (max_len // 256) + 2
    engine = LLMEngine(tname, hf_config=tcfg, draft=dname, draft_hf_config=dcfg, speculate=True, speculate_k=K,
                       num_gpus=args.gpus, max_num_seqs=1, max_model_len=max_len, max_num_batched_tokens=max_len,
                       kvcache_block_size=256, num_kvcache_blocks=blocks, num_draft_kvcache_blocks=blocks,
                       enforce_eager=args.eager)
    dev = engine.topo.device
    random.seed(0)
    prompt = [random.randint(0, 10000) for _ in range(args.input_len)]

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # ---- TTFT (chat.py:95-111 definition: generate() call -> first streamed token), p50 over a few runs ----
    ttfts = []
    for _ in range(max(1, args.ttft_samples)):
        first = []
        sync_all()
        t0 = time.perf_counter()
        engine.generate([prompt], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=1), use_tqdm=False,
                        stream_callback=lambda sid, toks: first.append(time.perf_counter()) if not first else None)
        ttfts.append((first[0] - t0) * 1e3)
    ttft_p50 = statistics.median(ttfts[1:] if len(ttfts) > 1 else ttfts)   # first run pays graph capture

    # ---- timed decode steps through the real engine ----
    total = args.warmup + args.steps
    engine.add_request(prompt, SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=total * (K + 1) + 8))
    for k_ in list(METRICS):
        METRICS[k_] = [] if isinstance(METRICS[k_], list) else 0
    step = engine.create_inference_step(engine.config)
    engine.step(step)                       # prefill
    for _ in range(args.warmup):
        engine.step(step)
    n0 = len(METRICS["accepted_suffix_lens_with_recovery"])
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.step(step)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    lens = METRICS["accepted_suffix_lens_with_recovery"][n0:]
    tokens = sum(lens)
    ms_step = dt / args.steps * 1e3
    seq = engine.scheduler.running[0] if engine.scheduler.running else None
    ctx = len(seq) if seq is not None else args.input_len

    tb, db = engine.model_runner.model.weight_bytes(), engine.draft_runner.model.weight_bytes()
    kv_tok = lambda m: 2 * m.cfg.num_layers * m.nkv * m.hd * 2
    # draft forwards actually run per step: K chained ones + the deferred KV-deposit forward, which only follows a
    # fully accepted round (engine/speculator_sync.py); the reference always runs K+1 (7 x 2.471 GB)
    draft_fwd = K + sum(1 for n in lens if n == K + 1) / max(1, len(lens))
    step_bytes = tb + draft_fwd * db + ctx * (kv_tok(engine.model_runner.model) + draft_fwd * kv_tok(engine.draft_runner.model))
    full_accept_step_s = (dt / args.steps) * (1 + (K + 1 - draft_fwd) * db / step_bytes)
    out = {
        "metric": "output tokens/sec (sync SD, b=1, temp 0), with p50 TTFT and mean accepted length",
        "value": round(tokens / dt, 3), "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic token ids (random.seed(0), randint(0,10000)) + synthetic seeded weights",
        "config": {"workload": f"{tname} target TP={args.gpus} + {dname} draft (replicated), sync speculative decoding "
                               f"k={K}, b=1, temp=0, input_len={args.input_len}, kv block 256",
                   "steps_are": "one speculate (k chained draft fwd; the reference's (k+1)-th, KV-deposit-only fwd runs only "
                                "after a fully accepted round) + verify ((k+1)-query target fwd) + accept round of the engine",
                   "parallelism": f"tp{args.gpus}", "hipgraph": not args.eager},
        "mean_accepted_len": round(tokens / max(1, len(lens)), 4),
        "ttft_p50_ms": round(ttft_p50, 3),
        "draft_forwards_per_step": round(draft_fwd, 3),
        "step_hbm_bytes_per_gpu": int(step_bytes),
        "step_roofline_frac": round(step_bytes / (dt / args.steps) / HBM_PEAK, 4),
        "tokens_per_s_if_all_accepted": round((K + 1) / full_accept_step_s, 2),
    }
    if rank == 0 and args.gpus == 1:
        if not args.no_roofline:
            out["roofline"] = gemm_roofline(engine, K, draft_fwd)
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(K)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
