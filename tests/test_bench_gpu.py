"""bench.py as the driver runs it: `python bench.py --gpus N` must start by itself at every N (it re-executes under
torch.distributed.run when it is not already inside a launch), and the JSON line carries the same keys at N > 1 as at
N = 1 (`roofline`, `cpu_baseline`, + `collective`).  On the one-GPU test box the N = 2 ranks share GPU 0
(SSD_DIST_BACKEND=gloo for the control collectives, SSD_LOCAL_DEVICE=0); the in-forward sums go through the one-shot
all-reduce between the two processes."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "roofline", "cpu_baseline", "reference_protocol"}


def bench(*flags, shared_gpu=False):
    env = dict(os.environ, PYTHONPATH=ROOT, SSD_BENCH_CPU_SECONDS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if shared_gpu:
        env.update(SSD_DIST_BACKEND="gloo", SSD_LOCAL_DEVICE="0", SSD_AR_DEVICE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--ttft-samples", "2", *flags],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def check(line, n):
    assert KEYS <= set(line), sorted(KEYS - set(line))
    assert line["n_gpus"] == n and line["value"] > 0 and line["ms_per_step"] > 0
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1.0 and r["achieved"] > 0 and r["peak"] == 8000.0
    assert line["cpu_baseline"]["value"] and line["cpu_baseline"]["cores"] >= 1, line["cpu_baseline"]
    assert "workload" in line["config"]
    rp = line["reference_protocol"]          # the reference's own bench protocol: 128-token prompts -> 512 output tokens
    assert rp["tokens_per_s_total"] > 0 and rp["tokens_per_s_decode"] >= rp["tokens_per_s_total"] and rp["final_context"] == 640
    assert 1.0 <= rp["mean_accepted_len"] <= 8.0


def test_bench_self_launches_at_every_n():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    one = bench("--gpus", "1", "--workload", "tiny")
    check(one, 1)
    two = bench("--gpus", "2", "--workload", "tiny", shared_gpu=True)           # NOT under torchrun: bench.py spawns the ranks
    check(two, 2)
    assert set(one) <= set(two) and "collective" in two
    assert two["collective"]["avg_us"] > 0
    # the correlated pair makes speculation accept draft tokens (random pairs sit at 1.0)
    assert one["mean_accepted_len"] > 1.2, one["mean_accepted_len"]


def test_bench_async_placements():
    """The metric's mode (async SSD, k=7 f=3) on toy shapes: co-located draft at N = 1 and N = 2 (TP = 2 + draft server on
    rank 0), and the dedicated draft rank over the p2p transport at N = 2."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    a1 = bench("--gpus", "1", "--workload", "tiny-async")
    check(a1, 1)
    assert a1["cache_hit_rate"] is not None and a1["cache_hit_rate"] > 0.05, a1["cache_hit_rate"]
    a2 = bench("--gpus", "2", "--workload", "tiny-async", shared_gpu=True)
    check(a2, 2)
    assert "tp2" in a2["config"]["parallelism"]
    d2 = bench("--gpus", "2", "--workload", "tiny-async", "--placement", "dedicated", shared_gpu=True)
    check(d2, 2)
    assert d2["config"]["parallelism"] == "tp1+draft1"
    # same models, same greedy decision rule: placement must not change what gets accepted
    assert abs(a1["mean_accepted_len"] - d2["mean_accepted_len"]) < 1e-9, (a1["mean_accepted_len"], d2["mean_accepted_len"])
    # draft data-parallelism: two draft ranks shard the speculation tree (HIP kernels on a 12-branch slice each)
    d3 = bench("--gpus", "3", "--workload", "tiny-async", "--placement", "dedicated", "--draft-dp", "2", shared_gpu=True)
    check(d3, 3)
    assert d3["config"]["parallelism"] == "tp1+draft2"
    assert abs(a1["mean_accepted_len"] - d3["mean_accepted_len"]) < 1e-9 and abs(a1["cache_hit_rate"] - d3["cache_hit_rate"]) < 1e-9


def test_bench_eagle_workload():
    """bench.py --workload tiny-eagle: the EAGLE-3 path (activation taps, wire tensors, EAGLE draft runner) through the bench
    harness, co-located at N = 1 and on a dedicated draft rank over the p2p transport at N = 2."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    e1 = bench("--gpus", "1", "--workload", "tiny-eagle", "--ref-seqs", "1", "--ref-output-len", "64")
    assert e1["config"]["eagle3"] and e1["value"] > 0 and e1["roofline"]["frac"] > 0
    assert 1.0 <= e1["mean_accepted_len"] <= 8.0 and e1["cache_hit_rate"] is not None
    e2 = bench("--gpus", "2", "--workload", "tiny-eagle", "--placement", "dedicated", "--ref-seqs", "0", shared_gpu=True)
    assert e2["config"]["parallelism"] == "tp1+draft1" and e2["value"] > 0
    assert abs(e1["mean_accepted_len"] - e2["mean_accepted_len"]) < 1e-9
