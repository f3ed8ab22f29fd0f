"""First contact with a multi-GPU box is unattended: whatever fails there, the launch must END within a bounded time and rank 0
must print ONE JSON line -- the result, or a failure record naming the stage (ssd_amd/utils/watchdog.py, used by bench.py; the
N-rank start-up it guards: reference ssd/engine/llm_engine.py:61-93, speculator_async.py:130-187).  Fault injection under gloo with
the real engine on oracle runners, launched the way the driver launches bench.py (`python -m torch.distributed.run`): a tensor-
parallel rank that dies mid-run, a dedicated draft rank that dies mid-run, a rank that wedges without dying."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(layout: str, world: int, fault: str, extra_env=None, limit: float = 60.0):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SSD_FAULT=fault, SSD_DIST_BACKEND="gloo", PYTHONPATH=ROOT, OMP_NUM_THREADS="1", **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "fault_worker.py"), layout]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit + 30)
    dt = time.time() - t0
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    return p.returncode, dt, lines, p.stderr


def test_clean_run_prints_the_result_line():
    rc, dt, lines, err = launch("tp2", 2, "")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["value"] == 1.0 and "error" not in lines[0]


@pytest.mark.parametrize("kind", ["exit", "raise"])
def test_a_tensor_parallel_rank_dies_mid_run(kind):
    rc, dt, lines, err = launch("tp2", 2, f"rank=1,after=3,kind={kind}")
    assert rc != 0 and dt < 60, (rc, dt)
    assert len(lines) == 1, (lines, err[-2000:])
    rec = lines[0]
    assert rec["value"] is None and rec["rank"] == 0 and rec["stage"] == "timed_steps" and rec["failure"] in ("exception", "terminated"), rec
    if kind == "raise":         # the culprit left its own record: rank 0's line carries it
        assert any(p["rank"] == 1 and "injected failure" in p["error"] for p in rec["peer_failures"]), rec


def test_the_dedicated_draft_rank_dies_mid_run():
    rc, dt, lines, err = launch("tp1+draft", 2, "rank=1,after=3,kind=exit")
    assert rc != 0 and dt < 60, (rc, dt)
    assert len(lines) == 1, (lines, err[-2000:])
    rec = lines[0]
    assert rec["value"] is None and rec["rank"] == 0 and rec["stage"] == "timed_steps", rec


def test_a_rank_that_wedges_without_dying_is_timed_out():
    rc, dt, lines, err = launch("tp2", 2, "rank=1,after=3,kind=hang", extra_env={"SSD_STAGE_TIMEOUT_S": "8"})
    assert rc != 0 and dt < 60, (rc, dt)
    assert len(lines) == 1, (lines, err[-2000:])
    rec = lines[0]
    assert rec["failure"] == "timeout" and rec["stage"] == "timed_steps" and rec["rank"] == 0, rec


def test_three_ranks_target_tp2_plus_draft_the_draft_wedges():
    """TP = 2 + a dedicated draft rank (the shape of configs[3] / [4]): the draft stops answering; the head rank waits in a recv,
    the other target rank in the reply broadcast -- both must be ended by their stage limit, one line on stdout."""
    rc, dt, lines, err = launch("tp2+draft", 3, "rank=2,after=2,kind=hang", extra_env={"SSD_STAGE_TIMEOUT_S": "10"})
    assert rc != 0 and dt < 60, (rc, dt)
    assert len(lines) == 1, (lines, err[-2000:])
    assert lines[0]["failure"] == "timeout" and lines[0]["stage"] == "timed_steps", lines[0]


def test_bench_py_itself_without_a_gpu_prints_one_failure_record():
    """The real `python bench.py --gpus 2` (self-launching, relaying launcher + RunGuard in every rank) on a box WITHOUT a GPU: the
    product path has no CPU fallback, so both ranks raise during start-up -- the launch must end at once with ONE json line that
    carries the benchmark's keys, value null, the error and the stage."""
    if __import__("torch").cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert p.returncode != 0 and time.time() - t0 < 60
    assert len(lines) == 1, p.stdout[-2000:]
    rec = lines[0]
    assert rec["value"] is None and rec["n_gpus"] == 2 and "no CPU fallback" in rec["error"] and rec["stage"].startswith("runner_init"), rec
    assert {"metric", "unit", "steps", "warmup", "higher_is_better", "config"} <= set(rec)
