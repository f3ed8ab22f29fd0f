"""LLM(model, num_gpus=N) from ONE process (the reference's calling convention, llm_engine.py:61-127): the engine
spawns the other ranks itself (engine/launcher.py).  CPU/gloo, oracle runners: tensor parallel (2 ranks) and
asynchronous speculation (target + draft rank) must reproduce the single-rank token stream exactly.  Runs in a
subprocess so the process group / environment of the pytest process stay untouched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
from ssd_amd.llm import LLM
from ssd_amd.model_config import ModelConfig
from ssd_amd.sampling_params import SamplingParams
t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 1024, False)
d = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 1024, True)
kw = dict(hf_config=t, max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=32, num_kvcache_blocks=48, weights_std=0.1,
          max_num_seqs=2, runner_factory="oracle.runner:oracle_runner_factory")
prompts = [[(5 * i + 3 * j) %% 256 for j in range(6 + 2 * i)] for i in range(2)]
sp = SamplingParams(temperature=0, max_new_tokens=12, ignore_eos=True)
mode = sys.argv[1]
if mode == "tp2":
    kw.update(num_gpus=2)
elif mode == "async":
    kw.update(num_gpus=2, draft="d", draft_hf_config=d, speculate=True, speculate_k=3, draft_async=True, async_fan_out=2, jit_speculate=True)
elif mode == "sync_tp2":
    kw.update(num_gpus=2, draft="d", draft_hf_config=d, speculate=True, speculate_k=3)
llm = LLM("t", **kw)
out, m = llm.generate(prompts, sp, use_tqdm=False)
out2, _ = llm.generate(prompts[:1], sp, use_tqdm=False)          # a second call goes through the same followers
llm.exit()
print("RESULT " + json.dumps([[o["token_ids"] for o in out], [o["token_ids"] for o in out2]]))
""" % ROOT


def run(mode):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", SCRIPT, mode], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
    assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(lines[-1][7:])


def test_single_process_launch_tp_and_async_match_one_rank():
    one = run("one")
    assert run("tp2") == one
    assert run("async") == one
    assert run("sync_tp2") == one
