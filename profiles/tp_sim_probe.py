"""What one rank of a TP=N 70B target does per step, measured on ONE GPU: a model config with exactly the per-rank
shard shapes (heads, kv heads, intermediate and vocabulary divided by N), tp_size 1 but the collective launches
forced through a 1-rank group (one-shot all-reduce kernel or RCCL), + the replicated 1B draft, sync k=6.
= the TP=N step minus the xGMI wait.   python profiles/tp_sim_probe.py 8 [custom|rccl]"""
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd.engine.llm_engine import LLMEngine, METRICS, hip_runner_factory  # noqa: E402
from ssd_amd.model_config import ModelConfig, PRESETS  # noqa: E402
from ssd_amd.sampling_params import SamplingParams  # noqa: E402
from ssd_amd.utils.topology import Topology  # noqa: E402


def main():
    tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    custom = (sys.argv[2] if len(sys.argv) > 2 else "custom") == "custom"
    full = PRESETS["llama-3.1-70b"]
    shard = ModelConfig("llama", full.hidden_size, full.num_layers, full.num_heads // tp, full.num_kv_heads // tp, full.head_dim,
                        full.intermediate_size // tp, full.vocab_size // tp, full.rms_norm_eps, full.rope_theta,
                        full.max_position_embeddings, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29588")
    dist.init_process_group("nccl", rank=0, world_size=1)
    grp = dist.new_group([0])
    dev = torch.device("cuda", 0)
    topo = Topology(0, 1, dev, "target", 0, 1, grp)

    def factory(config, model_cfg, *, is_draft, topo, **kw):
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, force_collectives=not is_draft,
                                  custom_ar=custom, **kw)

    K = 6
    # the draft must share the target's (sharded) vocabulary for token ids to be valid: shard its vocabulary too
    d = PRESETS["llama-3.2-1b"]
    draft = ModelConfig("llama", d.hidden_size, d.num_layers, d.num_heads, d.num_kv_heads, d.head_dim, d.intermediate_size,
                        full.vocab_size // tp, d.rms_norm_eps, d.rope_theta, d.max_position_embeddings, True)
    eng = LLMEngine("shard", hf_config=shard, draft="d", draft_hf_config=draft, speculate=True, speculate_k=K,
                    runner_factory=factory, topology=topo, max_num_seqs=1, max_model_len=2048, max_num_batched_tokens=2048,
                    kvcache_block_size=256, num_kvcache_blocks=10, num_draft_kvcache_blocks=10)
    random.seed(0)
    prompt = [random.randint(0, 10000) for _ in range(128)]
    eng.generate([prompt], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=16), use_tqdm=False)
    t0 = time.perf_counter()
    eng.generate([prompt], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=40), use_tqdm=False)
    dt = time.perf_counter() - t0
    steps = METRICS["target_step_times"][1:]
    ver = METRICS["target_verify_times"][1:]
    print(json.dumps({"tp": tp, "collectives": "one-shot kernel" if custom else "rccl", "note": "draft LM head is 1/tp of the real one",
                      "ms_per_step": round(sum(steps) / len(steps) * 1e3, 3), "verify_ms": round(sum(ver) / len(ver) * 1e3, 3),
                      "weights_MB_per_rank": eng.model_runner.model.weight_bytes() >> 20, "wall_s": round(dt, 2)}))


main()
