"""Async SSD (k=7, f=3) with the draft server CO-LOCATED on the target's GPU: does running the draft's glue + tree round
on its own stream, concurrently with the target's verify, beat doing them back to back?
python profiles/colocated_probe.py [target preset]   (env SSD_COLOCATED_OVERLAP=0 serialises the two)"""
import json
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd.engine.llm_engine import LLMEngine, METRICS  # noqa: E402
from ssd_amd.model_config import PRESETS  # noqa: E402
from ssd_amd.sampling_params import SamplingParams  # noqa: E402


def main():
    tname = sys.argv[1] if len(sys.argv) > 1 else "llama-3.1-8b"
    K, F = 7, 3
    eng = LLMEngine(tname, hf_config=PRESETS[tname], draft="llama-3.2-1b", draft_hf_config=PRESETS["llama-3.2-1b"], speculate=True,
                    speculate_k=K, draft_async=True, async_fan_out=F, jit_speculate=True, inprocess_draft=True, max_num_seqs=1,
                    max_model_len=2048, max_num_batched_tokens=2048, kvcache_block_size=256, num_kvcache_blocks=10,
                    num_draft_kvcache_blocks=10)
    random.seed(0)
    prompt = [random.randint(0, 10000) for _ in range(128)]
    sp = SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=24)
    eng.generate([prompt], sp, use_tqdm=False)                       # graph capture
    out, m = eng.generate([prompt], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=60), use_tqdm=False)
    torch.cuda.synchronize()
    st = m["target_step_times"][2:]
    lens = m["accepted_suffix_lens_with_recovery"]
    print(json.dumps({"target": tname, "overlap": os.environ.get("SSD_COLOCATED_OVERLAP", "1") != "0", "K": K, "F": F,
                      "ms_per_step": round(1e3 * sum(st) / len(st), 3), "steps": len(st),
                      "mean_accepted": round(sum(lens) / len(lens), 3), "cache_hit_rate": round(sum(m["cache_hits"]) / max(1, len(m["cache_hits"])), 3),
                      "tokens": out[0]["token_ids"][:8]}))


main()
