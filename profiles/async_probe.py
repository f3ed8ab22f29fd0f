"""Time the async (SSD) draft round on one GPU with the real shapes: JIT chain (miss path), glue + fork, K tree-decode
steps (MQ_LEN = F*(K+1) branches) for the Llama-3.2-1B draft, k=7 f=3, via the in-process loopback server; and the
K+1-query verify of the target next to it.  python profiles/async_probe.py [target preset]"""
import json
import random
import sys
import os
import time

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
    eng.generate([prompt], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=24), use_tqdm=False)   # capture graphs
    dr, tr = eng.draft_runner, eng.model_runner
    srv = eng.draft_server
    # time the pieces in isolation on the warmed-up graphs
    tables = [list(range(8))]
    nt = [300]

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    rec = [17]
    toks = dr.draft_jit(rec, nt, tables)
    glue = torch.cat([torch.tensor([rec], device=toks.device), toks], dim=1)
    fan = [[F] * (K + 1)]
    jl = [srv.j_hit]
    forks = dr.draft_glue_fork(glue, nt, tables, fan)
    out = {
        "target": tname, "K": K, "F": F, "MQ_LEN": F * (K + 1),
        "draft_jit_ms (K chained single-token decodes, miss path)": round(timed(lambda: dr.draft_jit(rec, nt, tables)), 3),
        "draft_glue_fork_ms (K+1-token forward + top-F fork)": round(timed(lambda: dr.draft_glue_fork(glue, nt, tables, fan)), 3),
        "draft_tree_ms (K steps x MQ_LEN tokens)": round(timed(lambda: dr.draft_tree(forks, nt, tables, jl)), 3),
    }
    out["draft_round_ms (glue + tree, overlaps the target verify)"] = round(
        out["draft_glue_fork_ms (K+1-token forward + top-F fork)"] + out["draft_tree_ms (K steps x MQ_LEN tokens)"], 3)
    vt = METRICS["target_verify_times"]
    out["target_verify_ms (K+1 queries, incl. accept + D2H)"] = round(sum(vt[2:]) / max(1, len(vt[2:])) * 1e3, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
