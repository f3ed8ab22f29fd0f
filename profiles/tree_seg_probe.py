"""Round 5: the async draft's round pieces on the 1B draft with the resident M-row layer segment (csrc/tree_segment.hip) on / off, one
process, the product runner's own hipGraphs: JIT chain (K single-token forwards), glue + fork (K+1 rows), K tree steps of MQ_LEN rows.
    python profiles/tree_seg_probe.py [ctx] [draft preset] [tree width] > gpurun_out/r05/tree_seg_probe.txt
tree width < MQ_LEN = one member's branch slice under draft data-parallelism (BASELINE configs[4]: Qwen3-0.6B x 4 -> 6 rows per step)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd.config import Config  # noqa: E402
from ssd_amd.engine.llm_engine import hip_runner_factory  # noqa: E402
from ssd_amd.model_config import PRESETS  # noqa: E402
from ssd_amd.utils.topology import Topology  # noqa: E402


def runner(seg: str, K: int, F: int, name: str = "llama-3.2-1b"):
    os.environ["SSD_TREE_SEG"] = seg
    cfg = Config(name, hf_config=PRESETS[name], draft=name, draft_hf_config=PRESETS[name], speculate=True, speculate_k=K,
                 draft_async=True, async_fan_out=F, jit_speculate=True, max_num_seqs=1, max_model_len=2048,
                 max_num_batched_tokens=2048, kvcache_block_size=256, num_kvcache_blocks=10, num_draft_kvcache_blocks=10)
    topo = Topology(0, 1, torch.device("cuda", 0), "target", 0, 1)
    return hip_runner_factory(cfg, PRESETS[name], is_draft=True, topo=topo, num_kvcache_blocks=10)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    name = sys.argv[2] if len(sys.argv) > 2 else "llama-3.2-1b"
    K, F = 7, 3
    MQ = F * (K + 1)
    width = int(sys.argv[3]) if len(sys.argv) > 3 else MQ
    tables, nt, rec = [list(range(8))], [ctx], [17]
    fan = [[F] * (K + 1)]
    jl = [[i // F for i in range(MQ)][:width]]
    res = {}
    for seg in ("0", "1"):
        dr = runner(seg, K, F, name)
        assert dr.model.tree_seg == (seg == "1")
        toks = dr.draft_jit(rec, nt, tables)
        glue = torch.cat([torch.tensor([rec], device=toks.device), toks], dim=1)
        forks = dr.draft_glue_fork(glue, nt, tables, fan)[:, :width].contiguous()
        out = {"draft": name, "tree_rows": width,
            "jit_chain_ms": round(timed(lambda: dr.draft_jit(rec, nt, tables)), 3),
            "glue_fork_ms": round(timed(lambda: dr.draft_glue_fork(glue, nt, tables, fan)), 3),
            "tree_ms": round(timed(lambda: dr.draft_tree(forks, nt, tables, jl)), 3),
        }
        out["tree_step_ms"] = round(out["tree_ms"] / K, 4)
        out["round_ms"] = round(out["glue_fork_ms"] + out["tree_ms"], 3)
        w = dr.model.weight_bytes()
        out["tree_step_TBps"] = round(w / (out["tree_step_ms"] * 1e-3) / 1e12, 3)
        if seg == "1":
            out["chain_err"] = int(dr.model.chain_err.item())
        tk = dr.draft_tree(forks, nt, tables, jl).cpu()
        res[seg] = (out, tk, forks.cpu())
        print(json.dumps({"SSD_TREE_SEG": seg, "ctx": ctx, **out}), flush=True)
        del dr
        torch.cuda.empty_cache()
    same_forks = torch.equal(res["0"][2], res["1"][2])
    agree = (res["0"][1] == res["1"][1]).float().mean().item()
    print(json.dumps({"forks_equal": same_forks, "tree_tokens_agree_frac (random weights: near-ties flip)": round(agree, 4)}))


if __name__ == "__main__":
    main()
