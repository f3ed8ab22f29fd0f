"""Where the 1B draft's chained decode goes: wall time per forward (hipGraph replay of K chained single-token
forwards) next to the sum of kernel durations from `rocprofv3 --kernel-trace` of the same command.
python profiles/draft_probe.py [K] [ctx]"""
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


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    name = "llama-3.2-1b"
    cfg = Config(name, hf_config=PRESETS[name], draft=name, draft_hf_config=PRESETS[name], speculate=True, speculate_k=K,
                 draft_async=True, async_fan_out=3, jit_speculate=True, max_num_seqs=1, max_model_len=2048,
                 max_num_batched_tokens=2048, kvcache_block_size=256, num_kvcache_blocks=10, num_draft_kvcache_blocks=10)
    topo = Topology(0, 1, torch.device("cuda", 0), "target", 0, 1)
    dr = hip_runner_factory(cfg, PRESETS[name], is_draft=True, topo=topo, num_kvcache_blocks=10)
    tables, nt, rec = [list(range(8))], [ctx], [17]

    def timed(fn, n=20):
        fn()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ms = timed(lambda: dr.draft_jit(rec, nt, tables))
    w = dr.model.weight_bytes() if hasattr(dr.model, "weight_bytes") else None
    print(json.dumps({"K": K, "ctx": ctx, "chain_ms": round(ms, 3), "ms_per_forward": round(ms / K, 4), "weight_bytes": w}))


if __name__ == "__main__":
    main()
