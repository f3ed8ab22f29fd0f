"""In-kernel ramp profile of the draft's decode kernels ("time-to-first-weight-tile and time-after-last").

Runs the Llama-3.2-1B draft's hipGraph rounds on the PROFILING build of the library (`make -C ssd_amd/csrc trace` ->
libssdhip_trace.so: the same kernels with KTRACE marks, csrc/common.h) and turns the marks -- thread 0 of every workgroup stores
the chip-wide 100 MHz clock at fixed points of the kernel -- into one table per kernel of the LAST decoder layer:
  * start:  first / median / last workgroup entering the kernel, relative to the first (dispatch skew);
  * each mark: median and max over workgroups of (mark - this kernel's first entry);
  * end:    the last workgroup's last mark;  gap: the next kernel's first entry minus this kernel's end (the boundary).
python profiles/ktrace_probe.py [preset] [ctx]     (default llama-3.2-1b, context 300)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd.hip.lib as L  # noqa: E402

L.lib_path = lambda: os.path.join(ROOT, "ssd_amd", "_lib", "libssdhip_trace.so")       # this process only: the profiling build

from ssd_amd.config import Config  # noqa: E402
from ssd_amd.engine.llm_engine import hip_runner_factory  # noqa: E402
from ssd_amd.model_config import PRESETS  # noqa: E402
from ssd_amd.utils.topology import Topology  # noqa: E402

MARKS, MAXWG, SLOTS = 12, 4096, 32
NAMES = {0: "fused norm+QKV+RoPE+store (M<=16)", 1: "fused norm+gate_up+SiLU (M<=16)", 2: "fused GEMM without prologue", 3: "qkv+RoPE m32",
         4: "gemm_sp o_proj -> slabs", 5: "gemm_sp down_proj -> slabs", 6: "attention", 7: "attention + o_proj -> slabs",
         8: "gemm_wf gate_up + SiLU", 9: "gemm_wf rows (LM head)", 10: "rmsnorm over slabs", 11: "rmsnorm",
         12: "chain segment (o -> gate_up -> down -> next QKV), csrc/chain.hip", 13: "chain segment, last layer (o -> gate_up -> down)",
         14: "M-row segment (o -> gate_up -> down -> next QKV), csrc/tree_segment.hip", 15: "M-row segment, last layer (o -> gate_up -> down)"}
MARK_NAMES = {
    0: ["entry", "weight loads issued", "x / slabs arrived (barrier)", "x^ in LDS (barrier)", "MFMAs done (weights arrived)", "combine barrier", "stores issued"],
    1: ["entry", "weight loads issued", "x / slabs arrived (barrier)", "x^ in LDS (barrier)", "MFMAs done (weights arrived)", "combine barrier", "stores issued"],
    2: ["entry", "weight loads issued", "-", "-", "MFMAs done", "combine barrier", "stores issued"],
    4: ["entry", "all loads issued", "MFMAs done (data arrived)", "combine barrier", "slab stored"],
    5: ["entry", "all loads issued", "MFMAs done (data arrived)", "combine barrier", "slab stored"],
    6: ["entry", "Q + first K/V tiles issued", "key loop done", "merge barrier", "merged + stored"],
    7: ["entry", "Q + first K/V tiles issued", "key loop done", "merge barrier", "merged (x in LDS)", "o_proj slab stored"],
    8: ["entry", "-", "-", "-", "last tile's MFMAs done", "combine barrier", "stores issued"],
    9: ["entry", "-", "-", "-", "last tile's MFMAs done", "combine barrier", "stores issued"],
    10: ["entry", "-", "-", "-", "-", "-", "stored"],
    11: ["entry", "-", "-", "-", "-", "-", "stored"],
    12: ["entry", "o_proj published", "edge 1: gathered, x^ ready", "gate_up published", "edge 2: activations gathered", "down_proj published",
         "edge 3: gathered, x^ ready", "QKV + RoPE stored"],
    13: ["entry", "o_proj published", "edge 1: gathered, x^ ready", "gate_up published", "edge 2: activations gathered", "down_proj published",
         "edge 3: gathered, rows stored"],
    14: ["entry", "o_proj rows stored", "o_proj published", "edge 1: all flags seen", "add + norm done, x^ in LDS", "gate_up published",
         "edge 2: all flags seen, x issued", "down_proj published", "edge 3: all flags seen", "add + norm done, x^ in LDS", "QKV + RoPE stored"],
    15: ["entry", "o_proj rows stored", "o_proj published", "edge 1: all flags seen", "add + norm done, x^ in LDS", "gate_up published",
         "edge 2: all flags seen, x issued", "down_proj rows stored"],
}


def report(buf, title, order):
    t = buf.cpu().view(SLOTS, MAXWG, MARKS).numpy()
    rows = {}
    for s in range(SLOTS):
        live = t[s][t[s][:, 0] != 0]
        if len(live):
            rows[s] = live
    if not rows:
        print(title, ": no marks (is this the trace build?)")
        return
    print(f"\n=== {title} ===  (us; 100 MHz clock, 0.01 us resolution)")
    seq = [s for s in order if s in rows] + [s for s in rows if s not in order]
    seq.sort(key=lambda s: rows[s][:, 0].min())
    prev_end = None
    for s in seq:
        live = rows[s].astype("int64")
        base = live[:, 0].min()
        ends = live.max(axis=1)
        gap = "" if prev_end is None else f"  gap after previous kernel's end {(base - prev_end) / 100:.2f}"
        import numpy as np
        print(f"[slot {s}] {NAMES.get(s, '?')}: {len(live)} workgroups; entry first 0.00 median {(np.median(live[:, 0]) - base) / 100:.2f} "
              f"last {(live[:, 0].max() - base) / 100:.2f}; end (last workgroup) {(ends.max() - base) / 100:.2f}{gap}")
        names = MARK_NAMES.get(s, [])
        for i in range(1, MARKS):
            col = live[:, i]
            col = col[col != 0]
            if len(col) == 0:
                continue
            nm = names[i] if i < len(names) else f"mark {i}"
            print(f"      {nm:38s} median {(np.median(col) - base) / 100:6.2f}  max {(col.max() - base) / 100:6.2f}   (in-workgroup: median "
                  f"{np.median(live[:, i][live[:, i] != 0] - live[:, 0][live[:, i] != 0]) / 100:5.2f} after its own entry)")
        prev_end = ends.max()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    K, F = 7, 3
    cfg = Config(name, hf_config=PRESETS[name], draft=name, draft_hf_config=PRESETS[name], speculate=True, speculate_k=K,
                 draft_async=True, async_fan_out=F, jit_speculate=True, max_num_seqs=1, max_model_len=2048,
                 max_num_batched_tokens=2048, kvcache_block_size=256, num_kvcache_blocks=10, num_draft_kvcache_blocks=10)
    topo = Topology(0, 1, torch.device("cuda", 0), "target", 0, 1)
    dr = hip_runner_factory(cfg, PRESETS[name], is_draft=True, topo=topo, num_kvcache_blocks=10)
    lib = L.load_library()
    buf = torch.zeros(SLOTS * MAXWG * MARKS, dtype=torch.int64, device="cuda")
    for tu in ("gemm", "gemm_fused", "gemm_sk", "attention", "norm", "chain", "tree_segment"):
        fn = getattr(lib, f"ssd_ktrace_set_{tu}")
        fn.argtypes, fn.restype = [C.c_void_p], C.c_int
        assert fn(buf.data_ptr()) == 0, tu
    tables, nt, rec = [list(range(8))], [ctx], [17]

    def settle(fn, n=4):
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()

    settle(lambda: dr.draft_jit(rec, nt, tables))
    report(buf, f"{name}: single-token forward (last layer + tail of the last chained forward), ctx {ctx}", [0, 6, 12, 7, 1, 5, 13, 11, 10, 9])
    toks = dr.draft_jit(rec, nt, tables)
    glue = torch.cat([torch.tensor([rec], device=toks.device), toks], dim=1)
    fan, jl = [[F] * (K + 1)], [[j for j in range(K + 1) for _ in range(F)]]
    settle(lambda: dr.draft_glue_fork(glue, nt, tables, fan))
    report(buf, f"{name}: glue forward, M = {K + 1}", [0, 6, 14, 7, 4, 1, 5, 15, 10, 11, 9])
    forks = dr.draft_glue_fork(glue, nt, tables, fan)
    settle(lambda: dr.draft_tree(forks, nt, tables, jl))
    report(buf, f"{name}: tree step, M = {F * (K + 1)} (last layer + tail of the last step)", [3, 6, 14, 4, 10, 8, 5, 15, 11, 9])


if __name__ == "__main__":
    main()
