"""Per-launch time of the paged-attention kernel inside a hipGraph (un-profiled), decode-side shapes.
Each launch in the graph uses a different layer's K/V cache so nothing is L2-resident by accident."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16


def graph_time(body, n_inner, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_inner * 1e6


@torch.inference_mode()
def main():
    L = 32
    for name, T, nh, nkv, hd in (("1b decode", 1, 32, 8, 64), ("70b verify", 7, 64, 8, 128), ("8b verify", 7, 32, 8, 128),
                                 ("1b tree T=24", 24, 32, 8, 64)):
        bs, blocks = 256, 20
        kc = torch.randn(L, blocks, nkv, bs, hd, device="cuda").to(BF)
        vc = torch.randn(L, blocks, nkv, bs, hd, device="cuda").to(BF)
        q = torch.randn(T, nh * hd, device="cuda").to(BF)
        out = torch.zeros(T, nh * hd, device="cuda", dtype=BF)
        bt = torch.arange(blocks, dtype=torch.int32, device="cuda").view(1, blocks)
        G = nh // nkv
        groups = (-(-(T * G) // 16) + 1) // 2
        for ctx in (140, 300, 640, 2048, 4096):
            cl = torch.tensor([ctx], dtype=torch.int32, device="cuda")
            res = []
            ws_o = torch.zeros(T * nh * 16 * hd, device="cuda")
            ws_ml = torch.zeros(T * nh * 16 * 2, device="cuda")
            for waves, splits in ((8, 1), (4, 1), (8, 2), (8, 4), (8, 8), (8, 16)):
                def body():
                    for li in range(L):
                        H.attn_paged(q, kc[li], vc[li], bt, blocks, cl, 1, T, T, nh, nkv, hd, bs, hd ** -0.5, q_per_seq=T,
                                     out_rows=out, waves=waves, splits=splits, ws_o=ws_o, ws_ml=ws_ml)
                res.append(f"w{waves}s{splits}:{graph_time(body, L):6.2f}")
            print(f"{name:14s} ctx={ctx:5d}  " + "  ".join(res) + "  us/launch")


main()
