"""Prefill-chunk GEMMs (M = 48..128) of the model shapes: skinny kernel (gemm.hip, x from L2 per wave) vs the
LDS-shared / split-K kernel (gemm_pf.hip).  hipGraph replays rotating over 4 weight copies (nothing L2/MALL-resident)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16


def graph_time(body, n_inner, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_inner * 1e6


@torch.inference_mode()
def main():
    shapes = [("70b qkv", 10240, 8192), ("70b o", 8192, 8192), ("70b gate_up", 57344, 8192), ("70b down", 8192, 28672),
              ("8b gate_up", 28672, 4096), ("8b down", 4096, 14336)]
    C = 4
    for name, N, K in shapes:
        ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(C)]
        for M in (128,):
            xf = torch.randn(H.frag_numel(M, K), device="cuda").to(BF)
            y = torch.zeros(M, N, device="cuda", dtype=BF)
            wsb = torch.zeros(H.gemm_pf_workspace_bytes(M, N, K) // 4 + 16 * M * N, dtype=torch.float32, device="cuda")
            t_old = graph_time(lambda: [H.gemm(xf, ws_[i % C], y, M, N, K, N) for i in range(8)], 8)
            res = [f"wf:{t_old:7.1f}us {N * K * 2 / t_old / 1e6:5.2f}TB/s"]
            for nt, sp in ((4, 1), (4, 2), (4, 4), (4, 8), (2, 1), (2, 2), (2, 4), (2, 8), (2 | 8 << 8, 1), (2 | 8 << 8, 2), (2 | 8 << 8, 4), (2 | 8 << 8, 8)):
                try:
                    t = graph_time(lambda: [H.gemm_pf(xf, ws_[i % C], y, M, N, K, N, wsb, splits=sp, nt=nt) for i in range(8)], 8)
                    res.append(f"n{nt & 255}{'w8' if nt >> 8 else ''}s{sp}:{t:6.1f} {N * K * 2 / t / 1e6:4.2f}")
                except RuntimeError:
                    res.append(f"n{nt & 255}{'w8' if nt >> 8 else ''}s{sp}: n/a")
            print(f"{name:16s} M={M:3d} " + " | ".join(res), flush=True)
        del ws_


main()
