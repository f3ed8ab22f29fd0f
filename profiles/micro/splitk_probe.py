"""Cross-workgroup split-K (csrc/gemm_sk.hip) against the plain skinny GEMM for the low-N shapes: hipGraph replays over
rotating weight copies."""
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
    L = 16
    shapes = (("1b head", 128256, 2048), ("8b head", 128256, 4096), ("70b/tp8 head", 16032, 8192), ("1b down", 2048, 8192), ("1b o", 2048, 2048), ("70b/tp8 qkv-size", 1280, 8192), ("8b o", 4096, 4096),
              ("8b down", 4096, 14336), ("70b o", 8192, 8192), ("70b down", 8192, 28672), ("70b/tp8 o", 8192, 1024),
              ("70b/tp8 down", 8192, 3584), ("70b/tp4 down", 8192, 7168))
    for name, N, K in shapes:
        L = max(3, min(16, int(1.2e9 // (N * K * 2))))
        for M in ((1, 7) if 'head' in name else (7,)):
            ws_ = [torch.randn(N * K, device="cuda").to(BF) for _ in range(L)]
            x = torch.randn(H.frag_numel(16, K), device="cuda").to(BF)
            y = torch.zeros(16, N, device="cuda", dtype=BF)
            counters = torch.zeros(N // 16, dtype=torch.int32, device="cuda")
            wsb = torch.zeros((N // 16) * 8 * 256, dtype=torch.float32, device="cuda")
            res = [f"plain:{graph_time(lambda: [H.gemm(x, ws_[i], y, M, N, K, N) for i in range(L)], L):5.2f}"]
            for S in (1,):
                for waves in (4, 8, 16):
                    t = graph_time(lambda: [H.gemm_splitk(x, ws_[i], y, M, N, K, N, S, waves, wsb, counters) for i in range(L)], L)
                    res.append(f"S{S}w{waves}:{t:5.2f}")
            print(f"{name:18s} M={M} " + " ".join(res), flush=True)


main()
