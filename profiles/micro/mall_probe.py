"""Does the 256 MiB Infinity Cache (MALL) serve re-reads of a weight matrix?  Time the M=7 skinny GEMM over a matrix
that is re-read every launch (1 copy), over 2 alternating copies, and over enough copies to defeat any cache; for
matrices from 34 MB to 470 MB.  If re-reads are fast, prefetching the next layer's weights during the latency-bound
attention launch would pay."""
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
    M = 7
    for N, K in ((4096, 4096), (8192, 4096), (8192, 8192), (8192, 28672)):
        mb = N * K * 2 / 1e6
        ncold = max(3, int(1200 // mb) + 1)
        ws = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(ncold)]
        x = torch.randn(H.frag_numel(M, K), device="cuda").to(BF)
        y = torch.zeros(16, N, device="cuda", dtype=BF)
        res = []
        for copies in (1, 2, ncold):
            t = graph_time(lambda: [H.gemm(x, ws[i % copies], y, M, N, K, N) for i in range(12)], 12)
            res.append(f"{copies} cop{'y' if copies == 1 else 'ies'}: {t:6.1f} us ({mb / t / 1e6 * 1e6 / 1e3:5.2f} TB/s)")
        print(f"{N}x{K} {mb:6.1f} MB  " + " | ".join(res), flush=True)
        del ws


main()
