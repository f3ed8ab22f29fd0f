"""What does one dependent launch cost inside a hipGraph on MI355X?  Graphs of N launches through the C ABI:
same tiny kernel repeated, alternating kernels, small GEMMs with L2-resident vs streaming weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

dev = "cuda"
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
    N = 200
    H_ = 2048
    x = torch.randn(1, H_, device=dev).to(BF)
    res = torch.randn(1, H_, device=dev).to(BF)
    w = torch.ones(H_, device=dev, dtype=BF)
    out = torch.zeros(1, H_, device=dev, dtype=BF)
    outf = torch.zeros(H.frag_numel(16, H_), device=dev, dtype=BF)
    ids = torch.zeros(1, dtype=torch.int64, device=dev)
    pos = torch.zeros(1, dtype=torch.int64, device=dev)
    nxt = torch.zeros(1, dtype=torch.int64, device=dev)
    slots = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = torch.ones(1, dtype=torch.int32, device=dev)
    bt = torch.zeros(1, 8, dtype=torch.int32, device=dev)
    spec = torch.zeros(1, N + 2, dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    emb = torch.randn(1024, H_, device=dev).to(BF)

    def adv():
        H.draft_advance(nxt, ids, pos, slots, ctx, bt, 8, 256, spec, N, step, 1)

    def norm():
        H.rmsnorm(x, w, 1e-5, 1, H_, res_in=res, res_out=res, out_rows=out, out_frag=outf)

    def embed():
        H.embedding(ids, emb, out, 1, H_)

    r = {}
    r["draft_advance x N"] = graph_time(lambda: [adv() for _ in range(N)], N)
    r["rmsnorm x N"] = graph_time(lambda: [norm() for _ in range(N)], N)
    r["embedding x N"] = graph_time(lambda: [embed() for _ in range(N)], N)
    r["alternate advance/rmsnorm"] = graph_time(lambda: [(adv(), norm()) for _ in range(N // 2)], N)
    r["torch add x N"] = graph_time(lambda: [out.add_(1) for _ in range(N)], N)

    # GEMM M=1, N=K=2048 (8.4 MB): one matrix reused (L2/MALL resident) vs 64 distinct matrices (536 MB rotation)
    K = Nn = 2048
    wf = [torch.randn(Nn * K, device=dev).to(BF) for _ in range(64)]
    xf = torch.randn(H.frag_numel(16, K), device=dev).to(BF)
    y = torch.zeros(16, Nn, device=dev, dtype=BF)
    r["gemm 2048x2048 same W x N"] = graph_time(lambda: [H.gemm(xf, wf[0], y, 1, Nn, K, Nn) for _ in range(N)], N)
    r["gemm 2048x2048 rotating W x N"] = graph_time(lambda: [H.gemm(xf, wf[i % 64], y, 1, Nn, K, Nn) for i in range(N)], N)
    r["gemm rotating + rmsnorm alternate"] = graph_time(lambda: [(H.gemm(xf, wf[i % 64], y, 1, Nn, K, Nn), norm()) for i in range(N // 2)], N)
    for k, v in r.items():
        print(f"{k:40s} {v:7.2f} us per launch")


main()
