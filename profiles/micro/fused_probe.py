"""(nt, waves) sweep of the fused decode-layer GEMMs (csrc/gemm_fused.hip) at the 1B draft's shapes, M = 1:
norm prologue + qkv + RoPE/KV-store epilogue, and norm prologue + gate_up + SiLU*mul epilogue.
hipGraph replays rotating over the 16 layers' worth of weight copies."""
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
    h, nh, nkv, hd, inter, L, M = 2048, 32, 8, 64, 8192, 16, 1
    dev = "cuda"
    hrows = torch.randn(16, h, device=dev).to(BF)
    res = torch.randn(16, h, device=dev).to(BF)
    res2 = torch.zeros(16, h, device=dev, dtype=BF)
    nw = torch.ones(h, device=dev, dtype=BF)
    pos = torch.zeros(16, dtype=torch.int64, device=dev)
    slots = torch.full((16,), -1, dtype=torch.int32, device=dev)
    cs = torch.randn(4096, hd, device=dev)
    q = torch.zeros(16, nh * hd, device=dev, dtype=BF)
    kc = torch.zeros(4, nkv, 256, hd, device=dev, dtype=BF)
    vc = torch.zeros_like(kc)
    qkv_n = (nh + 2 * nkv) * hd
    wq = [torch.randn(qkv_n * h, device=dev).to(BF) for _ in range(L)]
    wg = [torch.randn(2 * inter * h, device=dev).to(BF) for _ in range(L)]
    actf = torch.zeros(H.frag_numel(16, inter), device=dev, dtype=BF)
    for nt in (1, 2, 4):
        for waves in (4, 8, 16):
            try:
                t = graph_time(lambda: [H.gemm_fused(wq[i], M, qkv_n, h, H.FEPI_QKV_ROPE, h_rows=hrows, res_in=res, res_out=res2,
                                                     norm_w=nw, eps=1e-5, positions=pos, cos_sin=cs, slots=slots, q_out=q,
                                                     k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=256, nt=nt,
                                                     waves=waves) for i in range(L)], L)
                print(f"qkv+norm+rope  nt={nt} waves={waves:2d}: {t:6.2f} us", flush=True)
            except RuntimeError as e:
                print(f"qkv nt={nt} waves={waves}: n/a")
    for nt in (2, 4):
        for waves in (4, 8, 16):
            try:
                t = graph_time(lambda: [H.gemm_fused(wg[i], M, 2 * inter, h, H.FEPI_SILU_FRAG, h_rows=hrows, res_in=res, res_out=res2,
                                                     norm_w=nw, eps=1e-5, y=actf, nt=nt, waves=waves) for i in range(L)], L)
                print(f"gate_up+norm+silu nt={nt} waves={waves:2d}: {t:6.2f} us", flush=True)
            except RuntimeError:
                print(f"gate_up nt={nt} waves={waves}: n/a")


main()
