"""(nt, waves) sweep of the RoPE-epilogue GEMM (csrc/gemm_fused.hip, x fragment-major) at the 70B / 8B target qkv shapes, M = 7."""
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
    dev = "cuda"
    for name, h, nh, nkv, hd, L in (("70b", 8192, 64, 8, 128, 6), ("8b", 4096, 32, 8, 128, 16), ("70b/tp8", 8192, 8, 1, 128, 24)):
        M = 7
        qkv_n = (nh + 2 * nkv) * hd
        xf = torch.randn(H.frag_numel(16, h), device=dev).to(BF)
        pos = torch.zeros(16, dtype=torch.int64, device=dev)
        slots = torch.full((16,), -1, dtype=torch.int32, device=dev)
        cs = torch.randn(4096, hd, device=dev)
        q = torch.zeros(16, nh * hd, device=dev, dtype=BF)
        kc = torch.zeros(4, nkv, 256, hd, device=dev, dtype=BF)
        vc = torch.zeros_like(kc)
        wq = [torch.randn(qkv_n * h // 8, device=dev).to(BF).repeat(8) for _ in range(L)]
        res = []
        for nt in (1, 2, 4):
            for waves in (4, 8, 16):
                try:
                    t = graph_time(lambda: [H.gemm_fused(wq[i], M, qkv_n, h, H.FEPI_QKV_ROPE, x_frag=xf, positions=pos, cos_sin=cs,
                                                         slots=slots, q_out=q, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd,
                                                         block_size=256, nt=nt, waves=waves) for i in range(L)], L)
                    res.append(f"{nt},{waves}:{t:.1f}")
                except RuntimeError:
                    pass
        t = graph_time(lambda: [H.gemm_fused(wq[i], M, qkv_n, h, H.FEPI_QKV_ROPE, x_frag=xf, positions=pos, cos_sin=cs, slots=slots,
                                             q_out=q, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=256) for i in range(L)], L)
        print(f"{name} qkv {qkv_n}x{h} ({qkv_n * h * 2 / 1e6:.0f} MB): default {t:.1f} us | " + " ".join(res), flush=True)


main()
