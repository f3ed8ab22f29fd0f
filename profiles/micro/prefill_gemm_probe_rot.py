"""EXPERIMENT (round 6): the prefill GEMM with every workgroup's walk over its k-steps started at its own offset (wrapping), bit 30 of
ssd_gemm_pf_cfg's nt -- the same bytes and products, the workgroups de-phased (profiles/r06_campat*.txt: pure reads gain 7-15 %).  Another
summation order per workgroup: outputs compared with a tolerance.  Needs profiles/r06_pf_rot.patch applied."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16


def graph_time(body, n_inner, reps=8):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
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
    shapes = [("70b qkv", 10240, 8192, H.PF_EPI_PARTIALS, 4, 5), ("70b o", 8192, 8192, H.PF_EPI_PARTIALS, 4, 4),
              ("70b gate_up", 57344, 8192, H.EPI_SILU_FRAG, 1, 8), ("70b down", 8192, 28672, H.PF_EPI_PARTIALS, 8, 8),
              ("8b gate_up", 28672, 4096, H.EPI_SILU_FRAG, 1, 4), ("8b down", 4096, 14336, H.PF_EPI_PARTIALS, 8, 4)]
    C, M = 4, 128
    for name, N, K, epi, sp, waves in shapes:
        ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(C)]
        xf = (torch.randn(H.frag_numel(M, K), device="cuda") * 0.05).to(BF)
        y = torch.zeros(M * N, device="cuda", dtype=BF)
        wsb = torch.zeros(sp * M * N + 64, dtype=torch.float32, device="cuda")
        row, ref = [], None
        for rot, tag in ((0, "real"), (1, "de-phased"), (0, "real again"), (1, "de-phased again")):
            nt = 2 | waves << 8 | 8 << 16 | 2 << 24 | 2 << 28 | rot << 30
            y.zero_(); wsb.zero_()
            H.gemm_pf(xf, ws_[0], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt)
            torch.cuda.synchronize()
            o = wsb[: sp * M * N].view(sp, M * N).sum(0) if epi == H.PF_EPI_PARTIALS else y.float().clone()
            if ref is None:
                ref = o
            err = float((o - ref).abs().max()) / float(ref.pow(2).mean().sqrt())
            t = graph_time(lambda: [H.gemm_pf(xf, ws_[i % C], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt) for i in range(8)], 8)
            row.append(f"{tag}: {t:6.1f}us {N * K * 2 / t / 1e6:4.2f}TB/s" + (f" (rel err {err:.1e})" if rot else ""))
        print(f"M={M:3d} {name:12s} w{waves} s{sp}: " + " | ".join(row), flush=True)
        del ws_


main()
