"""Prefill-chunk GEMMs at M = 128, default launch shapes: the plain kernel against its BPRE form (bpre1: all B operands of a k-step read
from LDS before its MFMAs; bpre2: of both k-steps of a phase; bits 28-29 of ssd_gemm_pf_cfg's nt / SSD_PF_BPRE=1|2).  Every output is compared bit for bit with the plain
form's.  hipGraph replays rotating over 4 weight copies (nothing L2 / MALL resident)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16


def graph_time(body, n_inner, reps=8):
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
    shapes = [("70b qkv", 10240, 8192, H.EPI_ROWS, 4, 5), ("70b o", 8192, 8192, H.PF_EPI_PARTIALS, 4, 4),
              ("70b gate_up", 57344, 8192, H.EPI_SILU_FRAG, 1, 8), ("70b down", 8192, 28672, H.PF_EPI_PARTIALS, 8, 8),
              ("8b gate_up", 28672, 4096, H.EPI_SILU_FRAG, 1, 4), ("8b down", 4096, 14336, H.PF_EPI_PARTIALS, 8, 4)]
    C = 4
    ok = True
    for M in (128, 100):
        for name, N, K, epi, sp, waves in shapes:
            ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(C)]
            xf = (torch.randn(H.frag_numel(M, K), device="cuda") * 0.05).to(BF)
            y = torch.zeros(M * N, device="cuda", dtype=BF)
            wsb = torch.zeros(sp * M * N + 64, dtype=torch.float32, device="cuda")
            ref, row = None, []
            for bpre, tag in ((0, "plain"), (2, "bpre2"), (0, "plain again")):      # (bpre1 measured in round 4 and removed from the library)
                nt = 2 | waves << 8 | 8 << 16 | 2 << 24 | bpre << 28
                y.zero_(); wsb.zero_()
                H.gemm_pf(xf, ws_[0], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt)
                torch.cuda.synchronize()
                o = wsb[: sp * M * N].clone() if epi == H.PF_EPI_PARTIALS else y.clone()
                ref = o if ref is None else ref
                same = bool(torch.equal(o, ref))
                ok = ok and same
                t = graph_time(lambda: [H.gemm_pf(xf, ws_[i % C], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt) for i in range(8)], 8)
                row.append(f"{tag}:{t:6.1f}us {N * K * 2 / t / 1e6:4.2f}TB/s{'' if same else ' MISMATCH'}")
            print(f"M={M:3d} {name:12s} w{waves} s{sp}: " + " | ".join(row), flush=True)
            del ws_
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH FOUND")


main()
