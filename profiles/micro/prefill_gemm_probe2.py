"""Prefill-chunk GEMMs at M = 128, second sweep: W k-steps in flight per wave (8 | 16), k-steps per barrier (1 | 2), waves per
workgroup (4 | 7 | 8) and K splits, with the epilogue each GEMM has in the real prefill (qkv rows, o / down fp32 partials,
gate_up SiLU).  Every configuration is first compared bit for bit with the default configuration at the same split count.
hipGraph replays rotating over 4 weight copies (nothing L2/MALL-resident)."""
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
    shapes = [("70b qkv", 10240, 8192, H.EPI_ROWS, (2, 4, 8)), ("70b o", 8192, 8192, H.PF_EPI_PARTIALS, (2, 4, 8)),
              ("70b gate_up", 57344, 8192, H.EPI_SILU_FRAG, (1,)), 
              ("70b down", 8192, 28672, H.PF_EPI_PARTIALS, (2, 4, 8)),
              ("8b qkv", 6144, 4096, H.EPI_ROWS, (2, 4, 8)), ("8b o", 4096, 4096, H.PF_EPI_PARTIALS, (4, 8)),
              ("8b gate_up", 28672, 4096, H.EPI_SILU_FRAG, (1, 2)), ("8b down", 4096, 14336, H.PF_EPI_PARTIALS, (4, 8))]
    C = 4
    M = 128
    for name, N, K, epi, split_list in shapes:
        ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(C)]
        xf = (torch.randn(H.frag_numel(M, K), device="cuda") * 0.05).to(BF)
        y = torch.zeros(M * N, device="cuda", dtype=BF)
        wsb = torch.zeros(16 * M * N, dtype=torch.float32, device="cuda")

        def out_of(nt, sp):
            y.zero_(); wsb.zero_()
            H.gemm_pf(xf, ws_[0], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt)
            torch.cuda.synchronize()
            return (wsb[: sp * M * N].clone() if epi == H.PF_EPI_PARTIALS else y.clone())

        print(f"--- {name}: N={N} K={K} weights {N * K * 2 / 1e6:.0f} MB, epilogue {epi}", flush=True)
        for sp in split_list:
            ref = None
            row = []
            for ntile in (2, 1):
                if ntile == 1 and epi == H.EPI_SILU_FRAG:
                    continue
                for waves in (4, 8, 7, 5, 3, 6):
                    if N % (16 * ntile * waves) or (waves not in (4, 8) and ntile == 1):
                        continue
                    for uu in (8,):
                        for bps in (1, 2):
                            nt = ntile | waves << 8 | uu << 16 | bps << 24
                            tag = f"n{ntile}w{waves}u{uu}b{bps}"
                            try:
                                o = out_of(nt, sp)
                            except RuntimeError:
                                continue
                            if ref is None:
                                ref = o
                            same = bool(torch.equal(o, ref))
                            t = graph_time(lambda: [H.gemm_pf(xf, ws_[i % C], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt)
                                                    for i in range(8)], 8)
                            row.append((t, f"{tag}:{t:6.1f}us {N * K * 2 / t / 1e6:4.2f}TB/s{'' if same else ' MISMATCH'}"))
            row.sort()
            print(f"  s{sp}: " + " | ".join(r[1] for r in row[:10]), flush=True)
        del ws_


main()
