"""Prefill-chunk GEMMs at M = 128: the default launch shapes against the KS = 2 form (4 row slots x 2 k slots = 8 waves per workgroup,
the k slots' accumulators combined in LDS; bit 30 of ssd_gemm_pf_cfg's nt).  The K order changes (two half-range sums per split), so the
outputs are compared with a tolerance: max |sum of slabs - sum of slabs of the default form| relative to the output's rms.
hipGraph replays rotating over 4 weight copies (nothing L2 / MALL resident).
MEASURED SLOWER AND REMOVED FROM THE LIBRARY (profiles/r06_pf_probe_ks.txt, r06_pf_probe_ks_interleaved.txt, r06_ktrace_pf_ks.txt): this
probe needs profiles/r06_pf_ks.patch applied (git apply profiles/r06_pf_ks.patch; make -C ssd_amd/csrc)."""
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


def cfg(waves, bpre=2, ks=1, nt=2):
    return nt | waves << 8 | 8 << 16 | 2 << 24 | bpre << 28 | (1 << 30 if ks == 2 else 0)


@torch.inference_mode()
def main():
    # name, N, K, epilogue, default (splits, waves), KS candidates (splits, bpre)
    shapes = [("70b o", 8192, 8192, H.PF_EPI_PARTIALS, (4, 4), [(4, 2), (2, 2), (4, 0), (8, 2)]),
              ("70b down", 8192, 28672, H.PF_EPI_PARTIALS, (8, 8), [(4, 2), (8, 2), (2, 2), (4, 0)]),
              ("70b qkv", 10240, 8192, H.PF_EPI_PARTIALS, (4, 5), [(4, 2), (2, 2)]),
              ("70b gate_up", 57344, 8192, H.EPI_SILU_FRAG, (1, 8), [(1, 2)]),
              ("8b o", 4096, 4096, H.PF_EPI_PARTIALS, (8, 4), [(8, 2), (4, 2)]),
              ("8b down", 4096, 14336, H.PF_EPI_PARTIALS, (8, 4), [(8, 2), (4, 2)])]
    C = 4
    for M in (128, 100):
        for name, N, K, epi, (sp0, w0), cands in shapes:
            ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(C)]
            xf = (torch.randn(H.frag_numel(M, K), device="cuda") * 0.05).to(BF)
            y = torch.zeros(M * N, device="cuda", dtype=BF)
            wsb = torch.zeros(16 * M * N + 64, dtype=torch.float32, device="cuda")

            def run(nt, sp):
                y.zero_(); wsb.zero_()
                H.gemm_pf(xf, ws_[0], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt)
                torch.cuda.synchronize()
                if epi == H.PF_EPI_PARTIALS:
                    return wsb[: sp * M * N].view(sp, M * N).sum(0)
                return y.float().clone()

            def t_of(nt, sp):
                return graph_time(lambda: [H.gemm_pf(xf, ws_[i % C], y, M, N, K, N, wsb, epilogue=epi, splits=sp, nt=nt) for i in range(8)], 8)

            ref = run(cfg(w0), sp0)
            rms = float(ref.pow(2).mean().sqrt())
            row = [f"default w{w0} s{sp0}: {t_of(cfg(w0), sp0):6.1f}us"]
            for sp, bpre in cands:
                try:
                    o = run(cfg(8, bpre, 2), sp)
                    err = float((o - ref).abs().max()) / rms
                    t = t_of(cfg(8, bpre, 2), sp)
                    row.append(f"ks2 s{sp} bpre{bpre}: {t:6.1f}us {N * K * 2 / t / 1e6:4.2f}TB/s (rel err {err:.1e})")
                except RuntimeError as e:
                    row.append(f"ks2 s{sp} bpre{bpre}: n/a")
            row.append(f"default again: {t_of(cfg(w0), sp0):6.1f}us")
            print(f"M={M:3d} {name:12s} " + " | ".join(row), flush=True)
            del ws_


main()
