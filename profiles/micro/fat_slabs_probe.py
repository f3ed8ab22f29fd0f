"""Round 6 (end): down_proj / o_proj of the 70B verify (M = 8) as FEWER THAN 256 fat workgroups with the K range split over gridDim.y into
fp32 slabs (DEEP kernel, 2 row groups x tpw tiles per workgroup; bits 16..23 of ssd_gemm_wf_cfg's nt) -- 224 workgroups leave 32 CUs to
the co-located draft like the tuned gate_up does.  Kernel time of GEMM and of the norm that consumes rows / slabs; hipGraph of 24 launches
over 3 weight copies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16
M, h, qn, I = 8, 8192, 8192, 28672
COPIES, REPS = 3, 24


def timed(body):
    body(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            body(i % COPIES)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


@torch.inference_mode()
def main():
    nw = torch.ones(h, dtype=BF, device="cuda")
    res = torch.randn(M, h, device="cuda").to(BF)
    xf = torch.zeros(H.frag_numel(M, h), dtype=BF, device="cuda")
    for name, K in (("down_proj", I), ("o_proj", qn)):
        ws = [torch.empty(h * K, dtype=BF, device="cuda").normal_(0, 0.02) for _ in range(COPIES)]
        x = torch.empty(H.frag_numel(M, K), dtype=BF, device="cuda").normal_(0, 0.5)
        y = torch.zeros(M, h, dtype=BF, device="cuda")
        parts = torch.zeros(16 * M * h, dtype=torch.float32, device="cuda")
        H.gemm(x, ws[0], y, M, h, K, h)
        torch.cuda.synchronize()
        ref = y.float().clone()
        t0 = timed(lambda i: H.gemm(x, ws[i], y, M, h, K, h))
        tn = timed(lambda i: H.rmsnorm(y, nw, 1e-5, M, h, res_in=res, res_out=res, out_frag=xf))
        print(f"{name} [{h}x{K}] default rows: {t0:6.2f} us + norm over rows {tn:5.2f} us = {t0 + tn:6.2f}", flush=True)
        for S, tpw, wv in ((7, 8, 8), (7, 4, 8), (4, 8, 8), (3, 8, 8), (2, 8, 8), (7, 8, 4), (14, 16, 8)):
            cfg = (2 | 256 | S << 16, wv | tpw << 8)
            try:
                parts.zero_()
                H.gemm(x, ws[0], parts, M, h, K, h, epilogue=H.EPI_ROWS_F32, cfg=cfg)
                torch.cuda.synchronize()
            except RuntimeError as e:
                print(f"   S={S} tpw={tpw} waves={wv}: n/a ({e})")
                continue
            got = parts[: S * M * h].view(S, M, h).sum(0)
            err = float((got - ref).abs().max()) / float(ref.pow(2).mean().sqrt())
            tg = timed(lambda i: H.gemm(x, ws[i], parts, M, h, K, h, epilogue=H.EPI_ROWS_F32, cfg=cfg))
            tp = timed(lambda i: H.rmsnorm_parts(parts, S, M, nw, 1e-5, M, h, res_in=res, res_out=res, out_frag=xf))
            wgs = (h // 16 // 2 + tpw - 1) // tpw * S
            print(f"   slabs S={S:2d} tpw={tpw:2d} waves={wv} ({wgs:3d} workgroups): {tg:6.2f} us + norm over slabs {tp:5.2f} us = {tg + tp:6.2f}   (max |sum of slabs - rows| / rms {err:.1e})", flush=True)
        del ws


main()
