import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H
BF = torch.bfloat16
with torch.inference_mode():
    for name, N, K in (("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)):
        ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(3)]
        for M in (48, 128):
            xf = torch.randn(H.frag_numel(M, K), device="cuda").to(BF)
            y = torch.zeros(M, N, device="cuda", dtype=BF)
            wsb = torch.zeros(16 * M * N, dtype=torch.float32, device="cuda")
            for sp in (2, 8, 16):
                for i in range(6):
                    H.gemm_pf(xf, ws_[i % 3], y, M, N, K, N, wsb, splits=sp)
            torch.cuda.synchronize()
        del ws_
