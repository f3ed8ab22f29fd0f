"""RMSNorm launch time by threads per row (SSD_NORM_THREADS=256|1024 forces one; default = csrc/common.h ssd_norm_threads).
    for t in 256 1024; do SSD_NORM_THREADS=$t python profiles/micro/norm_probe.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16


@torch.inference_mode()
def main():
    tag = os.environ.get("SSD_NORM_THREADS", "default")
    for T, Hd, S in ((8, 8192, 0), (7, 8192, 0), (8, 4096, 0), (8, 5120, 0), (1, 2048, 0), (24, 2048, 4), (8, 2048, 4), (128, 8192, 0)):
        x = torch.randn(T, Hd, device="cuda").to(BF)
        res = torch.randn(T, Hd, device="cuda").to(BF)
        w = torch.ones(Hd, device="cuda", dtype=BF)
        ro = torch.zeros_like(res)
        xf = torch.zeros(H.frag_numel(T, Hd), device="cuda", dtype=BF)
        parts = torch.randn(max(S, 1), T, Hd, device="cuda")

        def one():
            if S:
                H.rmsnorm_parts(parts, S, T, w, 1e-5, T, Hd, res_in=res, res_out=ro, out_frag=xf)
            else:
                H.rmsnorm(x, w, 1e-5, T, Hd, res_in=res, res_out=ro, out_frag=xf)
        one()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(32):
                one()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"threads={tag:8s} T={T:4d} H={Hd:5d} slabs={S}: {e0.elapsed_time(e1) * 1e3 / (8 * 32):6.2f} us per launch", flush=True)


main()
