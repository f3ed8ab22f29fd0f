"""Does the split-K partial-slab GEMM (csrc/gemm_sk.hip gemm_sp_kernel) also beat the rows kernels on the TARGET's o_proj /
down_proj (8B, 70B, Qwen3-32B shapes, M = 8)?  us per launch over rotating weight copies (nothing cache-resident).
    python profiles/micro/parts_probe.py > gpurun_out/parts_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16
SHAPES = [("8b.o", 4096, 4096), ("8b.down", 4096, 14336), ("70b.o", 8192, 8192), ("70b.down", 8192, 28672),
          ("q32b.o", 5120, 8192), ("q32b.down", 5120, 25600), ("70b/tp4.o", 8192, 2048), ("70b/tp4.down", 8192, 7168)]


@torch.inference_mode()
def timed(fn, n, reps=4):
    fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


@torch.inference_mode()
def main():
    M = 8
    for name, N, K in SHAPES:
        mb = N * K * 2 / 1e6
        copies = max(4, min(32, int(800 // mb) + 1))
        ws = [torch.randn(N * K // 2, device="cuda", dtype=torch.float32).view(BF).view(-1)[:N * K].contiguous() for _ in range(copies)]
        x = torch.randn(H.frag_numel(M, K), device="cuda").to(BF)
        y = torch.zeros(16 * N, device="cuda", dtype=BF)
        parts = torch.zeros(16 * M * N, device="cuda", dtype=torch.float32)
        base = timed(lambda i: H.gemm(x, ws[i % copies], y, M, N, K, N), copies)
        line = f"{name:13s} {mb:7.1f} MB  rows default {base:7.2f} us ({mb / base:5.2f} TB/s) |"
        best = None
        KT = K // 32
        for S in (1, 2, 4, 8, 16):
            for waves in (8, 16):
                per = -(-KT // S)
                if KT < S or -(-per // waves) > 8:
                    continue
                t = timed(lambda i: H.gemm_parts(x, ws[i % copies], M, N, K, parts=parts, splits=S, waves=waves), copies)
                line += f" S{S}w{waves}:{t:.1f}"
                if best is None or t < best[0]:
                    best = (t, S, waves)
        if best:
            line += f"  | best parts S{best[1]}w{best[2]} {best[0]:.2f} us ({mb / best[0]:.2f} TB/s)"
        print(line, flush=True)
        del ws
        torch.cuda.empty_cache()


main()
