"""Sweep the (row groups per workgroup, K-split waves) decomposition of the skinny GEMM for the shapes of the
benchmark models, timing hipGraph replays that rotate through enough weight copies to stay out of the caches.
Run on the GPU box:  python profiles/tune_gemm.py > gpurun_out/tune.json   (prints a table on stderr)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16
SHAPES = {
    "1b": [(3072, 2048), (2048, 2048), (16384, 2048), (2048, 8192), (128256, 2048)],
    "8b": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)],
    "70b_tp1": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672)],
    "70b_tp2": [(5120, 8192), (8192, 4096), (28672, 8192), (8192, 14336), (64128, 8192)],
    "70b_tp4": [(2560, 8192), (8192, 2048), (14336, 8192), (8192, 7168), (32064, 8192)],
    "70b_tp8": [(1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584), (16032, 8192)],
    # BASELINE.json configs[4]: Qwen3-32B (h 5120, I 25600, 64/8 heads x 128, V 151936) at TP 1 / 4, Qwen3-0.6B draft
    "q32b_tp1": [(10240, 5120), (5120, 8192), (51200, 5120), (5120, 25600), (151936, 5120)],
    "q32b_tp4": [(2560, 5120), (5120, 2048), (12800, 5120), (5120, 6400), (37984, 5120)],
    "q06b": [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072), (151936, 1024)],
}
MS = {"1b": [1, 24], "8b": [7], "70b_tp1": [7], "70b_tp2": [7], "70b_tp4": [8], "70b_tp8": [8], "q32b_tp1": [8], "q32b_tp4": [8],
      "q06b": [1, 24]}
TPWS = (1, 2, 3, 4, 7, 8)      # consecutive tiles per workgroup (persistent variant), encoded in bits 8.. of `waves`
if len(sys.argv) > 1:
    SHAPES = {k: v for k, v in SHAPES.items() if k in sys.argv[1:]}


@torch.inference_mode()
def time_cfg(ws, x, y, M, N, K, epi, cfg, reps=3):
    def body():
        for w in ws:
            H.gemm(x, w, y, M, N, K, 0 if epi == H.EPI_SILU_FRAG else N, epi, cfg=cfg)
    try:
        body()
    except Exception:
        return None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * len(ws))


def main():
    out = {}
    for fam, shapes in SHAPES.items():
        for (N, K) in shapes:
            bytes_ = N * K * 2
            copies = max(2, min(64, int(700e6 // bytes_) + 1))
            ws = [torch.randn(N * K // 2, device="cuda", dtype=torch.float32).view(BF).view(-1)[:N * K].contiguous() for _ in range(copies)]
            for M in MS[fam]:
                x = torch.randn(H.frag_numel(M, K), device="cuda").to(BF)
                is_gu = (N, K) in [(16384, 2048), (28672, 4096), (57344, 8192), (28672, 8192), (14336, 8192), (7168, 8192),
                                   (51200, 5120), (12800, 5120), (6144, 1024)]
                epi = H.EPI_SILU_FRAG if is_gu else H.EPI_ROWS
                y = torch.zeros(max(M, 16) * N, device="cuda", dtype=BF)
                best = None
                rows = []
                for nt in (1, 2, 4):
                    if epi == H.EPI_SILU_FRAG and nt == 1:
                        continue
                    if (N // 16) % nt:
                        continue
                    if nt == 4 and M > 32:
                        continue
                    for waves in (4, 8, 16):
                        for tpw in TPWS:
                            if tpw > 1 and (N // 16 // nt) // tpw < 128:
                                continue
                            t = time_cfg(ws, x, y, M, N, K, epi, (nt, waves | (tpw << 8)))
                            if t is None:
                                continue
                            rows.append((nt, waves, tpw, t))
                            if best is None or t < best[-1]:
                                best = (nt, waves, tpw, t)
                tdef = time_cfg(ws, x, y, M, N, K, epi, None)
                key = f"{M},{N},{K}"
                out[key] = {"best": list(best[:3]), "best_us": round(best[3] * 1e6, 2), "default_us": round(tdef * 1e6, 2),
                            "GBps_best": round(bytes_ / best[3] / 1e9), "MB": round(bytes_ / 1e6, 1)}
                top = sorted(rows, key=lambda r: r[3])[:6]
                print(f"{fam:8s} M={M:3d} N={N:6d} K={K:5d} {bytes_ / 1e6:7.1f}MB default {tdef * 1e6:7.2f}us  best nt{best[0]},w{best[1]},t{best[2]} "
                      f"{best[3] * 1e6:7.2f}us ({bytes_ / best[3] / 1e9:5.0f} GB/s)  top: "
                      + " ".join(f"{a},{b},{c}:{d * 1e6:.1f}" for a, b, c, d in top)
                      + "  tpw=1 best: " + f"{min(r[3] for r in rows if r[2] == 1) * 1e6:.1f}", file=sys.stderr)
            del ws
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
