"""Round 6: decomposition sweep of the four layer GEMMs at the 70B TENSOR-PARALLEL shard shapes (TP = 4 and 8, M = 8 verify rows): the
default dispatch against explicit (row groups per workgroup, waves, tiles per workgroup) of gemm_wf_kernel, its DEEP form (bit 8 of nt),
the single-buffered gemm_sk kernel, the split-K slab kernel (gemm_sp, S slabs; its consumer -- the all-reduce + norm -- is not in the
time) and, for QKV, the fused QKV + RoPE + KV-store kernel.  hipGraph of 24 launches rotating over enough weight copies to stay out of the
256 MB Infinity Cache.
    python profiles/micro/tp_shard_tune.py [4|8 ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16
M, h, I_full, nh_full, nkv_full, hd = 8, 8192, 28672, 64, 8, 128
REPS = 24


def timed(body, copies):
    try:
        body(0)
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            body(i % copies)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def show(name, N, K, rows):
    mb = N * K * 2 / 1e6
    rows = [(k, v) for k, v in rows if v is not None]
    d = dict(rows)["default"]
    top = sorted(rows, key=lambda r: r[1])[:8]
    print(f"{name:22s} [{N}x{K}] {mb:6.1f} MB  default {d:6.2f} us ({mb / d:4.2f} TB/s) | " + "  ".join(f"{k}:{v:.2f}" for k, v in top), flush=True)


@torch.inference_mode()
def sweep(tp):
    nh, nkv, I = nh_full // tp, nkv_full // tp, I_full // tp
    qn, qkv_n = nh * hd, (nh + 2 * nkv) * hd
    print(f"==== TP = {tp} shard, M = {M} ====", flush=True)
    sweep_body(nh, nkv, I, qn, qkv_n)


def copies_of(n, k):
    c = max(3, min(24, int(600e6 // (n * k * 2)) + 1))
    return [torch.empty(n * k, dtype=BF, device="cuda").normal_(0, 0.02) for _ in range(c)]


def rows_sweep(name, N, K, xf, epi):
    if True:
        ws = copies_of(N, K)
        C = len(ws)
        y = torch.zeros(16 * max(N, 16), dtype=BF, device="cuda")
        ldy = 0 if epi == H.EPI_SILU_FRAG else N
        rows = [("default", timed(lambda i: H.gemm(xf, ws[i], y, M, N, K, ldy, epilogue=epi), C))]
        for nt in (1, 2, 4):
            if (epi == H.EPI_SILU_FRAG and nt == 1) or (N // 16) % nt:
                continue
            for wv in (4, 8, 16):
                for tpw in (1, 2, 3, 4):
                    if tpw > 1 and (N // 16 // nt) // tpw < 128:
                        continue
                    rows.append((f"nt{nt}w{wv}t{tpw}", timed(lambda i: H.gemm(xf, ws[i], y, M, N, K, ldy, epilogue=epi, cfg=(nt, wv | tpw << 8)), C)))
                    if nt >= 2 and wv <= 8:
                        rows.append((f"DEEPnt{nt}w{wv}t{tpw}", timed(lambda i: H.gemm(xf, ws[i], y, M, N, K, ldy, epilogue=epi, cfg=(nt | 256, wv | tpw << 8)), C)))
        if epi == H.EPI_ROWS:
            for wv in (4, 8, 16):
                rows.append((f"sk_w{wv}", timed(lambda i: H.gemm_splitk(xf, ws[i], y, M, N, K, N, 1, wv, None, None), C)))
            parts = torch.zeros(8 * M * N, dtype=torch.float32, device="cuda")
            for S in (2, 4):
                for wv in (4, 8, 16):
                    rows.append((f"slabs_S{S}w{wv}", timed(lambda i: H.gemm_parts(xf, ws[i], M, N, K, parts=parts, splits=S, waves=wv), C)))
        rows.append(("default_again", timed(lambda i: H.gemm(xf, ws[i], y, M, N, K, ldy, epilogue=epi), C)))      # (the first measurement runs on a cold clock)
        show(name, N, K, rows)
        del ws


@torch.inference_mode()
def sweep_qwen():
    """Qwen3-32B verify shapes at TP = 1 (BASELINE configs[4] on one GPU): q / k-norm model, so QKV is a plain rows GEMM."""
    hq, Iq, qn, qkv_n = 5120, 25600, 8192, 10240
    print(f"==== Qwen3-32B TP = 1, M = {M} ====", flush=True)
    xh = torch.empty(H.frag_numel(M, hq), dtype=BF, device="cuda").normal_(0, 1)
    xq = torch.empty(H.frag_numel(M, qn), dtype=BF, device="cuda").normal_(0, 1)
    xi = torch.empty(H.frag_numel(M, Iq), dtype=BF, device="cuda").normal_(0, 0.5)
    rows_sweep("qkv rows", qkv_n, hq, xh, H.EPI_ROWS)
    rows_sweep("o_proj", hq, qn, xq, H.EPI_ROWS)
    rows_sweep("gate_up+SiLU", 2 * Iq, hq, xh, H.EPI_SILU_FRAG)
    rows_sweep("down_proj", hq, Iq, xi, H.EPI_ROWS)


@torch.inference_mode()
def sweep_body(nh, nkv, I, qn, qkv_n):
    xh = torch.empty(H.frag_numel(M, h), dtype=BF, device="cuda").normal_(0, 1)
    xq = torch.empty(H.frag_numel(M, qn), dtype=BF, device="cuda").normal_(0, 1)
    xi = torch.empty(H.frag_numel(M, I), dtype=BF, device="cuda").normal_(0, 0.5)
    # QKV + RoPE + KV store (fused kernel)
    ws = copies_of(qkv_n, h)
    C = len(ws)
    pos = torch.arange(100, 100 + M, dtype=torch.int64, device="cuda")
    cs = torch.randn(1024, hd, device="cuda")
    slots = torch.arange(M, dtype=torch.int32, device="cuda")
    q_out = torch.zeros(M, nh * hd, dtype=BF, device="cuda")
    kc = torch.zeros(2, nkv, 256, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    rope = dict(positions=pos, cos_sin=cs, slots=slots, q_out=q_out, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=256)
    rows = [("default", timed(lambda i: H.gemm_fused(ws[i], M, qkv_n, h, H.FEPI_QKV_ROPE, x_frag=xh, **rope), C))]
    for nt in (1, 2, 4):
        if (qkv_n // 16) % nt:
            continue
        for wv in (4, 8, 16):
            rows.append((f"nt{nt}w{wv}", timed(lambda i: H.gemm_fused(ws[i], M, qkv_n, h, H.FEPI_QKV_ROPE, x_frag=xh, nt=nt, waves=wv, **rope), C)))
    show("qkv+RoPE+store (fused)", qkv_n, h, rows)
    del ws
    rows_sweep("o_proj", h, qn, xq, H.EPI_ROWS)
    rows_sweep("gate_up+SiLU", 2 * I, h, xh, H.EPI_SILU_FRAG)
    rows_sweep("down_proj", h, I, xi, H.EPI_ROWS)


for a in (sys.argv[1:] or ["4", "8"]):
    if a == "qwen":
        sweep_qwen()
    else:
        sweep(int(a))
    torch.cuda.empty_cache()
