"""Round 6: the 70B verify's skinny GEMMs (M = 8) -- the default dispatch, explicit decompositions and the DEEP form (twice the k-tiles
per stage at <= 8 waves, csrc/gemm.hip).  Each kind is a hipGraph of 24 launches rotating through 3 copies of its weights (nothing
cache-resident); per-launch time = graph time / 24.  (Until commit 2e5cd65 this probe also timed the norm-carrying "xsum" forms: their
results are profiles/r06_xsum_probe_v1..v3*.txt, the code profiles/r06_xsum_v3.patch.)
    python profiles/micro/deep_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402

BF = torch.bfloat16
M, h, qn, I, nh, nkv, hd = 8, 8192, 8192, 28672, 64, 8, 128
N_QKV = (nh + 2 * nkv) * hd
COPIES, REPS = 3, 24


def w(n, k):
    return [torch.empty(n * k, dtype=BF, device="cuda").normal_(0, 0.02) for _ in range(COPIES)]


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
    wo, wgu, wd, wq = w(h, qn), w(2 * I, h), w(h, I), w(N_QKV, h)
    a_f = torch.empty(H.frag_numel(M, qn), dtype=BF, device="cuda").normal_(0, 1)
    act_f = torch.empty(H.frag_numel(M, I), dtype=BF, device="cuda").normal_(0, 0.5)
    xf = torch.empty(H.frag_numel(M, h), dtype=BF, device="cuda").normal_(0, 1)
    y = torch.zeros(M, h, dtype=BF, device="cuda")
    res = torch.randn(M, h, device="cuda").to(BF)
    nw = torch.ones(h, dtype=BF, device="cuda")
    pos = torch.arange(100, 100 + M, dtype=torch.int64, device="cuda")
    cs = torch.randn(1024, hd, device="cuda")
    slots = torch.arange(M, dtype=torch.int32, device="cuda")
    q_out = torch.zeros(M, nh * hd, dtype=BF, device="cuda")
    kc = torch.zeros(2, nkv, 256, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    rope = dict(positions=pos, cos_sin=cs, slots=slots, q_out=q_out, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=256)
    t = {}
    t["o_proj rows (default dispatch)"] = timed(lambda i: H.gemm(a_f, wo[i], y, M, h, qn, h))
    t["down rows (default dispatch)"] = timed(lambda i: H.gemm(act_f, wd[i], y, M, h, I, h))
    t["rmsnorm"] = timed(lambda i: H.rmsnorm(y, nw, 1e-5, M, h, res_in=res, res_out=res, out_frag=xf))
    t["gate_up (default dispatch)"] = timed(lambda i: H.gemm(xf, wgu[i], act_f, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG))
    t["qkv+rope"] = timed(lambda i: H.gemm_fused(wq[i], M, N_QKV, h, H.FEPI_QKV_ROPE, x_frag=xf, **rope))

    # gate_up decompositions: the default (nt 4, 8 waves, 4 tiles per workgroup: 224 workgroups) against 256-workgroup forms
    for nt, wv, tpw in ((4, 8, 4), (2, 8, 7), (2, 16, 7), (4, 16, 4), (4, 8, 7)):
        try:
            t[f"gate_up cfg nt={nt} waves={wv} tpw={tpw} ({(2 * I // 16) // nt // tpw} wgs)"] = timed(
                lambda i: H.gemm(xf, wgu[i], act_f, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG, cfg=(nt, wv | (tpw << 8))))
        except Exception as e:
            print("cfg", nt, wv, tpw, "->", e)

    # DEEP (twice the k-tiles per stage, <= 8 waves, up to 256 VGPRs) against the plain kernel at its default decomposition
    t["gate_up DEEP nt=4 waves=8 tpw=4"] = timed(lambda i: H.gemm(xf, wgu[i], act_f, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG, cfg=(4 | 256, 8 | (4 << 8))))
    t["gate_up DEEP nt=2 waves=8 tpw=7"] = timed(lambda i: H.gemm(xf, wgu[i], act_f, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG, cfg=(2 | 256, 8 | (7 << 8))))
    t["o_proj plain nt=2 waves=8"] = timed(lambda i: H.gemm(a_f, wo[i], y, M, h, qn, h, cfg=(2, 8)))
    t["o_proj DEEP nt=2 waves=8"] = timed(lambda i: H.gemm(a_f, wo[i], y, M, h, qn, h, cfg=(2 | 256, 8)))
    t["o_proj DEEP nt=2 waves=4"] = timed(lambda i: H.gemm(a_f, wo[i], y, M, h, qn, h, cfg=(2 | 256, 4)))
    t["down plain nt=2 waves=8"] = timed(lambda i: H.gemm(act_f, wd[i], y, M, h, I, h, cfg=(2, 8)))
    t["down DEEP nt=2 waves=8"] = timed(lambda i: H.gemm(act_f, wd[i], y, M, h, I, h, cfg=(2 | 256, 8)))
    t["down DEEP nt=4 waves=8 (128 wgs)"] = timed(lambda i: H.gemm(act_f, wd[i], y, M, h, I, h, cfg=(4 | 256, 8)))

    for label, nt, wv in (("plain nt=4 waves=16 (default)", 4, 16), ("plain nt=4 waves=8", 4, 8)):        # (the DEEP form of this kernel was measured and deleted)
        t[f"qkv+rope {label}"] = timed(lambda i: H.gemm_fused(wq[i], M, N_QKV, h, H.FEPI_QKV_ROPE, x_frag=xf, nt=nt, waves=wv, **rope))

    def sep(i):
        H.gemm(a_f, wo[i], y, M, h, qn, h)
        H.rmsnorm(y, nw, 1e-5, M, h, res_in=res, res_out=res, out_frag=xf)
        H.gemm(xf, wgu[i], act_f, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG)
        H.gemm(act_f, wd[i], y, M, h, I, h)
        H.rmsnorm(y, nw, 1e-5, M, h, res_in=res, res_out=res, out_frag=xf)
        H.gemm_fused(wq[i], M, N_QKV, h, H.FEPI_QKV_ROPE, x_frag=xf, **rope)

    t["LAYER minus attention, separate (6 launches)"] = timed(sep)
    for k, v in t.items():
        print(f"{k:48s} {v:8.2f} us")


if __name__ == "__main__":
    main()
