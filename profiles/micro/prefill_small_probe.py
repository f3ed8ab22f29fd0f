"""Round 6: the non-GEMM launches of a 70B prefill layer at T = 128 (one chunk) -- what they cost one by one.
  * ssd_rope_store_kv_parts over the QKV GEMM's 4 fp32 slabs (RoPE + KV store)
  * ssd_rmsnorm_parts over o_proj's 4 / down_proj's 8 slabs (residual add + RMSNorm)
  * ssd_attn_paged, causal varlen prefill of 128 tokens: waves per workgroup x (two | one) row tiles per workgroup
hipGraph of 20 launches rotating over 4 buffer sets (~90 MB: not L2-resident, mostly Infinity-Cache resident -- as in the real prefill,
where the producer GEMM has just written the slabs).  Bit-identity of every attention variant against the default is printed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.hip import ops as H  # noqa: E402
from ssd_amd.model import make_cos_sin  # noqa: E402

BF = torch.bfloat16
T, h, nh, nkv, hd, bs = 128, 8192, 64, 8, 128, 256
if len(sys.argv) > 1:       # attention only, another geometry:  prefill_small_probe.py T nh nkv hd
    T, nh, nkv, hd = (int(a) for a in sys.argv[1:5])
NQ = (nh + 2 * nkv) * hd
REPS, C = 20, 4


def timed(body):
    body(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            body(i % C)
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
    dev = "cuda"
    if len(sys.argv) > 1:
        print(f"--- attention only: T {T}, nh {nh}, nkv {nkv}, hd {hd}")
        return attention(dev)
    pos = torch.arange(T, dtype=torch.int64, device=dev)
    slots = torch.arange(T, dtype=torch.int32, device=dev)
    cs = make_cos_sin(hd, 1024, 5e5, dev)
    parts = [torch.randn(4 * T * NQ, device=dev) for _ in range(C)]
    q = torch.zeros(T, nh * hd, dtype=BF, device=dev)
    kc = torch.zeros(2, nkv, bs, hd, dtype=BF, device=dev)
    vc = torch.zeros_like(kc)
    t = timed(lambda i: H.rope_store_kv_parts(parts[i], 4, pos, cs, slots, q, kc, vc, T, nh, nkv, hd, bs, qkv_perm=1))
    print(f"rope_store_kv_parts  S=4 T={T} [{NQ}]: {t:6.2f} us  ({4 * T * NQ * 4 / 1e6:.1f} MB of slabs)", flush=True)
    del parts
    nw = torch.ones(h, dtype=BF, device=dev)
    res = torch.randn(T, h, device=dev).to(BF)
    xf = torch.zeros(H.frag_numel(T, h), dtype=BF, device=dev)
    for S in (4, 8):
        ps = [torch.randn(S * T * h, device=dev) for _ in range(C)]
        t = timed(lambda i: H.rmsnorm_parts(ps[i], S, T, nw, 1e-5, T, h, res_in=res, res_out=res, out_frag=xf))
        print(f"rmsnorm_parts        S={S} T={T} [{h}]: {t:6.2f} us  ({S * T * h * 4 / 1e6:.1f} MB of slabs)", flush=True)
        del ps
    attention(dev)


@torch.inference_mode()
def attention(dev):
    # attention: T-token causal prefill over the cache the same forward has just filled
    qs = [torch.randn(T, nh * hd, device=dev).to(BF) for _ in range(C)]
    nb = (T + bs - 1) // bs + 1
    kcs = [(torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).to(BF) for _ in range(C)]
    vcs = [(torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).to(BF) for _ in range(C)]
    bt = torch.arange(8, dtype=torch.int32, device=dev).clamp_(max=nb - 1).view(1, 8)
    ctx = torch.full((1,), T, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
    af = torch.zeros(H.frag_numel(T, nh * hd), dtype=BF, device=dev)
    scale = hd ** -0.5
    ref = None
    for rt1 in (0, 4):
        for wv in (1, 2, 4, 8):
            def att(i, wv=wv, rt1=rt1):
                H.attn_paged(qs[i], kcs[i], vcs[i], bt, 8, ctx, 1, T, T, nh, nkv, hd, bs, scale, cu_q=cu, mode=H.MODE_CAUSAL, flags=rt1,
                             out_frag=af, waves=wv)
            att(0)
            torch.cuda.synchronize()
            o = af.clone()
            if ref is None and wv == 2:
                ref = o
            t = timed(att)
            same = "" if ref is None else (" == default" if torch.equal(o, ref) else f" max|d| vs default {float((o.float() - ref.float()).abs().max()):.3g}")
            print(f"attn prefill T={T}: {'one row tile ' if rt1 else 'two row tiles'} per workgroup, {wv} waves: {t:6.2f} us{same}", flush=True)


main()
