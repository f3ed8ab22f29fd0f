"""GPU parity of the fused decode-layer GEMMs (csrc/gemm_fused.hip) and of the rotation-paired QKV layout against
the oracle chain  add+RMSNorm -> linear -> RoPE -> KV store  /  add+RMSNorm -> linear -> SiLU*mul."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O
from oracle import layout as LY
from tests.util import assert_close_bf16

BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def dev(t):
    return t.cuda().contiguous()


def qkv_perm(nh, nkv, hd):
    half, gph = hd // 2, hd // 16
    idx = []
    for head in range(nh + nkv):
        for j in range(gph):
            idx.extend(head * hd + 8 * j + i for i in range(8))
            idx.extend(head * hd + half + 8 * j + i for i in range(8))
    idx.extend(range((nh + nkv) * hd, (nh + 2 * nkv) * hd))
    return torch.tensor(idx)


def test_qkv_weight_shuffle(H):
    nh, nkv, hd, K = 4, 2, 64, 128
    w = torch.randn((nh + 2 * nkv) * hd, K).to(BF)
    out = torch.zeros(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag_qkv(dev(w), out, nh, nkv, hd, K)
    ref = LY.rows_to_frag_ref(w[qkv_perm(nh, nkv, hd)])
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("xnorm", [True, False])
@pytest.mark.parametrize("M,nh,nkv,hd,K,with_res", [(1, 32, 8, 64, 2048, True), (7, 32, 8, 128, 4096, True), (16, 8, 2, 128, 1024, False),
                                                    (3, 4, 4, 64, 512, True)])
def test_fused_qkv_rope(H, M, nh, nkv, hd, K, with_res, xnorm):
    torch.manual_seed(M * 7 + K)
    bs, nb = 16, 8
    N = (nh + 2 * nkv) * hd
    h = torch.randn(M, K).to(BF)
    res = torch.randn(M, K).to(BF) if with_res else None
    nw = (1 + 0.1 * torch.randn(K)).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    bias = (torch.randn(N) * 0.1).to(BF)
    pos = torch.randint(0, 300, (M,), dtype=torch.int64)
    slots = torch.randperm(nb * bs)[:M].to(torch.int32)
    if M > 2:
        slots[1] = -1
    cache = O.make_cos_sin_cache(hd, 512, 5e5)
    # ---- oracle ----
    if with_res:
        x, res_ref = O.rmsnorm(h, nw, 1e-5, res)
    else:
        x, res_ref = O.rmsnorm(h, nw, 1e-5), h
    qkv = O.linear(x, w, bias)
    q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    q, k = O.rope(pos, q.contiguous(), k.contiguous(), cache, hd)
    kref = torch.zeros(nb, bs, nkv, hd, dtype=BF)
    vref = torch.zeros_like(kref)
    O.store_kv(k.view(M, nkv, hd), v.contiguous().view(M, nkv, hd), kref, vref, slots)
    # ---- HIP ----
    wf = torch.zeros(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag_qkv(dev(w), wf, nh, nkv, hd, K)
    bias_p = dev(bias[qkv_perm(nh, nkv, hd)])
    q_out = torch.zeros(M, nh * hd, dtype=BF, device="cuda")
    kc = torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    res_out = torch.zeros(M, K, dtype=BF, device="cuda")
    common = dict(bias=bias_p, positions=dev(pos), cos_sin=dev(cache), slots=dev(slots), q_out=q_out, k_cache=kc, v_cache=vc,
                  nh=nh, nkv=nkv, hd=hd, block_size=bs)
    if xnorm:
        H.gemm_fused(wf, M, N, K, H.FEPI_QKV_ROPE, h_rows=dev(h), res_in=None if res is None else dev(res), res_out=res_out,
                     norm_w=dev(nw), eps=1e-5, **common)
        assert torch.equal(res_out.cpu().view(torch.int16), res_ref.view(torch.int16))
    else:
        H.gemm_fused(wf, M, N, K, H.FEPI_QKV_ROPE, x_frag=dev(LY.rows_to_frag_ref(x)), **common)
    tol = dict(max_ulp=1, max_frac=0.04, rel_floor=2 ** -7)
    assert_close_bf16(q_out, q, what="fused q", **tol)
    assert_close_bf16(LY.kv_hnd_to_nhd(kc.cpu()), kref, what="fused k cache", **tol)
    assert_close_bf16(LY.kv_hnd_to_nhd(vc.cpu()), vref, what="fused v cache", **tol)
    # every decomposition of the RoPE-epilogue kernel (the tuned default picks nt = 4 for 70B-class shapes)
    # (a 1-ulp flip of a pre-RoPE value moves x*cos - y*sin by up to that ulp, however small the rotated result is:
    # other accumulation orders are compared with an absolute floor of one ulp of the largest pre-RoPE magnitude)
    tol_v = dict(tol, abs_floor=float(2.0 ** (math.floor(math.log2(max(q.abs().max().item(), 1e-3))) - 7)))
    if not xnorm:
        for nt, waves in ((1, 4), (2, 8), (4, 8), (4, 16), (2, 2)):
            if (N // 16) % nt:
                continue
            q3, kc3, vc3 = torch.zeros_like(q_out), torch.zeros_like(kc), torch.zeros_like(vc)
            H.gemm_fused(wf, M, N, K, H.FEPI_QKV_ROPE, x_frag=dev(LY.rows_to_frag_ref(x)), nt=nt, waves=waves,
                         **dict(common, q_out=q3, k_cache=kc3, v_cache=vc3))
            assert_close_bf16(q3, q, what=f"fused q nt{nt} w{waves}", **tol_v)
            assert_close_bf16(LY.kv_hnd_to_nhd(kc3.cpu()), kref, what=f"fused k nt{nt} w{waves}", **tol_v)
            assert_close_bf16(LY.kv_hnd_to_nhd(vc3.cpu()), vref, what=f"fused v nt{nt} w{waves}", **tol_v)
    # the unfused pair (generic GEMM on the same weights, then rope_store with qkv_perm=1) must agree too
    y = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm(dev(LY.rows_to_frag_ref(x)), wf, y, M, N, K, N, bias=bias_p)
    q2 = torch.zeros_like(q_out)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    H.rope_store_kv(y, dev(pos), dev(cache), dev(slots), q2, kc2, vc2, M, nh, nkv, hd, bs, qkv_perm=1)
    assert_close_bf16(q2, q, what="unfused perm q", **tol)
    assert_close_bf16(LY.kv_hnd_to_nhd(kc2.cpu()), kref, what="unfused perm k", **tol)
    assert torch.equal(vc2.cpu().view(torch.int16), vc.cpu().view(torch.int16))


@pytest.mark.parametrize("M,K,I", [(1, 2048, 1024), (7, 4096, 512), (16, 512, 256)])
def test_fused_norm_gateup_silu(H, M, K, I):
    torch.manual_seed(M + K)
    h, res = torch.randn(M, K).to(BF), torch.randn(M, K).to(BF)
    nw = (1 + 0.1 * torch.randn(K)).to(BF)
    w = (torch.randn(2 * I, K) * 0.06).to(BF)
    x, res_ref = O.rmsnorm(h, nw, 1e-6, res)
    ref = O.silu_mul(O.linear(x, w))
    wf = torch.zeros(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag(dev(w), wf, 2 * I, K, mode=1)
    act_f = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
    res_out = torch.zeros(M, K, dtype=BF, device="cuda")
    H.gemm_fused(wf, M, 2 * I, K, H.FEPI_SILU_FRAG, h_rows=dev(h), res_in=dev(res), res_out=res_out, norm_w=dev(nw), eps=1e-6,
                 y=act_f)
    assert torch.equal(res_out.cpu().view(torch.int16), res_ref.view(torch.int16))
    act = LY.frag_to_rows_ref(act_f.cpu(), M, I)
    assert_close_bf16(act, ref, max_ulp=1, max_frac=0.04, rel_floor=2 ** -7, what="fused norm+gate_up+silu")
    # plain rows epilogue with several decompositions
    y_ref = O.linear(x, w[:I])
    wf2 = dev(LY.rows_to_frag_ref(w[:I]))
    for nt, waves in [(1, 1), (1, 16), (2, 4), (4, 8)]:
        y = torch.zeros(M, I, dtype=BF, device="cuda")
        H.gemm_fused(wf2, M, I, K, H.FEPI_ROWS, h_rows=dev(h), res_in=dev(res), norm_w=dev(nw), eps=1e-6, y=y, ldy=I, nt=nt,
                     waves=waves)
        assert_close_bf16(y, y_ref, max_ulp=1, max_frac=0.04, rel_floor=2 ** -7, what=f"fused rows {nt},{waves}")


@pytest.mark.parametrize("M,nh,nkv,hd,K", [(24, 32, 8, 64, 2048), (17, 8, 2, 128, 1024), (32, 16, 8, 128, 1024), (24, 4, 4, 64, 96)])
def test_fused_qkv_rope_two_token_tiles(H, M, nh, nkv, hd, K):
    """17..32 token rows (the 24-branch tree-decode step): QKV GEMM + RoPE + paged KV store in one launch over two
    16-row token tiles (csrc/gemm_fused.hip gemm_qkv_rope_m32_kernel) vs the oracle chain linear -> RoPE -> store."""
    torch.manual_seed(M + K)
    bs, nb = 16, 8
    N = (nh + 2 * nkv) * hd
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    bias = (torch.randn(N) * 0.1).to(BF)
    pos = torch.randint(0, 300, (M,), dtype=torch.int64)
    slots = torch.randperm(nb * bs)[:M].to(torch.int32)
    slots[5] = -1
    cache = O.make_cos_sin_cache(hd, 512, 5e5)
    qkv = O.linear(x, w, bias)
    q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    q, k = O.rope(pos, q.contiguous(), k.contiguous(), cache, hd)
    kref = torch.zeros(nb, bs, nkv, hd, dtype=BF)
    vref = torch.zeros_like(kref)
    O.store_kv(k.view(M, nkv, hd), v.contiguous().view(M, nkv, hd), kref, vref, slots)
    wf = torch.zeros(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag_qkv(dev(w), wf, nh, nkv, hd, K)
    bias_p = dev(bias[qkv_perm(nh, nkv, hd)])
    tol = dict(max_ulp=1, max_frac=0.04, rel_floor=2 ** -7,
               abs_floor=float(2.0 ** (math.floor(math.log2(max(q.abs().max().item(), 1e-3))) - 7)))
    for nt, waves in ((0, 0), (1, 4), (2, 8), (1, 16)):
        if nt and (N // 16) % nt:
            continue
        q_out = torch.zeros(M, nh * hd, dtype=BF, device="cuda")
        kc = torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        H.gemm_fused(wf, M, N, K, H.FEPI_QKV_ROPE, x_frag=dev(LY.rows_to_frag_ref(x)), bias=bias_p, positions=dev(pos), cos_sin=dev(cache),
                     slots=dev(slots), q_out=q_out, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=bs, nt=nt, waves=waves)
        assert_close_bf16(q_out, q, what=f"m32 q nt{nt} w{waves}", **tol)
        assert_close_bf16(LY.kv_hnd_to_nhd(kc.cpu()), kref, what=f"m32 k nt{nt} w{waves}", **tol)
        assert_close_bf16(LY.kv_hnd_to_nhd(vc.cpu()), vref, what=f"m32 v nt{nt} w{waves}", max_ulp=1, max_frac=0.04, rel_floor=2 ** -7)
