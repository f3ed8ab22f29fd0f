"""Attention + o_proj in one launch (csrc/attention.hip OPROJ variant, ssd_attn_oproj_parts) against the two launches it
replaces: ssd_attn_paged (the same attention code: bit-identical bf16 output) followed by ssd_gemm_parts with one K split
per kv head -- the same K partition, so every slab must agree up to the fp32 summation order inside a slab -- and against
the fp32 oracle product (reference ssd/layers/attention.py:105-131 + ssd/layers/linear.py:186-199)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import layout as LY
from tests.test_hip_ops import make_paged, dev, BF


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


@pytest.mark.parametrize("nh,nkv,hd,N,bs", [(32, 8, 64, 2048, 256), (32, 8, 64, 2048, 16), (16, 8, 128, 1024, 256), (8, 2, 64, 512, 32)])
@pytest.mark.parametrize("T,ctx", [(1, 37), (1, 700), (8, 150), (8, 1000), (3, 260)])
def test_fused_attention_oproj_slabs(H, nh, nkv, hd, N, bs, T, ctx):
    G = nh // nkv
    if T * G > (32 if hd == 64 else 16):
        pytest.skip("outside the fused kernel's row budget")
    kc, vc, bt, mb = make_paged(1, [ctx], nkv, hd, bs, seed=ctx + T)
    torch.manual_seed(T * 7 + nh)
    q = torch.randn(T, nh, hd).to(BF)
    w = (torch.randn(N, nh * hd) * 0.05).to(BF)
    K = nh * hd
    ctx_t = torch.tensor([ctx], dtype=torch.int32)
    qd, kd, vd, btd, cd = dev(q.view(T, -1)), dev(LY.kv_nhd_to_hnd(kc)), dev(LY.kv_nhd_to_hnd(vc)), dev(bt), dev(ctx_t)
    wf = dev(LY.rows_to_frag_ref(w))
    # unfused: attention -> fragment-major activations -> split-K slabs, one split per kv head
    rows = torch.zeros(T, K, dtype=BF, device="cuda")
    xf = torch.zeros(H.frag_numel(T, K), dtype=BF, device="cuda")
    H.attn_paged(qd, kd, vd, btd, mb, cd, 1, T, T, nh, nkv, hd, bs, hd ** -0.5, q_per_seq=T, out_rows=rows, out_frag=xf, waves=8)
    ref_parts = torch.zeros(nkv, T, N, dtype=torch.float32, device="cuda")
    H.gemm_parts(xf, wf, T, N, K, parts=ref_parts, splits=nkv, waves=2 if (K // 32 // nkv) >= 2 else 1)
    got = torch.full((nkv, T, N), float("nan"), dtype=torch.float32, device="cuda")
    H.attn_oproj_parts(qd, kd, vd, btd, mb, cd, T, nh, nkv, hd, bs, hd ** -0.5, wf, N, got)
    torch.cuda.synchronize()
    got, ref_parts = got.cpu(), ref_parts.cpu()
    assert torch.isfinite(got).all()
    scale = ref_parts.abs().max().item()
    assert (got - ref_parts).abs().max().item() <= 2e-5 * max(1.0, scale) + 1e-5, "slabs differ by more than an fp32 reordering"
    # the consumer's view: bf16(sum of the slabs) vs the fp32 product of the (bit-identical) bf16 attention output
    exact = rows.cpu().float() @ w.float().t()
    y = got.sum(0).to(BF).float()
    ulp = torch.exp2(torch.floor(torch.log2(exact.abs().clamp_min(1e-20))) - 7)
    assert bool(((y - exact).abs() <= ulp + 1e-6).all())
