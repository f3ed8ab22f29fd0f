"""ssd_attn_paged_qkv (csrc/attention.hip, QKV variant): per-head q / k RMSNorm + RoPE + paged KV store + attention in ONE launch
must equal ssd_rope_store_kv followed by ssd_attn_paged BIT FOR BIT -- attention output rows, fragment-major copy, and the
K / V cache contents -- for single-token decode, K+1-row verify (batched), and the 24-branch tree step, head dims 64 / 128, with
and without the Qwen3 norms, natural and rotation-paired QKV column order.  (Reference: ssd/models/qwen3.py:96-104,
ssd/layers/rotary_embedding.py:40-60, ssd/layers/attention.py:10-41,105-131.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def _case(H, nh, nkv, hd, bs, B, qps, ctx_lens, norms, perm, mode, tree=None, waves=8, seed=0):
    from ssd_amd.model import make_cos_sin
    g = torch.Generator().manual_seed(seed)
    T = B * qps
    max_blocks = max((L + bs - 1) // bs for L in ctx_lens) + 1
    nblocks = B * max_blocks + 2
    pm = torch.randperm(nblocks, generator=g)
    bt = torch.full((B, max_blocks), -1, dtype=torch.int32)
    p0 = 0
    for b, L in enumerate(ctx_lens):
        n = (L + bs - 1) // bs
        bt[b, :n] = pm[p0:p0 + n].to(torch.int32)
        p0 += n
    kc0 = torch.randn(nblocks, nkv, bs, hd, generator=g).to(BF).cuda()          # HND layout, random "old" content
    vc0 = torch.randn(nblocks, nkv, bs, hd, generator=g).to(BF).cuda()
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, generator=g).to(BF).cuda()
    cos_sin = make_cos_sin(hd, 4096, 5e5, torch.device("cuda"))
    pos, slots = [], []
    for b, L in enumerate(ctx_lens):
        for i in range(qps):
            kidx = L - qps + i                       # the new tokens are the last qps keys of the sequence
            pos.append(kidx if mode == H.MODE_CAUSAL else 100 + (i % 7) + (tree["step"] if tree else 0))
            slots.append(int(bt[b, kidx // bs]) * bs + kidx % bs)
    pos = torch.tensor(pos, dtype=torch.int64).cuda()
    slots = torch.tensor(slots, dtype=torch.int32).cuda()
    ctx = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    qn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF).cuda() if norms else None
    kn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF).cuda() if norms else None
    tk = dict(mode=mode)
    if tree:
        tk.update(tree_K=tree["K"], tree_mq=qps, tree_step=tree["step"], tree_F=tree["F"])
    bt_d = bt.cuda()
    # path A: two launches
    kcA, vcA = kc0.clone(), vc0.clone()
    q_out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
    H.rope_store_kv(qkv, pos, cos_sin, slots, q_out, kcA, vcA, T, nh, nkv, hd, bs, q_norm_w=qn, k_norm_w=kn, eps=1e-6, qkv_perm=perm)
    outA = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
    fragA = torch.zeros(H.frag_numel(T, nh * hd), dtype=BF, device="cuda")
    H.attn_paged(q_out, kcA, vcA, bt_d, max_blocks, ctx, B, T, qps, nh, nkv, hd, bs, hd ** -0.5, q_per_seq=qps, splits=1,
                 out_rows=outA, out_frag=fragA, waves=waves, **tk)
    # path B: one launch
    kcB, vcB = kc0.clone(), vc0.clone()
    outB = torch.zeros_like(outA)
    fragB = torch.zeros_like(fragA)
    H.attn_paged_qkv(qkv, pos, cos_sin, slots, kcB, vcB, bt_d, max_blocks, ctx, B, T, qps, nh, nkv, hd, bs, hd ** -0.5,
                     q_norm_w=qn, k_norm_w=kn, eps=1e-6, qkv_perm=perm, out_rows=outB, out_frag=fragB, waves=waves, **tk)
    torch.cuda.synchronize()
    i16 = lambda t: t.view(torch.int16)
    assert torch.isfinite(outA.float()).all()
    assert torch.equal(i16(kcA), i16(kcB)) and torch.equal(i16(vcA), i16(vcB)), "K / V cache contents differ"
    assert not torch.equal(i16(kcA), i16(kc0)), "the store did not happen"
    assert torch.equal(i16(outA), i16(outB)), f"attention rows differ: max |d| {(outA.float() - outB.float()).abs().max().item()}"
    assert torch.equal(i16(fragA), i16(fragB))


@pytest.mark.parametrize("norms", [True, False])
@pytest.mark.parametrize("perm", [0, 1])
@pytest.mark.parametrize("nh,nkv,hd", [(16, 8, 128), (64, 8, 128), (32, 8, 64), (8, 1, 128)])
def test_decode_and_verify_rows(H, nh, nkv, hd, perm, norms):
    for (B, qps, ctx_lens) in [(1, 1, [37]), (3, 1, [1, 64, 333]), (1, 8, [135]), (2, 8, [640, 77]), (1, 7, [300])]:
        _case(H, nh, nkv, hd, 16, B, qps, ctx_lens, norms, perm, H.MODE_CAUSAL, seed=B * 10 + qps)
    _case(H, nh, nkv, hd, 256, 1, 8, [300], norms, perm, H.MODE_CAUSAL, waves=4, seed=5)


@pytest.mark.parametrize("norms", [True, False])
@pytest.mark.parametrize("nh,nkv,hd", [(16, 8, 128), (32, 8, 64)])
def test_tree_step(H, nh, nkv, hd, norms):
    K, F = 7, 3
    MQ = F * (K + 1)
    for step in (0, 3, 6):
        ctx_lens = [p + K + 1 + (step + 1) * MQ for p in (150, 41)]
        _case(H, nh, nkv, hd, 16, 2, MQ, ctx_lens, norms, 1, H.MODE_TREE, tree=dict(K=K, F=F, step=step), seed=step)
        _case(H, nh, nkv, hd, 256, 1, MQ, ctx_lens[:1], norms, 1, H.MODE_TREE, tree=dict(K=K, F=F, step=step), seed=step + 50)
