"""The oracle's attention restatement (oracle/ops.py attn_prefill_varlen / attn_paged / attn_tree + store_kv) against an
INDEPENDENT implementation: torch.nn.functional.scaled_dot_product_attention in float64 over keys gathered token by token
from the paged cache, with masks written from the position arithmetic (query i of Lq attends keys 0 .. Lk-Lq+i).  The
sgl-kernel / flashinfer wheels the reference calls are absent, so this is the strongest pin available for that part of the
oracle: two separately written implementations of the published algorithm agree to the last bf16 ulp."""
import random

import torch
import torch.nn.functional as F

from oracle import ops as O

BF = torch.bfloat16


def ulp_diff(a, b):
    return (a.contiguous().view(torch.int16).int() - b.contiguous().view(torch.int16).int()).abs()


def sdpa64(q, keys, vals, mask, scale):
    """q [Lq, nh, hd], keys/vals [Lk, nkv, hd] (bf16), mask bool [Lq, Lk] -> bf16 [Lq, nh, hd], all math in float64."""
    g = q.shape[1] // keys.shape[1]
    q64 = q.double().permute(1, 0, 2)
    k64 = keys.double().permute(1, 0, 2).repeat_interleave(g, dim=0)
    v64 = vals.double().permute(1, 0, 2).repeat_interleave(g, dim=0)
    o = F.scaled_dot_product_attention(q64, k64, v64, attn_mask=mask.unsqueeze(0), scale=scale)
    return o.permute(1, 0, 2).to(BF)


def paged_setup(B, nkv, hd, bs, max_ctx, seed):
    g = torch.Generator().manual_seed(seed)
    nblocks = B * (-(-max_ctx // bs)) + 3
    kc = torch.zeros(nblocks, bs, nkv, hd, dtype=BF)
    vc = torch.zeros_like(kc)
    perm = torch.randperm(nblocks, generator=g).tolist()
    tables, per = [], -(-max_ctx // bs)
    for b in range(B):
        tables.append(perm[b * per:(b + 1) * per])
    return kc, vc, tables, g


def fill(kc, vc, tables, ctx_lens, g, bs):
    """Write every position of every sequence through store_kv (the restated Triton store) and keep the plain copies."""
    keys, vals = [], []
    for tb, L in zip(tables, ctx_lens):
        k = torch.randn(L, kc.shape[2], kc.shape[3], generator=g).to(BF)
        v = torch.randn(L, kc.shape[2], kc.shape[3], generator=g).to(BF)
        slots = torch.tensor([tb[p // bs] * bs + p % bs for p in range(L)], dtype=torch.int32)
        O.store_kv(k, v, kc, vc, slots)
        keys.append(k)
        vals.append(v)
    return keys, vals


def test_paged_decode_verify_and_variable_glue():
    random.seed(0)
    for trial in range(6):
        B, nh, nkv, hd, bs = random.choice([1, 2, 3]), 4, 2, 32, random.choice([16, 32])
        ctx = [random.randint(20, 90) for _ in range(B)]
        kc, vc, tables, g = paged_setup(B, nkv, hd, bs, max(ctx), trial)
        keys, vals = fill(kc, vc, tables, ctx, g, bs)
        bt = torch.full((B, max(len(t) for t in tables)), -1, dtype=torch.int32)
        for b, t in enumerate(tables):
            bt[b, :len(t)] = torch.tensor(t)
        scale = hd ** -0.5
        # one query per sequence
        q = torch.randn(B, nh, hd, generator=g).to(BF)
        got = O.attn_paged(q, kc, vc, torch.tensor(ctx, dtype=torch.int32), bt, scale)
        for b in range(B):
            want = sdpa64(q[b:b + 1], keys[b], vals[b], torch.ones(1, ctx[b], dtype=torch.bool), scale)
            assert int(ulp_diff(got[b:b + 1], want).max()) <= 1
        # variable query counts per sequence (verify: K+1 each; EAGLE glue: extend rows make them differ), bottom-right aligned
        lq = [random.randint(1, 9) for _ in range(B)]
        cu = torch.tensor([0] + [sum(lq[:i + 1]) for i in range(B)], dtype=torch.int32)
        q = torch.randn(sum(lq), nh, hd, generator=g).to(BF)
        got = O.attn_paged(q, kc, vc, torch.tensor(ctx, dtype=torch.int32), bt, scale, cu_q=cu)
        for b in range(B):
            Lq, Lk = lq[b], ctx[b]
            mask = torch.tensor([[kpos <= Lk - Lq + i for kpos in range(Lk)] for i in range(Lq)])
            want = sdpa64(q[cu[b]:cu[b + 1]], keys[b], vals[b], mask, scale)
            d = ulp_diff(got[cu[b]:cu[b + 1]], want)
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.02


def test_varlen_prefill_and_store_skip():
    g = torch.Generator().manual_seed(3)
    nh, nkv, hd = 4, 2, 32
    lens = [5, 17, 1]
    cu = torch.tensor([0, 5, 22, 23], dtype=torch.int32)
    q = torch.randn(23, nh, hd, generator=g).to(BF)
    k = torch.randn(23, nkv, hd, generator=g).to(BF)
    v = torch.randn(23, nkv, hd, generator=g).to(BF)
    got = O.attn_prefill_varlen(q, k, v, cu, cu, hd ** -0.5)
    for b, L in enumerate(lens):
        s = slice(int(cu[b]), int(cu[b + 1]))
        mask = torch.tensor([[j <= i for j in range(L)] for i in range(L)])
        assert int(ulp_diff(got[s], sdpa64(q[s], k[s], v[s], mask, hd ** -0.5)).max()) <= 1
    # slot -1 stores nothing (reference store_kvcache semantics, ssd/layers/attention.py:23-25)
    kc = torch.zeros(4, 16, nkv, hd, dtype=BF)
    vc = torch.zeros_like(kc)
    slots = torch.tensor([3, -1, 40], dtype=torch.int32)
    O.store_kv(k[:3], v[:3], kc, vc, slots)
    flat = kc.view(-1, nkv, hd)
    assert torch.equal(flat[3], k[0]) and torch.equal(flat[40], k[2]) and int((flat != 0).any(-1).any(-1).sum()) == 2


def test_tree_attention_rows_see_prefix_glue_and_their_own_branch():
    """attn_tree with the structural mask == SDPA over an explicitly enumerated key set per branch: the trunk prefix, glue
    columns 0..j_i, and the branch's own column of every tree step so far (mask pinned to the reference's
    get_custom_mask in test_oracle_golden.py::test_tree_mask)."""
    g = torch.Generator().manual_seed(9)
    nh, nkv, hd, bs, K, F_ = 4, 2, 32, 16, 3, 2
    mq = F_ * (K + 1)
    jidx = [j for j in range(K + 1) for _ in range(F_)]
    for step in range(K):
        prefix = 37
        L = prefix + (K + 1) + (step + 1) * mq
        kc, vc, tables, _ = paged_setup(1, nkv, hd, bs, L, 11 + step)
        keys, vals = fill(kc, vc, tables, [L], g, bs)
        bt = torch.tensor([tables[0]], dtype=torch.int32)
        q = torch.randn(mq, nh, hd, generator=g).to(BF)
        got = O.attn_tree(q, kc, vc, torch.tensor([L], dtype=torch.int32), bt, hd ** -0.5, step, K, [jidx])
        for i in range(mq):
            visible = list(range(prefix)) + [prefix + c for c in range(jidx[i] + 1)] + [prefix + K + 1 + d * mq + i for d in range(step + 1)]
            idx = torch.tensor(visible)
            want = sdpa64(q[i:i + 1], keys[0][idx], vals[0][idx], torch.ones(1, len(visible), dtype=torch.bool), hd ** -0.5)
            assert int(ulp_diff(got[i:i + 1], want).max()) <= 1
