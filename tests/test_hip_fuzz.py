"""Randomised shape sweeps of the two kernels whose decomposition is data-dependent: paged attention (KV block size,
context length around page / tile boundaries, waves per workgroup, grid key-splits) and the skinny GEMM (row groups per
workgroup, K-split waves, consecutive tiles per workgroup incl. a ragged last workgroup).  Oracle = CPU restatement."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O
from tests.test_hip_ops import ATTN_TOL, BF, H, dev, make_paged, run_attn, to_frag_dev  # noqa: F401  (H is the fixture)
from tests.util import assert_close_bf16


def test_attention_block_sizes_and_boundaries(H):
    rnd = random.Random(1234)
    cases = []
    for bs in (16, 32, 64, 256):
        edge = [1, 15, 16, 17, 31, 32, 33, bs - 1, bs, bs + 1, 2 * bs - 1, 2 * bs, 2 * bs + 17, 255, 256, 257, 1023, 1025]
        for _ in range(7):
            B = rnd.choice((1, 2, 3))
            qps = rnd.choice((1, 1, 4, 7, 8, 24))
            ctx = [max(qps, rnd.choice(edge + [rnd.randint(1, 1400)])) for _ in range(B)]
            cases.append((bs, B, qps, ctx, rnd.choice((1, 1, 2, 5)), rnd.choice((1, 2, 4, 8))))
    for nh, nkv, hd in ((32, 8, 64), (16, 2, 128), (40, 8, 128)):
        for bs, B, qps, ctx_lens, splits, waves in cases[:: 2 if hd == 128 else 1]:
            kc, vc, bt, mb = make_paged(B, ctx_lens, nkv, hd, bs, seed=bs + qps + sum(ctx_lens))
            torch.manual_seed(sum(ctx_lens))
            q = torch.randn(B * qps, nh, hd).to(BF)
            ctx = torch.tensor(ctx_lens, dtype=torch.int32)
            cu = torch.arange(B + 1, dtype=torch.int32) * qps
            ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(B * qps, nh * hd)
            got = run_attn(H, q.view(B * qps, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=qps, splits=splits, waves=waves)
            assert_close_bf16(got, ref, what=f"attn bs{bs} B{B} q{qps} ctx{ctx_lens} s{splits} w{waves} heads{nh}/{nkv}x{hd}", **ATTN_TOL)


def test_gemm_decompositions(H):
    rnd = random.Random(99)
    for _ in range(40):
        M = rnd.choice((1, 3, 7, 16, 24))
        N = 16 * rnd.choice((4, 12, 20, 36, 60, 128))
        K = 32 * rnd.choice((2, 5, 8, 33, 64))
        nt = rnd.choice([n for n in (1, 2, 4) if (N // 16) % n == 0 and not (n == 4 and M > 32)])
        waves = rnd.choice((1, 2, 4, 8, 16))
        tpw = rnd.choice((1, 2, 3, 5, 8))
        torch.manual_seed(M + N + K)
        x = torch.randn(M, K).to(BF)
        w = (torch.randn(N, K) * 0.05).to(BF)
        b = torch.randn(N).to(BF)
        y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        H.gemm(to_frag_dev(x), to_frag_dev(w), y, M, N, K, N, bias=dev(b), cfg=(nt, waves | (tpw << 8)))
        assert_close_bf16(y, O.linear(x, w, b), max_ulp=1, max_frac=0.03, rel_floor=2 ** -7,
                          what=f"gemm M{M} N{N} K{K} nt{nt} w{waves} tpw{tpw}")
