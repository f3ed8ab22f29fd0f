"""LM head with per-workgroup argmax candidates (csrc/gemm.hip EPI_ROWS_ARGMAX) and the tail kernels that finish the
argmax from them (csrc/sample.hip ssd_argmax_parts / _verify / _advance) against the unfused launches they replace:
identical logits rows, identical tokens (incl. ties -> lowest index, reference torch argmax semantics,
ssd/layers/sampler.py:15-20), identical accept / reject and chain-advance state."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import layout as LY

BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def dev(t):
    return t.cuda().contiguous()


def head_inputs(M, N, K, seed, ties=True):
    torch.manual_seed(seed)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    if ties:                       # equal logits in different workgroups and inside one tile: the lowest index must win
        w[N - 3] = w[5]
        w[17] = w[5]
        w[N // 2 + 1] = w[5]
        x[0] = (w[5].float() * 4).to(BF)      # makes the tied rows the row-0 maximum
    return x, w


def run_head(H, x, w, M, N, K):
    xf, wf = dev(LY.rows_to_frag_ref(x)), dev(LY.rows_to_frag_ref(w))
    y0 = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm(xf, wf, y0, M, N, K, N)
    nparts = H.gemm_argmax_nparts(M, N, K)
    stride = nparts + 3
    pv = torch.full((M, stride), float("nan"), dtype=torch.float32, device="cuda")
    pi = torch.full((M, stride), -7, dtype=torch.int32, device="cuda")
    y1 = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm_argmax(xf, wf, y1, M, N, K, N, pv, pi, stride)
    return y0, y1, pv, pi, nparts, stride


@pytest.mark.parametrize("M,N,K", [(1, 512, 128), (1, 128256, 2048), (7, 4096, 256), (8, 32064, 1024), (8, 16032, 8192), (16, 2048, 512),
                                   (17, 4096, 256), (24, 128256, 2048), (32, 1024, 64)])
def test_head_rows_and_argmax_from_candidates(H, M, N, K):
    x, w = head_inputs(M, N, K, seed=M + N + K)
    y0, y1, pv, pi, nparts, stride = run_head(H, x, w, M, N, K)
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)), "the logits rows must not depend on the epilogue"
    want = y1.float().cpu().argmax(-1)          # torch CPU argmax: first maximal index
    vmax = y1.float().cpu().max(-1).values
    out = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    out2 = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    table = torch.full((M, 5), -1, dtype=torch.int64, device="cuda")
    val = torch.zeros(M, dtype=torch.float32, device="cuda")
    H.argmax_parts(pv, pi, nparts, stride, M, out, out2, table.view(-1)[3:], 5, out_val=val)
    torch.cuda.synchronize()
    assert out.cpu().tolist() == want.tolist()
    assert out2.cpu().tolist() == want.tolist() and table[:, 3].cpu().tolist() == want.tolist()
    assert bool((table[:, :3] == -1).all()) and bool((table[:, 4] == -1).all())
    assert torch.equal(val.cpu(), vmax)
    if N >= 4096:
        assert int(want[0]) == 5, "the tie construction should put the row-0 maximum on the duplicated rows"
    # candidates beyond nparts are never touched
    assert bool(torch.isnan(pv[:, nparts:]).all()) and bool((pi[:, nparts:] == -7).all())
    # global indices of a vocabulary shard
    off = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    H.argmax_parts(pv, pi, nparts, stride, M, off, idx_offset=1000)
    assert off.cpu().tolist() == (want + 1000).tolist()


@pytest.mark.parametrize("B,K,N,Kd", [(1, 6, 4096, 256), (3, 7, 32064, 512), (2, 1, 512, 128), (1, 15, 2048, 128)])
def test_verify_tail_equals_argmax_then_verify_greedy(H, B, K, N, Kd):
    M = B * (K + 1)
    x, w = head_inputs(M, N, Kd, seed=B * 100 + K, ties=False)
    y0, y1, pv, pi, nparts, stride = run_head(H, x, w, M, N, Kd)
    preds_ref = torch.zeros(M, dtype=torch.int64, device="cuda")
    H.argmax_rows(y0, N, M, N, preds_ref)
    pr = preds_ref.cpu().view(B, K + 1)
    # speculations that accept a different number of tokens per sequence
    spec = torch.zeros(B, K + 1, dtype=torch.int64)
    for b in range(B):
        n_ok = (b * 3 + 1) % (K + 1)
        spec[b, 1:] = pr[b, :K]
        if n_ok < K:
            spec[b, 1 + n_ok] = (pr[b, n_ok] + 1) % N
        spec[b, 0] = 7 + b
    sd = dev(spec)
    acc0, rec0 = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int64, device="cuda")
    pk0 = torch.zeros(B, K + 3, dtype=torch.int64, device="cuda")
    H.verify_greedy(preds_ref, sd, B, K, acc0, rec0, pk0)
    acc1, rec1 = torch.full((B,), -1, dtype=torch.int32, device="cuda"), torch.full((B,), -1, dtype=torch.int64, device="cuda")
    pk1 = torch.full((B, K + 3), -1, dtype=torch.int64, device="cuda")
    preds = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    H.argmax_parts_verify(pv, pi, nparts, stride, sd, B, K, acc1, rec1, pk1, preds=preds)
    torch.cuda.synchronize()
    assert torch.equal(preds.cpu(), preds_ref.cpu())
    assert torch.equal(acc0.cpu(), acc1.cpu()) and torch.equal(rec0.cpu(), rec1.cpu()) and torch.equal(pk0.cpu(), pk1.cpu())
    assert acc1.cpu().tolist() == [(b * 3 + 1) % (K + 1) for b in range(B)]


@pytest.mark.parametrize("B", [1, 3, 20])
def test_advance_tail_equals_argmax_then_draft_advance(H, B):
    K, bs, mb, N, Kd = 4, 16, 8, 4096, 256
    x, w = head_inputs(B, N, Kd, seed=B, ties=False)
    y0, y1, pv, pi, nparts, stride = run_head(H, x, w, B, N, Kd)
    nxt0 = torch.zeros(B, dtype=torch.int64, device="cuda")
    H.argmax_rows(y0, N, B, N, nxt0)

    def state():
        bt = torch.arange(B * mb, dtype=torch.int32).view(B, mb)
        bt[B - 1, 3:] = -1
        pos = torch.tensor([15 + 9 * b for b in range(B)], dtype=torch.int64) % (3 * bs)
        return [dev(t) for t in (torch.zeros(B, dtype=torch.int64), pos, torch.zeros(B, dtype=torch.int32), (pos + 1).to(torch.int32), bt,
                                 torch.zeros(B, K + 1, dtype=torch.int64), torch.tensor([1], dtype=torch.int32))]
    a, b_ = state(), state()
    H.draft_advance(nxt0, a[0], a[1], a[2], a[3], a[4], mb, bs, a[5], K, a[6], B)
    nxt1 = torch.full((B,), -1, dtype=torch.int64, device="cuda")
    H.argmax_parts_advance(pv, pi, nparts, stride, nxt1, b_[0], b_[1], b_[2], b_[3], b_[4], mb, bs, b_[5], K, b_[6], B)
    torch.cuda.synchronize()
    assert torch.equal(nxt0.cpu(), nxt1.cpu())
    for u, v in zip(a, b_):
        assert torch.equal(u.cpu(), v.cpu())
    assert int(b_[6].cpu()) == 2
