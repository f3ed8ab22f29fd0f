"""GPU parity of the temperature > 0 kernels (csrc/stochastic.hip).  The RNG differs from torch's, so parity is
(a) exact on every deterministic quantity (greedy rows, log-sum-exp, acceptance probabilities min(1, p/q)) and
(b) distributional on the random draws: chi-square goodness of fit of many independent draws against the
distribution the oracle (the reference's verify / Sampler restated) defines, at p > 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O

BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def dev(t):
    return t.cuda().contiguous()


def chi2_ok(counts: torch.Tensor, probs: torch.Tensor, what: str):
    from scipy.stats import chisquare
    n = counts.sum().item()
    keep = probs * n >= 5
    obs = torch.cat([counts[keep].double(), counts[~keep].double().sum().view(1)])
    exp = torch.cat([(probs[keep] * n).double(), (probs[~keep] * n).double().sum().view(1)])
    if exp[-1] < 1e-9:
        obs, exp = obs[:-1], exp[:-1]
    exp = exp * obs.sum() / exp.sum()
    stat, p = chisquare(obs.numpy(), exp.numpy())
    print(f"{what}: chi2={stat:.1f} p={p:.4f} bins={len(obs)} n={n}")
    assert p > 1e-4, f"{what}: chi-square p={p}"


def rng(seed=1):
    return torch.tensor([seed], dtype=torch.int64, device="cuda")


def test_sample_rows_distribution_and_greedy(H):
    torch.manual_seed(0)
    V, N = 96, 40000
    logits = (torch.randn(V) * 1.5).to(BF)
    T = 0.8
    rows = logits.unsqueeze(0).repeat(N, 1).contiguous()
    temps = torch.full((N,), T)
    out = torch.zeros(N, dtype=torch.int64, device="cuda")
    st = rng(3)
    H.sample_rows(dev(rows), V, N, V, dev(temps), 1, st, 11, out)
    probs = torch.softmax(logits.float() / T, dim=-1)
    chi2_ok(torch.bincount(out.cpu(), minlength=V), probs, "sample_rows T=0.8")
    # same state, same salt -> identical draws; advanced state -> different draws
    out2 = torch.zeros_like(out)
    H.sample_rows(dev(rows), V, N, V, dev(temps), 1, st, 11, out2)
    assert torch.equal(out, out2)
    H.rng_advance(st)
    H.sample_rows(dev(rows), V, N, V, dev(temps), 1, st, 11, out2)
    assert (out != out2).float().mean().item() > 0.5
    # temperature 0 rows are the argmax (lowest index on ties), mixed in one launch; one temperature per 3 rows
    x = torch.randn(6, 128256).to(BF)
    x[0, 5] = 20.0
    x[0, 9000] = 20.0
    t6 = torch.tensor([0.0, 1.0])
    o6 = torch.zeros(6, dtype=torch.int64, device="cuda")
    H.sample_rows(dev(x), 128256, 6, 128256, dev(t6), 3, st, 1, o6)
    assert o6.cpu()[:3].tolist() == O.argmax_rows(x[:3]).tolist()


def test_row_lse(H):
    torch.manual_seed(1)
    x = (torch.randn(9, 5000) * 3).to(BF)
    temps = torch.tensor([0.5, 1.0, 2.0])
    lse = torch.zeros(9, device="cuda")
    H.row_lse(dev(x), 5000, 9, 5000, dev(temps), 3, lse)
    want = torch.logsumexp(x.float() / temps.repeat_interleave(3).unsqueeze(1), dim=-1)
    assert (lse.cpu() - want).abs().max().item() < 2e-3


def run_verify(H, lp, lq, spec, tt, tq, ratio_rows, seed=5, salt=2):
    B, Kp1, V = lp.shape
    K = Kp1 - 1
    d_lp, d_lq = dev(lp.view(B * Kp1, V)), dev(lq.view(B * K, V))
    preds = torch.zeros(B * Kp1, dtype=torch.int64, device="cuda")
    H.argmax_rows(d_lp, V, B * Kp1, V, preds)
    lse_p = torch.zeros(B * Kp1, device="cuda")
    lse_q = torch.zeros(B * K, device="cuda")
    H.row_lse(d_lp, V, B * Kp1, V, dev(tt), Kp1, lse_p)
    H.row_lse(d_lq, V, B * K, V, dev(tq), K, lse_q)
    acc = torch.zeros(B, dtype=torch.int32, device="cuda")
    rec = torch.zeros(B, dtype=torch.int64, device="cuda")
    packed = torch.zeros(B, K + 3, dtype=torch.int64, device="cuda")
    ap = torch.zeros(B, K, device="cuda")
    H.verify_ratio(d_lp, V, d_lq, V, V, B, K, dev(spec), preds, lse_p, lse_q, dev(tt), dev(tq), dev(ratio_rows.to(torch.int32)),
                   rng(seed), salt, acc, rec, packed, ap)
    torch.cuda.synchronize()
    return acc.cpu(), rec.cpu(), packed.cpu(), ap.cpu()


def test_verify_ratio_deterministic_parts(H, golden):
    g = golden("stochastic_golden")
    lp, lq, spec, tt, tq, hits = g["lp"], g["lq"], g["spec"], g["tt"], g["tq"], g["hits"]
    B, Kp1, V = lp.shape
    V8 = 56                                    # argmax rows need 16-byte aligned rows: pad the vocabulary with -inf
    lp8 = torch.full((B, Kp1, V8), float("-inf"), dtype=BF)
    lq8 = torch.full((B, Kp1 - 1, V8), float("-inf"), dtype=BF)
    lp8[..., :V], lq8[..., :V] = lp, lq
    for jit in (False, True):
        ratio = torch.ones(B, dtype=torch.bool) if jit else hits.bool()
        acc, rec, packed, ap = run_verify(H, lp8, lq8, spec, tt, tq, ratio)
        _, _, want_ap = O.verify_full(lp, lq, spec, tt, tq, cache_hits=hits, jit_speculate=jit)
        base = (tt > 0) | (tq > 0)
        rows = base & ratio
        assert (ap[rows] - want_ap[rows]).abs().max().item() < 2e-3          # min(1, p/q) per position
        # rows that are NOT ratio rows accept exactly like the greedy branch
        preds = lp.argmax(-1)
        g_acc, g_rec = O.verify_greedy(preds, spec)
        assert acc[~rows].tolist() == g_acc[~rows].tolist()
        # temperature-0 targets recover greedily at whatever position they stopped
        z = tt == 0
        assert all(int(rec[b]) == int(preds[b, int(acc[b])]) for b in range(B) if z[b])
        assert torch.equal(packed[:, 0], acc.long()) and torch.equal(packed[:, 1], rec) and torch.equal(packed[:, 2:], spec)


def test_verify_ratio_distributions(H):
    """Many identical sequences in one launch: accepted-length and recovery-token statistics against the exact
    distributions implied by verify() (speculative sampling): P(n) = prod_{i<n} a_i * (1 - a_n), recovery | n ~
    normalise(max(0, p_n - q_n)) for n < K and ~ p_K for n = K."""
    torch.manual_seed(2)
    V, K, N = 48, 3, 30000
    lp1 = (torch.randn(K + 1, V) * 1.2).to(BF)
    lq1 = (lp1[:K].float() + torch.randn(K, V) * 0.8).to(BF)
    Tt, Tq = 0.9, 1.1
    p = torch.softmax(lp1.float() / Tt, -1)
    q = torch.softmax(lq1.float() / Tq, -1)
    x = torch.tensor([int(q[i].argmax()) for i in range(K)])        # a plausible draft: the mode of q
    spec1 = torch.cat([torch.tensor([7]), x])
    a = torch.tensor([min(1.0, float(p[i, x[i]] / q[i, x[i]])) for i in range(K)])
    lp = lp1.unsqueeze(0).repeat(N, 1, 1).contiguous()
    lq = lq1.unsqueeze(0).repeat(N, 1, 1).contiguous()
    spec = spec1.unsqueeze(0).repeat(N, 1).contiguous()
    tt, tq = torch.full((N,), Tt), torch.full((N,), Tq)
    acc, rec, _, ap = run_verify(H, lp, lq, spec, tt, tq, torch.ones(N, dtype=torch.bool))
    assert (ap[0] - a).abs().max().item() < 2e-3
    pn = torch.zeros(K + 1)
    run = 1.0
    for i in range(K):
        pn[i] = run * (1 - a[i])
        run *= a[i]
    pn[K] = run
    chi2_ok(torch.bincount(acc.long(), minlength=K + 1), pn, "accepted length")
    for n in range(K + 1):
        sel = acc == n
        if sel.sum() < 500:
            continue
        if n < K:
            r = (p[n] - q[n]).clamp(min=0)
            r = r / r.sum()
        else:
            r = p[K]
        chi2_ok(torch.bincount(rec[sel], minlength=V), r, f"recovery | n={n}")
    # non-ratio rows (cache miss without JIT): greedy acceptance, recovery ~ p at the stopping position
    acc2, rec2, _, _ = run_verify(H, lp, lq, spec, tt, tq, torch.zeros(N, dtype=torch.bool), seed=9)
    g_acc, _ = O.verify_greedy(lp1.argmax(-1).unsqueeze(0), spec1.unsqueeze(0))
    assert (acc2 == int(g_acc[0])).all()
    chi2_ok(torch.bincount(rec2, minlength=V), p[int(g_acc[0])], "recovery of a non-ratio row ~ p")


def test_sampler_x_boost_kernels(H, golden):
    """sampler_x: ssd_topk_rows picks the F+1 most probable tokens, ssd_sample_rows draws from the rescaled distribution
    (chi-square against the oracle's apply_sampler_x_rescaling), ssd_row_lse / ssd_verify_ratio use the rescaled q
    (acceptance probabilities against the oracle's verify(sampler_x=...))."""
    torch.manual_seed(3)
    V, N, F, X, T = 96, 40000, 3, 0.4, 0.8
    logits = (torch.randn(V) * 1.5).to(BF)
    rows = logits.unsqueeze(0).repeat(N, 1).contiguous()
    d_rows = dev(rows)
    top = torch.zeros(N, F + 1, dtype=torch.int32, device="cuda")
    H.topk_rows(d_rows, V, N, V, F + 1, top)
    want_top = torch.topk(logits.float(), F + 1).indices
    assert top[0].cpu().tolist() == want_top.tolist() and torch.equal(top[0], top[N - 1])
    out = torch.zeros(N, dtype=torch.int64, device="cuda")
    H.sample_rows(d_rows, V, N, V, dev(torch.full((N,), T)), 1, rng(4), 9, out, boost_idx=top, boost_k=F + 1, boost_x=X)
    probs = O.sampler_x_rescale(torch.softmax(logits.float() / T, -1).unsqueeze(0), X, F)[0]
    chi2_ok(torch.bincount(out.cpu(), minlength=V), probs, "sample_rows with sampler_x")
    # verify: accept probabilities with the rescaled q
    g = golden("stochastic_golden")
    lp, lq, spec, tt, tq, hits = g["lp"], g["lq"], g["spec"], g["tt"], g["tq"], g["hits"]
    B, Kp1, Vg = lp.shape
    K, V8 = Kp1 - 1, 56
    lp8 = torch.full((B, Kp1, V8), float("-inf"), dtype=BF)
    lq8 = torch.full((B, K, V8), float("-inf"), dtype=BF)
    lp8[..., :Vg], lq8[..., :Vg] = lp, lq
    d_lp, d_lq = dev(lp8.view(B * Kp1, V8)), dev(lq8.view(B * K, V8))
    preds = torch.zeros(B * Kp1, dtype=torch.int64, device="cuda")
    H.argmax_rows(d_lp, V8, B * Kp1, V8, preds)
    bq = torch.zeros(B * K, 4, dtype=torch.int32, device="cuda")
    H.topk_rows(d_lq, V8, B * K, V8, 4, bq)
    lse_p = torch.zeros(B * Kp1, device="cuda")
    lse_q = torch.zeros(B * K, device="cuda")
    H.row_lse(d_lp, V8, B * Kp1, V8, dev(tt), Kp1, lse_p)
    H.row_lse(d_lq, V8, B * K, V8, dev(tq), K, lse_q, boost_idx=bq, boost_k=4, boost_x=0.6)
    acc = torch.zeros(B, dtype=torch.int32, device="cuda")
    rec = torch.zeros(B, dtype=torch.int64, device="cuda")
    ap = torch.zeros(B, K, device="cuda")
    H.verify_ratio(d_lp, V8, d_lq, V8, V8, B, K, dev(spec), preds, lse_p, lse_q, dev(tt), dev(tq),
                   dev(torch.ones(B, dtype=torch.int32)), rng(5), 2, acc, rec, None, ap, boost_idx_q=bq, boost_k=4, boost_x=0.6)
    _, _, want_ap = O.verify_full(lp, lq, spec, tt, tq, cache_hits=hits, jit_speculate=True, sampler_x=0.6, async_fan_out=3)
    rows_ = (tt > 0) | (tq > 0)
    assert (ap.cpu()[rows_] - want_ap[rows_]).abs().max().item() < 2e-3
