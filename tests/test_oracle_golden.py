"""The CPU oracle against the golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  Pointwise ops and integer logic must be bit-exact; GEMM-backed results are
the same torch call as the reference so they are bit-exact on the generating machine and within one bf16
ulp elsewhere (different CPU GEMM kernels)."""
import torch

from oracle import ops as O
from oracle.model import OracleModel, Ctx
from ssd_amd.model_config import ModelConfig


def bits(t):
    return t.contiguous().view(torch.int16)


def assert_bits(a, b):
    assert a.shape == b.shape
    assert torch.equal(bits(a), bits(b)), f"{(bits(a) != bits(b)).sum().item()} of {a.numel()} elements differ"


def assert_ulp(a, b, max_ulp=1, max_frac=0.02):
    """bf16 tensors equal within max_ulp on at most max_frac of the elements (0 everywhere else)."""
    d = (bits(a).int() - bits(b).int()).abs()
    assert int(d.max()) <= max_ulp, f"max ulp diff {int(d.max())}"
    assert float((d > 0).float().mean()) <= max_frac, f"{float((d > 0).float().mean()):.4f} of elements differ"


def test_rmsnorm(golden):
    g = golden("ops_golden")
    assert_bits(O.rmsnorm(g["norm_x"], g["norm_w"], 1e-5), g["norm_y"])
    y, r = O.rmsnorm(g["norm_x"], g["norm_w"], 1e-5, g["norm_res"])
    assert_bits(y, g["addnorm_y"])
    assert_bits(r, g["addnorm_res"])
    assert_bits(O.rmsnorm(g["hnorm_x"], g["hnorm_w"], 1e-6), g["hnorm_y"])


def test_rope(golden):
    g = golden("ops_golden")
    cache = O.make_cos_sin_cache(64, 512, 500000.0)
    assert torch.equal(cache, g["rope_cache"])
    q, k = O.rope(g["rope_pos"], g["rope_q"], g["rope_k"], cache, 64)
    assert_bits(q, g["rope_qo"])
    assert_bits(k, g["rope_ko"])


def test_silu_mul(golden):
    g = golden("ops_golden")
    assert_bits(O.silu_mul(g["silu_x"]), g["silu_y"])


def test_linears_embedding_head(golden):
    g = golden("ops_golden")
    assert_ulp(O.linear(g["lin_x"], g["qkv_w"]), g["qkv_y"])
    assert_ulp(O.linear(g["lin_x"], g["gu_w"]), g["gu_y"])
    assert_ulp(O.silu_mul(O.linear(g["lin_x"], g["gu_w"])), g["gu_act"])
    assert_ulp(O.linear(g["dn_x"], g["dn_w"]), g["dn_y"])
    assert_bits(O.embedding(g["emb_ids"], g["emb_w"]), g["emb_y"])
    lg = O.linear(g["lin_x"], g["head_w"])
    assert_ulp(lg, g["head_logits"])
    assert torch.equal(O.argmax_rows(g["head_logits"]), g["sample_tokens"])


def test_verify_greedy(golden):
    g = golden("logic_golden")
    sfx, rec = O.verify_suffixes(g["v_logits_p"], g["v_spec"])
    assert [len(s) for s in sfx] == g["v_suffix_len"].tolist()
    for b, s in enumerate(sfx):
        assert s == g["v_suffix"][b, :len(s)].tolist()
    assert rec == g["v_rec"].tolist()
    # the cases were built to cover: all accepted, mid mismatch, first mismatch, last mismatch
    assert g["v_suffix_len"].tolist()[:4] == [7, 4, 1, 6]


def test_fork(golden):
    g = golden("logic_golden")
    lists = [g["f_list_hit"].tolist() if h else g["f_list_miss"].tolist() for h in g["f_hits"].tolist()]
    idx = O.fork_topf(g["f_logits"].view(2, 4, -1), g["f_returned"], lists)
    assert torch.equal(idx, g["f_idx"])


def test_tree_mask(golden):
    g = golden("logic_golden")
    K = 3
    for step in range(2):
        ctx = g[f"m_ctx{step}"].tolist()
        ref = g[f"m_mask{step}"].bool()
        off = 0
        for b, L in enumerate(ctx):
            fl = g["f_list_hit"].tolist() if g["f_hits"][b] else g["f_list_miss"].tolist()
            jidx = [j for j, f in enumerate(fl) for _ in range(f)]
            m = O.tree_mask(L, step, K, jidx)
            n = m.numel()
            assert torch.equal(m.view(-1), ref[off:off + n])
            off += n
        assert off == ref.numel()


def cfg_from(g, prefix=""):
    ci, cf = g[prefix + "cfg_i"].tolist(), g[prefix + "cfg_f"].tolist()
    return ci, cf


def make_oracle(g, family, wprefix="w.", cprefix="", tie=False, qk_norm=False):
    ci, cf = cfg_from(g, cprefix)
    cfg = ModelConfig(family, ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], cf[0], cf[1], ci[7], tie, qk_norm)
    w = {k[len(wprefix):]: v for k, v in g.items() if k.startswith(wprefix)}
    return OracleModel(cfg, w, num_blocks=24, block_size=16), cfg


def slots(table, positions, bs=16):
    return torch.tensor([table[p // bs] * bs + p % bs for p in positions], dtype=torch.int32)


def test_tiny_llama_forward(golden):
    g = golden("tiny_llama")
    m, cfg = make_oracle(g, "llama")
    bt = g["block_table"]
    table = bt[0].tolist()
    prompt = g["prompt"].tolist()
    P = len(prompt)
    cu = torch.tensor([0, P], dtype=torch.int32)
    h = m.forward(g["prompt"], torch.arange(P), Ctx("prefill", slot_mapping=slots(table, range(P)), cu_q=cu, cu_k=cu))
    assert_ulp(m.compute_logits(h), g["prefill_logits"], max_ulp=0, max_frac=0.0)
    # decode x2
    toks = g["decode_tokens"].tolist()
    for i, t in enumerate(toks):
        h = m.forward(torch.tensor([t]), torch.tensor([P + i]),
                      Ctx("decode", slot_mapping=slots(table, [P + i]), context_lens=torch.tensor([P + i + 1], dtype=torch.int32), block_tables=bt))
        assert_ulp(m.compute_logits(h), g["decode_logits"][i:i + 1], max_ulp=0, max_frac=0.0)
    # verify / glue
    vt = g["verify_tokens"]
    n = vt.numel()
    h = m.forward(vt, torch.arange(P, P + n),
                  Ctx("verify", slot_mapping=slots(table, range(P, P + n)), context_lens=torch.tensor([P + n], dtype=torch.int32),
                      block_tables=bt, cu_q=torch.tensor([0, n], dtype=torch.int32)))
    glue = m.compute_logits(h)
    assert_ulp(glue, g["verify_logits"], max_ulp=0, max_frac=0.0)
    # fork + tree decode
    K, F = g["tree_K_F"].tolist()
    MQ = F * (K + 1)
    forks = O.fork_topf(glue.view(1, K + 1, -1), vt.view(1, -1), [[F] * (K + 1)])
    assert torch.equal(forks, g["tree_forks"])
    jidx = [i // F for i in range(MQ)]
    toks = forks[0]
    for step in range(K):
        rope_pos = torch.tensor([P + j + 1 + step for j in jidx])
        cache_pos = [P + K + 1 + step * MQ + i for i in range(MQ)]
        h = m.forward(toks, rope_pos, Ctx("tree", slot_mapping=slots(table, cache_pos),
                                         context_lens=torch.tensor([cache_pos[-1] + 1], dtype=torch.int32), block_tables=bt,
                                         tree_step=step, tree_K=K, tree_jidx=[jidx]))
        lg = m.compute_logits(h)
        assert_ulp(lg, g["tree_logits"][step], max_ulp=0, max_frac=0.0)
        toks = lg.float().argmax(-1)


def test_tiny_qwen3_forward(golden):
    g = golden("tiny_qwen3")
    m, cfg = make_oracle(g, "qwen3", tie=True, qk_norm=True)
    bt = g["block_table"]
    table = bt[0].tolist()
    P = g["prompt"].numel()
    cu = torch.tensor([0, P], dtype=torch.int32)
    h = m.forward(g["prompt"], torch.arange(P), Ctx("prefill", slot_mapping=slots(table, range(P)), cu_q=cu, cu_k=cu))
    assert_ulp(m.compute_logits(h), g["prefill_logits"], max_ulp=0, max_frac=0.0)
    vt = g["verify_tokens"]
    n = vt.numel()
    h = m.forward(vt, torch.arange(P, P + n),
                  Ctx("verify", slot_mapping=slots(table, range(P, P + n)), context_lens=torch.tensor([P + n], dtype=torch.int32),
                      block_tables=bt, cu_q=torch.tensor([0, n], dtype=torch.int32)))
    assert_ulp(m.compute_logits(h), g["verify_logits"], max_ulp=0, max_frac=0.0)


def test_verify_stochastic_and_sampler(golden):
    """Temperature > 0: the oracle issues the reference's torch RNG calls in the same order, so a fixed CPU seed pins
    ratio acceptance, residual resampling, hit/miss rows, the jit switch and the exponential-noise sampler exactly."""
    g = golden("stochastic_golden")
    for jit in (0, 1):
        torch.manual_seed(123)
        sfx, rec, _ = O.verify_full(g["lp"], g["lq"], g["spec"], g["tt"], g["tq"], cache_hits=g["hits"], jit_speculate=bool(jit))
        for b, s in enumerate(sfx):
            assert s == g[f"sfx{jit}"][b, :len(s)].tolist()
            assert int((g[f"sfx{jit}"][b] >= 0).sum()) == len(s)
        assert rec == g[f"rec{jit}"].tolist()
    torch.manual_seed(5)
    assert torch.equal(O.sample(g["lp"][:, 0].clone(), g["tt"]), g["sample"])
    # sampler_x rescaling of the draft distribution (verify.py:101-105, sampler.py:29-31)
    torch.manual_seed(321)
    sfx, rec, _ = O.verify_full(g["lp"], g["lq"], g["spec"], g["tt"], g["tq"], cache_hits=g["hits"], jit_speculate=True,
                                sampler_x=0.6, async_fan_out=3)
    for b, s in enumerate(sfx):
        assert s == g["sfx_x"][b, :len(s)].tolist() and int((g["sfx_x"][b] >= 0).sum()) == len(s)
    assert rec == g["rec_x"].tolist()
    torch.manual_seed(6)
    assert torch.equal(O.sample(g["lp"][:, 1].clone(), g["tt"], sampler_x=0.6, F=3), g["sample_x"])
