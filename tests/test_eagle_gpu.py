"""EAGLE-3 on the GPU: the HIP draft model and the target's activation taps against the
golden vectors of the reference's own modules (tests/golden/tiny_eagle3.npz), and the whole asynchronous EAGLE engine
against the CPU oracle engine on the same weights.  Tolerances as in tests/test_model_gpu.py (bf16 intermediates)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import eagle_models_from_golden, ulp_stats, assert_stream_matches, seq_margins

BF = torch.bfloat16


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def i64(x):
    return torch.tensor(list(x), dtype=torch.int64, device="cuda")


def i32(x):
    return torch.tensor(list(x), dtype=torch.int32, device="cuda")


def slots(table, positions, bs=16):
    return i32([table[p // bs] * bs + p % bs for p in positions])


def close(got, want, what, max_abs=0.05, mean_abs=0.01):
    got, want = got.float().cpu(), want.float()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin), f"{what}: -inf pattern differs"
    d = (got[fin] - want[fin]).abs()
    print(f"{what}: max abs {d.max().item():.4f} mean {d.mean().item():.5f}")
    assert d.max().item() <= max_abs, f"{what}: max abs diff {d.max().item()}"
    assert d.mean().item() <= mean_abs, f"{what}: mean abs diff {d.mean().item()}"


def taps_close(got, want, what, h):
    """Residual-stream rows.  x = hidden + residual is a sum of GEMM outputs that are each rounded to bf16, so a one-ulp flip
    of a LARGE addend survives in a small sum: the bound is two ulps of the largest magnitude of the row's tap (not of the
    element), and the mean stays within a fraction of an ulp."""
    got, want = got.float().cpu(), want.float()
    d = (got - want).abs()
    T = want.shape[0]
    rowmax = want.view(T, -1, h).abs().amax(dim=-1, keepdim=True).expand(T, want.shape[1] // h, h).reshape(T, -1)
    bound = 0.02 + rowmax / 64.0
    print(f"{what}: max abs {d.max().item():.4f} mean {d.mean().item():.5f} worst d/bound {(d / bound).max().item():.3f}")
    assert bool((d <= bound).all()), f"{what}: {(d > bound).sum().item()} elements beyond two ulps of their row's magnitude"
    assert d.mean().item() <= 0.004 * (1.0 + want.abs().mean().item()), f"{what}: mean abs diff {d.mean().item()}"


def logits_close(got, want, what):
    close(got, want, what)
    g, w = got.float().cpu(), want.float()
    top2 = w.topk(2, dim=-1).values
    assert bool(((g.argmax(-1) == w.argmax(-1)) | (top2[:, 0] - top2[:, 1] < 0.0625)).all()), f"{what}: argmax differs beyond a near-tie"


def test_rmsnorm_pair_matches_two_norms(gpu):
    from oracle import ops as O
    from ssd_amd.hip import ops as H
    torch.manual_seed(0)
    for T, Hd in ((1, 128), (7, 4096), (24, 6144), (40, 2048)):
        x0, x1 = torch.randn(T, Hd).to(BF), (3.0 * torch.randn(T, Hd)).to(BF)
        w0, w1 = (1 + 0.1 * torch.randn(Hd)).to(BF), (1 + 0.1 * torch.randn(Hd)).to(BF)
        out = torch.zeros(H.frag_numel(T, 2 * Hd), dtype=BF, device=gpu)
        H.rmsnorm_pair(x0.cuda(), w0.cuda(), x1.cuda(), w1.cuda(), 1e-5, T, Hd, out_frag=out)
        rows = torch.zeros(T, 2 * Hd, dtype=BF, device=gpu)
        H.frag_to_rows(out, rows, T, 2 * Hd)
        want = torch.cat([O.rmsnorm(x0, w0, 1e-5), O.rmsnorm(x1, w1, 1e-5)], dim=-1)
        mx, frac = ulp_stats(rows.cpu(), want)
        assert mx <= 1 and frac <= 0.01, (T, Hd, mx, frac)


def mk_draft(g, gpu):
    from ssd_amd.eagle import HipEagleDraft
    _, _, dcfg, dw, _, K, F = eagle_models_from_golden(g)
    m = HipEagleDraft(dcfg, max_tokens=64, max_seqs=2, max_blocks=12, block_size=16, max_model_len=512, device=gpu)
    m.load_weights(iter(dw.items()))
    m.alloc_kv(24)
    return m, dcfg, K, F


def test_eagle_draft_forward_flavours_vs_reference_golden(gpu, golden):
    """Prefill (shifted, fc of target activations), the JIT chain (self-conditioning), the variable-length glue over
    [extend | recovery | spec] rows and the tree steps, each fed the reference's own inputs of that stage."""
    from ssd_amd.model import AttnMeta
    from ssd_amd.hip import ops as H
    g = golden("tiny_eagle3")
    m, dcfg, K, F = mk_draft(g, gpu)
    bt = g["d_block_table"].to(torch.int32).cuda().contiguous()
    table = g["d_block_table"][0].tolist()
    prompt = g["prompt"].tolist()
    P = len(prompt)
    full = torch.arange(dcfg.draft_vocab_size) + g["d.d2t"]
    assert torch.equal(m.target_index.cpu(), full)
    # prefill
    n = P - 1
    m.project(g["t_prefill_acts"][:-1].cuda().contiguous(), n, m.buf_cond)
    m.forward(i64(prompt[1:]), i64(range(n)), n, AttnMeta(H.MODE_CAUSAL, 1, n, slots(table, range(n)), i32([n]), bt, cu_q=i32([0, n])))
    close(m.buf_pre[:n], g["d_prefill_prenorm"], "prefill prenorm")
    m.compute_logits(n)
    logits_close(m.logits[:n], g["d_prefill_logits"], "prefill logits")

    def chain(rec, act, pos0, tk, lk, pk):
        tok = int(rec)
        m.project(act.cuda().contiguous(), 1, m.buf_cond)
        for i in range(K):
            p = pos0 + i
            m.forward(i64([tok]), i64([p]), 1, AttnMeta(H.MODE_CAUSAL, 1, 1, slots(table, [p]), i32([p + 1]), bt, q_per_seq=1))
            close(m.buf_pre[:1], g[pk][i:i + 1], f"{pk}[{i}]")
            m.compute_logits(1)
            logits_close(m.logits[:1], g[lk][i:i + 1], f"{lk}[{i}]")
            tok = int(g[tk][i])                              # follow the reference's chain
            m.buf_cond[:1].copy_(g[pk][i:i + 1].cuda())

    chain(g["rec0"][0], g["t_prefill_acts"][-1:], P - 1, "jit1_tokens", "jit1_logits", "jit1_prenorm")
    N = P + K + 2
    chain(g["rec1"][0], g["t_verify_acts"][K:K + 1], N - 2, "jit2_tokens", "jit2_logits", "jit2_prenorm")
    # glue: n_ext extend rows + recovery + K spec rows, bottom-right aligned causal attention over the paged cache
    n_ext = int(g["glue_n_ext"][0])
    ids = g["glue_ids"].tolist()
    n = len(ids)
    m.project(g["t_verify_acts"][:K + 1].cuda().contiguous(), K + 1, m.buf_cond)
    close(m.buf_cond[:K + 1], g["glue_hs"][:K + 1], "fc of extend + recovery rows", max_abs=0.02, mean_abs=0.003)
    m.buf_cond[:n].copy_(g["glue_hs"].cuda())
    base = N - 2 - n_ext
    m.forward(i64(ids), i64(range(base, base + n)), n,
              AttnMeta(H.MODE_CAUSAL, 1, n, slots(table, range(base, base + n)), i32([N - 1 + K]), bt, cu_q=i32([0, n])))
    close(m.buf_pre[:n], g["glue_prenorm"], "glue prenorm")
    m.compute_logits(K + 1, gather=i32(range(n_ext, n)), rows=K + 1)
    logits_close(m.logits[:K + 1], g["glue_logits"][n_ext:], "glue logits ([recovery | spec] rows)")
    # fork on the device from the reference's logits, then the tree steps
    MQ = F * (K + 1)
    counts = torch.full((1, K + 1), F, dtype=torch.int32)
    offs = (torch.cumsum(counts, 1) - counts).to(torch.int32)
    forks = torch.zeros(1, MQ, dtype=torch.int64, device="cuda")
    returned = torch.cat([g["rec1"], g["jit2_tokens"]]).view(1, -1)
    H.fork_topf(g["glue_logits"][n_ext:].cuda().contiguous(), m.V, m.V, returned.cuda(), counts.cuda(), offs.cuda(), 1, K, MQ, forks)
    assert forks.cpu().tolist() == g["tree_forks"].tolist()
    jidx = [i // F for i in range(MQ)]
    toks = g["tree_forks"][0].cuda()
    m.buf_cond[:MQ].copy_(g["glue_prenorm"][n_ext:][torch.tensor(jidx)].cuda())
    Pb = N - 2
    for step in range(K):
        rope_pos = [Pb + j + 1 + step for j in jidx]
        cache_pos = [Pb + K + 1 + step * MQ + i for i in range(MQ)]
        meta = AttnMeta(H.MODE_TREE, 1, MQ, slots(table, cache_pos), i32([cache_pos[-1] + 1]), bt, q_per_seq=MQ,
                        tree_K=K, tree_mq=MQ, tree_step=step, tree_F=F)
        m.forward(toks.contiguous(), i64(rope_pos), MQ, meta)
        close(m.buf_pre[:MQ], g["tree_prenorm"][step], f"tree prenorm {step}")
        m.compute_logits(MQ)
        logits_close(m.logits[:MQ], g["tree_logits"][step], f"tree logits {step}")
        toks = g["tree_logits"][step].float().argmax(-1).cuda()
        m.buf_cond[:MQ].copy_(g["tree_prenorm"][step].cuda())


def test_target_activation_taps_vs_reference_golden(gpu, golden):
    from ssd_amd.model import HipDecoder, AttnMeta
    from ssd_amd.hip import ops as H
    g = golden("tiny_eagle3")
    tcfg, tw, _, _, taps, K, _ = eagle_models_from_golden(g)
    dec = HipDecoder(tcfg, max_tokens=64, max_seqs=2, max_blocks=12, block_size=16, max_model_len=512, device=gpu, taps=taps)
    dec.load_weights(iter(tw.items()))
    dec.alloc_kv(24)
    bt = g["t_block_table"].to(torch.int32).cuda().contiguous()
    table = g["t_block_table"][0].tolist()
    prompt = g["prompt"].tolist()
    P = len(prompt)
    dec.forward(i64(prompt), i64(range(P)), P, AttnMeta(H.MODE_CAUSAL, 1, P, slots(table, range(P)), i32([P]), bt, cu_q=i32([0, P])))
    taps_close(dec.acts[:P], g["t_prefill_acts"], "prefill taps", tcfg.hidden_size)
    dec.compute_logits(P)
    logits_close(dec.logits[:P], g["t_prefill_logits"], "target prefill logits")
    vt = torch.cat([g["rec0"], g["jit1_tokens"]]).tolist()
    n = len(vt)
    dec.forward(i64(vt), i64(range(P, P + n)), n,
                AttnMeta(H.MODE_CAUSAL, 1, n, slots(table, range(P, P + n)), i32([P + n]), bt, q_per_seq=n))
    taps_close(dec.acts[:n], g["t_verify_acts"], "verify taps", tcfg.hidden_size)
    # the first tap is layer 0: the embedding rows themselves, bit-exact
    h = tcfg.hidden_size
    assert torch.equal(dec.acts[:n, :h].cpu().view(torch.int16), g["t_verify_acts"][:, :h].contiguous().view(torch.int16))


def _engine_pair():
    from tests.eagle_util import eagle_cfgs, peaky_weights
    t, d = eagle_cfgs(h_t=256, h_d=128, V=512, Vd=256, hd=64)
    tw, dw = peaky_weights(t, d)
    return t, d, tw, dw


@pytest.mark.parametrize("bs", [1, 3])
def test_eagle_engine_hip_vs_oracle(gpu, bs):
    """The asynchronous EAGLE-3 engine end to end on the HIP path (target taps -> wire -> draft server with JIT, cache
    hits, extend rows, tree) against (a) plain autoregressive decoding of the same HIP target: speculation is exact --
    and (b) the CPU oracle engine on the same weights: same stream and same acceptance trace up to a near-tie."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    from tests.eagle_util import eagle_kwargs, ENGINE_KW, PROMPTS
    from tests.test_model_gpu import hip_factory
    t, d, tw, dw = _engine_pair()
    prompts = [[x % t.vocab_size for x in p] for p in PROMPTS[:bs]]
    sp = SamplingParams(temperature=0, max_new_tokens=40, ignore_eos=True)
    kw_e = eagle_kwargs(t, d, bs=bs)
    kw_e.update(kvcache_block_size=16, num_kvcache_blocks=64, num_draft_kvcache_blocks=64)
    kw_ar = dict(ENGINE_KW, hf_config=t, max_num_seqs=bs, kvcache_block_size=16, num_kvcache_blocks=64)
    # oracle: AR stream + margins, and the EAGLE acceptance trace
    from ssd_amd.utils.topology import Topology
    cpu = Topology(0, 1, torch.device("cpu"), "target", 0, 1)       # the oracle engine stays on the host of the GPU box
    o_ar = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw), topology=cpu, **kw_ar)
    want, _ = o_ar.generate(prompts, sp, use_tqdm=False)
    margins = o_ar.model_runner.margin_log
    o_eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), inprocess_draft=True,
                      topology=cpu, **kw_e)
    o_out, o_m = o_eng.generate(prompts, sp, use_tqdm=False)
    o_lens = list(o_m["accepted_suffix_lens_with_recovery"])
    assert [o["token_ids"] for o in o_out] == [o["token_ids"] for o in want]
    # HIP
    eng = LLMEngine("t", runner_factory=hip_factory(tw, dw), inprocess_draft=True, **kw_e)
    out, m = eng.generate(prompts, sp, use_tqdm=False)
    stats = eng.draft_server.stats
    lens = list(m["accepted_suffix_lens_with_recovery"])
    print(f"EAGLE HIP: lens {lens}\n     oracle: lens {o_lens}\n hits {stats}")
    full = True
    for i, (a, b) in enumerate(zip(out, want)):
        n = assert_stream_matches(a["token_ids"], b["token_ids"], seq_margins(margins, i), len(prompts[i]), f"seq {i}")
        full = full and n == len(b["token_ids"])
    assert max(lens) >= 2 and stats["hits"] >= 1            # the extend rows and the cached-prenorm path ran
    # the draft's own near-ties are not recorded, so the acceptance traces may differ slightly; the step counts must not
    assert abs(len(lens) - len(o_lens)) <= max(3, len(o_lens) // 3), (lens, o_lens)
    eng.exit()


@pytest.mark.parametrize("preset", ["eagle3-llama-3.1-8b", "eagle3-llama-3.3-70b"])
def test_eagle_draft_at_real_shapes_vs_oracle(gpu, preset):
    """The published EAGLE-3 draft shapes (8B: h 4096, QKV over K = 8192, fc 12288 -> 4096, 32000-token head scattered into
    128256; 70B: h 6144, fc 24576 -> 6144), default kernel dispatch: a shifted prefill, a decode row, a 15-row glue with
    extend rows and a 24-row tree step, HIP vs the CPU oracle on the same synthetic weights."""
    import random
    from oracle.eagle import OracleEagleDraft
    from oracle.model import Ctx
    from ssd_amd import weights as W
    from ssd_amd.eagle import HipEagleDraft
    from ssd_amd.model import AttnMeta
    from ssd_amd.model_config import PRESETS
    from ssd_amd.hip import ops as H
    cfg = PRESETS[preset]
    full = W.synthetic_state_dict(cfg, seed=4, std=0.02)
    bs, nblocks = 256, 3
    m = HipEagleDraft(cfg, max_tokens=64, max_seqs=1, max_blocks=4, block_size=bs, max_model_len=1024, device=gpu)
    m.load_weights(iter(full.items()))
    m.alloc_kv(nblocks)
    orc = OracleEagleDraft(cfg, full, nblocks, bs)
    random.seed(2)
    torch.manual_seed(2)
    table = [2, 0, 1]
    bt = torch.tensor([table + [-1]], dtype=torch.int32)
    A = cfg.eagle_taps * cfg.d_model_target

    def sl(ps):
        return torch.tensor([table[p // bs] * bs + p % bs for p in ps], dtype=torch.int32)

    def t64(x):
        return torch.tensor(list(x), dtype=torch.int64)

    def compare(pre_ref, n, what, rows=None):
        lg_ref = orc.compute_logits(pre_ref if rows is None else pre_ref[rows])
        got_pre = m.buf_pre[:n].float().cpu()
        d = (got_pre - pre_ref.float()).abs()
        scale = pre_ref.float().std().item()
        print(f"{preset} {what}: prenorm std {scale:.3f} max |d| {d.max().item():.4f} mean {d.mean().item():.5f}")
        assert d.max().item() <= 0.05 * max(1.0, scale) and d.mean().item() <= 0.01 * max(1.0, scale)
        k = lg_ref.shape[0]
        got = m.logits[:k].float().cpu()
        fin = torch.isfinite(lg_ref.float())
        assert torch.equal(torch.isfinite(got), fin)
        dl = (got[fin] - lg_ref.float()[fin]).abs()
        ls = lg_ref.float()[fin].std().item()
        print(f"   logits std {ls:.3f} max |d| {dl.max().item():.4f} mean {dl.mean().item():.5f}")
        assert dl.max().item() <= 0.05 * max(1.0, ls) and dl.mean().item() <= 0.01 * max(1.0, ls)
        top2 = lg_ref.float().topk(2, dim=-1).values
        assert bool(((got.argmax(-1) == lg_ref.float().argmax(-1)) | (top2[:, 0] - top2[:, 1] < 0.0625)).all())

    # prefill of 40 shifted tokens from target activations (fc at M = 40: the prefill GEMM path)
    P = 40
    ids = [random.randint(0, 10000) for _ in range(P)]
    acts = torch.randn(P, A).to(BF)
    cu = torch.tensor([0, P], dtype=torch.int32)
    ref = orc.forward(t64(ids), t64(range(P)), acts, Ctx("prefill", slot_mapping=sl(range(P)), cu_q=cu, cu_k=cu))
    m.project(acts.cuda(), P, m.buf_cond)
    m.forward(t64(ids).cuda(), t64(range(P)).cuda(), P,
              AttnMeta(H.MODE_CAUSAL, 1, P, sl(range(P)).cuda(), torch.tensor([P], dtype=torch.int32).cuda(), bt.cuda(), cu_q=cu.cuda()))
    m.compute_logits(P)
    compare(ref, P, "prefill")
    # one decode row conditioned on a draft-width row
    cond = ref[-1:].clone()
    tok = [random.randint(0, 10000)]
    ref1 = orc.forward(t64(tok), t64([P]), cond, Ctx("decode", slot_mapping=sl([P]), context_lens=torch.tensor([P + 1], dtype=torch.int32), block_tables=bt))
    m.buf_cond[:1].copy_(cond.cuda())
    m.forward(t64(tok).cuda(), t64([P]).cuda(), 1, AttnMeta(H.MODE_CAUSAL, 1, 1, sl([P]).cuda(), torch.tensor([P + 1], dtype=torch.int32).cuda(), bt.cuda(), q_per_seq=1))
    m.compute_logits(1)
    compare(ref1, 1, "decode")
    # glue: 7 extend + recovery + 7 spec rows at positions P-6 .. P+8 (K = 7), logits only for the last 8 rows
    K = 7
    n = 2 * K + 1
    base = P + 1 - K
    gids = [random.randint(0, 10000) for _ in range(n)]
    tc = orc.project(torch.randn(K + 1, A).to(BF))
    hs = torch.cat([tc, (0.5 * torch.randn(K, cfg.hidden_size)).to(BF)], dim=0)
    ps = list(range(base, base + n))
    refg = orc.forward(t64(gids), t64(ps), hs, Ctx("verify", slot_mapping=sl(ps), context_lens=torch.tensor([base + n], dtype=torch.int32),
                                                 block_tables=bt, cu_q=torch.tensor([0, n], dtype=torch.int32)))
    m.buf_cond[:n].copy_(hs.cuda())
    m.forward(t64(gids).cuda(), t64(ps).cuda(), n,
              AttnMeta(H.MODE_CAUSAL, 1, n, sl(ps).cuda(), torch.tensor([base + n], dtype=torch.int32).cuda(), bt.cuda(),
                       cu_q=torch.tensor([0, n], dtype=torch.int32).cuda()))
    rows = torch.arange(K, n)
    m.compute_logits(n, gather=rows.to(torch.int32).cuda(), rows=K + 1)
    compare(refg, n, "glue", rows=rows)
    # one tree step: 24 branches (F = 3), step 0
    F = 3
    MQ = F * (K + 1)
    Pb = base + K                      # position of the recovery row
    jidx = [i // F for i in range(MQ)]
    toks = [random.randint(0, 10000) for _ in range(MQ)]
    hid = refg[K:][torch.tensor(jidx)].clone()
    rope_pos = [Pb + j + 1 for j in jidx]
    cache_pos = [Pb + K + 1 + i for i in range(MQ)]
    reft = orc.forward(t64(toks), t64(rope_pos), hid, Ctx("tree", slot_mapping=sl(cache_pos), context_lens=torch.tensor([cache_pos[-1] + 1], dtype=torch.int32),
                                                       block_tables=bt, tree_step=0, tree_K=K, tree_jidx=[jidx]))
    m.buf_cond[:MQ].copy_(hid.cuda())
    m.forward(t64(toks).cuda(), t64(rope_pos).cuda(), MQ,
              AttnMeta(H.MODE_TREE, 1, MQ, sl(cache_pos).cuda(), torch.tensor([cache_pos[-1] + 1], dtype=torch.int32).cuda(), bt.cuda(),
                       q_per_seq=MQ, tree_K=K, tree_mq=MQ, tree_step=0, tree_F=F))
    m.compute_logits(MQ)
    compare(reft, MQ, "tree step")


def test_eagle_engine_at_real_8b_shapes_is_exact(gpu):
    """Llama-3.1-8B + its EAGLE-3 draft (public shapes, synthetic weights with a few boosted head rows so that drafts get
    accepted) through the product engine with hipGraphs: asynchronous EAGLE speculation must reproduce the same target's
    autoregressive greedy stream (up to a recorded near-tie), with cache hits and multi-token acceptances along the way."""
    import gc
    from ssd_amd import weights as W
    from ssd_amd.engine.llm_engine import LLMEngine, hip_runner_factory
    from ssd_amd.model_config import PRESETS
    from ssd_amd.sampling_params import SamplingParams
    tcfg, dcfg = PRESETS["llama-3.1-8b"], PRESETS["eagle3-llama-3.1-8b"]
    d2t = W.synthetic_tensor("d2t", (dcfg.draft_vocab_size,), 1, 0.02, "cpu", recipe={"target_vocab": tcfg.vocab_size})
    g = torch.Generator().manual_seed(7)
    picks = torch.randperm(dcfg.draft_vocab_size, generator=g)[:3]
    boost = {"t": [int(p + d2t[p]) for p in picks], "d": [int(p) for p in picks]}

    def source(cfg, seed, which):
        for name, w in W.synthetic_weights(cfg, seed, 0.02, gen_device="cuda", out_device="cuda:0"):
            if name == "lm_head.weight":
                rows = torch.tensor(boost[which], device=w.device)
                w[rows] = (w[rows].float() * 6.0).to(w.dtype)
            yield name, w

    def factory(config, model_cfg, *, is_draft, topo, **kw):
        src = source(model_cfg, 1 if is_draft else 0, "d" if is_draft else "t")
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, weight_source=src, **kw)

    import random
    random.seed(3)
    prompt = [random.randint(0, 10000) for _ in range(96)]
    sp = SamplingParams(temperature=0, max_new_tokens=48, ignore_eos=True)
    common = dict(hf_config=tcfg, max_num_seqs=1, max_model_len=1024, max_num_batched_tokens=1024, kvcache_block_size=256,
                  num_kvcache_blocks=6, runner_factory=factory)
    ar = LLMEngine("llama-3.1-8b", **common)
    ar.model_runner.margin_log = {}
    want, _ = ar.generate([prompt], sp, use_tqdm=False)
    margins = seq_margins(ar.model_runner.margin_log, 0)
    del ar
    gc.collect()
    torch.cuda.empty_cache()
    eng = LLMEngine("llama-3.1-8b", draft="eagle3-llama-3.1-8b", draft_hf_config=dcfg, speculate=True, speculate_k=7, draft_async=True,
                    async_fan_out=3, jit_speculate=True, use_eagle=True, inprocess_draft=True, num_draft_kvcache_blocks=6, **common)
    assert eng.config.eagle_layers == [2, 16, 29]
    got, m = eng.generate([prompt], sp, use_tqdm=False)
    lens, stats = m["accepted_suffix_lens_with_recovery"], eng.draft_server.stats
    print(f"8B + EAGLE-3: accepted lens {lens}, server {stats}")
    assert_stream_matches(got[0]["token_ids"], want[0]["token_ids"], margins, len(prompt), "8B EAGLE vs AR")
    assert max(lens) >= 2 and stats["hits"] >= 1
    eng.exit()
