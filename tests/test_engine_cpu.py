"""The real engine (scheduler, block managers, SpecDecodeStep, SpeculatorSync, Verifier) driven on CPU through
the oracle runner, pinned against traces produced by the reference's own modules (tests/golden/make_golden.py:
gen_engine).  Same machine + same torch calls => token streams must be identical."""
import torch

from oracle.runner import oracle_runner_factory
from ssd_amd.engine.llm_engine import LLMEngine
from ssd_amd.model_config import ModelConfig
from ssd_amd.sampling_params import SamplingParams


def mcfg(g, prefix):
    ci, cf = g[prefix + "cfg_i"].tolist(), g[prefix + "cfg_f"].tolist()
    return ModelConfig("llama", ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], cf[0], cf[1], ci[7], False)


def weights(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix) and "cfg" not in k}


COMMON = dict(max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=16, num_kvcache_blocks=64)


def test_autoregressive_matches_reference(golden):
    g = golden("engine_golden")
    eng = LLMEngine("tiny", hf_config=mcfg(g, "t_"), runner_factory=oracle_runner_factory(weights(g, "t.")), **COMMON)
    n = g["ar_tokens"].numel()
    out, metrics = eng.generate([g["prompt"].tolist()], SamplingParams(temperature=0, max_new_tokens=n, ignore_eos=True), use_tqdm=False)
    assert out[0]["token_ids"] == g["ar_tokens"].tolist()
    # the prefill count includes the token appended by postprocess, exactly like the reference (step.py:47-48)
    assert metrics["decode_total_tokens"] == n - 1 and metrics["prefill_total_tokens"] == g["prompt"].numel() + 1


def _sd(g, tag):
    K = int(g["sd_K"])
    wt = weights(g, "t.")
    if tag == "same":
        cfg_d, wd = mcfg(g, "t_"), wt
    else:
        cfg_d, wd = mcfg(g, "d_"), weights(g, "d.")
    eng = LLMEngine("tiny", hf_config=mcfg(g, "t_"), draft="tiny-draft", draft_hf_config=cfg_d, speculate=True, speculate_k=K,
                    runner_factory=oracle_runner_factory(wt, wd), **COMMON)
    want = g[f"sd_{tag}_tokens"].tolist()
    streamed, rounds_at_first = [], []
    from ssd_amd.engine.llm_engine import METRICS

    def on_tokens(sid, toks):
        if not streamed:
            rounds_at_first.append(len(METRICS["accepted_suffix_lens_with_recovery"]))
        streamed.extend(toks)
    out, metrics = eng.generate([g["prompt"].tolist()], SamplingParams(temperature=0, max_new_tokens=len(want), ignore_eos=True),
                                use_tqdm=False, stream_callback=on_tokens)
    assert out[0]["token_ids"] == want
    assert streamed == want
    assert rounds_at_first == [0], "the prefill's token must reach the stream before the first speculation round"
    lens = metrics["accepted_suffix_lens_with_recovery"]
    ref_lens = [(row >= 0).sum().item() for row in g[f"sd_{tag}_suffix"]]
    assert lens == ref_lens
    return lens, K


def test_sync_sd_matches_reference_independent_draft(golden):
    lens, K = _sd(golden("engine_golden"), "diff")
    assert min(lens) >= 1


def test_sync_sd_draft_equals_target_accepts_everything(golden):
    g = golden("engine_golden")
    lens, K = _sd(g, "same")
    assert all(n == K + 1 for n in lens)
    # speculative decoding is exact: the stream equals plain autoregressive decoding
    n = min(g["ar_tokens"].numel(), g["sd_same_tokens"].numel())
    assert g["sd_same_tokens"][:n].tolist() == g["ar_tokens"][:n].tolist()


def test_batch_of_sequences_and_eos():
    """b > 1 scheduling, EOS truncation and block recycling on synthetic tiny models."""
    cfg = ModelConfig("llama", 64, 2, 2, 1, 32, 128, 256, 1e-5, 5e5, 512, False)
    dcfg = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 512, False)
    kw = dict(max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40, weights_std=0.1)
    prompts = [[(7 * i + j) % 256 for j in range(5 + 3 * i)] for i in range(4)]
    sp = SamplingParams(temperature=0, max_new_tokens=20, ignore_eos=True)
    ar = LLMEngine("t", hf_config=cfg, max_num_seqs=4, runner_factory=oracle_runner_factory(), **kw)
    ref, _ = ar.generate(prompts, sp, use_tqdm=False)
    sd = LLMEngine("t", hf_config=cfg, draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=3, max_num_seqs=4,
                   runner_factory=oracle_runner_factory(), **kw)
    got, m = sd.generate(prompts, sp, use_tqdm=False)
    assert [o["token_ids"] for o in got] == [o["token_ids"] for o in ref]
    # every block went back to the pool
    assert len(sd.scheduler.block_manager.free_block_ids) == 40
    assert len(sd.scheduler.draft_block_manager.free_block_ids) == sd.draft_runner.num_kvcache_blocks
    # EOS: declare the 3rd generated token of sequence 0 to be EOS
    eos = ref[0]["token_ids"][2]
    ar2 = LLMEngine("t", hf_config=cfg, eos=eos, max_num_seqs=4, runner_factory=oracle_runner_factory(), **kw)
    sd2 = LLMEngine("t", hf_config=cfg, eos=eos, draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=3,
                    max_num_seqs=4, runner_factory=oracle_runner_factory(), **kw)
    sp2 = SamplingParams(temperature=0, max_new_tokens=20, ignore_eos=False)
    a, _ = ar2.generate(prompts, sp2, use_tqdm=False)
    b, _ = sd2.generate(prompts, sp2, use_tqdm=False)
    assert [o["token_ids"] for o in a] == [o["token_ids"] for o in b]
    assert a[0]["token_ids"][-1] == eos and len(a[0]["token_ids"]) <= 3


def _tiny_pair():
    cfg = ModelConfig("llama", 64, 2, 2, 1, 32, 128, 256, 1e-5, 5e5, 512, False)
    dcfg = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 512, False)
    kw = dict(max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40, weights_std=0.1)
    return cfg, dcfg, kw


def test_temperature_sd_draft_equals_target_accepts_everything():
    """verify.py:50-120 -- with q == p the ratio min(1, p/q) is 1 at every position, so a JIT-speculating sync draft
    at temperature > 0 has every token accepted; without jit_speculate the sync path has no cache hits and the
    acceptance rule falls back to the greedy comparison (verify.py:57-62), so sampled drafts get rejected."""
    cfg, _, kw = _tiny_pair()
    prompts = [[3, 9, 27, 81, 243 % 256], [5, 6, 7]]
    sp = SamplingParams(temperature=0.9, max_new_tokens=16, ignore_eos=True)
    torch.manual_seed(0)
    eng = LLMEngine("t", hf_config=cfg, draft="d", draft_hf_config=cfg, speculate=True, speculate_k=3, max_num_seqs=2,
                    jit_speculate=True, draft_weights_seed=0, weights_seed=0, runner_factory=oracle_runner_factory(), **kw)
    out, m = eng.generate(prompts, sp, use_tqdm=False)
    assert all(n == 4 for n in m["accepted_suffix_lens_with_recovery"])
    assert all(len(o["token_ids"]) == 16 for o in out)
    torch.manual_seed(0)
    eng2 = LLMEngine("t", hf_config=cfg, draft="d", draft_hf_config=cfg, speculate=True, speculate_k=3, max_num_seqs=2,
                     draft_weights_seed=0, weights_seed=0, runner_factory=oracle_runner_factory(), **kw)
    _, m2 = eng2.generate(prompts, sp, use_tqdm=False)
    assert min(m2["accepted_suffix_lens_with_recovery"]) < 4


def test_temperature_mixed_batch_and_greedy_draft():
    """Per-sequence temperatures in one batch (greedy rows keep the exact greedy stream), and a greedy draft under a
    sampling target (draft_temperature=0: q is one-hot, acceptance probability is p(x))."""
    cfg, dcfg, kw = _tiny_pair()
    prompts = [[(7 * i + j) % 256 for j in range(5 + 3 * i)] for i in range(3)]
    sps = [SamplingParams(temperature=0, max_new_tokens=12, ignore_eos=True),
           SamplingParams(temperature=1.0, max_new_tokens=12, ignore_eos=True),
           SamplingParams(temperature=0.7, draft_temperature=0.0, max_new_tokens=12, ignore_eos=True)]
    ar = LLMEngine("t", hf_config=cfg, max_num_seqs=4, runner_factory=oracle_runner_factory(), **kw)
    ref, _ = ar.generate(prompts[:1], sps[0], use_tqdm=False)
    torch.manual_seed(1)
    sd = LLMEngine("t", hf_config=cfg, draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=3, max_num_seqs=4,
                   jit_speculate=True, runner_factory=oracle_runner_factory(), **kw)
    got, m = sd.generate(prompts, sps, use_tqdm=False)
    assert got[0]["token_ids"] == ref[0]["token_ids"]
    assert all(len(o["token_ids"]) == 12 for o in got)
    assert all(0 <= t < cfg.vocab_size for o in got for t in o["token_ids"])


def test_abort_all_returns_blocks_and_engine_stays_usable():
    """LLMEngine.abort_all (used by bench.py between its step timing and the reference-protocol run): every KV block of the
    dropped request returns to both pools and a following generate() is unaffected."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 1024, False)
    kw = dict(hf_config=t, max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=32, num_kvcache_blocks=48, weights_std=0.1)
    sp = SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=12)
    ref, _ = LLMEngine("t", runner_factory=oracle_runner_factory(), **kw).generate([[1, 2, 3, 4, 5]], sp, use_tqdm=False)
    eng = LLMEngine("t", runner_factory=oracle_runner_factory(), draft="d", draft_hf_config=t, draft_weights_seed=0, speculate=True,
                    speculate_k=3, num_draft_kvcache_blocks=48, **kw)
    eng.add_request([9, 8, 7, 6, 5, 4], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=100))
    step = eng.create_inference_step(eng.config)
    for _ in range(4):
        eng.step(step)
    assert not eng.is_finished()
    eng.abort_all()
    assert eng.is_finished()
    assert len(eng.scheduler.block_manager.free_block_ids) == 48 and len(eng.scheduler.draft_block_manager.free_block_ids) == 48
    out, _ = eng.generate([[1, 2, 3, 4, 5]], sp, use_tqdm=False)
    assert out[0]["token_ids"] == ref[0]["token_ids"]


def test_fully_cached_sequence_still_has_tokens_to_prefill():
    """A sequence of exactly n * block_size tokens whose blocks are all in the prefix cache (a preempted sequence coming
    back): the reference would schedule an empty prefill with no logits to sample from; here its last block is recomputed."""
    from ssd_amd.engine.block_manager import BlockManager
    from ssd_amd.engine.sequence import Sequence
    Sequence.block_size = 4
    bm = BlockManager(8, 4)
    a = Sequence(list(range(8)))
    bm.allocate(a)
    assert a.num_cached_tokens == 0
    bm.deallocate(a)                       # blocks return to the free list but stay hashed
    b = Sequence(list(range(8)))
    bm.allocate(b)
    assert b.num_cached_tokens == 4        # first block hit, the last full block is taken fresh
    c = Sequence(list(range(8)) + [99])    # one more token: both full blocks may hit, the partial block is computed
    bm.allocate(c)
    assert c.num_cached_tokens == 8


def test_generation_stops_at_max_model_len_instead_of_crashing():
    """A request that runs into max_model_len: the reference's scheduler refuses every further step, the sequence preempts
    itself and the engine dies on an empty batch.  Here it finishes with what it has -- a prefix of the unconstrained
    stream; the cap sits one lookahead below max_model_len in the speculative modes."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 1024, False)
    prompts = [[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [3, 4, 5, 6, 7, 8, 9]]
    sp = SamplingParams(temperature=0, max_new_tokens=200, ignore_eos=True)
    base = dict(hf_config=t, max_num_seqs=2, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40,
                num_draft_kvcache_blocks=40, weights_std=0.1)
    free, _ = LLMEngine("t", runner_factory=oracle_runner_factory(), max_model_len=256, **base).generate(
        prompts, SamplingParams(temperature=0, max_new_tokens=60, ignore_eos=True), use_tqdm=False)
    K, F = 3, 2
    for mode, lookahead in (("ar", 1), ("sync", K + 1), ("async", K + 1 + K * F * (K + 1))):
        kw = dict(base, max_model_len=64)
        if mode != "ar":
            kw.update(draft="d", draft_hf_config=t, draft_weights_seed=0, speculate=True, speculate_k=K)
        if mode == "async":
            kw.update(draft_async=True, async_fan_out=F, jit_speculate=True, inprocess_draft=True)
        out, _ = LLMEngine("t", runner_factory=oracle_runner_factory(), **kw).generate(prompts, sp, use_tqdm=False)
        for p, o, f in zip(prompts, out, free):
            n = len(p) + len(o["token_ids"])
            assert 64 - lookahead - K <= n <= 64, (mode, n)
            assert o["token_ids"] == f["token_ids"][:len(o["token_ids"])], mode
