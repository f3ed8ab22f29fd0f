"""Temperature > 0 through the whole engine on the GPU: sampling prefill/decode, sampled draft
chains with stored q logits, ratio verification with residual resampling (ssd/utils/verify.py:50-181).

The device RNG is not torch's, so parity is statistical: speculative sampling is exact, i.e. the token stream of the
speculative engine has the same distribution as plain autoregressive sampling from the target.  We draw a few
thousand continuations of one prompt from both engines and compare the marginals of the first generated positions
with a two-sample chi-square test (p > 1e-4), plus the deterministic consequences of the rule (q == p accepts
everything; temperature-0 rows inside a sampling batch stay exactly greedy)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def cfgs():
    from ssd_amd.model_config import ModelConfig
    t = ModelConfig("llama", 256, 2, 4, 2, 64, 512, 128, 1e-5, 5e5, 1024, False)
    d = ModelConfig("llama", 128, 1, 2, 1, 64, 256, 128, 1e-5, 5e5, 1024, True)
    return t, d


KW = dict(max_model_len=128, max_num_batched_tokens=1024, kvcache_block_size=16, num_kvcache_blocks=256,
          num_draft_kvcache_blocks=256, weights_std=0.25)
PROMPT = [(5 * j + 1) % 128 for j in range(9)]


def perturbed_pair(noise: float):
    """Target weights and a draft of the same architecture whose weights are the target's plus noise: a correlated
    draft, so that both the accept and the reject/residual branches are exercised."""
    from ssd_amd import weights as W
    from ssd_amd.engine.llm_engine import hip_runner_factory
    t, _ = cfgs()
    wt = W.synthetic_state_dict(t, 0, KW["weights_std"])
    g = torch.Generator().manual_seed(99)
    wd = {k: (v.float() + noise * KW["weights_std"] * torch.randn(v.shape, generator=g)).to(v.dtype) if "norm" not in k else v
          for k, v in wt.items()}

    def factory(config, model_cfg, *, is_draft, topo, **kw):
        ws = wd if is_draft else wt
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, weight_source=iter(ws.items()), **kw)
    return t, factory


def two_sample_ok(a: torch.Tensor, b: torch.Tensor, V: int, what: str):
    from scipy.stats import chi2_contingency
    ca, cb = torch.bincount(a, minlength=V).double(), torch.bincount(b, minlength=V).double()
    keep = (ca + cb) >= 10
    tab = torch.stack([torch.cat([ca[keep], ca[~keep].sum().view(1)]), torch.cat([cb[keep], cb[~keep].sum().view(1)])])
    tab = tab[:, tab.sum(0) > 0]
    stat, p, dof, _ = chi2_contingency(tab.numpy())
    print(f"{what}: chi2={stat:.1f} dof={dof} p={p:.4f}")
    assert p > 1e-4, f"{what}: two-sample chi-square p={p}"


def draw(eng, sp, rounds, per_round, n_new):
    toks = []
    lens = []
    for _ in range(rounds):
        out, m = eng.generate([PROMPT] * per_round, sp, use_tqdm=False)
        toks.extend(o["token_ids"][:n_new] for o in out)
        lens.extend(m.get("accepted_suffix_lens_with_recovery", []))
    return torch.tensor(toks), lens


@pytest.mark.parametrize("draft_temp", [None, 0.0])
def test_speculative_sampling_matches_autoregressive_distribution(gpu, draft_temp):
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, factory = perturbed_pair(0.03)
    B, R, n_new = 48, 40, 5
    sp = SamplingParams(temperature=0.7, draft_temperature=draft_temp, max_new_tokens=n_new, ignore_eos=True)
    ar = LLMEngine("t", hf_config=t, max_num_seqs=B, runner_factory=factory, **KW)
    a, _ = draw(ar, SamplingParams(temperature=0.7, max_new_tokens=n_new, ignore_eos=True), R, B, n_new)
    del ar
    sd = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=t, speculate=True, speculate_k=3, max_num_seqs=B,
                   jit_speculate=True, runner_factory=factory, **KW)
    s, lens = draw(sd, sp, R, B, n_new)
    print("mean accepted (+recovery):", sum(lens) / len(lens), "distinct tokens at pos 1:", a[:, 1].unique().numel())
    assert a[:, 1].unique().numel() >= 8, "degenerate test distribution"
    assert 1.4 < sum(lens) / len(lens) < 3.8          # some accepted, some rejected: both branches exercised
    for pos in range(1, n_new):
        two_sample_ok(a[:, pos], s[:, pos], t.vocab_size, f"marginal of generated position {pos} (draft_temp={draft_temp})")


def test_same_model_accepts_nearly_everything_and_greedy_rows_stay_greedy(gpu):
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, _ = cfgs()
    eng = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=t, speculate=True, speculate_k=3, max_num_seqs=8,
                    jit_speculate=True, weights_seed=0, draft_weights_seed=0, **KW)
    out, m = eng.generate([PROMPT] * 8, SamplingParams(temperature=0.9, max_new_tokens=24, ignore_eos=True), use_tqdm=False)
    lens = m["accepted_suffix_lens_with_recovery"]
    # q and p come from differently tiled launches (M=1 decode vs M=4 verify): equal up to bf16 noise, so min(1,p/q) ~ 1
    assert sum(lens) / len(lens) > 3.8, lens
    assert len({tuple(o["token_ids"]) for o in out}) > 1          # rows draw independently
    # mixed batch: the temperature-0 sequence reproduces the pure greedy stream
    sps = [SamplingParams(temperature=0, max_new_tokens=16, ignore_eos=True)] + \
          [SamplingParams(temperature=1.0, max_new_tokens=16, ignore_eos=True)] * 3
    mixed, _ = eng.generate([PROMPT] * 4, sps, use_tqdm=False)
    greedy, _ = eng.generate([PROMPT], sps[0], use_tqdm=False)
    assert mixed[0]["token_ids"] == greedy[0]["token_ids"]


def test_sync_without_jit_falls_back_to_greedy_acceptance(gpu):
    """verify.py:57-62: without JIT speculation and without cache hits no row takes the ratio path; acceptance is the
    greedy comparison and the recovery token is drawn from p at the stopping position."""
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = cfgs()
    eng = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=d, speculate=True, speculate_k=3, max_num_seqs=8, **KW)
    out, m = eng.generate([PROMPT] * 8, SamplingParams(temperature=0.7, max_new_tokens=12, ignore_eos=True), use_tqdm=False)
    assert all(len(o["token_ids"]) == 12 for o in out)
    assert all(1 <= n <= 4 for n in m["accepted_suffix_lens_with_recovery"])


def test_async_speculative_sampling_matches_autoregressive_distribution(gpu):
    """The asynchronous (SSD) protocol at temperature > 0 on the GPU, draft server in-process over the loopback
    transport: sampled JIT chains, sampled tree branches whose logits are cached and shipped as logits_q, ratio
    verification on hits and JIT rows.  Same exactness criterion as the synchronous test."""
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, factory = perturbed_pair(0.03)
    B, R, n_new = 8, 120, 6
    sp = SamplingParams(temperature=0.7, max_new_tokens=n_new, ignore_eos=True)
    ar = LLMEngine("t", hf_config=t, max_num_seqs=B, runner_factory=factory, **KW)
    a, _ = draw(ar, sp, R, B, n_new)
    del ar
    kw = dict(KW, num_draft_kvcache_blocks=256)
    sd = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=t, speculate=True, speculate_k=3, max_num_seqs=B,
                   draft_async=True, async_fan_out=3, jit_speculate=True, inprocess_draft=True, runner_factory=factory, **kw)
    s, lens = draw(sd, sp, R, B, n_new)
    st = sd.draft_server.stats
    print("async: mean accepted (+recovery)", sum(lens) / len(lens), "cache hit rate", st["hits"] / max(1, st["requests"]))
    assert st["hits"] > 0 and st["hits"] < st["requests"]          # both the cache path and the JIT path served requests
    assert 1.3 < sum(lens) / len(lens) < 3.9
    for pos in range(1, n_new):
        two_sample_ok(a[:, pos], s[:, pos], t.vocab_size, f"async: marginal of generated position {pos}")


def test_async_sampler_x_runs_on_the_hip_path(gpu):
    """sampler_x end to end on the GPU: top-(F+1) boost rows for the tree sampler and for q in the ratio test."""
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, factory = perturbed_pair(0.03)
    kw = dict(KW, num_draft_kvcache_blocks=256)
    sd = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=t, speculate=True, speculate_k=3, max_num_seqs=4,
                   draft_async=True, async_fan_out=3, jit_speculate=True, sampler_x=0.5, inprocess_draft=True,
                   runner_factory=factory, **kw)
    out, m = sd.generate([PROMPT] * 4, SamplingParams(temperature=0.8, max_new_tokens=16, ignore_eos=True), use_tqdm=False)
    assert all(len(o["token_ids"]) == 16 for o in out)
    assert all(1 <= n <= 4 for n in m["accepted_suffix_lens_with_recovery"])
    assert sd.draft_server.stats["hits"] > 0
