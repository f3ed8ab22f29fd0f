"""GPU replays of the goldens traced from the reference's own engine / DraftRunner code (tests/test_ref_engine_golden.py and
tests/test_draft_rounds_golden.py run them on the oracle backend): the HIP engine against what the reference's Scheduler /
Step / Speculator / Verifier / ModelRunner / DraftRunner produced (reference ssd/engine/step.py:91-163,
ssd/engine/draft_runner.py:186-378).  Validated on an MI355X in round 3 (profiles/r03_replays_gpu.txt)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu]

from tests.util import assert_stream_matches


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


from tests.test_ref_engine_golden import SCENARIOS, sampling, scenario_setup


@pytest.mark.parametrize("name", SCENARIOS)
def test_hip_engine_vs_reference_engine_run(gpu, golden, name):
    """The HIP engine against the completions of the reference's own engine classes, with the scenario's own requests (two, or
    three through two batch slots; per-request temperatures and length budgets).  A greedy request must reproduce the
    reference's stream to the end; a divergence is only legitimate where the reference's own top-2 margin of that decision is
    a near-tie.  A request sampled at temperature > 0 cannot be compared token by token (the device RNG is not torch's Philox
    stream; tests/test_hip_stochastic.py and tests/test_engine_temperature_gpu.py hold the distributional checks): it must
    run through the same engine path and deliver the same number of valid tokens -- and in the mixed batches the GREEDY
    neighbour of a sampling request is still held to the reference's stream."""
    from ssd_amd.engine.llm_engine import LLMEngine
    from tests.test_model_gpu import hip_factory
    g = golden("ref_engine")
    tw, dw, kw, new_tokens = scenario_setup(g, name)
    eng = LLMEngine("t", runner_factory=hip_factory(tw, dw), **kw)
    nreq = int(g[name + "/nreq"][0])
    prompts = [g[f"prompt{i}"].tolist() for i in range(nreq)]
    sps = sampling(g, name, kw, new_tokens)
    out, m = eng.generate(prompts, sps, use_tqdm=False)
    vocab = kw["hf_config"].vocab_size
    for b in range(nreq):
        want = g[f"{name}/completion{b}"].tolist()
        got = out[b]["token_ids"]
        if sps[b].temperature > 0:
            assert all(0 <= t < vocab for t in got)
            if kw["eos"] < 0:
                assert len(got) == len(want), f"{name} seq {b}: sampled request delivered {len(got)} tokens, reference {len(want)}"
            continue
        margins = {len(prompts[b]) + i: float(v) for i, v in enumerate(g[f"{name}/margins{b}"].tolist())}
        assert_stream_matches(got, want, margins, len(prompts[b]), f"{name} seq {b}")
    eng.exit()


@pytest.mark.parametrize("name,eagle", [("draft_rounds_llama", False), ("draft_rounds_eagle3", True)])
def test_hip_draft_server_replays_the_reference_runner_rounds(gpu, golden, name, eagle):
    """tests/test_draft_rounds_golden.py with the real runners (hipGraph JIT chain and tree, glue, device-side fork)."""
    from ssd_amd.engine import async_proto as P
    from ssd_amd.engine.draft_runner import DraftServer
    from ssd_amd.engine.llm_engine import hip_runner_factory
    from ssd_amd.utils.topology import Topology
    from tests.test_draft_rounds_golden import setup
    g = golden(name)
    cfg, _, _, K, F = setup(g, eagle)
    MQ = F * (K + 1)
    dw = {k[2:]: v for k, v in g.items() if k.startswith("d.")}
    runner = hip_runner_factory(cfg, cfg.draft_hf_config, is_draft=True, topo=Topology(0, 1, gpu, "draft", 0, 1),
                                weight_source=iter(dw.items()), num_kvcache_blocks=40)
    tx, server_end = P.LoopbackTransport.pair()
    server = DraftServer(cfg, runner, server_end)
    tables = g["draft_block_tables"].tolist()
    prompts = [g["prompt0"].tolist(), g["prompt1"].tolist()]
    toks = [p[1:] for p in prompts] if eagle else prompts
    payload = P.pack_prefill(toks, tables, cfg.max_blocks)
    tx.send_ints([P.CMD_PREFILL, 2, len(payload), P.FLAG_EAGLE if eagle else 0])
    tx.send_ints(payload)
    if eagle:
        tx.send_tensor(g["prefill_acts"].cuda())
    assert server.handle_one()
    alive, tainted, excused, compared = [True, True], [False, False], [], [0]
    for rnd in range(3):
        keys = [tuple(r) for r in g[f"r{rnd}_keys"].tolist()]
        nts = g[f"r{rnd}_num_tokens"].tolist()
        if eagle:
            payload = P.pack_speculate(keys, nts, tables, [0.0, 0.0], cfg.max_blocks, g[f"r{rnd}_ext_counts"].tolist(), g[f"r{rnd}_ext_ids"].tolist())
            acts = torch.cat([g[f"r{rnd}_ext_acts"], g[f"r{rnd}_rec_acts"].unsqueeze(1)], dim=1).cuda()
        else:
            payload = P.pack_speculate(keys, nts, tables, [0.0, 0.0], cfg.max_blocks)
        tx.send_ints([P.CMD_SPECULATE, 2, len(payload), P.FLAG_EAGLE if eagle else 0])
        tx.send_ints(payload)
        if eagle:
            tx.send_tensor(acts)
        assert server.handle_one()
        resp = tx.recv_tensor((2 + 2 * K,), torch.int64).tolist()
        forks, cached = server.pending_forks.cpu().tolist(), server.cache_tokens.cpu().tolist()
        if eagle:       # strict; prenorm vectors: two ulps of each row's magnitude (see tests/test_eagle_gpu.py taps_close)
            assert resp[:2] == g[f"r{rnd}_hits"].tolist(), f"round {rnd}: hits"
            assert resp[2:] == g[f"r{rnd}_tokens"].reshape(-1).tolist(), f"round {rnd}: replied tokens"
            assert forks == g[f"r{rnd}_forks"].tolist(), f"round {rnd}: forks"
            assert cached == g[f"r{rnd}_cache_tokens"].tolist(), f"round {rnd}: branch continuations"
            got = server.cache_acts.reshape(2 * MQ * K, -1).float().cpu()
            want = g[f"r{rnd}_cache_acts"].reshape(2 * MQ * K, -1).float()
            bound = 0.02 + want.abs().amax(dim=-1, keepdim=True) / 64.0
            assert bool(((got - want).abs() <= bound).all()), f"round {rnd}: branch prenorms"
            continue
        # Plain draft: every decision is compared; a difference is admissible only where the REFERENCE's own rank gap of that
        # decision (recorded by make_golden.py gen_draft_rounds, in bf16 ulps of the row's top logit) is a near-tie, and what
        # depends on a flipped decision is no longer compared (its sequence for the rest of the run / its branch this round).
        NEAR = 2.0
        jit_gap, glue_gap, tree_gap = g[f"r{rnd}_jit_gap2"], g[f"r{rnd}_glue_gapF"], g[f"r{rnd}_tree_gap2"]
        want_tok, want_forks, want_cached = g[f"r{rnd}_tokens"].tolist(), g[f"r{rnd}_forks"].tolist(), g[f"r{rnd}_cache_tokens"].tolist()
        for b in range(2):
            if not alive[b]:
                continue
            got_tok = resp[2 + b * K:2 + (b + 1) * K]
            if tainted[b]:      # last round left one differing branch: if THIS request named it, the sequence is no longer comparable
                tainted[b] = False
                if resp[b] != int(g[f"r{rnd}_hits"][b]) or got_tok != want_tok[b]:
                    alive[b] = False
                    continue
            assert resp[b] == int(g[f"r{rnd}_hits"][b]), f"round {rnd} seq {b}: hit flag"
            for j in range(K):
                compared[0] += 1
                if got_tok[j] != want_tok[b][j]:
                    # a hit replays a branch that was compared last round; only a JIT chain takes fresh decisions here
                    assert jit_gap.shape[0] > j and float(jit_gap[j, b]) <= NEAR, f"round {rnd} seq {b}: replied token {j}"
                    excused.append((rnd, b, "reply", j))
                    alive[b] = False
                    break
            if not alive[b]:
                continue
            for j in range(K + 1):
                branches = range(j * F, (j + 1) * F)
                ok = [forks[b][i] == want_forks[b][i] for i in branches]
                compared[0] += F
                if not all(ok):
                    assert float(glue_gap[b, j]) <= NEAR, f"round {rnd} seq {b}: forks of glue row {j}: {forks[b]} vs {want_forks[b]}"
                    excused.append((rnd, b, "fork", j))
                    tainted[b] = True            # the next request may name a fork this server does not hold
                for i, same in zip(branches, ok):
                    if not same:
                        continue
                    for d in range(K):
                        compared[0] += 1
                        if cached[b * MQ + i][d] != want_cached[b * MQ + i][d]:
                            assert float(tree_gap[d, b * MQ + i]) <= NEAR, f"round {rnd} seq {b} branch {i}: continuation token {d}"
                            excused.append((rnd, b, "tree", i, d))
                            tainted[b] = True
                            break
    if not eagle:
        # the fixture holds ~110 decisions, 19 of them within two ulps: a handful may flip, most must not
        assert compared[0] >= 60 and len(excused) <= 4, f"compared {compared[0]} decisions, excused {excused}"


def _full_size_pair(layer_gain: float = 0.05, target: str = "llama-3.1-8b", target_layers: int | None = None):
    """Llama-3.1-8B shapes (32 layers, h 4096, V 128256) + the full Llama-3.2-1B draft (16 layers): the correlated synthetic
    pair (ssd_amd/weights.py _pair_tensor), generated on the GPU (18.5 GB of bf16) and copied to the host once for the oracle.
    target / target_layers: another preset, cut to that many layers (the 70B geometry of the metric's own configuration)."""
    import dataclasses
    from ssd_amd import weights as W
    from ssd_amd.model_config import PRESETS
    tcfg = PRESETS[target]
    if target_layers is not None:
        tcfg = dataclasses.replace(tcfg, num_layers=target_layers)
    dcfg = dataclasses.replace(PRESETS["llama-3.2-1b"], tie_word_embeddings=False)     # the pair recipe unties the 1B head
    recipe = {"kind": "pair", "shared": dcfg.hidden_size, "snr": 8.0, "layer_gain": layer_gain}
    wt = {n: t.cpu() for n, t in W.synthetic_weights(tcfg, 0, 0.02, gen_device="cuda", recipe=recipe)}
    wd = {n: t.cpu() for n, t in W.synthetic_weights(dcfg, 1, 0.02, gen_device="cuda", recipe=recipe)}
    return tcfg, dcfg, wt, wd


def _lockstep_full_size(mode: str, n_new: int, layer_gain: float = 0.05, thr: float | None = None, min_rounds: float = 0.85,
                        min_tokens: float = 0.9, max_restarts: int = 16, target: str = "llama-3.1-8b", target_layers: int | None = None):
    """Product engine on the MI355X against the oracle engine on the host, round by round (tests/lockstep.py): hit flags,
    speculated tokens, accepted suffixes; an excused near-tie re-synchronises both runs on the oracle's tokens."""
    import random
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine, hip_runner_factory
    from ssd_amd.sampling_params import SamplingParams
    from tests.lockstep import compare_lockstep
    from tests.lockstep import NEAR_TIE
    tcfg, dcfg, wt, wd = _full_size_pair(layer_gain, target, target_layers)
    label = target + (f" x {target_layers} layers" if target_layers else "") + " + 1B"
    random.seed(5)
    prompt = [random.randint(0, 10000) for _ in range(96)]
    kw = dict(hf_config=tcfg, max_num_seqs=1, max_model_len=1024, max_num_batched_tokens=1024, kvcache_block_size=256,
              num_kvcache_blocks=6, num_draft_kvcache_blocks=6, draft="d", draft_hf_config=dcfg, speculate=True)
    if mode == "async":         # the metric's own mode (BASELINE.json: async SSD k = 7, f = 3, jit backup), draft co-located
        kw.update(speculate_k=7, draft_async=True, async_fan_out=3, jit_speculate=True)
    else:
        kw.update(speculate_k=6)

    def hipf(config, model_cfg, *, is_draft, topo, **k2):
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, weight_source=iter((wd if is_draft else wt).items()), **k2)

    gpu_eng = LLMEngine("t", runner_factory=hipf, inprocess_draft=mode == "async", **kw)
    from ssd_amd.utils.topology import Topology
    cpu_eng = LLMEngine("t", runner_factory=oracle_runner_factory(wt, wd), inprocess_draft=mode == "async",
                        topology=Topology(0, 1, torch.device("cpu"), "target", 0, 1), **kw)      # (the oracle's tensors live on the host)
    rep = compare_lockstep(gpu_eng, cpu_eng, prompt, n_new, lambda n: SamplingParams(temperature=0, max_new_tokens=n, ignore_eos=True),
                           fan_out=3 if mode == "async" else None, thr=NEAR_TIE if thr is None else thr, max_restarts=max_restarts,
                           what=f"{label} {mode} layer_gain {layer_gain}")
    gpu_eng.exit()
    print(f"full size {label} {mode} (layer_gain {layer_gain}): {rep.summary()}")
    assert rep.tokens == n_new
    assert rep.tokens_compared >= min_tokens * rep.tokens, rep.summary()  # only the disputed near-tie tokens themselves go uncompared
    assert rep.rounds_compared >= min_rounds * rep.rounds, rep.summary()  # (round 4 accepted 0.6; measured: sync 27 / 32, async 33 / 37)
    return rep


def test_full_size_configs1_llama8b_target_1b_draft_sync_k6(gpu):
    """BASELINE.json configs[1] at FULL size -- synchronous speculation k = 6, b = 1, temperature 0, KV block 256, hipGraphs
    (reference ssd/engine/step.py:91-163, speculator_sync.py:25-69 driven by bench/bench.py:34-51's configuration).  56 new
    tokens.  Every round: the K speculated tokens and the accepted suffix of the product engine equal the oracle engine's; a
    difference is admissible only where the ORACLE's own margin at that very decision is a near-tie, and then both runs are
    re-synchronised on the oracle's token and the comparison goes on to the end (round 3 stopped at the first near-tie, 16 of
    40 tokens, and its accepted-length comparison read one shared METRICS dict twice)."""
    rep = _lockstep_full_size("sync", 56, min_rounds=0.8)
    assert max(rep.accepted_lens) > 1 and min(rep.accepted_lens) < 7, "the pair should produce both accepts and rejections"


def test_full_size_async_k7_f3_llama8b_target_1b_draft(gpu):
    """The metric's own mode at real shapes against the oracle: asynchronous speculation k = 7, f = 3 (MQ_LEN 24), jit backup,
    co-located draft server, Llama-3.1-8B shapes + the full 1B draft, 64 new tokens (reference ssd/engine/step.py:91-163,
    draft_runner.py:186-378: cache lookup, JIT chain on a miss, glue + fork top-F at V = 128256, 7 tree steps of 24
    branches).  Hits, misses, partial and full acceptances must all occur; hit flags, replied tokens and accepted suffixes
    are compared round by round under the near-tie + re-synchronisation rule."""
    rep = _lockstep_full_size("async", 64)
    assert rep.hits > 0 and rep.real_misses > 0, rep.summary()        # misses beyond each run's first request: JIT chains on real misses
    assert rep.partial_accepts > 0 and max(rep.accepted_lens) > 2, rep.summary()


def test_full_size_async_with_the_resident_segments_beside_the_verify(gpu, monkeypatch):
    """The M-row resident segment at engine level and under the load it is hardest to get right under: SSD_TREE_SEG=1 forces it for
    the glue decode AND the 24-row tree steps of the co-located draft server, whose round runs on its own stream WHILE the target's
    verify streams 16 GB next to it -- the 256 workgroups come up one by one, spin on each other's flags and read each other's rows
    beside foreign traffic (the uneven-load case a hand-off protocol must survive: a stale row would flip a token).  Same lock-step
    comparison with the oracle engine as above: hit flags, replied tokens, accepted suffixes, near-ties only by the oracle's margin."""
    monkeypatch.setenv("SSD_TREE_SEG", "1")
    rep = _lockstep_full_size("async", 48, min_rounds=0.8)
    assert rep.hits > 0 and rep.partial_accepts > 0, rep.summary()


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_full_size_lockstep_with_undamped_layers(gpu, mode):
    """The same two lock-step runs with layer_gain 1.0: o_proj / down_proj at their plain N(0, 0.02) scale, so
    every attention / MLP kernel's error reaches the token decisions unattenuated (at 0.05 it arrives 20x smaller: those runs pin the
    protocol, these pin the layer kernels through it).  The decoder layers now drown the shared embedding: the pair agrees rarely,
    nearly every async round is a miss -> JIT chain + glue + fork at V = 128256 + 7 tree steps of 24 branches, all with undamped
    layers, compared decision by decision with the oracle.  Two bf16 pipelines of 32 (target) / 16 (draft) undamped layers differ by up
    to ~0.1-0.2 on a logit (measured: tests/test_hip_tree_segment.py 16 layers 0.10, test_eight_layer_70b_cut... 8 layers 0.19), yet the
    near-tie threshold stays at TWO bf16 ulps of the ORACLE's own margin (0.125): an undamped random model's logits are nearly flat,
    so near-ties are frequent BY CONSTRUCTION (first runs on MI355X, profiles/r05_lockstep_call8.txt: 10-11 of 24 rounds disputed, every
    one of them at an oracle margin of 0 - 0.0625) -- hence the lower bar on compared rounds -- while every decision with a real margin
    must match; an excused decision re-synchronises both runs (teacher forcing), so all 24 tokens are reached."""
    rep = _lockstep_full_size(mode, 24, layer_gain=1.0, thr=0.125, min_rounds=0.45, min_tokens=0.75, max_restarts=24)
    if mode == "async":
        assert rep.real_misses > 0, rep.summary()


def test_c4_shaped_async_lockstep_70b_geometry_cut_with_undamped_layers(gpu):
    """The METRIC's own configuration in shape (BASELINE.json configs[3] / bench.py c4): a target with Llama-3.1-70B's geometry (h 8192,
    64 / 8 heads, I 28672, V 128256 -- the launch shapes, kernels and dispatch choices of the benchmarked verify; cut to 8 layers so the
    oracle engine fits the host and the test stays under a minute) + the full 16-layer Llama-3.2-1B draft, asynchronous speculation
    k = 7, f = 3, jit backup, co-located draft server, hipGraphs, at layer_gain 1.0 (nothing damped: every attention / MLP kernel's error
    reaches the token decisions) -- lock-step against the oracle engine on the host.  Same criteria as the undamped
    8B runs above: every decision with a real oracle margin must match, near-ties (<= two bf16 ulps of the ORACLE's own margin) are
    excused and re-synchronised, all tokens are reached.  Reference: ssd/engine/step.py:91-163, speculator_async.py:92-187,
    draft_runner.py:186-378,713-812, utils/verify.py:28-48."""
    rep = _lockstep_full_size("async", 24, layer_gain=1.0, thr=0.125, min_rounds=0.45, min_tokens=0.75, max_restarts=24,
                              target="llama-3.1-70b", target_layers=8)
    assert rep.real_misses > 0, rep.summary()


def test_full_size_eagle3_llama8b_lockstep(gpu):
    """EAGLE-3 at real shapes with a pair that AGREES (VERDICT r3 missing #4), against the oracle: Llama-3.1-8B shapes as target +
    the eagle3-llama-3.1-8b draft (one layer, h 4096, 32000-token head, 3 tapped activations), the constructed pair of
    ssd_amd/weights.py eagle_pair_recipe (bench.py --workload c4e's recipe), asynchronous k = 7 f = 3, jit backup, co-located
    draft, 48 tokens -- lock-step (tests/lockstep.py): hit flags, replied tokens (cache hits carry the cached prenorm vectors
    into the next glue), extend rows, accepted suffixes.  Reference: ssd/engine/draft_runner.py:186-378 use_eagle branches,
    ssd/models/eagle3_draft_llama3.py, ssd/engine/helpers/cudagraph_helpers.py:636-774."""
    import dataclasses
    import random
    from oracle.runner import oracle_runner_factory
    from ssd_amd import weights as W
    from ssd_amd.engine.llm_engine import LLMEngine, hip_runner_factory
    from ssd_amd.model_config import PRESETS
    from ssd_amd.sampling_params import SamplingParams
    from ssd_amd.utils.topology import Topology
    from tests.lockstep import compare_lockstep
    tcfg = PRESETS["llama-3.1-8b"]
    dcfg = dataclasses.replace(PRESETS["eagle3-llama-3.1-8b"], d_model_target=tcfg.hidden_size)
    rec = W.eagle_pair_recipe(tcfg, dcfg, draft_seed=1, snr=8.0)
    wt = {n: t.cpu() for n, t in W.synthetic_weights(tcfg, 0, 0.02, gen_device="cuda", recipe=rec)}
    wd = {n: t.cpu() for n, t in W.synthetic_weights(dcfg, 1, 0.02, gen_device="cuda", recipe=rec)}
    random.seed(7)
    prompt = [random.randint(0, 10000) for _ in range(64)]
    kw = dict(hf_config=tcfg, max_num_seqs=1, max_model_len=1024, max_num_batched_tokens=1024, kvcache_block_size=256,
              num_kvcache_blocks=6, num_draft_kvcache_blocks=6, draft="e", draft_hf_config=dcfg, speculate=True, speculate_k=7,
              draft_async=True, async_fan_out=3, jit_speculate=True, use_eagle=True)

    def hipf(config, model_cfg, *, is_draft, topo, **k2):
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, weight_source=iter((wd if is_draft else wt).items()), **k2)

    gpu_eng = LLMEngine("t", runner_factory=hipf, inprocess_draft=True, **kw)
    cpu_eng = LLMEngine("t", runner_factory=oracle_runner_factory(wt, wd), inprocess_draft=True,
                        topology=Topology(0, 1, torch.device("cpu"), "target", 0, 1), **kw)
    n_new = 48
    rep = compare_lockstep(gpu_eng, cpu_eng, prompt, n_new, lambda n: SamplingParams(temperature=0, max_new_tokens=n, ignore_eos=True),
                           fan_out=3, what="8B + EAGLE-3")
    gpu_eng.exit()
    print(f"full size 8B + EAGLE-3 (constructed pair): {rep.summary()}")
    assert rep.tokens == n_new and rep.tokens_compared >= 0.9 * rep.tokens, rep.summary()
    assert rep.rounds_compared >= 0.6 * rep.rounds, rep.summary()
    assert rep.hits > 0 and max(rep.accepted_lens) > 2, rep.summary()
