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


def test_full_size_configs1_llama8b_target_1b_draft_sync_k6(gpu):
    """BASELINE.json configs[1] at FULL size -- Llama-3.1-8B shapes (32 layers, h 4096, V 128256) + the full Llama-3.2-1B
    draft (16 layers), synchronous speculation k = 6, b = 1, temperature 0, KV block 256, hipGraphs -- the product engine on
    the MI355X against the oracle engine on the host (reference ssd/engine/step.py:91-163 driven by
    bench/bench.py:34-51's configuration).  The pair is the correlated synthetic one (ssd_amd/weights.py _pair_tensor), so
    rounds end in rejections, partial and full acceptances.  40 new tokens; the streams must be identical to the end unless
    the ORACLE's own top-2 margin at the first differing decision is a near-tie (tests/util.py assert_stream_matches), and
    the accepted-suffix lengths must agree step by step over the common prefix.  Weights are generated on the GPU (18.5 GB
    of bf16) and copied to the host once for the oracle."""
    import dataclasses
    import random
    from oracle.runner import oracle_runner_factory
    from ssd_amd import weights as W
    from ssd_amd.engine.llm_engine import LLMEngine, hip_runner_factory
    from ssd_amd.model_config import PRESETS
    from ssd_amd.sampling_params import SamplingParams
    from tests.util import seq_margins
    tcfg = PRESETS["llama-3.1-8b"]
    dcfg = dataclasses.replace(PRESETS["llama-3.2-1b"], tie_word_embeddings=False)     # the pair recipe unties the 1B head
    recipe = {"kind": "pair", "shared": dcfg.hidden_size, "snr": 8.0, "layer_gain": 0.05}
    wt = {n: t.cpu() for n, t in W.synthetic_weights(tcfg, 0, 0.02, gen_device="cuda", recipe=recipe)}
    wd = {n: t.cpu() for n, t in W.synthetic_weights(dcfg, 1, 0.02, gen_device="cuda", recipe=recipe)}
    random.seed(5)
    prompt = [random.randint(0, 10000) for _ in range(96)]
    n_new = 40
    sp = SamplingParams(temperature=0, max_new_tokens=n_new, ignore_eos=True)
    kw = dict(hf_config=tcfg, max_num_seqs=1, max_model_len=1024, max_num_batched_tokens=1024, kvcache_block_size=256,
              num_kvcache_blocks=6, num_draft_kvcache_blocks=6, draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=6)

    def hipf(config, model_cfg, *, is_draft, topo, **k2):
        return hip_runner_factory(config, model_cfg, is_draft=is_draft, topo=topo, weight_source=iter((wd if is_draft else wt).items()), **k2)

    gpu_eng = LLMEngine("t", runner_factory=hipf, **kw)
    got, gm = gpu_eng.generate([prompt], sp, use_tqdm=False)
    gpu_eng.exit()
    del gpu_eng
    torch.cuda.empty_cache()
    cpu_eng = LLMEngine("t", runner_factory=oracle_runner_factory(wt, wd), **kw)
    want, cm = cpu_eng.generate([prompt], sp, use_tqdm=False)
    n = assert_stream_matches(got[0]["token_ids"], want[0]["token_ids"], seq_margins(cpu_eng.model_runner.margin_log, 0),
                              len(prompt), what="configs[1] 8B+1B sync k=6")
    gl, cl = list(gm["accepted_suffix_lens_with_recovery"]), list(cm["accepted_suffix_lens_with_recovery"])
    print(f"configs[1] full size: {n}/{n_new} tokens identical to the oracle engine; accepted lens gpu {gl} cpu {cl}")
    assert max(cl) > 1 and min(cl) < 7, "the pair should produce both accepts and rejections"
    # steps wholly inside the common prefix took the same accept / reject decisions
    done, k = 0, 0
    while k < min(len(gl), len(cl)) and done + cl[k] <= n:
        assert gl[k] == cl[k], f"step {k}: accepted {gl[k]} tokens, the oracle engine {cl[k]}"
        done += cl[k]
        k += 1
    assert k >= 3
