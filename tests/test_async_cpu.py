"""Asynchronous speculation (SSD) on CPU: the real engine + draft server + wire protocol, with oracle runners.
(1) in-process loopback, (2) two gloo processes (target rank 0, draft rank 1) as deployed.
Exactness: the async stream equals plain autoregressive decoding; with draft == target every verification is
fully accepted and every request after the first is a speculation-cache hit."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfgs():
    from ssd_amd.model_config import ModelConfig
    t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 1024, False)
    d = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 1024, True)
    return t, d


KW = dict(max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=32, num_kvcache_blocks=48, weights_std=0.1)
PROMPTS = [[(5 * i + 3 * j) % 256 for j in range(6 + 2 * i)] for i in range(2)]


def run(mode, same=False, bs=1, jit=True, fan=None, fan_miss=None, temperature=0.0, draft_temperature=None, **extra):
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = cfgs()
    kw = dict(KW, max_num_seqs=bs)
    kw.update(extra)
    if mode != "ar":
        kw.update(draft="d", draft_hf_config=t if same else d, speculate=True, speculate_k=3)
        if same:
            kw.update(draft_weights_seed=0)     # identical synthetic weights
    if mode == "async":
        kw.update(draft_async=True, async_fan_out=2, jit_speculate=jit, fan_out_list=fan, fan_out_list_miss=fan_miss)
    eng = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), inprocess_draft=(mode == "async" and "num_gpus" not in extra), **kw)
    out, m = eng.generate(PROMPTS[:max(1, bs)], SamplingParams(temperature=temperature, draft_temperature=draft_temperature,
                                                               max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    stats = eng.draft_server.stats if eng.draft_server is not None else None
    eng.exit()
    return [o["token_ids"] for o in out], m, stats


def test_async_loopback_is_exact():
    ar, _, _ = run("ar")
    sync, _, _ = run("sync")
    asy, m, stats = run("async")
    assert ar == sync == asy
    assert stats["requests"] == len(m["accepted_suffix_lens_with_recovery"])
    assert len(m["cache_hits"]) == stats["rounds"]


def test_deferred_draft_prefill_is_served_before_the_first_speculation_request(monkeypatch):
    """Co-located draft on one GPU: the draft's prefill command stays queued until the first speculation request's receive pumps
    it (engine/llm_engine.py, TTFT) -- commands are served in order, so streams, hits and acceptance are those of the eager
    order (batch of two requests)."""
    want, m0, s0 = run("async", same=True, bs=2)
    monkeypatch.setenv("SSD_DEFER_DRAFT_PREFILL", "1")
    got, m1, s1 = run("async", same=True, bs=2)
    assert got == want and m1["cache_hits"] == m0["cache_hits"]
    assert m1["accepted_suffix_lens_with_recovery"] == m0["accepted_suffix_lens_with_recovery"] and s1 == s0


def test_async_same_model_hits_and_accepts_everything():
    ar, _, _ = run("ar")
    asy, m, stats = run("async", same=True)
    assert asy == ar
    lens = m["accepted_suffix_lens_with_recovery"]
    assert all(n == 4 for n in lens[:-1])                 # K+1 tokens per step (last one may be clipped by max_new_tokens)
    assert m["cache_hits"][0] == 0.0 and all(h == 1.0 for h in m["cache_hits"][1:])
    assert stats["hits"] == stats["requests"] - 1


def test_async_batch_nonuniform_fanout_and_fast_backup():
    ar, _, _ = run("ar", bs=2)
    a1, _, _ = run("async", bs=2, fan=[1, 2, 2, 3], fan_miss=[3, 2, 2, 1])
    a2, _, _ = run("async", bs=2, jit=False)                # "fast" backup: misses carry filler tokens, still exact
    a3, _, s3 = run("async", bs=2, same=True, fan=[1, 1, 1, 5], fan_miss=[2, 2, 2, 2])
    assert ar == a1 == a2 == a3
    assert s3["hits"] > 0


def test_async_temperature_same_model_accepts_everything():
    """temperature > 0 over the async protocol (FLAG_WANT_LOGITS): the server samples the JIT chain and every tree
    branch and ships the q logits of the answered branch; with draft == target p == q, so min(1, p/q) = 1 and every
    round after the first (a JIT-served miss) is a cache hit that is fully accepted."""
    torch.manual_seed(0)
    asy, m, stats = run("async", same=True, temperature=0.8)
    lens = m["accepted_suffix_lens_with_recovery"]
    assert all(n == 4 for n in lens[:-1]), lens
    assert all(len(t) == 14 for t in asy)
    # the bonus token of an all-accepted round is a fresh draw from p: whether it is one of the F forked tokens is up
    # to chance, so hits are not guaranteed -- but a hit must have been answered from the sampled tree's logits
    assert stats["rounds"] == len(m["cache_hits"])


def test_async_temperature_greedy_draft_and_mixed_batch():
    torch.manual_seed(1)
    ar, _, _ = run("ar")
    # a greedy draft (draft_temperature = 0) under a sampling target: q is one-hot, the reply still carries logits
    out, m, _ = run("async", bs=2, temperature=0.7, draft_temperature=0.0)
    assert all(len(t) == 14 for t in out)
    assert all(1 <= n <= 4 for n in m["accepted_suffix_lens_with_recovery"])


def test_async_temperature_sampler_x_runs_end_to_end():
    """sampler_x (reference Sampler(is_tree=True) + verify(sampler_x=...)): the tree sampler and the verifier's q are
    rescaled on their top F+1 entries; bit-level semantics are pinned against the reference in test_oracle_golden."""
    torch.manual_seed(2)
    out, m, stats = run("async", same=True, temperature=0.9, sampler_x=0.5)
    assert all(len(t) == 14 for t in out)
    assert all(1 <= n <= 4 for n in m["accepted_suffix_lens_with_recovery"])
    assert stats["rounds"] >= 3


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    toks, m, _ = run("async", same=True, num_gpus=2)
    q.put((rank, toks, m["cache_hits"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_async_two_processes_gloo():
    ar, _, _ = run("ar")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    ps = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, toks, hits = q.get(timeout=300)
        got[r] = (toks, hits)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == ar                 # target rank produced the exact stream
    assert got[1][0] == []                 # draft rank only served
    assert all(h == 1.0 for h in got[0][1][1:])


def _worker_colocated(rank, port, q):
    """TP = 2 target with the draft server co-located on TP rank 0 (bench.py's default placement at N > 1)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = cfgs()
    eng = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), inprocess_draft=True, num_gpus=2, draft="d",
                    draft_hf_config=t, draft_weights_seed=0, speculate=True, speculate_k=3, draft_async=True, async_fan_out=2,
                    jit_speculate=True, **KW)
    assert eng.topo.tp_size == 2 and (eng.draft_server is not None) == (rank == 0)
    out, m = eng.generate(PROMPTS[:1], SamplingParams(temperature=0, max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    q.put((rank, [o["token_ids"] for o in out], m["cache_hits"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_async_colocated_draft_under_tensor_parallelism_gloo():
    ar, _, _ = run("ar")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 90)
    ps = [ctx.Process(target=_worker_colocated, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, toks, hits = q.get(timeout=300)
        got[r] = (toks, hits)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == ar and got[1][0] == ar          # SPMD: both TP ranks produce the exact stream
    assert got[0][1] == got[1][1] and all(h == 1.0 for h in got[0][1][1:])   # hits are broadcast to the non-head rank


def test_correlated_pair_recipe_gives_partial_acceptance():
    """weights._pair_tensor: a target and a NARROWER draft that agree on most greedy tokens by construction -- what
    bench.py uses so that acceptance, cache hits and the all-accepted path run at realistic rates without checkpoints.
    Speculation stays exact whatever the acceptance."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    t = ModelConfig("llama", 128, 2, 4, 2, 32, 256, 2048, 1e-5, 5e5, 1024, False)
    d = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 2048, 1e-5, 5e5, 1024, False)
    kw = dict(KW, hf_config=t, weights_recipe={"kind": "pair", "shared": 64, "snr": 8.0, "layer_gain": 0.05})
    sp = SamplingParams(temperature=0, max_new_tokens=48, ignore_eos=True)
    prompt = [[3, 1, 4, 1, 5, 9, 2, 6]]
    ar, _ = LLMEngine("t", runner_factory=oracle_runner_factory(), **kw).generate(prompt, sp, use_tqdm=False)
    sd, m = LLMEngine("t", runner_factory=oracle_runner_factory(), draft="d", draft_hf_config=d, speculate=True, speculate_k=4,
                      **kw).generate(prompt, sp, use_tqdm=False)
    asy, m2 = LLMEngine("t", runner_factory=oracle_runner_factory(), draft="d", draft_hf_config=d, speculate=True, speculate_k=4,
                        draft_async=True, async_fan_out=2, jit_speculate=True, inprocess_draft=True, **kw).generate(prompt, sp, use_tqdm=False)
    assert ar[0]["token_ids"] == sd[0]["token_ids"] == asy[0]["token_ids"]
    lens = m["accepted_suffix_lens_with_recovery"]
    assert 1.5 < sum(lens) / len(lens) < 5.0, lens                  # neither ~1.0 (random pair) nor always K+1 (draft == target)
    assert 0.0 < sum(m2["cache_hits"]) / len(m2["cache_hits"]) <= 1.0


def _worker_draft_dp(rank, world, port, q, temperature, extra=None):
    """1 target rank + (world - 1) draft ranks (draft data-parallel): branches and speculation cache sharded."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    torch.manual_seed(0)
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = cfgs()
    eng = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), num_gpus=world, num_draft_gpus=world - 1, draft="d",
                    draft_hf_config=t, draft_weights_seed=0, speculate=True, speculate_k=3, draft_async=True, async_fan_out=2,
                    max_num_seqs=2, **dict(dict(jit_speculate=True), **(extra or {})), **KW)
    out, m = eng.generate(PROMPTS, SamplingParams(temperature=temperature, max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    stats = eng.draft_server.stats if eng.draft_server is not None else None
    eng.exit()
    q.put((rank, [o["token_ids"] for o in out], m["cache_hits"], m["accepted_suffix_lens_with_recovery"], stats))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _run_draft_dp(world, temperature=0.0, extra=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90) + world + (7 if extra else 0)
    ps = [ctx.Process(target=_worker_draft_dp, args=(r, world, port, q, temperature, extra)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, toks, hits, lens, stats = q.get(timeout=300)
        got[r] = (toks, hits, lens, stats)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_draft_data_parallel_shards_the_tree_and_stays_exact():
    """BASELINE.json configs[4] shape (target + draft x D data-parallel) with D = 2 and D = 3 on gloo: same stream as
    autoregressive decoding; with draft == target every request after the first hits -- whichever member owns the branch --
    and everything is accepted, exactly like the single-draft run."""
    ar, _, _ = run("ar", bs=2)
    single, m1, _ = run("async", bs=2, same=True)
    for world in (3, 4):
        got = _run_draft_dp(world)
        assert got[0][0] == ar == single
        assert got[0][1] == m1["cache_hits"] and got[0][2] == m1["accepted_suffix_lens_with_recovery"]
        rounds = [got[r][3]["rounds"] for r in range(1, world)]
        hits = [got[r][3]["hits"] for r in range(1, world)]
        assert len(set(rounds)) == 1 and len(set(hits)) == 1          # every member sees every request and the merged verdict
        assert all(got[r][0] == [] for r in range(1, world))


def test_draft_data_parallel_with_sampling():
    """temperature > 0 through the draft group: the owner's branch logits reach the target via the leader; with
    draft == target p == q so every drafted token is accepted."""
    got = _run_draft_dp(3, temperature=0.8)
    lens = got[0][2]
    assert all(len(t) == 14 for t in got[0][0])
    assert all(n == 4 for n in lens[:-2]), lens


def test_draft_data_parallel_uneven_shards_and_fast_backup():
    """Non-uniform fan-out lists whose branch count (9) does not divide over the 2 draft ranks (5 + 4 branches), and the
    "fast" backup (no JIT chain on a miss): still the exact autoregressive stream, same hits as a single draft rank."""
    fan = dict(fan_out_list=[1, 2, 2, 4], fan_out_list_miss=[3, 2, 2, 2])
    ar, _, _ = run("ar", bs=2)
    for extra in (fan, dict(fan, jit_speculate=False)):
        single, m1, _ = run("async", bs=2, same=True, fan=fan["fan_out_list"], fan_miss=fan["fan_out_list_miss"],
                            jit=extra.get("jit_speculate", True))
        got = _run_draft_dp(3, extra=extra)
        assert got[0][0] == ar == single
        assert got[0][1] == m1["cache_hits"], (got[0][1], m1["cache_hits"])


def test_profile_flags_print_the_reference_trace_lines(capfd, monkeypatch):
    """SSD_PROFILE / SSD_PROFILE_TARGET / SSD_PROFILE_DRAFT (reference step.py:92-161, verifier.py:63-74,
    draft_runner.py:880-915): same switches, same line prefixes; the stream is unchanged by tracing."""
    ar, _, _ = run("ar")
    for flag in ("SSD_PROFILE", "SSD_PROFILE_TARGET", "SSD_PROFILE_DRAFT"):
        monkeypatch.setenv(flag, "1")
    asy, m, _ = run("async", same=True)
    out = capfd.readouterr().out
    assert asy == ar
    steps = len(m["accepted_suffix_lens_with_recovery"])
    assert out.count("[PROFILE target] handshake=") == steps and "hits=1/1 toks=4" in out
    assert out.count("[PROFILE verifier] target_call=") == steps
    assert out.count("[PROFILE draft] ") >= steps - 1 and "glue_fork=" in out and "tree[" in out


def _worker_tp_and_draft_dp(rank, world, ndraft, port, q):
    """The full-node shape of BASELINE.json configs[4]: a tensor-parallel target AND a data-parallel draft group."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    t = ModelConfig("llama", 128, 2, 8, 4, 16, 256, 256, 1e-5, 5e5, 1024, False)      # 4 kv heads: tensor-parallel up to 4
    eng = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), num_gpus=world, num_draft_gpus=ndraft, draft="d",
                    draft_hf_config=t, draft_weights_seed=0, speculate=True, speculate_k=3, draft_async=True, async_fan_out=2,
                    jit_speculate=True, max_num_seqs=2, **KW)
    tp = world - ndraft
    assert (eng.topo.role == "draft") == (rank >= tp)
    if rank < tp:
        assert eng.topo.tp_size == tp and eng.topo.tp_rank == rank
    out, m = eng.generate(PROMPTS, SamplingParams(temperature=0, max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    eng.exit()
    q.put((rank, [o["token_ids"] for o in out], m["cache_hits"], m["accepted_suffix_lens_with_recovery"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _run_tp_and_draft_dp(world, ndraft):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 90) + world
    ps = [ctx.Process(target=_worker_tp_and_draft_dp, args=(r, world, ndraft, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, toks, hits, lens = q.get(timeout=600)
        got[r] = (toks, hits, lens)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_tensor_parallel_target_with_draft_group_gloo():
    """world 4 = TP 2 target + 2 draft ranks (what `bench.py --gpus 8` launches with 4 + 4): every target rank produces the
    same stream, the draft ranks only serve, and with draft == target every request after the first is a hit."""
    world, ndraft = 4, 2
    got = _run_tp_and_draft_dp(world, ndraft)
    tp = world - ndraft
    assert all(got[r][0] == got[0][0] and len(got[r][0]) == 2 for r in range(tp))
    assert all(got[r][0] == [] for r in range(tp, world))
    assert all(got[r][1] == got[0][1] for r in range(tp))
    assert got[0][1][0] == 0.0 and all(h == 1.0 for h in got[0][1][1:])


import pytest


@pytest.mark.parametrize("world,ndraft", [(5, 1), (8, 4)])
def test_the_two_full_node_layouts_of_the_benchmark_gloo(world, ndraft):
    """The layouts `bench.py --gpus 5` and `--gpus 8` launch (BASELINE.json configs[3]: TP 4 + one draft GPU; configs[4]: TP 4 +
    draft x4 data-parallel), process for process, on gloo with oracle runners: every one of the four tensor-parallel target
    ranks produces the same stream -- the autoregressive one --, the draft ranks only serve, and with draft == target every
    request after the first is a speculation-cache hit whichever member of the draft group owns the branch.  (On the
    one-GPU test box these layouts cannot run on the device: four and more processes whose one-shot all-reduce kernels spin
    on each other's flags exceed what one GPU schedules concurrently -- the attempt cost the round a GPU box -- so the
    shared-GPU bench tests stop at three ranks, tests/test_bench_gpu.py.)"""
    got = _run_tp_and_draft_dp(world, ndraft)
    tp = world - ndraft
    assert tp == 4
    assert all(got[r][0] == got[0][0] and len(got[r][0]) == 2 for r in range(tp))
    assert all(got[r][0] == [] for r in range(tp, world))
    assert all(got[r][1] == got[0][1] and got[r][2] == got[0][2] for r in range(tp))
    # draft == target weights, but the target sums its row-parallel GEMMs in four shards: a near-tie may cost a hit or an
    # accepted token now and then -- most requests still hit
    hits = got[0][1]
    assert hits[0] == 0.0 and sum(hits[1:]) / max(1, len(hits) - 1) >= 0.5, hits
    # the same requests through ONE process: the same tokens up to such a near-tie
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    t = ModelConfig("llama", 128, 2, 8, 4, 16, 256, 256, 1e-5, 5e5, 1024, False)
    one, _ = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), max_num_seqs=2, **KW).generate(
        PROMPTS, SamplingParams(temperature=0, max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    for a, b in zip([o["token_ids"] for o in one], got[0][0]):
        n = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(a))
        assert len(a) == len(b) and n >= 4, (a, b)


def _logging_factory(log, fail_at=None):
    """Oracle runners whose draft side records the order of (speculation-cache lookup | JIT chain | chained speculation) and of the
    resident-segment error check (ModelRunner.check_segments on the HIP runner) -- optionally failing the n-th check."""
    from oracle.runner import oracle_runner_factory
    base = oracle_runner_factory()

    def factory(config, model_cfg, *, is_draft, **kw):
        r = base(config, model_cfg, is_draft=is_draft, **kw)
        if not is_draft:
            return r

        def wrap(name):
            orig = getattr(r, name)

            def f(*a, **k):
                log.append(name)
                return orig(*a, **k)
            setattr(r, name, f)
        for name in ("draft_jit", "cache_lookup", "speculate_chain", "draft_tree"):
            if hasattr(r, name):
                wrap(name)

        def check(sync=False):
            log.append(("check", sync))
            if fail_at is not None and sum(1 for x in log if isinstance(x, tuple)) == fail_at:
                raise RuntimeError("a bounded wait inside a resident layer segment gave up")
        r.check_segments = check
        return r
    return factory


def _run_logged(mode, log, fail_at=None, same=False):
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = cfgs()
    kw = dict(KW, max_num_seqs=1, draft="d", draft_hf_config=t if same else d, speculate=True, speculate_k=3)
    if same:
        kw.update(draft_weights_seed=0)
    if mode == "async":
        kw.update(draft_async=True, async_fan_out=2, jit_speculate=True)
    eng = LLMEngine("t", hf_config=t, runner_factory=_logging_factory(log, fail_at), inprocess_draft=mode == "async", **kw)
    try:
        out, _ = eng.generate(PROMPTS[:1], SamplingParams(temperature=0.0, max_new_tokens=14, ignore_eos=True), use_tqdm=False)
    finally:
        eng.exit()
    return out[0]["token_ids"]


def test_segment_error_word_is_checked_in_the_round_it_belongs_to():
    """until round 5 the resident segments' error word was looked at on entry of the NEXT draft-side call
    (one round late).  Now: synchronous speculation checks after the verify's read-back of the SAME round, before the accepted
    tokens are committed; the draft server checks right after the cache lookup's read-back (the tree round whose cache it is about
    to serve) and -- reading the device word -- right after a JIT chain, before the reply leaves."""
    log = []
    _run_logged("sync", log)
    kinds = [x for x in log if x == "speculate_chain" or isinstance(x, tuple)]
    assert kinds and kinds[0] == "speculate_chain"
    for a, b in zip(kinds[::2], kinds[1::2]):
        assert a == "speculate_chain" and b == ("check", False), kinds       # every chain is followed by its own round's check
    assert len(kinds) % 2 == 0
    # a failure reported by the n-th check ends THAT round: no further chain is launched, generate() raises
    log2 = []
    import pytest
    with pytest.raises(RuntimeError, match="resident layer segment"):
        _run_logged("sync", log2, fail_at=2)
    assert [x for x in log2 if x == "speculate_chain"].__len__() == 2

    # asynchronous: misses (independent draft) -> lookup, check(False), JIT chain, check(True) in every round
    log = []
    _run_logged("async", log)
    seq = [x for x in log if x in ("cache_lookup", "draft_jit") or isinstance(x, tuple)]
    assert "draft_jit" in seq
    for i, x in enumerate(seq):
        if x == "draft_jit":
            assert seq[i + 1] == ("check", True), seq[i:i + 3]
        if x == "cache_lookup":
            assert seq[i + 1] == ("check", False), seq[i:i + 3]
    # hits (draft == target): every served round is preceded by the check of the tree round that built its cache
    log = []
    _run_logged("async", log, same=True)
    seq = [x for x in log if x == "cache_lookup" or isinstance(x, tuple)]
    assert seq.count("cache_lookup") >= 2
    for i, x in enumerate(seq):
        if x == "cache_lookup":
            assert seq[i + 1] == ("check", False)
    log2 = []
    with pytest.raises(RuntimeError, match="resident layer segment"):
        _run_logged("async", log2, fail_at=3, same=True)
