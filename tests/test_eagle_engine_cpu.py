"""EAGLE-3 through the real engine on CPU (oracle runners): scheduler bookkeeping, the wire protocol with activation
tensors, the draft server's hit / JIT / extend paths.  The output must be the target's own greedy stream whatever the
draft proposes; the peaky weight pair (tests/eagle_util.py) makes it propose acceptable tokens often enough that cache
hits, multi-token acceptance and the extend rows all occur."""
import os
import sys

import torch
import torch.multiprocessing as mp

from tests.eagle_util import eagle_cfgs, peaky_weights, eagle_kwargs, ENGINE_KW, PROMPTS, K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def generate(mode, bs=1, new_tokens=40, **extra):
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = eagle_cfgs()
    tw, dw = peaky_weights(t, d)
    factory = oracle_runner_factory(weights_target=tw, weights_draft=dw)
    if mode == "ar":
        eng = LLMEngine("t", runner_factory=factory, **dict(ENGINE_KW, hf_config=t, max_num_seqs=bs))
    else:
        eng = LLMEngine("t", runner_factory=factory, inprocess_draft="num_gpus" not in extra, **eagle_kwargs(t, d, bs=bs, **extra))
    out, m = eng.generate(PROMPTS[:bs], SamplingParams(temperature=0, max_new_tokens=new_tokens, ignore_eos=True), use_tqdm=False)
    stats = eng.draft_server.stats if eng.draft_server is not None else None
    eng.exit()
    return [o["token_ids"] for o in out], m, stats


def test_eagle_async_stream_is_exact_and_uses_hit_and_extend_paths():
    ar, _, _ = generate("ar")
    asy, m, stats = generate("eagle")
    assert asy == ar
    lens = m["accepted_suffix_lens_with_recovery"]
    assert max(lens) >= 2, lens                     # some draft token was accepted -> the next request carried extend rows
    assert stats["hits"] >= 1                       # some request was answered from the speculation cache (with its prenorms)
    assert stats["requests"] == len(lens)


def test_constructed_eagle_pair_agrees_with_its_target():
    """weights.eagle_pair_recipe (bench.py --workload c4e): a target and an EAGLE-3 draft built so that the draft computes the
    target's own token map from the token it is fed (one-hot self attention, value = token half - conditioning half).  Through
    the real engine: the stream is the target's greedy stream, most speculated tokens are accepted and most requests hit the
    speculation cache -- for two geometries (2 kv heads of 32 dims; 1 kv head, draft narrower than the target)."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd import weights as W
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.model_config import ModelConfig
    from ssd_amd.sampling_params import SamplingParams
    from tests.eagle_util import TAPS
    for h_t, h_d, nkv_d, V, Vd in ((128, 128, 2, 2048, 512), (256, 96, 1, 2048, 512)):
        hd = 32
        t = ModelConfig("llama", h_t, 4, h_t // hd, 2, hd, 2 * h_t, V, 1e-5, 5e5, 1024, False)
        d = ModelConfig("eagle3", h_d, 1, h_d // hd, nkv_d, hd, 2 * h_d, V, 1e-5, 5e5, 1024, False, draft_vocab_size=Vd,
                        d_model_target=h_t, eagle_taps=len(TAPS))
        rec = W.eagle_pair_recipe(t, d, draft_seed=1)
        assert rec["sig"] == nkv_d * hd
        tw = W.synthetic_state_dict(t, 0, 0.1, recipe=rec)
        dw = W.synthetic_state_dict(d, 1, 0.1, recipe=rec)
        # the draft vocabulary's head rows are the target's shared head dims at d2t's positions; the target boosts exactly those
        tgt = torch.arange(Vd) + dw["d2t"]
        ds = rec["sig"]
        a, b = dw["lm_head.weight"][:, :ds].float(), tw["lm_head.weight"][tgt, :ds].float() / rec["boost"]
        assert torch.allclose(a, b, rtol=2e-2, atol=1e-3) and float(dw["lm_head.weight"][:, ds:].abs().max()) == 0.0
        factory = oracle_runner_factory(weights_target=tw, weights_draft=dw)
        prompt = [[(7 * j + 1) % V for j in range(13)]]
        sp = SamplingParams(temperature=0, max_new_tokens=48, ignore_eos=True)
        eng = LLMEngine("t", runner_factory=factory, **dict(ENGINE_KW, hf_config=t, max_num_seqs=1))
        ar, _ = eng.generate(prompt, sp, use_tqdm=False)
        eng.exit()
        eng = LLMEngine("t", runner_factory=factory, inprocess_draft=True, **eagle_kwargs(t, d))
        out, m = eng.generate(prompt, sp, use_tqdm=False)
        stats = eng.draft_server.stats
        eng.exit()
        assert out[0]["token_ids"] == ar[0]["token_ids"]
        assert all(int(x) in set(tgt.tolist()) for x in ar[0]["token_ids"][1:])       # the target stays inside the draft vocabulary
        lens = m["accepted_suffix_lens_with_recovery"]
        assert sum(lens) / len(lens) >= 2.5, lens                                       # of at most K + 1 = 4
        assert stats["hits"] >= 0.7 * stats["requests"], stats


def test_eagle_async_batch():
    ar, _, _ = generate("ar", bs=3, new_tokens=24)
    asy, m, stats = generate("eagle", bs=3, new_tokens=24)
    assert asy == ar


def test_eagle_config_rules():
    import pytest
    from ssd_amd.config import Config
    t, d = eagle_cfgs()
    base = dict(hf_config=t, draft="d", draft_hf_config=d, speculate=True, speculate_k=K, use_eagle=True)
    with pytest.raises(AssertionError):
        Config("t", **base)                                                     # synchronous speculation
    with pytest.raises(AssertionError):
        Config("t", draft_async=True, num_gpus=2, **base)                       # jit_speculate is mandatory
    c = Config("t", draft_async=True, num_gpus=2, jit_speculate=True, **base)
    assert c.eagle_layers == [2, 2, 1] and c.draft_hf_config.eagle_taps == 2    # the reference's default taps, de-duplicated
    assert c.draft_hf_config.family == "eagle3" and c.draft_hf_config.d_model_target == t.hidden_size


def test_proto_roundtrip_with_extend_block():
    from ssd_amd.engine import async_proto as P
    keys, nts, tabs, temps = [(3, 1, 9), (4, -2, 7)], [20, 31], [[1, 2], [5]], [0.0, 0.0]
    payload = P.pack_speculate(keys, nts, tabs, temps, 4, [2, 0], [[11, 12, 0], [0, 0, 0]])
    k2, n2, t2, tm2, counts, ids = P.unpack_speculate(payload, 2, 4, 3)
    assert k2 == keys and n2 == nts and counts == [2, 0] and ids == [[11, 12, 0], [0, 0, 0]]
    assert t2 == [[1, 2, -1, -1], [5, -1, -1, -1]]


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    toks, m, _ = generate("eagle", num_gpus=2)
    q.put((rank, toks, m["accepted_suffix_lens_with_recovery"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_eagle_two_processes_gloo():
    """Target on rank 0, the EAGLE draft server on rank 1: the activation tensors cross the process boundary."""
    ar, _, _ = generate("ar")
    _, m_loop, _ = generate("eagle")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90)
    ps = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, toks, lens = q.get(timeout=300)
        got[r] = (toks, lens)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == ar and got[1][0] == []
    assert got[0][1] == m_loop["accepted_suffix_lens_with_recovery"]      # same acceptance trace as the in-process server


def test_eagle_checkpoint_loader_maps_reference_names(tmp_path):
    """An EAGLE-3 checkpoint as published (flat names, q/k/v and gate/up unpacked, midlayer.hidden_norm, no embedding)
    + the target checkpoint that lends its embed_tokens (reference loader.py:64-183) -> this engine's parameter names."""
    from safetensors.torch import save_file
    from ssd_amd import weights as W
    t, d = eagle_cfgs()
    ref = W.synthetic_state_dict(d, 5, 0.05)
    nh, nkv, hd, I = d.num_heads, d.num_kv_heads, d.head_dim, d.intermediate_size
    q, k, v = ref["model.layer.self_attn.qkv_proj.weight"].split([nh * hd, nkv * hd, nkv * hd], dim=0)
    gate, up = ref["model.layer.mlp.gate_up_proj.weight"].split([I, I], dim=0)
    ckpt = {"midlayer.self_attn.q_proj.weight": q, "midlayer.self_attn.k_proj.weight": k, "midlayer.self_attn.v_proj.weight": v,
            "midlayer.self_attn.o_proj.weight": ref["model.layer.self_attn.o_proj.weight"],
            "midlayer.mlp.gate_proj.weight": gate, "midlayer.mlp.up_proj.weight": up,
            "midlayer.mlp.down_proj.weight": ref["model.layer.mlp.down_proj.weight"],
            "midlayer.input_layernorm.weight": ref["model.layer.input_layernorm.weight"],
            "midlayer.hidden_norm.weight": ref["model.layer.conditioning_feature_ln.weight"],
            "midlayer.post_attention_layernorm.weight": ref["model.layer.post_attention_layernorm.weight"],
            "norm.weight": ref["final_norm.weight"], "fc.weight": ref["fc.weight"], "lm_head.weight": ref["lm_head.weight"],
            "d2t": ref["d2t"], "t2d": torch.zeros(d.vocab_size, dtype=torch.bool)}
    ddir, tdir = tmp_path / "eagle", tmp_path / "target"
    ddir.mkdir()
    tdir.mkdir()
    save_file({k_: v_.contiguous() for k_, v_ in ckpt.items()}, str(ddir / "model.safetensors"))
    save_file({"model.embed_tokens.weight": ref["model.embed_tokens.weight"].contiguous()}, str(tdir / "model-00001-of-00001.safetensors"))
    got = dict(W.load_eagle_safetensors(d, str(ddir), target_dir=str(tdir)))
    assert set(got) == set(ref)
    for name in ref:
        assert torch.equal(got[name], ref[name]), name
    assert got["d2t"].dtype == torch.int64


def _worker_tp(rank, port, q):
    """TP = 2 target (taps are taken after the all-reduce: identical on both ranks) + the EAGLE draft co-located on rank 0."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = eagle_cfgs()
    tw, dw = peaky_weights(t, d)
    eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), inprocess_draft=True,
                    **eagle_kwargs(t, d, num_gpus=2))
    assert eng.topo.tp_size == 2 and (eng.draft_server is not None) == (rank == 0)
    out, m = eng.generate(PROMPTS[:1], SamplingParams(temperature=0, max_new_tokens=40, ignore_eos=True), use_tqdm=False)
    q.put((rank, [o["token_ids"] for o in out], m["accepted_suffix_lens_with_recovery"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_eagle_under_tensor_parallel_target_gloo():
    ar, _, _ = generate("ar")
    _, m_loop, _ = generate("eagle")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 90)
    ps = [ctx.Process(target=_worker_tp, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, toks, lens = q.get(timeout=300)
        got[r] = (toks, lens)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == ar and got[1][0] == ar
    assert got[0][1] == got[1][1]


def test_eagle_preemption_reprefills_with_fresh_activations():
    """A KV pool too small for three growing sequences: the scheduler preempts, the victim is re-prefilled as a longer prompt
    (target activations of ALL its tokens again, extend state dropped) and keeps following the target's greedy stream.  A
    preempted sequence restarts its completion count (reference scheduler semantics), so its output is a later window of
    the stream an unconstrained autoregressive run produces."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = eagle_cfgs()
    tw, dw = peaky_weights(t, d)
    prompts = [[(5 * i + 3 * j + 2) % 256 for j in range(20 + 6 * i)] for i in range(3)]
    base = dict(kvcache_block_size=16, max_model_len=256, max_num_batched_tokens=256)
    ar = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw),
                   **dict(ENGINE_KW, hf_config=t, max_num_seqs=3, num_kvcache_blocks=64, **base))
    want, _ = ar.generate(prompts, SamplingParams(temperature=0, max_new_tokens=120, ignore_eos=True), use_tqdm=False)
    eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), inprocess_draft=True,
                    **eagle_kwargs(t, d, bs=3, num_kvcache_blocks=9, num_draft_kvcache_blocks=40, **base))
    preempted = []
    orig = eng.scheduler.preempt

    def spy(seq):
        preempted.append(seq.seq_id)
        return orig(seq)
    eng.scheduler.preempt = spy
    got, m = eng.generate(prompts, SamplingParams(temperature=0, max_new_tokens=30, ignore_eos=True), use_tqdm=False)
    assert preempted, "the pool was meant to be too small"
    shifted = 0
    for o, w in zip(got, want):
        o, w = o["token_ids"], w["token_ids"]
        starts = [k for k in range(len(w) - len(o) + 1) if w[k:k + len(o)] == o]
        assert starts, "a preempted EAGLE sequence left the target's greedy stream"
        shifted += starts[0] > 0
    assert shifted >= 1


def test_streaming_does_not_change_a_preempted_sequence():
    """the token streamed at the prefill is pinned only until the round that appends it.  A sequence
    preempted LATER restarts its completion count at 0; honouring the stale pin there put the first completion token in
    place of the freshly computed one.  Same pool-too-small set-up as above, with and without a stream callback: the
    results must be identical, and every streamed first token must be the result's first token."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = eagle_cfgs()
    tw, dw = peaky_weights(t, d)
    prompts = [[(5 * i + 3 * j + 2) % 256 for j in range(20 + 6 * i)] for i in range(3)]
    base = dict(kvcache_block_size=16, max_model_len=256, max_num_batched_tokens=256)
    runs = []
    for streaming in (False, True):
        eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), inprocess_draft=True,
                        **eagle_kwargs(t, d, bs=3, num_kvcache_blocks=9, num_draft_kvcache_blocks=40, **base))
        preempted, first = [], {}
        orig = eng.scheduler.preempt

        def spy(seq, orig=orig, preempted=preempted):
            preempted.append(seq.seq_id)
            return orig(seq)
        eng.scheduler.preempt = spy

        def cb(seq_id, toks, first=first):
            first.setdefault(seq_id, toks[0])
        got, _ = eng.generate(prompts, SamplingParams(temperature=0, max_new_tokens=30, ignore_eos=True), use_tqdm=False,
                              stream_callback=cb if streaming else None)
        assert preempted, "the pool was meant to be too small"
        runs.append([o["token_ids"] for o in got])
        if streaming:
            assert len(first) == 3
    assert runs[0] == runs[1], "streaming changed the tokens of a preempted sequence"


def test_streamed_token_of_a_length_capped_sequence_is_in_its_result():
    """prompt + lookahead > max_model_len -> the scheduler finishes the sequence before its first round;
    the token the stream already delivered at the prefill must be in the returned token_ids."""
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = eagle_cfgs()
    tw, dw = peaky_weights(t, d)
    base = dict(kvcache_block_size=16, max_model_len=64, max_num_batched_tokens=256)
    eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), inprocess_draft=True,
                    **eagle_kwargs(t, d, bs=1, num_kvcache_blocks=16, num_draft_kvcache_blocks=40, **base))
    seen = []
    got, _ = eng.generate([[(3 * j + 1) % 256 for j in range(62)]], SamplingParams(temperature=0, max_new_tokens=30, ignore_eos=True),
                          use_tqdm=False, stream_callback=lambda sid, toks: seen.extend(toks))
    assert got[0]["finish_reason"] == "max_model_len"
    assert seen == got[0]["token_ids"] and len(seen) == 1


def test_glue_layout_matches_the_reference_mask_construction():
    """ssd_amd.engine.eagle_runner.glue_layout against the reference's construction of the same packed batch, restated
    here with the reference's own tensor formulas (draft_runner.py:555-566 lengths and offsets, :568-576 the extend /
    recovery / spec masks, :458-468 positions and context lengths, :640-644 the rows extracted for the fork)."""
    import random
    from ssd_amd.engine.eagle_runner import glue_layout
    random.seed(5)
    for _ in range(50):
        B, K_ = random.randint(1, 5), random.randint(1, 7)
        num_tokens_ref = torch.tensor([random.randint(K_ + 3, 300) for _ in range(B)])      # the reference's UNSHIFTED counts
        extend_counts = torch.tensor([random.randint(0, K_) for _ in range(B)])
        # --- reference formulas ---
        seqlens_q = (extend_counts + K_ + 1).to(torch.int32)
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(seqlens_q, 0)
        total = int(cu[-1])
        batch_idx = torch.repeat_interleave(torch.arange(B), seqlens_q)
        local_off = torch.arange(total) - cu[:-1].long().repeat_interleave(seqlens_q)
        n_ext_per_tok = extend_counts[batch_idx]
        is_extend, is_rec, is_spec = local_off < n_ext_per_tok, local_off == n_ext_per_tok, local_off > n_ext_per_tok
        base_pos = (num_tokens_ref - 2 - extend_counts).long()
        positions = base_pos[batch_idx] + local_off
        context_lens = num_tokens_ref - 1 + K_
        rec_offsets = cu[:-1].long() + extend_counts.long()
        extract_idx = (rec_offsets.unsqueeze(1) + torch.arange(K_ + 1).unsqueeze(0)).flatten()
        # --- this engine (the draft server hands the runner num_tokens - 1) ---
        shifted = [int(n) - 1 for n in num_tokens_ref]
        lay = glue_layout(shifted, extend_counts.tolist(), K_)
        assert lay["cu"] == cu.tolist() and lay["pos"] == positions.tolist()
        assert [n + K_ for n in shifted] == context_lens.tolist()                            # what the runners upload as context lengths
        assert lay["kp1"] == extract_idx.tolist()
        assert lay["tc_dst"] == torch.nonzero(is_extend | is_rec).flatten().tolist()
        assert lay["sp_dst"] == torch.nonzero(is_spec).flatten().tolist()
        # sources: extend row j of sequence b -> extend_eagle_acts[b, j]; recovery -> the recovery activation (row K of our tensor)
        want_src = []
        for t in torch.nonzero(is_extend | is_rec).flatten().tolist():
            b, j = int(batch_idx[t]), int(local_off[t])
            want_src.append(b * (K_ + 1) + (j if j < int(extend_counts[b]) else K_))
        assert lay["tc_src"] == want_src


def test_eagle_checkpoint_loader_matches_the_reference_loader(golden, tmp_path):
    """tests/golden/eagle_loader.npz: an EAGLE-3 checkpoint with the published flat names went through the REFERENCE's
    load_eagle_model (loader.py:64-183); weights.load_eagle_safetensors must put every tensor where the reference did --
    with the checkpoint's own embedding and with the target's borrowed one."""
    from safetensors.torch import save_file
    from ssd_amd import weights as W
    from ssd_amd.model_config import ModelConfig
    g = golden("eagle_loader")
    cfg = ModelConfig("eagle3", 128, 1, 2, 1, 64, 256, 512, 1e-5, 5e5, 512, False, draft_vocab_size=256, d_model_target=128, eagle_taps=3)
    ckpt = {k[5:]: v.contiguous() for k, v in g.items() if k.startswith("ckpt.")}
    for case in ("own", "borrowed"):
        ddir, tdir = tmp_path / f"{case}-draft", tmp_path / f"{case}-target"
        ddir.mkdir()
        tdir.mkdir()
        tensors = dict(ckpt)
        if case == "own":
            tensors["embed_tokens.weight"] = g["ckpt_embed_own"].contiguous()
        save_file(tensors, str(ddir / "model.safetensors"))
        save_file({"model.embed_tokens.weight": g["target_embed"].contiguous()}, str(tdir / "model-00001-of-00001.safetensors"))
        got = dict(W.load_eagle_safetensors(cfg, str(ddir), target_dir=str(tdir)))
        want = {k[len(case) + 1:]: v for k, v in g.items() if k.startswith(case + ".")}
        assert set(got) == set(want), set(got) ^ set(want)
        for name, v in want.items():
            assert torch.equal(got[name], v), f"{case}: {name}"
