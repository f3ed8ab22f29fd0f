"""Multi-rank paths on CPU: two gloo processes, tensor-parallel oracle runners under the REAL engine (SPMD: both
ranks run the same scheduler / speculation loop and only meet in collectives).  TP=2 must reproduce TP=1 token
for token, both autoregressive and speculative, and the weight sharding used by the product must equal the
reference's loaders (restated in oracle.model.shard_weights)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfgs():
    from ssd_amd.model_config import ModelConfig
    t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 512, False)
    d = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 512, True)
    return t, d


KW = dict(max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40, weights_std=0.1)
PROMPTS = [[(5 * i + 3 * j) % 256 for j in range(6 + 2 * i)] for i in range(3)]


def _generate(num_gpus, speculate):
    from oracle.runner import oracle_runner_factory
    from ssd_amd.engine.llm_engine import LLMEngine
    from ssd_amd.sampling_params import SamplingParams
    t, d = _cfgs()
    kw = dict(KW, num_gpus=num_gpus, max_num_seqs=3)
    if speculate:
        kw.update(draft="d", draft_hf_config=d, speculate=True, speculate_k=3)
    eng = LLMEngine("t", hf_config=t, runner_factory=oracle_runner_factory(), **kw)
    out, m = eng.generate(PROMPTS, SamplingParams(temperature=0, max_new_tokens=12, ignore_eos=True), use_tqdm=False)
    return [o["token_ids"] for o in out]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    res = {"ar": _generate(world, False), "sd": _generate(world, True)}
    q.put((rank, res))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_matches_tp1_under_gloo():
    ref = {"ar": _generate(1, False), "sd": _generate(1, True)}
    assert ref["ar"] == ref["sd"]           # speculative decoding is exact
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):                       # every rank holds the full result (SPMD)
        assert got[r]["ar"] == ref["ar"], f"rank {r} AR differs from TP=1"
        assert got[r]["sd"] == ref["sd"], f"rank {r} SD differs from TP=1"


def test_product_sharding_equals_reference_loaders():
    from oracle.model import shard_weights
    from ssd_amd import weights as W
    t, _ = _cfgs()
    full = W.synthetic_state_dict(t, 3, 0.1)
    for tp in (2,):
        for rank in range(tp):
            want = shard_weights(t, full, rank, tp)
            for name, w in full.items():
                got = W.shard_param(t, name, w, rank, tp)
                assert torch.equal(got, want[name]), name
    # rank-independent full tensors: the iterator for rank r is the shard of the same full tensors
    for rank in range(2):
        for name, w in W.synthetic_weights(t, 3, 0.1, rank, 2):
            assert torch.equal(w, W.shard_param(t, name, full[name], rank, 2))


def test_safetensors_roundtrip(tmp_path):
    """HF-named checkpoint (q_proj/k_proj/v_proj, gate_proj/up_proj) -> packed reference names."""
    from safetensors.torch import save_file
    from ssd_amd import weights as W
    t, _ = _cfgs()
    full = W.synthetic_state_dict(t, 9, 0.1)
    hf = {}
    hd, nh, nkv, I = t.head_dim, t.num_heads, t.num_kv_heads, t.intermediate_size
    for name, w in full.items():
        if "qkv_proj" in name:
            q, k, v = w.split([nh * hd, nkv * hd, nkv * hd], 0)
            for s, x in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                hf[name.replace("qkv_proj", s)] = x.contiguous()
        elif "gate_up_proj" in name:
            g, u = w.split([I, I], 0)
            hf[name.replace("gate_up_proj", "gate_proj")] = g.contiguous()
            hf[name.replace("gate_up_proj", "up_proj")] = u.contiguous()
        else:
            hf[name] = w
    save_file(hf, str(tmp_path / "model.safetensors"))
    assert W.has_safetensors(str(tmp_path))
    loaded = dict(W.load_safetensors(t, str(tmp_path)))
    assert set(loaded) == set(full)
    for k in full:
        assert torch.equal(loaded[k], full[k]), k


@pytest.mark.parametrize("family", ["llama", "qwen3"])
def test_safetensors_loader_matches_the_reference_loader_per_rank(golden, tmp_path, family):
    """tests/golden/loader_tp.npz: an HF-named checkpoint went through the REFERENCE's load_model into tensor-parallel
    (tp_size 2) reference models; weights.load_safetensors must hand every rank exactly the slices the reference's
    weight_loaders put there (packed q/k/v and gate/up, row-parallel o/down, vocabulary-parallel embedding and head)."""
    from safetensors.torch import save_file
    from ssd_amd import weights as W
    from ssd_amd.model_config import ModelConfig
    g = golden("loader_tp")
    hf = {k[len(family) + 4:]: v.contiguous() for k, v in g.items() if k.startswith(family + ".hf.")}
    save_file(hf, str(tmp_path / "model.safetensors"))
    qk = family == "qwen3"
    cfg = ModelConfig(family, 128, 2, 4, 2, 32 if not qk else 64, 256, 512, 1e-6 if qk else 1e-5, 1e6 if qk else 5e5, 512, False, qk)
    for rank in range(2):
        got = dict(W.load_safetensors(cfg, str(tmp_path), rank, 2))
        want = {k[len(family) + 7:]: v for k, v in g.items() if k.startswith(f"{family}.rank{rank}.")}
        assert set(got) == set(want), set(got) ^ set(want)
        for name, v in want.items():
            assert torch.equal(got[name], v), f"{family} rank {rank}: {name}"
