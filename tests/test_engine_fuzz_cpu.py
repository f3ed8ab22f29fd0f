"""Randomised engine configurations on the oracle backend: block size, speculation depth and fan-out, number of requests vs
batch slots, prompt / output lengths, EOS, KV pools with and without pressure, JIT vs fast backup, plain and EAGLE-3 drafts.
Invariants: nothing crashes, and without KV pressure (no preemption: a preempted request restarts its completion count)
every speculative mode reproduces the autoregressive stream -- up to a decision whose recorded top-2 margin is a near-tie
(the K+1-row verify and the 1-row decode run different CPU GEMM shapes)."""
import random

import pytest
import torch

from oracle.runner import oracle_runner_factory
from ssd_amd.engine.llm_engine import LLMEngine
from ssd_amd.model_config import ModelConfig
from ssd_amd.sampling_params import SamplingParams
from tests.eagle_util import eagle_cfgs, peaky_weights
from tests.util import assert_stream_matches, seq_margins


def draw(rnd, eagle):
    bs = rnd.choice([16, 32])
    K, F = rnd.choice([1, 2, 3, 4]), rnd.choice([1, 2, 3])
    nreq, slots = rnd.randint(1, 4), rnd.randint(1, 3)
    prompts = [[rnd.randrange(256) for _ in range(rnd.randint(2, 40))] for _ in range(nreq)]
    sps = [SamplingParams(temperature=0, max_new_tokens=rnd.randint(1, 24), ignore_eos=rnd.random() < 0.7) for _ in range(nreq)]
    look = K + 1 + K * F * (K + 1)
    need = sum(-(-(len(p) + sp.max_new_tokens + look) // bs) + 1 for p, sp in zip(prompts, sps))
    tight = rnd.random() < 0.35
    nb = max(-(-(max(len(p) for p in prompts) + 25 + look) // bs) + 2, need // 2) if tight else need + 4
    return dict(bs=bs, K=K, F=F, slots=slots, prompts=prompts, sps=sps, nb=nb, tight=tight, eos=rnd.randrange(256),
                same=rnd.random() < 0.5, jit=eagle or rnd.random() < 0.8)


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("eagle", [False, True])
def test_random_configuration(seed, eagle):
    rnd = random.Random(1000 * eagle + seed)
    c = draw(rnd, eagle)
    if eagle:
        t, d = eagle_cfgs()
        tw, dw = peaky_weights(t, d)
        factory = oracle_runner_factory(weights_target=tw, weights_draft=dw)
    else:
        t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 1024, False)
        d = t if c["same"] else ModelConfig("llama", 64, 1, 2, 1, 32, 128, 256, 1e-5, 5e5, 1024, True)
        factory = oracle_runner_factory()
    base = dict(hf_config=t, max_num_seqs=c["slots"], max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=c["bs"],
                num_kvcache_blocks=c["nb"], num_draft_kvcache_blocks=c["nb"], weights_std=0.1, eos=c["eos"])
    ar = LLMEngine("t", runner_factory=factory, **base)
    want, _ = ar.generate(c["prompts"], c["sps"], use_tqdm=False)
    margins = ar.model_runner.margin_log
    spec = dict(base, draft="d", draft_hf_config=d, speculate=True, speculate_k=c["K"])
    if not eagle and c["same"]:
        spec.update(draft_weights_seed=0)
    modes = {"async": dict(spec, draft_async=True, async_fan_out=c["F"], jit_speculate=c["jit"], inprocess_draft=True)}
    if eagle:
        modes["async"].update(use_eagle=True, eagle_layers=[0, 1, 3])
    else:
        modes["sync"] = spec
    for mode, kw in modes.items():
        eng = LLMEngine("t", runner_factory=factory, **kw)
        got, _ = eng.generate(c["prompts"], c["sps"], use_tqdm=False)      # must not crash, whatever the pool size
        eng.exit()
        assert len(got) == len(want)
        if c["tight"]:
            continue
        for i, (g, w) in enumerate(zip(got, want)):
            if len(g["token_ids"]) != len(w["token_ids"]):          # an EOS on one side of a near-tie flip
                n = next((j for j, (p, q) in enumerate(zip(g["token_ids"], w["token_ids"])) if p != q), None)
                assert n is not None and seq_margins(margins, i).get(len(c["prompts"][i]) + n, 1.0) <= 0.0625, (mode, i)
                continue
            assert_stream_matches(g["token_ids"], w["token_ids"], seq_margins(margins, i), len(c["prompts"][i]), f"{mode} request {i}")
