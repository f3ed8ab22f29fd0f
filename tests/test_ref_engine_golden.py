"""The engine against end-to-end runs of the REFERENCE'S OWN ENGINE CLASSES (tests/golden/make_golden.py gen_ref_engine):
its Scheduler, AutoRegressiveStep / SpecDecodeStep, SpeculatorSync / SpeculatorAsync, Verifier, ModelRunner.run and the
DraftRunner loop body executed on CPU (runner instances made without __init__; the async process group replaced by in-process
queues) for a batch of two requests -- autoregressive, synchronous speculation, asynchronous speculation with an independent
draft (every request misses -> JIT), with draft == target (hits, full acceptance; also with non-uniform hit / miss fan-out
lists), and with an EAGLE-3 draft (hits, partial acceptance, extend rows).  This engine, on the oracle backend with the same weights, must produce
the same completions, the same accepted-suffix length at every verification and the same cache-hit rates."""
import pytest
import torch

from oracle.runner import oracle_runner_factory
from ssd_amd.engine.llm_engine import LLMEngine
from ssd_amd.model_config import ModelConfig
from ssd_amd.sampling_params import SamplingParams


def cfg_of(g, prefix, family="llama", **kw):
    ci, cf = g[prefix + "cfg_i"].tolist(), g[prefix + "cfg_f"].tolist()
    return ModelConfig(family, ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], cf[0], cf[1], ci[7], False, **kw)


def weights(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix) and "cfg" not in k}


SCENARIOS = ["ar", "sync", "async_diff", "async_same", "eagle", "async_fanout", "qwen_sync", "qwen_async", "async_eos", "sync_eos", "sync_temp", "ar_temp", "async_temp", "async_same_temp", "async_temp_x", "sync_same", "async_diff_fanout", "async_peaky", "sync_peaky", "async_fast", "eagle_eos", "async_k1f1", "async_k5f3", "eagle_k5f3", "async_queue", "eagle_queue", "sync_queue", "async_dtemp", "sync_dtemp", "async_mixed", "sync_mixed", "async_peaky_fanout", "eagle_k1f1", "eagle_fanout"]


def scenario_setup(g, name):
    """(target weights, draft weights or None, LLMEngine keyword arguments, tokens to generate) of one golden scenario."""
    _, _, bs, nblocks, new_tokens = g["K_F_bs_blocks_new"].tolist()
    K, F = g[name + "/K_F"].tolist()
    qwen = name.startswith("qwen")            # Qwen3 target + Qwen3 draft (same weights): the per-head q / k norm path
    if qwen:
        tcfg, tw = cfg_of(g, "qwen/t_", "qwen3", qk_norm=True), weights(g, "qwen/t.")
    else:
        tcfg, tw = cfg_of(g, "t_"), weights(g, "t.")
    eagle = name.startswith("eagle")
    if eagle:
        tw["lm_head.weight"] = g["eagle/t.lm_head.weight"]
    kw = dict(hf_config=tcfg, max_num_seqs=2, max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=bs,
              num_kvcache_blocks=nblocks, num_draft_kvcache_blocks=nblocks)
    dw = None
    if name not in ("ar", "ar_temp"):
        if name in ("async_same", "async_fanout", "async_eos", "async_same_temp", "sync_same", "async_k1f1") or qwen:
            dw, dcfg = tw, tcfg
        elif eagle:
            dw = weights(g, "eagle/d.")
            dcfg = cfg_of(g, "eagle/d_", "eagle3", draft_vocab_size=int(dw["lm_head.weight"].shape[0]), d_model_target=tcfg.hidden_size,
                          eagle_taps=int(g["eagle/taps"].numel()))
        else:
            dw, dcfg = weights(g, "diff/d."), cfg_of(g, "diff/d_")
            if name.endswith("_peaky") or name in ("async_peaky_fanout","async_fast", "async_k5f3", "async_queue", "sync_queue", "async_dtemp", "sync_dtemp", "async_mixed", "sync_mixed"):         # the same independent draft, three head rows boosted in both models
                tw["lm_head.weight"], dw["lm_head.weight"] = g["peaky/t.lm_head.weight"], g["peaky/d.lm_head.weight"]
        kw.update(draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=K)
        if not name.startswith("sync") and name != "qwen_sync":
            kw.update(draft_async=True, async_fan_out=F, jit_speculate=bool(int(g[name + "/jit"][0])), inprocess_draft=True,
                      fan_out_list=g[name + "/fan"].tolist(), fan_out_list_miss=g[name + "/fan_miss"].tolist())
        if eagle:
            kw.update(use_eagle=True, eagle_layers=g["eagle/taps"].tolist())
    kw["eos"] = int(g[name + "/eos"][0])
    kw["max_model_len"] = int(g[name + "/max_model_len"][0])
    if float(g[name + "/sampler_x"][0]) > 0:
        kw["sampler_x"] = float(g[name + "/sampler_x"][0])
    return tw, dw, kw, new_tokens


def sampling(g, name, kw, new_tokens):
    """One SamplingParams per request (temperatures may differ inside the batch)."""
    dt = float(g[name + "/draft_temp"][0])
    nreq = int(g[name + "/nreq"][0])
    return [SamplingParams(temperature=float(t), draft_temperature=None if dt < 0 else dt, ignore_eos=kw["eos"] < 0,
                           max_new_tokens=new_tokens - (3 * i if nreq == 3 else 0)) for i, t in enumerate(g[name + "/temp"].tolist())]


@pytest.mark.parametrize("name", SCENARIOS)
def test_engine_matches_the_reference_engine_run(golden, name):
    g = golden("ref_engine")
    tw, dw, kw, new_tokens = scenario_setup(g, name)
    eng = LLMEngine("t", runner_factory=oracle_runner_factory(weights_target=tw, weights_draft=dw), **kw)
    nreq = int(g[name + "/nreq"][0])
    prompts = [g[f"prompt{i}"].tolist() for i in range(nreq)]
    sp = sampling(g, name, kw, new_tokens)      # (three requests through two batch slots: each has its own length budget)
    torch.manual_seed(777)          # the seed of the reference run: at temperature > 0 both draw from one global stream
    out, m = eng.generate(prompts, sp, use_tqdm=False)
    for i in range(nreq):
        assert out[i]["token_ids"] == g[name + f"/completion{i}"].tolist(), f"request {i}"
    if kw.get("speculate") and name != "async_fast":        # (a miss is answered with filler tokens: the reference's are random)
        assert list(m["accepted_suffix_lens_with_recovery"]) == g[name + "/accepted_lens"].tolist()
    if kw.get("draft_async") and name != "async_fast":      # (one lucky random filler token saves the reference a step there)
        assert [round(float(h), 4) for h in m["cache_hits"]] == [round(float(h), 4) for h in g[name + "/cache_hits"].tolist()]
    eng.exit()
