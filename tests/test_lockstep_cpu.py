"""The lock-step harness itself (tests/lockstep.py), on the CPU: oracle engine against oracle engine at tiny shapes.  Identical
engines must compare every token and every round with nothing excused; a wrong draft must be caught at the first speculated
token; and with the excuse threshold opened wide the re-synchronisation (teacher forcing) must carry a run with a different
draft to the end with every STREAM token still compared (speculation never changes the target's greedy stream)."""
import pytest

from oracle.runner import oracle_runner_factory
from ssd_amd import weights as W
from ssd_amd.engine.llm_engine import LLMEngine
from ssd_amd.model_config import ModelConfig
from ssd_amd.sampling_params import SamplingParams
from tests.lockstep import compare_lockstep

T = ModelConfig("llama", 128, 2, 4, 2, 32, 256, 512, 1e-5, 5e5, 1024, False)
D = ModelConfig("llama", 64, 1, 2, 1, 32, 128, 512, 1e-5, 5e5, 1024, False)
RECIPE = {"kind": "pair", "shared": D.hidden_size, "snr": 6.0, "layer_gain": 0.05}
PROMPT = [(7 * j + 3) % 512 for j in range(24)]


def _engine(mode, wrong_draft=False):
    wt = dict(W.synthetic_weights(T, 0, 0.05, recipe=RECIPE))
    wd = dict(W.synthetic_weights(D, 1, 0.05, recipe=dict(RECIPE, seed=99) if wrong_draft else RECIPE))     # other base vectors: it proposes other tokens
    kw = dict(hf_config=T, draft="d", draft_hf_config=D, speculate=True, max_num_seqs=1, max_model_len=512, max_num_batched_tokens=512,
              kvcache_block_size=16, num_kvcache_blocks=40, num_draft_kvcache_blocks=60)
    if mode == "async":
        kw.update(speculate_k=3, draft_async=True, async_fan_out=2, jit_speculate=True)
    else:
        kw.update(speculate_k=3)
    return LLMEngine("t", runner_factory=oracle_runner_factory(wt, wd), inprocess_draft=mode == "async", **kw)


def _sp(n):
    return SamplingParams(temperature=0, max_new_tokens=n, ignore_eos=True)


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_identical_engines_compare_everything(mode):
    rep = compare_lockstep(_engine(mode), _engine(mode), PROMPT, 40, _sp, fan_out=2 if mode == "async" else None, what=mode)
    assert rep.tokens == 40 and rep.tokens_compared == 40, rep.summary()
    assert rep.rounds_compared == rep.rounds and not rep.excused and rep.restarts == 0, rep.summary()
    assert max(rep.accepted_lens) > 1, "the pair was meant to agree some of the time: " + rep.summary()
    if mode == "async":
        assert rep.hits > 0 and rep.misses > 0, rep.summary()


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_a_wrong_draft_is_caught(mode):
    with pytest.raises(AssertionError, match="speculated token|hit flag"):
        compare_lockstep(_engine(mode, wrong_draft=True), _engine(mode), PROMPT, 40, _sp, fan_out=2 if mode == "async" else None, what=mode)


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_resynchronisation_carries_the_run_to_the_end(mode):
    rep = compare_lockstep(_engine(mode, wrong_draft=True), _engine(mode), PROMPT, 40, _sp, fan_out=2 if mode == "async" else None,
                           thr=float("inf"), max_restarts=64, what=mode)
    assert rep.tokens == 40 and rep.tokens_compared == 40, rep.summary()        # the target's stream is the same whatever the draft says
    assert rep.restarts > 0 and rep.rounds_compared < rep.rounds and all(e[0] != "target" for e in rep.excused), rep.summary()


def test_eagle3_engines_compare_everything():
    """The harness over the EAGLE-3 path (target taps, activations on the wire, extend rows, cached prenorms): identical oracle
    engines on the constructed agreeing pair (weights.eagle_pair_recipe) compare every token and round; hits and acceptances occur."""
    from ssd_amd.utils.topology import Topology
    import torch
    from tests.eagle_util import TAPS
    hd = 32
    t = ModelConfig("llama", 128, 4, 4, 2, hd, 256, 2048, 1e-5, 5e5, 1024, False)
    d = ModelConfig("eagle3", 128, 1, 4, 2, hd, 256, 2048, 1e-5, 5e5, 1024, False, draft_vocab_size=512, d_model_target=128, eagle_taps=len(TAPS))
    rec = W.eagle_pair_recipe(t, d, draft_seed=1)
    tw, dw = W.synthetic_state_dict(t, 0, 0.1, recipe=rec), W.synthetic_state_dict(d, 1, 0.1, recipe=rec)
    kw = dict(hf_config=t, draft="e", draft_hf_config=d, speculate=True, speculate_k=3, draft_async=True, async_fan_out=2, jit_speculate=True,
              use_eagle=True, eagle_layers=list(TAPS), max_num_seqs=1, max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=16,
              num_kvcache_blocks=40, num_draft_kvcache_blocks=60)
    cpu = Topology(0, 1, torch.device("cpu"), "target", 0, 1)
    engs = [LLMEngine("t", runner_factory=oracle_runner_factory(tw, dw), inprocess_draft=True, topology=cpu, **kw) for _ in range(2)]
    rep = compare_lockstep(engs[0], engs[1], [(7 * j + 1) % 2048 for j in range(13)], 40, _sp, fan_out=2, what="eagle3")
    assert rep.tokens == 40 and rep.tokens_compared == 40 and rep.rounds_compared == rep.rounds and not rep.excused, rep.summary()
    assert rep.hits > 0 and max(rep.accepted_lens) > 2, rep.summary()
