"""Shared fixtures of the EAGLE-3 engine tests (CPU oracle backend and HIP backend use the same models)."""
import torch

from ssd_amd import weights as W
from ssd_amd.model_config import ModelConfig

TAPS = [0, 1, 3]
K, F = 3, 2


def eagle_cfgs(h_t=128, h_d=64, V=256, Vd=128, hd=32):
    t = ModelConfig("llama", h_t, 4, h_t // hd, 2, hd, 2 * h_t, V, 1e-5, 5e5, 1024, False)
    d = ModelConfig("eagle3", h_d, 1, h_d // hd, 1, hd, 2 * h_d, V, 1e-5, 5e5, 1024, False, draft_vocab_size=Vd,
                    d_model_target=h_t, eagle_taps=len(TAPS))
    return t, d


def peaky_weights(t, d, seed_t=0, seed_d=1, std=0.1, gain=6.0, n_peaks=3):
    """Synthetic target + EAGLE-3 draft whose LM heads both favour the same few tokens: random models never agree, and
    without agreement the speculation-cache hit path and the extend rows of the glue would never run.  Rows of a few
    tokens (present in the draft vocabulary) are scaled by `gain` in both heads -- each model then picks one of them
    most of the time, the same one about a third of the time."""
    tw = W.synthetic_state_dict(t, seed_t, std)
    dw = W.synthetic_state_dict(d, seed_d, std)
    tgt_index = torch.arange(d.draft_vocab_size) + dw["d2t"]
    g = torch.Generator().manual_seed(99)
    picks = torch.randperm(d.draft_vocab_size, generator=g)[:n_peaks]
    for di in picks.tolist():
        ti = int(tgt_index[di])
        tw["lm_head.weight"][ti] = (tw["lm_head.weight"][ti].float() * gain).to(torch.bfloat16)
        dw["lm_head.weight"][di] = (dw["lm_head.weight"][di].float() * gain).to(torch.bfloat16)
    return tw, dw


ENGINE_KW = dict(max_model_len=512, max_num_batched_tokens=512, kvcache_block_size=32, num_kvcache_blocks=48,
                 num_draft_kvcache_blocks=48, weights_std=0.1)
PROMPTS = [[(7 * i + 3 * j + 1) % 256 for j in range(9 + 4 * i)] for i in range(3)]


def eagle_kwargs(t, d, bs=1, **extra):
    kw = dict(ENGINE_KW, hf_config=t, max_num_seqs=bs, draft="eagle-draft", draft_hf_config=d, speculate=True, speculate_k=K,
              draft_async=True, async_fan_out=F, jit_speculate=True, use_eagle=True, eagle_layers=list(TAPS))
    kw.update(extra)
    return kw
