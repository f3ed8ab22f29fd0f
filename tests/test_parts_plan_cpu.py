"""HipDecoder._parts_cfg (host logic, no GPU): every (splits, waves) plan it hands to ssd_gemm_parts must respect the kernel's
"<= 8 k-tiles per wave" register budget for the shapes that take the slab path -- ADVICE r2: the two-slab fused-consumer plan
violated it for h = 3072 with I = 14336 and hard-failed at launch."""
import itertools

from ssd_amd.model import HipDecoder


def tiles_per_wave(N, K, fused):
    S, wv = HipDecoder._parts_cfg(N, K, fused)
    assert 1 <= S <= 16 and 1 <= wv <= 16
    return -(-(-(-(K // 32) // S)) // wv)


def test_presets_on_the_slab_path_fit_the_register_budget():
    from ssd_amd.model_config import PRESETS
    for name, cfg in PRESETS.items():
        if cfg.family == "eagle3":
            continue
        h, qn, I = cfg.hidden_size, cfg.num_heads * cfg.head_dim, cfg.intermediate_size
        g = h // 16
        if not (g < 256 or 256 < g < 512):
            continue                         # rows kernels
        fused_possible = (not cfg.qk_norm) and h // 8 <= 1024
        ok = all(tiles_per_wave(h, k, f) <= 8 for k in (qn, I) for f in ((False, True) if fused_possible else (False,)))
        # a model whose plan does not fit must fall back to the rows kernels (use_parts False), never fail at launch: the same
        # predicate HipDecoder.__init__ evaluates
        if name in ("llama-3.2-1b", "qwen3-0.6b", "qwen3-32b"):
            assert ok, name


def test_unfused_consumer_plans_fit_for_arbitrary_shapes():
    for N, K in itertools.product((1024, 2048, 3072, 5120, 6144), (1024, 2048, 4096, 8192, 14336, 25600, 28672)):
        g = N // 16
        if g < 256 or 256 < g < 512:
            t = tiles_per_wave(N, K, False)
            assert t <= 8 or K // 32 > 16 * 16 * 8, (N, K, t)


def test_peaky_recipe_scales_the_same_tokens_in_target_and_eagle_draft():
    """bench.py --workload c4e builds its EAGLE pair with the "peaky" recipe (ssd_amd/weights.py peaky_rows): the LM-head rows
    of the same few tokens -- draft-vocabulary ids pushed through the draft's d2t -- are scaled in BOTH models, everything
    else is the plain seeded N(0, std)."""
    import torch
    from ssd_amd import weights as W
    from ssd_amd.model_config import ModelConfig
    t = ModelConfig("llama", 128, 2, 4, 2, 32, 256, 512, 1e-5, 5e5, 1024, False)
    d = ModelConfig("eagle3", 64, 1, 2, 1, 32, 128, 512, 1e-5, 5e5, 1024, False, draft_vocab_size=128, d_model_target=128, eagle_taps=3)
    rec = {"kind": "peaky", "gain": 6.0, "peaks": 3, "draft_seed": 1, "draft_vocab": 128, "target_vocab": 512}
    tw, dw = dict(W.synthetic_weights(t, 0, 0.1, recipe=rec)), dict(W.synthetic_weights(d, 1, 0.1, recipe=rec))
    plain_t, plain_d = dict(W.synthetic_weights(t, 0, 0.1)), dict(W.synthetic_weights(d, 1, 0.1, recipe={"target_vocab": 512}))
    dp, tp = W.peaky_rows(rec)
    assert ((torch.arange(128) + dw["d2t"])[dp] == tp).all()                 # the SAME tokens in both vocabularies
    for w, plain, rows in ((tw, plain_t, tp), (dw, plain_d, dp)):
        ratio = w["lm_head.weight"].float().norm(dim=1) / plain["lm_head.weight"].float().norm(dim=1)
        assert torch.allclose(ratio[rows], torch.full((3,), 6.0), rtol=0.02)
        mask = torch.ones(ratio.numel(), dtype=torch.bool)
        mask[rows] = False
        assert torch.allclose(ratio[mask], torch.ones(int(mask.sum())), rtol=1e-3)
        assert all(torch.equal(w[k], plain[k]) for k in w if k != "lm_head.weight")
