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
