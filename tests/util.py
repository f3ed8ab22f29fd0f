import torch


def bits(t):
    return t.contiguous().view(torch.int16).int()


def ulp_stats(a: torch.Tensor, b: torch.Tensor):
    """(max ulp distance, fraction differing) between two bf16 tensors (sign-magnitude aware)."""
    def key(t):
        x = bits(t)
        return torch.where(x < 0, -(x & 0x7FFF), x)
    d = (key(a.cpu()) - key(b.cpu())).abs()
    return int(d.max()), float((d > 0).float().mean())


def assert_close_bf16(a, b, max_ulp=1, max_frac=0.02, abs_floor=0.0, rel_floor=0.0, what=""):
    """bf16 tensors agree within max_ulp (elements whose abs diff is below abs_floor are exempt: near zero an
    ulp is meaninglessly small)."""
    a, b = a.cpu(), b.cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    af, bf = a.float(), b.float()
    assert torch.isfinite(af).all(), f"{what}: non-finite values"
    def key(t):
        x = bits(t)
        return torch.where(x < 0, -(x & 0x7FFF), x)
    d = (key(a) - key(b)).abs()
    # rel_floor: differences below rel_floor * mean|b| are exempt too -- outputs that cancel to ~0 carry the
    # absolute fp32 accumulation noise of the whole dot product, which is many ulps of a tiny value
    floor = max(abs_floor, rel_floor * float(bf.abs().mean()))
    small = (af - bf).abs() <= floor
    d = torch.where(small, torch.zeros_like(d), d)
    mx, frac = int(d.max()), float((d > 0).float().mean())
    assert mx <= max_ulp, f"{what}: max ulp diff {mx} (> {max_ulp}); max abs diff {(af - bf).abs().max().item():.4g}"
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ (> {max_frac})"


def common_prefix(a, b) -> int:
    n = 0
    for x, y in zip(a, b):
        if x != y:
            break
        n += 1
    return n


def seq_margins(margin_log: dict, index: int) -> dict:
    """{position: margin} of the index-th sequence (by seq_id order, the order generate() returns outputs in) out of a
    runner's margin_log {(seq_id, position): top-2 margin}."""
    sids = sorted({sid for sid, _ in margin_log})
    return {pos: m for (sid, pos), m in margin_log.items() if sid == sids[index]}


NEAR_TIE = 0.0625


def assert_stream_matches(got, want, margins: dict, prompt_len: int, what: str = "", thr: float = NEAR_TIE) -> int:
    """Greedy streams must be IDENTICAL to the end, unless the reference's own top-2 margin at the first differing
    decision is below `thr` (a near-tie that a different fp32 accumulation order may legitimately flip; after such a flip
    the two runs see different inputs and are no longer comparable).  Returns the common prefix length."""
    assert len(got) == len(want), f"{what}: lengths differ ({len(got)} vs {len(want)})"
    n = common_prefix(got, want)
    if n < len(want):
        m = margins.get(prompt_len + n)
        assert m is not None, f"{what}: diverged at token {n} and the reference recorded no margin there"
        # (bf16 logits: at magnitude 8-16 one ulp IS 0.0625, the smallest non-zero margin there -- inclusive bound)
        assert m <= thr, f"{what}: diverged at token {n} although the reference margin there is {m:.4f} (> {thr})"
    return n


def truth_forward(cfg, w: dict, tokens: list[int]) -> torch.Tensor:
    """Exact-arithmetic (float64) Llama / Qwen3 forward over one full sequence with causal attention: logits at every
    position.  The bf16 weights are upcast; NOTHING is rounded in between -- the "truth" both the reference's bf16
    pipeline and the HIP pipeline approximate.  Mathematically equal to prefill + cached decode / verify."""
    D = torch.float64
    T = len(tokens)
    h = w["model.embed_tokens.weight"].to(D)[torch.tensor(tokens)]
    hd, nh, nkv = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=D) / hd))
    fr = torch.arange(T, dtype=D)[:, None] * inv[None, :]
    cos, sin = fr.cos()[:, None, :], fr.sin()[:, None, :]

    def norm(x, wt, eps):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * wt.to(D)

    def rot(x):
        x1, x2 = x.chunk(2, -1)
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1)

    mask = torch.full((T, T), float("-inf"), dtype=D).triu(1)
    res = None
    for li in range(cfg.num_layers):
        p = f"model.layers.{li}."
        res = h if res is None else h + res
        x = norm(res, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        qkv = x @ w[p + "self_attn.qkv_proj.weight"].to(D).t()
        if p + "self_attn.qkv_proj.bias" in w:
            qkv = qkv + w[p + "self_attn.qkv_proj.bias"].to(D)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
        if cfg.qk_norm:
            q = norm(q, w[p + "self_attn.q_norm.weight"], cfg.rms_norm_eps)
            k = norm(k, w[p + "self_attn.k_norm.weight"], cfg.rms_norm_eps)
        q, k = rot(q), rot(k)
        g = nh // nkv
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        s = torch.einsum("qhd,khd->hqk", q, k) * hd ** -0.5 + mask
        o = torch.einsum("hqk,khd->qhd", s.softmax(-1), v).reshape(T, nh * hd)
        h = o @ w[p + "self_attn.o_proj.weight"].to(D).t()
        res = h + res
        x = norm(res, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        gu = x @ w[p + "mlp.gate_up_proj.weight"].to(D).t()
        a, b = gu.chunk(2, -1)
        h = (a * torch.sigmoid(a) * b) @ w[p + "mlp.down_proj.weight"].to(D).t()
    x = norm(h + res, w["model.norm.weight"], cfg.rms_norm_eps)
    head = w["model.embed_tokens.weight"] if cfg.tie_word_embeddings else w["lm_head.weight"]
    return x @ head.to(D).t()


def truth_forward_masked(cfg, w: dict, tokens: list[int], positions: list[int], visible: torch.Tensor) -> torch.Tensor:
    """truth_forward with explicit RoPE positions and an explicit visibility matrix (bool [T][T], row = query): the exact-arithmetic
    (float64, nothing rounded) logits of EVERY row of a speculation tree in one pass -- prompt + glue rows causal, tree rows under the
    structural mask of ssd/engine/helpers/mask_helpers.py:12-21 (oracle/ops.py tree_mask) -- mathematically equal to prefill + glue
    decode + K cached tree steps.  One layer's fp64 weights at a time."""
    D = torch.float64
    T = len(tokens)
    h = w["model.embed_tokens.weight"][torch.tensor(tokens)].to(D)
    hd, nh, nkv = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=D) / hd))
    fr = torch.tensor(positions, dtype=D)[:, None] * inv[None, :]
    cos, sin = fr.cos()[:, None, :], fr.sin()[:, None, :]

    def norm(x, wt, eps):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * wt.to(D)

    def rot(x):
        x1, x2 = x.chunk(2, -1)
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1)

    mask = torch.zeros(T, T, dtype=D).masked_fill(~visible, float("-inf"))
    res = None
    for li in range(cfg.num_layers):
        p = f"model.layers.{li}."
        res = h if res is None else h + res
        x = norm(res, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        qkv = x @ w[p + "self_attn.qkv_proj.weight"].to(D).t()
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
        if cfg.qk_norm:
            q = norm(q, w[p + "self_attn.q_norm.weight"], cfg.rms_norm_eps)
            k = norm(k, w[p + "self_attn.k_norm.weight"], cfg.rms_norm_eps)
        q, k = rot(q), rot(k)
        g = nh // nkv
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        s = torch.einsum("qhd,khd->hqk", q, k) * hd ** -0.5 + mask
        o = torch.einsum("hqk,khd->qhd", s.softmax(-1), v).reshape(T, nh * hd)
        h = o @ w[p + "self_attn.o_proj.weight"].to(D).t()
        res = h + res
        x = norm(res, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        gu = x @ w[p + "mlp.gate_up_proj.weight"].to(D).t()
        a, b = gu.chunk(2, -1)
        h = (a * torch.sigmoid(a) * b) @ w[p + "mlp.down_proj.weight"].to(D).t()
    x = norm(h + res, w["model.norm.weight"], cfg.rms_norm_eps)
    head = w["model.embed_tokens.weight"] if cfg.tie_word_embeddings else w["lm_head.weight"]
    return x @ head.to(D).t()


def eagle_models_from_golden(g):
    """(target ModelConfig, target weights, draft ModelConfig (family eagle3), draft weights incl. d2t, taps, K, F) of
    tests/golden/tiny_eagle3.npz (written by the reference's own LlamaForCausalLM / Eagle3DraftForCausalLM)."""
    from ssd_amd.model_config import ModelConfig

    def cfg(prefix, family, **kw):
        ci, cf = g[prefix + "cfg_i"].tolist(), g[prefix + "cfg_f"].tolist()
        return ModelConfig(family, ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], cf[0], cf[1], ci[7], False, **kw)

    tcfg = cfg("t_", "llama")
    tw = {k[2:]: v for k, v in g.items() if k.startswith("t.")}
    dw = {k[2:]: v for k, v in g.items() if k.startswith("d.")}
    dcfg = cfg("d_", "eagle3", draft_vocab_size=int(dw["lm_head.weight"].shape[0]), d_model_target=tcfg.hidden_size,
               eagle_taps=int(g["taps"].numel()))
    K, F = g["K_F"].tolist()
    return tcfg, tw, dcfg, dw, g["taps"].tolist(), K, F
