"""Weights: reference parameter names, TP sharding, synthetic generation and safetensors loading.

Parameter names are the reference's packed names (ssd/models/llama3.py:277-283, ssd/utils/loader.py:186-218):
  model.embed_tokens.weight [V,h]; model.layers.{i}.input_layernorm.weight; .self_attn.qkv_proj.weight
  [(nh+2nkv)hd, h] (+ .bias, + q_norm/k_norm for Qwen3); .self_attn.o_proj.weight [h, nh*hd];
  .post_attention_layernorm.weight; .mlp.gate_up_proj.weight [2I, h]; .mlp.down_proj.weight [h, I];
  model.norm.weight; lm_head.weight [V,h] (absent when tied).
Sharding follows the reference weight loaders (ssd/layers/linear.py:90-95,116-122,148-162,188-193;
ssd/layers/embed_head.py:41-47).

No weights exist offline, so the default source is synthetic: every FULL tensor is a deterministic function
of (seed, name) -- independent of rank -- and is sharded afterwards, so TP=N computes the same function as TP=1.
"""
from __future__ import annotations

import glob
import hashlib
import os
from typing import Iterator

import torch

from ssd_amd.model_config import ModelConfig

BF16 = torch.bfloat16


def eagle_param_shapes(cfg: ModelConfig) -> list[tuple[str, tuple[int, ...]]]:
    """Eagle3DraftForCausalLM's parameters under the reference's module names (eagle3_draft_llama3.py:101-140,159-194,
    209-262): QKV reads the 2h-wide [token | conditioning] concatenation, the LM head covers draft_vocab_size tokens and
    ``d2t`` (int64, target id = draft id + d2t[draft id], :325-327) places them in the target vocabulary."""
    h, hd, nh, nkv, I = cfg.hidden_size, cfg.head_dim, cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size
    p = "model.layer."
    return [("model.embed_tokens.weight", (cfg.vocab_size, h)),
            ("fc.weight", (h, cfg.eagle_taps * cfg.d_model_target)),
            (p + "input_layernorm.weight", (h,)),
            (p + "conditioning_feature_ln.weight", (h,)),
            (p + "self_attn.qkv_proj.weight", ((nh + 2 * nkv) * hd, 2 * h)),
            (p + "self_attn.o_proj.weight", (h, nh * hd)),
            (p + "post_attention_layernorm.weight", (h,)),
            (p + "mlp.gate_up_proj.weight", (2 * I, h)),
            (p + "mlp.down_proj.weight", (h, I)),
            ("final_norm.weight", (h,)),
            ("lm_head.weight", (cfg.draft_vocab_size, h)),
            ("d2t", (cfg.draft_vocab_size,))]


def param_shapes(cfg: ModelConfig) -> list[tuple[str, tuple[int, ...]]]:
    if cfg.family == "eagle3":
        return eagle_param_shapes(cfg)
    h, hd, nh, nkv, I, V = cfg.hidden_size, cfg.head_dim, cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size, cfg.vocab_size
    out = [("model.embed_tokens.weight", (V, h))]
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        out.append((p + "input_layernorm.weight", (h,)))
        out.append((p + "self_attn.qkv_proj.weight", ((nh + 2 * nkv) * hd, h)))
        if cfg.attention_bias:
            out.append((p + "self_attn.qkv_proj.bias", ((nh + 2 * nkv) * hd,)))
        if cfg.qk_norm:
            out.append((p + "self_attn.q_norm.weight", (hd,)))
            out.append((p + "self_attn.k_norm.weight", (hd,)))
        out.append((p + "self_attn.o_proj.weight", (h, nh * hd)))
        out.append((p + "post_attention_layernorm.weight", (h,)))
        out.append((p + "mlp.gate_up_proj.weight", (2 * I, h)))
        out.append((p + "mlp.down_proj.weight", (h, I)))
    out.append(("model.norm.weight", (h,)))
    if not cfg.tie_word_embeddings:
        out.append(("lm_head.weight", (V, h)))
    return out


def _name_seed(seed: int, name: str) -> int:
    return int.from_bytes(hashlib.blake2b(f"{seed}:{name}".encode(), digest_size=7).digest(), "little")


def _pair_tensor(name: str, shape, seed: int, std: float, gen_device: str, recipe: dict) -> torch.Tensor | None:
    """The "correlated pair" recipe: a target and a draft of DIFFERENT shapes that agree on most greedy tokens by
    construction, so that speculation runs at a realistic acceptance rate without trained checkpoints.
      * both models embed a token with the same base vectors E_b[V, ds] (ds = recipe["shared"], the draft's hidden
        size) and read it out with the same base head H_b[V, ds]; a wider model fills its remaining hidden dims with
        independent noise, down-weighted through the gain on the shared dims: logits_target = snr * signal + noise,
        logits_draft = signal (recipe["snr"]; Monte Carlo at V = 128256: snr 4 -> 45 % top-1 agreement, 8 -> ~65 %);
      * o_proj / down_proj are scaled by recipe["layer_gain"], so the decoder layers perturb the residual stream
        (context-dependent, independent between the two models) instead of drowning the embedding.
    Every matrix keeps its real shape and is streamed in full; only the VALUES differ from the plain N(0, std) recipe."""
    ds, snr, lg = int(recipe["shared"]), float(recipe.get("snr", 8.0)), float(recipe.get("layer_gain", 0.005))
    pseed = int(recipe.get("seed", 1234))

    def randn(tag, shp):
        g = torch.Generator(device=gen_device)
        g.manual_seed(_name_seed(pseed, tag))
        return torch.randn(shp, generator=g, device=gen_device, dtype=torch.float32)

    if name in ("model.embed_tokens.weight", "lm_head.weight"):
        V, h = shape
        assert h >= ds
        is_embed = name.startswith("model.embed")
        base = randn("pair.embed" if is_embed else "pair.head", (V, ds))
        if h == ds:
            return (base * (1.0 if is_embed else std)).to(BF16)
        gain = snr * ((h - ds) / ds) ** 0.5 if is_embed else 1.0       # logit SNR = gain * sqrt(ds / (h - ds))
        rest = randn(f"pair.rest.{seed}.{name}", (V, h - ds))
        out = torch.cat([base * gain, rest], dim=1)
        return (out * (1.0 if is_embed else std)).to(BF16)
    if name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
        g = torch.Generator(device=gen_device)
        g.manual_seed(_name_seed(seed, name))
        return (std * lg * torch.randn(shape, generator=g, device=gen_device, dtype=torch.float32)).to(BF16)
    return None


def eagle_pair_recipe(target: ModelConfig, draft: ModelConfig, draft_seed: int, snr: float = 8.0, layer_gain: float = 0.005,
                      boost: float = 2.0, seed: int = 1234) -> dict:
    """The recipe dict (config.weights_recipe) of the constructed target + EAGLE-3 draft pair: see _eagle_pair_tensor."""
    assert draft.family == "eagle3" and draft.d_model_target == target.hidden_size
    sig = min(draft.num_kv_heads * draft.head_dim, target.hidden_size, draft.hidden_size)
    return {"kind": "eagle_pair", "sig": sig, "snr": snr, "layer_gain": layer_gain, "boost": boost, "seed": seed,
            "draft_seed": draft_seed, "draft_vocab": draft.draft_vocab_size, "target_vocab": target.vocab_size,
            "target_hidden": target.hidden_size, "draft_heads": (draft.num_heads, draft.num_kv_heads, draft.head_dim)}


def eagle_pair_gain(recipe: dict) -> float:
    """Gain of the shared embedding dims inside the TARGET's embedding rows (logit SNR = recipe["snr"], as in _pair_tensor)."""
    ds, ht = int(recipe["sig"]), int(recipe["target_hidden"])
    return float(recipe.get("snr", 8.0)) * ((ht - ds) / ds) ** 0.5 if ht > ds else 1.0


def _eagle_pair_tensor(name: str, shape, seed: int, std: float, gen_device: str, recipe: dict) -> torch.Tensor | None:
    """A target and its EAGLE-3 draft that AGREE by construction (round 3; the "peaky" pair only made both heads favour the same
    three tokens).  The target is the correlated-pair target of _pair_tensor over ds = recipe["sig"] shared dims: it embeds a
    token as [gain * E_b[tok] | noise], reads out with [H_b | noise], its layers perturb the residual stream only slightly
    (layer_gain), so its greedy next token is mostly g(tok) = argmax_v H_b[v] . E_b[tok]; the head rows of the tokens the draft
    vocabulary contains are scaled by recipe["boost"], so that the argmax falls inside that vocabulary.  The one-layer draft is
    wired to compute the same g from the token it is fed, whatever its conditioning row holds:
      * embedding [E_b[tok] | 0]; both input norms carry the weight sqrt(ds / h), so a row [x | ~0] comes out as ~[x | 0];
      * q / k = beta * A_kvhead . E_b[tok] (the conditioning half of the 2h-wide input is ignored): a row's score with ITSELF is
        beta^2 * sqrt(hd), with any other row ~N(0, beta^4) (RoPE leaves same-position dot products unchanged), so the softmax is
        one-hot on the row itself;
      * v = (token half) - (conditioning half) on the ds shared dims (ds = nkv * hd: one 128-wide slice per kv head), o_proj puts
        the slice of the first q head of each group back in place: attention output = E_b[tok] - conditioning, and the residual
        add of the conditioning leaves [E_b[tok] | ~0] whatever the conditioning was (fc(target taps) or the previous prenorm);
      * the MLP is scaled down like the target's, the head is [H_b[d2t] | 0]: logits = H_b . E_b[tok] -> g(tok) within the draft
        vocabulary.  fc copies the shared dims of the first tapped activation (divided by the target's gain), i.e. the first
        conditioning row has the same form as a prenorm.
    Every matrix keeps its real shape and is streamed in full; only the values are constructed."""
    ds = int(recipe["sig"])
    pseed = int(recipe.get("seed", 1234))
    lg = float(recipe.get("layer_gain", 0.005))
    V, Vd, ht = int(recipe["target_vocab"]), int(recipe["draft_vocab"]), int(recipe["target_hidden"])
    is_draft = recipe.get("family") == "eagle3"

    def randn(tag, shp):
        g = torch.Generator(device=gen_device)
        g.manual_seed(_name_seed(pseed, tag))
        return torch.randn(shp, generator=g, device=gen_device, dtype=torch.float32)

    def own(shp):
        g = torch.Generator(device=gen_device)
        g.manual_seed(_name_seed(seed, name))
        return torch.randn(shp, generator=g, device=gen_device, dtype=torch.float32)

    def draft_targets():            # target-vocabulary id of every draft-vocabulary id (d2t of the draft seed)
        gc = torch.Generator()
        gc.manual_seed(_name_seed(int(recipe["draft_seed"]), "d2t"))
        return torch.randperm(V, generator=gc)[:Vd].sort().values

    if (name.endswith("o_proj.weight") and not is_draft) or name.endswith("down_proj.weight"):
        return (std * lg * own(shape)).to(BF16)
    if not is_draft:
        if name == "model.embed_tokens.weight":
            h = shape[1]
            assert shape[0] == V and h == ht and h >= ds
            base = randn("epair.embed", (V, ds)) * eagle_pair_gain(recipe)
            return (base if h == ds else torch.cat([base, own((V, h - ds))], dim=1)).to(BF16)
        if name == "lm_head.weight":
            h = shape[1]
            base = randn("epair.head", (V, ds))
            out = base if h == ds else torch.cat([base, own((V, h - ds))], dim=1)
            out[draft_targets().to(out.device)] *= float(recipe.get("boost", 2.0))
            return (out * std).to(BF16)
        return None
    # ---- the EAGLE-3 draft ----
    nh, nkv, hd = (int(x) for x in recipe["draft_heads"])
    assert ds <= nkv * hd
    if name == "model.embed_tokens.weight":
        h = shape[1]
        out = torch.zeros(shape, dtype=torch.float32, device=gen_device)
        out[:, :ds] = randn("epair.embed", (V, ds))
        return out.to(BF16)
    if name == "lm_head.weight":
        out = torch.zeros(shape, dtype=torch.float32, device=gen_device)
        out[:, :ds] = randn("epair.head", (V, ds))[draft_targets().to(gen_device)]
        return (out * std).to(BF16)
    if name == "fc.weight":
        out = torch.zeros(shape, dtype=torch.float32, device=gen_device)
        idx = torch.arange(ds, device=gen_device)
        out[idx, idx] = 1.0 / eagle_pair_gain(recipe)          # first tap, shared dims
        return out.to(BF16)
    if name.endswith("input_layernorm.weight") or name.endswith("conditioning_feature_ln.weight"):
        return torch.full(shape, (ds / shape[0]) ** 0.5, dtype=torch.float32, device=gen_device).to(BF16)
    if name.endswith("qkv_proj.weight"):
        h = shape[1] // 2
        out = torch.zeros(shape, dtype=torch.float32, device=gen_device)
        beta = (16.0 / max(hd ** 0.5 - 3.5, 1.0)) ** 0.5
        grp = nh // nkv
        for k in range(nkv):
            A = randn(f"epair.A.{k}", (hd, ds)) * (beta / ds ** 0.5)
            for i in range(k * grp, (k + 1) * grp):
                out[i * hd:(i + 1) * hd, :ds] = A                                   # q heads of the group
            out[(nh + k) * hd:(nh + k + 1) * hd, :ds] = A                           # its k head
            r = torch.arange(min(hd, max(ds - k * hd, 0)), device=gen_device)
            rows = (nh + nkv + k) * hd + r                                          # its v head: token half - conditioning half
            out[rows, k * hd + r] = 1.0
            out[rows, h + k * hd + r] = -1.0
        return out.to(BF16)
    if name.endswith("o_proj.weight"):
        out = torch.zeros(shape, dtype=torch.float32, device=gen_device)
        grp = nh // nkv
        for k in range(nkv):
            r = torch.arange(min(hd, max(ds - k * hd, 0)), device=gen_device)
            out[k * hd + r, (k * grp) * hd + r] = 1.0
        return out.to(BF16)
    return None


def peaky_rows(recipe: dict) -> tuple[torch.Tensor, torch.Tensor]:
    """(draft-vocabulary ids, target-vocabulary ids) of the few tokens whose LM-head rows the "peaky" recipe scales in BOTH a
    target and its EAGLE-3 draft.  Random models never agree, and without agreement the speculation-cache hit path, partial
    acceptance and the extend rows of the EAGLE glue never run; with both heads favouring the same few tokens each model picks
    one of them most of the time and the same one about a third of the time (tests/eagle_util.py uses the same construction).
    The target ids are the draft ids pushed through the draft's d2t map, which is a function of (draft seed, vocabularies)."""
    Vd, V = int(recipe["draft_vocab"]), int(recipe["target_vocab"])
    gc = torch.Generator()
    gc.manual_seed(_name_seed(int(recipe["draft_seed"]), "d2t"))
    tgt = torch.randperm(V, generator=gc)[:Vd].sort().values          # = arange(Vd) + d2t of synthetic_tensor("d2t", ...)
    gp = torch.Generator()
    gp.manual_seed(int(recipe.get("seed", 99)))
    picks = torch.randperm(Vd, generator=gp)[:int(recipe.get("peaks", 3))]
    return picks, tgt[picks]


def synthetic_tensor(name: str, shape, seed: int, std: float, gen_device: str, norm_jitter: float = 0.0,
                     recipe: dict | None = None) -> torch.Tensor:
    """Full (unsharded) synthetic parameter.  gen_device="cpu" gives values reproducible on any machine (tests,
    oracle comparisons); "cuda" is for the multi-GB benchmark models."""
    if recipe is not None and recipe.get("kind") == "pair":
        t = _pair_tensor(name, shape, seed, std, gen_device, recipe)
        if t is not None:
            return t
    if recipe is not None and recipe.get("kind") == "eagle_pair":
        t = _eagle_pair_tensor(name, shape, seed, std, gen_device, recipe)
        if t is not None:
            return t
    g = torch.Generator(device=gen_device)
    g.manual_seed(_name_seed(seed, name))
    if name == "d2t":       # an increasing map of the draft vocabulary into the target's (like the frequency-sorted real ones)
        V = int(recipe["target_vocab"]) if recipe and "target_vocab" in recipe else None
        assert V is not None and V >= shape[0], "d2t needs recipe['target_vocab']"
        gc = torch.Generator()
        gc.manual_seed(_name_seed(seed, name))
        idx = torch.randperm(V, generator=gc)[:shape[0]].sort().values
        return (idx - torch.arange(shape[0])).to(torch.int64).to(gen_device)
    if "norm" in name or name.endswith("_ln.weight"):
        if norm_jitter == 0.0:
            return torch.ones(shape, dtype=BF16, device=gen_device)
        return (1.0 + norm_jitter * torch.randn(shape, generator=g, device=gen_device, dtype=torch.float32)).to(BF16)
    t = std * torch.randn(shape, generator=g, device=gen_device, dtype=torch.float32)
    if recipe is not None and recipe.get("kind") == "peaky" and name == "lm_head.weight":
        dpicks, tpicks = peaky_rows(recipe)
        rows = dpicks if shape[0] == int(recipe["draft_vocab"]) else tpicks
        t[rows.to(t.device)] *= float(recipe.get("gain", 6.0))
    return t.to(BF16)


def shard_param(cfg: ModelConfig, name: str, w: torch.Tensor, rank: int, tp: int) -> torch.Tensor:
    if tp == 1:
        return w
    hd, nh, nkv, I = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size
    if name.endswith("qkv_proj.weight") or name.endswith("qkv_proj.bias"):
        q, k, v = w.split([nh * hd, nkv * hd, nkv * hd], dim=0)
        return torch.cat([q.chunk(tp, 0)[rank], k.chunk(tp, 0)[rank], v.chunk(tp, 0)[rank]], 0).contiguous()
    if name.endswith("gate_up_proj.weight"):
        g, u = w.split([I, I], dim=0)
        return torch.cat([g.chunk(tp, 0)[rank], u.chunk(tp, 0)[rank]], 0).contiguous()
    if name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
        return w.chunk(tp, 1)[rank].contiguous()
    if name.endswith("embed_tokens.weight") or name.endswith("lm_head.weight"):
        return w.chunk(tp, 0)[rank].contiguous()
    return w


def synthetic_weights(cfg: ModelConfig, seed: int, std: float, rank: int = 0, tp: int = 1, gen_device: str = "cpu",
                      out_device: str | None = None, norm_jitter: float = 0.0,
                      recipe: dict | None = None) -> Iterator[tuple[str, torch.Tensor]]:
    """Yields (name, this rank's shard) one tensor at a time (bounded transient memory for 70B)."""
    if cfg.family == "eagle3":          # d2t needs the target vocabulary size; the draft is never tensor-parallel
        assert tp == 1
        recipe = dict(recipe or {}, target_vocab=cfg.vocab_size)
        if recipe.get("kind") == "eagle_pair":
            assert int(recipe["draft_seed"]) == seed and int(recipe["draft_vocab"]) == cfg.draft_vocab_size, "eagle_pair recipe: draft seed / vocabulary mismatch"
        if recipe.get("kind") == "peaky":
            assert int(recipe["draft_seed"]) == seed and int(recipe["draft_vocab"]) == cfg.draft_vocab_size, "peaky recipe: draft seed / vocabulary mismatch"
    if recipe is not None and recipe.get("kind") == "eagle_pair":
        recipe = dict(recipe, family=cfg.family)
    for name, shape in param_shapes(cfg):
        w = shard_param(cfg, name, synthetic_tensor(name, shape, seed, std, gen_device, norm_jitter, recipe), rank, tp)
        yield name, (w.to(out_device) if out_device is not None else w)


def synthetic_state_dict(cfg: ModelConfig, seed: int, std: float, norm_jitter: float = 0.0, recipe: dict | None = None) -> dict:
    return dict(synthetic_weights(cfg, seed, std, gen_device="cpu", norm_jitter=norm_jitter, recipe=recipe))


# --------------------------------------------------------------------------------------------------
# HF safetensors (reference ssd/utils/loader.py:186-218): q/k/v and gate/up are packed by concatenation
# --------------------------------------------------------------------------------------------------


def has_safetensors(model_dir: str) -> bool:
    return os.path.isdir(model_dir) and bool(glob.glob(os.path.join(model_dir, "*.safetensors")))


def load_eagle_safetensors(cfg: ModelConfig, model_dir: str, target_dir: str | None = None,
                           out_device: str | None = None) -> Iterator[tuple[str, torch.Tensor]]:
    """EAGLE-3 checkpoints (reference load_eagle_model, ssd/utils/loader.py:64-183): a flat state dict with
    midlayer.* (q/k/v and gate/up unpacked, hidden_norm = the conditioning-feature norm), norm.weight (final norm),
    fc.weight, lm_head.weight, d2t / t2d, and embed_tokens.weight -- taken from the TARGET checkpoint when the draft
    ships none (:118-125, load_embedding_from_target :9-61)."""
    from safetensors import safe_open
    sd: dict[str, torch.Tensor] = {}
    for f in sorted(glob.glob(os.path.join(model_dir, "*.safetensors"))):
        with safe_open(f, "pt", "cpu") as sf:
            for k in sf.keys():
                sd[k] = sf.get_tensor(k)
    if "embed_tokens.weight" not in sd:
        assert target_dir is not None and has_safetensors(target_dir), "EAGLE-3 draft without embed_tokens needs the target checkpoint"
        for f in sorted(glob.glob(os.path.join(target_dir, "*.safetensors"))):
            with safe_open(f, "pt", "cpu") as sf:
                for k in sf.keys():
                    if k.endswith("embed_tokens.weight"):
                        sd["embed_tokens.weight"] = sf.get_tensor(k)
        assert "embed_tokens.weight" in sd, f"no embed_tokens.weight under {target_dir}"
    ml = "midlayer."
    src = {
        "model.embed_tokens.weight": lambda: sd["embed_tokens.weight"],
        "fc.weight": lambda: sd["fc.weight"],
        "model.layer.input_layernorm.weight": lambda: sd[ml + "input_layernorm.weight"],
        "model.layer.conditioning_feature_ln.weight": lambda: sd[ml + "hidden_norm.weight"],
        "model.layer.self_attn.qkv_proj.weight": lambda: torch.cat([sd[ml + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0),
        "model.layer.self_attn.o_proj.weight": lambda: sd[ml + "self_attn.o_proj.weight"],
        "model.layer.post_attention_layernorm.weight": lambda: sd[ml + "post_attention_layernorm.weight"],
        "model.layer.mlp.gate_up_proj.weight": lambda: torch.cat([sd[ml + "mlp.gate_proj.weight"], sd[ml + "mlp.up_proj.weight"]], 0),
        "model.layer.mlp.down_proj.weight": lambda: sd[ml + "mlp.down_proj.weight"],
        "final_norm.weight": lambda: sd["norm.weight"],
        "lm_head.weight": lambda: sd["lm_head.weight"],
        "d2t": lambda: sd["d2t"],
    }
    for name, shape in eagle_param_shapes(cfg):
        w = src[name]()
        w = w.to(torch.int64) if name == "d2t" else w.to(BF16)
        assert tuple(w.shape) == tuple(shape), f"{name}: {tuple(w.shape)} != {shape}"
        yield name, (w.to(out_device) if out_device is not None else w)


def load_safetensors(cfg: ModelConfig, model_dir: str, rank: int = 0, tp: int = 1,
                     out_device: str | None = None) -> Iterator[tuple[str, torch.Tensor]]:
    from safetensors import safe_open
    index: dict[str, str] = {}
    for f in sorted(glob.glob(os.path.join(model_dir, "*.safetensors"))):
        with safe_open(f, "pt", "cpu") as sf:
            for k in sf.keys():
                index[k] = f

    def get(name: str) -> torch.Tensor:
        with safe_open(index[name], "pt", "cpu") as sf:
            return sf.get_tensor(name).to(BF16)

    def packed_sources(name: str) -> list[str] | None:
        if "qkv_proj" in name:
            return [name.replace("qkv_proj", s) for s in ("q_proj", "k_proj", "v_proj")]
        if "gate_up_proj" in name:
            return [name.replace("gate_up_proj", s) for s in ("gate_proj", "up_proj")]
        return None

    for name, shape in param_shapes(cfg):
        srcs = packed_sources(name)
        if srcs is not None and name not in index:
            w = torch.cat([get(s) for s in srcs], dim=0)
        else:
            w = get(name)
        assert tuple(w.shape) == tuple(shape), f"{name}: {tuple(w.shape)} != {shape}"
        w = shard_param(cfg, name, w, rank, tp)
        yield name, (w.to(out_device) if out_device is not None else w)
