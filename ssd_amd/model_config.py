"""Architecture description shared by the HIP decoder and the test oracle.

The reference reads these fields from a HF ``config.json`` through ``AutoConfig`` (ssd/config.py:56-61) and
consumes them in ``LlamaAttention`` / ``Qwen3Attention`` (ssd/models/llama3.py:15-87, ssd/models/qwen3.py).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass
class ModelConfig:
    family: str                 # "llama" | "qwen3" | "eagle3" (the one-layer EAGLE-3 draft, eagle3_draft_llama3.py)
    hidden_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    tie_word_embeddings: bool = False
    qk_norm: bool = False       # Qwen3: per-head RMSNorm on q and k before RoPE (qwen3.py:96-104)
    attention_bias: bool = False
    # EAGLE-3 draft only (ssd/models/eagle3_draft_llama3.py:209-262): the LM head covers draft_vocab_size tokens that d2t maps
    # into the target vocabulary (vocab_size), and fc projects eagle_taps concatenated target activations of width
    # d_model_target each
    draft_vocab_size: int = 0
    d_model_target: int = 0
    eagle_taps: int = 3

    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    def to_dict(self) -> dict:
        return asdict(self)

    @staticmethod
    def from_hf(hf) -> "ModelConfig":
        """From a transformers config object (or anything with the same attributes)."""
        mt = getattr(hf, "model_type", "llama")
        family = "qwen3" if "qwen" in mt else "llama"
        if getattr(hf, "draft_vocab_size", None) and hf.num_hidden_layers == 1:
            family = "eagle3"       # EAGLE-3 checkpoints: a Llama config.json with one layer + draft_vocab_size
        nh = hf.num_attention_heads
        hd = getattr(hf, "head_dim", None) or hf.hidden_size // nh
        # transformers >= 5 moved rope_theta into rope_parameters; the reference's getattr(config,
        # "rope_theta", default) (llama3.py:163, qwen3.py:173) would silently fall back, so read both.
        theta = getattr(hf, "rope_theta", None)
        if theta is None:
            rp = getattr(hf, "rope_parameters", None) or {}
            theta = rp.get("rope_theta") if isinstance(rp, dict) else None
        if theta is None:
            theta = 1000000.0 if family == "qwen3" else 500000.0
        return ModelConfig(
            family=family,
            hidden_size=hf.hidden_size,
            num_layers=hf.num_hidden_layers,
            num_heads=nh,
            num_kv_heads=getattr(hf, "num_key_value_heads", nh),
            head_dim=hd,
            intermediate_size=hf.intermediate_size,
            vocab_size=hf.vocab_size,
            rms_norm_eps=getattr(hf, "rms_norm_eps", 1e-6),
            rope_theta=float(theta),
            max_position_embeddings=getattr(hf, "max_position_embeddings", 8192),
            tie_word_embeddings=bool(getattr(hf, "tie_word_embeddings", False)),
            qk_norm=(family == "qwen3"),
            attention_bias=bool(getattr(hf, "attention_bias", False)),
            draft_vocab_size=int(getattr(hf, "draft_vocab_size", 0) or 0),
        )


# Public shapes of the models BASELINE.json names (no weights exist offline).
PRESETS = {
    "llama-3.2-1b": ModelConfig("llama", 2048, 16, 32, 8, 64, 8192, 128256, 1e-5, 5e5, 131072, True),
    "llama-3.1-8b": ModelConfig("llama", 4096, 32, 32, 8, 128, 14336, 128256, 1e-5, 5e5, 131072, False),
    "llama-3.1-70b": ModelConfig("llama", 8192, 80, 64, 8, 128, 28672, 128256, 1e-5, 5e5, 131072, False),
    "qwen3-32b": ModelConfig("qwen3", 5120, 64, 64, 8, 128, 25600, 151936, 1e-6, 1e6, 40960, False, True),
    "qwen3-0.6b": ModelConfig("qwen3", 1024, 28, 16, 8, 128, 3072, 151936, 1e-6, 1e6, 40960, True, True),
    # the other sizes the reference's bench.py --size accepts (bench_helpers.py:96-120), public config.json shapes
    "llama-3.2-3b": ModelConfig("llama", 3072, 28, 24, 8, 128, 8192, 128256, 1e-5, 5e5, 131072, True),
    "qwen3-1.7b": ModelConfig("qwen3", 2048, 28, 16, 8, 128, 6144, 151936, 1e-6, 1e6, 40960, True, True),
    "qwen3-4b": ModelConfig("qwen3", 2560, 36, 32, 8, 128, 9728, 151936, 1e-6, 1e6, 40960, True, True),
    "qwen3-8b": ModelConfig("qwen3", 4096, 36, 32, 8, 128, 12288, 151936, 1e-6, 1e6, 40960, False, True),
    "qwen3-14b": ModelConfig("qwen3", 5120, 40, 40, 8, 128, 17408, 151936, 1e-6, 1e6, 40960, False, True),
    # EAGLE-3 drafts the reference's bench.py --eagle selects (bench_helpers.py:50-63, bench_paths.py:33-44); public config.json shapes
    "eagle3-llama-3.1-8b": ModelConfig("eagle3", 4096, 1, 32, 8, 128, 14336, 128256, 1e-5, 5e5, 131072, False,
                                       draft_vocab_size=32000, d_model_target=4096),
    "eagle3-llama-3.3-70b": ModelConfig("eagle3", 6144, 1, 48, 8, 128, 16384, 128256, 1e-5, 5e5, 131072, False,
                                        draft_vocab_size=32000, d_model_target=8192),
}
