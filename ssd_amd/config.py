"""Engine configuration: every keyword ``LLM(...)`` accepts, same names and defaults as the reference Config
(ssd/config.py:7-94).  Unknown kwargs are dropped by the engine exactly as the reference does
(ssd/engine/llm_engine.py:42-44).

Differences forced by the environment (no weights / tokenizers exist offline): ``model`` / ``draft`` may be a
HF-style directory (config.json [+ tokenizer]; read through AutoConfig as the reference does) OR one of the
preset architecture names in ``ssd_amd.model_config.PRESETS``; when the directory holds no *.safetensors the
weights are synthetic (seeded N(0, weights_std), norm weights 1)."""
from __future__ import annotations

import os
from dataclasses import dataclass

from ssd_amd.model_config import ModelConfig, PRESETS


def resolve_model_config(model: str) -> ModelConfig:
    if os.path.isdir(model):
        from transformers import AutoConfig
        return ModelConfig.from_hf(AutoConfig.from_pretrained(model))
    key = model.lower()
    if key in PRESETS:
        return PRESETS[key]
    raise ValueError(f"model must be a directory with config.json or one of {sorted(PRESETS)}; got {model!r}")


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 1
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.7
    num_gpus: int = 1
    enforce_eager: bool = False
    hf_config: ModelConfig | None = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1

    # speculation
    draft_hf_config: ModelConfig | None = None
    speculate: bool = False
    draft: str | None = None
    speculate_k: int = 1
    draft_async: bool = False

    # async speculation only
    async_fan_out: int = 3
    fan_out_list: list[int] | None = None
    fan_out_list_miss: list[int] | None = None
    sampler_x: float | None = None
    jit_speculate: bool = False
    # draft data-parallelism (BASELINE.json configs[4]; on the reference's roadmap only, README.md:129-130): the last
    # num_draft_gpus ranks form a draft group; every member keeps the trunk KV and the glue forward, the MQ_LEN tree
    # branches and the speculation cache are sharded over the members (engine/draft_runner.py)
    num_draft_gpus: int = 1

    # EAGLE-3 draft (reference config.py:36-40,72-92): a one-layer draft conditioned on target activations tapped at
    # eagle_layers; asynchronous + greedy + jit_speculate only, as in the reference (bench.py:83-87, draft_runner.py:42-44)
    use_eagle: bool = False
    eagle_layers: list[int] | None = None
    d_model_target: int | None = None
    tokenizer_path: str | None = None

    verbose: bool = False
    debug_mode: bool = False
    max_steps: int | None = None

    # synthetic-weight controls (not in the reference)
    weights_seed: int = 0
    weights_std: float = 0.02
    draft_weights_seed: int = 1
    num_draft_kvcache_blocks: int = -1      # -1: size from free memory like the reference (draft_runner.py:27)
    weights_recipe: dict | None = None      # e.g. {"kind": "pair", "shared": 2048, "snr": 8}: see weights._pair_tensor

    @property
    def max_blocks(self) -> int:
        return -(-self.max_model_len // self.kvcache_block_size)

    @property
    def MQ_LEN(self) -> int:
        return sum(self.fan_out_list) if self.fan_out_list else 0

    def __post_init__(self):
        assert 1 <= self.num_gpus <= 8, "single node only (reference ssd/config.py:55)"
        assert self.num_draft_gpus >= 1 and (self.num_draft_gpus == 1 or (self.speculate and self.draft_async)), \
            "num_draft_gpus > 1 needs draft_async"
        if self.hf_config is None:
            self.hf_config = resolve_model_config(self.model)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        if self.speculate:
            assert self.draft is not None, "speculate=True needs a draft model"
            if self.draft_hf_config is None:
                self.draft_hf_config = resolve_model_config(self.draft)
            self.max_model_len = min(self.max_model_len, self.draft_hf_config.max_position_embeddings)
            assert self.draft_hf_config.vocab_size == self.hf_config.vocab_size, "draft and target vocab must match"
            if self.use_eagle:
                from dataclasses import replace
                assert self.draft_async, "EAGLE-3 drafts are asynchronous-only (reference bench.py:87, speculator_sync.py:15)"
                assert self.jit_speculate, "EAGLE requires jit_speculate=True (cache misses need draft activations; draft_runner.py:42-44)"
                assert self.num_draft_gpus == 1, "EAGLE-3 drafts are not data-parallel"
                if self.eagle_layers is None:
                    L = self.hf_config.num_layers
                    self.eagle_layers = [2, L // 2, L - 3]          # config.py:73-77
                taps = sorted(set(self.eagle_layers))               # llama3.py:259-262 collects in layer order, once per layer
                assert taps and 0 <= taps[0] and taps[-1] < self.hf_config.num_layers
                self.d_model_target = self.hf_config.hidden_size
                d = self.draft_hf_config
                assert d.num_layers == 1, "an EAGLE-3 draft has exactly one decoder layer (eagle3_draft_llama3.py:192)"
                # the draft takes the target's rope_theta and position range (config.py:78-92)
                self.draft_hf_config = replace(d, family="eagle3", d_model_target=self.d_model_target, eagle_taps=len(taps),
                                               draft_vocab_size=d.draft_vocab_size or d.vocab_size, rope_theta=self.hf_config.rope_theta,
                                               max_position_embeddings=self.hf_config.max_position_embeddings)
                self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
            if self.draft_async:
                if self.fan_out_list is None:
                    self.fan_out_list = [self.async_fan_out] * (self.speculate_k + 1)
                if self.fan_out_list_miss is None:
                    self.fan_out_list_miss = list(self.fan_out_list)
                assert len(self.fan_out_list) == self.speculate_k + 1
                assert len(self.fan_out_list_miss) == self.speculate_k + 1
                # csrc/sample.hip fork_topf_kernel keeps the draft's own token + the picks so far in a 17-entry LDS list
                assert 0 <= min(self.fan_out_list + self.fan_out_list_miss) and max(self.fan_out_list + self.fan_out_list_miss) <= 15, \
                    "fan-out per position must be in 0..15"
                assert sum(self.fan_out_list_miss) == sum(self.fan_out_list), "hit and miss fan-out lists must have the same sum"
        if self.sampler_x is not None:
            assert self.speculate and self.draft_async, "sampler_x requires draft_async (reference model_runner.py:262-263)"
            assert self.sampler_x > 0 and self.async_fan_out + 1 <= 8
        assert self.kvcache_block_size & (self.kvcache_block_size - 1) == 0, "KV block sizes are powers of two (csrc/attention.hip)"
        assert self.kvcache_block_size >= 2 * self.speculate_k + 2, "block size < 2K+2 unsupported (reference llm_engine.py:48-49)"
        assert self.max_num_batched_tokens >= self.max_model_len
