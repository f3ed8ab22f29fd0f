"""Worker of tests/test_tp_one_gpu.py: one tensor-parallel rank of the HIP engine; both ranks share GPU 0 (gloo for
the control collectives, the one-shot all-reduce for the in-forward sums, eager launches)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ssd_amd.engine.llm_engine import LLMEngine  # noqa: E402
from ssd_amd.model_config import ModelConfig  # noqa: E402
from ssd_amd.sampling_params import SamplingParams  # noqa: E402


def main():
    tp = int(os.environ.get("WORLD_SIZE", "1"))
    t = ModelConfig("llama", 256, 2, 4, 2, 64, 512, 512, 1e-5, 5e5, 1024, False)
    d = ModelConfig("llama", 128, 1, 2, 1, 64, 256, 512, 1e-5, 5e5, 1024, True)
    eng = LLMEngine("t", hf_config=t, draft="d", draft_hf_config=d, speculate=True, speculate_k=3, num_gpus=tp, max_num_seqs=2,
                    max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40,
                    num_draft_kvcache_blocks=40, weights_std=0.1, enforce_eager=True)
    prompts = [[(5 * i + 3 * j) % 512 for j in range(9 + 2 * i)] for i in range(2)]
    if tp == 1:
        eng.model_runner.margin_log = {}
    out, m = eng.generate(prompts, SamplingParams(temperature=0, max_new_tokens=16, ignore_eos=True), use_tqdm=False)
    ar = eng.model_runner.model.custom_ar is not None
    margins = []
    if tp == 1:
        from tests.util import seq_margins
        margins = [{str(k): v for k, v in seq_margins(eng.model_runner.margin_log, i).items()} for i in range(len(prompts))]
    print("RESULT " + json.dumps({"tokens": [o["token_ids"] for o in out], "custom_ar": ar, "margins": margins,
                                  "prompt_lens": [len(p) for p in prompts],
                                  "lens": m["accepted_suffix_lens_with_recovery"]}), flush=True)
    if tp > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
