"""Command-line benchmark with the reference's flags (bench/bench.py:15-89 of tanishqkumar/ssd): a user's existing
invocations keep working --

    python bench/bench.py --size 70 --gpus 4 --spec --k 6 --random
    python bench/bench.py --size 70 --gpus 5 --spec --async --k 7 --f 3 --backup jit --random
    python bench/bench.py --qwen --size 32 --spec --async --temp 0.7 --x 0.5 --random

Differences, all forced by the environment rather than by design:
  * weights: `$SSD_HF_CACHE/models--<org>--<name>` snapshots are used when present (safetensors + tokenizer); otherwise
    the model is instantiated from its public config shapes with seeded synthetic weights and prompts are random token
    ids (the reference's own --random mode);
  * datasets: `$SSD_DATASET_DIR/<name>/<name>_data_10000.jsonl` as in ssd/paths.py:57-63 when present, else random ids;
  * --eagle selects the EAGLE-3 draft of the model (bench_helpers.py:50-63: Llama 8B / 70B; async + greedy only, as in
    the reference): its HF cache snapshot when present, else synthetic weights of its public shapes;
  * --wandb is accepted and ignored (no network);
  * multi-GPU: `--gpus N` spawns the other ranks itself (ssd_amd/engine/launcher.py), like the reference.
The driver-facing, roofline-reporting benchmark is ../bench.py; this file is the drop-in CLI.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from random import randint, seed

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd import LLM, SamplingParams  # noqa: E402

LLAMA = {"1": ("Llama-3.2-1B-Instruct", "llama-3.2-1b"), "3": ("Llama-3.2-3B-Instruct", "llama-3.2-3b"),
         "8": ("Llama-3.1-8B-Instruct", "llama-3.1-8b"), "70": ("Llama-3.1-70B-Instruct", "llama-3.1-70b")}
QWEN = {"0.6": ("Qwen3-0.6B", "qwen3-0.6b"), "1.7": ("Qwen3-1.7B", "qwen3-1.7b"), "4": ("Qwen3-4B", "qwen3-4b"),
        "8": ("Qwen3-8B", "qwen3-8b"), "14": ("Qwen3-14B", "qwen3-14b"), "32": ("Qwen3-32B", "qwen3-32b")}
EAGLE = {"8": ("models--yuhuili--EAGLE3-LLaMA3.1-Instruct-8B", "eagle3-llama-3.1-8b"),
         "70": ("models--lmsys--SGLang-EAGLE3-Llama-3.3-70B-Instruct-SpecForge", "eagle3-llama-3.3-70b")}
DATASETS = {"humaneval": "humaneval/humaneval_data_10000.jsonl", "alpaca": "alpaca/alpaca_data_10000.jsonl",
            "c4": "c4/c4_data_10000.jsonl", "gsm": "gsm8k/gsm8k_data_10000.jsonl",
            "ultrafeedback": "ultrafeedback/ultrafeedback_data_10000.jsonl"}


def parse_arguments():
    p = argparse.ArgumentParser(description="Benchmark the MI355X speculative-decoding engine (flags of the reference's bench.py)")
    p.add_argument("--size", type=str, choices=["0.6", "1.7", "4", "8", "14", "32", "1", "3", "70"], default="70")
    p.add_argument("--llama", action="store_true", default=True)
    p.add_argument("--qwen", action="store_true")
    p.add_argument("--draft", type=str, default=None, help="draft size (0.6 / 1) or a model directory")
    p.add_argument("--eager", action="store_true", help="no hipGraphs")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--spec", action="store_true")
    p.add_argument("--eagle", action="store_true")
    p.add_argument("--k", type=int, default=6)
    p.add_argument("--async", action="store_true")
    p.add_argument("--f", type=int, default=3)
    p.add_argument("--fl", type=int, nargs="+", default=None)
    p.add_argument("--flh", type=int, nargs="+", default=None)
    p.add_argument("--flm", type=int, nargs="+", default=None)
    p.add_argument("--backup", type=str, choices=["jit", "fast"], default="jit")
    p.add_argument("--block_sz", type=int, default=256)
    p.add_argument("--b", type=int, default=1)
    p.add_argument("--max_model_len", type=int, default=8192)
    p.add_argument("--input_len", type=int, default=128)
    p.add_argument("--output_len", type=int, default=512)
    p.add_argument("--numseqs", type=int, default=128)
    p.add_argument("--temp", type=float, default=0.0)
    p.add_argument("--dtemp", type=float, default=None)
    p.add_argument("--x", type=float, default=None, help="sampler_x")
    for name in ("example", "humaneval", "alpaca", "c4", "ultrafeedback", "random", "all", "chat_template", "verbose", "debug",
                 "wandb"):
        p.add_argument(f"--{name}", action="store_true")
    p.add_argument("--prompt_offset", type=int, default=0)
    p.add_argument("--max-steps", type=int, default=None)
    p.add_argument("--group", type=str, default=None)
    p.add_argument("--name", type=str, default=None)
    p.add_argument("--sweep", type=str, default=None, help='JSON list of {"temp": .., "b": ..} overrides run on one engine')
    args = p.parse_args()
    assert not (args.qwen and "--llama" in sys.argv), "--llama and --qwen are mutually exclusive"
    if args.qwen:
        args.llama = False
    if args.eagle:          # reference bench.py:83-87
        args.spec = True
        assert args.llama, "Eagle currently only supports llama models"
        assert args.temp == 0.0 and args.dtemp is None, "Eagle currently only supports greedy decoding (temp=0)"
        assert getattr(args, "async"), "Eagle currently only supports async speculative decoding"
    if args.wandb:
        print("note: --wandb ignored (no network); metrics are printed", flush=True)
    return args


def snapshot(base: str) -> str | None:
    """A HF cache entry -> the directory holding config.json (bench_helpers.py:14-43 semantics)."""
    if not os.path.isdir(base):
        return None
    if os.path.exists(os.path.join(base, "config.json")):
        return base
    for sub in (os.path.join(base, "snapshots"), base):
        if os.path.isdir(sub):
            for item in sorted(os.listdir(sub)):
                d = os.path.join(sub, item)
                if os.path.isdir(d) and os.path.exists(os.path.join(d, "config.json")):
                    return d
    return None


def resolve_models(args):
    """-> (display name, model path or preset name, draft path or preset name)."""
    table, org = (LLAMA, "meta-llama") if args.llama else (QWEN, "Qwen")
    if args.size not in table:
        sys.exit(f"size {args.size} not available for {'Llama' if args.llama else 'Qwen'} models: {sorted(table)}")
    cache = os.environ.get("SSD_HF_CACHE", "")
    name, preset = table[args.size]
    model = snapshot(os.path.join(cache, f"models--{org}--{name}")) if cache else None
    if args.eagle and args.draft is None:       # bench_helpers.py:50-63, bench_paths.py:33-40
        if args.size not in EAGLE:
            sys.exit(f"EAGLE draft not available for Llama size {args.size}")
        ecache, epreset = EAGLE[args.size]
        return name, model or preset, (snapshot(os.path.join(cache, ecache)) if cache else None) or epreset
    dsize = args.draft if args.draft is not None else ("1" if args.llama else "0.6")
    if os.path.isdir(dsize):
        draft = dsize
    else:
        if dsize not in table:
            sys.exit(f"draft size {dsize} unknown")
        dname, dpreset = table[dsize]
        draft = (snapshot(os.path.join(cache, f"models--{org}--{dname}")) if cache else None) or dpreset
    return name, model or preset, draft


def load_prompts(args, model_path: str):
    """Token-id prompts: dataset jsonl ({"text": ...} per line) when the files and a tokenizer exist, else random ids
    (bench_helpers.py:143-300)."""
    def random_ids(n):
        return [[randint(0, 10000) for _ in range(args.input_len)] for _ in range(n)]
    if args.random or not os.path.isdir(model_path):
        return random_ids(args.numseqs), None
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(model_path)
    if args.example:
        texts = ["introduce yourself", "explain the concept of recursion", "describe the color blue", "what are you doing?",
                 "how do you feel?", "what's the weather like today?", "tell me a joke", "what is the meaning of life?"][:min(args.numseqs, 8)]
        return texts, tok
    names = ["humaneval", "alpaca", "gsm", "ultrafeedback"] if args.all else \
        [next((n for n in ("humaneval", "alpaca", "c4", "ultrafeedback") if getattr(args, n)), "gsm")]
    root = os.environ.get("SSD_DATASET_DIR", "")
    prompts = []
    for n in names:
        path = os.path.join(root, DATASETS[n])
        if not os.path.exists(path):
            print(f"Warning: dataset file not found at {path}, falling back to random tokens", flush=True)
            prompts.extend(random_ids(args.numseqs))
            continue
        got = 0
        with open(path) as f:
            for i, line in enumerate(f):
                if i < args.prompt_offset:
                    continue
                if got >= args.numseqs:
                    break
                text = json.loads(line)["text"]
                if args.chat_template and hasattr(tok, "apply_chat_template"):
                    ids = tok.apply_chat_template([{"role": "system", "content": "You are a helpful assistant."},
                                                   {"role": "user", "content": text}], add_generation_prompt=True)
                else:
                    ids = tok.encode(text, add_special_tokens=False)
                prompts.append(list(ids)[:max(len(ids), args.input_len)])
                got += 1
    return prompts, tok


def main():
    args = parse_arguments()
    seed(0)
    name, model, draft = resolve_models(args)
    prompts, tok = load_prompts(args, model)
    kw = dict(enforce_eager=args.eager, num_gpus=args.gpus, speculate=args.spec, speculate_k=args.k,
              draft_async=getattr(args, "async"), async_fan_out=args.f, verbose=args.verbose, draft=draft if args.spec else None,
              kvcache_block_size=args.block_sz, max_num_seqs=args.b, max_model_len=args.max_model_len,
              max_num_batched_tokens=max(args.max_model_len, 8192), sampler_x=args.x, jit_speculate=(args.backup == "jit"),
              max_steps=args.max_steps)
    if args.eagle:
        kw["use_eagle"] = True
    if args.flh is not None or args.fl is not None:
        kw["fan_out_list"] = args.flh if args.flh is not None else args.fl
    if args.flm is not None:
        kw["fan_out_list_miss"] = args.flm
    if not os.path.isdir(model):
        print(f"note: no weights for {name} under $SSD_HF_CACHE; using seeded synthetic weights of its public shapes", flush=True)
    llm = LLM(model, **kw)
    sweeps = json.loads(args.sweep) if args.sweep else [{}]
    for si, cfg in enumerate(sweeps):
        assert not ({"backup", "flh", "flm"} & set(cfg)), "cannot sweep draft-side settings on a live engine"
        temp, b = cfg.get("temp", args.temp), cfg.get("b", args.b)
        assert b <= llm.config.max_num_seqs
        llm.config.max_num_seqs = b
        llm.scheduler.max_num_seqs = b
        sps = [SamplingParams(temperature=temp, draft_temperature=args.dtemp, ignore_eos=True, max_new_tokens=args.output_len)
               for _ in prompts]
        print(f"\n{'=' * 60}\nSWEEP [{si + 1}/{len(sweeps)}] temp={temp} b={b}\n{'=' * 60}", flush=True)
        t0 = time.time()
        outputs, metrics = llm.generate(prompts, sps)
        dt = time.time() - t0
        total = sum(sp.max_new_tokens for sp in sps)
        mode = ("Eager" if args.eager else "hipGraphs") + (f" + Speculative(k={args.k})" if args.spec else "") + \
               (" + Async" if getattr(args, "async") else "") + (" + JIT" if args.backup == "jit" else "") + \
               (f" + X({args.x})" if args.x else "")
        print(f"Model: {name}, Mode: {mode}, Total: {total}tok, Time: {dt:.2f}s, Total Throughput: {total / dt:.2f}tok/s", flush=True)
        if metrics.get("decode_total_time"):
            print(f"Decode Throughput: {metrics['decode_total_tokens'] / metrics['decode_total_time']:.2f}tok/s", flush=True)
        if metrics.get("target_step_times"):
            st = metrics["target_step_times"]
            print(f"Avg target step time: {1e3 * sum(st) / len(st):.2f}ms", flush=True)
        lens = metrics.get("accepted_suffix_lens_with_recovery")
        if lens:
            print(f"Avg tokens per step (incl. recovery): {sum(lens) / len(lens):.2f}", flush=True)
        if metrics.get("cache_hits"):
            print(f"Avg cache hits: {sum(metrics['cache_hits']) / len(metrics['cache_hits']):.2f}", flush=True)
        if tok is not None and si == 0:
            print("\n" + "=" * 80 + "\nGENERATIONS:\n" + "=" * 80)
            for i, (pr, out) in enumerate(zip(prompts[:10], outputs)):
                shown = pr if isinstance(pr, str) else tok.decode(pr, skip_special_tokens=True)
                print(f"\nPrompt {i + 1}: {shown!r}\nGeneration: {out['text']!r}\n" + "-" * 40)
    llm.exit()
    print("Engine exited!")


if __name__ == "__main__":
    main()
