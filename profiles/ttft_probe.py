"""Attribute one TTFT sample of the c4 workload (70B target + co-located 1B draft, async k = 7 f = 3, 128-token prompt):
host wall-clock segments of LLMEngine.generate(max_new_tokens = 1) and device time of the target prefill / the draft
prefill (HIP events on their own streams).  python profiles/ttft_probe.py [workload] > gpurun_out/r03/ttft_probe.txt"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ssd_amd.engine.llm_engine import LLMEngine  # noqa: E402
from ssd_amd.sampling_params import SamplingParams  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    import dataclasses
    tname, tcfg, dname, dcfg = bench.workload_models(wl)
    is_async = wl in bench.ASYNC_WORKLOADS
    K = 7 if is_async else 6
    dcfg = dataclasses.replace(dcfg, tie_word_embeddings=False)
    recipe = {"kind": "pair", "shared": min(dcfg.hidden_size, tcfg.hidden_size), "snr": 8.0, "layer_gain": 0.005}
    kw = dict(hf_config=tcfg, draft=dname, draft_hf_config=dcfg, speculate=True, speculate_k=K, num_gpus=1, max_num_seqs=1,
              max_model_len=8192, max_num_batched_tokens=8192, kvcache_block_size=256, num_kvcache_blocks=40,
              num_draft_kvcache_blocks=40, weights_recipe=recipe)
    if is_async:
        kw.update(draft_async=True, async_fan_out=3, jit_speculate=True, inprocess_draft=True, num_draft_gpus=1)
    eng = LLMEngine(tname, **kw)
    random.seed(0)
    prompt = [random.randint(0, 10000) for _ in range(128)]
    sp = SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=1)
    marks = []

    def mark(name):
        marks.append((name, time.perf_counter()))

    mr = eng.model_runner
    ev = {}
    real_run = mr.run

    def run(seqs, is_prefill, *a, **k):
        mark("target.run enter")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_prepare = mr._prepare_prefill

        def prep(s):
            r = real_prepare(s)
            mark("  target inputs staged")
            return r
        mr._prepare_prefill = prep
        real_read = mr._read_tokens

        def read(n):
            mark("  target graph launched (host returns)")
            e1.record()
            r = real_read(n)
            mark("  target tokens read (stream sync)")
            return r
        mr._read_tokens = read
        try:
            out = real_run(seqs, is_prefill, *a, **k)
        finally:
            mr._prepare_prefill, mr._read_tokens = real_prepare, real_read
        ev["target"] = (e0, e1)
        mark("target.run exit")
        return out
    mr.run = run
    srv = getattr(eng, "draft_server", None)
    if srv is not None:
        real_handle = srv._handle_one

        def handle():
            mark("draft server: command enter")
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = real_handle()
            e1.record(s)
            ev["draft"] = (e0, e1)
            mark("draft server: command exit (host)")
            return r
        srv._handle_one = handle
    real_step = eng.step

    def step(st):
        mark("engine.step enter")
        r = real_step(st)
        mark("engine.step exit")
        return r
    eng.step = step
    for it in range(5):
        torch.cuda.synchronize()
        marks.clear()
        first = []
        mark("generate() called")
        eng.generate([prompt], sp, use_tqdm=False, stream_callback=lambda sid, toks: first.append(time.perf_counter()) if not first else None)
        mark("generate() returned")
        torch.cuda.synchronize()
        t0 = marks[0][1]
        print(f"---- sample {it}: TTFT {(first[0] - t0) * 1e3:.2f} ms")
        if it >= 3:
            for name, t in marks:
                print(f"   {(t - t0) * 1e3:8.3f} ms  {name}")
            for k_, (e0, e1) in ev.items():
                print(f"   device time {k_:7s}: {e0.elapsed_time(e1):8.3f} ms (between its first and last enqueue on its stream)")
    # the prefill graph alone, replayed back to back (pure device time of the target prefill)
    g = [v for k_, v in mr.graphs.items() if k_[0] == "prefill"]
    if g:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(3):
            g[0].replay()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"target prefill hipGraph replayed 3x back to back: {e0.elapsed_time(e1) / 3:.3f} ms per replay on the device; "
              f"host time of the 3 replay calls {(t1 - t0) * 1e3:.3f} ms")
    eng.exit()


main()
