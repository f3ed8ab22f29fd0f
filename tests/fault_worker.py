"""One rank of a fault-injection launch (tests/test_fault_injection_cpu.py): the real engine on oracle runners over gloo, under the
same RunGuard as bench.py (ssd_amd/utils/watchdog.py) and with bench.py's stage names -- plus a fault switched on by the
environment:  SSD_FAULT="rank=R,after=S,kind=exit|hang|raise"  makes rank R misbehave after S timed steps (os._exit without a
word / sleep forever / raise).  Launched with `python -m torch.distributed.run`, like the driver launches bench.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    layout = sys.argv[1]                    # "tp2" | "tp1+draft" | "tp2+draft"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fault = dict(kv.split("=") for kv in os.environ.get("SSD_FAULT", "").split(",") if kv)
    from ssd_amd.utils.watchdog import RunGuard
    guard = RunGuard(rank, world, {"metric": "fault-injection worker", "value": None, "unit": "tokens/s", "n_gpus": world,
                                   "config": {"workload": layout}}).install()

    def body():
        import torch
        import torch.distributed as dist
        torch.set_num_threads(1)
        from oracle.runner import oracle_runner_factory
        from ssd_amd.engine.llm_engine import LLMEngine
        from ssd_amd.model_config import ModelConfig
        from ssd_amd.sampling_params import SamplingParams
        t = ModelConfig("llama", 64, 2, 4, 2, 32, 128, 256, 1e-5, 5e5, 512, False)
        kw = dict(hf_config=t, max_model_len=256, max_num_batched_tokens=256, kvcache_block_size=16, num_kvcache_blocks=40,
                  weights_std=0.1, num_gpus=world, draft="d", draft_hf_config=t, draft_weights_seed=0, speculate=True, speculate_k=3,
                  max_num_seqs=1)
        dedicated = layout.endswith("+draft")
        if dedicated:
            kw.update(draft_async=True, async_fan_out=2, jit_speculate=True, num_draft_gpus=1)
        guard.stage("engine_init")
        eng = LLMEngine("t", runner_factory=oracle_runner_factory(), **kw)

        def misbehave():
            kind = fault.get("kind", "exit")
            if kind == "exit":
                os._exit(9)                 # dies without a word (a segfault, an OOM kill)
            if kind == "raise":
                raise RuntimeError("injected failure")
            while True:                     # wedged: alive, silent, never reaches the next collective
                time.sleep(1)

        if eng.is_draft_process:
            guard.stage("draft_serve", guard.total_deadline)
            if fault and int(fault["rank"]) == rank:
                served = [0]
                orig = eng.draft_server.handle_one

                def handle_one():
                    served[0] += 1
                    if served[0] > int(fault["after"]) + 2:     # (+ hello and prefill)
                        misbehave()
                    return orig()
                eng.draft_server.handle_one = handle_one
            eng.serve()
            guard.stage("final_barrier", 60.0)
            dist.barrier()
            return
        guard.stage("warmup_steps")
        eng.add_request([(3 * j + 1) % 256 for j in range(12)], SamplingParams(temperature=0, ignore_eos=True, max_new_tokens=120))
        step = eng.create_inference_step(eng.config)
        eng.step(step)
        guard.stage("timed_steps")
        for i in range(12):
            if fault and int(fault["rank"]) == rank and i == int(fault["after"]):
                misbehave()
            eng.step(step)
        if rank == 0:
            print(json.dumps({"metric": "fault-injection worker", "value": 1.0, "unit": "tokens/s", "n_gpus": world}), flush=True)
        guard.done()
        eng.exit()
        dist.barrier()
    guard.run(body)
    os._exit(0)


if __name__ == "__main__":
    main()
