"""ssd_amd's Scheduler / BlockManager / Sequence replayed against decision traces recorded from the REFERENCE's own
scheduler (tests/golden/make_golden.py: gen_scheduler): prefix-cache sharing, speculation lookahead + rollback,
preemption under KV pressure, EOS / max_new_tokens.  Every visible decision must be identical: which sequences are
scheduled, prefill vs decode, block tables (hence KV slot numbers), cached-token counters, free-list size."""
import itertools
import json
import os
from types import SimpleNamespace

import pytest

from ssd_amd.engine.scheduler import Scheduler
from ssd_amd.engine.sequence import Sequence
from ssd_amd.sampling_params import SamplingParams
from tests.conftest import GOLDEN


@pytest.mark.parametrize("name", ["ar", "spec", "tight", "eagle"])
def test_scheduler_matches_reference_trace(name):
    tr = json.load(open(os.path.join(GOLDEN, "scheduler_golden.json")))[name]
    bs = 16
    Sequence.block_size = bs
    Sequence.counter = itertools.count()
    eagle = name == "eagle"         # the async + EAGLE-3 trace; prefix caching stays ON here, as in the reference's scheduler
    cfg = SimpleNamespace(max_num_seqs=tr["max_seqs"], max_num_batched_tokens=256, max_model_len=256, eos=5,
                          speculate=tr["speculate"], draft_async=eagle, speculate_k=tr["K"], kvcache_block_size=bs,
                          num_kvcache_blocks=tr["nblocks"], fan_out_list=[2] * (tr["K"] + 1) if eagle else None)
    sch = Scheduler(cfg, draft_num_blocks=tr["nblocks"] if tr["speculate"] else None)
    for toks, mnt, ign in tr["reqs"]:
        sch.add(Sequence(toks, SamplingParams(temperature=0.0, max_new_tokens=mnt, ignore_eos=ign)))
    for i, ev in enumerate(tr["events"]):
        assert not sch.is_finished()
        seqs, is_prefill = sch.schedule()
        ctx = f"{name} event {i}"
        assert is_prefill == ev["prefill"], ctx
        assert [s.seq_id for s in seqs] == ev["ids"], ctx
        assert [list(s.block_table) for s in seqs] == ev["bt"], ctx
        assert [list(s.draft_block_table) for s in seqs] == ev["dbt"], ctx
        assert [s.num_cached_tokens for s in seqs] == ev["cached"], ctx
        assert [s.num_draft_cached_tokens for s in seqs] == ev["dcached"], ctx
        assert len(sch.block_manager.free_block_ids) == ev["free"], ctx
        if not tr["speculate"]:
            sch.postprocess(seqs, ev["tokens"], is_prefill)
        elif is_prefill:
            for s, r in zip(seqs, ev["rec"]):
                s.recovery_token_id = r
                s.num_cached_tokens = s.num_prompt_tokens
                s.num_draft_cached_tokens = s.num_prompt_tokens
        elif eagle:
            import torch
            K = tr["K"]
            acts = torch.tensor([[[i, b, j] for j in range(K + 1)] for b in range(len(seqs))], dtype=torch.float32)
            sch.postprocess_speculate(seqs, ev["suffixes"], ev["rec"], eagle_acts=acts)
            for s, want in zip(seqs, ev["eagle"]):      # scheduler.py:303-320
                assert s.last_target_hidden_state.tolist() == want["last"], ctx
                assert s.extend_count == want["count"], ctx
                assert (s.extend_token_ids or []) == want["ids"], ctx
                assert ([] if s.extend_eagle_acts is None else s.extend_eagle_acts.tolist()) == want["acts"], ctx
        else:
            sch.postprocess_speculate(seqs, ev["suffixes"], ev["rec"])
        assert [s.num_tokens for s in seqs] == ev["after_len"], ctx
        assert [s.is_finished for s in seqs] == ev["finished"], ctx
        assert [s.seq_id for s in sch.waiting] == ev["waiting"], ctx
        assert [s.seq_id for s in sch.running] == ev["running"], ctx
    assert sch.is_finished()
    assert len(sch.block_manager.free_block_ids) == tr["nblocks"]
