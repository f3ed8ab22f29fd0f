"""The draft server (engine/draft_runner.py DraftServer, SURVEY.md section 8 rows a18-a21 + the EAGLE-3 flow of row f4) against
traces of the reference's OWN DraftRunner methods -- hit_cache_and_respond, jit_speculate, _build_tree_batch, _decode_tree,
_populate_tree_cache run on CPU by tests/golden/make_golden.py gen_draft_rounds -- for three speculation rounds of a batch of
two sequences: an empty cache (JIT chain), two hits (one with extend rows under EAGLE), a hit + a miss (whole batch JIT).
Requests go over the real wire protocol (loopback transport); replies, forks and the speculation cache after every round
must be identical (integers) / bit-identical (EAGLE prenorm vectors) to what the reference produced."""
import pytest
import torch

from oracle.runner import OracleRunner
from ssd_amd.config import Config
from ssd_amd.engine import async_proto as P
from ssd_amd.engine.draft_runner import DraftServer
from ssd_amd.model_config import ModelConfig


def setup(g, eagle):
    K, F = g["K_F"].tolist()
    ci, cf = g["d_cfg_i"].tolist(), g["d_cfg_f"].tolist()
    dw = {k[2:]: v for k, v in g.items() if k.startswith("d.")}
    extra = dict(draft_vocab_size=int(dw["lm_head.weight"].shape[0]), d_model_target=int(g["d_model_target"][0]),
                 eagle_taps=int(g["taps"].numel())) if eagle else {}
    dcfg = ModelConfig("eagle3" if eagle else "llama", ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], cf[0], cf[1], ci[7], False, **extra)
    ht = int(g["d_model_target"][0]) if eagle else 256
    tcfg = ModelConfig("llama", ht, 4, 4, 2, 64, 512, 512, 1e-5, 5e5, 512, False)        # only its shape fields are read here
    cfg = Config("t", hf_config=tcfg, draft="d", draft_hf_config=dcfg, speculate=True, speculate_k=K, draft_async=True,
                 async_fan_out=F, jit_speculate=True, num_gpus=2, max_num_seqs=2, kvcache_block_size=16, max_model_len=192,
                 max_num_batched_tokens=192,
                 use_eagle=eagle, eagle_layers=g["taps"].tolist() if eagle else None)
    runner = OracleRunner(cfg, cfg.draft_hf_config, is_draft=True, weights=dw, num_kvcache_blocks=40)
    client, server_end = P.LoopbackTransport.pair()
    return cfg, DraftServer(cfg, runner, server_end), client, K, F


@pytest.mark.parametrize("name,eagle", [("draft_rounds_llama", False), ("draft_rounds_eagle3", True)])
def test_draft_server_reproduces_the_reference_runner_rounds(golden, name, eagle):
    g = golden(name)
    cfg, server, tx, K, F = setup(g, eagle)
    MQ = F * (K + 1)
    tables = g["draft_block_tables"].tolist()
    prompts = [g["prompt0"].tolist(), g["prompt1"].tolist()]
    # prefill (EAGLE: shifted by one token, with the target activations of the previous position)
    toks = [p[1:] for p in prompts] if eagle else prompts
    payload = P.pack_prefill(toks, tables, cfg.max_blocks)
    tx.send_ints([P.CMD_PREFILL, 2, len(payload), P.FLAG_EAGLE if eagle else 0])
    tx.send_ints(payload)
    if eagle:
        tx.send_tensor(g["prefill_acts"])
    assert server.handle_one()
    for rnd in range(3):
        keys = [tuple(r) for r in g[f"r{rnd}_keys"].tolist()]
        nts = g[f"r{rnd}_num_tokens"].tolist()
        if eagle:
            counts, ids = g[f"r{rnd}_ext_counts"].tolist(), g[f"r{rnd}_ext_ids"].tolist()
            payload = P.pack_speculate(keys, nts, tables, [0.0, 0.0], cfg.max_blocks, counts, ids)
            acts = torch.cat([g[f"r{rnd}_ext_acts"], g[f"r{rnd}_rec_acts"].unsqueeze(1)], dim=1)      # [B, K extend rows | recovery]
        else:
            payload = P.pack_speculate(keys, nts, tables, [0.0, 0.0], cfg.max_blocks)
        tx.send_ints([P.CMD_SPECULATE, 2, len(payload), P.FLAG_EAGLE if eagle else 0])
        tx.send_ints(payload)
        if eagle:
            tx.send_tensor(acts)
        assert server.handle_one()
        resp = tx.recv_tensor((2 + 2 * K,), torch.int64).tolist()
        assert resp[:2] == g[f"r{rnd}_hits"].tolist(), f"round {rnd}: hits"
        assert resp[2:] == g[f"r{rnd}_tokens"].reshape(-1).tolist(), f"round {rnd}: replied tokens"
        # the next round's speculation cache, built after the reply
        assert torch.equal(server.pending_forks, g[f"r{rnd}_forks"]), f"round {rnd}: forks"
        assert torch.equal(server.cache_tokens, g[f"r{rnd}_cache_tokens"]), f"round {rnd}: branch continuations"
        if eagle:
            got, want = server.cache_acts, g[f"r{rnd}_cache_acts"]
            assert torch.equal(got.contiguous().view(torch.int16), want.contiguous().view(torch.int16)), f"round {rnd}: branch prenorms"
        ref_keys = g[f"r{rnd}_cache_keys"].tolist()
        assert len(ref_keys) == 2 * MQ
        forks, (seq_ids, jl) = server.pending_forks.tolist(), server.pending_meta
        mine = [[seq_ids[b], jl[b][i], forks[b][i]] for b in range(2) for i in range(MQ)]
        assert mine == ref_keys, f"round {rnd}: cache keys"
    assert server.stats == {"requests": 6, "hits": 3, "rounds": 3}
