"""Round 6: which side of tests/test_hip_chain.py::test_chain_is_replayable_in_a_graph was wrong on the driver's box (GPUTEST_r05)?

The test filled the KV cache with `kv_cache.normal_()` on the DEFAULT stream and ran the eager forward on a fresh non-blocking side
stream without ordering the two.  Three legs, each on a fresh decoder, each comparing the eager logits, the hipGraph replay's logits and
the ORACLE's logits (oracle/model.py over the same weights and the same -- fully written -- KV cache):

  A  the test as shipped (no ordering), repeated: does eager != replay ever happen spontaneously here?
  B  the same, with the default stream held busy in front of normal_ (torch.cuda._sleep): the fill is forced to land AFTER the eager
     forward started -- if the test race is the cause, eager is now the wrong side deterministically and the replay == the oracle.
  C  B + s.wait_stream(default): ordered; eager == replay == oracle-close however long the fill is delayed.
"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SSD_CHAIN_SEG"] = "1"

from oracle.model import OracleModel, Ctx          # noqa: E402  (diagnostic script: the checker, never the thing measured)
from ssd_amd import weights as W                   # noqa: E402
from ssd_amd.hip import ops as H                   # noqa: E402
from ssd_amd.model import AttnMeta, HipDecoder     # noqa: E402
from ssd_amd.model_config import PRESETS           # noqa: E402


def leg(full, cfg, delay_cycles: int, ordered: bool, seed: int):
    bs, nblocks = 256, 3
    dec = HipDecoder(cfg, max_tokens=64, max_seqs=1, max_blocks=4, block_size=bs, max_model_len=1024, device=torch.device("cuda", 0))
    dec.load_weights(iter(full.items()))
    dec.alloc_kv(nblocks)
    assert dec.chain_seg
    bt = torch.tensor([[0, 1, 2, -1]], dtype=torch.int32).cuda()
    ids = torch.tensor([77], dtype=torch.int64).cuda()
    pos = torch.tensor([5], dtype=torch.int64).cuda()
    meta = AttnMeta(H.MODE_CAUSAL, 1, 1, torch.tensor([5], dtype=torch.int32).cuda(), torch.tensor([6], dtype=torch.int32).cuda(), bt, q_per_seq=1)
    torch.cuda.synchronize()
    torch.manual_seed(seed)
    if delay_cycles:
        torch.cuda._sleep(delay_cycles)
    dec.kv_cache.normal_(0, 0.5)
    s = torch.cuda.Stream()
    if ordered:
        s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dec.forward(ids, pos, 1, meta)
        dec.compute_logits(1)
        s.synchronize()
        eager = dec.logits[:1].clone()
        torch.cuda.synchronize()
        # the cache as the fill left it + what the forward stored at slot 5 (the replay rewrites slot 5 with the same bits)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dec.forward(ids, pos, 1, meta)
            dec.compute_logits(1)
        outs = []
        for _ in range(3):
            dec.logits.zero_()
            g.replay()
            s.synchronize()
            outs.append(dec.logits[:1].clone())
    torch.cuda.synchronize()
    err = int(dec.chain_err.item())
    # oracle over the FINAL cache (positions 0..4 as filled; slot 5 is recomputed by the oracle's own store)
    orc = OracleModel(cfg, full, nblocks, bs)
    orc.kv_cache.copy_(dec.kv_cache.permute(1, 0, 2, 4, 3, 5).cpu())
    ctx = Ctx("verify", slot_mapping=torch.tensor([5], dtype=torch.int32), context_lens=torch.tensor([6], dtype=torch.int32),
              block_tables=bt.cpu(), cu_q=torch.tensor([0, 1], dtype=torch.int32))
    ref = orc.compute_logits(orc.forward(ids.cpu(), pos.cpu(), ctx)).float()
    e, r = eager.float().cpu(), outs[0].float().cpu()
    same_replays = all(torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) for o in outs)
    return dict(eager_eq_replay=bool(torch.equal(eager.view(torch.int16), outs[0].view(torch.int16))), replays_agree=same_replays, err=err,
                eager_vs_oracle=(e - ref).abs().max().item(), replay_vs_oracle=(r - ref).abs().max().item(),
                eager0=e[0, 0].item(), replay0=r[0, 0].item(), oracle0=ref[0, 0].item())


def main():
    cfg = dataclasses.replace(PRESETS["llama-3.2-1b"], num_layers=3)
    full = W.synthetic_state_dict(cfg, seed=5, std=0.02)
    print("leg A: as shipped (unordered), 20 fresh decoders")
    bad = 0
    for i in range(20):
        r = leg(full, cfg, 0, False, i)
        bad += not r["eager_eq_replay"]
        if not r["eager_eq_replay"] or i < 2:
            print("  A", i, r)
    print(f"leg A: eager != replay in {bad} / 20")
    print("leg B: unordered, default stream held busy in front of the fill")
    for cyc in (200_000, 2_000_000, 20_000_000):
        print("  B", cyc, leg(full, cfg, cyc, False, 100))
    print("leg C: the same delays, s.wait_stream(default) before the forward")
    for cyc in (200_000, 2_000_000, 20_000_000):
        print("  C", cyc, leg(full, cfg, cyc, True, 100))


if __name__ == "__main__":
    main()
