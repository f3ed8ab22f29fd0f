"""The resident M-row layer segment (csrc/tree_segment.hip, ssd_tree_segment: o_proj -> add + norm -> gate_up + SiLU -> down_proj ->
add + norm -> next layer's QKV + RoPE + KV store in ONE launch, for the async draft's K+1-row glue decode and its MQ_LEN-row tree
steps) at the REAL Llama-3.2-1B geometry (and Qwen3-0.6B's: q / k norm, h 1024) with PLAIN N(0, 0.02) weights (no damping):

  prefill -> glue (K + 1 = 8 rows, causal) -> device fork at V = 128256 -> K = 7 tree steps of 24 branches (structural tree mask)

compared, step by step, with (a) the separate launches it replaces, (b) the oracle model (reference
ssd/models/llama3.py:128-199, ssd/layers/attention.py:113-125, ssd/engine/helpers/mask_helpers.py:12-21,
ssd/utils/async_helpers/async_spec_helpers.py:26-78 restated) and (c) the float64 forward of the same weights under the same
visibility (tests/util.py truth_forward_masked) with the criterion of test_verify_logits_full_1b...: every row of the HIP logits
must be as close to exact arithmetic as the oracle pipeline's row is.  Same rounding points as the separate launches, a different
fp32 summation order inside o_proj / down_proj: tolerance-tested, identical argmax outside near-ties, forks bit-equal outside
near-ties, the new K / V rows compared.  The 16-layer case is the full model."""
import dataclasses
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from ssd_amd.model_config import PRESETS
from tests.util import truth_forward_masked


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def build(cfg, full, seg: bool, monkeypatch, nblocks=3, bs=256):
    from ssd_amd.model import HipDecoder
    monkeypatch.setenv("SSD_TREE_SEG", "1" if seg else "0")
    dec = HipDecoder(cfg, max_tokens=64, max_seqs=1, max_blocks=4, block_size=bs, max_model_len=1024, device=torch.device("cuda", 0))
    dec.load_weights(iter(full.items()))
    dec.alloc_kv(nblocks)
    assert dec.tree_seg == seg
    return dec


def i64(x):
    return torch.tensor(list(x), dtype=torch.int64)


def i32(x):
    return torch.tensor(list(x), dtype=torch.int32)


def test_hardware_bf16_conversion_equals_the_integer_rounding(H):
    """csrc/common.h pack_bf2_hw (v_cvt_pk_bf16_f32) == f2bf over ALL 2^32 fp32 patterns (NaNs stay NaNs)."""
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    H.selftest_bf16_cvt(counts)
    torch.cuda.synchronize()
    assert counts.tolist() == [0, 0], counts.tolist()


@pytest.mark.parametrize("preset,layers", [("llama-3.2-1b", 2), ("llama-3.2-1b", 16), ("qwen3-0.6b", 3)])
def test_glue_fork_tree_vs_separate_launches_oracle_and_truth(H, monkeypatch, preset, layers):
    from oracle import ops as O
    from oracle.model import OracleModel, Ctx
    from ssd_amd import weights as W
    from ssd_amd.model import AttnMeta
    # (qwen3-0.6b: h 1024, head dim 128, a per-head q / k RMSNorm between the projection and the rotation -- the segment then hands the
    #  next layer's raw QKV rows to ssd_rope_store_kv: reference ssd/models/qwen3.py:90-108; the draft of BASELINE configs[4])
    cfg = dataclasses.replace(PRESETS[preset], num_layers=layers)
    full = W.synthetic_state_dict(cfg, seed=13, std=0.02)
    bs, nblocks = 256, 3
    decs = {c: build(cfg, full, c, monkeypatch) for c in (False, True)}
    orc = OracleModel(cfg, full, nblocks, bs)
    random.seed(4)
    P, K, F = 29, 7, 3
    MQ = F * (K + 1)
    prompt = [random.randint(0, 100000) for _ in range(P)]
    glue_toks = [random.randint(0, 100000) for _ in range(K + 1)]
    table = [2, 0, 1]
    bt = torch.tensor([table + [-1]], dtype=torch.int32)

    def slots(ps):
        return torch.tensor([table[p // bs] * bs + p % bs for p in ps], dtype=torch.int32)

    def compare(got, ref, what, truth=None):
        """got: {seg: logits}; the bars of tests/test_hip_chain.py + (with truth) the exact-arithmetic criterion."""
        scale = max(1.0, ref.std().item())
        d_sep, d_seg = (got[False] - ref).abs(), (got[True] - ref).abs()
        print(f"layers {layers} {what}: |sep - oracle| max {d_sep.max():.4f} mean {d_sep.mean():.5f}; |seg - oracle| max {d_seg.max():.4f} "
              f"mean {d_seg.mean():.5f}; |seg - sep| max {(got[True] - got[False]).abs().max():.4f}; logit std {ref.std():.3f}")
        assert torch.isfinite(got[True]).all()
        assert d_seg.max().item() <= 0.05 * scale * max(1.0, layers / 4) and d_seg.mean().item() <= 0.01 * scale * max(1.0, layers / 4)
        assert d_seg.mean().item() <= 1.5 * d_sep.mean().item() + 1e-4, "the segment is further from the oracle than the separate launches"
        top2 = ref.topk(2, dim=-1).values          # a flipped argmax needs the oracle's own margin inside twice the row's deviation
        thr = torch.clamp(2 * d_seg.max(-1).values, min=0.0625)
        assert bool(((got[True].argmax(-1) == ref.argmax(-1)) | ((top2[:, 0] - top2[:, 1]) < thr)).all()), what
        if truth is not None:
            rms = lambda e: e.pow(2).mean(-1).sqrt()
            for c in (False, True):
                e_hip, e_ref = (got[c].double() - truth).abs(), (ref.double() - truth).abs()
                print(f"    {'segment ' if c else 'separate'} |HIP-truth| max {e_hip.max():.4f} rms {rms(e_hip).mean():.5f} | |oracle-truth| max "
                      f"{e_ref.max():.4f} rms {rms(e_ref).mean():.5f}")
                assert bool((rms(e_hip) <= 1.25 * rms(e_ref) + 1e-3).all()), f"{what}: a HIP row is further from exact arithmetic than the oracle's"
                assert e_hip.max().item() <= 1.5 * e_ref.max().item() + 1e-3

    def kv_check(pos_list, what):
        li = layers - 1
        for which in (0, 1):
            ref_rows = torch.stack([orc.kv_cache[which, li, table[p // bs], p % bs] for p in pos_list]).float()
            rows = {c: torch.stack([dec.kv_cache[li, which, table[p // bs], :, p % bs, :] for p in pos_list]).cpu().float() for c, dec in decs.items()}
            tol = 2.0 ** (math.floor(math.log2(ref_rows.abs().max().item())) - 6)
            dkv, dsep = (rows[True] - ref_rows).abs(), (rows[False] - ref_rows).abs()
            assert dkv.max().item() <= max(tol, 1.5 * dsep.max().item()), (what, which, dkv.max().item(), dsep.max().item(), tol)
            assert dkv.mean().item() <= max(tol / 8, 1.25 * dsep.mean().item() + tol / 64), (what, which, dkv.mean().item(), dsep.mean().item())

    # ---- prefill ----
    cu = torch.tensor([0, P], dtype=torch.int32)
    orc.forward(i64(prompt), i64(range(P)), Ctx("prefill", slot_mapping=slots(range(P)), cu_q=cu, cu_k=cu))
    for dec in decs.values():
        meta = AttnMeta(H.MODE_CAUSAL, 1, P, slots(range(P)).cuda(), i32([P]).cuda(), bt.cuda(), cu_q=cu.cuda())
        dec.forward(i64(prompt).cuda(), i64(range(P)).cuda(), P, meta)
    # ---- the whole tree under exact arithmetic: tokens are fixed below by following the ORACLE's branches ----
    gp = list(range(P, P + K + 1))
    ref_glue = orc.compute_logits(orc.forward(i64(glue_toks), i64(gp), Ctx("verify", slot_mapping=slots(gp), context_lens=i32([P + K + 1]),
                                                                            block_tables=bt, cu_q=i32([0, K + 1])))).float()
    got = {}
    for c, dec in decs.items():
        meta = AttnMeta(H.MODE_CAUSAL, 1, K + 1, slots(gp).cuda(), i32([P + K + 1]).cuda(), bt.cuda(), q_per_seq=K + 1)
        dec.forward(i64(glue_toks).cuda(), i64(gp).cuda(), K + 1, meta)
        n = dec.compute_logits(K + 1)
        got[c] = dec.logits[:n].float().cpu()
        assert int(dec.chain_err.item()) == 0 if c else True
    # ---- fork on the device at V = 128256 (reference async_spec_helpers.py:26-78) ----
    ref_forks = O.fork_topf(ref_glue.view(1, K + 1, -1).to(torch.bfloat16), i64(glue_toks).view(1, -1), [[F] * (K + 1)])
    counts = torch.full((1, K + 1), F, dtype=torch.int32)
    offs = (torch.cumsum(counts, 1) - counts).to(torch.int32)
    srt = ref_glue.clone()
    srt[:-1].scatter_(1, i64(glue_toks)[1:].view(-1, 1), float("-inf"))
    top = srt.topk(F + 1, dim=-1).values
    gaps = (top[:, :-1] - top[:, 1:]).min(-1).values            # smallest gap between consecutive top-(F+1) logits per glue row
    for c, dec in decs.items():
        forks = torch.zeros(1, MQ, dtype=torch.int64, device="cuda")
        H.fork_topf(dec.logits[:K + 1], dec.V, dec.V, i64(glue_toks).view(1, -1).cuda(), counts.cuda(), offs.cuda(), 1, K, MQ, forks)
        same = (forks.cpu().view(K + 1, F) == ref_forks.view(K + 1, F)).all(-1)
        thr = torch.clamp(2 * (got[c] - ref_glue).abs().max(-1).values, min=0.0625)
        assert bool((same | (gaps < thr)).all()), (c, forks.cpu().tolist(), ref_forks.tolist(), gaps.tolist())
    jidx = [i // F for i in range(MQ)]
    # tokens of every tree step, following the oracle
    toks = ref_forks.view(-1)
    tree_toks, tree_pos, ref_steps = [], [], []
    for step in range(K):
        rope_pos = [P + j + 1 + step for j in jidx]
        cache_pos = [P + K + 1 + step * MQ + i for i in range(MQ)]
        ctx = Ctx("tree", slot_mapping=slots(cache_pos), context_lens=i32([cache_pos[-1] + 1]), block_tables=bt, tree_step=step, tree_K=K,
                  tree_jidx=[jidx])
        ref_steps.append(orc.compute_logits(orc.forward(toks, i64(rope_pos), ctx)).float())
        tree_toks.append(toks.tolist())
        tree_pos.append(rope_pos)
        toks = ref_steps[-1].argmax(-1)
    # exact arithmetic over [prompt | glue | step 0 rows | step 1 rows | ...] with the same visibility
    all_toks = prompt + glue_toks + [t for st in tree_toks for t in st]
    all_pos = list(range(P + K + 1)) + [p for st in tree_pos for p in st]
    Ttot = len(all_toks)
    vis = torch.zeros(Ttot, Ttot, dtype=torch.bool)
    vis[:P + K + 1, :P + K + 1] = torch.ones(P + K + 1, P + K + 1, dtype=torch.bool).tril()
    for step in range(K):
        ctx_len = P + K + 1 + (step + 1) * MQ
        vis[P + K + 1 + step * MQ:P + K + 1 + (step + 1) * MQ, :ctx_len] = O.tree_mask(ctx_len, step, K, jidx)
    truth = truth_forward_masked(cfg, full, all_toks, all_pos, vis)
    compare(got, ref_glue, "glue", truth[P:P + K + 1])
    kv_check(gp, "glue")
    # ---- K tree steps of MQ branches ----
    for step in range(K):
        cache_pos = [P + K + 1 + step * MQ + i for i in range(MQ)]
        got = {}
        for c, dec in decs.items():
            meta = AttnMeta(H.MODE_TREE, 1, MQ, slots(cache_pos).cuda(), i32([cache_pos[-1] + 1]).cuda(), bt.cuda(), q_per_seq=MQ,
                            tree_K=K, tree_mq=MQ, tree_step=step, tree_F=F)
            dec.forward(i64(tree_toks[step]).cuda(), i64(tree_pos[step]).cuda(), MQ, meta)
            n = dec.compute_logits(MQ)
            got[c] = dec.logits[:n].float().cpu()
        assert int(decs[True].chain_err.item()) == 0, "a bounded wait inside the segment gave up"
        compare(got, ref_steps[step], f"tree step {step}", truth[P + K + 1 + step * MQ:P + K + 1 + (step + 1) * MQ])
        kv_check(cache_pos, f"tree step {step}")


@torch.inference_mode()
def test_segment_replays_in_a_graph_bit_identically_under_load(H, monkeypatch):
    """hipGraph replay of a 24-row forward: tags come from a device word bumped inside the graph, so replays need no re-initialisation;
    300 replays -- half of them next to a second stream that keeps the memory system busy (uneven load: a stale hand-off read shows up
    as a flipped bit) -- must give the bits of the eager run, every logit of every row."""
    from ssd_amd import weights as W
    from ssd_amd.model import AttnMeta
    cfg = dataclasses.replace(PRESETS["llama-3.2-1b"], num_layers=5)
    full = W.synthetic_state_dict(cfg, seed=5, std=0.02)
    dec = build(cfg, full, True, monkeypatch)
    bt = torch.tensor([[0, 1, 2, -1]], dtype=torch.int32).cuda()
    K, F, MQ, P = 7, 3, 24, 40
    random.seed(1)
    ids = i64([random.randint(0, 100000) for _ in range(MQ)]).cuda()
    jidx = [i // F for i in range(MQ)]
    pos = i64([P + j + 1 for j in jidx]).cuda()
    cache_pos = [P + K + 1 + i for i in range(MQ)]
    meta = AttnMeta(H.MODE_TREE, 1, MQ, i32(cache_pos).cuda(), i32([cache_pos[-1] + 1]).cuda(), bt, q_per_seq=MQ, tree_K=K, tree_mq=MQ,
                    tree_step=0, tree_F=F)
    dec.kv_cache.normal_(0, 0.5)
    s, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    s.wait_stream(torch.cuda.current_stream())      # side streams are non-blocking: order the cache fill in front of the forward
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dec.forward(ids, pos, MQ, meta)
        dec.compute_logits(MQ)
        s.synchronize()
        eager = dec.logits[:MQ].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dec.forward(ids, pos, MQ, meta)
            dec.compute_logits(MQ)
        for it in range(300):
            dec.logits.zero_()
            if it % 2:
                with torch.cuda.stream(s2):
                    big[:128 << 20].copy_(big[128 << 20:], non_blocking=True)
            g.replay()
            s.synchronize()
            assert torch.equal(dec.logits[:MQ].view(torch.int16), eager.view(torch.int16)), f"replay {it} differs from the eager run"
        torch.cuda.synchronize()
    assert int(dec.chain_err.item()) == 0
    assert int(dec.chain_gen.item()) >= 300
