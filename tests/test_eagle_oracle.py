"""EAGLE-3: the CPU oracle against tests/golden/tiny_eagle3.npz, which the reference's own
LlamaForCausalLM(use_eagle=True) and Eagle3DraftForCausalLM produced (tests/golden/make_golden.py gen_tiny_eagle)."""
import torch

from oracle import ops as O
from oracle.eagle import OracleEagleDraft
from oracle.model import OracleModel, Ctx
from tests.test_oracle_golden import assert_ulp, slots
from tests.util import eagle_models_from_golden


def exact(a, b):
    assert_ulp(a, b, max_ulp=0, max_frac=0.0)


def test_target_activation_taps(golden):
    g = golden("tiny_eagle3")
    tcfg, tw, _, _, taps, K, _ = eagle_models_from_golden(g)
    m = OracleModel(tcfg, tw, num_blocks=24, block_size=16)
    bt = g["t_block_table"]
    table = bt[0].tolist()
    prompt = g["prompt"]
    P = prompt.numel()
    cu = torch.tensor([0, P], dtype=torch.int32)
    h, acts = m.forward(prompt, torch.arange(P), Ctx("prefill", slot_mapping=slots(table, range(P)), cu_q=cu, cu_k=cu), taps=taps)
    exact(m.compute_logits(h), g["t_prefill_logits"])
    exact(acts, g["t_prefill_acts"])
    assert acts.shape == (P, len(taps) * tcfg.hidden_size)
    vt = torch.cat([g["rec0"], g["jit1_tokens"]])
    n = vt.numel()
    h, acts = m.forward(vt, torch.arange(P, P + n),
                        Ctx("verify", slot_mapping=slots(table, range(P, P + n)), context_lens=torch.tensor([P + n], dtype=torch.int32),
                            block_tables=bt, cu_q=torch.tensor([0, n], dtype=torch.int32)), taps=taps)
    exact(m.compute_logits(h), g["t_verify_logits"])
    exact(acts, g["t_verify_acts"])


def test_eagle_draft_forward_flavours(golden):
    g = golden("tiny_eagle3")
    _, _, dcfg, dw, _, K, F = eagle_models_from_golden(g)
    m = OracleEagleDraft(dcfg, dw, num_blocks=24, block_size=16)
    bt = g["d_block_table"]
    table = bt[0].tolist()
    prompt = g["prompt"].tolist()
    P = len(prompt)
    # prefill with the one-token shift
    cu = torch.tensor([0, P - 1], dtype=torch.int32)
    pre = m.forward(torch.tensor(prompt[1:]), torch.arange(P - 1), g["t_prefill_acts"][:-1],
                    Ctx("prefill", slot_mapping=slots(table, range(P - 1)), cu_q=cu, cu_k=cu))
    exact(pre, g["d_prefill_prenorm"])
    exact(m.compute_logits(pre), g["d_prefill_logits"])

    def chain(rec, act, pos0, toks_key, lg_key, pre_key):
        tok, hid = int(rec), act
        for i in range(K):
            p = pos0 + i
            pre = m.forward(torch.tensor([tok]), torch.tensor([p]), hid,
                            Ctx("decode", slot_mapping=slots(table, [p]), context_lens=torch.tensor([p + 1], dtype=torch.int32), block_tables=bt))
            lg = m.compute_logits(pre)
            exact(pre, g[pre_key][i:i + 1])
            exact(lg, g[lg_key][i:i + 1])
            tok, hid = int(O.argmax_rows(lg)[0]), pre
            assert tok == int(g[toks_key][i])

    chain(g["rec0"][0], g["t_prefill_acts"][-1:], P - 1, "jit1_tokens", "jit1_logits", "jit1_prenorm")
    N = P + K + 2
    chain(g["rec1"][0], g["t_verify_acts"][K:K + 1], N - 2, "jit2_tokens", "jit2_logits", "jit2_prenorm")
    # variable-length glue: K extend rows + recovery + K spec rows, all with draft-width conditioning rows
    n_ext = int(g["glue_n_ext"][0])
    ids = g["glue_ids"]
    n = ids.numel()
    exact(m.project(g["t_verify_acts"][:K + 1]), g["glue_hs"][:K + 1])
    base = N - 2 - n_ext
    pre = m.forward(ids, torch.arange(base, base + n), g["glue_hs"],
                    Ctx("verify", slot_mapping=slots(table, range(base, base + n)), context_lens=torch.tensor([N - 1 + K], dtype=torch.int32),
                        block_tables=bt, cu_q=torch.tensor([0, n], dtype=torch.int32)))
    lg = m.compute_logits(pre)
    exact(pre, g["glue_prenorm"])
    exact(lg, g["glue_logits"])
    # fork + tree
    MQ = F * (K + 1)
    returned = torch.cat([g["rec1"], g["jit2_tokens"]]).view(1, -1)
    forks = O.fork_topf(lg[n_ext:].view(1, K + 1, -1), returned, [[F] * (K + 1)])
    assert torch.equal(forks, g["tree_forks"])
    jidx = [i // F for i in range(MQ)]
    toks, hid = forks[0], pre[n_ext:][torch.tensor(jidx)]
    Pb = N - 2
    for step in range(K):
        rope_pos = torch.tensor([Pb + j + 1 + step for j in jidx])
        cache_pos = [Pb + K + 1 + step * MQ + i for i in range(MQ)]
        pre_t = m.forward(toks, rope_pos, hid, Ctx("tree", slot_mapping=slots(table, cache_pos),
                                                  context_lens=torch.tensor([cache_pos[-1] + 1], dtype=torch.int32), block_tables=bt,
                                                  tree_step=step, tree_K=K, tree_jidx=[jidx]))
        lg_t = m.compute_logits(pre_t)
        exact(pre_t, g["tree_prenorm"][step])
        exact(lg_t, g["tree_logits"][step])
        toks, hid = O.argmax_rows(lg_t), pre_t
