"""Per-kind table of ONE tensor-parallel rank's verify forward at the 70B shard shapes (TP = 4 and 8), on ONE GPU:
a model config with exactly the per-rank shapes (heads, kv heads, intermediate and vocabulary divided by N; a cut of
L_CUT layers so that it builds in seconds -- L_CUT x 430 MB of weights per pass is far more than the 256 MB Infinity
Cache, nothing is cache-resident), collectives forced through a 1-rank group so the one-shot all-reduce (+ fused add +
RMSNorm) kernels really launch.  = the TP = N verify minus the xGMI wait.  Every kind is a hipGraph of its L_CUT
launches, replayed; the whole verify body (all kinds in order + LM head + accept / reject) is timed the same way and
scaled to 80 layers.
    python profiles/tp_shard_probe.py [M=8] > gpurun_out/r03/tp_shard_per_kind.txt"""
import dataclasses
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_amd.config import Config  # noqa: E402
from ssd_amd.engine.llm_engine import hip_runner_factory  # noqa: E402
from ssd_amd.hip import ops as H  # noqa: E402
from ssd_amd.model import AttnMeta  # noqa: E402
from ssd_amd.model_config import ModelConfig, PRESETS  # noqa: E402
from ssd_amd.utils.graphs import capture  # noqa: E402
from ssd_amd.utils.topology import Topology  # noqa: E402

L_CUT = 8


def timed(fn, L, reps=6):
    for li in range(min(2, L)):
        fn(li)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        for li in range(L):
            fn(li)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * L)


@torch.inference_mode()
def probe(tp: int, M: int, grp, dev):
    full = PRESETS["llama-3.1-70b"]
    shard = ModelConfig("llama", full.hidden_size, L_CUT, full.num_heads // tp, full.num_kv_heads // tp, full.head_dim,
                        full.intermediate_size // tp, full.vocab_size // tp, full.rms_norm_eps, full.rope_theta,
                        full.max_position_embeddings, False)
    K = M - 1
    d = PRESETS["llama-3.2-1b"]        # (never built: Config only wants a draft whose vocabulary equals the target's)
    dshard = dataclasses.replace(d, vocab_size=full.vocab_size // tp)
    cfg = Config("shard", hf_config=shard, speculate=True, speculate_k=K, draft="d", draft_hf_config=dshard,
                 max_num_seqs=1, max_model_len=2048, max_num_batched_tokens=2048, kvcache_block_size=256, num_kvcache_blocks=10)
    topo = Topology(0, 1, dev, "target", 0, 1, grp)
    r = hip_runner_factory(cfg, shard, is_draft=False, topo=topo, force_collectives=True, custom_ar=True, num_kvcache_blocks=10)
    m = r.model
    ar = m.custom_ar
    assert ar is not None, "one-shot all-reduce did not come up"
    L = L_CUT
    ctx = 300
    r.d_slots[:64].fill_(-1)
    r.d_pos[:64].fill_(ctx - M)
    r.d_ctx[:1].fill_(ctx)
    r.d_bt[0, :8] = torch.arange(8, dtype=torch.int32, device=dev)
    r._ctx_hint = 1024
    meta = AttnMeta(H.MODE_CAUSAL, 1, M, r.d_slots, r.d_ctx, r.d_bt, q_per_seq=M, ctx_hint=1024)
    splits, waves = m._attn_cfg(M, meta)
    scale = m.hd ** -0.5
    w = m.w
    eps = shard.rms_norm_eps
    rows = []

    def row(name, us, mb, per_fwd):
        rows.append((name, us, mb, per_fwd))
        print(f"  {name:46s} {us:8.2f} us  {mb:8.1f} MB  {mb / us if us else 0:6.2f} TB/s   x{per_fwd:4d} = {us * per_fwd / 1e3:7.3f} ms", flush=True)

    print(f"==== TP = {tp}: per-rank shard of Llama-3.1-70B, M = {M} verify rows, ctx {ctx} ====")
    NL = 80
    row(f"qkv+RoPE+KV store [{m.qkv_n}x{m.h}]", timed(lambda li: m.launch_qkv(li, M, r.d_pos, r.d_slots, pre_normed=True), L), m.qkv_n * m.h * 2 / 1e6, NL)
    row(f"attention (nh {m.nh}, nkv {m.nkv}, {waves} waves, {splits} splits)",
        timed(lambda li: H.attn_paged(m.buf_q, m.kv_cache[li, 0], m.kv_cache[li, 1], meta.block_tables, m.max_blocks, meta.context_lens,
                                      1, M, M, m.nh, m.nkv, m.hd, m.block_size, scale, q_per_seq=M, mode=meta.mode, splits=splits,
                                      ws_o=m.ws_o, ws_ml=m.ws_ml, out_frag=m.buf_af, waves=waves), L), 2 * ctx * m.nkv * m.hd * 2 / 1e6, NL)
    row(f"o_proj [{m.h}x{m.qn}]", timed(lambda li: m.launch_o(li, M), L), m.h * m.qn * 2 / 1e6, NL)
    row("all-reduce + add + RMSNorm (1 rank: launch + flags only)",
        timed(lambda li: ar.all_reduce_add_rmsnorm(m.buf_h, m.buf_res, m.buf_res, w[f"model.layers.{li}.post_attention_layernorm.weight"],
                                                   eps, M, m.h, out_frag=m.buf_xf), L), M * m.h * 2 * 3 / 1e6, 2 * NL)
    row(f"gate_up+SiLU [{2 * m.I}x{m.h}]", timed(lambda li: m.launch_gate_up(li, M, pre_normed=True), L), 2 * m.I * m.h * 2 / 1e6, NL)
    row(f"down_proj [{m.h}x{m.I}]", timed(lambda li: m.launch_down(li, M), L), m.h * m.I * 2 / 1e6, NL)
    row("embedding + all-reduce", timed(lambda li: (H.embedding(r.d_ids, w["model.embed_tokens.weight"], m.buf_h, M, m.h, vocab_start=0,
                                                               vocab_count=m.V), m._allreduce(m.buf_h[:M])), 4), 0.0, 1)
    row(f"final norm + LM head [{m.V}x{m.h}] (+ argmax candidates)", timed(lambda li: m.compute_logits(M), 4), m.V * m.h * 2 / 1e6, 1)
    row("argmax (vocab-parallel: candidates -> all-gather -> merge)", timed(lambda li: m.argmax(M, r.d_next), 4), 0.0, 1)
    row("verify_greedy", timed(lambda li: H.verify_greedy(r.d_next, r.d_ids, 1, K, r.d_accept, r.d_recovery, r.d_packed), 4), 0.0, 1)
    total_kinds = sum(us * n for _, us, _, n in rows) / 1e3
    # the real body: all layers in order + head + accept / reject, one hipGraph
    body = timed(lambda _: r._body_verify(1, True), 1, reps=10)
    per_layer = None
    # split the body into (layers) + (tail) by timing the tail alone
    tail = timed(lambda _: (m.compute_logits(M), m.argmax(M, r.d_next), H.verify_greedy(r.d_next, r.d_ids, 1, K, r.d_accept, r.d_recovery, r.d_packed)), 1, reps=10)
    emb = rows[6][1]
    per_layer = (body - tail - emb) / L
    proj = (per_layer * NL + tail + emb) / 1e3
    gb = sum(mb * n for _, _, mb, n in rows[:6] if mb > 1.0) / 1e3 + m.V * m.h * 2 / 1e9      # layer matrices x 80 + LM head
    print(f"  sum of kinds x 80 layers                       {total_kinds:8.3f} ms")
    print(f"  verify body, {L} layers in one graph            {body:8.1f} us  -> per layer {per_layer:6.2f} us, tail {tail:6.1f} us")
    print(f"  PROJECTED TP = {tp} verify (80 layers, zero xGMI wait)  {proj:7.3f} ms for {gb:.2f} GB  = {gb / proj:.3f} TB/s = {gb / proj / 8.0:.3f} of 8 TB/s")
    for n_coll, us in ((161, 5.0), (161, 15.0)):
        t = proj + n_coll * us / 1e3 + 0.3
        print(f"    + {n_coll} collectives x {us:.0f} us xGMI + 0.3 ms hand-off: {t:6.3f} ms -> {gb / t / 8.0:.3f} of 8 TB/s")
    del r


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29589")
    dist.init_process_group("nccl", rank=0, world_size=1)
    grp = dist.new_group([0])
    dev = torch.device("cuda", 0)
    for tp in (4, 8):
        probe(tp, M, grp, dev)
        torch.cuda.empty_cache()
    dist.destroy_process_group()


main()
