"""Round-2 probe of the 1B draft decode layer on MI355X: every launch kind at M = 1 (and the M = 24 tree shapes), the
split-K partial-slab GEMM over (splits, waves), the fused kernels with rows vs slab prologues, and the whole decode
forward with the slab path on / off.  Each number = hipGraph of the 16 layers' launches, replayed; us per launch.
    python profiles/micro/draft_probe.py > gpurun_out/draft_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssd_amd.engine.llm_engine import LLMEngine  # noqa: E402
from ssd_amd.hip import ops as H  # noqa: E402
from ssd_amd.model import AttnMeta  # noqa: E402
from ssd_amd.model_config import PRESETS  # noqa: E402


def timed(fn, L, reps=8):
    for li in range(2):
        fn(li)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
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
def main():
    cfg = PRESETS["llama-3.2-1b"]
    eng = LLMEngine("llama-3.2-1b", hf_config=cfg, max_num_seqs=1, max_model_len=2048, max_num_batched_tokens=2048,
                    kvcache_block_size=256, num_kvcache_blocks=10)
    r = eng.model_runner
    m = r.model
    L = cfg.num_layers
    h, I, qn = m.h, m.I, m.qn
    r.d_slots[:64].fill_(-1)
    r.d_pos[:64].fill_(100)

    def row(name, us, mb):
        print(f"{name:58s} {us:8.2f} us   {mb / us if us else 0:7.2f} TB/s", flush=True)

    for M in (1, 8, 24):
        print(f"---- M = {M} ----")
        wo = lambda li: m.w[f"model.layers.{li}.self_attn.o_proj.weight"]
        wd = lambda li: m.w[f"model.layers.{li}.mlp.down_proj.weight"]
        mb_o, mb_d = h * qn * 2 / 1e6, h * I * 2 / 1e6
        row("o  current default (_gemm)", timed(lambda li: m._gemm(m.buf_af, qn, wo(li), h, m.buf_h, M, h), L), mb_o)
        row("down current default (_gemm)", timed(lambda li: m._gemm(m.buf_actf, I, wd(li), h, m.buf_h, M, h), L), mb_d)
        parts = torch.zeros(16 * 32 * h, dtype=torch.float32, device="cuda")
        for name, K, wsel, xbuf, mb in (("o", qn, wo, m.buf_af, mb_o), ("down", I, wd, m.buf_actf, mb_d)):
            for S in (2, 4):
                for waves in (4, 8, 16):
                    per = -(-(K // 32) // S)
                    if -(-per // waves) > 8 or (M > 16 and waves < 2):
                        continue
                    us = timed(lambda li: H.gemm_parts(xbuf, wsel(li), M, h, K, parts=parts, splits=S, waves=waves), L)
                    row(f"{name} parts S={S} waves={waves} ({h // 16 * S} WGs, {-(-per // waves)} tiles/wave)", us, mb)
        if M <= 16:
            small, nf = m.fusion_plan(M)
            print(f"fusion_plan: small={small} norm_fuse={nf}; parts_plan={m.parts_plan(M)} cfg_o={m._parts('o', M)} cfg_d={m._parts('d', M)}")
            for use in (False, True):
                m.use_parts = use
                row(f"qkv launch_qkv(li>0) use_parts={use}", timed(lambda li: m.launch_qkv(max(li, 1), M, r.d_pos, r.d_slots, gemm_only=True), L), m.qkv_n * h * 2 / 1e6)
                row(f"gate_up launch_gate_up use_parts={use}", timed(lambda li: m.launch_gate_up(li, M, gemm_only=True), L), 2 * I * h * 2 / 1e6)
            m.use_parts = True
            if nf:
                p = lambda li: f"model.layers.{li}."
                for waves in (8, 16):
                    for nt in (2, 4):
                        us = timed(lambda li: H.gemm_fused(m.w[p(li) + "mlp.gate_up_proj.weight"], M, 2 * I, h, H.FEPI_SILU_FRAG, res_in=m.buf_res2,
                                                            res_out=m.buf_res, norm_w=m.w[p(li) + "post_attention_layernorm.weight"], eps=1e-5,
                                                            y=m.buf_actf, h_parts=m.buf_parts_o, splits=m._parts('o', M)[0], nt=nt, waves=waves), L)
                        row(f"gate_up fused parts nt={nt} waves={waves}", us, 2 * I * h * 2 / 1e6)
        # attention at two context lengths
        for ctx in (150, 640):
            r.d_ctx[:1].fill_(ctx)
            r.d_bt[0, :8] = torch.arange(8, dtype=torch.int32, device="cuda")
            r._ctx_hint = 1024
            if M == 24:
                K_, mq = 7, 24
                jidx = torch.tensor([[i // 3 for i in range(24)]], dtype=torch.int32, device="cuda")
                r.d_ctx[:1].fill_(ctx + 8 + 24)
                meta = AttnMeta(H.MODE_TREE, 1, mq, r.d_slots, r.d_ctx, r.d_bt, q_per_seq=mq, tree_K=K_, tree_mq=mq, tree_step=0, tree_F=1,
                                tree_jidx=jidx, ctx_hint=1024)
            else:
                meta = AttnMeta(H.MODE_CAUSAL, 1, M, r.d_slots, r.d_ctx, r.d_bt, q_per_seq=M, ctx_hint=1024)
            splits, waves = m._attn_cfg(M, meta)
            scale = m.hd ** -0.5
            for wv, fl in [(w_, 0) for w_ in sorted({waves, 4, 8})] + ([(8, 4), (4, 4)] if M == 24 else []):
                us = timed(lambda li: H.attn_paged(m.buf_q, m.kv_cache[li, 0], m.kv_cache[li, 1], meta.block_tables, m.max_blocks, meta.context_lens,
                                                   1, M, M, m.nh, m.nkv, m.hd, m.block_size, scale, q_per_seq=meta.q_per_seq, mode=meta.mode,
                                                   tree_K=meta.tree_K, tree_mq=meta.tree_mq, tree_step=0, tree_F=1, tree_jidx=meta.tree_jidx,
                                                   splits=splits, flags=fl, ws_o=m.ws_o, ws_ml=m.ws_ml, out_frag=m.buf_af, waves=wv), L)
                row(f"attention ctx={ctx} waves={wv}{' rt=1' if fl else ''}{' (default)' if wv == waves and not fl else ''}", us, 2 * ctx * m.nkv * m.hd * 2 / 1e6)
        # attention + o_proj in one launch (round 3) vs the two launches
        if M <= 8:
            for ctx in (150, 640):
                r.d_ctx[:1].fill_(ctx)
                us = timed(lambda li: H.attn_oproj_parts(m.buf_q, m.kv_cache[li, 0], m.kv_cache[li, 1], r.d_bt, m.max_blocks, r.d_ctx, M, m.nh,
                                                         m.nkv, m.hd, m.block_size, m.hd ** -0.5, wo(li), h, m.buf_parts_o), L)
                row(f"attention + o_proj fused, ctx={ctx}", us, mb_o)
        # whole forward (all layers + head + argmax): slab path off / on, fused attention + o_proj off / on, fused argmax off / on
        r.d_ctx[:1].fill_(150)
        r._ctx_hint = 1024
        for use, ao, am in (((False, False, True), (True, False, False), (True, False, True), (True, True, True)) if M <= m.max_logit_rows else ()):
            m.use_parts, m.fuse_attn_o, m.argmax_fused = use, ao, am
            meta = AttnMeta(H.MODE_CAUSAL, 1, M, r.d_slots, r.d_ctx, r.d_bt, q_per_seq=M, ctx_hint=1024)

            def fwd(_):
                m.forward(r.d_ids, r.d_pos, M, meta)
                m.compute_logits(M)
                m.argmax(M, r.d_next)
            us = timed(fwd, 1, reps=20)
            print(f"WHOLE FORWARD M={M} slabs={use} attn+o fused={ao} argmax fused={am}: {us:9.1f} us   ({m.weight_bytes() / us * 1e-6:.2f} TB/s)", flush=True)
        m.use_parts, m.fuse_attn_o, m.argmax_fused = True, True, True
    # the small ops
    print("---- small ops, M = 1 ----")
    row("embedding", timed(lambda li: H.embedding(r.d_ids, m.w["model.embed_tokens.weight"], m.buf_h, 1, h), 16), 0)
    row("final rmsnorm", timed(lambda li: H.rmsnorm(m.buf_h, m.w["model.norm.weight"], 1e-5, 1, h, res_in=m.buf_res, out_frag=m.buf_lastf), 16), 0)
    row("lm head", timed(lambda li: m._gemm(m.buf_lastf, h, m.w["lm_head.weight"], m.V, m.logits, 1, m.V), 4), m.V * h * 2 / 1e6)
    row("argmax V=128256", timed(lambda li: H.argmax_rows(m.logits, m.V, 1, m.V, r.d_next), 16), 0.25)
    m._last_parts = False
    row("lm head + argmax candidates", timed(lambda li: m.compute_logits(1), 4), m.V * h * 2 / 1e6)
    row("argmax from candidates", timed(lambda li: m.argmax(1, r.d_next), 16), 0.03)


main()
