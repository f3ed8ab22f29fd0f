"""The residual add + RMSNorm carried by the GEMMs on either side of it (csrc/xsum.h; ssd_gemm_wf_res, ssd_gemm_wf_xs, ssd_gemm_fused_xs),
through the C ABI, against (a) the oracle (RMSDNorm.forward as compiled, ssd/layers/layernorm.py:64-88, between F.linear calls --
LlamaDecoderLayer.forward, ssd/models/llama3.py:185-199) and (b) the separate launches they replace (ssd_gemm_wf + ssd_rmsnorm +
ssd_gemm_wf / ssd_gemm_fused).  Same rounding points; the row's sum of squares is added in a different fp32 order, so:
  * what carries no sum -- the new residual, the fp32 x -- must be BIT-identical to the separate launches,
  * the row scale must agree to a few fp32 ulps,
  * the consumer's outputs are held to the same oracle bars as the separate launches' (tests/test_real_shapes_gpu.py) and may differ
    from them by one bf16 ulp on a small fraction of elements.
Shapes: the 70B geometry of the metric's verify (h 8192, I 28672, 64 / 8 heads x 128) and a second one that takes the two-row-group
decomposition of the consumer."""
import dataclasses
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import layout as LY
from oracle import ops as O
from ssd_amd.model_config import PRESETS
from tests.util import assert_close_bf16

BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    return ops


def rand_w(N, K, seed, std=0.03):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(N, K, device="cuda", generator=g) * std).to(BF)


def frag(H, w, mode=0):
    out = torch.empty(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag(w, out, w.shape[0], w.shape[1], mode)
    return out


def x32_rows(x32f: torch.Tensor, M: int, K: int) -> torch.Tensor:
    """fp32 fragment-major [16][K] -> rows [M][K]: tile kt = 64 lanes x 8 floats, lane = (m & 15) + 16 * ((k & 31) >> 3)."""
    t = x32f.cpu().view(K // 32, 4, 16, 8)              # [kt][k8 & 3][m][8]
    return t.permute(2, 0, 1, 3).reshape(16, K)[:M].contiguous()


@pytest.mark.parametrize("label,h,qn,I", [("70b", 8192, 8192, 28672), ("two-row-group gate_up", 8192, 8192, 8192)])
def test_res_epilogue_and_xs_gate_up_vs_separate_launches_and_oracle(H, label, h, qn, I):
    assert H.xsum_ok(8, h, qn, I, 10240)
    wo = rand_w(h, qn, 1)
    wgu = rand_w(2 * I, h, 2, std=0.04)
    wo_f, wgu_f = frag(H, wo), frag(H, wgu, mode=1)
    wo_c, wgu_c = wo.cpu(), wgu.cpu()
    for M in (1, 7, 8, 16):
        torch.manual_seed(10 + M)
        a = torch.randn(M, qn).to(BF)                    # attention output
        res = torch.randn(M, h).to(BF)
        nw = (1 + 0.1 * torch.randn(h)).to(BF)
        eps = 1e-5
        a_f = LY.rows_to_frag_ref(a).cuda()
        # --- the separate launches ---
        y = torch.zeros(M, h, dtype=BF, device="cuda")
        H.gemm(a_f, wo_f, y, M, h, qn, h)
        res_sep = res.cuda().clone()
        xf = torch.zeros(H.frag_numel(M, h), dtype=BF, device="cuda")
        H.rmsnorm(y, nw.cuda(), eps, M, h, res_in=res_sep, res_out=res_sep, out_frag=xf)
        act_sep = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
        H.gemm(xf, wgu_f, act_sep, M, 2 * I, h, 0, epilogue=H.EPI_SILU_FRAG)
        # --- the norm-carrying forms (residual in place) ---
        res_x = res.cuda().clone()
        x32 = torch.full((16 * h,), float("nan"), dtype=torch.float32, device="cuda")
        gss = torch.full((h,), float("nan"), dtype=torch.float32, device="cuda")
        H.gemm_res(a_f, wo_f, res_x, res_x, x32, gss, M, h, qn)
        act_x = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
        H.gemm_xs(x32, gss, nw.cuda(), eps, wgu_f, act_x, M, 2 * I, h)
        torch.cuda.synchronize()
        # residual: no sum involved -> the separate launches' bits
        assert torch.equal(res_x.cpu().view(torch.int16), res_sep.cpu().view(torch.int16)), f"{label} M={M}: residual differs"
        # fp32 x = fp32(bf16 y) + fp32(res), exactly
        want32 = y.cpu().float() + res.float()
        got32 = x32_rows(x32, M, h)
        assert torch.equal(got32, want32), f"{label} M={M}: fp32 x differs from fp32(y) + fp32(res)"
        # group sums -> the row's mean square within a few fp32 ulps of the float64 value
        g = gss.cpu().view(h // 16, 16)[:, :M].double().sum(0)
        exact = want32.double().pow(2).sum(-1)
        assert ((g - exact).abs() / exact).max().item() < 2e-6, f"{label} M={M}: group sums"
        # the consumer: vs the oracle at the separate launches' bar, vs the separate launches within one ulp on few elements
        ref_y = O.linear(a, wo_c)
        ref_x, _ = O.rmsnorm(ref_y, nw, eps, residual=res)
        ref_act = O.silu_mul(O.linear(ref_x, wgu_c))
        ax, asep = LY.frag_to_rows_ref(act_x.cpu(), M, I), LY.frag_to_rows_ref(act_sep.cpu(), M, I)
        d_x, d_s = (ax.float() - ref_act.float()).abs(), (asep.float() - ref_act.float()).abs()
        print(f"{label} M={M}: |xs - oracle| max {d_x.max():.5f} mean {d_x.mean():.6f}; |separate - oracle| max {d_s.max():.5f} mean {d_s.mean():.6f}; "
              f"xs != separate on {(ax.view(torch.int16) != asep.view(torch.int16)).float().mean():.5f} of the outputs")
        assert torch.isfinite(ax.float()).all()
        assert d_x.mean().item() <= 1.1 * d_s.mean().item() + 1e-6 and d_x.max().item() <= 1.5 * d_s.max().item() + 1e-6
        # (SiLU(g) * u of two operands that may each sit an ulp off: the product moves by up to ~4 ulps on a handful of outputs -- first
        #  run on MI355X: identical bits at M = 1 / 7 / 8, 0.3 % of the outputs different at M = 16, 3 ulps at most)
        assert_close_bf16(ax, asep, max_ulp=4, max_frac=0.02, rel_floor=2 ** -7, what=f"{label} xs vs separate M={M}")


def test_res_down_proj_and_xs_qkv_rope_store_vs_separate_launches_and_oracle(H):
    """down_proj + add, then the next layer's norm + QKV + RoPE + paged KV store (70B geometry)."""
    h, I, nh, nkv, hd = 8192, 28672, 64, 8, 128
    N = (nh + 2 * nkv) * hd
    bs, nb = 256, 2
    wd = rand_w(h, I, 3, std=0.02)
    wq = rand_w(N, h, 4)
    wd_f = frag(H, wd)
    wq_f = torch.empty(wq.numel(), dtype=BF, device="cuda")
    H.rows_to_frag_qkv(wq, wq_f, nh, nkv, hd, h)
    wd_c, wq_c = wd.cpu(), wq.cpu()
    cache = O.make_cos_sin_cache(hd, 1024, 5e5)
    for M in (1, 8, 16):
        torch.manual_seed(20 + M)
        act = (torch.randn(M, I) * 0.5).to(BF)
        res = torch.randn(M, h).to(BF)
        nw = (1 + 0.1 * torch.randn(h)).to(BF)
        eps = 1e-5
        pos = torch.randint(0, 900, (M,), dtype=torch.int64)
        slots = torch.randperm(nb * bs)[:M].to(torch.int32)
        act_f = LY.rows_to_frag_ref(act).cuda()
        rope = dict(positions=pos.cuda(), cos_sin=cache.cuda(), slots=slots.cuda(), nh=nh, nkv=nkv, hd=hd, block_size=bs)

        def outs():
            return (torch.zeros(M, nh * hd, dtype=BF, device="cuda"), torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda"),
                    torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda"))
        # separate launches
        y = torch.zeros(M, h, dtype=BF, device="cuda")
        H.gemm(act_f, wd_f, y, M, h, I, h)
        res_sep = res.cuda().clone()
        xf = torch.zeros(H.frag_numel(M, h), dtype=BF, device="cuda")
        H.rmsnorm(y, nw.cuda(), eps, M, h, res_in=res_sep, res_out=res_sep, out_frag=xf)
        q_s, k_s, v_s = outs()
        H.gemm_fused(wq_f, M, N, h, H.FEPI_QKV_ROPE, x_frag=xf, q_out=q_s, k_cache=k_s, v_cache=v_s, **rope)
        # norm-carrying forms
        res_x = res.cuda().clone()
        x32 = torch.zeros(16 * h, dtype=torch.float32, device="cuda")
        gss = torch.zeros(h, dtype=torch.float32, device="cuda")
        H.gemm_res(act_f, wd_f, res_x, res_x, x32, gss, M, h, I)
        q_x, k_x, v_x = outs()
        H.gemm_fused_xs(x32, gss, nw.cuda(), eps, wq_f, M, N, h, q_out=q_x, k_cache=k_x, v_cache=v_x, **rope)
        torch.cuda.synchronize()
        assert torch.equal(res_x.cpu().view(torch.int16), res_sep.cpu().view(torch.int16))
        # oracle
        ref_x, _ = O.rmsnorm(O.linear(act, wd_c), nw, eps, residual=res)
        qkv = O.linear(ref_x, wq_c)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
        q, k = O.rope(pos, q.contiguous(), k.contiguous(), cache, hd)
        kref = torch.zeros(nb, bs, nkv, hd, dtype=BF)
        vref = torch.zeros_like(kref)
        O.store_kv(k.view(M, nkv, hd), v.contiguous().view(M, nkv, hd), kref, vref, slots)
        floor = float(2.0 ** (math.floor(math.log2(max(q.abs().max().item(), 1e-3))) - 7))
        for what, gx, gs, ref in (("q", q_x.cpu(), q_s.cpu(), q), ("k", LY.kv_hnd_to_nhd(k_x.cpu()), LY.kv_hnd_to_nhd(k_s.cpu()), kref),
                                  ("v", LY.kv_hnd_to_nhd(v_x.cpu()), LY.kv_hnd_to_nhd(v_s.cpu()), vref)):
            d_x, d_s = (gx.float() - ref.float()).abs(), (gs.float() - ref.float()).abs()
            print(f"70b M={M} {what}: |xs - oracle| max {d_x.max():.5f} mean {d_x.mean():.6f}; |separate - oracle| max {d_s.max():.5f} mean {d_s.mean():.6f}")
            assert d_x.mean().item() <= 1.1 * d_s.mean().item() + 1e-6 and d_x.max().item() <= 1.5 * d_s.max().item() + floor
            assert_close_bf16(gx, gs, max_ulp=2, max_frac=0.02, rel_floor=2 ** -7, abs_floor=floor, what=f"xs vs separate {what} M={M}")


def test_three_layer_70b_cut_with_and_without_the_norm_carrying_gemms_vs_oracle(H):
    """HipDecoder.forward at the 70B geometry, M = 8 verify rows after a prefill, three layers + LM head: the default path (xsum on)
    and the separate launches (xsum off) against the oracle model; the xsum path may be no further from the oracle than the separate
    launches are, the greedy tokens agree outside near-ties, the new K / V rows of the last layer agree to the propagated-noise bar.
    Also under hipGraph replay (the form the engine runs): bit-identical to the eager run."""
    from oracle.model import OracleModel, Ctx
    from ssd_amd import weights as W
    from ssd_amd.model import HipDecoder, AttnMeta
    L = 3
    cfg = dataclasses.replace(PRESETS["llama-3.1-70b"], num_layers=L)
    bs, nblocks = 256, 2
    full = {}
    decs = {}
    for on in (True, False):
        dec = HipDecoder(cfg, max_tokens=64, max_seqs=1, max_blocks=2, block_size=bs, max_model_len=512, device=torch.device("cuda", 0))

        def both():
            for name, t in W.synthetic_weights(cfg, 31, 0.02, gen_device="cuda"):
                if on:
                    full[name] = t.cpu()
                yield name, t
        dec.load_weights(both())
        dec.alloc_kv(nblocks)
        assert dec.xsum, "the 70B geometry must qualify for the norm-carrying forms"
        dec.xsum = on
        decs[on] = dec
    orc = OracleModel(cfg, full, nblocks, bs)
    random.seed(3)
    P, M = 24, 8
    prompt = [random.randint(0, 100000) for _ in range(P)]
    vt = [random.randint(0, 100000) for _ in range(M)]
    table = [1, 0]
    bt = torch.tensor([table], dtype=torch.int32)

    def slots(ps):
        return torch.tensor([table[p // bs] * bs + p % bs for p in ps], dtype=torch.int32)

    def i64(x):
        return torch.tensor(list(x), dtype=torch.int64)

    def i32(x):
        return torch.tensor(list(x), dtype=torch.int32)

    cu = i32([0, P])
    orc.forward(i64(prompt), i64(range(P)), Ctx("prefill", slot_mapping=slots(range(P)), cu_q=cu, cu_k=cu))
    vp = list(range(P, P + M))
    ref = orc.compute_logits(orc.forward(i64(vt), i64(vp), Ctx("verify", slot_mapping=slots(vp), context_lens=i32([P + M]), block_tables=bt,
                                                             cu_q=i32([0, M])))).float()
    got = {}
    for on, dec in decs.items():
        dec.forward(i64(prompt).cuda(), i64(range(P)).cuda(), P, AttnMeta(H.MODE_CAUSAL, 1, P, slots(range(P)).cuda(), i32([P]).cuda(), bt.cuda(), cu_q=cu.cuda()))
        meta = AttnMeta(H.MODE_CAUSAL, 1, M, slots(vp).cuda(), i32([P + M]).cuda(), bt.cuda(), q_per_seq=M)
        ids, pos = i64(vt).cuda(), i64(vp).cuda()
        assert dec.xsum_plan(M, meta) == on
        dec.forward(ids, pos, M, meta)
        n = dec.compute_logits(M)
        torch.cuda.synchronize()
        got[on] = dec.logits[:n].float().cpu()
        if on:      # hipGraph replay == eager, bit for bit
            eager = dec.logits[:n].clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    dec.forward(ids, pos, M, meta)
                    dec.compute_logits(M)
                for _ in range(3):
                    dec.logits.zero_()
                    g.replay()
                    s.synchronize()
                    assert torch.equal(dec.logits[:n].view(torch.int16), eager.view(torch.int16))
            torch.cuda.synchronize()
    d_on, d_off = (got[True] - ref).abs(), (got[False] - ref).abs()
    print(f"70B x {L} layers M=8: |xsum - oracle| max {d_on.max():.4f} mean {d_on.mean():.5f}; |separate - oracle| max {d_off.max():.4f} mean "
          f"{d_off.mean():.5f}; |xsum - separate| max {(got[True] - got[False]).abs().max():.4f}")
    assert torch.isfinite(got[True]).all()
    assert d_on.mean().item() <= 1.25 * d_off.mean().item() + 1e-4 and d_on.max().item() <= 1.5 * d_off.max().item() + 1e-3
    top2 = ref.topk(2, dim=-1).values
    thr = torch.clamp(2 * d_on.max(-1).values, min=0.0625)
    assert bool(((got[True].argmax(-1) == ref.argmax(-1)) | ((top2[:, 0] - top2[:, 1]) < thr)).all())
    for which in (0, 1):
        ref_rows = torch.stack([orc.kv_cache[which, L - 1, table[p // bs], p % bs] for p in vp]).float()
        rows = {on: torch.stack([dec.kv_cache[L - 1, which, table[p // bs], :, p % bs, :] for p in vp]).cpu().float() for on, dec in decs.items()}
        tol = 2.0 ** (math.floor(math.log2(ref_rows.abs().max().item())) - 6)
        d1, d0 = (rows[True] - ref_rows).abs(), (rows[False] - ref_rows).abs()
        assert d1.max().item() <= max(tol, 1.5 * d0.max().item()) and d1.mean().item() <= max(tol / 8, 1.25 * d0.mean().item() + tol / 64)
