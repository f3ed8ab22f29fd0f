"""GPU parity of every libssdhip kernel, called through the C ABI, against the CPU oracle and against the
golden vectors produced by the reference's own code.  Bit-exact for integer results; for bf16 results the
bar is "identical up to the accumulation order": at most 1 bf16 ulp on a small fraction of elements (2 ulp for
attention, whose P operand is bf16 as in FlashAttention).  fp32 GEMM output is checked to 1e-3 absolute."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O
from oracle import layout as LY
from tests.util import assert_close_bf16

BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssd_amd.hip import ops
    from ssd_amd.hip.lib import load_library
    load_library()
    return ops


def dev(t):
    return t.cuda().contiguous()


def to_frag_dev(x):
    return dev(LY.rows_to_frag_ref(x))


# ------------------------------------------------------------------------------------------------
def test_layout_kernels(H):
    torch.manual_seed(0)
    for R, K in [(7, 64), (16, 256), (40, 4096), (512, 128)]:
        x = torch.randn(R, K).to(BF)
        ref = LY.rows_to_frag_ref(x)
        out = torch.zeros(H.frag_numel(R, K), dtype=BF, device="cuda")
        H.rows_to_frag(dev(x), out, R, K)
        assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
        back = torch.empty(R, K, dtype=BF, device="cuda")
        H.frag_to_rows(out, back, R, K)
        assert torch.equal(back.cpu().view(torch.int16), x.view(torch.int16))
    w = torch.randn(128, 64).to(BF)   # gate/up interleave
    out = torch.zeros(128 * 64, dtype=BF, device="cuda")
    H.rows_to_frag(dev(w), out, 128, 64, mode=1)
    assert torch.equal(out.cpu().view(torch.int16), LY.rows_to_frag_ref(LY.interleave_gate_up_rows(w)).view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (7, 6144, 4096), (8, 4096, 14336), (16, 512, 2048), (24, 2048, 2048),
                                   (33, 256, 1024), (100, 1024, 512), (128, 512, 4096), (7, 128256 // 8, 1024), (5, 272, 96)])
def test_gemm_vs_oracle(H, M, N, K):
    torch.manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    # asymmetric structure so a transposed / permuted tile cannot pass
    w[3, :] = 0.5
    x[M - 1, : K // 2] = -1.0
    ref = O.linear(x, w)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    H.gemm(xf, wf, y, M, N, K, N)
    assert_close_bf16(y, ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="gemm rows")
    # fp32 epilogue against an fp64 reference: <= 1e-3 abs (north_star logits tolerance)
    y32 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    H.gemm(xf, wf, y32, M, N, K, N, epilogue=H.EPI_ROWS_F32)
    ref64 = (x.double() @ w.double().t())
    assert (y32.cpu().double() - ref64).abs().max().item() <= 1e-3


def test_gemm_configs_and_bias(H):
    torch.manual_seed(5)
    M, N, K = 7, 1024, 2048
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    b = torch.randn(N).to(BF)
    ref = O.linear(x, w, b)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    outs = []
    for nt in (1, 2, 4):
        for waves in (1, 3, 4, 8, 16):
            y = torch.zeros(M, N, dtype=BF, device="cuda")
            H.gemm(xf, wf, y, M, N, K, N, bias=dev(b), cfg=(nt, waves))
            assert_close_bf16(y, ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what=f"gemm cfg {nt},{waves}")
            outs.append(y.clone())
    # determinism: same config twice -> identical bits
    y2 = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm(xf, wf, y2, M, N, K, N, bias=dev(b), cfg=(4, 16))
    assert torch.equal(y2.view(torch.int16), outs[-1].view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(1, 256, 2048), (8, 512, 8192), (16, 1024, 4128), (7, 256, 96)])
def test_gemm_deep_form_vs_oracle(H, M, N, K):
    """The DEEP form of the skinny GEMM (round 6: twice the k-tiles per stage at <= 8 waves; bit 8 of ssd_gemm_wf_cfg's nt), rows and
    SiLU epilogues, with and without a K remainder and consecutive tiles per workgroup: the plain form's bars against the oracle, the
    plain form's bits on repetition (deterministic), and -- a different dealing of k-tiles to waves, i.e. another fp32 order -- within
    one ulp of the plain form on a small fraction of the outputs."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    b = torch.randn(N).to(BF)
    ref = O.linear(x, w, b)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    for nt in (2, 4):
        for waves, tpw in ((8, 1), (4, 2), (1, 4), (8, 3)):
            plain = torch.zeros(M, N, dtype=BF, device="cuda")
            H.gemm(xf, wf, plain, M, N, K, N, bias=dev(b), cfg=(nt, waves | (tpw << 8)))
            outs = []
            for _ in range(2):
                y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
                H.gemm(xf, wf, y, M, N, K, N, bias=dev(b), cfg=(nt | 256, waves | (tpw << 8)))
                outs.append(y)
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
            assert_close_bf16(outs[0], ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what=f"deep gemm nt {nt} waves {waves} tpw {tpw}")
            assert_close_bf16(outs[0], plain, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what=f"deep vs plain nt {nt} waves {waves} tpw {tpw}")
    # SiLU epilogue (gate / up row groups interleaved)
    wg = (torch.randn(N, K) * 0.05).to(BF)
    refa = O.silu_mul(O.linear(x, wg))
    wgf = torch.zeros(N * K, dtype=BF, device="cuda")
    H.rows_to_frag(dev(wg), wgf, N, K, mode=1)
    for nt, waves, tpw in ((2, 8, 1), (4, 8, 2), (2, 4, 4)):
        act_f = torch.zeros(H.frag_numel(M, N // 2), dtype=BF, device="cuda")
        H.gemm(xf, wgf, act_f, M, N, K, 0, epilogue=H.EPI_SILU_FRAG, cfg=(nt | 256, waves | (tpw << 8)))
        act = LY.frag_to_rows_ref(act_f.cpu(), M, N // 2)
        assert_close_bf16(act, refa, max_ulp=2, max_frac=0.04, rel_floor=2 ** -7, what=f"deep silu nt {nt} waves {waves} tpw {tpw}")
    # what the DEEP form does not take is refused, not mis-launched
    with pytest.raises(Exception):
        H.gemm(xf, wf, plain, M, N, K, N, cfg=(2 | 256, 16))
    with pytest.raises(Exception):
        H.gemm(xf, wf, plain, M, N, K, N, cfg=(1 | 256, 8))


@pytest.mark.parametrize("M,N,K", [(1, 256, 2048), (7, 128, 8192), (16, 2048, 2048), (3, 80, 96), (1, 2048, 8192)])
def test_gemm_splitk_vs_oracle(H, M, N, K):
    """csrc/gemm_sk.hip: K split across workgroups, last arriver reduces in z order; counters must return to zero and
    the result must not depend on the run (deterministic order)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    b = torch.randn(N).to(BF)
    ref = O.linear(x, w, b)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    counters = torch.zeros(N // 16, dtype=torch.int32, device="cuda")
    for splits in (1, 2, 3, 4, 8):
        if (K // 32) < splits:
            continue
        ws = torch.zeros((N // 16) * splits * 256, dtype=torch.float32, device="cuda")
        outs = []
        for waves in (1, 4, 8, 16):
            for _ in range(2):
                y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
                H.gemm_splitk(xf, wf, y, M, N, K, N, splits, waves, ws, counters, bias=dev(b))
                outs.append(y)
            assert torch.equal(outs[-1].view(torch.int16), outs[-2].view(torch.int16))
            assert_close_bf16(outs[-1], ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what=f"split-K gemm S{splits} w{waves}")
        assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("M,N,K,splits", [(128, 512, 4096, 0), (100, 1024, 512, 0), (17, 256, 1024, 2), (64, 256, 128, 1),
                                          (128, 512, 1024, 1), (90, 128, 2048, 1),
                                          (33, 2048, 2048, 0), (128, 128, 8192, 16), (65, 384, 256, 0)])
def test_gemm_prefill_vs_oracle(H, M, N, K, splits):
    """csrc/gemm_pf.hip: x tile shared through LDS, K split across workgroups, fp32 partials summed in order."""
    torch.manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    b = torch.randn(N).to(BF)
    w[3, :] = 0.5
    x[M - 1, : K // 2] = -1.0
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    ws = torch.zeros(max(H.gemm_pf_workspace_bytes(M, N, K), 16 * M * N * 4) // 4, dtype=torch.float32, device="cuda")
    for bias in (None, b):
        y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        H.gemm_pf(xf, wf, y, M, N, K, N, ws, bias=None if bias is None else dev(bias), splits=splits)
        assert_close_bf16(y, O.linear(x, w, bias), max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="prefill gemm rows")
    y2 = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm_pf(xf, wf, y2, M, N, K, N, ws, bias=dev(b), splits=splits)
    assert torch.equal(y2.view(torch.int16), y.view(torch.int16))          # deterministic


@pytest.mark.parametrize("M,N,K,splits", [(128, 26880, 512, 2), (97, 26880, 256, 1), (128, 10240, 2048, 4)])
def test_gemm_prefill_launch_shapes_are_bit_identical(H, M, N, K, splits):
    """The launch-shape variants of csrc/gemm_pf.hip (3..8 waves per workgroup, one or two 16-row groups per wave, one or two
    k-steps per barrier) only change WHO computes a tile, never the K order inside a split: every variant must equal the plain
    4-wave form bit for bit, and that one the oracle.  (The 5-wave form is the default for the 70B qkv matrix, 10240 rows.)"""
    torch.manual_seed(N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    ws = torch.zeros(16 * M * N, dtype=torch.float32, device="cuda")

    def run(nt):
        y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        H.gemm_pf(xf, wf, y, M, N, K, N, ws, splits=splits, nt=nt)
        return y

    base = run(2 | 4 << 8 | 8 << 16 | 1 << 24)
    assert_close_bf16(base, O.linear(x, w), max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="prefill gemm rows")
    tried = 0
    for ntile, waves_list in ((2, (3, 4, 5, 6, 7, 8)), (1, (4, 8))):
        for waves in waves_list:
            if N % (16 * ntile * waves):
                continue
            for bps in (1, 2):
                if (ntile, waves, bps) in ((2, 6, 1),):
                    continue                      # not instantiated
                y = run(ntile | waves << 8 | 8 << 16 | bps << 24)
                assert torch.equal(y.view(torch.int16), base.view(torch.int16)), (ntile, waves, bps)
                tried += 1
    assert tried >= 8
    ydef = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    H.gemm_pf(xf, wf, ydef, M, N, K, N, ws, splits=splits)           # the default launch shape for this split count
    assert torch.equal(ydef.view(torch.int16), base.view(torch.int16))


@pytest.mark.parametrize("M,splits", [(40, 0), (128, 0), (100, 1), (128, 1)])
def test_gemm_prefill_silu_epilogue(H, M, splits):
    torch.manual_seed(M)
    I, K = 512, 1024
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(2 * I, K) * 0.06).to(BF)
    ref = O.silu_mul(O.linear(x, w))
    wf = torch.zeros(2 * I * K, dtype=BF, device="cuda")
    H.rows_to_frag(dev(w), wf, 2 * I, K, mode=1)
    act_f = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
    ws = torch.zeros(H.gemm_pf_workspace_bytes(M, 2 * I, K) // 4, dtype=torch.float32, device="cuda")
    H.gemm_pf(to_frag_dev(x), wf, act_f, M, 2 * I, K, 0, ws, epilogue=H.EPI_SILU_FRAG, splits=splits)   # splits=1: in-kernel epilogue
    act = LY.frag_to_rows_ref(act_f.cpu(), M, I)
    assert_close_bf16(act, ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="prefill gemm+silu")


@pytest.mark.parametrize("M", [1, 7, 24, 70])
def test_gemm_silu_epilogue(H, M):
    torch.manual_seed(M)
    I, K = 512, 1024
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(2 * I, K) * 0.06).to(BF)
    ref = O.silu_mul(O.linear(x, w))
    wf = torch.zeros(2 * I * K, dtype=BF, device="cuda")
    H.rows_to_frag(dev(w), wf, 2 * I, K, mode=1)
    act_f = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
    H.gemm(to_frag_dev(x), wf, act_f, M, 2 * I, K, 0, epilogue=H.EPI_SILU_FRAG)
    act = LY.frag_to_rows_ref(act_f.cpu(), M, I)
    assert_close_bf16(act, ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="gemm+silu")


def test_stand_alone_silu_mul_and_head_rmsnorm(H, golden):
    """The module-level entry points (ssd_silu_mul = SiluAndMul.forward, activation.py:11-14; ssd_head_rmsnorm = RMSHeadNorm.forward,
    layernorm.py:16-40) against the goldens the reference's own compiled modules wrote (ops_golden.npz), against the oracle at real
    widths, and against the fused forms the hot path runs: the gate_up epilogue's arithmetic, and -- bit for bit -- the q / k norm
    inside ssd_rope_store_kv (at position 0 the rotation is the identity, so its q output IS the normalised head)."""
    g = golden("ops_golden")
    T, I2 = g["silu_x"].shape
    y = torch.zeros(T, I2 // 2, dtype=BF, device="cuda")
    yf = torch.zeros(H.frag_numel(T, I2 // 2), dtype=BF, device="cuda")
    H.silu_mul(dev(g["silu_x"]), T, I2 // 2, out_rows=y, out_frag=yf)
    assert_close_bf16(y, g["silu_y"], max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="silu_mul golden")
    assert torch.equal(LY.frag_to_rows_ref(yf.cpu(), T, I2 // 2).view(torch.int16), y.cpu().view(torch.int16))
    torch.manual_seed(3)
    x = (torch.randn(24, 2 * 8192) * 2).to(BF)
    y = torch.zeros(24, 8192, dtype=BF, device="cuda")
    H.silu_mul(dev(x), 24, 8192, out_rows=y)
    assert_close_bf16(y, O.silu_mul(x), max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what="silu_mul 24 x 8192")
    # head norm: the reference's compiled RMSHeadNorm (28 heads of 64)
    n, hd = g["hnorm_x"].shape
    out = torch.zeros(n, hd, dtype=BF, device="cuda")
    H.head_rmsnorm(dev(g["hnorm_x"]), dev(g["hnorm_w"]), 1e-6, out, 1, n, hd)
    assert_close_bf16(out, g["hnorm_y"], max_ulp=1, max_frac=0.01, what="head_rmsnorm golden")
    for hd, nh, nkv in ((128, 16, 8), (64, 4, 2), (256, 2, 1)):
        T = 5
        qkv = torch.randn(T, (nh + 2 * nkv) * hd).to(BF)
        w = (1 + 0.1 * torch.randn(hd)).to(BF)
        out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
        q = qkv[:, :nh * hd].contiguous()
        H.head_rmsnorm(dev(q), dev(w), 1e-6, out, T, nh, hd)
        assert_close_bf16(out, O.rmsnorm(q.view(-1, hd), w, 1e-6).view(T, nh * hd), max_ulp=1, max_frac=0.01, what=f"head_rmsnorm hd {hd}")
        cache = torch.cat([torch.ones(4, hd // 2), torch.zeros(4, hd // 2)], dim=1).float().contiguous()     # cos = 1, sin = 0: no rotation
        q_out, _, _ = run_rope(H, qkv, torch.zeros(T, dtype=torch.int64), cache, torch.full((T,), -1, dtype=torch.int32), nh, nkv, hd, 16, 2,
                               qn=w, kn=w, eps=1e-6)
        assert torch.equal(q_out.view(torch.int16), out.cpu().view(torch.int16)), f"stand-alone head norm != the fused one, hd {hd}"


def test_embedding(H, golden):
    g = golden("ops_golden")
    out = torch.zeros(7, 256, dtype=BF, device="cuda")
    H.embedding(dev(g["emb_ids"]), dev(g["emb_w"]), out, 7, 256)
    assert torch.equal(out.cpu().view(torch.int16), g["emb_y"].view(torch.int16))
    # TP-style shard [500, 1000): rows outside give zeros
    H.embedding(dev(g["emb_ids"]), dev(g["emb_w"][500:].contiguous()), out, 7, 256, vocab_start=500, vocab_count=500)
    ref = O.embedding(g["emb_ids"], g["emb_w"][500:], 500)
    # masked rows: the reference's `mask * y` leaves -0.0 where y < 0; the sign of zero vanishes in the TP
    # all-reduce that always follows, so compare values, not bits
    assert torch.equal(out.cpu().float(), ref.float())


def test_rmsnorm_golden(H, golden):
    g = golden("ops_golden")
    T, Hd = g["norm_x"].shape
    y = torch.zeros(T, Hd, dtype=BF, device="cuda")
    yf = torch.zeros(H.frag_numel(T, Hd), dtype=BF, device="cuda")
    H.rmsnorm(dev(g["norm_x"]), dev(g["norm_w"]), 1e-5, T, Hd, out_rows=y, out_frag=yf)
    assert_close_bf16(y, g["norm_y"], max_ulp=1, max_frac=0.005, what="rmsnorm")
    assert torch.equal(LY.frag_to_rows_ref(yf.cpu(), T, Hd).view(torch.int16), y.cpu().view(torch.int16))
    res = torch.zeros(T, Hd, dtype=BF, device="cuda")
    H.rmsnorm(dev(g["norm_x"]), dev(g["norm_w"]), 1e-5, T, Hd, res_in=dev(g["norm_res"]), res_out=res, out_rows=y)
    assert torch.equal(res.cpu().view(torch.int16), g["addnorm_res"].view(torch.int16))
    assert_close_bf16(y, g["addnorm_y"], max_ulp=1, max_frac=0.005, what="add+rmsnorm")


@pytest.mark.parametrize("T,Hd", [(1, 2048), (7, 4096), (24, 8192), (3, 1024), (130, 5120)])
def test_rmsnorm_shapes(H, T, Hd):
    torch.manual_seed(T + Hd)
    x, r = torch.randn(T, Hd).to(BF), torch.randn(T, Hd).to(BF)
    w = (1 + 0.1 * torch.randn(Hd)).to(BF)
    yr, rr = O.rmsnorm(x, w, 1e-6, r)
    y = torch.zeros(T, Hd, dtype=BF, device="cuda")
    res = torch.zeros(T, Hd, dtype=BF, device="cuda")
    H.rmsnorm(dev(x), dev(w), 1e-6, T, Hd, res_in=dev(r), res_out=res, out_rows=y)
    assert torch.equal(res.cpu().view(torch.int16), rr.view(torch.int16))
    assert_close_bf16(y, yr, max_ulp=1, max_frac=0.005, what="add+rmsnorm")
    # gather (prefill last-token rows)
    idx = torch.tensor([T - 1, 0], dtype=torch.int32)
    yg = torch.zeros(2, Hd, dtype=BF, device="cuda")
    H.rmsnorm(dev(x), dev(w), 1e-6, 2, Hd, res_in=dev(r), out_rows=yg, gather=dev(idx))
    assert torch.equal(yg.cpu().view(torch.int16), y.cpu()[idx.long()].view(torch.int16))


def run_rope(H, qkv, pos, cache, slots, nh, nkv, hd, bs, nblocks, qn=None, kn=None, eps=0.0):
    T = qkv.shape[0]
    q_out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
    kc = torch.zeros(nblocks, nkv, bs, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    H.rope_store_kv(dev(qkv), dev(pos), dev(cache), dev(slots), q_out, kc, vc, T, nh, nkv, hd, bs,
                    q_norm_w=None if qn is None else dev(qn), k_norm_w=None if kn is None else dev(kn), eps=eps)
    return q_out.cpu(), kc.cpu(), vc.cpu()


def test_rope_store_golden(H, golden):
    g = golden("ops_golden")
    T, nh, nkv, hd, bs, nb = 7, 4, 2, 64, 16, 6
    v = torch.randn(T, nkv * hd).to(BF)
    qkv = torch.cat([g["rope_q"], g["rope_k"], v], dim=1)
    slots = torch.tensor([5, 17, -1, 40, 95, 0, 33], dtype=torch.int32)
    q_out, kc, vc = run_rope(H, qkv, g["rope_pos"], g["rope_cache"], slots, nh, nkv, hd, bs, nb)
    assert torch.equal(q_out.view(torch.int16), g["rope_qo"].view(torch.int16))          # bit-exact vs the reference
    kref = torch.zeros(nb, bs, nkv, hd, dtype=BF)
    vref = torch.zeros_like(kref)
    O.store_kv(g["rope_ko"].view(T, nkv, hd), v.view(T, nkv, hd), kref, vref, slots)
    assert torch.equal(LY.kv_hnd_to_nhd(kc).view(torch.int16), kref.view(torch.int16))
    assert torch.equal(LY.kv_hnd_to_nhd(vc).view(torch.int16), vref.view(torch.int16))


@pytest.mark.parametrize("qk_norm,S", [(False, 4), (True, 5), (False, 1), (True, 8)])
def test_rope_store_from_prefill_partials_with_several_workgroups_per_row(H, qk_norm, S):
    """Round 6: ssd_rope_store_kv_parts spreads a token row over 256-thread workgroups (70B-like head counts: 3 per row) and issues
    the slabs' loads four at a time -- still bit-identical to the row form over bf16(slab 0 + slab 1 + ...), for slab counts on both
    sides of the batch size, with per-head norms (the shuffles of a head stay inside one wave), -0.0 sums and skipped slots."""
    torch.manual_seed(3 + S)
    T, nh, nkv, hd, bs, nb = 70, 40, 8, 128, 16, 8
    N = (nh + 2 * nkv) * hd
    parts = torch.randn(S, T, N) * 0.5
    parts[:, 3, :200] = -0.0
    rows = parts[0].clone()
    for z in range(1, S):
        rows = rows + parts[z]
    rows = rows.to(BF)
    qn, kn = ((1 + 0.1 * torch.randn(hd)).to(BF), (1 + 0.1 * torch.randn(hd)).to(BF)) if qk_norm else (None, None)
    pos = torch.randint(0, 250, (T,), dtype=torch.int64)
    cache = O.make_cos_sin_cache(hd, 256, 5e5)
    slots = torch.randperm(nb * bs)[:T].to(torch.int32)
    slots[11] = -1

    def run(fn, src, *extra):
        q_out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
        kc = torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        fn(src, *extra, dev(pos), dev(cache), dev(slots), q_out, kc, vc, T, nh, nkv, hd, bs,
           q_norm_w=None if qn is None else dev(qn), k_norm_w=None if kn is None else dev(kn), eps=1e-6, qkv_perm=1)
        torch.cuda.synchronize()
        return [t.view(torch.int16) for t in (q_out, kc, vc)]

    a = run(H.rope_store_kv, dev(rows))
    b = run(H.rope_store_kv_parts, dev(parts.contiguous()), S)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("qk_norm,perm", [(False, 1), (True, 1), (False, 0)])
def test_rope_store_from_prefill_partials_equals_rows(H, qk_norm, perm):
    """ssd_rope_store_kv_parts (round 3): the prefill QKV GEMM leaves its split-K slabs in the workspace and the RoPE / KV-store
    kernel sums them -- bit-identical to the GEMM's epilogue launch followed by ssd_rope_store_kv, both on hand-made slabs (incl.
    -0.0 sums and skipped slots) and behind the real GEMM at its default split count."""
    torch.manual_seed(11 + perm)
    T, nh, nkv, hd, bs, nb, S = 100, 8, 2, 128, 16, 8, 4
    N = (nh + 2 * nkv) * hd
    parts = torch.randn(S, T, N) * 0.5
    parts[:, 3, :64] = -0.0                                   # a -0.0 + -0.0 + ... sum must stay -0.0 like the epilogue's
    parts[1:, 5, :] = 0.0
    rows = parts[0].clone()
    for z in range(1, S):
        rows = rows + parts[z]                                # fp32, slab order
    rows = rows.to(BF)
    qn, kn = ((1 + 0.1 * torch.randn(hd)).to(BF), (1 + 0.1 * torch.randn(hd)).to(BF)) if qk_norm else (None, None)
    pos = torch.randint(0, 250, (T,), dtype=torch.int64)
    cache = O.make_cos_sin_cache(hd, 256, 5e5)
    slots = torch.randperm(nb * bs)[:T].to(torch.int32)
    slots[7] = -1

    def run(fn, src, *extra):
        q_out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
        kc = torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        fn(src, *extra, dev(pos), dev(cache), dev(slots), q_out, kc, vc, T, nh, nkv, hd, bs,
           q_norm_w=None if qn is None else dev(qn), k_norm_w=None if kn is None else dev(kn), eps=1e-6, qkv_perm=perm)
        torch.cuda.synchronize()
        return [t.view(torch.int16) for t in (q_out, kc, vc)]

    a = run(H.rope_store_kv, dev(rows))
    b = run(H.rope_store_kv_parts, dev(parts.contiguous()), S)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # behind the real prefill GEMM (default split count of this shape)
    M, K = T, 2048
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.03).to(BF)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    nbytes = H.gemm_pf_workspace_bytes(M, N, K)
    Sd = nbytes // (4 * M * N)
    assert Sd > 1
    ws = torch.full((nbytes // 4 + 64,), float("nan"), dtype=torch.float32, device="cuda")
    y = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm_pf(xf, wf, y, M, N, K, N, ws)
    a = run(H.rope_store_kv, y)
    ws.fill_(float("nan"))
    H.gemm_pf(xf, wf, None, M, N, K, N, ws, epilogue=H.PF_EPI_PARTIALS)
    b = run(H.rope_store_kv_parts, ws, Sd)
    for x_, y_ in zip(a, b):
        assert torch.equal(x_, y_)


def test_rope_head_norm(H):
    torch.manual_seed(3)
    T, nh, nkv, hd, bs, nb = 5, 8, 2, 128, 16, 4
    qkv = torch.randn(T, (nh + 2 * nkv) * hd).to(BF)
    qn, kn = (1 + 0.1 * torch.randn(hd)).to(BF), (1 + 0.1 * torch.randn(hd)).to(BF)
    pos = torch.tensor([0, 3, 9, 100, 101], dtype=torch.int64)
    cache = O.make_cos_sin_cache(hd, 256, 1e6)
    slots = torch.arange(T, dtype=torch.int32) + 7
    q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=1)
    qr = O.rmsnorm(q.reshape(-1, hd), qn, 1e-6).reshape(q.shape)
    kr = O.rmsnorm(k.reshape(-1, hd), kn, 1e-6).reshape(k.shape)
    qr, kr = O.rope(pos, qr, kr, cache, hd)
    q_out, kc, vc = run_rope(H, qkv, pos, cache, slots, nh, nkv, hd, bs, nb, qn, kn, 1e-6)
    assert_close_bf16(q_out, qr, max_ulp=1, max_frac=0.01, what="qk-norm+rope q")
    kref = torch.zeros(nb, bs, nkv, hd, dtype=BF)
    vref = torch.zeros_like(kref)
    O.store_kv(kr.view(T, nkv, hd), v.contiguous().view(T, nkv, hd), kref, vref, slots)
    assert_close_bf16(LY.kv_hnd_to_nhd(kc), kref, max_ulp=1, max_frac=0.01, what="qk-norm+rope k")
    assert torch.equal(LY.kv_hnd_to_nhd(vc).view(torch.int16), vref.view(torch.int16))


# ------------------------------------------------------------------------------------------------
def make_paged(B, ctx_lens, nkv, hd, bs, seed):
    """Random paged K/V in the reference layout with a shuffled page table."""
    g = torch.Generator().manual_seed(seed)
    max_blocks = max((L + bs - 1) // bs for L in ctx_lens) + 1
    nblocks = B * max_blocks + 3
    perm = torch.randperm(nblocks, generator=g)
    bt = torch.full((B, max_blocks), -1, dtype=torch.int32)
    p = 0
    for b, L in enumerate(ctx_lens):
        n = (L + bs - 1) // bs
        bt[b, :n] = perm[p:p + n].to(torch.int32)
        p += n
    kc = torch.randn(nblocks, bs, nkv, hd, generator=g).to(BF)
    vc = torch.randn(nblocks, bs, nkv, hd, generator=g).to(BF)
    return kc, vc, bt, max_blocks


def run_attn(H, q, kc, vc, bt, max_blocks, ctx, nh, nkv, hd, bs, cu_q=None, q_per_seq=0, splits=1, flags=0, waves=1, **tree):
    T = q.shape[0]
    B = ctx.numel()
    max_q = q_per_seq if cu_q is None else int((cu_q[1:] - cu_q[:-1]).max())
    out = torch.full((T, nh * hd), float("nan"), dtype=BF, device="cuda")
    outf = torch.zeros(H.frag_numel(T, nh * hd), dtype=BF, device="cuda")
    ws_o = torch.zeros(T * nh * splits * hd, dtype=torch.float32, device="cuda")
    ws_ml = torch.zeros(T * nh * splits * 2, dtype=torch.float32, device="cuda")
    H.attn_paged(dev(q), dev(LY.kv_nhd_to_hnd(kc)), dev(LY.kv_nhd_to_hnd(vc)), dev(bt), max_blocks, dev(ctx), B, T, max_q,
                 nh, nkv, hd, bs, hd ** -0.5, cu_q=None if cu_q is None else dev(cu_q), q_per_seq=q_per_seq, splits=splits,
                 flags=flags, ws_o=ws_o, ws_ml=ws_ml, out_rows=out, out_frag=outf, waves=waves, **tree)
    torch.cuda.synchronize()
    rows = out.cpu()
    assert torch.equal(LY.frag_to_rows_ref(outf.cpu(), T, nh * hd).view(torch.int16), rows.view(torch.int16))
    return rows


# P is rounded to bf16 before P.V (as FlashAttention-3 does in the reference): |delta o| <= 2^-9 * sum_i p_i |v_i|,
# i.e. up to ~4e-3 absolute for N(0,1) values on short contexts, whatever the magnitude of o itself.
ATTN_TOL = dict(max_ulp=1, max_frac=0.05, abs_floor=2e-4)
ATTN_TOL_FA = dict(max_ulp=2, max_frac=0.25, abs_floor=4e-3)   # flags bit1: single-bf16 P


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("nh,nkv,hd", [(32, 8, 128), (32, 8, 64), (16, 8, 128), (8, 1, 128)])
def test_attn_decode_and_verify(H, nh, nkv, hd, flags):
    bs = 16
    for (B, qps, ctx_lens, splits) in [(1, 1, [37], 1), (3, 1, [1, 64, 333], 4), (1, 7, [135], 1), (2, 8, [640, 77], 5),
                                       (1, 24, [200], 3)]:
        kc, vc, bt, mb = make_paged(B, ctx_lens, nkv, hd, bs, seed=B * 100 + qps)
        torch.manual_seed(qps)
        q = torch.randn(B * qps, nh, hd).to(BF)
        ctx = torch.tensor(ctx_lens, dtype=torch.int32)
        cu = torch.arange(B + 1, dtype=torch.int32) * qps
        ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(B * qps, nh * hd)
        got = run_attn(H, q.view(B * qps, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=qps, splits=splits, flags=flags)
        assert_close_bf16(got, ref, what=f"attn B{B} q{qps} splits{splits}", **ATTN_TOL)


def test_attn_prefill_varlen(H):
    nh, nkv, hd, bs = 8, 2, 128, 16
    lens = [5, 128, 33]
    kc, vc, bt, mb = make_paged(3, lens, nkv, hd, bs, seed=9)
    torch.manual_seed(1)
    T = sum(lens)
    q = torch.randn(T, nh, hd).to(BF)
    cu = torch.tensor([0, 5, 133, 166], dtype=torch.int32)
    ctx = torch.tensor(lens, dtype=torch.int32)
    ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(T, nh * hd)
    got = run_attn(H, q.view(T, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu, splits=2)
    assert_close_bf16(got, ref, what="prefill varlen", **ATTN_TOL)
    # prefix-cache style: fewer queries than keys (bottom-right alignment)
    cu2 = torch.tensor([0, 2, 66, 70], dtype=torch.int32)
    q2 = q[:70]
    ref2 = O.attn_paged(q2, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu2).reshape(70, nh * hd)
    got2 = run_attn(H, q2.reshape(70, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu2, splits=1)
    assert_close_bf16(got2, ref2, what="prefill suffix", **ATTN_TOL)


@pytest.mark.parametrize("nh,nkv,hd,waves", [(16, 2, 128, 2), (16, 4, 64, 2), (16, 2, 128, 4)])
def test_attn_prefill_one_row_tile_per_workgroup_is_bit_identical(H, nh, nkv, hd, waves):
    """Round 6 (model._attn_flags): a prefill-sized query block (> 8 row tiles per kv head) run with ONE 16-row tile per workgroup
    (flags bit 2) instead of the kernel's default two -- the same key split per wave, so the same bits; and both against the oracle.
    Ragged: the last row tile is partial, the lengths straddle key tiles."""
    bs = 16
    lens = [131, 77]
    kc, vc, bt, mb = make_paged(2, lens, nkv, hd, bs, seed=21)
    torch.manual_seed(5)
    T = sum(lens)
    q = torch.randn(T, nh, hd).to(BF)
    cu = torch.tensor([0, 131, 208], dtype=torch.int32)
    ctx = torch.tensor(lens, dtype=torch.int32)
    ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(T, nh * hd)
    two = run_attn(H, q.view(T, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu, waves=waves)
    one = run_attn(H, q.view(T, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu, flags=4, waves=waves)
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))
    assert_close_bf16(one, ref, what="prefill, one row tile per workgroup", **ATTN_TOL)


@pytest.mark.parametrize("splits", [1, 3])
def test_attn_tree(H, splits):
    nh, nkv, hd, bs = 32, 8, 64, 16
    K, F = 7, 3
    MQ = F * (K + 1)
    for step in (0, 3, 6):
        prefix = [150, 41]
        ctx_lens = [p + K + 1 + (step + 1) * MQ for p in prefix]
        kc, vc, bt, mb = make_paged(2, ctx_lens, nkv, hd, bs, seed=step)
        torch.manual_seed(step)
        q = torch.randn(2 * MQ, nh, hd).to(BF)
        ctx = torch.tensor(ctx_lens, dtype=torch.int32)
        jidx = [i // F for i in range(MQ)]
        ref = O.attn_tree(q, kc, vc, ctx, bt, hd ** -0.5, step, K, [jidx, jidx]).reshape(2 * MQ, nh * hd)
        got = run_attn(H, q.view(2 * MQ, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=MQ, splits=splits,
                       mode=H.MODE_TREE, tree_K=K, tree_mq=MQ, tree_step=step, tree_F=F)
        assert_close_bf16(got, ref, what=f"tree step {step}", **ATTN_TOL)
    # non-uniform fan-out through the explicit branch->glue-position table
    fl = [[2, 2, 3, 1], [3, 2, 2, 1]]
    K2, MQ2 = 3, 8
    jl = [[j for j, f in enumerate(l) for _ in range(f)] for l in fl]
    ctx_lens = [30 + K2 + 1 + MQ2, 55 + K2 + 1 + MQ2]
    kc, vc, bt, mb = make_paged(2, ctx_lens, nkv, hd, bs, seed=77)
    q = torch.randn(2 * MQ2, nh, hd).to(BF)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    ref = O.attn_tree(q, kc, vc, ctx, bt, hd ** -0.5, 0, K2, jl).reshape(2 * MQ2, nh * hd)
    got = run_attn(H, q.view(2 * MQ2, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=MQ2, splits=splits, mode=H.MODE_TREE,
                   tree_K=K2, tree_mq=MQ2, tree_step=0, tree_F=1, tree_jidx=dev(torch.tensor(jl, dtype=torch.int32)))
    assert_close_bf16(got, ref, what="tree non-uniform", **ATTN_TOL)


def test_attention_under_the_reference_call_site_names_and_the_c_graph_helpers(H):
    """ssd_attn_prefill_varlen (attention.py:90-93) and ssd_attn_tree (attention.py:113-125) are ssd_attn_paged under the names of the call
    sites they replace: bit-equal to the mode they wrap.  ssd_graph_begin / _end / _launch / _destroy capture library calls into a hipGraph
    without torch's graph machinery: replays reproduce the eager bits."""
    nh, nkv, hd, bs = 8, 2, 128, 16
    lens = [5, 128, 33]
    kc, vc, bt, mb = make_paged(3, lens, nkv, hd, bs, seed=9)
    torch.manual_seed(1)
    T = sum(lens)
    q = torch.randn(T, nh, hd).to(BF)
    cu = torch.tensor([0, 5, 133, 166], dtype=torch.int32)
    ctx = torch.tensor(lens, dtype=torch.int32)
    want = run_attn(H, q.view(T, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu, splits=1)
    kd, vd = dev(LY.kv_nhd_to_hnd(kc)), dev(LY.kv_nhd_to_hnd(vc))
    out = torch.zeros(T, nh * hd, dtype=BF, device="cuda")
    H.attn_prefill_varlen(dev(q.view(T, -1)), kd, vd, dev(bt), mb, dev(ctx), dev(cu), 3, T, 128, nh, nkv, hd, bs, hd ** -0.5, out_rows=out)
    assert torch.equal(out.cpu().view(torch.int16), want.view(torch.int16))
    K, F = 3, 2
    MQ = F * (K + 1)
    ctx_lens = [40 + K + 1 + 2 * MQ, 17 + K + 1 + 2 * MQ]
    kc, vc, bt, mb = make_paged(2, ctx_lens, nkv, hd, bs, seed=5)
    q = torch.randn(2 * MQ, nh, hd).to(BF)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    want = run_attn(H, q.view(2 * MQ, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=MQ, splits=1, mode=H.MODE_TREE, tree_K=K, tree_mq=MQ,
                    tree_step=1, tree_F=F)
    kd, vd = dev(LY.kv_nhd_to_hnd(kc)), dev(LY.kv_nhd_to_hnd(vc))
    out = torch.zeros(2 * MQ, nh * hd, dtype=BF, device="cuda")
    qd, btd, ctxd = dev(q.view(2 * MQ, -1)), dev(bt), dev(ctx)
    H.attn_tree(qd, kd, vd, btd, mb, ctxd, 2, K, MQ, 1, F, nh, nkv, hd, bs, hd ** -0.5, out_rows=out)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(torch.int16), want.view(torch.int16))
    # the same call + a norm over its output, captured through the C ABI's graph helpers and replayed
    w = dev((1 + 0.1 * torch.randn(nh * hd)).to(BF))
    y = torch.zeros(2 * MQ, nh * hd, dtype=BF, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())      # (non-blocking side stream: order it behind the zero fills above)
    with torch.cuda.stream(s):
        H.attn_tree(qd, kd, vd, btd, mb, ctxd, 2, K, MQ, 1, F, nh, nkv, hd, bs, hd ** -0.5, out_rows=out)
        H.rmsnorm(out, w, 1e-6, 2 * MQ, nh * hd, out_rows=y)
        s.synchronize()
        eager = y.clone()
        with H.CGraph(s) as g:
            H.attn_tree(qd, kd, vd, btd, mb, ctxd, 2, K, MQ, 1, F, nh, nkv, hd, bs, hd ** -0.5, out_rows=out)
            H.rmsnorm(out, w, 1e-6, 2 * MQ, nh * hd, out_rows=y)
        for _ in range(3):
            y.zero_()
            out.zero_()
            g.launch()
            s.synchronize()
            assert torch.equal(y.view(torch.int16), eager.view(torch.int16))
        g.destroy()


def test_attn_softmax_spike(H):
    """A key that dominates one query row late in the scan forces the online-softmax rescale branch."""
    nh, nkv, hd, bs = 4, 1, 128, 16
    kc, vc, bt, mb = make_paged(1, [300], nkv, hd, bs, seed=4)
    q = torch.randn(1, nh, hd).to(BF)
    blk, off = int(bt[0, 250 // bs]), 250 % bs
    kc[blk, off, 0] = (q[0, 2].float() * 4).to(BF)
    ctx = torch.tensor([300], dtype=torch.int32)
    ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5).reshape(1, nh * hd)
    for splits in (1, 4):
        got = run_attn(H, q.view(1, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=1, splits=splits)
        assert_close_bf16(got, ref, what="spike", **ATTN_TOL)


# ------------------------------------------------------------------------------------------------
def test_argmax_ties_and_golden(H, golden):
    g = golden("ops_golden")
    lg = g["head_logits"]
    out = torch.zeros(7, dtype=torch.int64, device="cuda")
    H.argmax_rows(dev(lg), lg.shape[1], 7, lg.shape[1], out)
    assert out.cpu().tolist() == g["sample_tokens"].tolist()
    V = 128256
    x = torch.randn(3, V).to(BF)
    x[0, 77] = 9.0
    x[0, 100000] = 9.0          # tie -> lowest index
    x[1, V - 1] = 11.0          # last element
    x[2, 0] = 12.0
    H.argmax_rows(dev(x), V, 3, V, out)
    assert out.cpu()[:3].tolist() == [77, V - 1, 0] == O.argmax_rows(x).tolist()
    y = torch.randn(5, 151936).to(BF)
    o2 = torch.zeros(5, dtype=torch.int64, device="cuda")
    H.argmax_rows(dev(y), 151936, 5, 151936, out, o2)
    assert out.cpu()[:5].tolist() == O.argmax_rows(y).tolist() == o2.cpu().tolist()


def test_verify_greedy_golden(H, golden):
    g = golden("logic_golden")
    lp, spec = g["v_logits_p"], g["v_spec"]
    B, Kp1, V = lp.shape
    preds = torch.zeros(B * Kp1, dtype=torch.int64, device="cuda")
    ld = (V + 7) // 8 * 8            # rows must be 16-byte aligned (real vocab sizes are)
    padded = torch.full((B * Kp1, ld), float("-inf"), dtype=BF)
    padded[:, :V] = lp.view(B * Kp1, V)
    H.argmax_rows(dev(padded), ld, B * Kp1, V, preds)
    acc = torch.zeros(B, dtype=torch.int32, device="cuda")
    rec = torch.zeros(B, dtype=torch.int64, device="cuda")
    packed = torch.zeros(B, Kp1 + 2, dtype=torch.int64, device="cuda")
    H.verify_greedy(preds, dev(spec), B, Kp1 - 1, acc, rec, packed)
    assert (acc.cpu() + 1).tolist() == g["v_suffix_len"].tolist()
    assert rec.cpu().tolist() == g["v_rec"].tolist()
    pk = packed.cpu()
    assert pk[:, 0].tolist() == acc.cpu().tolist() and pk[:, 1].tolist() == rec.cpu().tolist()
    assert torch.equal(pk[:, 2:], spec)
    for b in range(B):   # suffix = [spec_0] + accepted draft tokens, exactly what verify() returns
        n = int(pk[b, 0])
        assert pk[b, 2:3 + n].tolist() == g["v_suffix"][b, :n + 1].tolist()


def test_argmax_vocab_parallel_merge(H):
    torch.manual_seed(8)
    T, V, tp = 6, 4096, 4
    x = torch.randn(T, V).to(BF)
    x[1, 100] = 7.0
    x[1, 3000] = 7.0          # tie across shards -> lowest global index
    Vs = V // tp
    vals = torch.zeros(tp, T, dtype=torch.float32, device="cuda")
    idxs = torch.zeros(tp, T, dtype=torch.int64, device="cuda")
    for r in range(tp):
        shard = dev(x[:, r * Vs:(r + 1) * Vs].contiguous())
        H.argmax_rows_val(shard, Vs, T, Vs, r * Vs, idxs[r], vals[r])
    out = torch.zeros(T, dtype=torch.int64, device="cuda")
    H.argmax_merge(vals, idxs, tp, T, T, out)
    assert out.cpu().tolist() == O.argmax_rows(x).tolist()


def test_fork_golden(H, golden):
    g = golden("logic_golden")
    lg = g["f_logits"]                      # [2, 4, 500]
    B, Kp1, V = lg.shape
    lists = [g["f_list_hit"].tolist() if h else g["f_list_miss"].tolist() for h in g["f_hits"].tolist()]
    counts = torch.tensor(lists, dtype=torch.int32)
    offsets = torch.cumsum(counts, 1).to(torch.int32) - counts
    mq = int(counts[0].sum())
    out = torch.zeros(B, mq, dtype=torch.int64, device="cuda")
    # V=500 is not a multiple of 8: pad the row stride
    ld = 504
    padded = torch.full((B * Kp1, ld), float("-inf"), dtype=BF)
    padded[:, :V] = lg.view(B * Kp1, V)
    H.fork_topf(dev(padded), ld, V, dev(g["f_returned"]), dev(counts), dev(offsets), B, Kp1 - 1, mq, out)
    assert torch.equal(out.cpu(), g["f_idx"])


def test_fork_split_golden_and_equal_to_the_row_walk(H, golden):
    """ssd_fork_topf_split (per-slice top-F candidates + one merge launch, round 5) == the reference fork golden (logic_golden.npz:
    get_forked_recovery_tokens_from_logits, async_spec_helpers.py:26-78) and == ssd_fork_topf bit for bit at the real vocabulary sizes,
    with non-uniform fan-out lists and logits quantised to a handful of values (ties everywhere: the lowest index must win in the
    slices AND in the merge, also across slice borders)."""
    from oracle import ops as O
    g = golden("logic_golden")
    lg = g["f_logits"]                      # [2, 4, 500]
    B, Kp1, V = lg.shape
    lists = [g["f_list_hit"].tolist() if h else g["f_list_miss"].tolist() for h in g["f_hits"].tolist()]
    counts = torch.tensor(lists, dtype=torch.int32)
    offsets = torch.cumsum(counts, 1).to(torch.int32) - counts
    mq = int(counts[0].sum())
    ld = 504                                # V = 500 padded to a multiple of 8 with -inf logits (never picked)
    padded = torch.full((B * Kp1, ld), float("-inf"), dtype=BF)
    padded[:, :V] = lg.view(B * Kp1, V)
    out = torch.zeros(B, mq, dtype=torch.int64, device="cuda")
    ws = torch.zeros(H.fork_topf_workspace_bytes(ld, B, Kp1 - 1) // 8, dtype=torch.int64, device="cuda")
    H.fork_topf_split(dev(padded), ld, ld, dev(g["f_returned"]), dev(counts), dev(offsets), B, Kp1 - 1, mq, ws, out)
    assert torch.equal(out.cpu(), g["f_idx"])
    for V, B, K, levels, seed in ((128256, 1, 7, 0, 0), (128256, 2, 7, 9, 1), (151936, 3, 5, 5, 2), (4096, 1, 2, 3, 3), (32, 1, 1, 0, 4)):
        torch.manual_seed(seed)
        x = torch.randn(B * (K + 1), V)
        if levels:
            x = (x * levels).round() / levels * 0.5          # a handful of distinct values: ties by the thousand
        x = x.to(BF)
        returned = torch.randint(0, V, (B, K + 1), dtype=torch.int64)
        returned[:, 1] = x[::K + 1].float().argmax(-1)       # the draft's own next token IS the row's maximum: the exclusion matters
        row0 = torch.randint(1, 6, (K + 1,), dtype=torch.int32)             # every sequence: the same fan-outs in another order (MQ_LEN is fixed)
        cnt = torch.stack([row0[torch.randperm(K + 1)] for _ in range(B)]).contiguous()
        offs = torch.cumsum(cnt, 1).to(torch.int32) - cnt
        mq = int(row0.sum())
        a = torch.full((B, mq), -1, dtype=torch.int64, device="cuda")
        b_ = torch.full((B, mq), -1, dtype=torch.int64, device="cuda")
        H.fork_topf(dev(x), V, V, dev(returned), dev(cnt), dev(offs), B, K, mq, a)
        ws = torch.zeros(H.fork_topf_workspace_bytes(V, B, K) // 8, dtype=torch.int64, device="cuda")
        H.fork_topf_split(dev(x), V, V, dev(returned), dev(cnt), dev(offs), B, K, mq, ws, b_)
        assert torch.equal(a.cpu(), b_.cpu()), (V, B, K)
        assert torch.equal(b_.cpu(), O.fork_topf(x.view(B, K + 1, V), returned, cnt.tolist())), (V, B, K)


def test_draft_advance(H):
    B, K, bs, mb = 3, 4, 16, 8
    bt = torch.arange(B * mb, dtype=torch.int32).view(B, mb)
    nxt = torch.tensor([11, 22, 33], dtype=torch.int64)
    ids = torch.zeros(B, dtype=torch.int64)
    pos = torch.tensor([15, 16, 40], dtype=torch.int64)
    slots = torch.zeros(B, dtype=torch.int32)
    ctx = (pos + 1).to(torch.int32)
    spec = torch.zeros(B, K + 1, dtype=torch.int64)
    step = torch.tensor([1], dtype=torch.int32)
    d = [dev(t) for t in (nxt, ids, pos, slots, ctx, bt, spec, step)]
    H.draft_advance(d[0], d[1], d[2], d[3], d[4], d[5], mb, bs, d[6], K, d[7], B)
    assert d[1].cpu().tolist() == [11, 22, 33]
    assert d[2].cpu().tolist() == [16, 17, 41]
    assert d[4].cpu().tolist() == [17, 18, 42]
    assert d[3].cpu().tolist() == [int(bt[0, 1]) * bs + 0, int(bt[1, 1]) * bs + 1, int(bt[2, 2]) * bs + 9]
    assert d[6].cpu()[:, 2].tolist() == [11, 22, 33]
    assert d[7].cpu().item() == 2


@pytest.mark.parametrize("waves", [2, 8])
def test_attn_multiwave_inblock_merge(H, waves):
    """Up to 8 waves of one workgroup split the key range and merge through LDS (single launch); also combined
    with grid splits + the merge kernel."""
    bs = 16
    for nh, nkv, hd in [(32, 8, 128), (32, 8, 64), (8, 1, 128)]:
        for (B, qps, ctx_lens, splits) in [(1, 1, [37], 1), (2, 7, [640, 77], 1), (1, 24, [1500], 2), (3, 1, [1, 64, 333], 3)]:
            kc, vc, bt, mb = make_paged(B, ctx_lens, nkv, hd, bs, seed=B * 10 + qps)
            torch.manual_seed(qps + waves)
            q = torch.randn(B * qps, nh, hd).to(BF)
            ctx = torch.tensor(ctx_lens, dtype=torch.int32)
            cu = torch.arange(B + 1, dtype=torch.int32) * qps
            ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(B * qps, nh * hd)
            got = run_attn(H, q.view(B * qps, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=qps, splits=splits, waves=waves)
            assert_close_bf16(got, ref, what=f"attn waves{waves} B{B} q{qps} splits{splits}", **ATTN_TOL)
    # tree mask through the multi-wave path
    nh, nkv, hd, K, F = 32, 8, 64, 7, 3
    MQ = F * (K + 1)
    ctx_lens = [150 + K + 1 + 4 * MQ]
    kc, vc, bt, mb = make_paged(1, ctx_lens, nkv, hd, bs, seed=5)
    q = torch.randn(MQ, nh, hd).to(BF)
    ctx = torch.tensor(ctx_lens, dtype=torch.int32)
    jidx = [i // F for i in range(MQ)]
    ref = O.attn_tree(q, kc, vc, ctx, bt, hd ** -0.5, 3, K, [jidx]).reshape(MQ, nh * hd)
    got = run_attn(H, q.view(MQ, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=MQ, splits=1, waves=waves,
                   mode=H.MODE_TREE, tree_K=K, tree_mq=MQ, tree_step=3, tree_F=F)
    assert_close_bf16(got, ref, what="tree multiwave", **ATTN_TOL)


# ------------------------------------------------------------------------------------------------
# split-K partial slabs (csrc/gemm_sk.hip gemm_sp_kernel) and their consumers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 2048, 2048), (1, 2048, 8192), (7, 1024, 3072), (24, 2048, 2048), (24, 2048, 8192),
                                   (16, 256, 256), (3, 80, 96), (32, 512, 1024)])
def test_gemm_parts_vs_oracle(H, M, N, K):
    """y = x . W^T as `splits` fp32 partial slabs: their in-order sum, rounded once to bf16, is the F.linear result up to
    the accumulation order; every (splits, waves) decomposition is deterministic; splits = 1 can write bf16 rows directly."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.05).to(BF)
    b = torch.randn(N).to(BF)
    w[3, :] = 0.5
    x[M - 1, : K // 2] = -1.0
    ref = O.linear(x, w)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    KT = K // 32
    tried = 0
    for splits in (1, 2, 3, 4, 8):
        for waves in (2, 4, 8, 16):
            per = -(-KT // splits)
            if KT < splits or -(-per // waves) > 8:
                continue
            tried += 1
            outs = []
            for _ in range(2):
                parts = torch.full((splits, M, N), float("nan"), dtype=torch.float32, device="cuda")
                H.gemm_parts(xf, wf, M, N, K, parts=parts, splits=splits, waves=waves)
                outs.append(parts)
            assert torch.equal(outs[0], outs[1]), "partial slabs are not deterministic"
            acc = outs[0][0].clone()
            for z in range(1, splits):
                acc += outs[0][z]
            assert_close_bf16(acc.to(BF), ref, max_ulp=1, max_frac=0.03, rel_floor=2 ** -7, what=f"gemm_parts S{splits} w{waves}")
            if splits == 1:
                y = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
                H.gemm_parts(xf, wf, M, N, K, y=y, ldy=N, splits=1, waves=waves, bias=dev(b))
                want = (outs[0][0] + dev(b).float()).to(BF)
                assert torch.equal(y.view(torch.int16), want.view(torch.int16))
    assert tried >= 3


@pytest.mark.parametrize("T,Hd,S", [(1, 2048, 2), (7, 1024, 4), (24, 2048, 2), (5, 256, 8)])
def test_rmsnorm_parts_equals_rmsnorm_of_the_rounded_sum(H, T, Hd, S):
    torch.manual_seed(T + Hd)
    parts = torch.randn(S, T + 3, Hd, device="cuda")            # slab_rows > T: extra rows are ignored
    res = torch.randn(T, Hd).to(BF).cuda()
    w = (1 + 0.1 * torch.randn(Hd)).to(BF).cuda()
    acc = parts[0].clone()
    for z in range(1, S):
        acc += parts[z]
    x = acc[:T].to(BF).contiguous()
    want = [torch.zeros(T, Hd, dtype=BF, device="cuda") for _ in range(2)] + [torch.zeros(H.frag_numel(T, Hd), dtype=BF, device="cuda")]
    got = [torch.zeros_like(t) for t in want]
    H.rmsnorm(x, w, 1e-5, T, Hd, res_in=res, res_out=want[0], out_rows=want[1], out_frag=want[2])
    H.rmsnorm_parts(parts, S, T + 3, w, 1e-5, T, Hd, res_in=res, res_out=got[0], out_rows=got[1], out_frag=got[2])
    for a, b_ in zip(got, want):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16))


@pytest.mark.parametrize("M,K,S", [(1, 2048, 2), (4, 2048, 4), (7, 4096, 2), (16, 512, 3)])
def test_fused_gemm_with_partial_slab_prologue(H, M, K, S):
    """The fused norm + GEMM kernels fed by partial slabs must equal the same kernels fed the bf16 rounding of the slab sum
    bit for bit (QKV + RoPE + KV store, and gate_up + SiLU)."""
    torch.manual_seed(M * K + S)
    nh, nkv, hd, bs, nb = 8, 2, 64, 16, 8
    N = (nh + 2 * nkv) * hd
    parts = (torch.randn(S, M, K, device="cuda") * 0.7)
    acc = parts[0].clone()
    for z in range(1, S):
        acc += parts[z]
    h = acc.to(BF).contiguous()
    res = torch.randn(M, K).to(BF).cuda()
    nw = (1 + 0.1 * torch.randn(K)).to(BF).cuda()
    w = (torch.randn(N, K) * 0.05).to(BF).cuda()
    wf = torch.zeros(w.numel(), dtype=BF, device="cuda")
    H.rows_to_frag_qkv(w, wf, nh, nkv, hd, K)
    pos = torch.randint(0, 300, (M,), dtype=torch.int64).cuda()
    slots = torch.randperm(nb * bs)[:M].to(torch.int32).cuda()
    cache = O.make_cos_sin_cache(hd, 512, 5e5).cuda()
    outs = []
    for src in (dict(h_rows=h), dict(h_parts=parts, splits=S)):
        q_out = torch.zeros(M, nh * hd, dtype=BF, device="cuda")
        kc = torch.zeros(nb, nkv, bs, hd, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        res_out = torch.zeros(M, K, dtype=BF, device="cuda")
        H.gemm_fused(wf, M, N, K, H.FEPI_QKV_ROPE, res_in=res, res_out=res_out, norm_w=nw, eps=1e-5, positions=pos, cos_sin=cache,
                     slots=slots, q_out=q_out, k_cache=kc, v_cache=vc, nh=nh, nkv=nkv, hd=hd, block_size=bs, **src)
        outs.append((q_out, kc, vc, res_out))
    for a, b_ in zip(*outs):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16))
    I = 256
    wg = (torch.randn(2 * I, K) * 0.06).to(BF).cuda()
    wgf = torch.zeros(wg.numel(), dtype=BF, device="cuda")
    H.rows_to_frag(wg, wgf, 2 * I, K, mode=1)
    outs = []
    for src in (dict(h_rows=h), dict(h_parts=parts, splits=S)):
        act = torch.zeros(H.frag_numel(M, I), dtype=BF, device="cuda")
        res_out = torch.zeros(M, K, dtype=BF, device="cuda")
        H.gemm_fused(wgf, M, 2 * I, K, H.FEPI_SILU_FRAG, res_in=res, res_out=res_out, norm_w=nw, eps=1e-5, y=act, **src)
        outs.append((act, res_out))
    for a, b_ in zip(*outs):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16))


def test_cache_lookup_equals_the_tensor_compare_of_the_reference(H):
    """ssd_cache_lookup vs the reference's vectorized membership test (draft_runner.py:215-252, restated in
    oracle/runner.py cache_lookup): first matching entry, -1 on a miss; duplicates, wrong-sequence and wrong-position keys."""
    from oracle.runner import OracleRunner
    torch.manual_seed(4)
    Bc, W = 3, 24
    forks = torch.randint(0, 50, (Bc, W), dtype=torch.int64)
    forks[1, 7] = forks[1, 3]                       # duplicate token in one row ...
    seq = torch.tensor([11, 5, 9], dtype=torch.int64)
    cj = torch.tensor([[i // 3 for i in range(W)]] * Bc, dtype=torch.int32)
    cj[1, 7] = cj[1, 3]                             # ... at the same glue position: the first entry wins
    keys = [(5, int(cj[1, 3]), int(forks[1, 3])), (11, int(cj[0, 23]), int(forks[0, 23])), (9, 0, 10 ** 9), (7, 0, int(forks[0, 0])),
            (9, int(cj[2, 5]) + 1, int(forks[2, 5])), (9, int(cj[2, 0]), int(forks[2, 0]))]
    want = OracleRunner.cache_lookup(None, keys, seq, cj, forks).tolist()
    assert want[0] == 1 * W + 3 and want[1] == 23 and want[2] == -1 and want[3] == -1 and want[5] == 2 * W
    req = dev(torch.tensor([list(k) for k in keys], dtype=torch.int64))
    out = torch.full((len(keys),), -5, dtype=torch.int32, device="cuda")
    H.cache_lookup(req, dev(seq), dev(cj), dev(forks), len(keys), Bc, W, out)
    assert out.cpu().tolist() == want


def test_attn_long_context_prefill_and_decode(H):
    """Above the contexts the other tests reach (max_model_len allows 8 K prompts): a 6000-token causal prefill through the
    page table (KV block 256, 8 waves per workgroup, two row tiles per workgroup) and, on the same cache, a 7-row verify and a
    single-token decode at context 6000 with 12 grid key-splits + the merge kernel -- against the fp32 oracle
    (reference ssd/layers/attention.py:90-93,105-111,126-131)."""
    nh, nkv, hd, bs = 4, 1, 128, 256
    L = 6000
    kc, vc, bt, mb = make_paged(1, [L], nkv, hd, bs, seed=21)
    torch.manual_seed(3)
    q = torch.randn(L, nh, hd).to(BF)
    cu = torch.tensor([0, L], dtype=torch.int32)
    ctx = torch.tensor([L], dtype=torch.int32)
    ref = O.attn_paged(q, kc, vc, ctx, bt, hd ** -0.5, cu_q=cu).reshape(L, nh * hd)
    got = run_attn(H, q.view(L, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, cu_q=cu, splits=1, flags=8 << 8)
    assert_close_bf16(got, ref, what="prefill 6000", **ATTN_TOL)
    for qps, splits in ((7, 12), (1, 12), (1, 1)):
        qd = q[:qps].contiguous()
        cud = torch.tensor([0, qps], dtype=torch.int32)
        refd = O.attn_paged(qd, kc, vc, ctx, bt, hd ** -0.5, cu_q=cud).reshape(qps, nh * hd)
        gotd = run_attn(H, qd.view(qps, -1), kc, vc, bt, mb, ctx, nh, nkv, hd, bs, q_per_seq=qps, splits=splits, flags=8 << 8)
        assert_close_bf16(gotd, refd, what=f"ctx 6000 q{qps} splits{splits}", **ATTN_TOL)


@pytest.mark.parametrize("M,N,K", [(128, 8192, 8192), (100, 8192, 7168), (40, 4096, 14336), (128, 1024, 512)])
def test_prefill_partials_consumed_by_the_norm_equal_epilogue_then_norm(H, M, N, K):
    """PF_EPI_PARTIALS (round 3): o_proj / down_proj of a single-chunk prefill leave their split-K partials in the workspace and
    the add + RMSNorm that follows sums them (ssd_rmsnorm_parts) -- bit-identical to the epilogue launch + ssd_rmsnorm it
    replaces (same slab order, same single rounding to bf16), for the default split counts incl. the unsplit case."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(BF)
    w = (torch.randn(N, K) * 0.03).to(BF)
    res = torch.randn(M, N).to(BF)
    nw = (1 + 0.1 * torch.randn(N)).to(BF)
    xf, wf = to_frag_dev(x), to_frag_dev(w)
    nbytes = H.gemm_pf_workspace_bytes(M, N, K)
    S = nbytes // (4 * M * N)
    assert S >= 1 and S * 4 * M * N == nbytes
    ws = torch.full((nbytes // 4 + 64,), float("nan"), dtype=torch.float32, device="cuda")
    # reference: GEMM with its epilogue -> rows, then add + norm
    y = torch.zeros(M, N, dtype=BF, device="cuda")
    H.gemm_pf(xf, wf, y, M, N, K, N, ws)
    r0, o0 = torch.zeros(M, N, dtype=BF, device="cuda"), torch.zeros(H.frag_numel(M, N), dtype=BF, device="cuda")
    H.rmsnorm(y, dev(nw), 1e-5, M, N, res_in=dev(res), res_out=r0, out_frag=o0)
    # partials only -> the norm sums them
    ws.fill_(float("nan"))
    H.gemm_pf(xf, wf, None, M, N, K, N, ws, epilogue=H.PF_EPI_PARTIALS)
    r1, o1 = torch.zeros(M, N, dtype=BF, device="cuda"), torch.zeros(H.frag_numel(M, N), dtype=BF, device="cuda")
    H.rmsnorm_parts(ws, S, M, dev(nw), 1e-5, M, N, res_in=dev(res), res_out=r1, out_frag=o1)
    torch.cuda.synchronize()
    assert torch.isfinite(ws[:S * M * N]).all()
    assert torch.equal(r0.view(torch.int16), r1.view(torch.int16)) and torch.equal(o0.view(torch.int16), o1.view(torch.int16))
