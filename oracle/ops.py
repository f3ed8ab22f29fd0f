"""CPU oracle: restatement of the reference's per-op arithmetic on the draft->verify hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ssd_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.  The product path is the HIP
library and fails loudly when it is missing.

Every function cites the reference file:line it restates (paths relative to /root/reference).  Parity
of this restatement is PINNED against the reference itself: ``tests/golden/make_golden.py`` imports the
reference's own modules (two stub modules for flashinfer / sgl_kernel) in the build container and dumps
golden vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays them through this file.
Where the arithmetic lives in un-vendored wheels (FlashAttention-3 in sgl-kernel 0.3.17.post1, flashinfer
0.5.2, Triton 3.4.0 store kernel) the published algorithm softmax(q k^T * scale + mask) v in fp32 is
restated and anchored on the reference's call sites (ssd/layers/attention.py:73-134).

Numerics contract ("as run"): the reference wraps its pointwise layers in @torch.compile; Inductor
computes bf16 pointwise chains in fp32 and rounds once at each store.  These functions implement that
single-rounding form (verified bit-exact against the compiled modules on CPU, see make_golden.py).
"""
from __future__ import annotations

import torch

BF16 = torch.bfloat16


# --------------------------------------------------------------------------------------------------
# pointwise / norm
# --------------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: torch.Tensor | None = None):
    """RMSDNorm / RMSHeadNorm forward -- ssd/layers/layernorm.py:16-40,64-88 (as compiled).

    Returns y (and the new residual when ``residual`` is given): x32 = x + res (fp32);
    res_out = bf16(x32); y = bf16(x32 * rsqrt(mean(x32^2) + eps) * w32).
    """
    dt = x.dtype
    x32 = x.float()
    if residual is not None:
        x32 = x32 + residual.float()
        res_out = x32.to(dt)
    var = x32.pow(2).mean(dim=-1, keepdim=True)
    y = (x32 * torch.rsqrt(var + eps)) * weight.float()
    y = y.to(dt)
    if residual is not None:
        return y, res_out
    return y


def make_cos_sin_cache(head_size: int, max_position: int, base: float) -> torch.Tensor:
    """RotaryEmbedding.__init__ -- ssd/layers/rotary_embedding.py:20-38 (fp32 [max_pos, hd] = cos || sin)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float) / head_size))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def rope(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin_cache: torch.Tensor, head_size: int):
    """RotaryEmbedding.forward + apply_rotary_emb -- ssd/layers/rotary_embedding.py:6-16,40-60 (neox halves, fp32)."""
    T = positions.shape[0]
    cos, sin = cos_sin_cache[positions].chunk(2, dim=-1)
    cos, sin = cos.unsqueeze(-2), sin.unsqueeze(-2)

    def rot(x):
        shp = x.shape
        x = x.view(T, -1, head_size)
        x1, x2 = torch.chunk(x.float(), 2, dim=-1)
        y1 = x1 * cos - x2 * sin
        y2 = x2 * cos + x1 * sin
        return torch.cat((y1, y2), dim=-1).to(x.dtype).view(shp)

    return rot(q), rot(k)


def silu_mul(x: torch.Tensor) -> torch.Tensor:
    """SiluAndMul.forward -- ssd/layers/activation.py:11-14 (as compiled: fp32 x*sigmoid(x)*y, one rounding)."""
    a, b = x.float().chunk(2, -1)
    return (a * torch.sigmoid(a) * b).to(x.dtype)


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """F.linear on bf16 operands (fp32 accumulation, one rounding) -- ssd/layers/linear.py:65,98,196.
    The same torch call the reference makes, so on one machine it is bit-identical to the reference."""
    return torch.nn.functional.linear(x, w, b)


def embedding(ids: torch.Tensor, table: torch.Tensor, vocab_start: int = 0) -> torch.Tensor:
    """VocabParallelEmbedding.forward -- ssd/layers/embed_head.py:49-57 (masked gather, pre all-reduce)."""
    n = table.shape[0]
    local = ids - vocab_start
    mask = (local >= 0) & (local < n)
    y = torch.nn.functional.embedding(torch.where(mask, local, torch.zeros_like(local)), table)
    return y * mask.unsqueeze(1).to(y.dtype)


# --------------------------------------------------------------------------------------------------
# KV cache + attention (reference cache layout: [num_blocks, block_size, n_kv, hd], "NHD")
# --------------------------------------------------------------------------------------------------
def store_kv(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, slot_mapping: torch.Tensor):
    """store_kvcache (Triton) -- ssd/layers/attention.py:10-41: row i -> cache[slot[i]] unless slot == -1."""
    D = k.shape[1] * k.shape[2]
    ok = slot_mapping >= 0
    idx = slot_mapping[ok].long()
    k_cache.view(-1, D)[idx] = k.reshape(-1, D)[ok]
    v_cache.view(-1, D)[idx] = v.reshape(-1, D)[ok]


def _sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale masked) v in fp32; q [Lq, nh, hd], k/v [Lk, nkv, hd], mask bool [Lq, Lk]."""
    nh, nkv = q.shape[1], k.shape[1]
    g = nh // nkv
    qf = q.float().transpose(0, 1)                                  # [nh, Lq, hd]
    kf = k.float().transpose(0, 1).repeat_interleave(g, dim=0)      # [nh, Lk, hd]
    vf = v.float().transpose(0, 1).repeat_interleave(g, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale
    s = s.masked_fill(~mask.unsqueeze(0), float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)  # fully masked rows (never produced by valid inputs)
    o = torch.matmul(p, vf)
    return o.transpose(0, 1).to(q.dtype)                            # [Lq, nh, hd]


def gather_paged(cache: torch.Tensor, block_table: torch.Tensor, length: int) -> torch.Tensor:
    """Keys/values 0..length-1 of one sequence from the paged cache ([blocks, bs, nkv, hd])."""
    bs = cache.shape[1]
    nblk = (length + bs - 1) // bs
    pages = cache[block_table[:nblk].long()]
    return pages.reshape(-1, cache.shape[2], cache.shape[3])[:length]


def attn_prefill_varlen(q, k, v, cu_q, cu_k, scale):
    """flash_attn_varlen_func(causal=True) -- ssd/layers/attention.py:90-93; bottom-right aligned when Lq < Lk."""
    outs = []
    for b in range(cu_q.numel() - 1):
        qs = q[cu_q[b]:cu_q[b + 1]]
        ks, vs = k[cu_k[b]:cu_k[b + 1]], v[cu_k[b]:cu_k[b + 1]]
        Lq, Lk = qs.shape[0], ks.shape[0]
        mask = torch.ones(Lq, Lk, dtype=torch.bool).tril(diagonal=Lk - Lq)
        outs.append(_sdpa(qs, ks, vs, mask, scale))
    return torch.cat(outs, dim=0)


def attn_paged(q, k_cache, v_cache, context_lens, block_tables, scale, cu_q=None):
    """flash_attn_with_kvcache(causal=True) -- ssd/layers/attention.py:105-111 (verify / glue: cu_seqlens_q
    queries per sequence, bottom-right causal over context_lens) and :126-131 (one query per sequence)."""
    B = context_lens.numel()
    outs = []
    for b in range(B):
        if cu_q is None:
            qs = q[b:b + 1]
        else:
            qs = q[cu_q[b]:cu_q[b + 1]]
        L = int(context_lens[b])
        ks = gather_paged(k_cache, block_tables[b], L)
        vs = gather_paged(v_cache, block_tables[b], L)
        Lq = qs.shape[0]
        mask = torch.ones(Lq, L, dtype=torch.bool).tril(diagonal=L - Lq)
        outs.append(_sdpa(qs, ks, vs, mask, scale))
    return torch.cat(outs, dim=0)


def tree_mask(context_len: int, step: int, K: int, jidx: list[int]) -> torch.Tensor:
    """Tree-decode custom mask -- ssd/engine/helpers/mask_helpers.py:12-21,56-90.

    Row i (branch i, forked at glue position jidx[i]) sees: the trunk prefix, glue columns 0..jidx[i],
    and its own column in each of the step+1 diagonal blocks.  [MQ_LEN, context_len] bool.
    """
    mq = len(jidx)
    prefix = context_len - (K + 1) - (step + 1) * mq
    assert prefix >= 0
    m = torch.zeros(mq, context_len, dtype=torch.bool)
    m[:, :prefix] = True
    for i, j in enumerate(jidx):
        m[i, prefix:prefix + j + 1] = True
        for d in range(step + 1):
            m[i, prefix + K + 1 + d * mq + i] = True
    return m


def attn_tree(q, k_cache, v_cache, context_lens, block_tables, scale, step: int, K: int, jidx_per_seq: list[list[int]]):
    """flashinfer BatchPrefillWithPagedKVCacheWrapper.run with the custom mask -- ssd/layers/attention.py:113-125."""
    B = context_lens.numel()
    mq = len(jidx_per_seq[0])
    outs = []
    for b in range(B):
        L = int(context_lens[b])
        ks = gather_paged(k_cache, block_tables[b], L)
        vs = gather_paged(v_cache, block_tables[b], L)
        mask = tree_mask(L, step, K, jidx_per_seq[b])
        outs.append(_sdpa(q[b * mq:(b + 1) * mq], ks, vs, mask, scale))
    return torch.cat(outs, dim=0)


# --------------------------------------------------------------------------------------------------
# sampling / verification / fork (integer results: bit-exact contract)
# --------------------------------------------------------------------------------------------------
def argmax_rows(logits: torch.Tensor) -> torch.Tensor:
    """Sampler.forward at temperature 0 -- ssd/layers/sampler.py:18-19; verify.py:34 (lowest index on ties)."""
    return logits.to(torch.float).argmax(dim=-1)


def verify_greedy(preds_p: torch.Tensor, speculations: torch.Tensor):
    """Greedy branch of verify() -- ssd/utils/verify.py:28-48.  preds_p [B,K+1] = argmax(logits_p);
    speculations [B,K+1] = (recovery, x_1..x_K).  Returns (accept_len [B], recovery [B])."""
    K = speculations.shape[1] - 1
    draft = speculations[:, 1:]
    matches = draft == preds_p[:, :-1]
    any_mismatch = (~matches).any(dim=1)
    first_mismatch = (~matches).int().argmax(dim=1)
    accept = torch.where(any_mismatch, first_mismatch, torch.full_like(first_mismatch, K))
    rec = preds_p[torch.arange(preds_p.shape[0]), accept]
    return accept, rec


def verify_suffixes(logits_p: torch.Tensor, speculations: torch.Tensor):
    """verify() end to end at temperature 0 -- ssd/utils/verify.py:5-48,169-181: (new_suffixes, recovery_tokens)."""
    preds = logits_p.argmax(dim=-1)
    accept, rec = verify_greedy(preds, speculations)
    starts = speculations[:, 0].tolist()
    out = []
    for b, n in enumerate(accept.tolist()):
        out.append([starts[b]] + speculations[b, 1:1 + n].tolist())
    return out, rec.tolist()


def fork_topf(logits: torch.Tensor, returned_tokens: torch.Tensor, fan_out_lists: list[list[int]]) -> torch.Tensor:
    """get_forked_recovery_tokens_from_logits -- ssd/utils/async_helpers/async_spec_helpers.py:26-78.

    logits [B,K+1,V]; returned_tokens [B,K+1] = (rec, x_1..x_K); fan_out_lists[b] = per-position fan-out
    (the hit or the miss list, chosen per sequence).  Returns [B, MQ_LEN] int64.
    """
    B, Kp1, V = logits.shape
    lg = logits.clone()
    lg[:, :-1, :] = lg[:, :-1, :].scatter(2, returned_tokens[:, 1:].unsqueeze(2), float("-inf"))
    kmax = max(max(f) for f in fan_out_lists)
    # stable descending order (lowest index first among equal logits)
    order = torch.argsort(lg.float(), dim=-1, descending=True, stable=True)[..., :kmax]
    rows = []
    for b in range(B):
        toks = []
        for j in range(Kp1):
            toks.extend(order[b, j, :fan_out_lists[b][j]].tolist())
        rows.append(toks)
    return torch.tensor(rows, dtype=torch.int64)


# --------------------------------------------------------------------------------------------------
# temperature > 0 (stochastic) -- same torch RNG call sequence as the reference, so with the same manual seed the
# CPU results are identical to the reference's (pinned in tests/test_oracle_golden.py::test_verify_stochastic)
# --------------------------------------------------------------------------------------------------
def sampler_x_rescale(probs: torch.Tensor, sampler_x: float, F: int) -> torch.Tensor:
    """apply_sampler_x_rescaling -- ssd/utils/async_helpers/async_spec_helpers.py:79-105: the F+1 largest probabilities
    are multiplied by sampler_x, then the row is renormalised."""
    _, top = torch.topk(probs, F + 1, dim=-1)
    mask = torch.zeros_like(probs, dtype=torch.bool)
    mask.scatter_(dim=-1, index=top, value=True)
    probs = torch.where(mask, probs * sampler_x, probs)
    return probs / probs.sum(dim=-1, keepdim=True)


def sample(logits: torch.Tensor, temperatures: torch.Tensor, sampler_x: float | None = None, F: int | None = None) -> torch.Tensor:
    """Sampler.forward -- ssd/layers/sampler.py:15-36: greedy where T == 0, else argmax(softmax(logits / T) / Exp(1));
    with sampler_x (tree decode only, is_tree=True) the distribution is rescaled first."""
    lg = logits.to(torch.float)
    greedy = lg.argmax(dim=-1)
    lg = lg / temperatures.unsqueeze(1)
    probs = torch.softmax(lg, dim=-1, dtype=torch.float)
    if sampler_x is not None:
        probs = sampler_x_rescale(probs, sampler_x, F)
    scores = probs.div_(torch.empty_like(probs).exponential_(1) + 1e-10)
    return torch.where(temperatures == 0, greedy, scores.argmax(dim=-1))


def verify_full(logits_p, logits_q, speculations, temps_t, temps_q, cache_hits=None, jit_speculate=False, sampler_x=None,
                async_fan_out=None):
    """verify() -- ssd/utils/verify.py:5-181.  Returns (suffixes, recovery, accept_probs or None)."""
    B, Kp1, V = logits_p.shape
    K = Kp1 - 1
    draft = speculations[:, 1:]
    preds_p = logits_p.argmax(dim=-1)
    matches = draft == preds_p[:, :-1]
    any_mis = (~matches).any(dim=1)
    first_mis = (~matches).int().argmax(dim=1)
    accept_greedy = torch.where(any_mis, first_mis, torch.full_like(first_mis, K))
    bidx = torch.arange(B)
    rec_greedy = preds_p[bidx, accept_greedy]
    base = (temps_t > 0) | (temps_q > 0)
    if jit_speculate:
        ratio_rows = base
    else:
        ratio_rows = base & (cache_hits.to(torch.bool) if cache_hits is not None else torch.zeros_like(base))
    do_ratio = bool(ratio_rows.any())
    need_p = bool((temps_t > 0).any()) or do_ratio
    probs_p = None
    if need_p:
        probs_p = torch.zeros(B, Kp1, V, dtype=torch.float32)
        nz = temps_t > 0
        if nz.any():
            t = temps_t[nz].unsqueeze(1).unsqueeze(2).clamp(min=1e-8)
            probs_p[nz] = torch.softmax((logits_p[nz] / t).to(torch.float32), dim=-1)
        if (~nz).any():
            oh = torch.zeros_like(logits_p[~nz], dtype=torch.float32)
            oh.scatter_(2, logits_p[~nz].argmax(dim=-1).unsqueeze(-1), 1.0)
            probs_p[~nz] = oh
    accept_probs = None
    if do_ratio:
        probs_q = torch.zeros(B, K, V, dtype=torch.float32)
        nzq = temps_q > 0
        if nzq.any():
            tq = temps_q[nzq].unsqueeze(1).unsqueeze(2).clamp(min=1e-8)
            probs_q[nzq] = torch.softmax((logits_q[nzq] / tq).to(torch.float32), dim=-1)
        if (~nzq).any():
            ohq = torch.zeros_like(logits_q[~nzq], dtype=torch.float32)
            ohq.scatter_(2, logits_q[~nzq].argmax(dim=-1).unsqueeze(-1), 1.0)
            probs_q[~nzq] = ohq
        if sampler_x is not None:           # verify.py:101-105
            probs_q = sampler_x_rescale(probs_q, sampler_x, async_fan_out)
        gi = draft.unsqueeze(2)
        p_vals = probs_p[:, :K, :].gather(2, gi).squeeze(2)
        q_vals = probs_q.gather(2, gi).squeeze(2)
        accept_probs = (p_vals / (q_vals + 1e-10)).clamp(max=1.0)
        rand = torch.rand_like(accept_probs)
        accepts = rand <= accept_probs
        rej_any = (~accepts).any(dim=1)
        first_rej = (~accepts).int().argmax(dim=1)
        accept_ratio = torch.where(rej_any, first_rej, torch.full_like(first_rej, K))
        accept_until = torch.where(ratio_rows, accept_ratio, accept_greedy)
    else:
        accept_until = accept_greedy
    if probs_p is None:
        rec_ratio = rec_greedy
    else:
        p_fb = probs_p[bidx, accept_until]
        fb = p_fb / p_fb.sum(dim=1, keepdim=True)
        if do_ratio:
            q_slice = probs_q[bidx, accept_until.clamp(max=K - 1)]
            mask_adjust = (temps_t > 0) & (accept_until < K) & ratio_rows
            adj = (p_fb - q_slice).clamp(min=0.0)
            sums = adj.sum(dim=1, keepdim=True)
            adj_norm = torch.where(sums > 0, adj / sums, fb)
            r1 = torch.multinomial(adj_norm, 1).squeeze(1)
            r2 = torch.multinomial(fb, 1).squeeze(1)
            rec_ratio = torch.where(mask_adjust, r1, r2)
        else:
            rec_ratio = torch.multinomial(fb, 1).squeeze(1)
    rec = torch.where(temps_t > 0, rec_ratio, rec_greedy)
    starts = speculations[:, 0].tolist()
    sfx = [[starts[b]] + draft[b, :n].tolist() for b, n in enumerate(accept_until.tolist())]
    return sfx, rec.tolist(), accept_probs
