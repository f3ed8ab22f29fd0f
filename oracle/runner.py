"""CPU stand-in for ssd_amd.engine.model_runner.ModelRunner built on the oracle model (TEST INFRASTRUCTURE ONLY).

Same public surface (run / speculate_chain / verify_chain / num_kvcache_blocks) so the REAL engine code --
scheduler, block managers, SpecDecodeStep, speculators, verifier -- can be exercised on CPU in the ``not gpu``
tests, under ``gloo`` for the multi-rank paths, and timed as the CPU baseline in bench.py.  Input preparation
restates ssd/engine/helpers/runner_helpers.py:50-180; sampling is Sampler at temperature 0 and verification is
the greedy branch of ssd/utils/verify.py.  It is injected explicitly (``LLMEngine(..., runner_factory=...)``);
the product never selects it.
"""
from __future__ import annotations

import torch

from oracle import ops as O
from oracle.model import OracleModel, Ctx, shard_weights
from ssd_amd import weights as W


class OracleRunner:
    def __init__(self, config, model_cfg, *, is_draft: bool, topo=None, weights: dict | None = None,
                 num_kvcache_blocks: int = -1, **_):
        self.config, self.cfg, self.is_draft = config, model_cfg, is_draft
        self.block_size = config.kvcache_block_size
        self.K = config.speculate_k if config.speculate else 0
        tp_rank = topo.tp_rank if topo is not None else 0
        tp_size = topo.tp_size if topo is not None else 1
        tp_group = topo.tp_group if topo is not None else None
        if weights is None:
            seed = config.draft_weights_seed if is_draft else config.weights_seed
            weights = W.synthetic_state_dict(model_cfg, seed, config.weights_std, recipe=getattr(config, "weights_recipe", None))
        self.num_kvcache_blocks = num_kvcache_blocks if num_kvcache_blocks > 0 else 64
        self.eagle = model_cfg.family == "eagle3"
        # EAGLE-3 target: tap the residual stream entering these layers on every forward (llama3.py:256-271)
        self.taps = sorted(set(config.eagle_layers)) if (getattr(config, "use_eagle", False) and not is_draft) else None
        self._acts = None
        if self.eagle:
            from oracle.eagle import OracleEagleDraft
            self.model = OracleEagleDraft(model_cfg, weights, self.num_kvcache_blocks, self.block_size)
        else:
            self.model = OracleModel(model_cfg, shard_weights(model_cfg, weights, tp_rank, tp_size), self.num_kvcache_blocks,
                                     self.block_size, tp_rank, tp_size, tp_group)
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        # (seq_id, absolute position of the decided token) -> top-2 logit margin of that greedy decision; tests use it
        # to tell a legitimate near-tie flip from a real divergence of the HIP engine
        self.margin_log: dict[tuple[int, int], float] = {}
        # draft-side greedy decisions in the order they were taken: (kind, gaps) with kind in "chain" (synchronous K-step chain,
        # [K, B]), "jit" ([K, B]), "glue" (fork rows, [B, K+1]: smallest gap between consecutive logits among the F + 1 largest
        # that the fork may pick from) and "tree" ([K, B * MQ]); gaps in logit units.  tests/lockstep.py uses them to tell a
        # draft near-tie that flipped on the GPU from a real divergence of the speculation
        self.decision_gaps: list[tuple[str, torch.Tensor]] = []
        self.log_decisions = False      # switched on by the lock-step tests only (the CPU baseline leg of bench.py times this runner)

    @staticmethod
    def _gap(lg, top: int = 2):
        v = lg.float().topk(top, dim=-1).values
        return (v[..., :-1] - v[..., 1:]).min(dim=-1).values

    def _log_margins(self, lg, keys):
        top = lg.float().topk(2, dim=-1).values
        for (sid, pos), m in zip(keys, (top[:, 0] - top[:, 1]).tolist()):
            self.margin_log[(sid, pos)] = m

    # ---- helpers ----
    def _table(self, s):
        return s.draft_block_table if self.is_draft else s.block_table

    def _slot(self, table, p):
        return table[p // self.block_size] * self.block_size + p % self.block_size

    def _bt(self, seqs):
        n = max(len(self._table(s)) for s in seqs)
        return torch.tensor([self._table(s) + [-1] * (n - len(self._table(s))) for s in seqs], dtype=torch.int32)

    def _logits(self, hidden):
        """Full-vocab logits on every rank (the engine is SPMD)."""
        lg = O.linear(hidden, self.model.w["model.embed_tokens.weight" if self.cfg.tie_word_embeddings else "lm_head.weight"])
        if self.tp_size > 1:
            import torch.distributed as dist
            parts = [torch.empty_like(lg) for _ in range(self.tp_size)]
            dist.all_gather(parts, lg.contiguous(), group=self.tp_group)
            lg = torch.cat(parts, dim=-1)
        return lg

    def call(self, method, *args):
        return getattr(self, method)(*args)

    def _forward(self, ids, pos, ctx):
        """Target forward; under EAGLE-3 also keeps the tapped activations of this call for `eagle_acts`."""
        if self.taps is None:
            return self.model.forward(ids, pos, ctx)
        h, self._acts = self.model.forward(ids, pos, ctx, taps=self.taps)
        return h

    def eagle_acts(self, n: int) -> torch.Tensor:
        """[n, taps * h] activations of the last prefill / verify forward (reference model_runner.py:613-616)."""
        assert self._acts is not None and self._acts.shape[0] == n
        return self._acts

    def _temps(self, seqs):
        return torch.tensor([float(s.draft_temperature) if (self.is_draft and s.draft_temperature is not None) else float(s.temperature)
                             for s in seqs])

    def _pick(self, lg, seqs):
        """Sampler.forward (sampler.py:15-36): greedy rows at temperature 0, exponential-noise draw otherwise."""
        t = self._temps(seqs)
        return (O.sample(lg, t) if bool((t > 0).any()) else O.argmax_rows(lg)).tolist()

    @torch.inference_mode()
    def run(self, seqs, is_prefill: bool, last_only: bool = True, draft_return_logits: bool = False):
        if is_prefill:
            ids, pos, slots, cu_q, cu_k = [], [], [], [0], [0]
            for s in seqs:
                cached = s.num_draft_cached_tokens if self.is_draft else s.num_cached_tokens
                n = len(s)
                ids.extend(s[cached:])
                pos.extend(range(cached, n))
                slots.extend(self._slot(self._table(s), p) for p in range(cached, n))
                cu_q.append(cu_q[-1] + n - cached)
                cu_k.append(cu_k[-1] + n)
            cu_q_t, cu_k_t = torch.tensor(cu_q, dtype=torch.int32), torch.tensor(cu_k, dtype=torch.int32)
            paged = cu_k[-1] > cu_q[-1]
            ctx = Ctx("prefill", slot_mapping=torch.tensor(slots, dtype=torch.int32), cu_q=cu_q_t, cu_k=cu_k_t,
                      block_tables=self._bt(seqs) if paged else None)
            h = self._forward(torch.tensor(ids), torch.tensor(pos), ctx)
            lg = self._logits(h[(cu_q_t[1:] - 1).long()])
            self._log_margins(lg, [(s.seq_id, len(s)) for s in seqs])
            toks = self._pick(lg, seqs)
            return (toks, lg) if draft_return_logits else toks
        if not last_only:
            lg = self._verify_logits(seqs, None)
            return lg
        ids, pos, slots, ctx_lens = [], [], [], []
        for s in seqs:
            ids.append(s.last_token)
            pos.append(len(s) - 1)
            slots.append(self._slot(self._table(s), len(s) - 1))
            ctx_lens.append(len(s))
        lg = self._decode(ids, pos, slots, ctx_lens, self._bt(seqs))
        self._log_margins(lg, [(s.seq_id, len(s)) for s in seqs])
        toks = self._pick(lg, seqs)
        return (toks, lg) if draft_return_logits else toks

    def _decode(self, ids, pos, slots, ctx_lens, bt):
        ctx = Ctx("decode", slot_mapping=torch.tensor(slots, dtype=torch.int32),
                  context_lens=torch.tensor(ctx_lens, dtype=torch.int32), block_tables=bt)
        return self._logits(self.model.forward(torch.tensor(ids), torch.tensor(pos), ctx))

    def _verify_logits(self, seqs, speculations):
        K = self.K
        ids, pos, slots, ctx_lens = [], [], [], []
        for b, s in enumerate(seqs):
            pos0 = s.num_tokens - (K + 1)
            ids.extend(s[pos0:] if speculations is None else speculations[b].tolist())
            pos.extend(range(pos0, pos0 + K + 1))
            slots.extend(self._slot(self._table(s), p) for p in range(pos0, pos0 + K + 1))
            ctx_lens.append(len(s))
        B = len(seqs)
        ctx = Ctx("verify", slot_mapping=torch.tensor(slots, dtype=torch.int32),
                  context_lens=torch.tensor(ctx_lens, dtype=torch.int32), block_tables=self._bt(seqs),
                  cu_q=torch.arange(B + 1, dtype=torch.int32) * (K + 1))
        return self._logits(self._forward(torch.tensor(ids), torch.tensor(pos), ctx))

    @torch.inference_mode()
    def speculate_chain(self, seqs, recovery_tokens):
        """K sequential draft decodes from the recovery token (speculator_sync.py:47-66); the (K+1)-th, KV-deposit-only
        decode is deferred to deposit_pending, exactly like ssd_amd.engine.model_runner.ModelRunner."""
        B, K = len(seqs), self.K
        spec = torch.zeros(B, K + 1, dtype=torch.int64)
        spec[:, 0] = torch.tensor(recovery_tokens)
        cur = list(recovery_tokens)
        bt = self._bt(seqs)
        lq = []
        for k in range(K):
            pos = [len(s) - 1 + k for s in seqs]
            slots = [self._slot(self._table(s), p) for s, p in zip(seqs, pos)]
            lg = self._decode(cur, pos, slots, [p + 1 for p in pos], bt)
            lq.append(lg)
            cur = self._pick(lg, seqs)
            spec[:, k + 1] = torch.tensor(cur)
        self._lq = torch.stack(lq, dim=1)
        if self.log_decisions:
            self.decision_gaps.append(("chain", torch.stack([self._gap(l) for l in lq])))
        if bool((self._temps(seqs) > 0).any()):
            # the reference's (K+1)-th draft forward (speculator_sync.py:47-56; deferred to deposit_pending here) goes through
            # run() and its Sampler too: one more exponential draw per logit, discarded.  The oracle consumes it at the same
            # point of the random stream, so that a seeded reference run can be replayed token for token.
            torch.empty(B, lg.shape[-1], dtype=torch.float32).exponential_(1)
        return spec

    @torch.inference_mode()
    def deposit_pending(self, seqs):
        pos = [len(s) - 2 for s in seqs]
        slots = [self._slot(self._table(s), p) for s, p in zip(seqs, pos)]
        self._decode([s[p] for s, p in zip(seqs, pos)], pos, slots, [p + 1 for p in pos], self._bt(seqs))

    def logits_q(self, B):
        return self._lq[:B]

    @torch.inference_mode()
    def verify_chain(self, seqs, speculations, logits_q=None, temps_q=None, ratio_rows=None):
        B, K = len(seqs), self.K
        lg = self._verify_logits(seqs, speculations).view(B, K + 1, -1)
        # row j decides the token at position pos0 + j + 1 (rows past the first rejection are re-decided, and re-logged, later)
        self._log_margins(lg.reshape(B * (K + 1), -1), [(s.seq_id, s.num_tokens - (K + 1) + j + 1) for s in seqs for j in range(K + 1)])
        if temps_q is None:
            return O.verify_suffixes(lg, speculations)
        # verify() with explicit ratio rows == jit_speculate=True on the rows flagged, greedy fallback elsewhere
        sfx, rec, _ = O.verify_full(lg, logits_q if logits_q is not None else torch.zeros(B, K, lg.shape[-1], dtype=lg.dtype),
                                    speculations, self._temps(seqs), torch.tensor(temps_q), cache_hits=torch.tensor(ratio_rows),
                                    jit_speculate=False, sampler_x=self.config.sampler_x, async_fan_out=self.config.async_fan_out)
        return sfx, rec

    # ---- draft-server operations of asynchronous speculation (explicit arrays, no Sequence objects) ----
    def cache_index(self, seq_ids, jlists):
        return torch.tensor(list(seq_ids), dtype=torch.int64), torch.tensor([list(j) for j in jlists], dtype=torch.int32)

    def cache_lookup(self, keys, cache_seq, cache_j, forks):
        """The reference's vectorized membership test (ssd/engine/draft_runner.py:215-252): first matching entry or -1."""
        req = torch.tensor([list(k) for k in keys], dtype=torch.int64)                      # [B, 3]
        W = forks.shape[1]
        ck = torch.stack([cache_seq.repeat_interleave(W), cache_j.reshape(-1).to(torch.int64), forks.reshape(-1)], dim=1)   # [Bc*W, 3]
        match = (req.unsqueeze(1) == ck.unsqueeze(0)).all(dim=2)                            # [B, Bc*W]
        first = match.float().argmax(dim=1).to(torch.int32)
        return torch.where(match.any(dim=1), first, torch.full_like(first, -1))

    def zeros_tokens(self, B, K):
        return torch.zeros(B, K, dtype=torch.int64)

    def _bt_from(self, tables):
        return torch.tensor(tables, dtype=torch.int32)

    @torch.inference_mode()
    def draft_prefill(self, token_lists, tables, acts=None):
        """draft_async_prefill (draft_runner.py:51-101): trunk KV for the whole prompt.  EAGLE-3: `acts` holds one target
        activation row per token (the caller already applied the one-token shift)."""
        ids, pos, slots, cu = [], [], [], [0]
        for toks, tb in zip(token_lists, tables):
            n = len(toks)
            ids.extend(toks)
            pos.extend(range(n))
            slots.extend(self._slot(tb, p) for p in range(n))
            cu.append(cu[-1] + n)
        cu_t = torch.tensor(cu, dtype=torch.int32)
        ctx = Ctx("prefill", slot_mapping=torch.tensor(slots, dtype=torch.int32), cu_q=cu_t, cu_k=cu_t)
        if self.eagle:
            self.model.forward(torch.tensor(ids), torch.tensor(pos), acts, ctx)
        else:
            self.model.forward(torch.tensor(ids), torch.tensor(pos), ctx)

    def _eagle_logits(self, ids, pos, cond, ctx):
        """(full-vocabulary logits, prenorm) of one EAGLE draft forward (reference model_runner.py:606-612)."""
        pre = self.model.forward(ids, pos, cond, ctx)
        return self.model.compute_logits(pre), pre

    @torch.inference_mode()
    def draft_jit(self, rec, num_tokens, tables, temps=None, cond=None):
        """jit_speculate (draft_runner.py:124-184): K single-token decodes from the recovery token at P = n - 1.  EAGLE-3:
        the first step is conditioned on fc(cond) (the recovery token's target activation), every later one on the
        previous step's prenorm; the K prenorms are kept for `jit_acts`."""
        B, K = len(rec), self.K
        bt = self._bt_from(tables)
        out = torch.zeros(B, K, dtype=torch.int64)
        cur = list(rec)
        t = None if temps is None or not any(x > 0 for x in temps) else torch.tensor(temps, dtype=torch.float32)
        if t is not None:
            self._mirror_response_init(B)
        lq, pres = [], []
        for i in range(K):
            pos = [n - 1 + i for n in num_tokens]
            slots = [self._slot(tb, p) for tb, p in zip(tables, pos)]
            if self.eagle:
                ctx = Ctx("decode", slot_mapping=torch.tensor(slots, dtype=torch.int32),
                          context_lens=torch.tensor([p + 1 for p in pos], dtype=torch.int32), block_tables=bt)
                lg, cond = self._eagle_logits(torch.tensor(cur), torch.tensor(pos), cond, ctx)
                pres.append(cond)
            else:
                lg = self._decode(cur, pos, slots, [p + 1 for p in pos], bt)
            lq.append(lg)
            # the JIT chain samples with is_tree=True (draft_runner.py:172): sampler_x rescales it like the tree steps
            cur = (O.argmax_rows(lg) if t is None else O.sample(lg, t, self.config.sampler_x, self.config.async_fan_out)).tolist()
            out[:, i] = torch.tensor(cur)
        self._lq = torch.stack(lq, dim=1)
        self._jit_acts = torch.stack(pres, dim=1) if pres else None
        if self.log_decisions:
            self.decision_gaps.append(("jit", torch.stack([self._gap(l) for l in lq])))
        return out

    def jit_acts(self, B):
        return self._jit_acts[:B]

    def tree_acts(self, T):
        return self._tree_acts[:T]

    def _eagle_glue_fork(self, glue_ids, num_tokens, tables, fan_lists, eagle):
        """_build_tree_batch, EAGLE branch (draft_runner.py:538-612,660-700): per sequence the packed rows
        [extend_0..extend_{n-1} | recovery | spec_1..spec_K] at positions P-n .. P+K (P = num_tokens - 1, already shifted);
        extend + recovery rows are conditioned on fc(target activation), spec rows on the previous round's prenorms."""
        B, K = glue_ids.shape[0], self.K
        acts, counts, ext_ids, prev = eagle["acts"], eagle["ext_counts"], eagle["ext_ids"], eagle["prev_acts"]
        ids, pos, slots, cu, conds = [], [], [], [0], []
        for b, (n, tb) in enumerate(zip(num_tokens, tables)):
            ne = counts[b]
            tc = self.model.project(torch.cat([acts[b, :ne], acts[b, K:K + 1]], dim=0))         # one fc call for extend + recovery
            conds.extend([tc, prev[b]])
            ids.extend(list(ext_ids[b][:ne]) + glue_ids[b].tolist())
            for p in range(n - 1 - ne, n + K):
                pos.append(p)
                slots.append(self._slot(tb, p))
            cu.append(cu[-1] + ne + K + 1)
        ctx = Ctx("verify", slot_mapping=torch.tensor(slots, dtype=torch.int32),
                  context_lens=torch.tensor([n + K for n in num_tokens], dtype=torch.int32), block_tables=self._bt_from(tables),
                  cu_q=torch.tensor(cu, dtype=torch.int32))
        lg, pre = self._eagle_logits(torch.tensor(ids), torch.tensor(pos), torch.cat(conds, dim=0), ctx)
        rows = torch.tensor([cu[b] + counts[b] + j for b in range(B) for j in range(K + 1)])    # the K+1 [recovery | spec] rows
        self._glue_pre = pre[rows].view(B, K + 1, -1)
        glg = lg[rows].view(B, K + 1, -1)
        if self.log_decisions:
            ex = glg.clone()        # the fork never picks the token that follows in the chain (async_spec_helpers.py:45-52)
            ex[:, :-1, :] = ex[:, :-1, :].scatter(2, glue_ids[:, 1:].unsqueeze(2), float("-inf"))
            self.decision_gaps.append(("glue", self._gap(ex, max(max(f) for f in fan_lists) + 1)))
        return O.fork_topf(glg, glue_ids, fan_lists)

    def _mirror_response_init(self, B):
        """Random-stream bookkeeping for seeded replays of a reference run at temperature > 0 (tests/test_ref_engine_golden.py):
        the reference's hit_cache_and_respond starts every request by filling its reply buffer with uniform noise
        (draft_runner.py:192-193) -- B*K*V bf16 draws from the global generator, before the JIT chain on a miss and before
        the tree sampling on a hit.  The oracle consumes the same draws at the same point: at the top of draft_jit, or --
        when the round had no JIT chain -- at the top of the glue."""
        torch.empty((B, self.K, self.cfg.vocab_size), dtype=torch.bfloat16).uniform_()
        self._response_init_done = True

    @torch.inference_mode()
    def draft_glue_fork(self, glue_ids, num_tokens, tables, fan_lists, eagle=None):
        """Glue decode + fork (draft_runner.py:620-700; async_spec_helpers.py:26-78)."""
        if getattr(self, "_sampling_rounds", False) and not getattr(self, "_response_init_done", False):
            self._mirror_response_init(glue_ids.shape[0])
        self._response_init_done = False
        if eagle is not None:
            return self._eagle_glue_fork(glue_ids, num_tokens, tables, fan_lists, eagle)
        B, K = glue_ids.shape[0], self.K
        pos, slots = [], []
        for n, tb in zip(num_tokens, tables):
            for p in range(n - 1, n + K):
                pos.append(p)
                slots.append(self._slot(tb, p))
        ctx = Ctx("verify", slot_mapping=torch.tensor(slots, dtype=torch.int32),
                  context_lens=torch.tensor([n + K for n in num_tokens], dtype=torch.int32), block_tables=self._bt_from(tables),
                  cu_q=torch.arange(B + 1, dtype=torch.int32) * (K + 1))
        lg = self._logits(self.model.forward(glue_ids.reshape(-1), torch.tensor(pos), ctx)).view(B, K + 1, -1)
        if self.log_decisions:
            ex = lg.clone()         # the fork never picks the token that follows in the chain (async_spec_helpers.py:45-52)
            ex[:, :-1, :] = ex[:, :-1, :].scatter(2, glue_ids[:, 1:].unsqueeze(2), float("-inf"))
            self.decision_gaps.append(("glue", self._gap(ex, max(max(f) for f in fan_lists) + 1)))
        return O.fork_topf(lg, glue_ids, fan_lists)

    @torch.inference_mode()
    def draft_tree(self, forks, num_tokens, tables, jlists, temps=None, eagle=False):
        """K tree-decode steps (draft_runner.py:713-812): returns tokens [B*MQ, K] (and keeps the per-branch logits
        for `tree_logits` when some temperature is > 0)."""
        B, K = forks.shape[0], self.K
        mq = forks.shape[1]
        bt = self._bt_from(tables)
        toks = forks.reshape(-1)
        out = torch.zeros(B * mq, K, dtype=torch.int64)
        t = None if temps is None or not any(x > 0 for x in temps) else torch.tensor(temps, dtype=torch.float32).repeat_interleave(mq)
        self._sampling_rounds = t is not None
        tl, pres = [], []
        if eagle:       # branch i starts from the glue prenorm of its position j_i (draft_runner.py:660-676), then conditions on itself
            cond = torch.cat([self._glue_pre[b, torch.tensor(jlists[b])] for b in range(B)], dim=0)
        for d in range(K):
            pos, slots, ctx_lens = [], [], []
            for b, (n, tb) in enumerate(zip(num_tokens, tables)):
                Pb = n - 1
                for i in range(mq):
                    pos.append(Pb + jlists[b][i] + 1 + d)
                    slots.append(self._slot(tb, Pb + K + 1 + d * mq + i))
                ctx_lens.append(Pb + K + 1 + (d + 1) * mq)
            ctx = Ctx("tree", slot_mapping=torch.tensor(slots, dtype=torch.int32), context_lens=torch.tensor(ctx_lens, dtype=torch.int32),
                      block_tables=bt, tree_step=d, tree_K=K, tree_jidx=jlists)
            if eagle:
                lg, cond = self._eagle_logits(toks, torch.tensor(pos), cond, ctx)
                pres.append(cond)
            else:
                lg = self._logits(self.model.forward(toks, torch.tensor(pos), ctx))
            toks = O.argmax_rows(lg) if t is None else O.sample(lg, t, self.config.sampler_x, self.config.async_fan_out)
            tl.append(lg)
            out[:, d] = toks
        if self.log_decisions:
            self.decision_gaps.append(("tree", torch.stack([self._gap(l) for l in tl])))
        self._tree_lq = torch.stack(tl, dim=1) if t is not None else None
        self._tree_acts = torch.stack(pres, dim=1) if pres else None
        return out

    def tree_logits(self, T):
        return self._tree_lq[:T]

    def exit(self, *a):
        pass


def oracle_runner_factory(weights_target: dict | None = None, weights_draft: dict | None = None):
    def factory(config, model_cfg, *, is_draft: bool, topo, **kw):
        return OracleRunner(config, model_cfg, is_draft=is_draft, topo=topo,
                            weights=weights_draft if is_draft else weights_target,
                            num_kvcache_blocks=kw.get("num_kvcache_blocks", -1))
    return factory
