"""The decoder forward on MI355X: a fixed sequence of libssdhip launches over pre-allocated buffers.

Functionally LlamaForCausalLM.forward / Qwen3ForCausalLM.forward + compute_logits of the reference
(ssd/models/llama3.py:89-99,185-199,248-273; ssd/models/qwen3.py:90-108; ssd/layers/embed_head.py:78-116), with
the process-global attention Context (ssd/utils/context.py) turned into explicit arguments.

Per layer the launch list is: add+RMSNorm -> QKV GEMM -> (head-norm)+RoPE+KV-store -> paged attention
(+split merge) -> O GEMM [+all-reduce] -> add+RMSNorm -> gate_up GEMM with fused SiLU*mul -> down GEMM
[+all-reduce].  Activations that feed a GEMM are produced directly in the fragment-major layout; nothing is
allocated during a forward, so the whole thing can be captured in a hipGraph (torch.cuda.CUDAGraph is
hipGraph on ROCm) including the RCCL all-reduces.
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import torch
import torch.distributed as dist

from ssd_amd.hip import ops as H
from ssd_amd.model_config import ModelConfig

BF16 = torch.bfloat16


@dataclass
class AttnMeta:
    """What the reference keeps in its global Context, plus the tree geometry."""
    mode: int                     # H.MODE_CAUSAL | H.MODE_TREE
    B: int
    max_q: int
    slot_mapping: torch.Tensor    # int32 [T]
    context_lens: torch.Tensor    # int32 [B]
    block_tables: torch.Tensor    # int32 [B, max_blocks]
    cu_q: torch.Tensor | None = None
    q_per_seq: int = 0
    tree_K: int = 0
    tree_mq: int = 0
    tree_step: int = 0
    tree_F: int = 1
    tree_jidx: torch.Tensor | None = None
    ctx_hint: int = 0             # host-side upper bound of the context lengths (0 = unknown -> max_model_len)


def make_cos_sin(head_dim: int, max_pos: int, theta: float, device) -> torch.Tensor:
    """fp32 [max_pos, hd] = cos || sin, computed on the host exactly as RotaryEmbedding.__init__ does
    (ssd/layers/rotary_embedding.py:28-36) -- never sinf/cosf on the device."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous().to(device)


class HipDecoder:
    def __init__(self, cfg: ModelConfig, *, max_tokens: int, max_seqs: int, max_blocks: int, block_size: int,
                 max_model_len: int, device: torch.device, tp_rank: int = 0, tp_size: int = 1, tp_group=None,
                 max_logit_rows: int | None = None, max_split_tokens: int = 256, force_collectives: bool = False,
                 taps: list[int] | None = None):
        self.cfg, self.device = cfg, device
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        # force_collectives: issue the RCCL calls even at tp_size == 1 (lets a single-GPU box exercise the
        # collective + hipGraph-capture path that the multi-GPU runs depend on)
        self.use_coll = tp_size > 1 or (force_collectives and tp_group is not None)
        self.custom_ar = None      # OneShotAllReduce (ssd_amd/utils/custom_ar.py) once the runner has validated it
        self.fuse_ar_norm = True            # (all-reduce + add + RMSNorm in one launch; a probe may clear the attribute for an A/B run)
        assert cfg.num_heads % tp_size == 0 and cfg.num_kv_heads % tp_size == 0
        assert cfg.intermediate_size % (tp_size * 32) == 0 and cfg.vocab_size % (tp_size * 16) == 0
        self.nh, self.nkv = cfg.num_heads // tp_size, cfg.num_kv_heads // tp_size
        self.hd, self.h = cfg.head_dim, cfg.hidden_size
        self.I = cfg.intermediate_size // tp_size
        self.V = cfg.vocab_size // tp_size
        self.qkv_n = (self.nh + 2 * self.nkv) * self.hd
        self.qn = self.nh * self.hd
        self.block_size, self.max_blocks = block_size, max_blocks
        self.max_tokens = max_tokens
        self.max_logit_rows = max_logit_rows or max_tokens
        self.max_split_tokens = max_split_tokens
        self.max_model_len = max_model_len
        self.w: dict[str, torch.Tensor] = {}
        self.kv_cache: torch.Tensor | None = None
        self.cos_sin = make_cos_sin(self.hd, max_model_len, cfg.rope_theta, device)

        def z(*shape, dtype=BF16):
            return torch.zeros(*shape, dtype=dtype, device=device)

        T = max_tokens
        self.buf_h = z(T, self.h)
        self.buf_res = z(T, self.h)
        self.buf_res2 = z(min(T, 16), self.h)      # second residual buffer of the fused decode path (ping-pong)
        self.buf_xf = z(H.frag_numel(T, self.h))
        self.buf_qkv = z(T, self.qkv_n)
        self.buf_q = z(T, self.qn)
        self.buf_af = z(H.frag_numel(T, self.qn))
        self.buf_actf = z(H.frag_numel(T, self.I))
        self.buf_lastf = z(H.frag_numel(self.max_logit_rows, self.h))
        # split-K partial slabs of o_proj / down_proj for models whose hidden size gives too few 16-row groups to fill
        # the chip (csrc/gemm_sk.hip gemm_sp_kernel): fp32 [S][T][h], summed by the consumer's prologue / ssd_rmsnorm_parts
        # (< 256 row groups: the 1B / 0.6B drafts; 256 < groups < 512: a quarter-empty second round of workgroups --
        # Qwen3-32B o_proj 20.6 -> 15.6 us, down_proj 58.1 -> 48.1 us at 16 splits, profiles/r02_parts_probe.txt; at exactly
        # 256 or >= 512 groups the rows kernels are as fast or faster)
        g_ = self.h // 16
        self.use_parts = g_ < 256 or 256 < g_ < 512
        # ssd_gemm_parts keeps a wave's whole K share in registers: <= 8 k-tiles per wave.  A shape whose (splits, waves) plan
        # cannot meet that (e.g. h = 3072 with I = 14336 under the two-slab fused consumer) runs the rows kernels instead
        def fits(N, K, fused):
            S, wv = self._parts_cfg(N, K, fused)
            return -(-(-(-(K // 32) // S)) // wv) <= 8
        fused_possible = (not cfg.qk_norm) and self.h // 8 <= 1024          # fusion_plan(): the norm prologue exists at T = 1 only
        if self.use_parts and not all(fits(self.h, k, f) for k in (self.qn, self.I) for f in ((False, True) if fused_possible else (False,))):
            self.use_parts = False
        self.fuse_attn_o = True
        self.pf_parts = True
        # the single-token chain (one sequence, T = 1) with everything between two attention launches in ONE resident launch
        # (csrc/chain.hip): 1 + 2 per layer launches instead of 4 per layer
        # Default ("auto"): on at the geometry it was validated and measured at on the MI355X -- Llama-3.2-1B's, the draft of every
        # Llama configuration in BASELINE.json (tests/test_hip_chain.py, the full-size lock-step tests; draft forward 0.740 ->
        # 0.705 ms, c2 7.78 -> 7.50 ms / step, profiles/r04_chain_segment.txt).  SSD_CHAIN_SEG=1 forces it for every shape the
        # kernel accepts, =0 turns it off.
        _cs = os.environ.get("SSD_CHAIN_SEG", "auto")
        _geo = (self.h, self.qn, self.I, self.qkv_n, self.hd)
        _validated = _geo == (2048, 2048, 8192, 3072, 64)
        self.chain_seg = ((_cs == "1" or (_cs == "auto" and _validated)) and not cfg.qk_norm and tp_size == 1 and not self.use_coll
                          and taps is None and H.chain_segment_ok(self.h, self.qn, self.I, self.qkv_n, self.nh, self.nkv, self.hd))
        # the same segment for 2..30 token rows (csrc/tree_segment.hip): attention + ONE resident launch per layer instead of 7
        # launches.  Measured on MI355X at the 1B draft's geometry (profiles/r05_tree_seg_probe_v1.txt, parity tests/test_hip_tree_segment.py):
        # the K+1 = 8-row glue decode 0.945 -> 0.857 ms (kept), the 24-row tree step 0.943 -> 0.965 ms (NOT kept: at 24 rows every
        # all-to-all edge costs ~4.5 us of flag latency + ~5 us to read 96-192 KB of freshly written rows in every workgroup -- what the
        # six kernel boundaries it replaces cost).  "auto" = on at that geometry for forwards of 2..16 rows;
        # 1 = every shape and row count the kernel accepts (tests); 0 = off.
        _ts = os.environ.get("SSD_TREE_SEG", "auto")
        self.tree_seg_rows = 32 if _ts == "1" else 16
        self.tree_seg_colocated = False     # set by the engine for a draft server that shares its GPU with the target (llm_engine.py)
        # (a q / k norm model -- Qwen3-0.6B, the draft of BASELINE configs[4] -- leaves the segment with raw QKV rows; the norm + RoPE +
        #  KV store stay with ssd_rope_store_kv in front of the attention launch: 3 launches per layer instead of 8.  Parity-green at
        #  that geometry (tests/test_hip_tree_segment.py) but NOT faster -- a 0.6B layer streams 15 MB, so the segment's fixed edge cost
        #  is all there is: tree step 1.248 -> 1.240 ms at 6 rows, 1.275 -> 1.299 at 12, 1.363 -> 1.457 at 24,
        #  profiles/r05_tree_seg_probe_qwen.txt -- so "auto" leaves it off there)
        self.tree_seg = ((_ts == "1" or (_ts == "auto" and _validated)) and tp_size == 1 and not self.use_coll
                         and taps is None and max_tokens >= 2
                         and H.tree_segment_ok(2, self.h, self.qn, self.I, self.qkv_n, self.nh, self.nkv, self.hd))
        if self.chain_seg:
            self.chain_gr = z(H.chain_granule_bytes(self.h, self.I) // 8, dtype=torch.int64)
            self.buf_res3 = z(1, self.h)             # the chain's residual ping-pongs between buf_res2 and this (every workgroup re-reads res_in)
        if self.chain_seg or self.tree_seg:
            self.chain_gen = z(1, dtype=torch.int32)
            self.chain_err = z(1, dtype=torch.int32)
        if self.tree_seg:
            self.tree_ws = z(H.tree_segment_workspace_bytes(self.h, self.I) // 8, dtype=torch.int64)
            self.buf_res_b = z(min(T, 32), self.h)        # the residual ping-pongs: every workgroup re-reads a layer's input residual
        self._last_parts = False        # set by forward() for the compute_logits that follows it
        pt = min(T, 32)
        self.buf_parts_o = z(16 * pt * self.h, dtype=torch.float32) if self.use_parts else None
        self.buf_parts_d = z(16 * pt * self.h, dtype=torch.float32) if self.use_parts else None
        self.logits = z(self.max_logit_rows, self.V)
        # EAGLE-3 target (ssd/models/llama3.py:256-271): the residual stream ENTERING the tapped layers (bf16(hidden + residual),
        # = the residual the layer's input add+norm writes anyway), concatenated in layer order for the draft's fc
        self.taps = sorted(set(taps)) if taps else None
        self.acts = z(T, len(self.taps) * self.h) if self.taps else None
        self.max_splits = 16
        # fp32 split-K partials of the prefill GEMM (csrc/gemm_pf.hip), sized ONCE for the largest matrix that can take that
        # path (layer matrices and the LM head): prefill hipGraphs bake this pointer, so it must never be reallocated
        pf_shapes = [(self.qkv_n, self.h), (self.h, self.qn), (2 * self.I, self.h), (self.h, self.I), (self.V, self.h)]
        need = max([H.gemm_pf_workspace_bytes(128, n, k) // 4 for n, k in pf_shapes if self._pf_eligible(n, k)], default=0)
        self._ws_pf = z(need, dtype=torch.float32) if need and max_tokens > 32 else None
        st = min(T, max_split_tokens)
        self.ws_o = z(st * self.nh * self.max_splits * self.hd, dtype=torch.float32)
        self.ws_ml = z(st * self.nh * self.max_splits * 2, dtype=torch.float32)
        # LM-head argmax candidates (csrc/gemm.hip EPI_ROWS_ARGMAX): one (value, index) per token row and workgroup, finished
        # by ssd_argmax_parts* -- on the greedy path for up to 32 logit rows (decode / verify / tree step)
        self.argmax_fused = True
        self.ap_rows = min(self.max_logit_rows, 32)
        self.ap_stride = max(H.gemm_argmax_nparts(m, self.V, self.h) for m in ({1, self.ap_rows} if self.ap_rows > 16 else {1}))
        self.ap_val = z(self.ap_rows, self.ap_stride, dtype=torch.float32)
        self.ap_idx = z(self.ap_rows, self.ap_stride, dtype=torch.int32)
        # (whether candidates exist for n rows is a function of n alone -- never of Python-side state, which a hipGraph replay
        #  of compute_logits would not update)
        # vocab-parallel argmax scratch
        self.am_val = z(tp_size, self.max_logit_rows, dtype=torch.float32)
        self.am_idx = z(tp_size, self.max_logit_rows, dtype=torch.int64)
        self.am_val_l = z(self.max_logit_rows, dtype=torch.float32)
        self.am_idx_l = z(self.max_logit_rows, dtype=torch.int64)
        # packed form for the one-shot exchange: per rank [R int64 indices | R fp32 values (+ pad to 8 bytes)]
        self.am_words = self.max_logit_rows + (self.max_logit_rows + 1) // 2
        self.am_pack_l = z(self.am_words, dtype=torch.int64)
        self.am_pack = z(tp_size, self.am_words, dtype=torch.int64)

    # ---------------------------------------------------------------------------------------------
    def load_weights(self, weight_iter) -> None:
        """Consumes (name, row-major bf16 shard) pairs; matrices are re-tiled once into the fragment-major
        layout (gate_up with gate/up row groups interleaved for the fused SiLU epilogue)."""
        for name, w in weight_iter:
            w = w.to(self.device).contiguous()
            if name.endswith("qkv_proj.weight"):
                # rotation-paired row order: RoPE pairs share an accumulator tile (fused epilogue) -- layout.hip
                out = torch.empty(w.numel(), dtype=BF16, device=self.device)
                H.rows_to_frag_qkv(w, out, self.nh, self.nkv, self.hd, w.shape[1])
                self.w[name] = out
            elif name.endswith("qkv_proj.bias"):
                self.w[name] = w[self._qkv_row_perm().to(self.device)].contiguous()
            elif w.dim() == 2 and not name.endswith("embed_tokens.weight"):
                R, K = w.shape
                out = torch.empty(H.frag_numel(R, K), dtype=BF16, device=self.device)
                H.rows_to_frag(w, out, R, K, mode=1 if name.endswith("gate_up_proj.weight") else 0)
                self.w[name] = out
            elif name.endswith("embed_tokens.weight"):
                self.w[name] = w                                   # row-major for the gather
                if self.cfg.tie_word_embeddings:                   # tied LM head: same values, GEMM layout
                    out = torch.empty(H.frag_numel(w.shape[0], w.shape[1]), dtype=BF16, device=self.device)
                    H.rows_to_frag(w, out, w.shape[0], w.shape[1])
                    self.w["lm_head.weight"] = out
            else:
                self.w[name] = w
        torch.cuda.synchronize(self.device)

    def overwrite_weights(self, weight_iter) -> None:
        """New VALUES into the existing weight tensors (same names and shapes): captured hipGraphs keep pointing at the same memory, so a
        model can be given other weights without recapturing anything (bench.py: the independent-draft leg after the correlated one)."""
        old, self.w = self.w, {}
        self.load_weights(weight_iter)
        new, self.w = self.w, old
        assert set(new) == set(old), sorted(set(new) ^ set(old))
        for n, t in new.items():
            old[n].copy_(t)
        torch.cuda.synchronize(self.device)

    def _qkv_row_perm(self) -> torch.Tensor:
        """Source row of every destination row of the rotation-paired QKV order (same map as ssd_rows_to_frag_qkv)."""
        hd, half, gph = self.hd, self.hd // 2, self.hd // 16
        idx = []
        for head in range(self.nh + self.nkv):
            for j in range(gph):
                idx.extend(head * hd + 8 * j + i for i in range(8))
                idx.extend(head * hd + half + 8 * j + i for i in range(8))
        idx.extend(range((self.nh + self.nkv) * hd, (self.nh + 2 * self.nkv) * hd))
        return torch.tensor(idx, dtype=torch.int64)

    def weight_bytes(self) -> int:
        """HBM bytes one forward must stream (every matrix once; the embedding table is only gathered)."""
        return sum(t.numel() * t.element_size() for n, t in self.w.items() if n != "model.embed_tokens.weight")

    def kv_block_bytes(self) -> int:
        return 2 * self.cfg.num_layers * self.block_size * self.nkv * self.hd * 2

    def alloc_kv(self, num_blocks: int) -> None:
        # [L][2][blocks][nkv][block_size][hd]: one (page, kv head) is a contiguous run for the attention kernel
        self.num_blocks = num_blocks
        self.kv_cache = torch.zeros(self.cfg.num_layers, 2, num_blocks, self.nkv, self.block_size, self.hd,
                                    dtype=BF16, device=self.device)

    # ---------------------------------------------------------------------------------------------
    PF_MIN_WEIGHT_BYTES = 100 << 20

    @classmethod
    def _pf_eligible(cls, N: int, K: int) -> bool:
        return 2 * N * K >= cls.PF_MIN_WEIGHT_BYTES and N % 128 == 0 and K % 128 == 0

    def _gemm_chunk(self, xf, K, w, N, y, m, ldy, epi, bias):
        """One <= 128-row chunk.  Prefill-sized chunks of a big matrix go to the LDS-shared / split-K kernel
        (csrc/gemm_pf.hip): measured on MI355X at M = 128, 70B layer GEMMs 611 -> 383 us, 8B gate_up 83 -> 62 us; small
        matrices (<100 MB: too few workgroups without deep K-splits) stay on the skinny kernel."""
        if m > 32 and self._pf_eligible(N, K):
            assert self._ws_pf is not None and self._ws_pf.numel() * 4 >= H.gemm_pf_workspace_bytes(128, N, K), (N, K)
            H.gemm_pf(xf, w, y, m, N, K, ldy, self._ws_pf, epilogue=epi, bias=bias)
        else:
            H.gemm(xf, w, y, m, N, K, ldy, epi, bias)

    def _pf_partials_ok(self, T: int, N: int, K: int) -> bool:
        """o_proj / down_proj of ONE prefill chunk may leave their split-K partials in the workspace for the norm that follows."""
        return self.pf_parts and not self.use_coll and 32 < T <= 128 and self._pf_eligible(N, K) and self._ws_pf is not None

    @staticmethod
    def _pf_splits(T: int, N: int, K: int) -> int:
        return H.gemm_pf_workspace_bytes(T, N, K) // (4 * T * N)

    def _gemm_pf_partials(self, xf, K, w, N, T) -> int:
        """Prefill GEMM stopped after its split-K stage (csrc/gemm_pf.hip PF_EPI_PARTIALS): fp32 slabs [S][T][N] in the workspace;
        returns S.  The consumer is ssd_rmsnorm_parts -- one launch per GEMM less."""
        S = H.gemm_pf_workspace_bytes(T, N, K) // (4 * T * N)
        H.gemm_pf(xf, w, None, T, N, K, N, self._ws_pf, epilogue=H.PF_EPI_PARTIALS)
        return S

    def _gemm(self, xf, K, w, N, y, T, ldy, epi=H.EPI_ROWS, bias=None):
        if T <= 128:
            self._gemm_chunk(xf, K, w, N, y, T, ldy, epi, bias)
            return
        yf = y.view(-1)
        for m0 in range(0, T, 128):
            m = min(128, T - m0)
            x_off = (m0 // 16) * (K // 32) * 512
            y_off = (m0 // 16) * ((N // 2) // 32) * 512 if epi == H.EPI_SILU_FRAG else m0 * ldy
            self._gemm_chunk(xf[x_off:], K, w, N, yf[y_off:], m, ldy, epi, bias)

    def _allreduce(self, t):
        if not self.use_coll:
            return
        if self.custom_ar is not None and self.custom_ar.fits(t):
            self.custom_ar.all_reduce(t)        # one-shot full-mesh sum over xGMI (csrc/comm.hip)
        else:
            dist.all_reduce(t, group=self.tp_group)

    @staticmethod
    def ctx_bucket(ctx: int) -> int:
        """Power-of-two context bucket (>= 1024): the attention decomposition is static per hipGraph, and up to
        1024 keys a single workgroup per (sequence, kv head) with no merge kernel is the fastest one."""
        b = 1024
        while b < ctx:
            b *= 2
        return b

    def _attn_cfg(self, T: int, meta: AttnMeta) -> tuple[int, int]:
        """(grid key-splits, waves per workgroup).  Up to 8 waves of one workgroup split the key range and merge
        in LDS (one launch).  A decode-side launch occupies only B*nkv*row_tiles CUs and each CU sustains a few
        tens of GB/s of K/V loads, so beyond ~1K keys per workgroup the scan is spread over more workgroups with
        grid key-splits + the merge kernel (measured on MI355X, 1B decode: ctx 2048 22 -> 13 us, 4096 39 -> 17 us;
        below 1K keys the extra launch costs more than it saves)."""
        G = self.nh // self.nkv
        row_tiles = -(-(meta.max_q * G) // 16)
        groups = -(-row_tiles // 2) if row_tiles > 8 else row_tiles       # csrc/attention.hip attn_launch
        base = max(1, groups * meta.B * self.nkv)
        waves = max(1, min(8, 512 // base))
        # (prefill with 4 / 8 waves per workgroup, measured on c4: TTFT 30.92 ms (this heuristic, 2 waves) / 30.68 / 30.74 -- within noise, and
        #  another accumulation order: not adopted, the override switch retired in round 5)
        ctx = self.ctx_bucket(meta.ctx_hint if meta.ctx_hint > 0 else self.max_model_len)
        splits = 1 if ctx <= 1024 else max(1, min(self.max_splits, ctx // 512))
        if base >= 256 or T > self.max_split_tokens:
            splits = 1
        return splits, waves

    def _attn_flags(self, meta: AttnMeta) -> int:
        """ssd_attn_paged flags.  Query blocks of more than 8 row tiles per kv head (prefill chunks) take ONE 16-row tile per workgroup
        (bit 2) instead of the kernel's default two: twice the workgroups, K tiles prefetched (the two-tile variant of head_dim 128 has no
        registers for that) -- the same key split per wave, so bit-identical; measured at T = 128 (profiles/r06_prefill_small_kernels.txt):
        70B 17.0 -> 12.2 us, 8B 16.8 -> 10.6, 1B 12.4 -> 8.1; T = 512: 48.8 -> 46.6."""
        G = self.nh // self.nkv
        return 4 if -(-(meta.max_q * G) // 16) > 8 else 0

    @staticmethod
    def _parts_cfg(N: int, K: int, fused_consumer: bool = False) -> tuple[int, int]:
        """(K splits, waves) of the split-K partial-slab GEMM for a [N, K] matrix, from profiles/r02_draft_probe.txt (1B
        shapes, us per launch; rows kernel -> slabs): M = 1: o_proj 4.5 -> 3.7 (4 splits, 8 waves) / 3.9 (2, 16), down_proj
        10.0 -> 7.6 (4, 8) / 8.0 (2, 16); M = 24: 8.1 -> 4.2 and 16.7 -> 9.0.  Enough workgroups to put >= 2 on every CU and
        <= 8 k-tiles per wave, so that every wave has its whole share in flight at once.  When the consumer is the fused
        norm + GEMM prologue (every workgroup of the NEXT kernel re-reads the slabs) two slabs are the optimum: a third
        and fourth cost the consumers more than they save here."""
        groups, KT = N // 16, K // 32
        smax = 2 if fused_consumer else (16 if groups > 256 else 4)
        S = 1
        while groups * S < (5120 if groups > 256 else 512) and S < smax and KT // (S * 2) >= 8:
            S *= 2
        per = -(-KT // S)
        waves = 16 if fused_consumer else 8
        while waves < 16 and -(-per // waves) > 8:
            waves *= 2
        while waves > 2 and per // waves < 1:
            waves //= 2
        # a very long K: the kernel holds a wave's whole share in registers (<= 8 k-tiles).  More slabs where the consumer is a
        # stand-alone norm (up to the 16 the slab buffers hold); the two-slab fused consumer cannot take more -- __init__ then
        # keeps such a model on the rows kernels
        while not fused_consumer and S < 16 and -(-(-(-KT // S)) // waves) > 8:
            S *= 2
            waves = 16
        return S, waves

    def _parts(self, which: str, T: int) -> tuple[int, int]:
        fused = self.fusion_plan(T)[1]
        N, K = (self.h, self.qn) if which == "o" else (self.h, self.I)
        return self._parts_cfg(N, K, fused)

    def parts_plan(self, T: int) -> bool:
        """o_proj / down_proj as split-K partial slabs consumed by the next norm: single-rank models with < 256 row groups at
        decode-sized T (the draft's chain / glue / tree forwards)."""
        return self.use_parts and not self.use_coll and T <= 32

    def attn_o_plan(self, T: int, meta: AttnMeta, splits: int) -> bool:
        """Attention + o_proj as ONE launch (csrc/attention.hip OPROJ variant): single-rank models on the slab path, one
        sequence, causal decode rows, context scanned inside one workgroup (bucket <= 1024).  Only while the query rows of a
        kv head fit ONE 16-row tile (the single-token chain, up to 4 rows at G = 4): the N/128 workgroups of a head each
        repeat its attention, and with two row tiles (the K+1 = 8-row glue) that costs more than the saved launch --
        measured on the 1B draft (profiles/r03_draft_probe.txt): M = 1 forward 734 -> 718 us, M = 8 forward 838 -> 867 us.
        (Grouped-query models only, G >= 2: the shapes the kernel was validated at on the MI355X -- G = 4 and 2.)"""
        G = self.nh // self.nkv
        return (self.fuse_attn_o and self.parts_plan(T) and meta.mode == H.MODE_CAUSAL and meta.cu_q is None and meta.B == 1
                and splits == 1 and G >= 2 and T * G <= 16 and G * self.hd <= 256
                and self.h % 128 == 0 and self.nkv <= 16)

    # ---- the four GEMM launches of a layer (also used one by one by bench.py's roofline timing) ----
    def fusion_plan(self, T: int) -> tuple[bool, bool]:
        """(small, norm_fuse).  T <= 16: RoPE + KV store ride the QKV epilogue (csrc/gemm_fused.hip).  The residual
        add + RMSNorm ride the GEMM prologue only while M*K is tiny (single-token draft decode) and no all-reduce sits
        between producer and norm: the prologue is paid by EVERY workgroup and its LDS image limits residency --
        measured on MI355X, M=7 x K=4096 made gate_up 73 us vs 49 us unfused, M=1 x K=2048 made norm+qkv+rope 5.6 us
        vs 14.8 us."""
        small = T <= 16 and not self.cfg.qk_norm
        return small, small and not self.use_coll and T * self.h // 8 <= 1024

    def chain_plan(self, T: int, meta: AttnMeta, splits: int) -> bool:
        """The resident single-token chain (csrc/chain.hip): one sequence, one new token, causal attention scanned inside one
        workgroup, the fused norm + QKV launch available for layer 0, no biases."""
        return (self.chain_seg and T == 1 and meta.B == 1 and meta.cu_q is None and meta.mode == H.MODE_CAUSAL and splits == 1
                and self.fusion_plan(T)[1] and "model.layers.0.self_attn.qkv_proj.bias" not in self.w)

    def _forward_chain(self, positions, meta: AttnMeta, attn_waves: int) -> None:
        cfg, w = self.cfg, self.w
        L = cfg.num_layers
        scale = self.hd ** -0.5
        H.chain_tick(self.chain_gen)
        self.launch_qkv(0, 1, positions, meta.slot_mapping, parts=False)       # layer 0: norm(embedding) + QKV + RoPE + KV store
        for li in range(L):
            H.attn_paged(self.buf_q, self.kv_cache[li, 0], self.kv_cache[li, 1], meta.block_tables, self.max_blocks,
                         meta.context_lens, meta.B, 1, meta.max_q, self.nh, self.nkv, self.hd, self.block_size, scale,
                         cu_q=None, q_per_seq=meta.q_per_seq, mode=meta.mode, splits=1, ws_o=self.ws_o, ws_ml=self.ws_ml,
                         out_frag=self.buf_af, waves=attn_waves)
            p = f"model.layers.{li}."
            last = li + 1 == L
            nxt = {} if last else dict(
                w_qkv_next=w[f"model.layers.{li + 1}.self_attn.qkv_proj.weight"], ln_next=w[f"model.layers.{li + 1}.input_layernorm.weight"],
                positions=positions, cos_sin=self.cos_sin, slots=meta.slot_mapping, q_out=self.buf_q,
                k_cache=self.kv_cache[li + 1, 0], v_cache=self.kv_cache[li + 1, 1])
            rin, rout = (self.buf_res2, self.buf_res3) if li % 2 == 0 else (self.buf_res3, self.buf_res2)
            H.chain_segment(self.buf_af, rin, self.buf_res if last else rout, w[p + "self_attn.o_proj.weight"],
                            w[p + "mlp.gate_up_proj.weight"], w[p + "mlp.down_proj.weight"], w[p + "post_attention_layernorm.weight"],
                            cfg.rms_norm_eps, self.h, self.qn, self.I, self.qkv_n, self.nh, self.nkv, self.hd, self.block_size, li,
                            self.chain_gr, self.chain_gen, self.chain_err, h_out=self.buf_h if last else None, **nxt)

    def tree_plan(self, T: int, meta: AttnMeta) -> bool:
        """The resident M-row layer segment (csrc/tree_segment.hip): decode-side forwards of 2..30 rows (tree steps, glue), no biases."""
        return (self.tree_seg and not self.tree_seg_colocated and 2 <= T <= self.tree_seg_rows and meta.cu_q is None and "model.layers.0.self_attn.qkv_proj.bias" not in self.w
                and H.tree_segment_ok(T, self.h, self.qn, self.I, self.qkv_n, self.nh, self.nkv, self.hd))

    def _forward_tree_seg(self, positions, T: int, meta: AttnMeta, splits: int, attn_waves: int) -> None:
        """embedding rows in buf_h -> [norm + QKV + RoPE + KV store of layer 0] -> per layer: attention, segment."""
        cfg, w = self.cfg, self.w
        L = cfg.num_layers
        scale = self.hd ** -0.5
        res = [self.buf_res, self.buf_res_b]
        start = L % 2               # layer li reads res[(start + li) % 2] and writes the other: the last layer's lands in buf_res
        H.chain_tick(self.chain_gen)
        H.rmsnorm(self.buf_h, w["model.layers.0.input_layernorm.weight"], cfg.rms_norm_eps, T, self.h, res_in=None, res_out=res[start],
                  out_frag=self.buf_xf)
        qk = cfg.qk_norm
        if qk:
            self._gemm(self.buf_xf, self.h, w["model.layers.0.self_attn.qkv_proj.weight"], self.qkv_n, self.buf_qkv, T, self.qkv_n)
        else:
            H.gemm_fused(w["model.layers.0.self_attn.qkv_proj.weight"], T, self.qkv_n, self.h, H.FEPI_QKV_ROPE, x_frag=self.buf_xf,
                         positions=positions, cos_sin=self.cos_sin, slots=meta.slot_mapping, q_out=self.buf_q, k_cache=self.kv_cache[0, 0],
                         v_cache=self.kv_cache[0, 1], nh=self.nh, nkv=self.nkv, hd=self.hd, block_size=self.block_size)
        for li in range(L):
            if qk:      # per-head q / k RMSNorm + RoPE + KV store of the rows the previous segment (or the GEMM above) left in buf_qkv
                p_ = f"model.layers.{li}.self_attn."
                H.rope_store_kv(self.buf_qkv, positions, self.cos_sin, meta.slot_mapping, self.buf_q, self.kv_cache[li, 0], self.kv_cache[li, 1],
                                T, self.nh, self.nkv, self.hd, self.block_size, q_norm_w=w[p_ + "q_norm.weight"],
                                k_norm_w=w[p_ + "k_norm.weight"], eps=cfg.rms_norm_eps, qkv_perm=1)
            H.attn_paged(self.buf_q, self.kv_cache[li, 0], self.kv_cache[li, 1], meta.block_tables, self.max_blocks,
                         meta.context_lens, meta.B, T, meta.max_q, self.nh, self.nkv, self.hd, self.block_size, scale,
                         cu_q=None, q_per_seq=meta.q_per_seq, mode=meta.mode, tree_K=meta.tree_K, tree_mq=meta.tree_mq,
                         tree_step=meta.tree_step, tree_F=meta.tree_F, tree_jidx=meta.tree_jidx, splits=splits,
                         ws_o=self.ws_o, ws_ml=self.ws_ml, out_frag=self.buf_af, waves=attn_waves)
            p = f"model.layers.{li}."
            last = li + 1 == L
            nxt = {} if last else dict(
                w_qkv_next=w[f"model.layers.{li + 1}.self_attn.qkv_proj.weight"], ln_next=w[f"model.layers.{li + 1}.input_layernorm.weight"],
                qkv_rows_next=self.buf_qkv) if qk else dict(
                w_qkv_next=w[f"model.layers.{li + 1}.self_attn.qkv_proj.weight"], ln_next=w[f"model.layers.{li + 1}.input_layernorm.weight"],
                positions=positions, cos_sin=self.cos_sin, slots=meta.slot_mapping, q_out=self.buf_q,
                k_cache=self.kv_cache[li + 1, 0], v_cache=self.kv_cache[li + 1, 1])
            H.tree_segment(self.buf_af, res[(start + li) % 2], res[(start + li + 1) % 2], w[p + "self_attn.o_proj.weight"],
                           w[p + "mlp.gate_up_proj.weight"], w[p + "mlp.down_proj.weight"], w[p + "post_attention_layernorm.weight"],
                           cfg.rms_norm_eps, T, self.h, self.qn, self.I, self.qkv_n, self.nh, self.nkv, self.hd, self.block_size, li,
                           self.tree_ws, self.chain_gen, self.chain_err, h_out=self.buf_h if last else None, **nxt)

    def launch_qkv(self, li: int, T: int, positions, slot_mapping, gemm_only: bool = False, pre_normed: bool = False,
                   parts: bool | None = None, pf_src: int = 0) -> None:
        """gemm_only: skip the separate add+RMSNorm / RoPE launches of the unfused variants (kernel timing).
        pre_normed: buf_xf / buf_res already hold this layer's normalised input and residual (written by the fused
        all-reduce + add + RMSNorm that closed the previous layer)."""
        cfg, w = self.cfg, self.w
        p = f"model.layers.{li}."
        small, norm_fuse = self.fusion_plan(T)
        kc, vc = self.kv_cache[li, 0], self.kv_cache[li, 1]
        rope = dict(positions=positions, cos_sin=self.cos_sin, slots=slot_mapping, q_out=self.buf_q, k_cache=kc, v_cache=vc,
                    nh=self.nh, nkv=self.nkv, hd=self.hd, block_size=self.block_size)
        h, res, xf = self.buf_h, self.buf_res, self.buf_xf
        parts = (self.parts_plan(T) if parts is None else parts) and li > 0   # the previous layer's down_proj left fp32 slabs, not rows
        if norm_fuse:
            src = dict(h_parts=self.buf_parts_d, splits=self._parts("d", T)[0]) if parts else dict(h_rows=h)
            H.gemm_fused(w[p + "self_attn.qkv_proj.weight"], T, self.qkv_n, self.h, H.FEPI_QKV_ROPE,
                         res_in=None if li == 0 else res, res_out=self.buf_res2, norm_w=w[p + "input_layernorm.weight"],
                         eps=cfg.rms_norm_eps, bias=w.get(p + "self_attn.qkv_proj.bias"), waves=16, **src, **rope)
            return
        # residual is None on layer 0 (llama3.py:187-190): residual := embeddings, x := norm(embeddings)
        if not gemm_only and not pre_normed:
            if pf_src:       # the previous layer's down_proj left pf_src split-K slabs [pf_src][T][h] in the prefill workspace
                H.rmsnorm_parts(self._ws_pf, pf_src, T, w[p + "input_layernorm.weight"], cfg.rms_norm_eps, T, self.h,
                                res_in=res, res_out=res, out_frag=xf)
            elif parts:
                H.rmsnorm_parts(self.buf_parts_d, self._parts("d", T)[0], T, w[p + "input_layernorm.weight"], cfg.rms_norm_eps, T, self.h,
                                res_in=res, res_out=res, out_frag=xf)
            else:
                H.rmsnorm(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps, T, self.h, res_in=None if li == 0 else res,
                          res_out=res, out_frag=xf)
        if small or (16 < T <= 32 and not cfg.qk_norm):      # T in 17..32 (tree-decode step): the two-token-tile variant
            H.gemm_fused(w[p + "self_attn.qkv_proj.weight"], T, self.qkv_n, self.h, H.FEPI_QKV_ROPE, x_frag=xf,
                         bias=w.get(p + "self_attn.qkv_proj.bias"), **rope)
        elif (not gemm_only and w.get(p + "self_attn.qkv_proj.bias") is None and self._pf_partials_ok(T, self.qkv_n, self.h)
              and self._pf_splits(T, self.qkv_n, self.h) > 1):
            # single-chunk prefill of a big QKV matrix: its split-K slabs stay in the workspace and the RoPE / KV-store kernel
            # sums them (bit-identical to the GEMM's epilogue launch + rope_store_kv; one launch less per layer)
            S = self._gemm_pf_partials(xf, self.h, w[p + "self_attn.qkv_proj.weight"], self.qkv_n, T)
            H.rope_store_kv_parts(self._ws_pf, S, positions, self.cos_sin, slot_mapping, self.buf_q, kc, vc, T, self.nh, self.nkv,
                                  self.hd, self.block_size, q_norm_w=w.get(p + "self_attn.q_norm.weight"),
                                  k_norm_w=w.get(p + "self_attn.k_norm.weight"), eps=cfg.rms_norm_eps, qkv_perm=1)
        else:
            self._gemm(xf, self.h, w[p + "self_attn.qkv_proj.weight"], self.qkv_n, self.buf_qkv, T, self.qkv_n,
                       bias=w.get(p + "self_attn.qkv_proj.bias"))
            if gemm_only:
                return
            H.rope_store_kv(self.buf_qkv, positions, self.cos_sin, slot_mapping, self.buf_q, kc, vc, T, self.nh, self.nkv,
                            self.hd, self.block_size, q_norm_w=w.get(p + "self_attn.q_norm.weight"),
                            k_norm_w=w.get(p + "self_attn.k_norm.weight"), eps=cfg.rms_norm_eps, qkv_perm=1)

    def launch_o(self, li: int, T: int, parts: bool | None = None, pf_partials: bool = False) -> int:
        """Returns the number of split-K slabs left in the prefill workspace (pf_partials), else 0."""
        w = self.w[f"model.layers.{li}.self_attn.o_proj.weight"]
        if pf_partials and self._pf_partials_ok(T, self.h, self.qn):
            return self._gemm_pf_partials(self.buf_af, self.qn, w, self.h, T)
        if self.parts_plan(T) if parts is None else parts:
            S, wv = self._parts("o", T)
            H.gemm_parts(self.buf_af, w, T, self.h, self.qn, parts=self.buf_parts_o, splits=S, waves=wv)
        else:
            self._gemm(self.buf_af, self.qn, w, self.h, self.buf_h, T, self.h)
        return 0

    def launch_gate_up(self, li: int, T: int, gemm_only: bool = False, pre_normed: bool = False, parts: bool | None = None,
                       o_splits: int | None = None, pf_src: int = 0) -> None:
        """o_splits: slabs o_proj left in buf_parts_o (default: the split-K GEMM's; the fused attention + o_proj leaves nkv)."""
        cfg, w = self.cfg, self.w
        p = f"model.layers.{li}."
        _, norm_fuse = self.fusion_plan(T)
        parts = self.parts_plan(T) if parts is None else parts      # o_proj left fp32 partial slabs
        So = o_splits if o_splits is not None else (self._parts("o", T)[0] if parts else 1)
        if norm_fuse:
            src = dict(h_parts=self.buf_parts_o, splits=So) if parts else dict(h_rows=self.buf_h)
            H.gemm_fused(w[p + "mlp.gate_up_proj.weight"], T, 2 * self.I, self.h, H.FEPI_SILU_FRAG,
                         res_in=self.buf_res2, res_out=self.buf_res, norm_w=w[p + "post_attention_layernorm.weight"],
                         eps=cfg.rms_norm_eps, y=self.buf_actf, waves=8, **src)     # profiles/micro/fused_probe.py: 13.2 us vs 16.4 (16 waves)
        else:
            if not gemm_only and not pre_normed:
                if pf_src:       # o_proj left pf_src split-K slabs in the prefill workspace
                    H.rmsnorm_parts(self._ws_pf, pf_src, T, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps,
                                    T, self.h, res_in=self.buf_res, res_out=self.buf_res, out_frag=self.buf_xf)
                elif parts:
                    H.rmsnorm_parts(self.buf_parts_o, So, T, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps,
                                    T, self.h, res_in=self.buf_res, res_out=self.buf_res, out_frag=self.buf_xf)
                else:
                    H.rmsnorm(self.buf_h, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, T, self.h,
                              res_in=self.buf_res, res_out=self.buf_res, out_frag=self.buf_xf)
            self._gemm(self.buf_xf, self.h, w[p + "mlp.gate_up_proj.weight"], 2 * self.I, self.buf_actf, T, 0, epi=H.EPI_SILU_FRAG)

    def launch_down(self, li: int, T: int, parts: bool | None = None, pf_partials: bool = False) -> int:
        w = self.w[f"model.layers.{li}.mlp.down_proj.weight"]
        if pf_partials and self._pf_partials_ok(T, self.h, self.I):
            return self._gemm_pf_partials(self.buf_actf, self.I, w, self.h, T)
        if self.parts_plan(T) if parts is None else parts:
            S, wv = self._parts("d", T)
            H.gemm_parts(self.buf_actf, w, T, self.h, self.I, parts=self.buf_parts_d, splits=S, waves=wv)
        else:
            self._gemm(self.buf_actf, self.I, w, self.h, self.buf_h, T, self.h)
        return 0

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, T: int, meta: AttnMeta) -> None:
        """Runs all layers; leaves the final (pre-norm) hidden state in buf_h and the residual in buf_res."""
        cfg, w = self.cfg, self.w
        h = self.buf_h
        H.embedding(input_ids, w["model.embed_tokens.weight"], h, T, self.h,
                    vocab_start=self.tp_rank * self.V if self.tp_size > 1 else 0, vocab_count=self.V if self.tp_size > 1 else cfg.vocab_size)
        self._allreduce(h[:T])
        splits, attn_waves = self._attn_cfg(T, meta)
        attn_flags = self._attn_flags(meta)
        scale = self.hd ** -0.5
        # a varlen prefill never takes the slab path (its last-token gather reads rows); compute_logits, which always follows
        # in the same Python body, must know whether the last down_proj left rows or partial slabs
        parts = self.parts_plan(T) and meta.cu_q is None
        self._fwd_T, self._last_parts = T, parts
        if self.chain_plan(T, meta, splits):
            self._last_parts = False            # the last segment leaves rows (buf_h) + the residual (buf_res) for compute_logits
            self._forward_chain(positions, meta, attn_waves)
            return
        if self.tree_plan(T, meta):
            self._last_parts = False            # rows (buf_h) + the residual (buf_res), as above
            self._forward_tree_seg(positions, T, meta, splits, attn_waves)
            return
        fuse_ao = parts and self.attn_o_plan(T, meta, splits)
        # tensor parallel with the one-shot collective: the all-reduce after o_proj / down_proj absorbs the residual add
        # and the RMSNorm that follow it (csrc/comm.hip), 2 launches fewer per half layer
        ar = self.custom_ar
        fuse = self.use_coll and ar is not None and self.fuse_ar_norm and ar.fits_rows(T, self.h)
        eps, res, xf = cfg.rms_norm_eps, self.buf_res, self.buf_xf
        L = cfg.num_layers
        pf_d = 0            # split-K slabs the previous layer's down_proj left for this layer's input norm (single-chunk prefill)
        for li in range(L):
            self.launch_qkv(li, T, positions, meta.slot_mapping, pre_normed=fuse and li > 0, parts=parts, pf_src=pf_d)
            if self.taps is not None and li in self.taps:
                # launch_qkv has just written x + residual: to buf_res2 on the fused-prologue path, to buf_res otherwise
                src = self.buf_res2 if self.fusion_plan(T)[1] else res
                i = self.taps.index(li)
                self.acts[:T, i * self.h:(i + 1) * self.h].copy_(src[:T])
            pf_o = 0
            if fuse_ao:
                H.attn_oproj_parts(self.buf_q, self.kv_cache[li, 0], self.kv_cache[li, 1], meta.block_tables, self.max_blocks,
                                   meta.context_lens, T, self.nh, self.nkv, self.hd, self.block_size, scale,
                                   w[f"model.layers.{li}.self_attn.o_proj.weight"], self.h, self.buf_parts_o)
            else:
                H.attn_paged(self.buf_q, self.kv_cache[li, 0], self.kv_cache[li, 1], meta.block_tables, self.max_blocks,
                             meta.context_lens, meta.B, T, meta.max_q, self.nh, self.nkv, self.hd, self.block_size, scale,
                             cu_q=meta.cu_q, q_per_seq=meta.q_per_seq, mode=meta.mode, tree_K=meta.tree_K, tree_mq=meta.tree_mq,
                             tree_step=meta.tree_step, tree_F=meta.tree_F, tree_jidx=meta.tree_jidx, splits=splits,
                             flags=attn_flags, ws_o=self.ws_o, ws_ml=self.ws_ml, out_frag=self.buf_af, waves=attn_waves)
                pf_o = self.launch_o(li, T, parts=parts, pf_partials=not parts)
            if fuse:
                ar.all_reduce_add_rmsnorm(h, res, res, w[f"model.layers.{li}.post_attention_layernorm.weight"], eps, T, self.h, out_frag=xf)
            else:
                self._allreduce(h[:T])
            self.launch_gate_up(li, T, pre_normed=fuse, parts=parts, o_splits=self.nkv if fuse_ao else None, pf_src=pf_o)
            # (the last layer's down_proj keeps its epilogue: compute_logits' last-token gather reads rows)
            pf_d = self.launch_down(li, T, parts=parts, pf_partials=not parts and li + 1 < L)
            if fuse and li + 1 < L:
                ar.all_reduce_add_rmsnorm(h, res, res, w[f"model.layers.{li + 1}.input_layernorm.weight"], eps, T, self.h, out_frag=xf)
            else:
                self._allreduce(h[:T])

    def compute_logits(self, T: int, gather: torch.Tensor | None = None, rows: int | None = None) -> int:
        """Final add+RMSNorm (optionally only the `gather` rows: prefill last-token, embed_head.py:81-84) and the
        LM-head GEMM into self.logits[:rows] (this rank's vocab shard).  Returns the number of logit rows."""
        n = T if gather is None else rows
        assert n <= self.max_logit_rows
        if self._last_parts:
            assert gather is None and n == self._fwd_T
            H.rmsnorm_parts(self.buf_parts_d, self._parts("d", n)[0], n, self.w["model.norm.weight"], self.cfg.rms_norm_eps, n, self.h,
                            res_in=self.buf_res, out_frag=self.buf_lastf)
        else:
            H.rmsnorm(self.buf_h, self.w["model.norm.weight"], self.cfg.rms_norm_eps, n, self.h, res_in=self.buf_res,
                      out_frag=self.buf_lastf, gather=gather)
        if self._ap_ok(n):
            H.gemm_argmax(self.buf_lastf, self.w["lm_head.weight"], self.logits, n, self.V, self.h, self.V, self.ap_val, self.ap_idx,
                          self.ap_stride)
        else:
            self._gemm(self.buf_lastf, self.h, self.w["lm_head.weight"], self.V, self.logits, n, self.V)
        return n

    def _ap_ok(self, n: int) -> bool:
        """compute_logits(n) leaves argmax candidates for its n rows."""
        return self.argmax_fused and n <= self.ap_rows

    def _ap_parts(self, n: int) -> int:
        return H.gemm_argmax_nparts(n, self.V, self.h)

    def has_argmax_parts(self, n: int) -> bool:
        """compute_logits(n) leaves argmax candidates and the vocabulary is not sharded (the fused tails apply)."""
        return self._ap_ok(n) and not self.use_coll

    def argmax_verify(self, B: int, K: int, speculations, preds, accept_len, recovery, packed) -> None:
        """argmax of the B*(K+1) verify rows + greedy accept / reject (utils/verify.py:28-48) in one launch (TP = 1)."""
        assert self.has_argmax_parts(B * (K + 1))
        H.argmax_parts_verify(self.ap_val, self.ap_idx, self._ap_parts(B * (K + 1)), self.ap_stride, speculations, B, K, accept_len, recovery,
                              packed, preds=preds)

    def argmax_advance(self, B: int, next_ids, input_ids, positions, slots, context_lens, block_tables, max_blocks, block_size,
                       spec, K, step) -> None:
        """argmax of the B decode rows + the device-side chain advance (csrc/misc.hip draft_advance) in one launch."""
        assert self.has_argmax_parts(B)
        H.argmax_parts_advance(self.ap_val, self.ap_idx, self._ap_parts(B), self.ap_stride, next_ids, input_ids, positions, slots,
                               context_lens, block_tables, max_blocks, block_size, spec, K, step, B)

    def argmax(self, n: int, out: torch.Tensor, out2: torch.Tensor | None = None, out3: torch.Tensor | None = None,
               out3_stride: int = 0) -> None:
        """Greedy tokens of logits[:n] over the FULL vocabulary (identical on every TP rank).  out3 (optional): a third
        destination written with a row stride (the tree step's [T][K] token table; only with candidates available)."""
        if self._ap_ok(n):          # candidates from the LM-head epilogue: a few thousand per row instead of V logits
            np_ = self._ap_parts(n)
            if not self.use_coll:
                H.argmax_parts(self.ap_val, self.ap_idx, np_, self.ap_stride, n, out, out2, out3, out3_stride)
                return
            assert out3 is None
            R = self.max_logit_rows
            if self.custom_ar is not None:
                idx_l = self.am_pack_l[:R]
                val_l = self.am_pack_l[R:].view(torch.float32)[:R]
                H.argmax_parts(self.ap_val, self.ap_idx, np_, self.ap_stride, n, idx_l, out_val=val_l, idx_offset=self.tp_rank * self.V)
                self.custom_ar.all_gather_words(self.am_pack_l, self.am_pack, self.am_words)
                vals = self.am_pack.view(-1)[R:].view(torch.float32)
                H.argmax_merge(vals, self.am_pack, self.tp_size, n, 2 * self.am_words, out, out2, stride_idx=self.am_words)
                return
            H.argmax_parts(self.ap_val, self.ap_idx, np_, self.ap_stride, n, self.am_idx_l, out_val=self.am_val_l,
                           idx_offset=self.tp_rank * self.V)
            dist.all_gather_into_tensor(self.am_val.view(-1), self.am_val_l, group=self.tp_group)
            dist.all_gather_into_tensor(self.am_idx.view(-1), self.am_idx_l, group=self.tp_group)
            H.argmax_merge(self.am_val, self.am_idx, self.tp_size, n, R, out, out2)
            return
        assert out3 is None
        if not self.use_coll:
            H.argmax_rows(self.logits, self.V, n, self.V, out, out2)
            return
        R = self.max_logit_rows
        if self.custom_ar is not None:
            # local (value, global index) straight into the packed buffer, one one-shot all-gather, merge
            idx_l = self.am_pack_l[:R]
            val_l = self.am_pack_l[R:].view(torch.float32)[:R]
            H.argmax_rows_val(self.logits, self.V, n, self.V, self.tp_rank * self.V, idx_l, val_l)
            self.custom_ar.all_gather_words(self.am_pack_l, self.am_pack, self.am_words)
            # rank r's indices start at word r*am_words, its values R words later: one strided merge
            vals = self.am_pack.view(-1)[R:].view(torch.float32)
            H.argmax_merge(vals, self.am_pack, self.tp_size, n, 2 * self.am_words, out, out2, stride_idx=self.am_words)
            return
        H.argmax_rows_val(self.logits, self.V, n, self.V, self.tp_rank * self.V, self.am_idx_l, self.am_val_l)
        dist.all_gather_into_tensor(self.am_val.view(-1), self.am_val_l, group=self.tp_group)
        dist.all_gather_into_tensor(self.am_idx.view(-1), self.am_idx_l, group=self.tp_group)
        # gathered layout is [tp][R]; rows beyond n are ignored by the merge (T = n, stride R)
        H.argmax_merge(self.am_val, self.am_idx, self.tp_size, n, R, out, out2)

    def full_logits(self, n: int) -> torch.Tensor:
        """[n, V_full] logits on every rank (only needed off the greedy path)."""
        if not self.use_coll:
            return self.logits[:n]
        parts = [torch.empty(n, self.V, dtype=BF16, device=self.device) for _ in range(self.tp_size)]
        dist.all_gather(parts, self.logits[:n].contiguous(), group=self.tp_group)
        return torch.cat(parts, dim=-1)
