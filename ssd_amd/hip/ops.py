"""Thin torch-tensor front end over the C ABI (include/ssd_hip.h).

PyTorch is used here only as a device-memory carrier: every function passes raw device pointers and the
current HIP stream to libssdhip and returns nothing new (callers own all buffers, so the calls are
hipGraph-capturable).  No function in this module computes anything in torch.
"""
from __future__ import annotations

import torch

from .lib import load_library, SsdHipError

EPI_ROWS, EPI_SILU_FRAG, EPI_ROWS_F32 = 0, 1, 2
PF_EPI_PARTIALS = 2          # ssd_gemm_pf only: leave the split-K partials in the workspace for ssd_rmsnorm_parts
MODE_CAUSAL, MODE_TREE = 0, 1


def _p(t) -> int:
    if t is None:
        return 0
    assert t.is_cuda and t.is_contiguous(), "libssdhip needs contiguous device tensors"
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _check(rc: int, name: str):
    if rc != 0:
        raise SsdHipError(f"{name} failed with code {rc}")


def frag_numel(M: int, K: int) -> int:
    return ((M + 15) // 16) * 16 * K


def rows_to_frag(src: torch.Tensor, dst: torch.Tensor, R: int, K: int, mode: int = 0):
    _check(load_library().ssd_rows_to_frag(_p(src), _p(dst), R, K, mode, _stream()), "ssd_rows_to_frag")


def frag_to_rows(src: torch.Tensor, dst: torch.Tensor, R: int, K: int):
    _check(load_library().ssd_frag_to_rows(_p(src), _p(dst), R, K, _stream()), "ssd_frag_to_rows")


def embedding(ids, table, out_rows, T: int, H: int, vocab_start: int = 0, vocab_count: int | None = None):
    vc = table.shape[0] if vocab_count is None else vocab_count
    _check(load_library().ssd_embedding(_p(ids), _p(table), _p(out_rows), T, H, vocab_start, vc, _stream()), "ssd_embedding")


def rmsnorm(x_rows, weight, eps: float, T: int, H: int, res_in=None, res_out=None, out_rows=None, out_frag=None, gather=None):
    _check(load_library().ssd_rmsnorm(_p(x_rows), _p(res_in), _p(res_out), _p(weight), eps, _p(out_rows), _p(out_frag),
                                      _p(gather), T, H, _stream()), "ssd_rmsnorm")


def head_rmsnorm(x_rows, weight, eps: float, out_rows, T: int, heads: int, hd: int):
    """RMSHeadNorm.forward as a call of its own (the hot path runs it inside rope_store_kv)."""
    _check(load_library().ssd_head_rmsnorm(_p(x_rows), _p(weight), eps, _p(out_rows), T, heads, hd, _stream()), "ssd_head_rmsnorm")


def silu_mul(x_rows, T: int, I: int, out_rows=None, out_frag=None):
    """SiluAndMul.forward as a call of its own (the hot path runs it as the gate_up GEMM's epilogue)."""
    _check(load_library().ssd_silu_mul(_p(x_rows), _p(out_rows), _p(out_frag), T, I, _stream()), "ssd_silu_mul")


def gemm(x_frag, w_frag, y, M: int, N: int, K: int, ldy: int, epilogue: int = EPI_ROWS, bias=None, cfg=None):
    lib = load_library()
    if cfg is None:
        rc = lib.ssd_gemm_wf(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, epilogue, _stream())
    else:
        rc = lib.ssd_gemm_wf_cfg(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, epilogue, cfg[0], cfg[1], _stream())
    _check(rc, "ssd_gemm_wf")


def gemm_splitk(x_frag, w_frag, y, M: int, N: int, K: int, ldy: int, splits: int, waves: int, workspace, counters, bias=None):
    """Skinny GEMM with K split across workgroups (csrc/gemm_sk.hip); workspace: >= (N/16)*splits KiB, counters: zeroed uint32[N/16]."""
    _check(load_library().ssd_gemm_splitk(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, splits, waves, _p(workspace),
                                          _p(counters), _stream()), "ssd_gemm_splitk")


def gemm_parts(x_frag, w_frag, M: int, N: int, K: int, *, parts=None, splits: int = 1, waves: int = 8, y=None, ldy: int = 0, bias=None):
    """Latency-optimal small-matrix GEMM (csrc/gemm_sk.hip gemm_sp_kernel): K split over `splits` workgroups per row group,
    fp32 partial slabs [splits][M][N] into `parts` (summed by the consumer), or bf16 rows into y when splits == 1."""
    _check(load_library().ssd_gemm_parts(_p(x_frag), _p(w_frag), _p(bias), _p(y), _p(parts), M, N, K, ldy, splits, waves,
                                         _stream()), "ssd_gemm_parts")


def rmsnorm_pair(x0_rows, weight0, x1_rows, weight1, eps: float, T: int, H: int, out_frag):
    """cat([norm(x0) * w0, norm(x1) * w1], -1) as one fragment-major [T][2H] activation (EAGLE-3 draft layer input)."""
    _check(load_library().ssd_rmsnorm_pair(_p(x0_rows), _p(weight0), _p(x1_rows), _p(weight1), eps, _p(out_frag), T, H, _stream()),
           "ssd_rmsnorm_pair")


def rmsnorm_parts(parts, splits: int, slab_rows: int, weight, eps: float, T: int, H: int, res_in=None, res_out=None, out_rows=None,
                  out_frag=None):
    _check(load_library().ssd_rmsnorm_parts(_p(parts), splits, slab_rows, _p(res_in), _p(res_out), _p(weight), eps, _p(out_rows),
                                            _p(out_frag), T, H, _stream()), "ssd_rmsnorm_parts")


def gemm_pf_workspace_bytes(M: int, N: int, K: int) -> int:
    import ctypes
    out = ctypes.c_int64(0)
    _check(load_library().ssd_gemm_pf_workspace_bytes(M, N, K, ctypes.addressof(out)), "ssd_gemm_pf_workspace_bytes")
    return out.value


def gemm_pf(x_frag, w_frag, y, M: int, N: int, K: int, ldy: int, workspace, epilogue: int = EPI_ROWS, bias=None, splits: int = 0,
            nt: int = 0):
    """Prefill-chunk GEMM (16 < M <= 128); workspace: a float32 device tensor of >= gemm_pf_workspace_bytes."""
    wb = workspace.numel() * workspace.element_size()
    if nt:
        _check(load_library().ssd_gemm_pf_cfg(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, epilogue, _p(workspace),
                                              wb, nt, splits, _stream()), "ssd_gemm_pf_cfg")
    else:
        _check(load_library().ssd_gemm_pf(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, epilogue, _p(workspace),
                                          wb, splits, _stream()), "ssd_gemm_pf")


def rope_store_kv(qkv_rows, positions, cos_sin, slot_mapping, q_out, k_cache, v_cache, T, nh, nkv, hd, block_size,
                  q_norm_w=None, k_norm_w=None, eps: float = 0.0, qkv_perm: int = 0):
    _check(load_library().ssd_rope_store_kv(_p(qkv_rows), _p(positions), _p(cos_sin), _p(slot_mapping), _p(q_out),
                                            _p(k_cache), _p(v_cache), _p(q_norm_w), _p(k_norm_w), eps, T, nh, nkv, hd,
                                            block_size, qkv_perm, _stream()), "ssd_rope_store_kv")


def rope_store_kv_parts(parts, splits: int, positions, cos_sin, slot_mapping, q_out, k_cache, v_cache, T, nh, nkv, hd, block_size,
                        q_norm_w=None, k_norm_w=None, eps: float = 0.0, qkv_perm: int = 0):
    """rope_store_kv over the prefill GEMM's fp32 split-K slabs [splits][T][(nh + 2 nkv) * hd] (gemm_pf, PF_EPI_PARTIALS)."""
    _check(load_library().ssd_rope_store_kv_parts(_p(parts), splits, _p(positions), _p(cos_sin), _p(slot_mapping), _p(q_out),
                                                  _p(k_cache), _p(v_cache), _p(q_norm_w), _p(k_norm_w), eps, T, nh, nkv, hd,
                                                  block_size, qkv_perm, _stream()), "ssd_rope_store_kv_parts")


def rows_to_frag_qkv(src, dst, nh: int, nkv: int, hd: int, K: int):
    _check(load_library().ssd_rows_to_frag_qkv(_p(src), _p(dst), nh, nkv, hd, K, _stream()), "ssd_rows_to_frag_qkv")


FEPI_ROWS, FEPI_SILU_FRAG, FEPI_QKV_ROPE = 0, 1, 3


def gemm_fused(w_frag, M: int, N: int, K: int, epilogue: int, *, x_frag=None, h_rows=None, h_parts=None, splits: int = 0,
               res_in=None, res_out=None,
               norm_w=None, eps: float = 0.0, bias=None, y=None, ldy: int = 0, positions=None, cos_sin=None, slots=None,
               q_out=None, k_cache=None, v_cache=None, nh: int = 0, nkv: int = 0, hd: int = 0, block_size: int = 0,
               nt: int = 0, waves: int = 0):
    if h_parts is not None:     # norm prologue fed by the producer's fp32 split-K slabs [splits][M][K]
        _check(load_library().ssd_gemm_fused_parts(_p(h_parts), splits, _p(res_in), _p(res_out), _p(norm_w), eps, _p(w_frag),
                                                   _p(bias), M, N, K, epilogue, _p(y), ldy, _p(positions), _p(cos_sin), _p(slots),
                                                   _p(q_out), _p(k_cache), _p(v_cache), nh, nkv, hd, block_size, nt, waves,
                                                   _stream()), "ssd_gemm_fused_parts")
        return
    _check(load_library().ssd_gemm_fused(_p(x_frag), _p(h_rows), _p(res_in), _p(res_out), _p(norm_w), eps, _p(w_frag),
                                         _p(bias), M, N, K, epilogue, _p(y), ldy, _p(positions), _p(cos_sin), _p(slots),
                                         _p(q_out), _p(k_cache), _p(v_cache), nh, nkv, hd, block_size, nt, waves,
                                         _stream()), "ssd_gemm_fused")


def attn_paged(q_rows, k_cache, v_cache, block_tables, max_blocks, context_lens, B, T, max_q, nh, nkv, hd, block_size,
               scale, cu_q=None, q_per_seq=0, mode=MODE_CAUSAL, tree_K=0, tree_mq=0, tree_step=0, tree_F=1, tree_jidx=None,
               splits=1, flags=0, ws_o=None, ws_ml=None, out_rows=None, out_frag=None, waves=1):
    flags = (flags & 0xff) | ((waves & 0xf) << 8)
    _check(load_library().ssd_attn_paged(_p(q_rows), _p(k_cache), _p(v_cache), _p(block_tables), max_blocks,
                                         _p(context_lens), _p(cu_q), q_per_seq, B, T, max_q, nh, nkv, hd, block_size,
                                         scale, mode, tree_K, tree_mq, tree_step, tree_F, _p(tree_jidx), splits, flags,
                                         _p(ws_o), _p(ws_ml), _p(out_rows), _p(out_frag), _stream()), "ssd_attn_paged")


def attn_oproj_parts(q_rows, k_cache, v_cache, block_tables, max_blocks: int, context_lens, T: int, nh: int, nkv: int, hd: int,
                     block_size: int, scale: float, w_o_frag, N: int, parts):
    """One sequence's decode / glue attention fused with o_proj: fp32 slabs [nkv][T][N] (csrc/attention.hip OPROJ variant)."""
    _check(load_library().ssd_attn_oproj_parts(_p(q_rows), _p(k_cache), _p(v_cache), _p(block_tables), max_blocks, _p(context_lens), T, nh,
                                               nkv, hd, block_size, scale, _p(w_o_frag), N, _p(parts), _stream()), "ssd_attn_oproj_parts")


def gemm_argmax_nparts(M: int, N: int, K: int) -> int:
    """Candidates per token row that gemm_argmax writes for this shape (= its workgroups); host-side query."""
    n = load_library().ssd_gemm_wf_argmax_parts(M, N, K)
    if n <= 0:
        _check(n if n < 0 else -1, "ssd_gemm_wf_argmax_parts")
    return n


def gemm_argmax(x_frag, w_frag, y, M: int, N: int, K: int, ldy: int, part_val, part_idx, part_stride: int, bias=None):
    """LM head: logits rows + per-workgroup argmax candidates [M][part_stride] (csrc/gemm.hip EPI_ROWS_ARGMAX)."""
    _check(load_library().ssd_gemm_wf_argmax(_p(x_frag), _p(w_frag), _p(bias), _p(y), M, N, K, ldy, _p(part_val), _p(part_idx),
                                             part_stride, _stream()), "ssd_gemm_wf_argmax")


def chain_segment_ok(h: int, qn: int, I: int, qkv_n: int, nh: int, nkv: int, hd: int) -> bool:
    return load_library().ssd_chain_segment_ok(h, qn, I, qkv_n, nh, nkv, hd) == 0


def chain_granule_bytes(h: int, I: int) -> int:
    return load_library().ssd_chain_granule_bytes(h, I)


def chain_tick(gen):
    """Bumps the forward generation the chain segments tag their hand-offs with (once per forward, before the first segment)."""
    _check(load_library().ssd_chain_tick(_p(gen), _stream()), "ssd_chain_tick")


def chain_segment(a_frag, res_in, res_out, w_o, w_gu, w_d, ln_post, eps: float, h: int, qn: int, I: int, qkv_n: int, nh: int, nkv: int,
                  hd: int, block_size: int, layer: int, granules, gen, err, *, h_out=None, w_qkv_next=None, ln_next=None, positions=None,
                  cos_sin=None, slots=None, q_out=None, k_cache=None, v_cache=None):
    """One single-token decoder layer between two attention launches as ONE launch (csrc/chain.hip): o_proj, residual add + RMSNorm,
    gate_up + SiLU * mul, down_proj, residual add, and either the next layer's norm + QKV + RoPE + KV store (w_qkv_next given) or the
    last layer's hand-over to the final norm (h_out given)."""
    _check(load_library().ssd_chain_segment(_p(a_frag), _p(res_in), _p(res_out), _p(h_out), _p(w_o), _p(w_gu), _p(w_d), _p(w_qkv_next),
                                            _p(ln_post), _p(ln_next), eps, _p(positions), _p(cos_sin), _p(slots), _p(q_out), _p(k_cache),
                                            _p(v_cache), h, qn, I, qkv_n, nh, nkv, hd, block_size, layer, _p(granules), _p(gen), _p(err),
                                            _stream()), "ssd_chain_segment")


def attn_prefill_varlen(q_rows, k_cache, v_cache, block_tables, max_blocks, context_lens, cu_q, B, T, max_q, nh, nkv, hd, block_size, scale,
                        out_rows=None, out_frag=None):
    _check(load_library().ssd_attn_prefill_varlen(_p(q_rows), _p(k_cache), _p(v_cache), _p(block_tables), max_blocks, _p(context_lens), _p(cu_q),
                                                  B, T, max_q, nh, nkv, hd, block_size, scale, _p(out_rows), _p(out_frag), _stream()),
           "ssd_attn_prefill_varlen")


def attn_tree(q_rows, k_cache, v_cache, block_tables, max_blocks, context_lens, B, tree_K, tree_mq, tree_step, tree_F, nh, nkv, hd, block_size,
              scale, tree_jidx=None, out_rows=None, out_frag=None):
    _check(load_library().ssd_attn_tree(_p(q_rows), _p(k_cache), _p(v_cache), _p(block_tables), max_blocks, _p(context_lens), B, tree_K, tree_mq,
                                        tree_step, tree_F, _p(tree_jidx), nh, nkv, hd, block_size, scale, _p(out_rows), _p(out_frag), _stream()),
           "ssd_attn_tree")


class CGraph:
    """hipGraph capture through the C ABI alone (ssd_graph_begin / _end / _launch / _destroy): what a host without torch would use; torch hosts
    keep torch.cuda.CUDAGraph.  `with CGraph(stream) as g: <libssdhip calls on that stream>`, then g.launch()."""

    def __init__(self, stream: "torch.cuda.Stream"):
        self.stream, self.exec = stream, None

    def __enter__(self):
        _check(load_library().ssd_graph_begin(self.stream.cuda_stream), "ssd_graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        import ctypes as C
        out = C.c_void_p()
        rc = load_library().ssd_graph_end(self.stream.cuda_stream, C.byref(out))
        if et is None:
            _check(rc, "ssd_graph_end")
            self.exec = out.value
        elif rc == 0 and out.value:          # the body raised: the capture still ended in an executable graph nobody will launch
            load_library().ssd_graph_destroy(out.value)
        return False

    def launch(self):
        _check(load_library().ssd_graph_launch(self.exec, self.stream.cuda_stream), "ssd_graph_launch")

    def destroy(self):
        if self.exec:
            _check(load_library().ssd_graph_destroy(self.exec), "ssd_graph_destroy")
            self.exec = None


def tree_segment_ok(M: int, h: int, qn: int, I: int, qkv_n: int, nh: int, nkv: int, hd: int) -> bool:
    return load_library().ssd_tree_segment_ok(M, h, qn, I, qkv_n, nh, nkv, hd) == 0


def tree_segment_workspace_bytes(h: int, I: int) -> int:
    return load_library().ssd_tree_segment_workspace_bytes(h, I)


def tree_segment(a_frag, res_in, res_out, w_o, w_gu, w_d, ln_post, eps: float, M: int, h: int, qn: int, I: int, qkv_n: int, nh: int,
                 nkv: int, hd: int, block_size: int, layer: int, workspace, gen, err, *, h_out=None, w_qkv_next=None, ln_next=None,
                 positions=None, cos_sin=None, slots=None, q_out=None, k_cache=None, v_cache=None, qkv_rows_next=None):
    """chain_segment for M token rows (csrc/tree_segment.hip): the tree-decode step / the glue decode of the async draft.
    qkv_rows_next: raw QKV rows of the next layer instead of the RoPE + KV-store epilogue (q / k norm models)."""
    _check(load_library().ssd_tree_segment(_p(a_frag), _p(res_in), _p(res_out), _p(h_out), _p(w_o), _p(w_gu), _p(w_d), _p(w_qkv_next),
                                           _p(ln_post), _p(ln_next), eps, _p(positions), _p(cos_sin), _p(slots), _p(q_out), _p(k_cache),
                                           _p(v_cache), _p(qkv_rows_next), M, h, qn, I, qkv_n, nh, nkv, hd, block_size, layer, _p(workspace), _p(gen), _p(err),
                                           _stream()), "ssd_tree_segment")


def selftest_bf16_cvt(counts2):
    _check(load_library().ssd_selftest_bf16_cvt(_p(counts2), _stream()), "ssd_selftest_bf16_cvt")


def argmax_parts(part_val, part_idx, nparts: int, part_stride: int, T: int, out=None, out2=None, out3=None, out3_stride: int = 0,
                 out_val=None, idx_offset: int = 0):
    _check(load_library().ssd_argmax_parts(_p(part_val), _p(part_idx), nparts, part_stride, T, idx_offset, _p(out), _p(out2), _p(out3),
                                           out3_stride, _p(out_val), _stream()), "ssd_argmax_parts")


def argmax_parts_verify(part_val, part_idx, nparts: int, part_stride: int, speculations, B: int, K: int, accept_len, recovery,
                        packed=None, preds=None):
    _check(load_library().ssd_argmax_parts_verify(_p(part_val), _p(part_idx), nparts, part_stride, _p(speculations), B, K, _p(preds),
                                                  _p(accept_len), _p(recovery), _p(packed), _stream()), "ssd_argmax_parts_verify")


def argmax_parts_advance(part_val, part_idx, nparts: int, part_stride: int, next_ids, input_ids, positions, slots, context_lens,
                         block_tables, max_blocks, block_size, spec, K, step, B):
    _check(load_library().ssd_argmax_parts_advance(_p(part_val), _p(part_idx), nparts, part_stride, _p(next_ids), _p(input_ids),
                                                   _p(positions), _p(slots), _p(context_lens), _p(block_tables), max_blocks, block_size,
                                                   _p(spec), K, _p(step), B, _stream()), "ssd_argmax_parts_advance")


def argmax_rows(logits, ld: int, T: int, V: int, out, out2=None):
    _check(load_library().ssd_argmax_rows(_p(logits), ld, T, V, _p(out), _p(out2), _stream()), "ssd_argmax_rows")


def argmax_rows_val(logits, ld: int, T: int, V: int, idx_offset: int, out_idx, out_val):
    _check(load_library().ssd_argmax_rows_val(_p(logits), ld, T, V, idx_offset, _p(out_idx), _p(out_val), _stream()),
           "ssd_argmax_rows_val")


def argmax_merge(vals, idxs, tp: int, T: int, stride: int, out, out2=None, stride_idx: int | None = None):
    _check(load_library().ssd_argmax_merge(_p(vals), _p(idxs), tp, T, stride, stride if stride_idx is None else stride_idx,
                                           _p(out), _p(out2), _stream()), "ssd_argmax_merge")


def verify_greedy(preds, speculations, B: int, K: int, accept_len, recovery, packed=None):
    _check(load_library().ssd_verify_greedy(_p(preds), _p(speculations), B, K, _p(accept_len), _p(recovery), _p(packed),
                                            _stream()), "ssd_verify_greedy")


def fork_topf(logits, ld: int, V: int, returned, counts, offsets, B: int, K: int, mq: int, out):
    _check(load_library().ssd_fork_topf(_p(logits), ld, V, _p(returned), _p(counts), _p(offsets), B, K, mq, _p(out),
                                        _stream()), "ssd_fork_topf")


def fork_topf_workspace_bytes(V: int, B: int, K: int) -> int:
    """Workspace of ssd_fork_topf_split; 0 when the split form does not take this vocabulary (the caller uses fork_topf)."""
    n = load_library().ssd_fork_topf_workspace_bytes(V, B, K)
    return max(n, 0)


def fork_topf_split(logits, ld: int, V: int, returned, counts, offsets, B: int, K: int, mq: int, workspace, out):
    """ssd_fork_topf spread over the chip (per-slice candidates + merge): bit-equal, ~10x shorter on the draft's critical path."""
    _check(load_library().ssd_fork_topf_split(_p(logits), ld, V, _p(returned), _p(counts), _p(offsets), B, K, mq, _p(workspace), _p(out),
                                              _stream()), "ssd_fork_topf_split")


def cache_lookup(req_keys, cache_seq, cache_j, cache_forks, B: int, Bc: int, W: int, out_idx):
    _check(load_library().ssd_cache_lookup(_p(req_keys), _p(cache_seq), _p(cache_j), _p(cache_forks), B, Bc, W, _p(out_idx), _stream()),
           "ssd_cache_lookup")


def draft_advance(next_ids, input_ids, positions, slots, context_lens, block_tables, max_blocks, block_size, spec, K, step, B):
    _check(load_library().ssd_draft_advance(_p(next_ids), _p(input_ids), _p(positions), _p(slots), _p(context_lens),
                                            _p(block_tables), max_blocks, block_size, _p(spec), K, _p(step), B, _stream()),
           "ssd_draft_advance")


def topk_rows(logits, ld: int, T: int, V: int, k: int, out_idx):
    _check(load_library().ssd_topk_rows(_p(logits), ld, T, V, k, _p(out_idx), _stream()), "ssd_topk_rows")


def sample_rows(logits, ld: int, T: int, V: int, temps, rows_per_temp: int, rng_state, salt: int, out, out2=None,
                boost_idx=None, boost_k: int = 0, boost_x: float = 1.0):
    _check(load_library().ssd_sample_rows(_p(logits), ld, T, V, _p(temps), rows_per_temp, _p(rng_state), salt, _p(out), _p(out2),
                                          _p(boost_idx), boost_k, boost_x, _stream()), "ssd_sample_rows")


def rng_advance(rng_state):
    _check(load_library().ssd_rng_advance(_p(rng_state), _stream()), "ssd_rng_advance")


def row_lse(logits, ld: int, T: int, V: int, temps, rows_per_temp: int, lse, boost_idx=None, boost_k: int = 0, boost_x: float = 1.0):
    _check(load_library().ssd_row_lse(_p(logits), ld, T, V, _p(temps), rows_per_temp, _p(lse), _p(boost_idx), boost_k, boost_x,
                                      _stream()), "ssd_row_lse")


def verify_ratio(logits_p, ld_p: int, logits_q, ld_q: int, V: int, B: int, K: int, spec, preds_p, lse_p, lse_q, temps_t, temps_q,
                 ratio_rows, rng_state, salt: int, accept_len, recovery, packed=None, accept_prob=None, boost_idx_q=None,
                 boost_k: int = 0, boost_x: float = 1.0):
    _check(load_library().ssd_verify_ratio(_p(logits_p), ld_p, _p(logits_q), ld_q, V, B, K, _p(spec), _p(preds_p), _p(lse_p),
                                           _p(lse_q), _p(temps_t), _p(temps_q), _p(ratio_rows), _p(rng_state), salt,
                                           _p(accept_len), _p(recovery), _p(packed), _p(accept_prob), _p(boost_idx_q), boost_k,
                                           boost_x, _stream()), "ssd_verify_ratio")


def store_step_rows(src, src_ld: int, dst, B: int, V: int, K: int, step):
    _check(load_library().ssd_store_step_rows(_p(src), src_ld, _p(dst), B, V, K, _p(step), _stream()), "ssd_store_step_rows")


