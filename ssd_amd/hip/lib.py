"""ctypes binding of libssdhip.so (the C ABI declared in include/ssd_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this raises.
The library is built in-tree (``ssd_amd/_lib/libssdhip.so``) by ``ssd_amd/csrc/Makefile`` so it travels with
the repo snapshot to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_LIB = None

c_void_p, c_int, c_long, c_float, c_int64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_int64

ABI_VERSION = 2       # include/ssd_hip.h SSD_HIP_ABI_VERSION (tests/test_abi.py compares the two)

# name -> argtypes, exactly include/ssd_hip.h (+ include/ssd_hip_tune.h)
SIGNATURES = {
    "ssd_abi_version": [],
    "ssd_rows_to_frag": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "ssd_frag_to_rows": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "ssd_rows_to_frag_qkv": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_gemm_fused": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int,
                       c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                       c_int, c_int, c_int, c_int, c_void_p],
    "ssd_embedding": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_long, c_void_p],
    "ssd_rmsnorm": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "ssd_gemm_wf": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_gemm_wf_cfg": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_gemm_splitk": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ssd_gemm_parts": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_gemm_fused_parts": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_void_p],
    "ssd_tune_deep": [c_int],
    "ssd_head_rmsnorm": [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p],
    "ssd_silu_mul": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "ssd_rmsnorm_pair": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    "ssd_rmsnorm_parts": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "ssd_gemm_pf_workspace_bytes": [c_int, c_int, c_int, c_void_p],
    "ssd_gemm_pf": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_void_p],
    "ssd_gemm_pf_cfg": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int, c_void_p],
    "ssd_rope_store_kv": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_rope_store_kv_parts": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ssd_attn_paged": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                       c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ssd_attn_prefill_varlen": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_float, c_void_p, c_void_p, c_void_p],
    "ssd_attn_tree": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                      c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "ssd_graph_begin": [c_void_p],
    "ssd_graph_end": [c_void_p, C.POINTER(c_void_p)],
    "ssd_graph_launch": [c_void_p, c_void_p],
    "ssd_graph_destroy": [c_void_p],
    "ssd_attn_oproj_parts": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                             c_void_p, c_int, c_void_p, c_void_p],
    "ssd_gemm_wf_argmax_parts": [c_int, c_int, c_int],
    "ssd_chain_tick": [c_void_p, c_void_p],
    "ssd_chain_granule_bytes": [c_int, c_int],
    "ssd_chain_segment_ok": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ssd_chain_segment": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ssd_tree_segment_workspace_bytes": [c_int, c_int],
    "ssd_tree_segment_ok": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ssd_tree_segment": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ssd_selftest_bf16_cvt": [c_void_p, c_void_p],
    "ssd_gemm_wf_argmax": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "ssd_argmax_parts": [c_void_p, c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p],
    "ssd_argmax_parts_verify": [c_void_p, c_void_p, c_int, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ssd_argmax_parts_advance": [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "ssd_argmax_rows": [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ssd_argmax_rows_val": [c_void_p, c_long, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p],
    "ssd_argmax_merge": [c_void_p, c_void_p, c_int, c_int, c_long, c_long, c_void_p, c_void_p, c_void_p],
    "ssd_verify_greedy": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ssd_fork_topf": [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ssd_fork_topf_workspace_bytes": [c_int, c_int, c_int],
    "ssd_fork_topf_split": [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ssd_comm_alloc": [C.POINTER(c_void_p), c_long],
    "ssd_comm_free": [c_void_p],
    "ssd_comm_ipc_export": [c_void_p, c_void_p],
    "ssd_comm_ipc_open": [c_void_p, C.POINTER(c_void_p)],
    "ssd_comm_ipc_close": [c_void_p],
    "ssd_allreduce_bf16": [c_void_p, c_void_p, c_long, c_int, c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), c_long,
                           c_void_p, c_void_p, c_long, c_void_p],
    "ssd_allgather_u64": [c_void_p, c_void_p, c_long, c_int, c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), c_long,
                          c_void_p, c_void_p, c_long, c_void_p],
    "ssd_allreduce_add_rmsnorm_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       C.POINTER(c_void_p), C.POINTER(c_void_p), c_long, c_void_p, c_void_p, c_long, c_void_p],
    "ssd_allreduce_gr_bf16": [c_void_p, c_void_p, c_long, c_int, c_int, C.POINTER(c_void_p), c_long, c_void_p, c_void_p, c_long, c_void_p],
    "ssd_allreduce_add_rmsnorm_gr_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          C.POINTER(c_void_p), c_long, c_void_p, c_void_p, c_long, c_void_p],
    "ssd_topk_rows": [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p],
    "ssd_sample_rows": [c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p, C.c_uint, c_void_p, c_void_p, c_void_p, c_int,
                        c_float, c_void_p],
    "ssd_rng_advance": [c_void_p, c_void_p],
    "ssd_row_lse": [c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p],
    "ssd_verify_ratio": [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_void_p, c_void_p, c_void_p, C.c_uint, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_int, c_float, c_void_p],
    "ssd_store_step_rows": [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ssd_cache_lookup": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ssd_draft_advance": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                          c_void_p, c_int, c_void_p],
}


def lib_path() -> str:
    return os.path.join(_PKG, "_lib", "libssdhip.so")


def build_library(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_PKG, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j8"])
    return lib_path()


class SsdHipError(RuntimeError):
    pass


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SsdHipError(f"{path} not found: build it with `make -C ssd_amd/csrc` (or __graft_entry__.build()); "
                          "there is no CPU fallback for the hot path")
    lib = C.CDLL(path)
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SsdHipError(f"libssdhip.so does not export {name}") from e
        fn.argtypes = args
        fn.restype = c_int
    if lib.ssd_abi_version() != ABI_VERSION:
        raise SsdHipError("libssdhip.so ABI version mismatch")
    _LIB = lib
    return lib
