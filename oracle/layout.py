"""Host-side (torch, CPU) statement of the fragment-major layout of include/ssd_hip.h, used by tests to build
and decode device buffers independently of the HIP conversion kernels."""
from __future__ import annotations

import torch


def rows_to_frag_ref(x: torch.Tensor) -> torch.Tensor:
    """[R, K] -> flat [ceil(R/16)*16*K]: 16x32 tiles, lane = (r & 15) + 16 * ((k & 31) >> 3), 8 elements per lane."""
    R, K = x.shape
    assert K % 32 == 0
    MT = (R + 15) // 16
    xp = torch.zeros(MT * 16, K, dtype=x.dtype)
    xp[:R] = x
    t = xp.view(MT, 16, K // 32, 4, 8)          # [mt, r, kt, q, e]
    return t.permute(0, 2, 3, 1, 4).contiguous().view(-1)   # [mt, kt, q, r, e]


def frag_to_rows_ref(xf: torch.Tensor, R: int, K: int) -> torch.Tensor:
    MT = (R + 15) // 16
    t = xf.view(MT, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous().view(MT * 16, K)
    return t[:R]


def interleave_gate_up_rows(w: torch.Tensor) -> torch.Tensor:
    """[2I, K] (gate rows then up rows) -> rows reordered so 16-row groups alternate gate, up, gate, up ..."""
    I = w.shape[0] // 2
    g = w[:I].view(I // 16, 16, -1)
    u = w[I:].view(I // 16, 16, -1)
    return torch.stack([g, u], dim=1).reshape(2 * I, -1)


def kv_nhd_to_hnd(cache: torch.Tensor) -> torch.Tensor:
    """reference cache [blocks, bs, nkv, hd] -> ssd_amd cache [blocks, nkv, bs, hd]."""
    return cache.permute(0, 2, 1, 3).contiguous()


def kv_hnd_to_nhd(cache: torch.Tensor) -> torch.Tensor:
    return cache.permute(0, 2, 1, 3).contiguous()
