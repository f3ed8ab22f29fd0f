"""npz (de)serialisation of golden vectors; bf16 tensors travel as int16 views (numpy has no bf16)."""
from __future__ import annotations

import numpy as np
import torch


def save_npz(path: str, tensors: dict) -> None:
    out = {}
    for k, v in tensors.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.contiguous().view(torch.int16).numpy()
            else:
                out[k] = v.contiguous().numpy()
        else:
            out[k] = np.asarray(v)
    np.savez_compressed(path, **out)


def load_npz(path: str) -> dict:
    z = np.load(path, allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(a.copy()).view(torch.bfloat16)
        else:
            out[k] = torch.from_numpy(a.copy()) if a.ndim > 0 else torch.tensor(a.item())
    return out
