"""In-kernel timeline of the prefill-chunk GEMM (csrc/gemm_pf.hip) at M = 128 for the four 70B layer matrices, default launch shapes
(profiling build of the library: `make -C ssd_amd/csrc trace`): where a workgroup's time goes -- ramp to the first barrier, the k loop,
the store tail -- and how far the workgroups of one launch are apart.    python profiles/ktrace_pf_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd.hip.lib as L  # noqa: E402

L.lib_path = lambda: os.path.join(ROOT, "ssd_amd", "_lib", "libssdhip_trace.so")
from ssd_amd.hip import ops as H  # noqa: E402

MARKS, MAXWG, SLOTS = 12, 4096, 32
NAMES = ["entry", "prologue loads issued", "first barrier (x staged)", "k loop done", "stores issued"]


@torch.inference_mode()
def main():
    lib = L.load_library()
    buf = torch.zeros(SLOTS * MAXWG * MARKS, dtype=torch.int64, device="cuda")
    fn = lib.ssd_ktrace_set_gemm_pf
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    assert fn(buf.data_ptr()) == 0
    BF = torch.bfloat16
    M = 128
    shapes = [("70b qkv", 10240, 8192, H.EPI_ROWS), ("70b o", 8192, 8192, H.PF_EPI_PARTIALS), ("70b gate_up", 57344, 8192, H.EPI_SILU_FRAG),
              ("70b down", 8192, 28672, H.PF_EPI_PARTIALS)]
    for name, N, K, epi in shapes:
        ws_ = [torch.randn(N * K // 16, device="cuda").to(BF).repeat(16) for _ in range(3)]
        xf = (torch.randn(H.frag_numel(M, K), device="cuda") * 0.05).to(BF)
        y = torch.zeros(M * N, device="cuda", dtype=BF)
        wsb = torch.zeros(16 * M * N + 64, dtype=torch.float32, device="cuda")
        for i in range(3):
            H.gemm_pf(xf, ws_[i], y, M, N, K, N, wsb, epilogue=epi)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.gemm_pf(xf, ws_[0], y, M, N, K, N, wsb, epilogue=epi)
        e1.record()
        torch.cuda.synchronize()
        t = buf.cpu().view(SLOTS, MAXWG, MARKS).numpy()
        print(f"\n=== {name} [{N} x {K}] M = {M}: {N * K * 2 / 1e6:.0f} MB, event time {e0.elapsed_time(e1) * 1e3:.1f} us (first kernel of the call; an epilogue kernel may follow) ===")
        for s in range(16, 19):
            live = t[s][t[s][:, 0] != 0].astype("int64")
            if not len(live):
                continue
            base = live[:, 0].min()
            print(f"[slot {s}] {len(live)} workgroups; entry median {(np.median(live[:, 0]) - base) / 100:.2f} last {(live[:, 0].max() - base) / 100:.2f}; "
                  f"end median {(np.median(live[:, 4]) - base) / 100:.2f} last {(live[:, 4].max() - base) / 100:.2f}")
            for i in range(1, 5):
                col = live[:, i]
                print(f"      {NAMES[i]:28s} median {(np.median(col) - base) / 100:7.2f}  max {(col.max() - base) / 100:7.2f}   in-workgroup median "
                      f"{np.median(col - live[:, 0]) / 100:7.2f}   (phase: {np.median(col - live[:, i - 1]) / 100:6.2f})")
        del ws_


main()
