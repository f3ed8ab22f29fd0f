"""One-shot full-mesh all-reduce over hipIpc-shared buffers (device side: ssd_amd/csrc/comm.hip).

Used for the small, latency-bound tensor-parallel sums of the decode / verify forwards; larger messages (prefill)
stay on RCCL.  Safety net: the kernel's spins are bounded and raise an error word instead of hanging; before the engine
enables this path every rank validates it in a throw-away helper process (``python -m ssd_amd.utils.custom_ar``),
so that a driver / topology problem (no peer access, IPC refused, a fault) can only kill the helper -- the engine then
simply keeps using RCCL.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import torch
import torch.distributed as dist

from ssd_amd.hip.lib import load_library
from ssd_amd.utils.graphs import capture

SLOT_ELEMS = 1 << 19          # bf16 elements per staging slot (1 MiB); messages above this go to RCCL
GR_MAX_ELEMS = 1 << 16        # messages up to this many bf16 elements travel as data-tagged granules (csrc/comm.hip): [8, 8192] fits
GR_CAP = GR_MAX_ELEMS // 2    # granules (8 bytes: 2 bf16 + epoch tag) per (parity, source rank) inbox region
AR_MAX_RANKS = 8
FLAG_BYTES = 4096
SPIN_BUDGET = 100_000_000     # polls (~1 us each after the first 4096: ~100 s) before a wait gives up and sets the error word


class OneShotAllReduce:
    def __init__(self, group, device: torch.device, granules: bool = True):
        self.lib = load_library()
        self.group, self.device = group, device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # granules for the small messages (the default); False keeps every message on the stage -> flag -> peer-read protocol
        # (the self-test runs both: `python -m ssd_amd.utils.custom_ar` with SSD_AR_PROTO=flag)
        self.granules = granules
        torch.cuda.set_device(device)
        self._own, self._opened = [], []
        # Every rank takes part in every collective of this constructor even if a local step failed, and the outcome is
        # agreed on at the end: either all ranks hold a working object or all of them raise -- never a rank left
        # waiting in a collective for a peer that bailed out.
        mine = None
        try:
            slot = self._alloc(2 * SLOT_ELEMS * 2)
            flags = self._alloc(FLAG_BYTES)
            inbox = self._alloc(2 * AR_MAX_RANKS * GR_CAP * 8)
            mine = (self._export(slot), self._export(flags), self._export(inbox))
        except Exception:
            mine = None
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        ok = all(h is not None for h in handles)
        if ok:
            try:
                slots, flgs, boxes = [], [], []
                for r, (hs, hf, hb) in enumerate(handles):
                    if r == self.rank:
                        slots.append(slot)
                        flgs.append(flags)
                        boxes.append(inbox)
                    else:
                        slots.append(self._open(hs))
                        flgs.append(self._open(hf))
                        boxes.append(self._open(hb))
                self.slots = (C.c_void_p * self.world)(*slots)
                self.flags = (C.c_void_p * self.world)(*flgs)
                self.inboxes = (C.c_void_p * self.world)(*boxes)
                self.counters = torch.zeros(8, dtype=torch.int32, device=device)
                self.err = torch.zeros(1, dtype=torch.int32, device=device)
            except Exception:
                ok = False
        verdict = [None] * self.world
        dist.all_gather_object(verdict, bool(ok), group=group)      # doubles as the barrier after the IPC opens
        if not all(verdict):
            self.close()
            raise RuntimeError("one-shot all-reduce setup failed on at least one rank")

    def _alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        if self.lib.ssd_comm_alloc(C.byref(p), nbytes) != 0:
            raise RuntimeError("ssd_comm_alloc failed")
        self._own.append(p.value)
        return p.value

    def _export(self, ptr: int) -> bytes:
        buf = C.create_string_buffer(64)
        if self.lib.ssd_comm_ipc_export(C.c_void_p(ptr), buf) != 0:
            raise RuntimeError("hipIpcGetMemHandle failed")
        return buf.raw

    def _open(self, handle: bytes) -> int:
        p = C.c_void_p()
        if self.lib.ssd_comm_ipc_open(C.create_string_buffer(handle, 64), C.byref(p)) != 0:
            raise RuntimeError("hipIpcOpenMemHandle failed")
        self._opened.append(p.value)
        return p.value

    def fits(self, t: torch.Tensor) -> bool:
        n = t.numel()
        return t.dtype == torch.bfloat16 and n <= SLOT_ELEMS and n % 4 == 0 and t.is_contiguous() and t.data_ptr() % 8 == 0

    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place sum over the group (bf16, fp32 accumulation in rank order); enqueued on the current stream."""
        if self.granules and t.numel() <= GR_MAX_ELEMS:
            rc = self.lib.ssd_allreduce_gr_bf16(t.data_ptr(), t.data_ptr(), t.numel(), self.rank, self.world, self.inboxes, GR_CAP,
                                                self.counters.data_ptr(), self.err.data_ptr(), SPIN_BUDGET,
                                                torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"ssd_allreduce_gr_bf16 failed with code {rc}")
            return
        rc = self.lib.ssd_allreduce_bf16(t.data_ptr(), t.data_ptr(), t.numel(), self.rank, self.world, self.slots, self.flags,
                                         SLOT_ELEMS, self.counters.data_ptr(), self.err.data_ptr(), SPIN_BUDGET,
                                         torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ssd_allreduce_bf16 failed with code {rc}")

    def fits_rows(self, T: int, H: int) -> bool:
        return T * H <= SLOT_ELEMS and H % 32 == 0 and H <= 16384

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, res_in: torch.Tensor, res_out: torch.Tensor, weight: torch.Tensor,
                               eps: float, T: int, H: int, out_rows=None, out_frag=None) -> None:
        """res_out = bf16(sum_over_ranks(x[:T]) + res_in); out = RMSNorm(that) * weight (csrc/comm.hip): the all-reduce
        after o_proj / down_proj and the add + RMSNorm that follows it, in one launch."""
        if self.granules and T * H <= GR_MAX_ELEMS:
            rc = self.lib.ssd_allreduce_add_rmsnorm_gr_bf16(x.data_ptr(), res_in.data_ptr(), res_out.data_ptr(), weight.data_ptr(), eps,
                                                            0 if out_rows is None else out_rows.data_ptr(),
                                                            0 if out_frag is None else out_frag.data_ptr(), T, H, self.rank, self.world,
                                                            self.inboxes, GR_CAP, self.counters.data_ptr(), self.err.data_ptr(),
                                                            SPIN_BUDGET, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"ssd_allreduce_add_rmsnorm_gr_bf16 failed with code {rc}")
            return
        rc = self.lib.ssd_allreduce_add_rmsnorm_bf16(x.data_ptr(), res_in.data_ptr(), res_out.data_ptr(), weight.data_ptr(), eps,
                                                     0 if out_rows is None else out_rows.data_ptr(),
                                                     0 if out_frag is None else out_frag.data_ptr(), T, H, self.rank, self.world,
                                                     self.slots, self.flags, SLOT_ELEMS, self.counters.data_ptr(),
                                                     self.err.data_ptr(), SPIN_BUDGET, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ssd_allreduce_add_rmsnorm_bf16 failed with code {rc}")

    def all_gather_words(self, src: torch.Tensor, dst: torch.Tensor, n8: int) -> None:
        """dst [world][n8] 8-byte words <- every rank's src[:n8] (device buffers, 8-byte aligned)."""
        rc = self.lib.ssd_allgather_u64(src.data_ptr(), dst.data_ptr(), n8, self.rank, self.world, self.slots, self.flags,
                                        SLOT_ELEMS, self.counters.data_ptr(), self.err.data_ptr(), SPIN_BUDGET,
                                        torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ssd_allgather_u64 failed with code {rc}")

    def failed(self) -> bool:
        return bool(self.err.item())

    def close(self) -> None:
        for p in self._opened:
            self.lib.ssd_comm_ipc_close(C.c_void_p(p))
        for p in self._own:
            self.lib.ssd_comm_free(C.c_void_p(p))
        self._opened, self._own = [], []


# --------------------------------------------------------------------------------------------------
# validation helper: run as its own process group (gloo rendezvous), one helper per rank
# --------------------------------------------------------------------------------------------------
def _rank_order_sum(xs) -> torch.Tensor:
    """fp32 sum in rank order, one rounding to bf16 -- the kernel's arithmetic (torch.sum may use another order)."""
    acc = xs[0].float()
    for x in xs[1:]:
        acc = acc + x.float()
    return acc.to(torch.bfloat16)


def _selftest_main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("SSD_AR_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = OneShotAllReduce(dist.group.WORLD, dev, granules=os.environ.get("SSD_AR_PROTO", "granule") != "flag")      # (test plumbing)
    ok = True
    g = torch.Generator().manual_seed(1234)
    for n in (4, 4096, 7 * 8192, 24 * 2048, SLOT_ELEMS):
        for it in range(6):
            xs = [torch.randn(n, generator=g).to(torch.bfloat16) for _ in range(world)]      # same on every rank
            want = _rank_order_sum(xs)
            t = xs[rank].to(dev)
            ar.all_reduce(t)
            torch.cuda.synchronize()
            if ar.failed() or not torch.equal(t.cpu().view(torch.int16), want.view(torch.int16)):
                print(f"[custom_ar selftest] rank {rank}: mismatch at n={n} it={it} failed={ar.failed()}", flush=True)
                ok = False
                break
    # captured in a hipGraph and replayed (the way the engine uses it): two dependent all-reduces per replay
    if ok:
        n = 7 * 8192
        buf = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        src = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        s = torch.cuda.Stream()

        def body():
            buf.copy_(src)
            ar.all_reduce(buf)
            ar.all_reduce(buf)
        with torch.cuda.stream(s):
            body()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with capture(graph, stream=s):
            body()
        for it in range(20):
            xs = [torch.randn(n, generator=g).to(torch.bfloat16) for _ in range(world)]
            once = _rank_order_sum(xs)
            want = _rank_order_sum([once] * world)
            src.copy_(xs[rank].to(dev))
            graph.replay()
            torch.cuda.synchronize()
            if ar.failed() or not torch.equal(buf.cpu().view(torch.int16), want.view(torch.int16)):
                print(f"[custom_ar selftest] rank {rank}: graph replay mismatch it={it} failed={ar.failed()}", flush=True)
                ok = False
                break
    # the fused all-reduce + residual add + RMSNorm must equal the unfused pair bit for bit (interleaved with plain calls)
    if ok:
        from ssd_amd.hip import ops as H
        for T, Hd in ((1, 2048), (7, 8192), (8, 5120), (20, 1024)):
            xs = [torch.randn(T, Hd, generator=g).to(torch.bfloat16) for _ in range(world)]
            res = torch.randn(T, Hd, generator=g).to(torch.bfloat16).to(dev)
            w = (1 + 0.1 * torch.randn(Hd, generator=g)).to(torch.bfloat16).to(dev)
            x = xs[rank].to(dev)
            ref_h = x.clone()
            ar.all_reduce(ref_h)
            ref_res = torch.zeros_like(res)
            ref_rows = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
            ref_frag = torch.zeros(H.frag_numel(T, Hd), dtype=torch.bfloat16, device=dev)
            H.rmsnorm(ref_h, w, 1e-5, T, Hd, res_in=res, res_out=ref_res, out_rows=ref_rows, out_frag=ref_frag)
            got_res = res.clone()
            rows = torch.zeros_like(ref_rows)
            frag = torch.zeros_like(ref_frag)
            ar.all_reduce_add_rmsnorm(x, got_res, got_res, w, 1e-5, T, Hd, out_rows=rows, out_frag=frag)
            torch.cuda.synchronize()
            same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in ((got_res, ref_res), (rows, ref_rows), (frag, ref_frag)))
            if ar.failed() or not same:
                print(f"[custom_ar selftest] rank {rank}: fused all-reduce+norm mismatch at T={T} H={Hd} failed={ar.failed()}", flush=True)
                ok = False
                break
    # mismatched shapes interleaved (plain: words dealt to the workgroups; fused: whole rows; gather: a handful of
    # words) while one rank is delayed by a long device-side sleep: every workgroup must stay on one global epoch
    if ok:
        from ssd_amd.hip import ops as H
        T, Hd, ng = 7, 4096, 12
        res0 = torch.randn(T, Hd, generator=g).to(torch.bfloat16)
        w = (1 + 0.1 * torch.randn(Hd, generator=g)).to(torch.bfloat16).to(dev)
        for it in range(24):
            xs = [torch.randn(T, Hd, generator=g).to(torch.bfloat16) for _ in range(world)]
            ys = [torch.randn(4 * (it + 1), generator=g).to(torch.bfloat16) for _ in range(world)]
            gs = [torch.randint(0, 1 << 40, (ng,), generator=g) for _ in range(world)]
            if it % world == rank:
                torch.cuda._sleep(20_000_000)           # ~10 ms: this rank enters the sequence late
            y = ys[rank].to(dev)
            ar.all_reduce(y)
            x = xs[rank].to(dev)
            res = res0.to(dev)
            rows = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
            ar.all_reduce_add_rmsnorm(x, res, res, w, 1e-5, T, Hd, out_rows=rows)
            gout = torch.zeros(world, ng, dtype=torch.int64, device=dev)
            ar.all_gather_words(gs[rank].to(dev), gout, ng)
            x2 = xs[(rank + 1) % world].to(dev)
            ar.all_reduce(x2.view(-1))
            torch.cuda.synchronize()
            want_y = _rank_order_sum(ys)
            sum_x = _rank_order_sum(xs)
            want_x2 = _rank_order_sum([xs[(r + 1) % world] for r in range(world)])
            ref_res = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
            ref_rows = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
            H.rmsnorm(sum_x.to(dev), w, 1e-5, T, Hd, res_in=res0.to(dev), res_out=ref_res, out_rows=ref_rows)
            torch.cuda.synchronize()
            good = (torch.equal(y.cpu().view(torch.int16), want_y.view(torch.int16))
                    and torch.equal(res.view(torch.int16), ref_res.view(torch.int16))
                    and torch.equal(rows.view(torch.int16), ref_rows.view(torch.int16))
                    and torch.equal(gout.cpu(), torch.stack(gs))
                    and torch.equal(x2.cpu().view(torch.int16), want_x2.view(torch.int16)))
            if ar.failed() or not good:
                print(f"[custom_ar selftest] rank {rank}: interleaved mismatched-shape sequence broke at it={it} failed={ar.failed()}", flush=True)
                ok = False
                break
    # SSD_AR_STRESS=n (tests): n more rounds of randomly sized plain / fused / gather calls in random order -- message
    # sizes on both sides of the granule / flag protocol boundary, every result compared bit for bit with the rank-order sum
    stress = int(os.environ.get("SSD_AR_STRESS", "0"))
    if ok and stress > 0:
        import random
        from ssd_amd.hip import ops as H
        rnd = random.Random(99)                  # the same sequence on every rank
        for it in range(stress):
            kind = rnd.choice(("plain", "plain", "fused", "gather"))
            if kind == "plain":
                n = 4 * rnd.randint(1, (GR_MAX_ELEMS * 3) // 4)
                xs = [torch.randn(n, generator=g).to(torch.bfloat16) for _ in range(world)]
                t = xs[rank].to(dev)
                ar.all_reduce(t)
                torch.cuda.synchronize()
                good = torch.equal(t.cpu().view(torch.int16), _rank_order_sum(xs).view(torch.int16))
            elif kind == "fused":
                Hd = 32 * rnd.randint(1, 256)
                T = rnd.randint(1, max(1, min(24, (2 * GR_MAX_ELEMS) // Hd)))
                xs = [torch.randn(T, Hd, generator=g).to(torch.bfloat16) for _ in range(world)]
                res0 = torch.randn(T, Hd, generator=g).to(torch.bfloat16)
                w = (1 + 0.1 * torch.randn(Hd, generator=g)).to(torch.bfloat16).to(dev)
                res = res0.to(dev)
                rows = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
                ar.all_reduce_add_rmsnorm(xs[rank].to(dev), res, res, w, 1e-5, T, Hd, out_rows=rows)
                ref_res = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
                ref_rows = torch.zeros(T, Hd, dtype=torch.bfloat16, device=dev)
                H.rmsnorm(_rank_order_sum(xs).to(dev), w, 1e-5, T, Hd, res_in=res0.to(dev), res_out=ref_res, out_rows=ref_rows)
                torch.cuda.synchronize()
                good = torch.equal(res.view(torch.int16), ref_res.view(torch.int16)) and torch.equal(rows.view(torch.int16), ref_rows.view(torch.int16))
            else:
                ng = rnd.randint(1, 64)
                gs = [torch.randint(0, 1 << 40, (ng,), generator=g) for _ in range(world)]
                gout = torch.zeros(world, ng, dtype=torch.int64, device=dev)
                ar.all_gather_words(gs[rank].to(dev), gout, ng)
                torch.cuda.synchronize()
                good = torch.equal(gout.cpu(), torch.stack(gs))
            if ar.failed() or not good:
                print(f"[custom_ar selftest] rank {rank}: stress round {it} ({kind}) mismatch, failed={ar.failed()}", flush=True)
                ok = False
                break
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    ar.close()
    dist.destroy_process_group()
    return 0 if int(flag.item()) == 1 else 1


def validate_in_subprocess(rank: int, world: int, local_rank: int, port: int, timeout: float = 180.0) -> bool:
    """Spawn this rank's validation helper; True iff the whole helper group succeeded."""
    # under torchrun the workers carry TORCHELASTIC_USE_AGENT_STORE=True, which makes env:// rendezvous look for the
    # agent's store on MASTER_PORT instead of creating one: the helpers form their own little group, so drop those
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC") and not k.startswith("TORCH_NCCL")}
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        p = subprocess.Popen([sys.executable, "-m", "ssd_amd.utils.custom_ar"], env=env, stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL)
        try:
            return p.wait(timeout=timeout) == 0
        except subprocess.TimeoutExpired:
            p.kill()
            return False
    except Exception:
        return False


if __name__ == "__main__":
    sys.exit(_selftest_main())
