"""Data-parallel plumbing for the fine-tune step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

Only the fine-tune step has an exchange; inference and independent students are replicas with
no collective.  The HIP engine calls back into ``ArenaAllReduce`` at every point where full-batch semantics
need cross-rank sums:
  * per BN layer, forward : (sum(z-c), sum((z-c)^2))   2*C float64   -> SyncBN statistics over the global batch
  * loss                  : (CE sum, valid-pixel count) 2 float64    -> the mean's denominator is global
  * per BN layer, backward: (sum dy, sum dy*xhat)       2*C float64
  * gradients             : the flat trainable arena    2 113 043 float32 (8.45 MB), once, before Adam
Every rank then applies the identical Adam update to identical weights, as the reference's single-process step
would on the concatenated batch.  The 8.45 MB gradient message is one flat buffer by construction (the engine
trains in a flat arena), so there is exactly one bandwidth-bound collective per step; the BN messages (<= 15 KB)
are latency-bound and sequentially dependent (108 per step) — see DESIGN.md for what that costs on xGMI.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch

from . import hip


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun's contract).

    Returns (rank, world_size, local_rank).  With WORLD_SIZE unset or 1 no process group is created."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [begin, end) of n_items for this rank (earlier ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class RcclComm:
    """The library's own RCCL communicator (``ams_comm_*`` in include/ams_hip.h): the production transport of the data-parallel
    step.  Every cross-rank sum of ``StudentEngine.train_step(..., comm=...)`` is one ``ncclAllReduce`` on the launch stream,
    enqueued by the engine in launch order — no Python, no host synchronisation between the 110 collectives of a step.

    Construct on every rank after ``torch.cuda.set_device(local_rank)``; the 128-byte unique id travels from rank 0 through the
    default ``torch.distributed`` group (any backend).  ``world == 1`` gives a no-op communicator."""

    def __init__(self, rank: int, world: int, device: Optional[torch.device] = None, real_single_rank: bool = False):
        import ctypes as C
        self.lib = hip.lib()
        self.rank, self.world = int(rank), int(world)
        ident = (C.c_uint8 * 128)()
        have_id = self.world > 1 or real_single_rank
        if self.world == 1 and real_single_rank:      # a genuine one-rank RCCL communicator (tests): every call goes through librccl
            hip.check(self.lib.ams_comm_unique_id(ident, 128), "ams_comm_unique_id")
        if self.world > 1:
            import torch.distributed as dist
            box = [None]
            if self.rank == 0:
                hip.check(self.lib.ams_comm_unique_id(ident, 128), "ams_comm_unique_id")
                box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            C.memmove(ident, box[0], 128)
        handle = C.c_void_p()
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            hip.check(self.lib.ams_comm_create(ident if have_id else None, 128, self.rank, self.world, C.byref(handle)), "ams_comm_create")
        self._h = handle

    def stats(self) -> Tuple[int, int]:
        """(collectives issued, bytes reduced) since construction."""
        import ctypes as C
        calls, nbytes = C.c_int64(), C.c_int64()
        hip.check(self.lib.ams_comm_stats(self._h, None, None, C.byref(calls), C.byref(nbytes)), "ams_comm_stats")
        return int(calls.value), int(nbytes.value)

    def set_timing(self, on: bool) -> None:
        """Bracket every collective with a HIP event pair (diagnostic steps only: the pairs cost stream time)."""
        hip.check(self.lib.ams_comm_set_timing(self._h, int(bool(on))), "ams_comm_set_timing")

    def timing(self) -> Tuple[float, float, int]:
        """(summed ms, longest ms, count) of the collectives' spans since ``set_timing``; synchronises the device."""
        import ctypes as C
        tot, mx, n = C.c_double(), C.c_double(), C.c_int64()
        hip.check(self.lib.ams_comm_timing_read(self._h, C.byref(tot), C.byref(mx), C.byref(n)), "ams_comm_timing_read")
        return float(tot.value), float(mx.value), int(n.value)

    def rank_world(self) -> Tuple[int, int]:
        """(rank, world size) as the library's communicator sees them (RCCL's own view, not the launcher's environment)."""
        import ctypes as C
        r, w = C.c_int32(), C.c_int32()
        hip.check(self.lib.ams_comm_stats(self._h, C.byref(r), C.byref(w), None, None), "ams_comm_stats")
        return int(r.value), int(w.value)

    def all_reduce(self, t: torch.Tensor) -> None:
        """Sum ``t`` (float32 / float64, contiguous, on this rank's GPU) in place across ranks on torch's current stream."""
        import ctypes as C
        code = {torch.float32: hip.DT_F32, torch.float64: hip.DT_F64}[t.dtype]
        assert t.is_cuda and t.is_contiguous()
        hip.check(self.lib.ams_comm_allreduce(self._h, C.c_void_p(t.data_ptr()), t.numel(), code,
                                              C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)), "ams_comm_allreduce")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.ams_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class ArenaAllReduce:
    """The engine's ``ams_allreduce_cb``: sum ``count`` elements at byte ``offset`` of the arena across ranks.

    ``reduce_fn`` defaults to ``torch.distributed.all_reduce`` on ``group``.  Instances count calls and bytes so
    tests and DESIGN.md can state what a step exchanges."""

    def __init__(self, arena: torch.Tensor, group=None, reduce_fn: Optional[Callable[[torch.Tensor], None]] = None):
        assert arena.dtype == torch.uint8 and arena.dim() == 1
        self.arena = arena
        self.group = group
        self.reduce_fn = reduce_fn
        self.calls = 0
        self.bytes = 0
        self.error: Optional[BaseException] = None

    def view(self, offset: int, count: int, dtype_code: int) -> torch.Tensor:
        if dtype_code == hip.DT_F64:
            return self.arena[offset:offset + 8 * count].view(torch.float64)
        if dtype_code == hip.DT_F32:
            return self.arena[offset:offset + 4 * count].view(torch.float32)
        raise ValueError("all-reduce of dtype code %d is not part of the ABI" % dtype_code)

    def __call__(self, _user, offset, count, dtype_code) -> int:
        try:
            t = self.view(int(offset), int(count), int(dtype_code))
            if self.reduce_fn is not None:
                self.reduce_fn(t)
            else:
                import torch.distributed as dist
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.calls += 1
            self.bytes += t.numel() * t.element_size()
            return 0
        except BaseException as e:  # noqa: BLE001  (must not propagate through the C frame)
            self.error = e
            return 1
