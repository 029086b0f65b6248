"""Python side of ``csrc/xgmi_collectives.hip``: hand-written one-shot / two-shot collectives
over xGMI peer memory for the latency-bound messages of an 8 x MI355X node.

The reference has no collectives at all (its multi-GPU telemetry goes through Prometheus
scrapes, ``energy/collector.py:203-251``, and tensor parallelism is a string forwarded to the
served backend, ``runners/backends/vllm/deploy.sh:78-79``); here every rank is one process on
one GPU and the small, frequent messages — the lock-step telemetry table, the per-token
histogram bins, the all-reduce behind every row-parallel projection of a TP model — go
GPU to GPU through peer-mapped buffers instead of a library ring:

* :meth:`XgmiComm.all_gather` — every rank stores its payload into every peer's slot (all
  seven links of a GPU at once), one flag per workgroup, no host synchronisation;
* :meth:`XgmiComm.all_reduce` — one-shot (<= 256 KiB) or two-shot reduce-scatter + all-gather
  (<= the slot size, 4 MiB by default), bf16 on the wire, fp32 sums in rank order (every rank
  gets bitwise the same result);
* :meth:`XgmiComm.residual_allreduce` — the TP epilogue in one launch: split-K slab sum ->
  all-reduce -> residual add -> per-row sums of squares.

Bootstrap: each rank allocates its buffer, the 64-byte IPC handles travel over the
``torch.distributed`` group that already exists (gloo or RCCL), peers are mapped with
``hipIpcOpenMemHandle``.  A group of one rank is its own peer (tests on one GPU run the same
kernels).  Calls alternate between two sets of slots; a captured HIP graph must therefore hold
an even number of calls (a decoder layer makes two) — :attr:`XgmiComm.calls` lets callers check.
"""

from __future__ import annotations

from typing import Any, List, Optional

import torch

ONE_SHOT_MAX_BYTES = 256 * 1024


class XgmiComm:
    def __init__(self, group: Any = None, device: Optional[torch.device] = None, slot_bytes: int = 4 << 20, world: Optional[int] = None, rank: Optional[int] = None,
                 handles: Optional[List[bytes]] = None):
        """Collective over ``group`` (``torch.distributed``), or — with ``world`` / ``rank`` given —
        over ranks whose IPC handles the caller exchanges itself: pass ``handles=None`` first, ship
        :attr:`handle` to the peers by any channel, then call :meth:`open_peers`."""
        import torch.distributed as dist

        from kvmini.ops import native

        self.C = native()
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank = int(world), int(rank or 0)
        if not 1 <= self.world <= 8:
            raise ValueError("xGMI collectives serve one node: 1..8 ranks")
        self.slot_bytes = (int(slot_bytes) + 4095) // 4096 * 4096
        self.h = self.C.xgmi_create(self.world, self.rank, self.slot_bytes)
        self.handle: bytes = self.C.xgmi_handle(self.h) if self.world > 1 else b""
        if self.world > 1 and handles is None and dist.is_initialized() and world == dist.get_world_size(group):
            gathered: List[Any] = [None] * self.world
            dist.all_gather_object(gathered, self.handle, group=group)
            handles = gathered
        if handles is not None:
            self.open_peers(handles)

    def open_peers(self, handles: List[bytes]) -> None:
        assert len(handles) == self.world
        for r, hb in enumerate(handles):
            if r != self.rank:
                self.C.xgmi_open_peer(self.h, r, hb)
        assert self.C.xgmi_ready(self.h)

    # ------------------------------------------------------------------
    @property
    def calls(self) -> int:
        return int(self.C.xgmi_calls(self.h))

    def errors(self) -> int:
        """Spin timeouts counted by the kernels (synchronises the device)."""
        return int(self.C.xgmi_errors(self.h))

    def fits(self, nbytes: int) -> bool:
        return 0 < nbytes <= self.slot_bytes

    def fits_residual(self, t: int, h: int) -> bool:
        return t * h * 2 <= self.slot_bytes and t * ((h + 511) // 512) <= int(self.C.xgmi_max_wg()) and h % 8 == 0

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """``x``: contiguous device tensor whose byte size is a multiple of 16 -> ``[world, *x.shape]``."""
        x = x.contiguous()
        out = torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        self.C.xgmi_allgather(self.h, x, out)
        return out

    def all_reduce(self, x: torch.Tensor, two_shot: Optional[bool] = None) -> torch.Tensor:
        """Sum of a bf16 tensor over the ranks, in place.  ``two_shot=None`` picks by size."""
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        n = x.numel()
        if two_shot is None:
            two_shot = n * 2 > ONE_SHOT_MAX_BYTES and n % (8 * self.world) == 0
        self.C.xgmi_allreduce(self.h, x.view(-1), x.view(-1), bool(two_shot))
        return x

    def residual_allreduce(self, x: torch.Tensor, residual: torch.Tensor, has_residual: bool) -> torch.Tensor:
        """``residual += all_reduce(x)`` (``x``: bf16 ``[T, H]`` or fp32 slabs ``[S, T, H]``), in
        place; returns the partial sums of squares ``[T, ceil(H / 512)]`` of the new residual."""
        return self.C.xgmi_residual_allreduce(self.h, x, residual, bool(has_residual))

    def close(self) -> None:
        if self.h:
            self.C.xgmi_destroy(self.h)
            self.h = 0
