"""Host model of the signalling protocol of ``csrc/xgmi_collectives.hip``, step for step, over
``multiprocessing.shared_memory`` — one process per rank, each rank's "device buffer" a region
of one shared segment that the other ranks write into the way GPU peers store over xGMI.

It exists so that the protocol's logic — per-workgroup flags that carry a constant 1 and are
reset by their reader, two alternating sets of slots, no barrier between consecutive calls, ranks
that run arbitrarily far out of step — is exercised by real concurrent processes on a CPU-only
machine (the kernels themselves run on one GPU with one rank, and with two rank processes
sharing that GPU).  What it cannot show is the memory-model side (system-scope release /
acquire, uncached peer memory): x86 orders these plain stores by itself.

A rank's region: ``flags[2 parities][2 phases][world][n_wg]`` int32, then
``slots[2 parities][world][slot_elems]`` float32, then ``gather[2 parities][slot_elems]``.
"""

from __future__ import annotations

import time
from multiprocessing import shared_memory
from typing import List, Optional

import numpy as np


class RankRegion:
    def __init__(self, buf: memoryview, base: int, world: int, n_wg: int, slot_elems: int):
        nf = 2 * 2 * world * n_wg
        self.flags = np.ndarray((2, 2, world, n_wg), dtype=np.int32, buffer=buf, offset=base)
        self.slots = np.ndarray((2, world, slot_elems), dtype=np.float32, buffer=buf, offset=base + nf * 4)
        self.gather = np.ndarray((2, slot_elems), dtype=np.float32, buffer=buf, offset=base + nf * 4 + 2 * world * slot_elems * 4)

    @staticmethod
    def nbytes(world: int, n_wg: int, slot_elems: int) -> int:
        return 2 * 2 * world * n_wg * 4 + 2 * world * slot_elems * 4 + 2 * slot_elems * 4


class HostComm:
    """One rank's view: ``peer[r]`` is rank r's region (``peer[rank]`` its own)."""

    def __init__(self, name: Optional[str], world: int, rank: int, n_wg: int = 4, wg_elems: int = 64, create: bool = False):
        self.world, self.rank, self.n_wg, self.wg_elems = world, rank, n_wg, wg_elems
        self.slot_elems = n_wg * wg_elems
        per = RankRegion.nbytes(world, n_wg, self.slot_elems)
        if create:
            self.shm = shared_memory.SharedMemory(create=True, size=per * world, name=name)
            self.shm.buf[:] = bytes(per * world)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.owner = create
        self.peer: List[RankRegion] = [RankRegion(self.shm.buf, r * per, world, n_wg, self.slot_elems) for r in range(world)]
        self.parity = 0
        self.timeouts = 0

    def _signal_and_wait(self, parity: int, phase: int, wg: int, timeout_s: float) -> None:
        for r in range(self.world):  # "lane r": one flag on every peer
            self.peer[r].flags[parity, phase, self.rank, wg] = 1
        mine = self.peer[self.rank].flags
        for r in range(self.world):
            t0 = time.perf_counter()
            while mine[parity, phase, r, wg] != 1:
                if time.perf_counter() - t0 > timeout_s:
                    self.timeouts += 1
                    break
                time.sleep(0)
            mine[parity, phase, r, wg] = 0  # ready for the call after next

    def all_reduce(self, x: np.ndarray, two_shot: bool = False, timeout_s: float = 20.0) -> np.ndarray:
        """Sum of ``x`` (float32, ``n_wg * wg_elems`` elements; two-shot: that many per chunk) over the
        ranks; the "workgroups" of a rank run one after the other here."""
        parity = self.parity
        self.parity ^= 1
        W, we = self.world, self.wg_elems
        out = np.empty_like(x)
        if not two_shot:
            for wg in range(self.n_wg):
                sl = slice(wg * we, (wg + 1) * we)
                for k in range(W):
                    j = (self.rank + 1 + k) % W
                    self.peer[j].slots[parity, self.rank, sl] = x[sl]
                self._signal_and_wait(parity, 0, wg, timeout_s)
                out[sl] = sum(self.peer[self.rank].slots[parity, r, sl] for r in range(W))
            return out
        cn = x.size // W
        assert cn * W == x.size and x.size <= self.slot_elems and cn % we == 0
        for wg in range(cn // we):
            sl = slice(wg * we, (wg + 1) * we)
            for k in range(W):
                j = (self.rank + 1 + k) % W
                self.peer[j].slots[parity, self.rank, sl] = x[j * cn + wg * we : j * cn + (wg + 1) * we]
            self._signal_and_wait(parity, 0, wg, timeout_s)
            red = sum(self.peer[self.rank].slots[parity, r, sl] for r in range(W))
            for k in range(W):
                j = (self.rank + 1 + k) % W
                self.peer[j].gather[parity, self.rank * cn + wg * we : self.rank * cn + (wg + 1) * we] = red
            self._signal_and_wait(parity, 1, wg, timeout_s)
            for r in range(W):
                out[r * cn + wg * we : r * cn + (wg + 1) * we] = self.peer[self.rank].gather[parity, r * cn + wg * we : r * cn + (wg + 1) * we]
        return out

    def close(self) -> None:
        self.peer = []
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass
