"""Host-side step channel of a tensor-parallel group: the scheduling rank publishes the staged
inputs of every engine step (one int32 image: tokens, positions, slots, block tables, header)
in a POSIX shared-memory ring; the follower ranks — one process per GPU of the same node —
pick each image up from there and enqueue the same step on their GPU.

Nothing on this path touches a device: the followers never synchronise with their GPU to learn
what to run next (no broadcast of a device tensor, no read-back of a header), so a follower can
enqueue step n+1 while step n is still running — launch-ahead stepping works under tensor
parallelism, and the only device-side coupling between the ranks is the all-reduce itself.

Protocol (single writer, W-1 readers, ring of R slots):

* writer: wait until every reader has acknowledged image ``seq - R`` (the slot is free), copy the
  image into slot ``seq % R``, then store ``seq + 1`` into the head word;
* reader r: wait until ``head > mine``, copy slot ``mine % R`` out, store ``mine + 1`` into
  ``ack[r]``.

The head / ack words are 8-byte aligned and written with single stores after the data they
publish; x86 keeps stores in order, so a reader that sees the new head sees the image.
"""

from __future__ import annotations

import time
from multiprocessing import shared_memory
from typing import Optional

import numpy as np

HEADER_BYTES = 64
MAX_READERS = 8


class StepChannel:
    def __init__(self, name: Optional[str], n_ints: int, ring: int = 4, create: bool = False, reader: int = -1, n_readers: int = 0):
        """``create=True`` (the writer) makes the segment; readers attach by ``name`` with their
        index ``reader`` in ``[0, n_readers)``."""
        self.n_ints, self.ring = int(n_ints), int(ring)
        size = HEADER_BYTES + MAX_READERS * 8 + self.ring * self.n_ints * 4
        if create:
            self.shm = shared_memory.SharedMemory(create=True, size=size, name=name)
            self.shm.buf[: HEADER_BYTES + MAX_READERS * 8] = bytes(HEADER_BYTES + MAX_READERS * 8)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.name = self.shm.name
        self.owner = create
        self.head = np.ndarray((1,), dtype=np.int64, buffer=self.shm.buf, offset=0)
        self.nread = np.ndarray((1,), dtype=np.int64, buffer=self.shm.buf, offset=8)
        self.acks = np.ndarray((MAX_READERS,), dtype=np.int64, buffer=self.shm.buf, offset=HEADER_BYTES)
        self.slots = np.ndarray((self.ring, self.n_ints), dtype=np.int32, buffer=self.shm.buf, offset=HEADER_BYTES + MAX_READERS * 8)
        self.writer_pid = np.ndarray((1,), dtype=np.int64, buffer=self.shm.buf, offset=16)
        self.reader = reader
        self.seq = 0  # writer: images published; reader: images consumed
        if create:
            import os

            self.nread[0] = n_readers
            self.writer_pid[0] = os.getpid()

    def writer_alive(self) -> bool:
        """Readers (same host): is the process that publishes the images still there?  A follower
        whose scheduling rank died must exit and give its GPU back instead of polling for ever."""
        import os

        pid = int(self.writer_pid[0])
        if pid <= 0:
            return True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        return True

    # ---- writer ---------------------------------------------------------
    def publish(self, image: np.ndarray, timeout_s: float = 60.0) -> None:
        n_readers = int(self.nread[0])
        if n_readers:
            t0 = time.perf_counter()
            while int(self.acks[:n_readers].min()) < self.seq - self.ring + 1:
                if time.perf_counter() - t0 > timeout_s:
                    raise TimeoutError("tensor-parallel follower did not pick up a step image")
                time.sleep(20e-6)
        self.slots[self.seq % self.ring, : image.size] = image
        self.seq += 1
        self.head[0] = self.seq  # published after the data

    # ---- reader ---------------------------------------------------------
    def receive(self, out: np.ndarray, timeout_s: Optional[float] = None) -> bool:
        """Copy the next image into ``out``.  False on timeout."""
        t0 = time.perf_counter()
        spins = 0
        while int(self.head[0]) <= self.seq:
            spins += 1
            if spins > 2000:  # ~ms of busy polling, then yield the core between polls
                time.sleep(50e-6)
            if timeout_s is not None and time.perf_counter() - t0 > timeout_s:
                return False
        out[:] = self.slots[self.seq % self.ring, : out.size]
        self.seq += 1
        self.acks[self.reader] = self.seq
        return True

    def close(self) -> None:
        # drop the numpy views first: SharedMemory.close() refuses while exported buffers exist
        self.head = self.nread = self.acks = self.slots = self.writer_pid = None  # type: ignore[assignment]
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass
