"""Multi-process CPU tests of the two host-visible protocols of the tensor-parallel path: the
shared-memory step channel (scheduling rank -> followers) and the flag / parity protocol of
the xGMI peer-memory collectives (host model of csrc/xgmi_collectives.hip)."""
import multiprocessing as mp
import os
import random
import time

import numpy as np

from kvmini.parallel.step_channel import StepChannel
from kvmini.parallel.xgmi_protocol import HostComm


def _reader(name, n_ints, idx, n_images, q):
    ch = StepChannel(name, n_ints, ring=4, reader=idx)
    rng = random.Random(idx)
    out = np.empty(n_ints, dtype=np.int32)
    ok = True
    for k in range(n_images):
        assert ch.receive(out, timeout_s=30.0)
        ok = ok and out[0] == k and out[-1] == 7 * k and int(out.sum()) == k + 7 * k + int(out[1:-1].sum()) and np.all(out[1:-1] == k % 97)
        if rng.random() < 0.05:
            time.sleep(rng.random() * 2e-3)  # a slow follower: the writer must wait for the slot
    q.put((idx, ok, ch.seq))
    ch.close()


def test_step_channel_single_writer_two_readers_processes():
    """500 step images through a ring of 4 slots to two reader processes with random stalls:
    every reader sees every image once, in order and untorn; the writer never overruns a slot."""
    n_ints, n_images = 1000, 500
    ch = StepChannel(f"kvt_{os.getpid()}", n_ints, ring=4, create=True, n_readers=2)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reader, args=(ch.name, n_ints, i, n_images, q)) for i in range(2)]
    for p in procs:
        p.start()
    img = np.empty(n_ints, dtype=np.int32)
    for k in range(n_images):
        img[:] = k % 97
        img[0], img[-1] = k, 7 * k
        ch.publish(img, timeout_s=30.0)
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    ch.close()
    assert res == [(0, True, n_images), (1, True, n_images)]


def _rank(name, world, rank, iters, q):
    comm = HostComm(name, world, rank, n_wg=4, wg_elems=48)
    rng = random.Random(1000 + rank)
    ok = True
    n = comm.slot_elems
    for it in range(iters):
        two = it % 3 == 2
        # every rank's input is a function of (rank, it): the expected sum needs no communication
        x = (np.arange(n, dtype=np.float32) % 13) * (rank + 1) + it
        want = sum((np.arange(n, dtype=np.float32) % 13) * (r + 1) + it for r in range(world))
        if two:
            m = n // world // comm.wg_elems * comm.wg_elems * world
            got = comm.all_reduce(x[:m], two_shot=True)
            ok = ok and np.array_equal(got, want[:m])
        else:
            ok = ok and np.array_equal(comm.all_reduce(x), want)
        if rng.random() < 0.1:
            time.sleep(rng.random() * 3e-3)  # ranks drift apart by whole calls
    q.put((rank, bool(ok), comm.timeouts))
    comm.close()


def test_xgmi_flag_protocol_three_rank_processes():
    """Three rank processes, 150 back-to-back all-reduces (one-shot and two-shot mixed) with no
    barrier in between and random stalls: constant-1 flags reset by their reader plus two
    alternating slot sets never mix two calls' data and never deadlock."""
    world, iters = 3, 150
    host = HostComm(f"kvx_{os.getpid()}", world, 0, n_wg=4, wg_elems=48, create=True)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(host.shm.name, world, r, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    host.close()
    assert res == [(r, True, 0) for r in range(world)]
