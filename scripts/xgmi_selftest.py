#!/usr/bin/env python3
"""xGMI peer-memory collectives (csrc/xgmi_collectives.hip) exercised by W rank processes.

On a node with W GPUs every rank takes its own GPU; on a one-GPU box all ranks share GPU 0
(`--same-gpu`): the buffers are still exported / mapped through HIP IPC between processes and
the kernels of the ranks still spin on each other's flags, so the whole protocol runs — only
the links are missing.  Each rank checks every result against sums it forms from a gloo
all-gather of the inputs, and the parent prints one JSON line.

    python scripts/xgmi_selftest.py --world 2 --same-gpu --iters 40
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a) -> int:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0 if a.same_gpu else rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    from kvmini.parallel.xgmi import XgmiComm

    comm = XgmiComm(device=dev, slot_bytes=4 << 20)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(100 + rank)
    worst = {"allgather": 0.0, "allreduce1": 0.0, "allreduce2": 0.0, "residual": 0.0}
    t_us = {"allreduce1": [], "allreduce2": [], "residual": [], "allgather": []}

    def gather_cpu(t):
        lst = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        return lst

    for it in range(a.iters):
        # all-gather of a telemetry-sized table
        tab = torch.randn(8 + it % 5, 16, generator=gen, dtype=torch.float64)
        got = comm.all_gather(tab.to(dev))
        want = torch.stack(gather_cpu(tab))
        worst["allgather"] = max(worst["allgather"], float((got.cpu() - want).abs().max()))
        # one-shot / two-shot all-reduce
        for name, n, two in (("allreduce1", 8 * (1000 + 8 * it), False), ("allreduce2", 8 * world * (4000 + it), True)):
            x = torch.randn(n, generator=gen).bfloat16()
            ref = sum(v.float() for v in gather_cpu(x)).bfloat16().float()
            xd = x.to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            comm.all_reduce(xd, two_shot=two)
            torch.cuda.synchronize()
            t_us[name].append((time.perf_counter() - t0) * 1e6)
            worst[name] = max(worst[name], float((xd.float().cpu() - ref).abs().max() / ref.abs().max()))
        # fused TP epilogue on split-K slabs
        T, H, S = 4 + it % 13, 1024, 3
        slabs = torch.randn(S, T, H, generator=gen)
        res = torch.randn(T, H, generator=gen).bfloat16()
        local = slabs.sum(0).bfloat16()
        tot = sum(v.float() for v in gather_cpu(local)).bfloat16().float()
        want_res = (tot + res.float()).bfloat16().float()
        rd = res.to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ss = comm.residual_allreduce(slabs.to(dev), rd, True)
        torch.cuda.synchronize()
        t_us["residual"].append((time.perf_counter() - t0) * 1e6)
        worst["residual"] = max(worst["residual"], float((rd.float().cpu() - want_res).abs().max() / want_res.abs().max()),
                                float((ss.sum(-1).cpu() - (want_res ** 2).sum(-1)).abs().max() / (want_res ** 2).sum(-1).max()))
    errs = comm.errors()
    out = {"rank": rank, "world": world, "iters": a.iters, "worst": worst, "timeouts": errs, "calls": comm.calls,
           "median_us": {k: sorted(v)[len(v) // 2] for k, v in t_us.items() if v}}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    comm.close()
    dist.destroy_process_group()
    if rank == 0:
        ok = all(g["timeouts"] == 0 and max(g["worst"].values()) < 1e-2 and g["worst"]["allgather"] == 0.0 for g in gathered)
        print(json.dumps({"ok": ok, "same_gpu": a.same_gpu, "ranks": gathered}))
        return 0 if ok else 1
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--same-gpu", action="store_true")
    a = ap.parse_args()
    if "RANK" in os.environ:
        return worker(a)
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 240
    for p in procs:
        try:
            rc = rc or p.wait(max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 124
    return rc


if __name__ == "__main__":
    sys.exit(main())
