#!/usr/bin/env python3
"""One streaming-GEMM configuration launched repeatedly (for rocprofv3 --pmc / --kernel-trace passes).

    rocprofv3 --pmc SQ_WAVE_CYCLES ... -- python3 scripts/stream_gemm_profile.py --rows 64 --n 28672 --k 4096 --epi 2 --nw 7 --nb 2
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64, help="16, 32, 48, 64, 96, 128 or 256")
    ap.add_argument("--n", type=int, default=28672)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--epi", type=int, default=2)
    ap.add_argument("--nw", type=int, default=7)
    ap.add_argument("--nb", type=int, default=2)
    ap.add_argument("--ft", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--time", action="store_true")
    a = ap.parse_args()
    from kvmini.ops import hip_ops, native

    C = native()
    dev = torch.device("cuda:0")
    copies = max(2, int(600e6 // (a.n * a.k * 2)) + 1)
    Ws = [hip_ops.PackedLinear((torch.randn(a.n, a.k, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(a.rows, a.k, device=dev).bfloat16()
    y = torch.empty((a.rows, a.n // 2 if a.epi == 2 else a.n), dtype=torch.bfloat16, device=dev)
    gps = a.k // (128 * a.nb)

    def run(i):
        w = Ws[i % copies]
        C.stream_gemm3(w.wp, a.n, a.k, x, y, a.epi, a.nb, 1, gps, None, 0.0, w.wscale, a.nw, 0, ft=a.ft)

    for i in range(a.iters):
        run(i)
    torch.cuda.synchronize()
    if a.time:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"rows={a.rows} n={a.n} k={a.k} nw={a.nw} nb={a.nb} ft={a.ft}: {us:.1f} us, {a.n * a.k * 2 / us / 1e6:.2f} TB/s of weights", flush=True)


if __name__ == "__main__":
    main()
