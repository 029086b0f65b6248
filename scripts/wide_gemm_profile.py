#!/usr/bin/env python3
"""One wide-GEMM configuration launched repeatedly (for rocprofv3 --kernel-trace / --pmc passes).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -- python3 scripts/wide_gemm_profile.py --m 2048 --n 28672 --k 4096 --tile 256x256 --variant 0
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=28672)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--tile", default="256x256")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--epi", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--n-fastest", action="store_true")
    ap.add_argument("--library", action="store_true")
    ap.add_argument("--time", action="store_true", help="print the mean time per launch (events around a second round of launches)")
    a = ap.parse_args()
    from kvmini.ops import native

    C = native()
    dev = torch.device("cuda:0")
    bm, bn = (int(v) for v in a.tile.split("x"))
    copies = max(2, int(600e6 // (a.n * a.k * 2)) + 1)
    Ws = [(torch.randn(a.n, a.k, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    Wps = [C.pack_weight(w) for w in Ws]
    x = torch.randn(a.m, a.k, device=dev).bfloat16()
    y = torch.empty((a.m, a.n // 2 if a.epi == 2 else a.n), dtype=torch.bfloat16, device=dev)
    nst = a.k // (32 * C.wide_gemm_stage_ksteps(bm, bn, a.variant))
    def run(i):
        if a.library:
            F.linear(x, Ws[i % copies])
        else:
            C.wide_gemm(Wps[i % copies], a.n, a.k, x, y, a.epi, bm, bn, a.variant, 1, nst, None, 0.0, None, not a.n_fastest)

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
        if not a.library and a.epi == 0:
            ref = x.float() @ Ws[(a.iters - 1) % copies].float().T
            err = (y.float() - ref).abs().max().item()
        else:
            err = None
        print(f"m={a.m} n={a.n} k={a.k} {'library' if a.library else f'tile {a.tile} variant {a.variant} epi {a.epi}'}: {us:.1f} us, {2.0 * a.m * a.n * a.k / us / 1e6:.0f} TFLOP/s, max err {err}", flush=True)


if __name__ == "__main__":
    main()
