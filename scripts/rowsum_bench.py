#!/usr/bin/env python3
"""Row-parallel projection + residual update on one MI355X: the row-sum GEMM (K split inside the
workgroup, residual add in the epilogue, one launch) against the split-K streaming GEMM followed by
the residual kernel, on the Llama-3-8B output and down projections, with cold weights (several
copies in rotation, so the Infinity Cache cannot serve re-reads).  JSON lines on stdout."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from scripts.gpu_microbench import time_fns  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="16,32,64")
    ap.add_argument("--shapes", default="o:4096x4096,down:4096x14336")
    ap.add_argument("--fp8", action="store_true")
    a = ap.parse_args()
    from kvmini.ops import hip_ops

    dev = torch.device("cuda:0")
    ones = None
    for item in a.shapes.split(","):
        name, _, nk = item.partition(":")
        N, K = (int(v) for v in nk.split("x"))
        copies = max(3, int(700e6 // (N * K * (1 if a.fp8 else 2))) + 1)
        pws = [hip_ops.PackedLinear((torch.randn(N, K, device=dev) * 0.02).bfloat16(), fp8=a.fp8) for _ in range(copies)]
        ones = torch.ones(N, device=dev, dtype=torch.bfloat16)
        for M in (int(v) for v in a.rows.split(",")):
            x = torch.randn(M, K, device=dev).bfloat16()
            res = torch.randn(M, N, device=dev).bfloat16()
            st = {"i": 0}

            def nxt():
                st["i"] = (st["i"] + 1) % copies
                return pws[st["i"]]

            def pair():
                hip_ops.add_rmsnorm(hip_ops.linear(x, nxt(), allow_slabs=True), res, ones, 1e-5)

            def gemm_only():
                hip_ops.linear(x, nxt(), allow_slabs=True)

            fns = {"gemm+residual_kernel": pair, "gemm_alone": gemm_only}
            for nb in (2, 3, 4):
                if nb > 2 and M > 32:
                    continue

                def rs(nb=nb):
                    hip_ops.ROWSUM_NB = nb
                    hip_ops._linear_residual_rowsum(x, nxt(), res, 1e-5)

                fns[f"rowsum_nb{nb}"] = rs
            t = time_fns(fns, rounds=15, inner=20)
            wbytes = N * K * (1 if a.fp8 else 2)
            row = {"bench": "rowsum_gemm", "shape": name, "N": N, "K": K, "rows": M, "fp8": a.fp8, "us": {k: round(v, 2) for k, v in t.items()},
                   "TBps": {k: round(wbytes / v / 1e6, 2) for k, v in t.items()}}
            print(json.dumps(row), flush=True)
        del pws
        torch.cuda.empty_cache()
    return 0


if __name__ == "__main__":
    sys.exit(main())
