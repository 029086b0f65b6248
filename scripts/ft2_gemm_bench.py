#!/usr/bin/env python3
"""Streaming GEMM with two feature tiles per wave (FT = 2 of csrc/stream_gemm.hip) against one,
and against the path the engine takes today, on the Llama-3-8B projection shapes (one MI355X;
weights rotated through > 600 MB of copies so no cache serves re-reads; fp32 check of every variant).

    python scripts/ft2_gemm_bench.py --out gpurun_out/ft2_gemm.json [--rows 32,128,256] [--formats bf16,fp8,mxfp4]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def time_fns(fns, rounds=8, inner=4):
    res = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            e1.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1000.0 / inner)
    return {k: statistics.median(v) for k, v in res.items()}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/ft2_gemm.json")
    ap.add_argument("--rows", default="32,128,256")
    ap.add_argument("--formats", default="bf16,fp8,mxfp4")
    ap.add_argument("--shapes", default="qkv,o,gate_up,down")
    a = ap.parse_args()
    from kvmini.ops import hip_ops, native

    C = native()
    dev = torch.device("cuda:0")
    shapes = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 0), "gate_up": (28672, 4096, 2), "down": (4096, 14336, 0)}
    out = []
    torch.manual_seed(0)
    for fmt in a.formats.split(","):
        bits = {"bf16": 16, "fp8": 8, "mxfp4": 4}[fmt]
        for name in a.shapes.split(","):
            N, K, epi = shapes[name]
            copies = max(2, int(600e6 // (N * K * bits // 8)) + 1)
            Ws = [hip_ops.PackedLinear((torch.randn(N, K, device=dev) * 0.02).bfloat16(), fp8=fmt == "fp8", mxfp4=fmt == "mxfp4") for _ in range(copies)]
            for M in [int(r) for r in a.rows.split(",")]:
                if M == 256 and bits == 4:
                    continue
                x = torch.randn(M, K, device=dev).bfloat16()
                st = {"i": 0}

                def nxt():
                    st["i"] = (st["i"] + 1) % copies
                    return st["i"]

                n_out = N // 2 if epi == 2 else N
                fns, outs = {}, {}

                def engine_path():
                    w = Ws[nxt()]
                    return hip_ops.linear_swiglu(x, w) if epi == 2 else hip_ops.linear(x, w, allow_slabs=True)

                fns["engine_today"] = engine_path
                cols = 128
                for ft in (1, 2):
                    if ft == 2 and M <= 64 and bits != 4:
                        continue  # FT = 2 at few rows is built for 4-bit weights only
                    for nw in (4, 8) if ft == 2 else (4, 6, 7, 8):
                        if N % (16 * nw * ft) or (M == 256 and ft == 2 and nw == 8):
                            continue
                        wgs_x = N // (16 * nw * ft)
                        if M <= 64:
                            nbs = ((8, 4, 2) if bits != 16 else (4, 2)) if epi else (4, 2)
                        else:
                            nbs = (2,)
                        for nb in nbs:
                            if K % (cols * nb) or (nb == 8 and M > 32):
                                continue
                            ngroups = K // (cols * nb)
                            split_opts = (1,) if epi else sorted({s for s in (1, 2, 4, 6, 8, 16) if s <= ngroups})
                            for want in split_opts:
                                gps = -(-ngroups // want)
                                splits = -(-ngroups // gps)
                                if epi == 0 and (wgs_x * splits < 128 or wgs_x * splits > 640):
                                    continue
                                key = f"ft{ft}_nw{nw}_nb{nb}_s{splits}"
                                if key in fns:
                                    continue
                                if splits > 1:
                                    buf = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
                                    e = 1
                                else:
                                    buf = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
                                    e = epi

                                def mk(buf=buf, e=e, nb=nb, splits=splits, gps=gps, nw=nw, ft=ft):
                                    def f(i=None):
                                        w = Ws[nxt() if i is None else i]
                                        C.stream_gemm3(w.wp, N, K, x, buf, e, nb, splits, gps, None, 0.0, w.wscale, nw, bits if bits == 4 else 0, ft=ft)
                                        return buf
                                    return f
                                fns[key] = mk()
                                outs[key] = (fns[key], splits)
                # every variant against the first FT = 1 one on copy 0 (that kernel is tested against fp32 in tests/)
                ref = None
                for key, (f, splits) in outs.items():
                    y = f(0)
                    torch.cuda.synchronize()
                    y = y.sum(0) if splits > 1 else y.float()
                    if ref is None:
                        ref = y.clone()
                    err = ((y - ref).abs().max() / ref.abs().max()).item()
                    assert err < 2e-2, (fmt, name, M, key, err)
                t = time_fns(fns)
                best = min((v, k) for k, v in t.items() if k != "engine_today")
                best1 = min((v, k) for k, v in t.items() if k.startswith("ft1"))
                best2 = min(((v, k) for k, v in t.items() if k.startswith("ft2")), default=(0.0, None))
                rec = {"format": fmt, "shape": name, "M": M, "N": N, "K": K, "engine_today_us": round(t["engine_today"], 2), "best_ft1": [best1[1], round(best1[0], 2)],
                       "best_ft2": [best2[1], round(best2[0], 2)], "weight_TBps_best": round(N * K * bits / 8 / best[0] / 1e6, 2), "us": {k: round(v, 2) for k, v in sorted(t.items(), key=lambda kv: kv[1])}}
                out.append(rec)
                print(json.dumps({k: v for k, v in rec.items() if k != "us"}), flush=True)
            del Ws
            torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
