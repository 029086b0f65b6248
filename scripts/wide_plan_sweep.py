#!/usr/bin/env python3
"""Tiled GEMM at 129-512 rows: every (tile, K split) the kernel offers, timed on cold weights, next to the
planner's pick for the same shape (``kvmini.ops.hip_ops._plan_wide``).  Slab consumers are priced as the planner
prices them (2 x splits x m x n x 4 bytes at 2.5 TB/s).

    python3 scripts/wide_plan_sweep.py --rows 160,256,512 --out gpurun_out/wide_plan_sweep.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="160,192,256,384,512")
    ap.add_argument("--shapes", default="qkv:6144:4096,o:4096:4096,gate_up:28672:4096,down:4096:14336")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--out", default="gpurun_out/wide_plan_sweep.json")
    a = ap.parse_args()
    from kvmini.ops import hip_ops, native

    C = native()
    dev = torch.device("cuda:0")
    rows = [int(v) for v in a.rows.split(",")]
    res = []
    for spec in a.shapes.split(","):
        name, n, k = spec.split(":")
        n, k = int(n), int(k)
        copies = max(2, int(600e6 // (n * k * 2)) + 1)
        Wps = [C.pack_weight((torch.randn(n, k, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        may_split = name != "gate_up"
        for m in rows:
            x = torch.randn(m, k, device=dev).bfloat16()
            pick = hip_ops._plan_wide(m, n, k, 0, may_split)
            cands = []
            for (bm, bn, variant), _step in hip_ops._WIDE_TILES_BF16:
                if n % bn:
                    continue
                tiles = -(-m // bm) * (n // bn)
                per_stage = 32 * C.wide_gemm_stage_ksteps(bm, bn, variant)
                nst = k // per_stage
                for splits in ((1, 2, 3, 4, 5, 6, 8, 10) if may_split else (1,)):
                    if splits > 1 and (tiles * splits > 768 or nst < 4 * splits):
                        continue
                    sps = -(-nst // splits)
                    if (splits - 1) * sps >= nst:
                        continue
                    if splits > 1:
                        out = torch.empty((splits, m, n), dtype=torch.float32, device=dev)
                        epi = 1
                    else:
                        out = torch.empty((m, n // 2 if name == "gate_up" else n), dtype=torch.bfloat16, device=dev)
                        epi = 2 if name == "gate_up" else 0
                    def run(i):
                        C.wide_gemm(Wps[i % copies], n, k, x, out, epi, bm, bn, variant, splits, sps, None, 0.0, None, True)
                    for i in range(3):
                        run(i)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(a.iters):
                        run(i)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / a.iters * 1e3
                    slab = 1.0 * splits * m * n * 4 / 2.5e6 if splits > 1 else 0.0  # the consumer's read; the write is inside the measured kernel
                    cands.append({"tile": [bm, bn, variant], "splits": splits, "sps": sps, "wgs": tiles * splits, "us": round(us, 1), "with_slabs_us": round(us + slab, 1)})
            cands.sort(key=lambda c: c["with_slabs_us"])
            mine = next((c for c in cands if tuple(c["tile"]) == tuple(pick[:3]) and c["splits"] == pick[3]), None)
            res.append({"shape": name, "n": n, "k": k, "rows": m, "planner": list(pick), "planner_us": mine and mine["with_slabs_us"], "best": cands[:4]})
            b = cands[0]
            print(f"{name:8s} m={m:4d} planner {pick[:4]} {mine and mine['with_slabs_us']} us | best {b['tile']} x{b['splits']} {b['with_slabs_us']} us (kernel {b['us']})", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
