#!/usr/bin/env python3
"""A short new turn behind a long cached prefix (the chat-continuation step): attention time per layer of
the tile kernel walking the whole context per tile against its split form.  GPU box.

  python scripts/prefill_split_bench.py --prefix 131072 --new 64,256,512 --out gpurun_out/prefill_split.json
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.ops import hip_ops


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefix", type=int, default=131072)
    ap.add_argument("--new", default="32,64,128,256,512")
    ap.add_argument("--nh", type=int, default=32)
    ap.add_argument("--nkv", type=int, default=8)
    ap.add_argument("--kv-dtypes", default="bf16,fp8")
    ap.add_argument("--splits", default="", help="comma list to sweep instead of the planner's choice")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/prefill_split.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hd, bs = 128, 16
    rows_out = []
    for kvd in a.kv_dtypes.split(","):
        for n in (int(v) for v in a.new.split(",")):
            total = a.prefix + n
            nblk = (total + bs - 1) // bs
            if kvd == "fp8":
                kc = (torch.randn(nblk, a.nkv, bs, hd, device=dev) * 0.5).to(torch.float8_e4m3fn)
                vc = (torch.randn(nblk, a.nkv, bs, hd, device=dev) * 0.5).to(torch.float8_e4m3fn)
                kw = {"kv_scale": torch.ones(2, a.nkv, device=dev)}
            else:
                kc = (torch.randn(nblk, a.nkv, bs, hd, device=dev) * 0.5).bfloat16()
                vc = (torch.randn(nblk, a.nkv, bs, hd, device=dev) * 0.5).bfloat16()
                kw = {}
            bt = torch.arange(nblk, device=dev, dtype=torch.int32).view(1, nblk)
            q = torch.randn(n, a.nh, hd, device=dev).bfloat16()
            tiles = [(off, min(16, n - off), a.prefix + off, 0) for off in range(0, n, 16)]
            tt = torch.tensor(tiles, device=dev, dtype=torch.int32).T.contiguous()
            out0 = torch.zeros(n, a.nh * hd, device=dev, dtype=torch.bfloat16)
            out1 = torch.zeros_like(out0)
            scale = 1 / math.sqrt(hd)
            t_one = timed(lambda: hip_ops.prefill_attention(q, kc, vc, bt, tt, out0, scale, **kw), a.reps)
            plan = hip_ops.prefill_split_plan(len(tiles), a.nh, a.nkv, total)
            for ns in ([int(v) for v in a.splits.split(",")] if a.splits else [plan]):
                if ns < 1:
                    continue
                t_split = timed(lambda: hip_ops.prefill_attention_split(q, kc, vc, bt, tt, len(tiles), out1, scale, ns, **kw), a.reps)
                err = (out0.float() - out1.float()).abs().max().item()
                flops = 4.0 * n * a.nh * hd * (a.prefix + (n + 1) / 2)
                kv_bytes = 2.0 * total * a.nkv * hd * kc.element_size()
                row = {"kv": kvd, "prefix": a.prefix, "new_tokens": n, "tiles": len(tiles), "n_split": ns, "planned": ns == plan, "one_pass_us": round(t_one, 1), "split_us": round(t_split, 1),
                       "speedup": round(t_one / t_split, 2), "split_TFLOPs": round(flops / t_split / 1e6, 1), "split_kv_TBps": round(kv_bytes / t_split / 1e6, 2), "max_abs_diff": err}
                print(json.dumps(row), flush=True)
                rows_out.append(row)
            del kc, vc
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"bench": "prefill_split", "heads": a.nh, "kv_heads": a.nkv, "rows": rows_out}, f, indent=1)


if __name__ == "__main__":
    main()
