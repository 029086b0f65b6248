#!/usr/bin/env python3
"""Decode tokens that ride along in a step with prompt chunks: per-layer attention time of the
per-token kernel (one workgroup per token and kv head) against the split-KV kernel over listed
rows, Llama-3-8B head shapes (32 query / 8 kv heads of 128), one JSON line per case.

    python scripts/dec_split_bench.py [--ctx 8192,32768,131072] [--rows 1,4,16] [--kv8]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", default="8192,32768,131072")
    ap.add_argument("--rows", default="1,4,16")
    ap.add_argument("--step-rows", type=int, default=2048, help="token rows of the step the decode tokens sit in")
    ap.add_argument("--kv8", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from kvmini.ops import hip_ops

    dev = torch.device("cuda", 0)
    nh, nkv, hd, bs = 32, 8, 128, 16
    T = args.step_rows
    scale = 1 / math.sqrt(hd)
    for ctx_len in (int(c) for c in args.ctx.split(",")):
        for n_rows in (int(r) for r in args.rows.split(",")):
            max_blocks = (ctx_len + bs - 1) // bs
            nb = n_rows * max_blocks + 8
            if args.kv8:
                kc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).to(torch.float8_e4m3fn)
                vc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).to(torch.float8_e4m3fn)
                sc = torch.ones(2, nkv, device=dev)
            else:
                kc = torch.randn(nb, nkv, bs, hd, device=dev).bfloat16()
                vc = torch.randn(nb, nkv, bs, hd, device=dev).bfloat16()
                sc = None
            bt = torch.randperm(nb, device=dev)[: n_rows * max_blocks].view(n_rows, max_blocks).int()
            live = torch.arange(n_rows, device=dev)  # consecutive rows, as a step stages its one-token chunks (rows spaced 2^k apart would put the per-token kernel's live workgroups on one XCD)
            ctx = torch.full((T,), -1, device=dev, dtype=torch.int32)
            ctx[live] = ctx_len
            tok_seq = torch.zeros(T, device=dev, dtype=torch.int32)
            tok_seq[live] = torch.arange(n_rows, device=dev, dtype=torch.int32)
            q = torch.randn(T, nh, hd, device=dev).bfloat16()
            rows = torch.full((1 << (n_rows - 1).bit_length(),), -1, device=dev, dtype=torch.int32)
            rows[:n_rows] = live.int()
            out = torch.zeros(T, nh * hd, device=dev, dtype=torch.bfloat16)

            def timed(fn) -> float:
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / args.iters * 1e3

            t_tok = timed(lambda: hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, max_blocks, sc))
            t_split = timed(lambda: hip_ops.split_decode_attention(q, kc, vc, bt, tok_seq, ctx, rows, out, scale, kv_scale=sc))
            ref = hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, scale, max_blocks, sc)
            err = (out[live].float() - ref[live].float()).abs().max().item()
            kv_bytes = n_rows * ctx_len * nkv * hd * 2 * (1 if args.kv8 else 2)
            print(json.dumps({"ctx": ctx_len, "decode_rows": n_rows, "step_rows": T, "kv": "e4m3" if args.kv8 else "bf16", "per_token_kernel_us": round(t_tok, 1),
                              "split_kernel_us": round(t_split, 1), "speedup": round(t_tok / t_split, 2), "split_tb_per_s": round(kv_bytes / t_split / 1e6, 2),
                              "splits": hip_ops.decode_attention_splits(rows.numel(), nkv), "max_abs_diff": round(err, 5)}), flush=True)
            del kc, vc
    return 0


if __name__ == "__main__":
    sys.exit(main())
