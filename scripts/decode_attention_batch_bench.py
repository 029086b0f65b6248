#!/usr/bin/env python3
"""Decode attention at large batch: N sequences x ctx tokens of KV per layer, microseconds and TB/s of K/V read.

    python3 scripts/decode_attention_batch_bench.py --out gpurun_out/decode_attention_batch.json
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", default="64,128,256")
    ap.add_argument("--ctx", default="128,256,512,1024,2048")
    ap.add_argument("--out", default="gpurun_out/decode_attention_batch.json")
    ap.add_argument("--sequential", action="store_true", help="consecutive blocks per sequence (a fresh engine) instead of a random permutation")
    a = ap.parse_args()
    from kvmini.ops import hip_ops

    dev = torch.device("cuda:0")
    nh, nkv, hd, bs = 32, 8, 128, 16
    res = []
    for nseq in (int(v) for v in a.seqs.split(",")):
        for ctx in (int(v) for v in a.ctx.split(",")):
            per = (ctx + bs - 1) // bs
            nb = nseq * per
            for kv8 in (False, True):
                kc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
                vc = (torch.randn(nb, nkv, bs, hd, device=dev) * 0.5).bfloat16()
                if kv8:
                    kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
                scale = torch.ones(2, nkv, device=dev) if kv8 else None
                q = torch.randn(nseq, nh, hd, device=dev).bfloat16()
                tok_seq = torch.arange(nseq, device=dev, dtype=torch.int32)
                ctx_lens = torch.full((nseq,), ctx, device=dev, dtype=torch.int32)
                bt = (torch.arange(nb, device=dev) if a.sequential else torch.randperm(nb, device=dev)).view(nseq, per).int().contiguous()
                kw = dict(kv_scale=scale) if kv8 else {}
                fn = lambda: hip_ops.paged_attention(q, kc, vc, bt, tok_seq, ctx_lens, 1 / math.sqrt(hd), **kw)
                # the same rows as one-query tiles of the MFMA prompt-tile kernel
                tiles = torch.stack([torch.arange(nseq), torch.ones(nseq, dtype=torch.int64), torch.full((nseq,), ctx - 1), torch.arange(nseq)]).int().to(dev).contiguous()
                out_t = torch.zeros(nseq, nh * hd, dtype=torch.bfloat16, device=dev)
                fn_t = lambda: hip_ops.prefill_attention(q, kc, vc, bt, tiles, out_t, 1 / math.sqrt(hd), kv_scale=scale)
                ref = fn().float().view(nseq, -1)
                fn_t()
                torch.cuda.synchronize()
                err = (out_t.float() - ref).abs().max().item()
                for _ in range(5):
                    fn_t()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn_t()
                e1.record()
                torch.cuda.synchronize()
                us_t = e0.elapsed_time(e1) / 20 * 1e3
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 20 * 1e3
                gb = 2.0 * nseq * ctx * nkv * hd * kc.element_size() / 1e9
                row = {"seqs": nseq, "ctx": ctx, "kv": "e4m3" if kv8 else "bf16", "us": round(us, 1), "tiles_us": round(us_t, 1), "tiles_err": round(err, 4), "kv_GB": round(gb, 3), "TBps": round(gb / us * 1e3, 2)}
                print(json.dumps(row), flush=True)
                res.append(row)
                del kc, vc
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
