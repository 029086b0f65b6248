#!/usr/bin/env python3
"""Decode step time against context length at a fixed batch: N sequences with ctx-token prompts, 24 decode steps timed
after the prompts are in (engine steps, launch-ahead off the clock: device time per step from events).

    python3 scripts/decode_step_vs_context.py --batch 32 --ctx 256,1024,4096 --kv auto,fp8
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="32")
    ap.add_argument("--ctx", default="256,1024,4096")
    ap.add_argument("--kv", default="auto,fp8")
    ap.add_argument("--out", default="gpurun_out/decode_step_vs_context.json")
    a = ap.parse_args()
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    res = []
    for kv in a.kv.split(","):
        for batch in (int(v) for v in a.batch.split(",")):
            for ctx in (int(v) for v in a.ctx.split(",")):
                extra = batch * (ctx // 2048 + 2)  # the first sequences decode while the later prompts are fed: enough tokens for all to be running when timed
                eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=max(32, batch), max_batched_tokens=2048, max_model_len=ctx + 128 + extra, kernels="hip",
                                             kv_memory_fraction=0.3, kv_cache_dtype=kv, enable_prefix_caching=False))
                eng.warmup()
                seqs = [eng.make_sequence([1] + [(7 * i + 13 * b) % 5000 + 10 for i in range(ctx - 1)], SamplingParams(max_tokens=64 + extra, ignore_eos=True)) for b in range(batch)]
                for s in seqs:
                    assert eng.scheduler.add(s) is None
                while any(s.first_token_time is None for s in seqs):
                    eng.step()
                for _ in range(8):
                    eng.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                while n < 24 and eng.scheduler.has_work():
                    eng.step()
                    n += 1
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / max(1, n) * 1e3
                running = len(eng.scheduler.running)
                mean_ctx = sum(s.num_tokens for s in eng.scheduler.running) / max(1, running)
                per_tok = eng.runner.cfg.num_layers * 2 * eng.runner.nkv * eng.runner.hd * eng.runner.kv_cache.element_size()
                kv_gb = running * mean_ctx * per_tok / 1e9
                row = {"kv": kv, "batch": batch, "ctx": ctx, "running": running, "mean_ctx": round(mean_ctx), "ms_per_step": round(ms, 3), "kv_GB_per_step": round(kv_gb, 2), "weights_GB": round(eng.weights.nbytes() / 1e9, 2)}
                print(json.dumps(row), flush=True)
                res.append(row)
                del eng, seqs
                torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
