#!/usr/bin/env python3
"""Chat continuation behind a long cached conversation (Llama-3-8B): PREFIX tokens are in the prefix cache and every
request adds a turn of NEW tokens.  Prints the time to the turn's first token with the tile kernel walking the whole
context per tile and with its split form (first use eager, second captures a graph, later ones replay).  GPU box.

  python scripts/chat_continuation_step.py --prefix 131072 --new 64,256 --out gpurun_out/chat_continuation.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefix", type=int, default=131072)
    ap.add_argument("--new", default="64,256")
    ap.add_argument("--kv-cache-dtype", default="auto")
    ap.add_argument("--mark", action="store_true", help="bracket the last split turn with two arange launches (for scripts/trace_between_markers.py)")
    ap.add_argument("--out", default="gpurun_out/chat_continuation.json")
    a = ap.parse_args()
    eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda:0", max_num_seqs=4, max_batched_tokens=2048, max_model_len=a.prefix + 2048, kernels="hip",
                                 kv_cache_dtype=a.kv_cache_dtype, kv_memory_fraction=0.25))
    eng.warmup()
    history = [1] + [(13 * j) % 5000 + 10 for j in range(a.prefix - 1)]
    t0 = time.perf_counter()
    eng.generate([history], SamplingParams(max_tokens=1, ignore_eos=True))
    t_prefill = time.perf_counter() - t0
    print(f"prefill of {a.prefix} tokens: {t_prefill:.2f} s ({a.prefix / t_prefill / 1e3:.1f} k tok/s)", flush=True)
    rows = []
    turn = 0

    def first_token_ms(n_new, mark=False):
        nonlocal turn
        turn += 1
        prompt = history + [(17 * j + 101 * turn) % 5000 + 10 for j in range(n_new)]
        torch.cuda.synchronize()
        if mark:
            torch.arange(0, 4099, device="cuda:0")
        t0 = time.perf_counter()
        eng.generate([prompt], SamplingParams(max_tokens=1, ignore_eos=True))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if mark:
            torch.arange(0, 4099, device="cuda:0")
            torch.cuda.synchronize()
        return dt

    for n_new in (int(v) for v in a.new.split(",")):
        eng.runner.pf_split_ok = False
        one = [first_token_ms(n_new) for _ in range(2)]
        eng.runner.pf_split_ok = True
        g0 = eng.runner.variant_graph_steps
        split = [first_token_ms(n_new) for _ in range(3)] + [first_token_ms(n_new, mark=a.mark)]
        row = {"prefix": a.prefix, "new_tokens": n_new, "kv": a.kv_cache_dtype, "one_pass_ms": [round(v, 2) for v in one], "split_ms": [round(v, 2) for v in split],
               "graph_replays": eng.runner.variant_graph_steps - g0, "prefix_hits": eng.bm.prefix_hits}
        print(json.dumps(row), flush=True)
        rows.append(row)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"bench": "chat_continuation", "model": "llama-3-8b", "prefill_s": round(t_prefill, 2), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
