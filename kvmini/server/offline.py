"""``kvmini generate``: offline batch generation on the native engine — no server, no HTTP.

Prompts come from ``--prompt`` (repeatable) and / or ``--prompts-file`` (one prompt per line, or JSON lines
with ``{"prompt": ...}`` or ``{"messages": [...]}``); all of them go through one engine at once (continuous
batching, prefix cache, HIP graphs — the serving path, minus the front-end) and the replies are printed or
written as JSON lines together with the token counts and the run's throughput.  The engine flags are the
server's (``kvmini serve``): ``--model`` takes a preset (random-init weights) or a Hugging Face checkpoint
directory, whose tokenizer and chat template are used.  The reference has no offline mode (its harness only
talks to deployed endpoints, ``scripts/loadtest.py:1-60``); this is for checking a checkpoint or timing the
engine without the HTTP path.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Any, Dict, List, Optional

from kvmini.engine.engine import LLMEngine
from kvmini.engine.scheduler import SamplingParams
from kvmini.engine.tokenizer import get_tokenizer


def read_prompts(a: argparse.Namespace) -> List[Dict[str, Any]]:
    items: List[Dict[str, Any]] = [{"prompt": p} for p in (a.prompt or [])]
    if a.prompts_file:
        with (sys.stdin if a.prompts_file == "-" else open(a.prompts_file)) as f:
            for ln in f:
                ln = ln.rstrip("\n")
                if not ln.strip():
                    continue
                obj: Any = None
                if ln.lstrip().startswith("{"):
                    try:
                        obj = json.loads(ln)
                    except ValueError:
                        obj = None
                if isinstance(obj, dict) and ("prompt" in obj or "messages" in obj):
                    items.append(obj)
                else:
                    items.append({"prompt": ln})
    return items


def main(argv: Optional[List[str]] = None) -> int:
    from kvmini.server.native import add_engine_args, engine_config_from_args

    ap = argparse.ArgumentParser(prog="kvmini generate", description="Offline batch generation on the native engine")
    ap.add_argument("--prompt", action="append", help="a prompt (repeatable)")
    ap.add_argument("--prompts-file", default=None, help="one prompt per line, or JSON lines with 'prompt' or 'messages'; '-' = stdin")
    ap.add_argument("--chat", action="store_true", help="wrap plain prompts as one user message in the model's chat template")
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--min-tokens", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=0.0)
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--top-k", type=int, default=0)
    ap.add_argument("--sampling-seed", type=int, default=None, help="per-request seed (request i gets seed + i)")
    ap.add_argument("--presence-penalty", type=float, default=0.0)
    ap.add_argument("--frequency-penalty", type=float, default=0.0)
    ap.add_argument("--repetition-penalty", type=float, default=1.0)
    ap.add_argument("--out", default=None, help="write JSON lines here instead of printing the replies")
    add_engine_args(ap)
    a = ap.parse_args(argv)
    if a.tensor_parallel_size > 1:
        raise SystemExit("kvmini generate runs one GPU; tensor-parallel models are served with `kvmini serve --tensor-parallel-size N`")
    items = read_prompts(a)
    if not items:
        raise SystemExit("no prompts: give --prompt and / or --prompts-file")
    t_load = time.perf_counter()
    engine = LLMEngine(engine_config_from_args(a))
    cfg = engine.mcfg
    tok_path = a.tokenizer
    if tok_path is None and os.path.isfile(os.path.join(a.model, "tokenizer.json")):
        tok_path = os.path.join(a.model, "tokenizer.json")
    tok = get_tokenizer(cfg.vocab_size, cfg.bos_token_id, cfg.eos_token_id, tok_path)
    engine.warmup()
    t_load = time.perf_counter() - t_load
    ignore_eos = not a.honor_eos
    stops = [] if ignore_eos else [t for t in getattr(cfg, "extra_eos_token_ids", ())]
    ids: List[List[int]] = []
    params: List[SamplingParams] = []
    for i, it in enumerate(items):
        if "messages" in it:
            ids.append(tok.apply_chat_template(it["messages"]))
        elif a.chat:
            ids.append(tok.apply_chat_template([{"role": "user", "content": str(it["prompt"])}]))
        else:
            ids.append(tok.encode(str(it["prompt"]), add_bos=True))
        params.append(SamplingParams(
            max_tokens=int(it.get("max_tokens", a.max_tokens)), min_tokens=a.min_tokens, temperature=a.temperature, top_p=a.top_p, top_k=a.top_k,
            seed=None if a.sampling_seed is None else a.sampling_seed + i, ignore_eos=ignore_eos, stop_token_ids=list(stops),
            presence_penalty=a.presence_penalty, frequency_penalty=a.frequency_penalty, repetition_penalty=a.repetition_penalty))
        bad = engine.scheduler.validate(engine.make_sequence(ids[-1], params[-1]))
        if bad:
            raise SystemExit(f"prompt {i}: {bad}")
    t0 = time.perf_counter()
    outs = engine.generate(ids, params)
    dt = time.perf_counter() - t0
    eos = {cfg.eos_token_id, *stops}
    sink = open(a.out, "w") if a.out else None
    n_out = 0
    for i, (it, p, o) in enumerate(zip(items, ids, outs)):
        n_out += len(o)
        shown = o[:-1] if (o and not ignore_eos and o[-1] in eos) else o  # the end-of-sequence token itself is not text
        text = tok.decode(shown)
        if sink is not None:
            sink.write(json.dumps({"index": i, "prompt_tokens": len(p), "completion_tokens": len(o), "text": text, "token_ids": o}) + "\n")
        else:
            print(f"--- [{i}] {len(p)} prompt tokens, {len(o)} generated\n{text}")
    if sink is not None:
        sink.close()
    st = engine.metrics()
    summary = {"requests": len(items), "prompt_tokens": sum(len(p) for p in ids), "completion_tokens": n_out, "seconds": round(dt, 4),
               "output_tokens_per_s": round(n_out / dt, 1) if dt > 0 else None, "engine_steps": st["steps"], "step_ms_ewma": round(st["step_ms_ewma"], 3), "prefix_cache_hits": st["prefix_cache_hits"],
               "load_and_warmup_s": round(t_load, 2), "model": cfg.name, "device": str(engine.device)}
    print(json.dumps(summary), file=sys.stderr if sink is None else sys.stdout)
    engine.stop()
    return 0


if __name__ == "__main__":
    sys.exit(main())
