"""Persistent load-generator child process.

``bench.py`` keeps the model replica and its HTTP front-end in the rank process
and drives load from a *separate* process (so the generator's Python work never
competes for the engine's GIL).  Starting an interpreter per pass would put
~0.5 s of startup into every timed step, so the child stays alive and executes
one JSON command per stdin line::

    {"cmd": "run", "url": ..., "requests": 500, "concurrency": 20, "max_tokens": 64, "run_dir": ..., ...}
    {"cmd": "quit"}

and answers each with one JSON line on stdout (per-request arrays inline so the
parent can all-gather them without touching the filesystem).
"""

from __future__ import annotations

import asyncio
import json
import sys
from typing import Any, Dict

from .protocols import RequestSpec
from .runner import LoadConfig, run_load


def _run(cmd: Dict[str, Any]) -> Dict[str, Any]:
    spec = RequestSpec(
        model=cmd.get("model", "placeholder"),
        prompt=cmd.get("prompt", "Hello, world!"),
        max_tokens=int(cmd.get("max_tokens", 64)),
        stream=bool(cmd.get("stream", True)),
        temperature=cmd.get("temperature"),
    )
    cfg = LoadConfig(
        url=cmd["url"],
        run_dir=cmd.get("run_dir"),
        requests=int(cmd.get("requests", 500)),
        concurrency=int(cmd.get("concurrency", 20)),
        pattern=cmd.get("pattern", "steady"),
        target_rps=cmd.get("target_rps"),
        protocol=cmd.get("protocol", "openai"),
        spec=spec,
        seed=cmd.get("seed", 0),
        trace=bool(cmd.get("trace", False)),
        token_events=bool(cmd.get("token_events", False)),
        quiet=True,
        timeout_s=float(cmd.get("timeout_s", 120.0)),
    )
    res = asyncio.run(run_load(cfg))
    gaps = []
    for r in res:
        tt = r.token_times_ms
        gaps.extend(b - a for a, b in zip(tt, tt[1:]))
    return {
        "ok": True,
        "n": len(res),
        "status": [r.status for r in res],
        "start_ms": [r.start_ms for r in res],
        "latency_ms": [r.latency_ms for r in res],
        "ttfb_ms": [r.ttfb_ms for r in res],
        "ttft_ms": [r.ttft_ms for r in res],
        "tllt_ms": [r.tllt_ms for r in res],
        "completion_tokens": [r.completion_tokens or 0 for r in res],
        "total_tokens": [r.total_tokens or 0 for r in res],
        "itl_ms": gaps if cmd.get("return_itl", True) else [],
        "errors": [r.error for r in res if r.error][:5],
    }


def main() -> int:
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            cmd = json.loads(line)
        except ValueError as e:
            print(json.dumps({"ok": False, "error": f"bad command: {e}"}), flush=True)
            continue
        if cmd.get("cmd") == "quit":
            break
        if cmd.get("cmd") == "ping":
            print(json.dumps({"ok": True, "pong": True}), flush=True)
            continue
        try:
            out = _run(cmd)
        except Exception as e:
            out = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
