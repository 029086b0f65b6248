"""OpenAI API parity probe (``kvmini parity``).

Five live probes against ``/v1/chat/completions`` — function calling, parallel
tool calls (>= 2 in one completion), JSON mode (content must parse), logprobs,
streaming (>= 2 chunks and first chunk within 5 s) — aggregated into
``capability_matrix.json`` with the same shape the reference writes
(``scripts/openai_parity_probe.py:250-275``) plus an optional HTML page.
Uses the in-tree asyncio HTTP client, so no third-party HTTP stack is needed.
"""

from __future__ import annotations

import argparse
import asyncio
import html
import json
import sys
import time
from dataclasses import asdict, dataclass
from typing import Any, Dict, List, Optional, Tuple

from kvmini.loadgen.httpclient import HttpClient
from kvmini.loadgen.sse import SSEParser
from kvmini.utils import write_json

CAPS = (
    ("tools_function_calling", "Tools / function calling"),
    ("parallel_tool_calls", "Parallel tool calls"),
    ("json_mode", "JSON mode"),
    ("logprobs", "Logprobs"),
    ("streaming", "Streaming"),
)


@dataclass
class ProbeResult:
    supported: bool
    details: Optional[str] = None


def _fn(name: str, props: Dict[str, Any], required: Optional[List[str]] = None, desc: Optional[str] = None) -> Dict[str, Any]:
    params: Dict[str, Any] = {"type": "object", "properties": props}
    if required:
        params["required"] = required
    f: Dict[str, Any] = {"name": name, "parameters": params}
    if desc:
        f["description"] = desc
    return {"type": "function", "function": f}


class ParityProber:
    def __init__(self, base_url: str, model: str, api_key: Optional[str] = None, insecure: bool = False, timeout_s: float = 60.0):
        self.base_url = base_url.rstrip("/")
        self.model = model
        self.api_key = api_key
        self.timeout_s = timeout_s
        self.client = HttpClient(pooled=True, verify_tls=not insecure)

    def _headers(self) -> Dict[str, str]:
        h = {"Content-Type": "application/json"}
        if self.api_key:
            h["Authorization"] = f"Bearer {self.api_key}"
        return h

    async def _chat(self, payload: Dict[str, Any]) -> Tuple[int, Dict[str, Any]]:
        body = {"model": self.model, **payload}
        resp = await self.client.post(self.base_url + "/v1/chat/completions", headers=self._headers(), body=json.dumps(body).encode(), timeout=self.timeout_s)
        raw = await resp.read()
        try:
            data = json.loads(raw) if raw else {}
        except ValueError:
            data = {}
        return resp.status, data if isinstance(data, dict) else {}

    @staticmethod
    def _choice(data: Dict[str, Any]) -> Dict[str, Any]:
        ch = data.get("choices") or [{}]
        return ch[0] if isinstance(ch[0], dict) else {}

    RETRIES = 6  # sampled re-asks when the model's own choice under "auto" did not show the capability

    async def _tool_calls(self, payload: Dict[str, Any], enough: int) -> Tuple[int, int, str]:
        """(status, most tool calls seen in one completion, which request showed them).  The first
        request is the reference's (``tool_choice: auto``, ``scripts/openai_parity_probe.py:98,147``).
        Under "auto" a server may let the MODEL decide between text and calls, and how many: a text
        answer there says nothing about the server, so the question is asked again with
        ``tool_choice: required`` (sampled, a few times) before the capability counts as absent."""
        status, data = await self._chat({**payload, "tool_choice": "auto"})
        calls = (self._choice(data).get("message") or {}).get("tool_calls")
        best, how = (len(calls), "auto") if status == 200 and isinstance(calls, list) else (-1, "auto")
        for k in range(self.RETRIES if status == 200 and best < enough else 0):
            st2, data = await self._chat({**payload, "tool_choice": "required", "temperature": 1.0, "max_tokens": 256})
            calls = (self._choice(data).get("message") or {}).get("tool_calls")
            if st2 == 200 and isinstance(calls, list) and len(calls) > best:
                best, how = len(calls), f"required, sampled (attempt {k + 1}); under auto the model chose otherwise"
            if best >= enough:
                break
        return status, best, how

    async def probe_tools(self) -> ProbeResult:
        status, best, how = await self._tool_calls(
            {
                "messages": [{"role": "user", "content": "What is the weather in San Francisco?"}],
                "tools": [_fn("get_current_weather", {"location": {"type": "string"}, "unit": {"type": "string", "enum": ["celsius", "fahrenheit"]}}, ["location"], "Get the current weather")],
            },
            enough=0,
        )
        ok = status == 200 and best >= 0
        return ProbeResult(ok, (None if how == "auto" else how) if ok else f"status={status}, no tool_calls list")

    async def probe_parallel_tools(self) -> ProbeResult:
        status, best, how = await self._tool_calls(
            {
                "messages": [{"role": "user", "content": "For NYC and SF, get weather and timezone."}],
                "tools": [_fn("get_weather", {"city": {"type": "string"}}), _fn("get_timezone", {"city": {"type": "string"}})],
            },
            enough=2,
        )
        if status != 200:
            return ProbeResult(False, f"status={status}")
        return ProbeResult(best >= 2, f"tool_calls={max(best, 0)}" + ("" if how == "auto" else f" ({how})"))

    async def probe_json_mode(self) -> ProbeResult:
        status, data = await self._chat({"messages": [{"role": "user", "content": "Return a JSON with keys city and temp."}], "response_format": {"type": "json_object"}})
        if status != 200:
            return ProbeResult(False, f"status={status}")
        content = (self._choice(data).get("message") or {}).get("content") or ""
        try:
            json.loads(content)
            return ProbeResult(True)
        except ValueError as e:
            return ProbeResult(False, f"not json: {e}")

    async def probe_logprobs(self) -> ProbeResult:
        status, data = await self._chat({"messages": [{"role": "user", "content": "Say: test"}], "max_tokens": 4, "logprobs": True})
        if status != 200:
            return ProbeResult(False, f"status={status}")
        ch = self._choice(data)
        has = bool(ch.get("logprobs") or ch.get("top_logprobs") or (ch.get("delta") or {}).get("logprobs"))
        return ProbeResult(has, None if has else "no logprobs in response")

    async def probe_streaming(self) -> ProbeResult:
        body = {"model": self.model, "messages": [{"role": "user", "content": "Stream 3 words."}], "max_tokens": 8, "stream": True}
        t0 = time.perf_counter()
        resp = await self.client.post(self.base_url + "/v1/chat/completions", headers=self._headers(), body=json.dumps(body).encode(), timeout=self.timeout_s)
        if resp.status != 200:
            await resp.read()
            return ProbeResult(False, f"status={resp.status}")
        parser = SSEParser("openai")
        first: Optional[float] = None
        chunks = 0
        async for piece in resp.iter_bytes():
            now = time.perf_counter()
            before = parser.n_events
            parser.feed(piece, now * 1000.0)
            if parser.n_events > before and first is None:
                first = now - t0
        chunks = parser.n_events - (1 if parser.done else 0)
        ok = chunks >= 2 and first is not None and first < 5.0
        return ProbeResult(ok, f"chunks={chunks}, ttft={first if first is not None else -1:.3f}s")

    async def run(self) -> Dict[str, Any]:
        out: Dict[str, ProbeResult] = {}
        probes = (self.probe_tools, self.probe_parallel_tools, self.probe_json_mode, self.probe_logprobs, self.probe_streaming)
        try:
            for (key, _), fn in zip(CAPS, probes):
                try:
                    out[key] = await fn()
                except Exception as e:  # connection errors count as unsupported
                    out[key] = ProbeResult(False, f"{type(e).__name__}: {e}")
        finally:
            await self.client.aclose()
        return {
            "endpoint": self.base_url,
            "model": self.model,
            "capabilities": {k: asdict(v) for k, v in out.items()},
            "summary": {"supported_count": sum(1 for v in out.values() if v.supported), "total": len(CAPS)},
        }


def render_html(res: Dict[str, Any]) -> str:
    rows = "".join(
        f"<tr><td>{html.escape(label)}</td><td class='{'ok' if res['capabilities'][k]['supported'] else 'no'}'>{'yes' if res['capabilities'][k]['supported'] else 'no'}</td><td>{html.escape(str(res['capabilities'][k].get('details') or ''))}</td></tr>"
        for k, label in CAPS
    )
    return (
        "<!DOCTYPE html><html><head><meta charset='utf-8'><title>Capability matrix</title>"
        "<style>body{font-family:system-ui,Arial;padding:20px}td,th{padding:4px 12px;border-bottom:1px solid #ddd;text-align:left}.ok{color:#1a7f37;font-weight:600}.no{color:#cf222e;font-weight:600}</style></head><body>"
        f"<h1>OpenAI parity capability matrix</h1><p><b>Endpoint:</b> {html.escape(res['endpoint'])} &nbsp; <b>Model:</b> {html.escape(res['model'])} &nbsp; "
        f"<b>{res['summary']['supported_count']}/{res['summary']['total']}</b> supported</p><table><tr><th>capability</th><th>supported</th><th>details</th></tr>{rows}</table></body></html>"
    )


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini parity", description="OpenAI API parity probe")
    ap.add_argument("--url", required=True, help="Base URL of an OpenAI-compatible endpoint")
    ap.add_argument("--model", required=True)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--insecure", action="store_true")
    ap.add_argument("--out", default="capability_matrix.json")
    ap.add_argument("--html", default=None)
    a = ap.parse_args(argv)
    res = asyncio.run(ParityProber(a.url, a.model, a.api_key, a.insecure).run())
    write_json(a.out, res)
    print(f"Capability matrix written to {a.out} ({res['summary']['supported_count']}/{res['summary']['total']} supported)")
    if a.html:
        with open(a.html, "w") as f:
            f.write(render_html(res))
        print(f"HTML report written to {a.html}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
