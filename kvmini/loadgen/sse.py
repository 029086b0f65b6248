"""Incremental Server-Sent-Events parser with per-token timestamps.

The reference treats the first *HTTP body chunk* as the first token and never
parses ``data:`` frames, so usage is never found in real SSE streams
(``scripts/loadtest.py:325-331,468-478``).  This parser frames the byte stream
properly (``\\n\\n`` / ``\\r\\n\\r\\n`` event boundaries, multi-line ``data:``,
``[DONE]`` sentinel, comments) and records, against the caller's clock:

* ``first_byte_ms``     — first body byte (the reference's "TTFB/TTFT"),
* ``first_token_ms``    — first event that carries generated content,
* ``last_token_ms``     — last content event (TLLT),
* ``token_times_ms``    — one timestamp per content event (for inter-token latency),
* ``usage``             — the OpenAI usage object when the server sends one
  (requested with ``stream_options.include_usage``).

It understands three wire dialects: OpenAI chat/completions chunks, TGI
``/generate_stream`` token events and KServe-v2 / Triton ``generate_stream``
``text_output`` events.
"""

from __future__ import annotations

import json
from typing import Any, Callable, Dict, List, Optional


def _openai_content(ev: Dict[str, Any]) -> int:
    """Number of content-bearing deltas in an OpenAI chunk (0 for role-only /
    usage-only chunks).  Tool-call argument deltas count as content."""
    n = 0
    for ch in ev.get("choices") or []:
        delta = ch.get("delta") or {}
        if delta.get("content") or delta.get("reasoning_content") or delta.get("tool_calls"):
            n += 1
        elif ch.get("text"):  # /v1/completions
            n += 1
    return n


def _tgi_content(ev: Dict[str, Any]) -> int:
    tok = ev.get("token")
    if isinstance(tok, dict) and not tok.get("special", False):
        return 1
    return 0


def _kserve_v2_content(ev: Dict[str, Any]) -> int:
    if ev.get("text_output"):
        return 1
    for out in ev.get("outputs") or []:
        if out.get("name") == "text_output" and out.get("data"):
            return 1
    return 0


CONTENT_COUNTERS: Dict[str, Callable[[Dict[str, Any]], int]] = {
    "openai": _openai_content,
    "tgi": _tgi_content,
    "kserve-v2": _kserve_v2_content,
}


class SSEParser:
    def __init__(self, dialect: str = "openai", keep_events: bool = False, keep_text: bool = True):
        self._buf = bytearray()
        self.dialect = dialect
        self._count = CONTENT_COUNTERS[dialect]
        self.keep_events = keep_events
        self.keep_text = keep_text
        self.events: List[Dict[str, Any]] = []
        self.first_byte_ms: Optional[float] = None
        self.first_token_ms: Optional[float] = None
        self.last_token_ms: Optional[float] = None
        self.last_byte_ms: Optional[float] = None
        self.token_times_ms: List[float] = []
        self.usage: Optional[Dict[str, Any]] = None
        self.done = False
        self.n_events = 0
        self.n_chunks = 0
        self.text_parts: List[str] = []
        self.finish_reason: Optional[str] = None
        self.tool_calls = 0
        self.generated_text: Optional[str] = None  # TGI final event
        self.parse_errors = 0
        # called as on_token(index, t_ms) for every content event (live inter-token statistics)
        self.on_token: Optional[Callable[[int, float], None]] = None

    def feed(self, data: bytes, t_ms: float) -> None:
        if not data:
            return
        self.n_chunks += 1
        if self.first_byte_ms is None:
            self.first_byte_ms = t_ms
        self.last_byte_ms = t_ms
        self._buf += data
        while True:
            # event boundary: blank line (LF LF or CRLF CRLF)
            i = self._buf.find(b"\n\n")
            j = self._buf.find(b"\r\n\r\n")
            if i < 0 and j < 0:
                break
            if j >= 0 and (i < 0 or j < i):
                raw, adv = bytes(self._buf[:j]), j + 4
            else:
                raw, adv = bytes(self._buf[:i]), i + 2
            del self._buf[:adv]
            self._event(raw, t_ms)

    def close(self, t_ms: float) -> None:
        """Flush a trailing event that was not terminated by a blank line."""
        if self._buf.strip():
            raw = bytes(self._buf)
            self._buf.clear()
            self._event(raw, t_ms)

    def _event(self, raw: bytes, t_ms: float) -> None:
        datas: List[str] = []
        for line in raw.splitlines():
            if not line or line.startswith(b":"):
                continue
            if line.startswith(b"data:"):
                datas.append(line[5:].lstrip(b" ").decode("utf-8", "replace"))
            elif line.startswith(b"{") and not datas:
                # bare NDJSON line (some servers stream without the data: prefix)
                datas.append(line.decode("utf-8", "replace"))
        if not datas:
            return
        payload = "\n".join(datas)
        self.n_events += 1
        if payload.strip() == "[DONE]":
            self.done = True
            return
        try:
            ev = json.loads(payload)
        except ValueError:
            self.parse_errors += 1
            return
        if not isinstance(ev, dict):
            return
        if self.keep_events:
            self.events.append(ev)
        if isinstance(ev.get("usage"), dict):
            self.usage = ev["usage"]
        n = self._count(ev)
        if n:
            if self.first_token_ms is None:
                self.first_token_ms = t_ms
            self.last_token_ms = t_ms
            self.token_times_ms.append(t_ms)
            if self.on_token is not None:
                self.on_token(len(self.token_times_ms) - 1, t_ms)
        self._harvest(ev)

    def _harvest(self, ev: Dict[str, Any]) -> None:
        if self.dialect == "openai":
            for ch in ev.get("choices") or []:
                delta = ch.get("delta") or {}
                if self.keep_text:
                    c = delta.get("content") or ch.get("text")
                    if c:
                        self.text_parts.append(c)
                if delta.get("tool_calls"):
                    for tc in delta["tool_calls"]:
                        if (tc.get("function") or {}).get("name"):
                            self.tool_calls += 1
                if ch.get("finish_reason"):
                    self.finish_reason = ch["finish_reason"]
        elif self.dialect == "tgi":
            tok = ev.get("token") or {}
            if self.keep_text and tok.get("text") and not tok.get("special"):
                self.text_parts.append(tok["text"])
            if ev.get("generated_text") is not None:
                self.generated_text = ev["generated_text"]
                det = ev.get("details") or {}
                self.finish_reason = det.get("finish_reason", self.finish_reason)
                if det.get("generated_tokens") is not None:
                    self.usage = {"completion_tokens": det["generated_tokens"]}
        else:
            if self.keep_text and ev.get("text_output"):
                self.text_parts.append(ev["text_output"])

    @property
    def text(self) -> str:
        return "".join(self.text_parts)

    @property
    def content_events(self) -> int:
        return len(self.token_times_ms)
