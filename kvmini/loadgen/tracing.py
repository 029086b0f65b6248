"""Client-side request-lifecycle tracing, exported as OTLP-shaped JSON.

Span tree per request (same operation names as the reference,
``scripts/loadtest.py:360-462``): ``client.request`` -> ``client.wait_scheduled``,
``http.request`` -> ``server.ttft``, ``server.tllt``.  A W3C ``traceparent`` header
(``00-<trace>-<span>-01``) is sent with every request so a server-side tracer can
join the trace.  Attribute typing tests ``bool`` before ``int`` (``bool`` is an
``int`` subclass, which makes the reference's bool branch unreachable:
``scripts/loadtest.py:157-164``); integers are exported as ``intValue``.
"""

from __future__ import annotations

import json
import os
import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

from kvmini.utils import now_ms

_rng = random.Random()


def generate_trace_id() -> str:
    return f"{_rng.getrandbits(128):032x}"


def generate_span_id() -> str:
    return f"{_rng.getrandbits(64):016x}"


def create_traceparent_header(trace_id: str, span_id: str) -> str:
    return f"00-{trace_id}-{span_id}-01"


@dataclass
class TraceSpan:
    trace_id: str
    span_id: str
    parent_span_id: Optional[str]
    operation_name: str
    start_time: float  # ms
    end_time: Optional[float] = None
    status: str = "ok"
    attributes: Dict[str, Any] = field(default_factory=dict)


def _otlp_value(v: Any) -> Dict[str, Any]:
    if isinstance(v, bool):
        return {"boolValue": v}
    if isinstance(v, int):
        return {"intValue": str(v)}  # OTLP/JSON encodes int64 as string
    if isinstance(v, float):
        return {"doubleValue": v}
    if isinstance(v, str):
        return {"stringValue": v}
    return {"stringValue": str(v)}


class TraceCollector:
    def __init__(self, service_name: str = "kvmini-loadgen", service_version: str = "0.1.0"):
        self.spans: List[TraceSpan] = []
        self.service_name = service_name
        self.service_version = service_version

    def start_span(
        self,
        trace_id: str,
        operation_name: str,
        parent_span_id: Optional[str] = None,
        attributes: Optional[Dict[str, Any]] = None,
        start_time: Optional[float] = None,
    ) -> TraceSpan:
        span = TraceSpan(trace_id, generate_span_id(), parent_span_id, operation_name, now_ms() if start_time is None else start_time, attributes=dict(attributes or {}))
        self.spans.append(span)
        return span

    def finish_span(self, span: TraceSpan, status: str = "ok", attributes: Optional[Dict[str, Any]] = None, end_time: Optional[float] = None) -> None:
        span.end_time = now_ms() if end_time is None else end_time
        span.status = status
        if attributes:
            span.attributes.update(attributes)

    def to_otlp(self) -> Dict[str, Any]:
        spans = []
        for s in self.spans:
            o: Dict[str, Any] = {
                "traceId": s.trace_id,
                "spanId": s.span_id,
                "name": s.operation_name,
                "startTimeUnixNano": str(int(s.start_time * 1_000_000)),
                "endTimeUnixNano": str(int((s.end_time if s.end_time is not None else s.start_time) * 1_000_000)),
                "attributes": [{"key": k, "value": _otlp_value(v)} for k, v in s.attributes.items()],
                "status": {"code": 1 if s.status == "ok" else 2},
            }
            if s.parent_span_id:
                o["parentSpanId"] = s.parent_span_id
            spans.append(o)
        return {
            "resourceSpans": [
                {
                    "resource": {
                        "attributes": [
                            {"key": "service.name", "value": {"stringValue": self.service_name}},
                            {"key": "service.version", "value": {"stringValue": self.service_version}},
                        ]
                    },
                    "scopeSpans": [{"scope": {"name": "kvmini", "version": self.service_version}, "spans": spans}],
                }
            ]
        }

    def export_traces(self, output_file: str) -> None:
        os.makedirs(os.path.dirname(output_file) or ".", exist_ok=True)
        with open(output_file, "w") as f:
            json.dump(self.to_otlp(), f)

    def export_http(self, endpoint: str, timeout_s: float = 10.0, batch: int = 2000, headers: Optional[Dict[str, str]] = None) -> int:
        """POST the spans to an OpenTelemetry collector over OTLP/HTTP with the JSON encoding
        (``<endpoint>/v1/traces``; an endpoint that already ends in ``/v1/traces`` is used as it
        is), ``batch`` spans per request.  Returns the number of spans the collector accepted
        (HTTP 2xx); raises ``OSError`` when it cannot be reached.  (The reference only writes the
        file and leaves "future OTLP integration" as a placeholder: ``tools/bundle_run.sh:235``.)"""
        import urllib.request

        url = endpoint.rstrip("/")
        if not url.endswith("/v1/traces"):
            url += "/v1/traces"
        doc = self.to_otlp()
        spans = doc["resourceSpans"][0]["scopeSpans"][0]["spans"]
        sent = 0
        for i in range(0, len(spans), max(1, batch)):
            part = {"resourceSpans": [{**doc["resourceSpans"][0], "scopeSpans": [{**doc["resourceSpans"][0]["scopeSpans"][0], "spans": spans[i : i + batch]}]}]}
            req = urllib.request.Request(url, data=json.dumps(part).encode(), headers={"Content-Type": "application/json", **(headers or {})}, method="POST")
            with urllib.request.urlopen(req, timeout=timeout_s) as resp:
                if 200 <= resp.status < 300:
                    sent += len(part["resourceSpans"][0]["scopeSpans"][0]["spans"])
        return sent
