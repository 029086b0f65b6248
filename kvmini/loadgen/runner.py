"""The asyncio load engine (``kvmini loadtest`` / ``load-test.sh``).

Open-loop arrival schedule capped by a concurrency semaphore — the reference's
hybrid model (``scripts/loadtest.py:536-560``) — with these differences:

* one dispatcher walks the sorted schedule and spawns a request task when it is
  due (the reference creates every task at t0 and lets each one sleep);
* connections are pooled (``--no-pool`` restores connection-per-request);
* the body is parsed as SSE while it streams: ``ttfb_ms`` is the first body byte
  (reference meaning), ``ttft_ms`` the first *content* token, ``tllt_ms`` the last
  content token; usage comes from ``stream_options.include_usage`` or, failing
  that, from counting content events; non-stream bodies are parsed for ``usage``;
* per-token timestamps go to ``token_events.npy`` for inter-token statistics;
* prompts can rotate from ``--prompts-file`` or be synthesised to ``--prompt-tokens``.

Artifacts: ``requests.csv`` (canonical 12 columns + ``ttft_ms,chunks``),
``meta.json``, ``traces/traces.json``, ``token_events.npy``.
"""

from __future__ import annotations

import argparse
import asyncio
import csv
import json
import os
import random
import sys
import time
from dataclasses import dataclass, field
from typing import Any, Awaitable, Callable, Dict, List, Optional

from kvmini.utils import write_json

from .arrivals import PATTERNS, calculate_duration_and_rps, generate_arrival_times
from .httpclient import HttpClient
from .protocols import PROTOCOLS, Protocol, RequestSpec, get_protocol
from .sse import SSEParser
from .tracing import TraceCollector, create_traceparent_header, generate_span_id, generate_trace_id

CSV_HEADER = [
    "id",
    "scheduled_ms",
    "start_ms",
    "ttfb_ms",
    "tllt_ms",
    "latency_ms",
    "status",
    "prompt_tokens",
    "completion_tokens",
    "total_tokens",
    "error",
    "trace_id",
    # extensions (appended so positional readers of the canonical 12 still work)
    "ttft_ms",
    "chunks",
]


@dataclass
class ReqResult:
    id: int
    scheduled_ms: float
    start_ms: float
    ttfb_ms: Optional[float]
    tllt_ms: Optional[float]
    latency_ms: float
    status: int
    prompt_tokens: Optional[int]
    completion_tokens: Optional[int]
    total_tokens: Optional[int]
    error: Optional[str]
    trace_id: Optional[str] = None
    ttft_ms: Optional[float] = None
    chunks: int = 0
    is_cold_start: Optional[bool] = None
    token_times_ms: List[float] = field(default_factory=list)
    text: str = ""
    tenant: Optional[str] = None
    # structured output (requests that carried a JSON schema): the answer parses / satisfies it
    json_ok: Optional[bool] = None
    schema_ok: Optional[bool] = None
    schema_why: Optional[str] = None

    def csv_row(self) -> List[Any]:
        f3 = lambda v: "" if v is None else f"{v:.3f}"  # noqa: E731
        iv = lambda v: "" if v is None else v  # noqa: E731
        return [
            self.id,
            f"{self.scheduled_ms:.3f}",
            f"{self.start_ms:.3f}",
            f3(self.ttfb_ms),
            f3(self.tllt_ms),
            f"{self.latency_ms:.3f}",
            self.status,
            iv(self.prompt_tokens),
            iv(self.completion_tokens),
            iv(self.total_tokens),
            self.error or "",
            self.trace_id or "",
            f3(self.ttft_ms),
            self.chunks,
        ]


@dataclass
class LoadConfig:
    url: str
    run_dir: Optional[str] = None
    requests: int = 200
    concurrency: int = 10
    pattern: str = "steady"
    target_rps: Optional[float] = None
    protocol: str = "openai"
    spec: RequestSpec = field(default_factory=RequestSpec)
    api_key: Optional[str] = None
    insecure: bool = False
    pooled: bool = True
    connect_retry_s: float = 0.0  # wait this long for a refused port (cold-starting server)
    timeout_s: float = 60.0
    seed: Optional[int] = None
    prompts: Optional[List[str]] = None
    extra_headers: Dict[str, str] = field(default_factory=dict)
    trace: bool = True
    otlp_endpoint: Optional[str] = None  # OpenTelemetry collector the spans are also POSTed to (OTLP/HTTP JSON)
    keep_text: bool = False
    token_events: bool = True
    quiet: bool = False
    # hook called with each finished ReqResult (fairness guard, live histograms)
    on_result: Optional[Callable[[ReqResult], None]] = None
    # gate consulted right before a request is sent; returning False records a 429
    # without touching the network (client-side throttling for the fairness guard)
    admit: Optional[Callable[[int], bool]] = None
    # live inter-token statistics (kvmini.loadgen.live.LiveQuantiles): fed by the SSE parser
    # while the load runs; its final snapshot is written to live_itl.json
    live: Any = None


_WORDS = (
    "the quick brown fox jumps over lazy dog while seven wizards quietly mix "
    "bright liquid for a jovial king and queen of distant northern lands"
).split()


def synthetic_prompt(n_tokens: int, rng: Optional[random.Random] = None) -> str:
    """A prompt of roughly ``n_tokens`` tokens (one short word ~ one token for
    BPE vocabularies; exactly one token per word for the built-in servers'
    whitespace tokenizer)."""
    rng = rng or random.Random(0)
    return " ".join(rng.choice(_WORDS) for _ in range(max(1, n_tokens)))


async def _one_request(
    rid: int,
    scheduled_s: float,
    t0_wall: float,
    t0_perf: float,
    cfg: LoadConfig,
    proto: Protocol,
    client: HttpClient,
    tracer: Optional[TraceCollector],
    sem: asyncio.Semaphore,
    results: List[ReqResult],
) -> None:
    spec = cfg.spec
    url = cfg.url.rstrip("/") + proto.path(spec)
    trace_id = generate_trace_id()
    wall = lambda perf: (t0_wall + (perf - t0_perf)) * 1000.0  # noqa: E731
    root = wait_span = http_span = None
    if tracer:
        root = tracer.start_span(
            trace_id,
            "client.request",
            attributes={"request.id": rid, "http.url": url, "http.method": "POST", "llm.model": spec.model, "llm.max_tokens": spec.max_tokens},
            start_time=wall(time.perf_counter()),
        )
        wait_span = tracer.start_span(trace_id, "client.wait_scheduled", root.span_id, start_time=root.start_time)

    async with sem:
        p_start = time.perf_counter()
        start_ms = wall(p_start)
        if tracer and wait_span:
            tracer.finish_span(wait_span, attributes={"wait_time_ms": (p_start - t0_perf - scheduled_s) * 1000.0}, end_time=start_ms)

        status = 0
        err: Optional[str] = None
        usage: Optional[Dict[str, Any]] = None
        ttfb = ttft = tllt = None
        token_times: List[float] = []
        chunks = 0
        text = ""
        answer: Optional[str] = None  # kept only when the answer is checked against a schema
        check_schema = spec.response_schema is not None

        if cfg.admit is not None and not cfg.admit(rid):
            # throttled client-side: recorded as HTTP 429, like a server-side shed
            status, err = 429, "throttled"
        else:
            headers = {"Content-Type": "application/json"}
            if cfg.api_key:
                headers["Authorization"] = f"Bearer {cfg.api_key}"
            headers["traceparent"] = create_traceparent_header(trace_id, generate_span_id())
            if spec.stream:
                headers["Accept"] = "text/event-stream"
            headers.update(cfg.extra_headers)
            prompt = cfg.prompts[(rid - 1) % len(cfg.prompts)] if cfg.prompts else None
            payload = proto.payload(spec, prompt)
            if tracer and root:
                http_span = tracer.start_span(trace_id, "http.request", root.span_id, {"http.url": url}, start_time=start_ms)
            try:
                resp = await client.post(url, headers=headers, json_body=payload, timeout=cfg.timeout_s)
                status = resp.status
                ctype = resp.headers.get("content-type", "")
                if spec.stream and status == 200 and ("event-stream" in ctype or "json" not in ctype):
                    parser = SSEParser(proto.sse_dialect, keep_text=cfg.keep_text or check_schema)
                    if cfg.live is not None:
                        base_ns, live = int(p_start * 1e9), cfg.live
                        parser.on_token = lambda idx, t_ms, _rid=rid: live.on_token(_rid, idx, base_ns + int(t_ms * 1e6))

                    async def _consume() -> None:
                        async for data in resp.iter_bytes():
                            parser.feed(data, (time.perf_counter() - p_start) * 1000.0)
                        parser.close((time.perf_counter() - p_start) * 1000.0)

                    await asyncio.wait_for(_consume(), cfg.timeout_s)
                    chunks = parser.n_chunks
                    ttfb = parser.first_byte_ms
                    ttft = parser.first_token_ms
                    # TLLT = last content token; fall back to last byte when the
                    # stream carried no recognisable content events
                    tllt = parser.last_token_ms if parser.last_token_ms is not None else parser.last_byte_ms
                    token_times = parser.token_times_ms
                    usage = parser.usage
                    text = parser.text if cfg.keep_text else ""
                    answer = parser.text if check_schema else None
                    if usage is None and parser.content_events:
                        usage = {"completion_tokens": parser.content_events}
                else:
                    t_first = None
                    body_parts = []
                    async for data in resp.iter_bytes():
                        if t_first is None:
                            t_first = (time.perf_counter() - p_start) * 1000.0
                        body_parts.append(data)
                    raw = b"".join(body_parts)
                    chunks = len(body_parts)
                    if spec.stream:
                        ttfb = t_first
                    try:
                        body = json.loads(raw.decode("utf-8", "replace")) if raw else None
                    except ValueError:
                        body = None
                    if status == 200:
                        usage, text_full = proto.parse_body(body)
                        text = text_full if cfg.keep_text else ""
                        answer = text_full if check_schema else None
                    elif isinstance(body, dict):
                        e = body.get("error")
                        err = (e.get("message") if isinstance(e, dict) else str(e)) if e else None
                if tracer and http_span:
                    tracer.finish_span(http_span, "ok", {"http.status_code": status, "response.chunk_count": chunks})
            except asyncio.TimeoutError:
                err = f"timeout after {cfg.timeout_s}s"
                if tracer and http_span:
                    tracer.finish_span(http_span, "timeout", {"error.message": err})
            except Exception as e:  # network / protocol failure -> status 0 + error text
                err = f"{type(e).__name__}: {e}"
                if tracer and http_span:
                    tracer.finish_span(http_span, "error", {"error.message": err})

        p_end = time.perf_counter()
        latency = (p_end - p_start) * 1000.0
        pt = ct = tt = None
        if isinstance(usage, dict):
            pt, ct, tt = usage.get("prompt_tokens"), usage.get("completion_tokens"), usage.get("total_tokens")
            if tt is None and ct is not None:
                tt = (pt or 0) + ct

        if tracer and root and http_span:
            if ttfb is not None:
                s = tracer.start_span(trace_id, "server.ttft", http_span.span_id, {"ttft_ms": ttfb, "first_content_token_ms": ttft if ttft is not None else ttfb}, start_time=start_ms)
                tracer.finish_span(s, end_time=start_ms + ttfb)
            if tllt is not None:
                s = tracer.start_span(trace_id, "server.tllt", http_span.span_id, {"tllt_ms": tllt}, start_time=start_ms)
                tracer.finish_span(s, end_time=start_ms + tllt)
        if tracer and root:
            tracer.finish_span(
                root,
                "error" if (err and status != 429) else "ok",
                {
                    "http.status_code": status,
                    "request.latency_ms": latency,
                    "llm.prompt_tokens": pt or 0,
                    "llm.completion_tokens": ct or 0,
                    "llm.total_tokens": tt or 0,
                },
                end_time=wall(p_end),
            )

        res = ReqResult(
            id=rid,
            scheduled_ms=(t0_wall + scheduled_s) * 1000.0,
            start_ms=start_ms,
            ttfb_ms=ttfb,
            tllt_ms=tllt,
            latency_ms=latency,
            status=status,
            prompt_tokens=pt,
            completion_tokens=ct,
            total_tokens=tt,
            error=err,
            trace_id=trace_id,
            ttft_ms=ttft,
            chunks=chunks,
            token_times_ms=token_times,
            text=text,
            tenant=cfg.extra_headers.get("X-Tenant"),
        )
        if check_schema and status == 200 and answer is not None:
            from .schema_check import check_document

            res.json_ok, res.schema_ok, res.schema_why = check_document(answer, spec.response_schema)
        results.append(res)
        if cfg.live is not None:
            cfg.live.finished(rid)
        if cfg.on_result:
            try:
                cfg.on_result(res)
            except Exception:
                pass


async def run_load(cfg: LoadConfig, tracer: Optional[TraceCollector] = None, client: Optional[HttpClient] = None) -> List[ReqResult]:
    """Run the schedule and return per-request results (also persisted when
    ``cfg.run_dir`` is set)."""
    if cfg.pattern not in PATTERNS:
        raise ValueError(f"Unknown pattern: {cfg.pattern}")
    proto = get_protocol(cfg.protocol)
    rng = random.Random(cfg.seed)
    duration, target_rps = calculate_duration_and_rps(cfg.requests, cfg.concurrency, cfg.pattern, cfg.target_rps)
    arrivals = generate_arrival_times(cfg.pattern, cfg.requests, duration, target_rps, rng)
    if not cfg.quiet:
        print(f"Generated {cfg.pattern} pattern: {cfg.requests} requests over {duration:.1f}s (target {target_rps:.1f} RPS)")

    own_tracer = tracer is None and cfg.trace
    if own_tracer:
        tracer = TraceCollector()
    own_client = client is None
    if own_client:
        client = HttpClient(pooled=cfg.pooled, verify_tls=not cfg.insecure, connect_retry_s=cfg.connect_retry_s)
    sem = asyncio.Semaphore(cfg.concurrency)
    results: List[ReqResult] = []
    tasks: List[asyncio.Task] = []

    t0_wall = time.time()
    t0_perf = time.perf_counter()
    if cfg.live is not None:
        cfg.live.start()
    try:
        for i, at in enumerate(arrivals):
            delay = at - (time.perf_counter() - t0_perf)
            if delay > 0:
                await asyncio.sleep(delay)
            tasks.append(asyncio.create_task(_one_request(i + 1, at, t0_wall, t0_perf, cfg, proto, client, tracer if cfg.trace else None, sem, results)))
        if tasks:
            await asyncio.gather(*tasks)
    finally:
        if own_client:
            await client.aclose()
    live_final = await cfg.live.stop() if cfg.live is not None else None

    results.sort(key=lambda r: r.id)
    if cfg.run_dir and live_final is not None:
        os.makedirs(cfg.run_dir, exist_ok=True)
        write_json(os.path.join(cfg.run_dir, "live_itl.json"), {**live_final, "interval_s": cfg.live.interval_s, "history": [{"t": t, "p50": q[0.5], "p95": q[0.95], "p99": q[0.99]} for t, q in cfg.live.history]})
    if live_final is not None and not cfg.quiet:
        print(f"Live inter-token latency ({live_final['engine']}, {live_final['updates']} histogram updates during the run): p50 {live_final['itl_p50_ms']} ms, p95 {live_final['itl_p95_ms']} ms, p99 {live_final['itl_p99_ms']} ms over {live_final['itl_samples']} gaps")
    if cfg.run_dir:
        persist_run(cfg, results, tracer if cfg.trace else None, duration, target_rps, t0_wall, client.connections_opened if client else 0)
    return results


def structured_summary(results: List[ReqResult]) -> Optional[Dict[str, Any]]:
    """Compliance of the answers that were checked against a schema (None: no request carried one)."""
    checked = [r for r in results if r.json_ok is not None]
    if not checked:
        return None
    ok = sum(1 for r in checked if r.schema_ok)
    return {"checked": len(checked), "valid_json": sum(1 for r in checked if r.json_ok), "schema_valid": ok, "schema_valid_rate": ok / len(checked),
            "first_failures": [{"id": r.id, "why": r.schema_why} for r in checked if not r.schema_ok][:5]}


def persist_run(cfg: LoadConfig, results: List[ReqResult], tracer: Optional[TraceCollector], duration: float, target_rps: float, t0_wall: float, conns: int = 0) -> None:
    run_dir = cfg.run_dir
    assert run_dir
    os.makedirs(run_dir, exist_ok=True)
    if tracer is not None:
        trace_file = os.path.join(run_dir, "traces", "traces.json")
        tracer.export_traces(trace_file)
        if not cfg.quiet:
            print(f"Exported {len(tracer.spans)} trace spans to {trace_file}")
        if cfg.otlp_endpoint:
            try:  # a collector that is down must not cost the run its artifacts
                n = tracer.export_http(cfg.otlp_endpoint)
                if not cfg.quiet:
                    print(f"Sent {n} trace spans to {cfg.otlp_endpoint}")
            except (OSError, ValueError) as e:
                print(f"OTLP export to {cfg.otlp_endpoint} failed: {e}", file=sys.stderr)
    with open(os.path.join(run_dir, "requests.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_HEADER)
        for r in results:
            w.writerow(r.csv_row())
    if cfg.token_events:
        import numpy as np

        n = sum(len(r.token_times_ms) for r in results)
        ev = np.empty((n, 3), dtype=np.float64)
        k = 0
        for r in results:
            m = len(r.token_times_ms)
            if m:
                ev[k : k + m, 0] = r.id
                ev[k : k + m, 1] = np.arange(m)
                ev[k : k + m, 2] = r.token_times_ms
                k += m
        np.save(os.path.join(run_dir, "token_events.npy"), ev)
    spec = cfg.spec
    structured = structured_summary(results)
    if structured is not None:
        write_json(os.path.join(run_dir, "structured.json"), structured)
        if not cfg.quiet:
            print(f"Structured output: {structured['schema_valid']} / {structured['checked']} answers satisfy the schema ({structured['valid_json']} parse as JSON)")
    write_json(
        os.path.join(run_dir, "meta.json"),
        {
            "url": cfg.url,
            "model": spec.model,
            "prompt": spec.prompt if not cfg.prompts else f"<{len(cfg.prompts)} rotating prompts>",
            "max_tokens": spec.max_tokens,
            "concurrency": cfg.concurrency,
            "requests": cfg.requests,
            "pattern": cfg.pattern,
            "duration_sec": duration,
            "target_rps": target_rps,
            "test_start_time": t0_wall,
            "stream": spec.stream,
            "protocol": cfg.protocol,
            "pooled_connections": cfg.pooled,
            "connections_opened": conns,
            "seed": cfg.seed,
        },
    )


def run_load_sync(cfg: LoadConfig) -> List[ReqResult]:
    return asyncio.run(run_load(cfg))


def build_arg_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="kvmini loadtest", description="OpenAI/TGI/KServe-v2 compatible load generator")
    ap.add_argument("--url", required=True, help="Base URL of the endpoint, e.g. http://host:8000")
    ap.add_argument("--model", default="placeholder")
    ap.add_argument("--prompt", default="Hello, world!")
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--concurrency", type=int, default=10)
    ap.add_argument("--pattern", choices=list(PATTERNS), default="steady")
    ap.add_argument("--target-rps", type=float, default=None, help="Override the pattern's rate heuristic")
    ap.add_argument("--run-dir", required=True)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--insecure", action="store_true", help="Disable TLS verification")
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--stream", dest="stream", action="store_true", default=None, help="Stream responses (SSE); needed for TTFT")
    g.add_argument("--no-stream", dest="stream", action="store_false")
    ap.add_argument("--protocol", choices=sorted(PROTOCOLS), default="openai")
    ap.add_argument("--temperature", type=float, default=None)
    ap.add_argument("--top-p", type=float, default=None)
    ap.add_argument("--top-k", type=int, default=None)
    ap.add_argument("--num-completions", type=int, default=None)
    ap.add_argument("--presence-penalty", type=float, default=None)
    ap.add_argument("--frequency-penalty", type=float, default=None)
    ap.add_argument("--json-mode", action="store_true")
    ap.add_argument("--extra-openai-json", default=None, help="JSON file with extra payload fields")
    ap.add_argument("--tools-json", default=None, help="JSON file with an OpenAI tools array")
    ap.add_argument("--schema-json", default=None, help="JSON-schema file for structured output")
    ap.add_argument("--prompts-file", default=None, help="One prompt per line (or a JSON list); rotated per request")
    ap.add_argument("--prompt-tokens", type=int, default=None, help="Synthesise a prompt of about N tokens")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--procs", type=int, default=1, help="Generator processes (requests, concurrency and rate are split evenly; artifacts are merged)")
    ap.add_argument("--no-pool", action="store_true", help="New TCP/TLS connection per request (reference behaviour)")
    ap.add_argument("--connect-retry-s", type=float, default=0.0, help="Keep retrying a refused connection for this long (requests wait out a server cold start)")
    ap.add_argument("--no-trace", action="store_true")
    ap.add_argument("--otlp-endpoint", default=os.environ.get("OTEL_EXPORTER_OTLP_ENDPOINT"),
                    help="also POST the trace spans to this OpenTelemetry collector (OTLP/HTTP JSON, <endpoint>/v1/traces; default: $OTEL_EXPORTER_OTLP_ENDPOINT)")
    ap.add_argument("--timeout", type=float, default=60.0)
    ap.add_argument("--header", action="append", default=[], help="Extra header 'Name: value' (repeatable)")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--live-quantiles", action="store_true", help="rolling inter-token p50/p95/p99 while the load runs (streaming histogram: HIP kernel on a GPU host, numpy otherwise)")
    ap.add_argument("--live-interval", type=float, default=0.25, help="seconds between histogram updates of --live-quantiles")
    ap.add_argument("--live-metrics-port", type=int, default=None, help="serve the live quantiles as Prometheus text on this port (0 = any free port)")
    ap.add_argument("--live-device", default=None, choices=["cuda", "cpu"], help="where the live histogram lives (default: the GPU when one is usable)")
    return ap


def config_from_args(args: argparse.Namespace) -> LoadConfig:
    extra: Dict[str, Any] = {}
    if args.extra_openai_json:
        with open(args.extra_openai_json) as f:  # read once, not per request
            extra = json.load(f)
    tools = None
    if args.tools_json:
        with open(args.tools_json) as f:
            tools = json.load(f)
    schema = None
    if args.schema_json:
        with open(args.schema_json) as f:
            schema = json.load(f)
    prompts = None
    if args.prompts_file:
        with open(args.prompts_file) as f:
            txt = f.read()
        try:
            prompts = [str(p) for p in json.loads(txt)]
        except ValueError:
            prompts = [ln for ln in txt.splitlines() if ln.strip()]
    prompt = args.prompt
    if args.prompt_tokens:
        prompt = synthetic_prompt(args.prompt_tokens, random.Random(args.seed or 0))
    headers = {}
    for h in args.header:
        k, _, v = h.partition(":")
        headers[k.strip()] = v.strip()
    spec = RequestSpec(
        model=args.model,
        prompt=prompt,
        max_tokens=args.max_tokens,
        stream=True if args.stream is None else args.stream,
        temperature=args.temperature,
        top_p=args.top_p,
        top_k=args.top_k,
        n=args.num_completions,
        presence_penalty=args.presence_penalty,
        frequency_penalty=args.frequency_penalty,
        json_mode=args.json_mode,
        extra_payload=extra,
        tools=tools,
        response_schema=schema,
    )
    live = None
    if getattr(args, "live_quantiles", False) or getattr(args, "live_metrics_port", None) is not None:
        from .live import LiveQuantiles

        live = LiveQuantiles(interval_s=args.live_interval, device=args.live_device, verbose=not args.quiet)
        if args.live_metrics_port is not None:
            port = live.serve_metrics(args.live_metrics_port)
            if not args.quiet:
                print(f"live quantiles: http://127.0.0.1:{port}/metrics")
    return LoadConfig(
        live=live,
        url=args.url,
        run_dir=args.run_dir,
        requests=args.requests,
        concurrency=args.concurrency,
        pattern=args.pattern,
        target_rps=args.target_rps,
        protocol=args.protocol,
        spec=spec,
        api_key=args.api_key,
        insecure=args.insecure,
        pooled=not args.no_pool,
        connect_retry_s=args.connect_retry_s,
        timeout_s=args.timeout,
        seed=args.seed,
        prompts=prompts,
        extra_headers=headers,
        trace=not args.no_trace,
        otlp_endpoint=getattr(args, "otlp_endpoint", None),
        quiet=args.quiet,
    )


def run_sharded(argv: List[str], args: argparse.Namespace) -> int:
    """``--procs N``: N generator processes, each with 1/N of the requests, of the
    concurrency and of the target rate, started together against the same endpoint; their
    run directories (``<run-dir>/shard<k>``) are merged into the canonical artifacts
    (request ids renumbered, token events and trace spans concatenated).  One asyncio
    process tops out around 10^5 token events per second; shards scale that with cores."""
    import json
    import subprocess

    import numpy as np

    n = args.procs
    base = [a for a in (argv if argv is not None else sys.argv[1:])]

    def strip(flag: str, has_value: bool = True) -> None:
        while flag in base:
            i = base.index(flag)
            del base[i : i + (2 if has_value else 1)]

    for flag in ("--procs", "--requests", "--concurrency", "--run-dir", "--target-rps", "--seed"):
        strip(flag)
    procs = []
    for k in range(n):
        req = args.requests // n + (1 if k < args.requests % n else 0)
        conc = max(1, args.concurrency // n + (1 if k < args.concurrency % n else 0))
        shard = ["--requests", str(req), "--concurrency", str(conc), "--run-dir", os.path.join(args.run_dir, f"shard{k}"), "--seed", str((args.seed or 0) * 1000 + k)]
        if args.target_rps:
            shard += ["--target-rps", str(args.target_rps / n)]
        procs.append(subprocess.Popen([sys.executable, "-m", "kvmini.loadgen.runner", *base, *shard, "--quiet"]))
    rc = max(p.wait() for p in procs)
    rows: List[List[str]] = []
    events, spans, metas = [], [], []
    for k in range(n):
        d = os.path.join(args.run_dir, f"shard{k}")
        off = len(rows)
        with open(os.path.join(d, "requests.csv"), newline="") as f:
            rd = csv.reader(f)
            header = next(rd)
            for r in rd:
                r[0] = str(int(r[0]) + off)
                rows.append(r)
        ev_path = os.path.join(d, "token_events.npy")
        if os.path.exists(ev_path):
            ev = np.load(ev_path)
            ev[:, 0] += off
            events.append(ev)
        tr = os.path.join(d, "traces", "traces.json")
        if os.path.exists(tr):
            with open(tr) as f:
                doc = json.load(f)
            spans += doc["resourceSpans"][0]["scopeSpans"][0]["spans"]
        with open(os.path.join(d, "meta.json")) as f:
            metas.append(json.load(f))
    os.makedirs(args.run_dir, exist_ok=True)
    with open(os.path.join(args.run_dir, "requests.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)
    if events:
        np.save(os.path.join(args.run_dir, "token_events.npy"), np.concatenate(events))
    if spans:
        doc["resourceSpans"][0]["scopeSpans"][0]["spans"] = spans
        os.makedirs(os.path.join(args.run_dir, "traces"), exist_ok=True)
        with open(os.path.join(args.run_dir, "traces", "traces.json"), "w") as f:
            json.dump(doc, f)
    meta = dict(metas[0])
    meta.update({"requests": args.requests, "concurrency": args.concurrency, "target_rps": sum(m.get("target_rps") or 0 for m in metas), "duration_sec": max(m.get("duration_sec") or 0 for m in metas),
                 "test_start_time": min(m.get("test_start_time") or 0 for m in metas), "connections_opened": sum(m.get("connections_opened") or 0 for m in metas), "generator_processes": n})
    write_json(os.path.join(args.run_dir, "meta.json"), meta)
    if not args.quiet:
        ok = sum(1 for r in rows if r[header.index("status")] == "200")
        print(f"Completed {len(rows)} requests ({ok} ok) in {n} generator processes -> {args.run_dir}")
    return rc


def main(argv: Optional[List[str]] = None) -> int:
    args = build_arg_parser().parse_args(argv)
    if getattr(args, "procs", 1) > 1:
        return run_sharded(list(argv) if argv is not None else sys.argv[1:], args)
    cfg = config_from_args(args)
    try:
        results = run_load_sync(cfg)
    except KeyboardInterrupt:
        print("Aborted")
        return 130
    ok = sum(1 for r in results if r.status == 200)
    if not cfg.quiet:
        print(f"Completed {len(results)} requests ({ok} ok) -> {cfg.run_dir}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
