"""Deterministic CPU mock of an LLM endpoint (``kvmini mock``).

Speaks the three wire protocols the harness can drive — OpenAI
(``/v1/chat/completions``, ``/v1/completions``, ``/v1/models``), TGI
(``/generate``, ``/generate_stream``, ``/info``) and KServe-v2 generate
(``/v2/models/<m>/generate[_stream]``) — plus ``/health`` and Prometheus-text
``/metrics``.  Latency model: ``ttft_ms`` before the first token (multiplied by
``cold_multiplier`` during the first ``cold_seconds`` after start, divided for
prompts whose prefix was seen before when ``prefix_cache`` is on), then
``tpot_ms`` per token, both with multiplicative jitter from a seeded RNG;
``error_rate`` of requests fail with HTTP 500.  It exists so that the whole
pipeline — load generator, SSE/TTFT parser, analyzer, cost, gates, sweeps, A/B,
fairness, parity, cache probe, chaos — runs in CPU-only CI; the reference has no
such server and therefore no test of its load generator at all.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import random
import sys
import time
import uuid
from dataclasses import dataclass
from typing import Any, AsyncIterator, Dict, List, Optional

from .httpd import HttpServer, Request, Response, StreamingResponse, sse_event

_VOCAB = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau upsilon phi chi psi omega".split()


@dataclass
class MockConfig:
    model: str = "mock-llm"
    ttft_ms: float = 8.0
    tpot_ms: float = 1.0
    jitter: float = 0.1
    error_rate: float = 0.0
    cold_seconds: float = 0.0
    cold_multiplier: float = 5.0
    prefix_cache: bool = True
    prefix_cache_speedup: float = 4.0
    seed: int = 0
    api_key: Optional[str] = None
    max_concurrency: int = 0  # >0: extra queueing delay once this many requests are in flight
    supports_tools: bool = True
    supports_json_mode: bool = True
    supports_logprobs: bool = True
    tenant_shed: Optional[str] = None  # tenant name whose requests get 429 when shedding is on


def tokenize(text: str) -> List[str]:
    return text.split()


class MockLLM:
    def __init__(self, cfg: Optional[MockConfig] = None):
        self.cfg = cfg or MockConfig()
        self.rng = random.Random(self.cfg.seed)
        self.started_at = time.time()
        self.inflight = 0
        self.seen_prompts: set = set()
        self.stats = {"requests": 0, "errors": 0, "prompt_tokens": 0, "generation_tokens": 0, "cache_hits": 0, "cache_queries": 0}
        self.extra_latency_ms = 0.0  # chaos: injected latency
        self.down_until = 0.0  # chaos: "restarting" until this wall-clock time
        self.drop_rate = 0.0  # chaos: injected connection failure rate

    # --- latency model ---------------------------------------------------
    def _jit(self, ms: float) -> float:
        j = self.cfg.jitter
        return max(0.0, ms * (1.0 + self.rng.uniform(-j, j))) / 1000.0

    def _ttft_s(self, prompt: str) -> float:
        ms = self.cfg.ttft_ms
        if self.cfg.cold_seconds and time.time() - self.started_at < self.cfg.cold_seconds:
            ms *= self.cfg.cold_multiplier
        if self.cfg.prefix_cache:
            self.stats["cache_queries"] += 1
            key = prompt[:256]
            if key in self.seen_prompts:
                self.stats["cache_hits"] += 1
                ms /= self.cfg.prefix_cache_speedup
            else:
                self.seen_prompts.add(key)
        if self.cfg.max_concurrency and self.inflight > self.cfg.max_concurrency:
            ms *= self.inflight / self.cfg.max_concurrency
        # injected (network-like) delay is not something a prefix cache can hide
        return self._jit(ms) + self.extra_latency_ms / 1000.0

    def _gen_words(self, prompt: str, n: int, json_mode: bool = False) -> List[str]:
        if json_mode:
            words = ['{"answer":', '"' + _VOCAB[len(prompt) % len(_VOCAB)] + '",', '"ok":', "true}"]
            return words[: max(n, 4)] if n >= 4 else ['{"a":', "1}"][:n] or ["{}"]
        h = sum(prompt.encode()) % len(_VOCAB)
        return [_VOCAB[(h + i) % len(_VOCAB)] for i in range(n)]

    # --- helpers -----------------------------------------------------------
    def _auth_ok(self, req: Request) -> bool:
        if not self.cfg.api_key:
            return True
        return req.headers.get("authorization", "") == f"Bearer {self.cfg.api_key}"

    def _should_fail(self) -> bool:
        return self.cfg.error_rate > 0 and self.rng.random() < self.cfg.error_rate

    # --- OpenAI ------------------------------------------------------------
    async def chat(self, req: Request) -> Any:
        return await self._openai(req, chat=True)

    async def completions(self, req: Request) -> Any:
        return await self._openai(req, chat=False)

    async def _openai(self, req: Request, chat: bool) -> Any:
        if not self._auth_ok(req):
            return Response.json({"error": {"message": "invalid api key", "type": "authentication_error"}}, 401)
        body = req.json() or {}
        self.stats["requests"] += 1
        if time.time() < self.down_until:
            self.stats["errors"] += 1
            return Response.json({"error": {"message": "service restarting", "type": "unavailable"}}, 503)
        if self.cfg.tenant_shed and req.headers.get("x-tenant") == self.cfg.tenant_shed:
            return Response.json({"error": {"message": "tenant throttled", "type": "rate_limit"}}, 429)
        if self._should_fail():
            self.stats["errors"] += 1
            return Response.json({"error": {"message": "injected failure", "type": "internal_error"}}, 500)
        if chat:
            msgs = body.get("messages") or []
            prompt = " ".join(str(m.get("content", "")) for m in msgs if isinstance(m, dict))
        else:
            prompt = str(body.get("prompt", ""))
        max_tokens = int(body.get("max_tokens") or body.get("max_completion_tokens") or 16)
        n_choices = int(body.get("n") or 1)
        stream = bool(body.get("stream"))
        include_usage = bool((body.get("stream_options") or {}).get("include_usage"))
        json_mode = self.cfg.supports_json_mode and (body.get("response_format") or {}).get("type") in ("json_object", "json_schema")
        tools = body.get("tools") if self.cfg.supports_tools else None
        want_logprobs = self.cfg.supports_logprobs and bool(body.get("logprobs"))
        model = body.get("model") or self.cfg.model
        rid = ("chatcmpl-" if chat else "cmpl-") + uuid.uuid4().hex[:24]
        created = int(time.time())
        ptoks = len(tokenize(prompt))
        self.stats["prompt_tokens"] += ptoks

        tool_calls = None
        if tools and chat:
            # call every offered function once when the prompt asks for several, else the first
            many = "and" in prompt.lower() or body.get("parallel_tool_calls")
            chosen = tools if many else tools[:1]
            tool_calls = []
            for i, t in enumerate(chosen):
                fn = (t.get("function") or {}).get("name", f"fn{i}")
                props = (((t.get("function") or {}).get("parameters") or {}).get("properties") or {})
                args = {k: ("Paris" if (v or {}).get("type") == "string" else 1) for k, v in props.items()}
                tool_calls.append({"index": i, "id": f"call_{uuid.uuid4().hex[:12]}", "type": "function", "function": {"name": fn, "arguments": json.dumps(args)}})

        words = self._gen_words(prompt, max_tokens, json_mode)
        ctoks = len(words) * n_choices if not tool_calls else len(tool_calls)
        usage = {"prompt_tokens": ptoks, "completion_tokens": ctoks, "total_tokens": ptoks + ctoks}
        obj = "chat.completion" if chat else "text_completion"

        def lp(w: str) -> Dict[str, Any]:
            return {"content": [{"token": w, "logprob": -0.1, "bytes": list(w.encode()), "top_logprobs": [{"token": w, "logprob": -0.1}]}]}

        if not stream:
            self.inflight += 1
            try:
                await asyncio.sleep(self._ttft_s(prompt) + sum(self._jit(self.cfg.tpot_ms) for _ in range(max(0, len(words) - 1))))
            finally:
                self.inflight -= 1
            self.stats["generation_tokens"] += ctoks
            choices = []
            for i in range(n_choices):
                text = " ".join(words)
                if chat:
                    msg: Dict[str, Any] = {"role": "assistant", "content": None if tool_calls else text}
                    if tool_calls:
                        msg["tool_calls"] = [{k: v for k, v in tc.items() if k != "index"} for tc in tool_calls]
                    ch: Dict[str, Any] = {"index": i, "message": msg, "finish_reason": "tool_calls" if tool_calls else "length"}
                    if want_logprobs:
                        ch["logprobs"] = {"content": [lp(w)["content"][0] for w in words]}
                else:
                    ch = {"index": i, "text": text, "finish_reason": "length"}
                choices.append(ch)
            return Response.json({"id": rid, "object": obj, "created": created, "model": model, "choices": choices, "usage": usage})

        async def gen() -> AsyncIterator[bytes]:
            self.inflight += 1
            try:
                base = {"id": rid, "object": obj + ".chunk" if chat else obj, "created": created, "model": model}
                await asyncio.sleep(self._ttft_s(prompt))
                if chat:
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {"role": "assistant", "content": ""}, "finish_reason": None}]})
                if tool_calls:
                    for tc in tool_calls:
                        yield sse_event({**base, "choices": [{"index": 0, "delta": {"tool_calls": [tc]}, "finish_reason": None}]})
                        await asyncio.sleep(self._jit(self.cfg.tpot_ms))
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {}, "finish_reason": "tool_calls"}]})
                else:
                    for k, w in enumerate(words):
                        if k:
                            await asyncio.sleep(self._jit(self.cfg.tpot_ms))
                        piece = w if k == 0 else " " + w
                        for i in range(n_choices):
                            if chat:
                                ch = {"index": i, "delta": {"content": piece}, "finish_reason": None}
                                if want_logprobs:
                                    ch["logprobs"] = lp(w)
                            else:
                                ch = {"index": i, "text": piece, "finish_reason": None}
                            yield sse_event({**base, "choices": [ch]})
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {}, "finish_reason": "length"} if chat else {"index": 0, "text": "", "finish_reason": "length"}]})
                if include_usage:
                    yield sse_event({**base, "choices": [], "usage": usage})
                yield sse_event("[DONE]")
                self.stats["generation_tokens"] += ctoks
            finally:
                self.inflight -= 1

        return StreamingResponse(gen())

    async def models(self, req: Request) -> Response:
        return Response.json({"object": "list", "data": [{"id": self.cfg.model, "object": "model", "created": int(self.started_at), "owned_by": "kvmini"}]})

    async def health(self, req: Request) -> Response:
        if time.time() < self.down_until:  # "restarting" (POST /admin/chaos {"restart": true, "down_s": ...})
            return Response.json({"status": "starting"}, 503)
        return Response.json({"status": "ok", "uptime_s": time.time() - self.started_at})

    async def metrics(self, req: Request) -> Response:
        s = self.stats
        lines = [
            "# TYPE vllm:num_requests_running gauge",
            f'vllm:num_requests_running{{model_name="{self.cfg.model}"}} {self.inflight}',
            "# TYPE vllm:prompt_tokens_total counter",
            f'vllm:prompt_tokens_total{{model_name="{self.cfg.model}"}} {s["prompt_tokens"]}',
            "# TYPE vllm:generation_tokens_total counter",
            f'vllm:generation_tokens_total{{model_name="{self.cfg.model}"}} {s["generation_tokens"]}',
            "# TYPE vllm:gpu_prefix_cache_hits_total counter",
            f'vllm:gpu_prefix_cache_hits_total{{model_name="{self.cfg.model}"}} {s["cache_hits"]}',
            "# TYPE vllm:gpu_prefix_cache_queries_total counter",
            f'vllm:gpu_prefix_cache_queries_total{{model_name="{self.cfg.model}"}} {s["cache_queries"]}',
            "# TYPE kvmini_requests_total counter",
            f"kvmini_requests_total {s['requests']}",
            f"kvmini_request_errors_total {s['errors']}",
        ]
        return Response.text("\n".join(lines) + "\n", content_type="text/plain; version=0.0.4")

    # --- TGI ---------------------------------------------------------------
    async def tgi_generate(self, req: Request) -> Any:
        body = req.json() or {}
        self.stats["requests"] += 1
        if self._should_fail():
            self.stats["errors"] += 1
            return Response.json({"error": "injected failure", "error_type": "generation"}, 500)
        prompt = str(body.get("inputs", ""))
        params = body.get("parameters") or {}
        n = int(params.get("max_new_tokens") or 20)
        words = self._gen_words(prompt, n)
        stream = req.path.endswith("_stream") or bool(body.get("stream")) and req.path.endswith("_stream")
        ptoks = len(tokenize(prompt))
        self.stats["prompt_tokens"] += ptoks
        details = {"finish_reason": "length", "generated_tokens": len(words), "seed": None, "prefill": [{"id": i, "text": t} for i, t in enumerate(tokenize(prompt))], "tokens": []}
        if not stream:
            await asyncio.sleep(self._ttft_s(prompt) + sum(self._jit(self.cfg.tpot_ms) for _ in range(max(0, n - 1))))
            self.stats["generation_tokens"] += len(words)
            return Response.json({"generated_text": " ".join(words), "details": details})

        async def gen() -> AsyncIterator[bytes]:
            self.inflight += 1
            try:
                await asyncio.sleep(self._ttft_s(prompt))
                for k, w in enumerate(words):
                    if k:
                        await asyncio.sleep(self._jit(self.cfg.tpot_ms))
                    ev: Dict[str, Any] = {"index": k, "token": {"id": k, "text": (" " if k else "") + w, "logprob": -0.1, "special": False}, "generated_text": None, "details": None}
                    if k == len(words) - 1:
                        ev["generated_text"] = " ".join(words)
                        ev["details"] = {"finish_reason": "length", "generated_tokens": len(words), "seed": None}
                    yield sse_event(ev)
                self.stats["generation_tokens"] += len(words)
            finally:
                self.inflight -= 1

        return StreamingResponse(gen())

    async def tgi_info(self, req: Request) -> Response:
        return Response.json({"model_id": self.cfg.model, "max_total_tokens": 8192, "version": "mock", "model_device_type": "rocm"})

    # --- KServe v2 generate --------------------------------------------------
    async def v2_generate(self, req: Request) -> Any:
        body = req.json() or {}
        self.stats["requests"] += 1
        if self._should_fail():
            self.stats["errors"] += 1
            return Response.json({"error": "injected failure"}, 500)
        prompt, n = "", 16
        for t in body.get("inputs") or []:
            if t.get("name") == "text_input":
                prompt = str((t.get("data") or [""])[0])
            if t.get("name") == "max_tokens":
                n = int((t.get("data") or [16])[0])
        if "text_input" in body:
            prompt = str(body["text_input"])
            n = int(body.get("max_tokens", n))
        words = self._gen_words(prompt, n)
        model_name = req.path.split("/")[3] if req.path.count("/") >= 4 else self.cfg.model
        if not req.path.endswith("_stream"):
            await asyncio.sleep(self._ttft_s(prompt) + sum(self._jit(self.cfg.tpot_ms) for _ in range(max(0, n - 1))))
            return Response.json({"model_name": model_name, "outputs": [{"name": "text_output", "datatype": "BYTES", "shape": [1], "data": [" ".join(words)]}, {"name": "sequence_length", "datatype": "INT32", "shape": [1], "data": [len(words)]}]})

        async def gen() -> AsyncIterator[bytes]:
            await asyncio.sleep(self._ttft_s(prompt))
            for k, w in enumerate(words):
                if k:
                    await asyncio.sleep(self._jit(self.cfg.tpot_ms))
                yield sse_event({"model_name": model_name, "text_output": (" " if k else "") + w, "token_id": k})

        return StreamingResponse(gen())

    async def v2_ready(self, req: Request) -> Response:
        return Response.json({"ready": True})

    # --- chaos / admin -------------------------------------------------------
    async def admin(self, req: Request) -> Response:
        """POST /admin/chaos {"extra_latency_ms":..,"error_rate":..,"restart":true,"down_s":..,"tenant_shed":..}"""
        body = req.json() or {}
        if "extra_latency_ms" in body:
            self.extra_latency_ms = float(body["extra_latency_ms"])
        if "error_rate" in body:
            self.cfg.error_rate = float(body["error_rate"])
        if "tenant_shed" in body:
            self.cfg.tenant_shed = body["tenant_shed"]
        if body.get("restart"):
            self.started_at = time.time()
            self.seen_prompts.clear()
            self.down_until = time.time() + float(body.get("down_s") or 0.0)  # /health and the API answer 503 until then
        return Response.json({"ok": True, "started_at": self.started_at})

    def build_server(self) -> HttpServer:
        s = HttpServer()
        s.route("POST", "/v1/chat/completions", self.chat)
        s.route("POST", "/v1/completions", self.completions)
        s.route("GET", "/v1/models", self.models)
        s.route("GET", "/health", self.health)
        s.route("GET", "/", self.health)
        s.route("GET", "/metrics", self.metrics)
        s.route("POST", "/generate", self.tgi_generate)
        s.route("POST", "/generate_stream", self.tgi_generate)
        s.route("GET", "/info", self.tgi_info)
        s.route("GET", "/v2/health/ready", self.v2_ready)
        s.route("GET", "/v2", self.v2_ready)
        s.route("POST", "/v2/models/", self.v2_generate, prefix=True)
        s.route("POST", "/admin/chaos", self.admin)
        return s


def start_mock_in_thread(cfg: Optional[MockConfig] = None, port: int = 0):
    """(mock, base_url, stop) — a mock server on a background thread."""
    from .httpd import run_server_in_thread

    mock = MockLLM(cfg)
    _srv, p, stop = run_server_in_thread(mock.build_server, port=port)
    return mock, f"http://127.0.0.1:{p}", stop


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini mock", description="Deterministic mock LLM endpoint (OpenAI / TGI / KServe-v2)")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--model", default="mock-llm")
    ap.add_argument("--ttft-ms", type=float, default=8.0)
    ap.add_argument("--tpot-ms", type=float, default=1.0)
    ap.add_argument("--jitter", type=float, default=0.1)
    ap.add_argument("--error-rate", type=float, default=0.0)
    ap.add_argument("--cold-seconds", type=float, default=0.0)
    ap.add_argument("--cold-multiplier", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--service", default="mock-llm", help="Name to register in the local cluster registry")
    ap.add_argument("--register", action="store_true", help="Record this server in the local cluster registry")
    a = ap.parse_args(argv)
    cfg = MockConfig(model=a.model, ttft_ms=a.ttft_ms, tpot_ms=a.tpot_ms, jitter=a.jitter, error_rate=a.error_rate, cold_seconds=a.cold_seconds, cold_multiplier=a.cold_multiplier, seed=a.seed, api_key=a.api_key)
    mock = MockLLM(cfg)

    async def _main() -> None:
        srv = mock.build_server()
        port = await srv.start(a.host, a.port)
        print(f"kvmini mock listening on http://{a.host}:{port} (model={cfg.model})", flush=True)
        if a.register:
            import os

            from kvmini.cluster import LocalBackend

            LocalBackend().register_server(a.service, f"http://{a.host}:{port}", os.getpid(), mock.started_at, gpus=0, image="kvmini-mock")
        await srv.serve_forever()

    try:
        asyncio.run(_main())
    except KeyboardInterrupt:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
