"""Load generator, SSE parser, arrival patterns, tracing — against the mock server."""

import asyncio
import json
import os
import random

import pytest

from kvmini.loadgen.arrivals import PATTERNS, calculate_duration_and_rps, generate_arrival_times
from kvmini.loadgen.protocols import RequestSpec, count_tokens_from_outputs, get_protocol, update_tokens_from_stream_event
from kvmini.loadgen.runner import CSV_HEADER, LoadConfig, build_arg_parser, config_from_args, run_load_sync, synthetic_prompt
from kvmini.loadgen.sse import SSEParser
from kvmini.loadgen.tracing import TraceCollector, create_traceparent_header


def test_arrival_patterns():
    assert calculate_duration_and_rps(200, 10, "steady") == (10.0, 20.0)
    d, r = calculate_duration_and_rps(300, 10, "poisson")
    assert r == 15.0 and d == pytest.approx(30.0)
    rng = random.Random(0)
    for p in PATTERNS:
        d, r = calculate_duration_and_rps(400, 10, p)
        t = generate_arrival_times(p, 400, d, r, rng)
        assert len(t) == 400 and t == sorted(t) and t[0] >= 0
    st = generate_arrival_times("steady", 5, 1.0, 10.0)
    assert st == pytest.approx([0, 0.1, 0.2, 0.3, 0.4])
    d, r = calculate_duration_and_rps(1000, 10, "bursty")
    b = generate_arrival_times("bursty", 1000, d, r, random.Random(1))
    assert sum(1 for x in b if x <= 0.2 * d) == 800
    h = generate_arrival_times("heavy", 1000, d, r, random.Random(2))
    assert max(h) <= 0.95 * d + 1e-9
    with pytest.raises(ValueError):
        generate_arrival_times("nope", 1, 1, 1)
    # reproducible with a seed
    assert generate_arrival_times("poisson", 10, 1, 5, random.Random(7)) == generate_arrival_times("poisson", 10, 1, 5, random.Random(7))


def test_sse_parser_openai_framing_and_usage():
    p = SSEParser("openai")
    role = b'data: {"choices":[{"delta":{"role":"assistant","content":""}}]}\n\n'
    tok = b'data: {"choices":[{"delta":{"content":"hi"}}]}\n\n'
    usage = b'data: {"choices":[],"usage":{"prompt_tokens":3,"completion_tokens":2,"total_tokens":5}}\n\ndata: [DONE]\n\n'
    p.feed(role, 1.0)
    p.feed(tok[:10], 2.0)  # event split across chunks
    p.feed(tok[10:] + tok, 3.0)
    p.feed(usage, 4.0)
    assert p.first_byte_ms == 1.0 and p.first_token_ms == 3.0 and p.last_token_ms == 3.0
    assert p.content_events == 2 and p.usage["total_tokens"] == 5 and p.done and p.text == "hihi"
    # CRLF framing, comments, tool calls
    q = SSEParser("openai")
    q.feed(b': keep-alive\r\n\r\ndata: {"choices":[{"delta":{"tool_calls":[{"function":{"name":"f","arguments":"{}"}}]},"finish_reason":null}]}\r\n\r\n', 5.0)
    assert q.content_events == 1 and q.tool_calls == 1


def test_sse_parser_tgi_and_kserve():
    t = SSEParser("tgi")
    t.feed(b'data: {"token":{"id":1,"text":"a","special":false},"generated_text":null}\n\n', 1.0)
    t.feed(b'data: {"token":{"id":2,"text":"b","special":false},"generated_text":"ab","details":{"finish_reason":"length","generated_tokens":2}}\n\n', 2.0)
    assert t.content_events == 2 and t.usage == {"completion_tokens": 2} and t.generated_text == "ab"
    k = SSEParser("kserve-v2")
    k.feed(b'data: {"text_output":"x"}\n\ndata: {"text_output":"y"}\n\n', 1.0)
    assert k.content_events == 2 and k.text == "xy"


def test_token_count_helpers():
    assert count_tokens_from_outputs([{"name": "output_lengths", "data": [17]}, {"name": "text_output", "data": ["hello"]}]) == 17
    assert count_tokens_from_outputs([{"name": "text_output", "data": ["hello world!"]}], "hello world!") >= 2
    assert update_tokens_from_stream_event(0, 0, {"token_id": 123}) == (1, 0)
    tk, prev = update_tokens_from_stream_event(0, 0, {"text": "abcd"})
    assert tk >= 1 and prev == 4
    assert update_tokens_from_stream_event(3, 0, "garbage") == (3, 0)


def test_payload_builders():
    spec = RequestSpec(model="m", prompt="p", max_tokens=7, top_p=0.9, top_k=5, n=2, presence_penalty=0.1, json_mode=True, extra_payload={"use_beam_search": True})
    p = get_protocol("openai").payload(spec)
    assert p["messages"][0]["content"] == "p" and p["temperature"] == 0 and p["n"] == 2 and p["top_k"] == 5
    assert p["response_format"] == {"type": "json_object"} and p["use_beam_search"] is True
    assert p["stream_options"] == {"include_usage": True}
    t = get_protocol("tgi").payload(spec)
    assert t["inputs"] == "p" and t["parameters"]["max_new_tokens"] == 7 and t["parameters"]["do_sample"] is False
    k = get_protocol("kserve-v2")
    assert k.path(spec).endswith("/generate_stream")
    assert [i["name"] for i in k.payload(spec)["inputs"]] == ["text_input", "max_tokens", "bad_words", "stop_words"]


def test_tracing_otlp_export(tmp_path):
    tc = TraceCollector()
    s = tc.start_span("a" * 32, "client.request", attributes={"flag": True, "n": 3, "f": 1.5, "s": "x"})
    tc.finish_span(s, "error", {"why": "boom"})
    out = tmp_path / "t" / "traces.json"
    tc.export_traces(str(out))
    sp = json.loads(out.read_text())["resourceSpans"][0]["scopeSpans"][0]["spans"][0]
    attrs = {a["key"]: a["value"] for a in sp["attributes"]}
    assert attrs["flag"] == {"boolValue": True} and attrs["n"] == {"intValue": "3"} and attrs["f"] == {"doubleValue": 1.5}
    assert sp["status"]["code"] == 2
    assert create_traceparent_header("t", "s") == "00-t-s-01"


@pytest.mark.parametrize("protocol", ["openai", "openai-completions", "tgi", "kserve-v2"])
def test_load_run_streaming_all_protocols(mock_server, tmp_path, protocol):
    _mock, url = mock_server
    cfg = LoadConfig(url=url, run_dir=str(tmp_path / protocol), requests=20, concurrency=5, protocol=protocol, spec=RequestSpec(model="m", max_tokens=8), seed=1, quiet=True)
    res = run_load_sync(cfg)
    assert len(res) == 20 and all(r.status == 200 for r in res)
    r0 = res[0]
    assert r0.ttfb_ms is not None and r0.ttft_ms is not None and r0.ttfb_ms <= r0.ttft_ms <= r0.tllt_ms <= r0.latency_ms
    assert r0.completion_tokens == 8 and len(r0.token_times_ms) == 8
    header = (tmp_path / protocol / "requests.csv").read_text().splitlines()[0].split(",")
    assert header == CSV_HEADER and header[:12] == ["id", "scheduled_ms", "start_ms", "ttfb_ms", "tllt_ms", "latency_ms", "status", "prompt_tokens", "completion_tokens", "total_tokens", "error", "trace_id"]
    meta = json.loads((tmp_path / protocol / "meta.json").read_text())
    for k in ("url", "model", "max_tokens", "concurrency", "requests", "pattern", "duration_sec", "target_rps", "test_start_time"):
        assert k in meta
    spans = json.loads((tmp_path / protocol / "traces" / "traces.json").read_text())["resourceSpans"][0]["scopeSpans"][0]["spans"]
    assert {"client.request", "client.wait_scheduled", "http.request", "server.ttft", "server.tllt"} <= {s["name"] for s in spans}
    import numpy as np

    ev = np.load(tmp_path / protocol / "token_events.npy")
    assert ev.shape == (160, 3)


def test_load_run_non_stream_errors_and_pooling(mock_server, tmp_path):
    mock, url = mock_server
    cfg = LoadConfig(url=url, run_dir=str(tmp_path / "ns"), requests=12, concurrency=3, spec=RequestSpec(model="m", max_tokens=4, stream=False), quiet=True)
    res = run_load_sync(cfg)
    assert all(r.status == 200 and r.ttfb_ms is None and r.total_tokens and r.completion_tokens == 4 for r in res)
    assert json.loads((tmp_path / "ns" / "meta.json").read_text())["connections_opened"] <= 3
    mock.cfg.error_rate = 1.0
    bad = run_load_sync(LoadConfig(url=url, requests=5, concurrency=2, spec=RequestSpec(max_tokens=2), quiet=True))
    assert all(r.status == 500 for r in bad)
    mock.cfg.error_rate = 0.0
    dead = run_load_sync(LoadConfig(url="http://127.0.0.1:9", requests=2, concurrency=1, spec=RequestSpec(max_tokens=2), quiet=True, timeout_s=2))
    assert all(r.status == 0 and r.error for r in dead)
    # admit hook -> client-side 429
    thr = run_load_sync(LoadConfig(url=url, requests=4, concurrency=2, spec=RequestSpec(max_tokens=2), quiet=True, admit=lambda rid: rid % 2 == 0))
    assert sorted(r.status for r in thr) == [200, 200, 429, 429]


def test_cli_args_and_files(tmp_path, mock_server):
    _m, url = mock_server
    extra = tmp_path / "extra.json"
    extra.write_text(json.dumps({"use_beam_search": True}))
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("one two\nthree four five\n")
    a = build_arg_parser().parse_args(["--url", url, "--run-dir", str(tmp_path / "r"), "--requests", "4", "--concurrency", "2", "--extra-openai-json", str(extra), "--prompts-file", str(prompts), "--temperature", "0.5", "--header", "X-Tenant: A", "--quiet", "--seed", "3"])
    cfg = config_from_args(a)
    assert cfg.spec.extra_payload == {"use_beam_search": True} and cfg.prompts == ["one two", "three four five"] and cfg.extra_headers == {"X-Tenant": "A"} and cfg.spec.stream is True
    res = run_load_sync(cfg)
    assert {r.prompt_tokens for r in res} == {2, 3} and res[0].tenant == "A"
    assert len(synthetic_prompt(100).split()) == 100
    with pytest.raises(SystemExit):  # unknown flags are rejected, not silently forwarded
        build_arg_parser().parse_args(["--url", url, "--run-dir", "x", "--speculative-model", "foo"])


def test_sharded_load_generation_merges_artifacts(mock_server, tmp_path):
    """``--procs 2``: two generator processes, one merged run directory that the analyzer reads."""
    import csv
    import json

    import numpy as np

    from kvmini.analysis.reduce import analyze_run
    from kvmini.cluster import FakeBackend
    from kvmini.loadgen import runner

    _mock, url = mock_server
    run = tmp_path / "sharded"
    rc = runner.main(["--url", url, "--model", "m", "--requests", "21", "--concurrency", "5", "--max-tokens", "6", "--run-dir", str(run), "--procs", "2", "--quiet", "--target-rps", "500"])
    assert rc == 0
    rows = list(csv.DictReader(open(run / "requests.csv")))
    assert len(rows) == 21 and sorted(int(r["id"]) for r in rows) == list(range(1, 22))  # ids stay 1-based and unique after the merge
    assert all(r["status"] == "200" and int(float(r["completion_tokens"])) == 6 for r in rows)
    ev = np.load(run / "token_events.npy")
    assert ev.shape == (21 * 6, 3) and set(ev[:, 0].astype(int)) == set(range(1, 22))
    meta = json.load(open(run / "meta.json"))
    assert meta["generator_processes"] == 2 and meta["requests"] == 21 and meta["concurrency"] == 5
    res = analyze_run(str(run), "ns", "svc", backend=FakeBackend(), quiet=True)
    assert res["requests"] == {"total": 21, "success": 21} and res["itl_samples"] == 21 * 5


def test_live_inter_token_quantiles_final_flush_equals_offline(tmp_path):
    """--live-quantiles: token events go through the ring buffer into the streaming histogram
    WHILE the load runs (several updates before the last request ends); after the final flush
    the histogram holds exactly the gaps the offline analysis derives from token_events.npy —
    same count, same mean, quantiles inside one 1 % bin — and /metrics serves them."""
    import urllib.request

    import numpy as np

    from kvmini.analysis.reduce import inter_token_latency_metrics
    from kvmini.loadgen.live import LiveQuantiles
    from kvmini.loadgen.protocols import RequestSpec
    from kvmini.loadgen.runner import LoadConfig, run_load_sync
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    srv, url, stop = start_mock_in_thread(MockConfig(ttft_ms=5.0, tpot_ms=4.0, jitter=0.5, seed=3))
    try:
        live = LiveQuantiles(interval_s=0.05, device="cpu")
        port = live.serve_metrics(0)
        run_dir = str(tmp_path / "run")
        cfg = LoadConfig(url=url, run_dir=run_dir, requests=40, concurrency=8, spec=RequestSpec(model="mock", max_tokens=24, stream=True), quiet=True, live=live, seed=1)
        res = run_load_sync(cfg)
        assert all(r.status == 200 for r in res)
        off = inter_token_latency_metrics(run_dir, engine="python")
        fin = json.load(open(os.path.join(run_dir, "live_itl.json")))
        assert fin["itl_samples"] == off["itl_samples"] == sum(len(r.token_times_ms) - 1 for r in res)
        assert abs(fin["itl_mean_ms"] - off["itl_mean_ms"]) < 1e-3 * off["itl_mean_ms"]
        for k in ("itl_p50_ms", "itl_p95_ms", "itl_p99_ms"):
            assert abs(fin[k] - off[k]) <= 0.012 * off[k] + 1e-3, (k, fin[k], off[k])
        assert fin["updates"] >= 3 and fin["dropped_events"] == 0  # binned during the run, not after it
        assert len(fin["history"]) >= 3 and fin["engine"] == "numpy"
        assert port > 0 and "kvmini_itl_ms{quantile=\"0.95\"}" in live.metrics_text() and f"kvmini_itl_ms_count {fin['itl_samples']}" in live.metrics_text()
    finally:
        stop()
    # the ring alone: wrap-around keeps order, overflow drops the oldest and counts them
    from kvmini.loadgen.live import TokenEventRing

    ring = TokenEventRing(8)
    for i in range(5):
        ring.append(1, i, 100 + i)
    assert ring.take()[:, 2].tolist() == [100, 101, 102, 103, 104]
    for i in range(6):
        ring.append(2, i, 200 + i)
    assert ring.take()[:, 2].tolist() == [200, 201, 202, 203, 204, 205]  # wrapped
    for i in range(20):
        ring.append(3, i, 300 + i)
    assert ring.take()[:, 2].tolist() == list(range(312, 320)) and ring.dropped == 12


def test_traces_are_posted_to_an_otlp_collector(mock_server, tmp_path):
    """--otlp-endpoint: the spans of a run arrive at <endpoint>/v1/traces as OTLP/JSON batches that
    add up to the file export; a collector that is down does not fail the run."""
    import http.server
    import json
    import threading

    from kvmini.loadgen.protocols import RequestSpec
    from kvmini.loadgen.runner import LoadConfig, run_load_sync

    got = []

    class Collector(http.server.BaseHTTPRequestHandler):
        def do_POST(self):  # noqa: N802
            body = self.rfile.read(int(self.headers["Content-Length"]))
            got.append((self.path, self.headers["Content-Type"], json.loads(body)))
            self.send_response(200)
            self.end_headers()
            self.wfile.write(b"{}")

        def log_message(self, *a):
            pass

    srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), Collector)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    try:
        _mock, url = mock_server
        run = tmp_path / "r"
        res = run_load_sync(LoadConfig(url=url, run_dir=str(run), requests=8, concurrency=4, quiet=True, spec=RequestSpec(model="m", max_tokens=4),
                                       otlp_endpoint=f"http://127.0.0.1:{srv.server_address[1]}"))
        assert all(r.status == 200 for r in res)
        file_spans = json.load(open(run / "traces" / "traces.json"))["resourceSpans"][0]["scopeSpans"][0]["spans"]
        assert got and all(p == "/v1/traces" and "json" in ct for p, ct, _ in got)
        sent = [s for _, _, d in got for s in d["resourceSpans"][0]["scopeSpans"][0]["spans"]]
        assert [s["spanId"] for s in sent] == [s["spanId"] for s in file_spans] and len(sent) >= 8 * 3
        assert got[0][2]["resourceSpans"][0]["resource"]["attributes"][0]["key"] == "service.name"
    finally:
        srv.shutdown()
    # nobody listening: the run still completes and keeps its artifacts
    res = run_load_sync(LoadConfig(url=url, run_dir=str(tmp_path / "r2"), requests=2, concurrency=1, quiet=True, spec=RequestSpec(model="m", max_tokens=2), otlp_endpoint="http://127.0.0.1:9"))
    assert all(r.status == 200 for r in res) and (tmp_path / "r2" / "requests.csv").exists()

def _gaps_model(batches):
    """Per-event reference: the gap of an event with index > 0 is its time minus the time its request was last seen."""
    last, out = {}, []
    for ev in batches:
        g = []
        for req, idx, t in ev.tolist():
            if idx < 0:
                continue
            prev = last.get(req)
            if prev is not None and idx > 0:
                g.append((t - prev) * 1e-6)
            last[req] = t
        out.append(sorted(g))
    return out


def _event_batches(seed, n_req=300, n_batches=7):
    import numpy as np

    rng = np.random.default_rng(seed)
    evs = []
    for r in range(n_req):
        t = int(rng.integers(0, 5_000_000_000))
        for k in range(int(rng.integers(1, 90))):
            t += int(rng.integers(200_000, 30_000_000))
            evs.append((t, r, k))
        evs.append((t + 1, r, -1))  # end-of-request marker
    evs.sort()
    arr = np.array([(r, (k if k >= 0 else -1), (t if k >= 0 else 0)) for t, r, k in evs], dtype=np.int64)
    cuts = sorted(rng.integers(0, arr.shape[0], n_batches - 1).tolist())
    return [b for b in np.split(arr, cuts)]


def test_token_gaps_vectorised_host_path_equals_the_per_event_model():
    """The live path's gap computation without per-event Python: batches cut anywhere (inside a
    request's stream too), requests interleaved, end markers in between, the request table growing."""
    import numpy as np

    from kvmini.ops.hist_quantile import TokenGaps

    batches = _event_batches(3)
    want = _gaps_model(batches)
    tg = TokenGaps(device="cpu", requests_hint=16)
    for ev, w in zip(batches, want):
        got = np.sort(tg.gaps(ev))
        assert got.shape[0] == len(w) and np.allclose(got, np.asarray(w, dtype=np.float32), rtol=1e-6)
    assert tg.gaps(np.empty((0, 3), dtype=np.int64)).shape[0] == 0
    assert sum(len(w) for w in want) > 5000
