"""Data-parallel front-end: one endpoint in front of N engine replicas (``kvmini router``).

On a Kubernetes cluster Knative's activator / the Service spread requests over the
predictor pods (``isvc.yaml:9-12`` sets ``minScale`` / ``maxScale``).  On a bare 8 x MI355X
node the same role is played here: ``kvmini router --replicas 8 -- --model llama-3-8b`` starts
one ``kvmini serve`` process per GPU (``HIP_VISIBLE_DEVICES`` pins each to its device; a
replica that itself uses tensor parallelism gets ``--gpus-per-replica`` devices) and serves
the OpenAI / TGI routes on one port, forwarding every request to the replica with the fewest
requests in flight (ties: round robin) and streaming the response through chunk by chunk.
``--backends URL ...`` fronts servers that are already running instead.

``GET /health`` is healthy when at least one replica is; ``GET /metrics`` concatenates the
replicas' metrics with a ``replica`` label and adds the router's own counters;
``GET /v1/models`` comes from the first healthy replica.
"""

from __future__ import annotations

import argparse
import asyncio
import itertools
import os
import re
import signal
import subprocess
import sys
import time
from typing import Any, AsyncIterator, List, Optional

from kvmini.loadgen.httpclient import HttpClient
from kvmini.server.httpd import HttpServer, Request, Response, StreamingResponse

FORWARD_ROUTES = [("POST", "/v1/chat/completions"), ("POST", "/v1/completions"), ("POST", "/generate"), ("POST", "/generate_stream"), ("POST", "/tokenize"), ("POST", "/detokenize"), ("GET", "/v1/models"), ("GET", "/info")]


class Replica:
    def __init__(self, url: str, index: int):
        self.url = url.rstrip("/")
        self.index = index
        self.inflight = 0
        self.served = 0
        self.errors = 0
        self.healthy = True


class Router:
    def __init__(self, backends: List[str], timeout_s: float = 600.0):
        self.replicas = [Replica(u, i) for i, u in enumerate(backends)]
        self.client = HttpClient(pooled=True)
        self.timeout_s = timeout_s
        self._rr = itertools.count()

    def pick(self) -> Optional[Replica]:
        """Least requests in flight among the healthy replicas; round robin among equals."""
        live = [r for r in self.replicas if r.healthy] or self.replicas
        if not live:
            return None
        low = min(r.inflight for r in live)
        cands = [r for r in live if r.inflight == low]
        return cands[next(self._rr) % len(cands)]

    async def forward(self, req: Request) -> Any:
        headers = {k.title(): v for k, v in req.headers.items() if k in ("content-type", "authorization", "traceparent", "accept")}
        up = rep = None
        last_err: Optional[Exception] = None
        # a replica that cannot be reached has not seen the request: try the next one
        for _ in range(max(1, len(self.replicas))):
            rep = self.pick()
            if rep is None:
                break
            rep.inflight += 1
            try:
                up = await self.client.request(req.method, rep.url + req.path, body=req.body or None, headers=headers, timeout=self.timeout_s)
                break
            except Exception as e:
                rep.inflight -= 1
                rep.errors += 1
                rep.healthy = False  # the health loop brings it back
                last_err = e
                if not any(r.healthy for r in self.replicas):
                    break
        if up is None or rep is None:
            return Response.json({"error": {"message": f"no replica available: {last_err}", "type": "service_unavailable"}}, 503)
        ctype = up.headers.get("content-type", "application/json")
        if "text/event-stream" in ctype or "chunked" in up.headers.get("transfer-encoding", "").lower():

            async def relay() -> AsyncIterator[bytes]:
                try:
                    async for piece in up.iter_bytes():
                        yield piece
                    rep.served += 1
                except Exception:
                    rep.errors += 1
                    raise
                finally:
                    rep.inflight -= 1

            return StreamingResponse(relay(), up.status, {"X-Kvmini-Replica": str(rep.index)}, ctype)
        try:
            body = b"".join([piece async for piece in up.iter_bytes()])
            rep.served += 1
        except Exception as e:
            rep.errors += 1
            return Response.json({"error": {"message": f"replica {rep.index} failed mid-response: {e}", "type": "bad_gateway"}}, 503)
        finally:
            rep.inflight -= 1
        return Response(up.status, body, {"X-Kvmini-Replica": str(rep.index)}, ctype)

    async def health(self, _req: Request) -> Response:
        n = sum(1 for r in self.replicas if r.healthy)
        return Response.json({"status": "ok" if n else "unavailable", "replicas": len(self.replicas), "healthy": n}, 200 if n else 503)

    async def metrics(self, _req: Request) -> Response:
        lines = ["# TYPE kvmini_router_requests_total counter", "# TYPE kvmini_router_inflight gauge"]
        for r in self.replicas:
            lb = f'{{replica="{r.index}"}}'
            lines += [f"kvmini_router_requests_total{lb} {r.served}", f"kvmini_router_errors_total{lb} {r.errors}", f"kvmini_router_inflight{lb} {r.inflight}", f"kvmini_router_replica_healthy{lb} {int(r.healthy)}"]
        for r in self.replicas:
            if not r.healthy:
                continue
            try:
                up = await self.client.request("GET", r.url + "/metrics", timeout=5.0)
                text = b"".join([p async for p in up.iter_bytes()]).decode()
            except Exception:
                continue
            for ln in text.splitlines():
                if not ln or ln.startswith("#"):
                    continue
                m = re.match(r"^([A-Za-z_:][A-Za-z0-9_:]*)(\{[^}]*\})?\s+(.*)$", ln)
                if m:
                    labels = m.group(2)
                    labels = f'{{replica="{r.index}"}}' if not labels else labels[:-1] + f',replica="{r.index}"}}'
                    lines.append(f"{m.group(1)}{labels} {m.group(3)}")
        return Response.text("\n".join(lines) + "\n", content_type="text/plain; version=0.0.4")

    async def health_loop(self, period_s: float = 2.0) -> None:
        while True:
            for r in self.replicas:
                try:
                    up = await self.client.request("GET", r.url + "/health", timeout=2.0)
                    async for _ in up.iter_bytes():
                        pass
                    r.healthy = up.status == 200
                except Exception:
                    r.healthy = False
            await asyncio.sleep(period_s)

    def build_server(self) -> HttpServer:
        s = HttpServer()
        for method, path in FORWARD_ROUTES:
            s.route(method, path, self.forward)
        s.route("GET", "/health", self.health)
        s.route("GET", "/", self.health)
        s.route("GET", "/metrics", self.metrics)
        return s


def spawn_replicas(n: int, base_port: int, serve_args: List[str], gpus_per_replica: int = 1, device: str = "cuda", log_dir: str = "runs/.local_logs") -> List[subprocess.Popen]:
    """One ``kvmini serve`` per replica, each pinned to its GPU(s).  The pinning is done with
    ``HIP_VISIBLE_DEVICES`` in the CHILD's environment, before that process touches the GPU."""
    os.makedirs(log_dir, exist_ok=True)
    pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs = []
    for i in range(n):
        env = dict(os.environ)
        env["PYTHONPATH"] = pkg_root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        if device != "cpu":
            env["HIP_VISIBLE_DEVICES"] = ",".join(str(i * gpus_per_replica + k) for k in range(gpus_per_replica))
        port = base_port + 1 + i
        cmd = [sys.executable, "-m", "kvmini", "serve"]
        if gpus_per_replica > 1:  # the command starts its own ranks (kvmini.server.native.launch_tensor_parallel)
            cmd += ["--tensor-parallel-size", str(gpus_per_replica)]
        cmd += ["--port", str(port), "--service", f"replica-{i}", "--no-register", *serve_args]
        log = open(os.path.join(log_dir, f"router-replica-{i}.log"), "ab")
        procs.append(subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, start_new_session=True))
    return procs


async def wait_ready(urls: List[str], procs: List[subprocess.Popen], timeout_s: float) -> None:
    client = HttpClient(pooled=False)
    deadline = time.time() + timeout_s
    pending = set(range(len(urls)))
    while pending and time.time() < deadline:
        for i in list(pending):
            if procs and procs[i].poll() is not None:
                raise SystemExit(f"replica {i} exited with code {procs[i].returncode} (see runs/.local_logs/router-replica-{i}.log)")
            try:
                up = await client.request("GET", urls[i] + "/health", timeout=1.0)
                async for _ in up.iter_bytes():
                    pass
                if up.status == 200:
                    pending.discard(i)
            except Exception:
                pass
        if pending:
            await asyncio.sleep(0.3)
    if pending:
        raise SystemExit(f"replicas {sorted(pending)} not ready after {timeout_s:.0f}s")


def main(argv: Optional[List[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    serve_args: List[str] = []
    if "--" in argv:
        k = argv.index("--")
        argv, serve_args = argv[:k], argv[k + 1:]
    ap = argparse.ArgumentParser(prog="kvmini router", description="Data-parallel front-end: least-loaded routing over N engine replicas (flags after `--` go to every `kvmini serve`)")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--replicas", type=int, default=0, help="start this many `kvmini serve` processes (ports port+1 ...), one GPU each")
    ap.add_argument("--gpus-per-replica", type=int, default=1, help="tensor-parallel degree of every replica")
    ap.add_argument("--backends", nargs="*", default=[], help="URLs of already-running servers to front instead of spawning replicas")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu: do not pin GPUs (tests)")
    ap.add_argument("--ready-timeout", type=float, default=900.0)
    ap.add_argument("--service", default="demo-llm", help="name in the local cluster registry")
    ap.add_argument("--namespace", default="local")
    ap.add_argument("--no-register", action="store_true")
    a = ap.parse_args(argv)
    if not a.replicas and not a.backends:
        ap.error("give --replicas N or --backends URL ...")

    procs: List[subprocess.Popen] = []
    urls = list(a.backends)
    if a.replicas:
        if a.device == "cpu" and "--device" not in serve_args:
            serve_args += ["--device", "cpu"]
        procs = spawn_replicas(a.replicas, a.port, serve_args, a.gpus_per_replica, a.device)
        urls = [f"http://127.0.0.1:{a.port + 1 + i}" for i in range(a.replicas)]
    t0 = time.time()

    def _term(*_: Any) -> None:
        raise KeyboardInterrupt

    signal.signal(signal.SIGTERM, _term)

    async def run() -> None:
        await wait_ready(urls, procs, a.ready_timeout)
        router = Router(urls)
        srv = router.build_server()
        port = await srv.start(a.host, a.port)
        if not a.no_register:
            from kvmini.cluster import LocalBackend

            LocalBackend().register_server(a.service, f"http://{a.host}:{port}", os.getpid(), t0, namespace=a.namespace, gpus=max(1, len(urls)) * a.gpus_per_replica if a.device != "cpu" else 0)
        print(f"kvmini router: {len(urls)} replicas behind http://{a.host}:{port} (ready in {time.time() - t0:.1f}s)", flush=True)
        hl = asyncio.ensure_future(router.health_loop())
        try:
            await srv.serve_forever()
        finally:
            hl.cancel()

    try:
        asyncio.run(run())
    except KeyboardInterrupt:
        pass
    finally:
        for p in procs:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except OSError:
                pass
        for p in procs:
            try:
                p.wait(20)
            except subprocess.TimeoutExpired:
                p.kill()
        if not a.no_register:
            from kvmini.cluster import LocalBackend

            LocalBackend().mark_finished(os.getpid())
    return 0


if __name__ == "__main__":
    sys.exit(main())
