"""Dependency-free asyncio HTTP/1.1 server used by the mock and the native engine server.

Keep-alive, ``Content-Length`` request bodies, fixed-length JSON responses and
chunked streaming responses (one HTTP chunk per SSE event, written with
TCP_NODELAY so the first token leaves the host immediately).  A tiny router maps
``(method, path)`` — exact or by prefix — to ``async handler(Request)``.
"""

from __future__ import annotations

import asyncio
import json
import socket
import urllib.parse
from dataclasses import dataclass, field
from typing import Any, AsyncIterator, Awaitable, Callable, Dict, List, Optional, Tuple, Union

_REASONS = {200: "OK", 204: "No Content", 400: "Bad Request", 401: "Unauthorized", 404: "Not Found", 405: "Method Not Allowed", 413: "Payload Too Large", 422: "Unprocessable Entity", 429: "Too Many Requests", 500: "Internal Server Error", 503: "Service Unavailable"}


@dataclass
class Request:
    method: str
    path: str
    query: Dict[str, List[str]]
    headers: Dict[str, str]
    body: bytes
    peer: Any = None

    def json(self) -> Any:
        return json.loads(self.body.decode("utf-8")) if self.body else None


@dataclass
class Response:
    status: int = 200
    body: bytes = b""
    headers: Dict[str, str] = field(default_factory=dict)
    content_type: str = "application/json"

    @classmethod
    def json(cls, obj: Any, status: int = 200, headers: Optional[Dict[str, str]] = None) -> "Response":
        return cls(status, json.dumps(obj, separators=(",", ":")).encode(), headers or {})

    @classmethod
    def text(cls, s: str, status: int = 200, content_type: str = "text/plain; charset=utf-8") -> "Response":
        return cls(status, s.encode(), {}, content_type)


@dataclass
class StreamingResponse:
    """``gen`` yields body pieces; each piece is sent as one HTTP chunk."""

    gen: AsyncIterator[bytes]
    status: int = 200
    headers: Dict[str, str] = field(default_factory=dict)
    content_type: str = "text/event-stream"


Handler = Callable[[Request], Awaitable[Union[Response, StreamingResponse]]]


def sse_event(obj: Any) -> bytes:
    if isinstance(obj, (bytes, bytearray)):
        return b"data: " + bytes(obj) + b"\n\n"
    if isinstance(obj, str):
        return b"data: " + obj.encode() + b"\n\n"
    return b"data: " + json.dumps(obj, separators=(",", ":")).encode() + b"\n\n"


class HttpServer:
    def __init__(self, max_body: int = 64 << 20):
        self._exact: Dict[Tuple[str, str], Handler] = {}
        self._prefix: List[Tuple[str, str, Handler]] = []
        self._server: Optional[asyncio.AbstractServer] = None
        self.max_body = max_body
        self.port: Optional[int] = None
        self.host: str = "127.0.0.1"
        self.requests_served = 0
        self._conns: set = set()

    def route(self, method: str, path: str, handler: Handler, prefix: bool = False) -> None:
        if prefix:
            self._prefix.append((method.upper(), path, handler))
        else:
            self._exact[(method.upper(), path)] = handler

    def _find(self, method: str, path: str) -> Tuple[Optional[Handler], bool]:
        h = self._exact.get((method, path))
        if h:
            return h, True
        for m, p, hh in self._prefix:
            if m == method and path.startswith(p):
                return hh, True
        known_path = any(pp == path for (_m, pp) in self._exact)
        return None, known_path

    async def start(self, host: str = "127.0.0.1", port: int = 0) -> int:
        self._server = await asyncio.start_server(self._client, host, port, limit=1 << 20, backlog=1024, reuse_address=True)
        sock = self._server.sockets[0]
        self.host, self.port = host, sock.getsockname()[1]
        return self.port

    async def serve_forever(self) -> None:
        assert self._server is not None
        async with self._server:
            await self._server.serve_forever()

    async def stop(self) -> None:
        if self._server is not None:
            self._server.close()
            for w in list(self._conns):
                try:
                    w.close()
                except Exception:
                    pass
            try:
                await asyncio.wait_for(self._server.wait_closed(), 2.0)
            except (asyncio.TimeoutError, Exception):
                pass
            self._server = None

    async def _client(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> None:
        sock = writer.get_extra_info("socket")
        if sock is not None:
            try:
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            except OSError:
                pass
        self._conns.add(writer)
        try:
            while True:
                line = await reader.readline()
                if not line:
                    break
                try:
                    method, target, _ver = line.decode("latin-1").rstrip("\r\n").split(" ", 2)
                except ValueError:
                    await self._send(writer, Response.json({"error": "bad request line"}, 400), close=True)
                    break
                headers: Dict[str, str] = {}
                while True:
                    h = await reader.readline()
                    if h in (b"\r\n", b"\n", b""):
                        break
                    k, _, v = h.decode("latin-1").partition(":")
                    headers[k.strip().lower()] = v.strip()
                n = int(headers.get("content-length", "0") or 0)
                if n > self.max_body:
                    await self._send(writer, Response.json({"error": "payload too large"}, 413), close=True)
                    break
                body = await reader.readexactly(n) if n else b""
                u = urllib.parse.urlsplit(target)
                req = Request(method.upper(), u.path, urllib.parse.parse_qs(u.query), headers, body, writer.get_extra_info("peername"))
                close = headers.get("connection", "").lower() == "close"
                handler, known = self._find(req.method, req.path)
                if handler is None:
                    resp: Union[Response, StreamingResponse] = Response.json({"error": {"message": "method not allowed" if known else f"no route for {req.path}", "type": "invalid_request_error"}}, 405 if known else 404)
                else:
                    try:
                        resp = await handler(req)
                    except json.JSONDecodeError as e:
                        resp = Response.json({"error": {"message": f"invalid JSON body: {e}", "type": "invalid_request_error"}}, 400)
                    except Exception as e:  # handler bug -> 500, keep serving
                        resp = Response.json({"error": {"message": f"{type(e).__name__}: {e}", "type": "internal_error"}}, 500)
                self.requests_served += 1
                if isinstance(resp, StreamingResponse):
                    ok = await self._send_stream(writer, resp, close)
                    if not ok:
                        break
                else:
                    await self._send(writer, resp, close)
                if close:
                    break
        except (ConnectionError, asyncio.IncompleteReadError, asyncio.CancelledError):
            pass
        finally:
            self._conns.discard(writer)
            try:
                writer.close()
            except Exception:
                pass

    @staticmethod
    def _head(status: int, headers: Dict[str, str]) -> bytes:
        return (f"HTTP/1.1 {status} {_REASONS.get(status, 'OK')}\r\n" + "".join(f"{k}: {v}\r\n" for k, v in headers.items()) + "\r\n").encode("latin-1")

    async def _send(self, writer: asyncio.StreamWriter, resp: Response, close: bool) -> None:
        headers = {"Content-Type": resp.content_type, "Content-Length": str(len(resp.body))}
        headers.update(resp.headers)
        if close:
            headers["Connection"] = "close"
        writer.write(self._head(resp.status, headers) + resp.body)
        await writer.drain()

    async def _send_stream(self, writer: asyncio.StreamWriter, resp: StreamingResponse, close: bool) -> bool:
        headers = {"Content-Type": resp.content_type, "Cache-Control": "no-cache", "Transfer-Encoding": "chunked"}
        headers.update(resp.headers)
        if close:
            headers["Connection"] = "close"
        writer.write(self._head(resp.status, headers))
        try:
            async for piece in resp.gen:
                if not piece:
                    continue
                writer.write(b"%x\r\n" % len(piece) + piece + b"\r\n")
                await writer.drain()
            writer.write(b"0\r\n\r\n")
            await writer.drain()
            return True
        except (ConnectionError, asyncio.CancelledError):
            # client went away: let the generator clean up (abort the request)
            aclose = getattr(resp.gen, "aclose", None)
            if aclose:
                try:
                    await aclose()
                except Exception:
                    pass
            return False


def run_server_in_thread(build: Callable[[], HttpServer], host: str = "127.0.0.1", port: int = 0) -> Tuple[Any, int, Callable[[], None]]:
    """Start a server on a private event loop in a daemon thread (tests, bench).
    Returns ``(server, port, stop)``."""
    import threading

    ready = threading.Event()
    box: Dict[str, Any] = {}

    def _run() -> None:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
        srv = build()
        box["srv"], box["loop"] = srv, loop
        box["port"] = loop.run_until_complete(srv.start(host, port))
        ready.set()
        try:
            loop.run_forever()
        finally:
            loop.run_until_complete(srv.stop())
            loop.close()

    th = threading.Thread(target=_run, daemon=True, name="kvmini-httpd")
    th.start()
    if not ready.wait(30):
        raise RuntimeError("server did not start")

    def stop() -> None:
        loop = box["loop"]
        loop.call_soon_threadsafe(loop.stop)
        th.join(5)

    return box["srv"], box["port"], stop
