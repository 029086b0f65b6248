"""Minimal pooled asyncio HTTP/1.1 client for the load generator.

The reference opens a brand-new ``httpx.AsyncClient`` (TCP + TLS handshake) for
every request (``scripts/loadtest.py:407-409``), which puts connection setup into
every TTFB sample.  This client keeps per-origin keep-alive connections in a
pool (``pooled=False`` restores connection-per-request for parity runs), speaks
just enough HTTP/1.1 for OpenAI/TGI/KServe-v2 endpoints — ``Content-Length``,
``Transfer-Encoding: chunked`` and read-until-close bodies — and hands body bytes
to the caller *as they arrive* so the SSE parser can timestamp every token.
"""

from __future__ import annotations

import asyncio
import json
import ssl
import time
import urllib.parse
from dataclasses import dataclass, field
from typing import Any, AsyncIterator, Dict, List, Optional, Tuple


class HttpError(Exception):
    pass


@dataclass
class _Conn:
    reader: asyncio.StreamReader
    writer: asyncio.StreamWriter
    reusable: bool = True


@dataclass
class Response:
    status: int
    headers: Dict[str, str]
    _client: "HttpClient"
    _origin: Tuple[str, int, bool]
    _conn: _Conn
    _done: bool = False
    _chunks: List[bytes] = field(default_factory=list)

    async def iter_bytes(self) -> AsyncIterator[bytes]:
        """Yield body bytes in network arrival order (one HTTP chunk per item for
        chunked bodies)."""
        rd = self._conn.reader
        te = self.headers.get("transfer-encoding", "").lower()
        try:
            if "chunked" in te:
                while True:
                    line = await rd.readline()
                    if not line:
                        raise HttpError("connection closed inside chunked body")
                    size = int(line.split(b";", 1)[0].strip() or b"0", 16)
                    if size == 0:
                        # trailers until blank line
                        while True:
                            t = await rd.readline()
                            if t in (b"\r\n", b"\n", b""):
                                break
                        break
                    data = await rd.readexactly(size + 2)
                    yield data[:-2]
            elif "content-length" in self.headers:
                remaining = int(self.headers["content-length"])
                while remaining > 0:
                    data = await rd.read(min(65536, remaining))
                    if not data:
                        raise HttpError("connection closed before content-length bytes arrived")
                    remaining -= len(data)
                    yield data
            else:
                self._conn.reusable = False
                while True:
                    data = await rd.read(65536)
                    if not data:
                        break
                    yield data
        except BaseException:
            self._conn.reusable = False
            raise
        finally:
            self._finish()

    async def read(self) -> bytes:
        async for c in self.iter_bytes():
            self._chunks.append(c)
        return b"".join(self._chunks)

    async def json(self) -> Any:
        return json.loads((await self.read()).decode("utf-8", "replace"))

    def _finish(self) -> None:
        if self._done:
            return
        self._done = True
        if self.headers.get("connection", "").lower() == "close":
            self._conn.reusable = False
        self._client._release(self._origin, self._conn)

    def close(self) -> None:
        """Abandon the response without reading the body (connection is dropped)."""
        if not self._done:
            self._conn.reusable = False
            self._finish()


class HttpClient:
    def __init__(self, pooled: bool = True, verify_tls: bool = True, connect_timeout: float = 10.0, max_idle_per_origin: int = 256, connect_retry_s: float = 0.0):
        self.pooled = pooled
        self.verify_tls = verify_tls
        self.connect_timeout = connect_timeout
        # keep knocking on a refused port for this long: what Knative's activator does for a
        # scaled-to-zero / restarting revision (the request waits, its latency shows the cold start)
        self.connect_retry_s = connect_retry_s
        self.max_idle = max_idle_per_origin
        self._idle: Dict[Tuple[str, int, bool], List[_Conn]] = {}
        self.connections_opened = 0

    async def _connect(self, origin: Tuple[str, int, bool]) -> _Conn:
        host, port, tls = origin
        ctx = None
        if tls:
            ctx = ssl.create_default_context()
            if not self.verify_tls:
                ctx.check_hostname = False
                ctx.verify_mode = ssl.CERT_NONE
        deadline = time.monotonic() + self.connect_retry_s
        while True:
            try:
                reader, writer = await asyncio.wait_for(asyncio.open_connection(host, port, ssl=ctx, limit=1 << 20), self.connect_timeout)
                break
            except (ConnectionRefusedError, ConnectionResetError):
                if time.monotonic() >= deadline:
                    raise
                await asyncio.sleep(0.05)
        sock = writer.get_extra_info("socket")
        if sock is not None:
            import socket

            try:
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            except OSError:
                pass
        self.connections_opened += 1
        return _Conn(reader, writer)

    def _release(self, origin: Tuple[str, int, bool], conn: _Conn) -> None:
        if self.pooled and conn.reusable and not conn.reader.at_eof():
            pool = self._idle.setdefault(origin, [])
            if len(pool) < self.max_idle:
                pool.append(conn)
                return
        try:
            conn.writer.close()
        except Exception:
            pass

    async def request(
        self,
        method: str,
        url: str,
        headers: Optional[Dict[str, str]] = None,
        json_body: Any = None,
        body: Optional[bytes] = None,
        timeout: float = 60.0,
    ) -> Response:
        """Send the request and return once the response *headers* have arrived."""
        u = urllib.parse.urlsplit(url)
        tls = u.scheme == "https"
        host = u.hostname or "127.0.0.1"
        port = u.port or (443 if tls else 80)
        origin = (host, port, tls)
        path = u.path or "/"
        if u.query:
            path += "?" + u.query
        if json_body is not None:
            body = json.dumps(json_body, separators=(",", ":")).encode()
        hdrs = {"Host": u.netloc, "Accept": "*/*", "User-Agent": "kvmini-loadgen/0.1"}
        if headers:
            hdrs.update(headers)
        if body is not None:
            hdrs["Content-Length"] = str(len(body))
        if not self.pooled:
            hdrs["Connection"] = "close"
        head = f"{method} {path} HTTP/1.1\r\n" + "".join(f"{k}: {v}\r\n" for k, v in hdrs.items()) + "\r\n"
        payload = head.encode("latin-1") + (body or b"")

        last_exc: Optional[BaseException] = None
        for attempt in range(2):
            reused = False
            conn: Optional[_Conn] = None
            pool = self._idle.get(origin)
            while self.pooled and pool:
                c = pool.pop()
                if c.reader.at_eof() or c.writer.is_closing():
                    try:
                        c.writer.close()
                    except Exception:
                        pass
                    continue
                conn, reused = c, True
                break
            if conn is None:
                conn = await self._connect(origin)
            try:
                conn.writer.write(payload)
                await conn.writer.drain()
                status_line = await asyncio.wait_for(conn.reader.readline(), timeout)
                if not status_line:
                    raise HttpError("connection closed before status line")
                parts = status_line.decode("latin-1").split(" ", 2)
                status = int(parts[1])
                resp_headers: Dict[str, str] = {}
                while True:
                    line = await asyncio.wait_for(conn.reader.readline(), timeout)
                    if line in (b"\r\n", b"\n", b""):
                        break
                    k, _, v = line.decode("latin-1").partition(":")
                    resp_headers[k.strip().lower()] = v.strip()
                return Response(status, resp_headers, self, origin, conn)
            except (HttpError, ConnectionError, asyncio.IncompleteReadError, OSError) as e:
                last_exc = e
                try:
                    conn.writer.close()
                except Exception:
                    pass
                # a pooled connection the server closed while idle: retry once on a fresh one
                if reused and attempt == 0:
                    continue
                raise
        raise HttpError(str(last_exc))

    async def get(self, url: str, **kw: Any) -> Response:
        return await self.request("GET", url, **kw)

    async def post(self, url: str, **kw: Any) -> Response:
        return await self.request("POST", url, **kw)

    async def aclose(self) -> None:
        for pool in self._idle.values():
            for c in pool:
                try:
                    c.writer.close()
                except Exception:
                    pass
        self._idle.clear()

    async def __aenter__(self) -> "HttpClient":
        return self

    async def __aexit__(self, *exc: Any) -> None:
        await self.aclose()
