"""Live inter-token latency: token events -> ring buffer -> streaming histogram, while the load runs.

The reference has one aggregate per request, ``(tllt - ttfb) / completion_tokens``
(``analyze.py:143-146``), computed after the run.  Here every content event the SSE parser
sees is appended as ``(request id, token index, t_ns)`` to a preallocated ring; a background
task wakes every ``interval_s`` (250 ms) and takes the events that arrived since its last pass
(the *delta* — nothing is re-read) in ONE copy.  The raw events go to the device as they are:
``kvmini.ops.hist_quantile.TokenGaps`` (three small kernels around a batch-local hash table and a
per-request "last seen" table that lives on the device) turns them into inter-token gaps, and
``StreamingHistogram`` (one launch of ``stream_hist_kernel``) bins the gaps into the persistent
2048-bin log-spaced histogram — no per-event work on the host.  Without a GPU on the load
generator's host the same two steps run vectorised in numpy.  Rolling
p50 / p95 / p99 are printed and exported as Prometheus text; after the last request the final
flush holds exactly the gaps the offline analysis derives from ``token_events.npy``.
"""

from __future__ import annotations

import asyncio
import sys
import threading
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

QS = (0.5, 0.95, 0.99)


class TokenEventRing:
    """Single-producer / single-consumer ring of ``(request id, token index, t_ns)`` records.
    The producer (the event loop that parses SSE) only ever advances ``head``, the consumer
    only ``tail``; a full ring drops the oldest unread records and counts them."""

    def __init__(self, capacity: int = 1 << 18):
        self.cap = int(capacity)
        self.buf = np.zeros((self.cap, 3), dtype=np.int64)
        self.head = 0  # records ever written
        self.tail = 0  # records ever read
        self.dropped = 0

    def append(self, req: int, idx: int, t_ns: int) -> None:
        row = self.buf[self.head % self.cap]
        row[0], row[1], row[2] = req, idx, t_ns
        self.head += 1

    def take(self) -> np.ndarray:
        """The records written since the last call (a copy), oldest first."""
        head = self.head
        n = head - self.tail
        if n <= 0:
            return np.empty((0, 3), dtype=np.int64)
        if n > self.cap:
            self.dropped += n - self.cap
            self.tail = head - self.cap
            n = self.cap
        a, b = self.tail % self.cap, head % self.cap
        out = self.buf[a:b].copy() if a < b else np.concatenate((self.buf[a:], self.buf[:b]))
        self.tail = head
        return out


class LiveQuantiles:
    def __init__(self, interval_s: float = 0.25, device: Optional[str] = None, capacity: int = 1 << 18, verbose: bool = False):
        from kvmini.ops import hist_quantile as hq

        if device is None:
            device = "cuda" if hq.available() else "cpu"
        self.hist = hq.StreamingHistogram(1, device=device)
        self.engine = "hip:token_gaps+stream_hist_kernel" if self.hist.on_gpu else "numpy"
        self.ring = TokenEventRing(capacity)
        self.interval_s = float(interval_s)
        self.verbose = verbose
        self.gaps = hq.TokenGaps(device=device)  # per-request "last seen" table + the gap kernels (numpy on a host without a GPU)
        self.updates = 0
        self.gaps_seen = 0
        self.history: List[Tuple[float, Dict[float, float]]] = []  # (wall time, quantiles) per update
        self._task: Optional["asyncio.Task[None]"] = None
        self._stop = False
        self._lock = threading.Lock()
        self._metrics_srv: Any = None

    # ---- producer side (SSE parser hook) ---------------------------------
    def on_token(self, req: int, idx: int, t_ns: int) -> None:
        self.ring.append(req, idx, t_ns)

    def finished(self, req: int) -> None:
        """A request is over: its last timestamp is not needed once its events are drained."""
        self.ring.append(req, -1, 0)

    # ---- consumer side -----------------------------------------------------
    def drain(self) -> int:
        """Bin the gaps of the events that arrived since the last pass: one ``take()`` of the
        ring's new slice, then the gap kernels and the histogram kernel on it (no per-event host
        work).  Returns the number of token events that can carry a gap (index > 0)."""
        ev = self.ring.take()
        if not ev.shape[0]:
            return 0
        n = int((ev[:, 1] > 0).sum())
        g = self.gaps.gaps(ev)
        if n:
            with self._lock:
                self.hist.update(g)
                self.updates += 1
                self.gaps_seen += n
        return n

    def snapshot(self) -> Dict[str, Any]:
        with self._lock:
            st = self.hist.snapshot(QS)[0]
        q = st.get("quantiles", {})
        return {"itl_p50_ms": q.get(0.5), "itl_p95_ms": q.get(0.95), "itl_p99_ms": q.get(0.99), "itl_mean_ms": st.get("mean"), "itl_samples": int(st.get("count", 0)),
                "itl_min_ms": st.get("min"), "itl_max_ms": st.get("max"), "updates": self.updates, "dropped_events": self.ring.dropped, "engine": self.engine}

    async def run(self) -> None:
        t0 = time.time()
        while not self._stop:
            await asyncio.sleep(self.interval_s)
            if self.drain() and (self.verbose or self._metrics_srv is not None):
                s = self.snapshot()
                self.history.append((time.time(), {0.5: s["itl_p50_ms"], 0.95: s["itl_p95_ms"], 0.99: s["itl_p99_ms"]}))
                if self.verbose:
                    print(f"[live {time.time() - t0:6.1f}s] inter-token ms p50 {s['itl_p50_ms']:.3f}  p95 {s['itl_p95_ms']:.3f}  p99 {s['itl_p99_ms']:.3f}  ({s['itl_samples']} gaps, {self.engine})",
                          file=sys.stderr, flush=True)

    def start(self) -> None:
        self._stop = False
        self._task = asyncio.get_running_loop().create_task(self.run())

    async def stop(self) -> Dict[str, Any]:
        """Final flush: everything still in the ring is binned; returns the final snapshot."""
        self._stop = True
        if self._task is not None:
            self._task.cancel()
            try:
                await self._task
            except asyncio.CancelledError:
                pass
        self.drain()
        if self._metrics_srv is not None:
            self._metrics_srv.shutdown()
        return self.snapshot()

    # ---- Prometheus text ---------------------------------------------------
    def metrics_text(self) -> str:
        s = self.snapshot()
        lines = ["# TYPE kvmini_itl_ms summary"]
        for q, key in ((0.5, "itl_p50_ms"), (0.95, "itl_p95_ms"), (0.99, "itl_p99_ms")):
            if s[key] is not None:
                lines.append(f'kvmini_itl_ms{{quantile="{q}"}} {s[key]:.6f}')
        lines += [f"kvmini_itl_ms_count {s['itl_samples']}", f"kvmini_itl_ms_sum {(s['itl_mean_ms'] or 0.0) * s['itl_samples']:.6f}",
                  "# TYPE kvmini_itl_histogram_updates_total counter", f"kvmini_itl_histogram_updates_total {s['updates']}",
                  f"kvmini_itl_dropped_events_total {s['dropped_events']}"]
        return "\n".join(lines) + "\n"

    def serve_metrics(self, port: int) -> int:
        """``/metrics`` on a background thread; returns the bound port."""
        from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

        live = self

        class H(BaseHTTPRequestHandler):
            def do_GET(self) -> None:  # noqa: N802
                body = live.metrics_text().encode()
                self.send_response(200)
                self.send_header("Content-Type", "text/plain; version=0.0.4")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def log_message(self, *a: Any) -> None:
                pass

        self._metrics_srv = ThreadingHTTPServer(("127.0.0.1", port), H)
        threading.Thread(target=self._metrics_srv.serve_forever, daemon=True, name="kvmini-live-metrics").start()
        return int(self._metrics_srv.server_address[1])
