"""Python side of ``csrc/hist_quantile.hip``: exact batched quantiles, 20-bin
histograms and a persistent streaming histogram on the GPU.

Exactness: the kernel selects the order statistics ``floor(p*(n-1))`` and
``ceil(p*(n-1))`` exactly (radix select on order-preserving keys); the linear
interpolation between them happens here in float64 with the reference's formula
(``analyze.py:75-81``), so results equal the CPU oracle bit for bit on float64
input.
"""

from __future__ import annotations

import math
import struct
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import gpu_available, native, native_built

BINS = 2048
# streaming histogram range: 2^-10 ms (~1 us) .. 2^20 ms (~17 min), ~68 bins per octave (~1 % wide)
LO_LOG2 = -10.0
HI_LOG2 = 20.0


def available() -> bool:
    return native_built() and gpu_available()


def _torch():
    import torch

    return torch


def _key_to_value(key: int, is_double: bool) -> float:
    if is_double:
        bits = (key & 0x7FFFFFFFFFFFFFFF) if (key >> 63) else (~key & 0xFFFFFFFFFFFFFFFF)
        return struct.unpack("<d", struct.pack("<Q", bits))[0]
    k32 = key >> 32
    bits = (k32 & 0x7FFFFFFF) if (k32 >> 31) else (~k32 & 0xFFFFFFFF)
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def _new_stream_state(m: int, device: Any):
    torch = _torch()
    C = native()
    sb = C.stream_state_bytes()
    one = np.zeros(sb, dtype=np.uint8)
    one[16:24] = 0xFF  # min_key = ~0
    state = torch.from_numpy(np.tile(one, m)).to(device)
    hist = torch.zeros((m, BINS), dtype=torch.int32, device=device)
    return state, hist


def _decode_stream_state(state_cpu: np.ndarray, m: int, is_double: bool) -> List[Dict[str, Any]]:
    sb = state_cpu.size // m
    out = []
    for i in range(m):
        count, total, kmin, kmax = struct.unpack("<QdQQ", state_cpu[i * sb : i * sb + 32].tobytes())
        d: Dict[str, Any] = {"count": count, "sum": total}
        if count:
            d["min"] = _key_to_value(kmin, is_double)
            d["max"] = _key_to_value(kmax, is_double)
            d["mean"] = total / count
        out.append(d)
    return out


def batched_quantiles(samples: Any, ps: Sequence[float] = (0.5, 0.95, 0.99), return_stats: bool = False):
    """Quantiles of every row of a ``[n_metrics, n]`` matrix (NaN = missing).

    ``samples``: numpy array or torch tensor, float32/float64.  Returns an array
    ``[n_metrics, len(ps)]`` (NaN for empty rows) and, optionally, per-row
    count/sum/min/max/mean."""
    torch = _torch()
    C = native()
    if not gpu_available():
        raise RuntimeError("HIP quantiles need a GPU")
    if not 1 <= len(ps) <= 4:
        raise ValueError("1..4 quantile levels per call")
    if isinstance(samples, np.ndarray):
        if samples.dtype not in (np.float32, np.float64):
            samples = samples.astype(np.float64)
        x = torch.from_numpy(np.ascontiguousarray(samples)).cuda()
    else:
        x = samples if samples.is_cuda else samples.cuda()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.double()
        x = x.contiguous()
    if x.dim() == 1:
        x = x.unsqueeze(0)
    m, n = x.shape
    is_double = x.dtype == torch.float64
    dev = x.device
    state, hist = _new_stream_state(m, dev)
    C.stream_hist(x, n, hist, state, LO_LOG2, HI_LOG2)
    rstate = torch.zeros(m * C.radix_state_bytes(), dtype=torch.uint8, device=dev)
    rhist = torch.zeros((m, 8, BINS), dtype=torch.int32, device=dev)
    ps_dev = torch.tensor(list(ps), dtype=torch.float64, device=dev)
    C.radix_select(x, n, state, ps_dev, rstate, rhist)
    rs = rstate.cpu().numpy()
    stats = _decode_stream_state(state.cpu().numpy(), m, is_double)
    rb = C.radix_state_bytes()
    out = np.full((m, len(ps)), np.nan, dtype=np.float64)
    for i in range(m):
        cnt = stats[i]["count"]
        if cnt == 0:
            continue
        prefixes = struct.unpack("<8Q", rs[i * rb : i * rb + 64].tobytes())
        for j, p in enumerate(ps):
            p = max(0.0, min(float(p), 1.0))
            pos = p * (cnt - 1)
            lo = math.floor(pos)
            v_lo = _key_to_value(prefixes[2 * j], is_double)
            v_hi = _key_to_value(prefixes[2 * j + 1], is_double)
            out[i, j] = v_lo if cnt == 1 or lo == math.ceil(pos) else v_lo + (pos - lo) * (v_hi - v_lo)
    return (out, stats) if return_stats else out


def _float_keys(x: np.ndarray) -> np.ndarray:
    """Order-preserving unsigned keys of float32 / float64 values (the kernels' ``KeyOf``): flip the
    sign bit of non-negative numbers, all bits of negative ones."""
    if x.dtype == np.float32:
        b = x.view(np.uint32)
        return np.where(b >> np.uint32(31), ~b, b | np.uint32(0x80000000)).astype(np.uint64) << np.uint64(32)
    b = x.view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | np.uint64(0x8000000000000000))


def distributed_quantiles(samples: Any, ps: Sequence[float], dist: Any, group: Any = None, device: Any = None) -> List[float]:
    """EXACT quantiles (the reference's linear interpolation, ``analyze.py:59-81``) of samples that
    are spread over the ranks of a ``torch.distributed`` group, without gathering them: an MSD
    radix select in which every rank histograms the next 11-bit digit of its own samples, the ranks
    sum the histograms (one all-reduce of 8 x 2048 counts per pass: 3 passes for float32, 6 for
    float64) and every rank walks the summed counts — so the selection state is the same
    everywhere and ends at the order statistics of the union.  NaN = missing.  On a GPU the
    passes are ``radix_hist_kernel`` / ``radix_scan_kernel`` (csrc/hist_quantile.hip) and the
    all-reduce is RCCL's; on the host the same protocol runs in numpy (CPU rehearsals, tests).
    Every rank returns the same list; all NaN when no rank has a sample."""
    torch = _torch()
    ps = [max(0.0, min(float(p), 1.0)) for p in ps]
    if not 1 <= len(ps) <= 4:
        raise ValueError("1..4 quantile levels per call")
    x = np.asarray(samples.cpu().numpy() if hasattr(samples, "cpu") else samples)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    x = np.ascontiguousarray(x.reshape(-1))
    x = x[~np.isnan(x)]
    is_double = x.dtype == np.float64
    dev = torch.device(device) if device is not None else torch.device("cuda" if available() else "cpu")
    on_gpu = dev.type == "cuda" and available()
    cnt = torch.tensor([x.size], dtype=torch.int64, device=dev if on_gpu else "cpu")
    dist.all_reduce(cnt, group=group)
    n_all = int(cnt.item())
    if n_all == 0:
        return [float("nan")] * len(ps)
    ranks: List[int] = []
    for p in ps:
        pos = p * (n_all - 1)
        ranks += [math.floor(pos), math.ceil(pos)]
    if on_gpu:
        C = native()
        xd = torch.from_numpy(x if x.size else np.zeros(1, dtype=x.dtype)).to(dev).reshape(1, -1)
        state, _h = _new_stream_state(1, dev)
        state.view(torch.int64)[0] = n_all  # the selection's ranks come from the global count
        rstate = torch.zeros(C.radix_state_bytes(), dtype=torch.uint8, device=dev)
        rhist = torch.zeros((1, 8, BINS), dtype=torch.int32, device=dev)
        C.radix_init(state, torch.tensor(ps, dtype=torch.float64, device=dev), rstate)
        for k in range(int(C.radix_num_passes(is_double))):
            C.radix_pass_hist(xd, int(x.size), rstate, rhist, k)
            dist.all_reduce(rhist, group=group)
            C.radix_pass_scan(rstate, rhist, is_double, k)
        prefixes = struct.unpack("<8Q", rstate.cpu().numpy()[:64].tobytes())
        vals = [_key_to_value(prefixes[i], is_double) for i in range(2 * len(ps))]
    else:
        keys = _float_keys(x) if x.size else np.zeros(0, dtype=np.uint64)
        key_bits = 64 if is_double else 32
        prefix = [0] * len(ranks)   # decided high bits, left-aligned in 64
        rem = list(ranks)           # rank remaining inside the prefix
        top, decided = key_bits, 0
        while top > 0:
            db = 11 if top >= 11 else top
            shift = top - db + (64 - key_bits)
            uniq = sorted(set(prefix))
            h = np.zeros((len(uniq), 1 << 11), dtype=np.int64)
            for u, pre in enumerate(uniq):
                sel = keys if decided == 0 else keys[(keys >> np.uint64(64 - decided)) == np.uint64(pre >> (64 - decided))]
                if sel.size:
                    h[u] = np.bincount(((sel >> np.uint64(shift)) & np.uint64((1 << db) - 1)).astype(np.int64), minlength=1 << 11)
            ht = torch.from_numpy(h)
            dist.all_reduce(ht, group=group)
            h = ht.numpy()
            for r in range(len(ranks)):
                row = h[uniq.index(prefix[r])]
                cum = np.cumsum(row)
                b = int(np.searchsorted(cum, rem[r], side="right"))
                rem[r] -= int(cum[b - 1]) if b else 0
                prefix[r] |= b << shift
            decided += db
            top -= db
        vals = [_key_to_value(pf, is_double) for pf in prefix]
    out = []
    for j, p in enumerate(ps):
        pos = p * (n_all - 1)
        lo = math.floor(pos)
        v_lo, v_hi = vals[2 * j], vals[2 * j + 1]
        out.append(v_lo if n_all == 1 or lo == math.ceil(pos) else v_lo + (pos - lo) * (v_hi - v_lo))
    return out


def quantiles(a: Iterable[float], ps: Sequence[float] = (0.5, 0.95, 0.99)) -> List[float]:
    arr = np.asarray(a, dtype=np.float64).reshape(1, -1)
    out: List[float] = []
    ps = list(ps)
    for i in range(0, len(ps), 4):
        out += batched_quantiles(arr, ps[i : i + 4])[0].tolist()
    return out


def equal_width_histogram(data: Iterable[float], num_bins: int = 20) -> Dict[str, Any]:
    """Same dict as ``analysis.histograms.compute_histogram`` (reference
    ``analyze.py:84-122``), computed on the GPU."""
    torch = _torch()
    C = native()
    arr = np.asarray(data, dtype=np.float64).reshape(1, -1)
    x = torch.from_numpy(arr).cuda()
    n = x.shape[1]
    state, hist = _new_stream_state(1, x.device)
    C.stream_hist(x, n, hist, state, LO_LOG2, HI_LOG2)
    out = torch.zeros((1, 64), dtype=torch.int64, device=x.device)
    C.equal_hist(x, n, state, out, num_bins)
    st = _decode_stream_state(state.cpu().numpy(), 1, True)[0]
    if st["count"] == 0:
        return {"bins": [], "counts": [], "bin_edges": []}
    lo, hi = st["min"], st["max"]
    if lo == hi:
        return {"bins": [lo], "counts": [int(st["count"])], "bin_edges": [lo, hi]}
    width = (hi - lo) / num_bins
    edges = [lo + i * width for i in range(num_bins + 1)]
    counts = out[0, :num_bins].cpu().tolist()
    return {
        "bins": [(edges[i] + edges[i + 1]) / 2 for i in range(num_bins)],
        "counts": [int(c) for c in counts],
        "bin_edges": edges,
        "min": lo,
        "max": hi,
        "total_samples": int(st["count"]),
    }


def log_bins_numpy(x: np.ndarray) -> np.ndarray:
    """Bin index of every sample, the formula of ``log_bin`` in ``csrc/hist_quantile.hip``
    (float32 ``log2``, truncation, clamped to ``[0, BINS)``; non-positive values -> bin 0)."""
    v = np.asarray(x, dtype=np.float32)
    inv_step = np.float32(BINS / (HI_LOG2 - LO_LOG2))
    with np.errstate(divide="ignore", invalid="ignore"):
        f = (np.log2(v, dtype=np.float32) - np.float32(LO_LOG2)) * inv_step
    f = np.where(v > 0, f, np.float32(0.0))
    return np.clip(f.astype(np.int64), 0, BINS - 1)


class TokenGaps:
    """Inter-token gaps from raw ``(request id, token index, t_ns)`` events, batch after batch,
    without per-event host work.  The gap of an event with index k > 0 is its time minus the time
    of the same request's previous event (in this batch, or remembered from an earlier one).

    On a GPU: the batch is copied once and three launches of ``csrc/hist_quantile.hip`` (insert
    into a batch-local hash table, emit the gaps, commit the per-request time table) leave an
    fp32 device vector (ms, NaN where an event has no gap) that :class:`StreamingHistogram` bins
    as it is.  Without one: the same result with a stable argsort by request in numpy."""

    def __init__(self, device: Any = "cuda", requests_hint: int = 4096):
        torch = _torch()
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        n = max(16, int(requests_hint))
        if self.on_gpu:
            self.last_t = torch.full((n,), -1, dtype=torch.int64, device=self.device)
            self._keys = self._vals = None
        else:
            self.last_np = np.full(n, -1, dtype=np.int64)

    def _grow(self, need: int) -> None:
        torch = _torch()
        if self.on_gpu:
            if need > self.last_t.numel():
                bigger = torch.full((max(need, 2 * self.last_t.numel()),), -1, dtype=torch.int64, device=self.device)
                bigger[: self.last_t.numel()] = self.last_t
                self.last_t = bigger
        elif need > self.last_np.size:
            bigger_np = np.full(max(need, 2 * self.last_np.size), -1, dtype=np.int64)
            bigger_np[: self.last_np.size] = self.last_np
            self.last_np = bigger_np

    def gaps(self, events: np.ndarray) -> Any:
        """``events``: int64 ``[n, 3]`` in arrival order.  Returns the gaps in ms: a device fp32
        tensor ``[n]`` with NaN for events without a gap (GPU), or a float32 numpy vector of the
        gaps only (host)."""
        torch = _torch()
        ev = np.ascontiguousarray(events, dtype=np.int64).reshape(-1, 3)
        n = ev.shape[0]
        if n == 0:
            return torch.empty(0, dtype=torch.float32, device=self.device) if self.on_gpu else np.empty(0, dtype=np.float32)
        if int(ev[:, 0].min()) < 0:
            raise ValueError("request ids must be non-negative")
        self._grow(int(ev[:, 0].max()) + 1)
        if self.on_gpu:
            slots = 1 << max(5, int(2 * n - 1).bit_length())
            if self._keys is None or self._keys.numel() < slots:
                self._keys = torch.empty(slots, dtype=torch.int64, device=self.device)
                self._vals = torch.empty(slots, dtype=torch.int64, device=self.device)
            ev_d = torch.from_numpy(ev).to(self.device, non_blocking=True)
            out = torch.empty(n, dtype=torch.float32, device=self.device)
            native().token_gaps(ev_d, self._keys[:slots], self._vals[:slots], self.last_t, out)
            return out
        tok = ev[ev[:, 1] >= 0]
        if not tok.shape[0]:
            return np.empty(0, dtype=np.float32)
        order = np.argsort(tok[:, 0], kind="stable")  # by request, arrival order kept inside a request
        req, idx, t = tok[order, 0], tok[order, 1], tok[order, 2]
        first = np.ones(req.size, dtype=bool)
        first[1:] = req[1:] != req[:-1]
        prev = np.empty_like(t)
        prev[1:] = t[:-1]
        prev[first] = self.last_np[req[first]]
        keep = (idx > 0) & (prev >= 0)
        last = np.ones(req.size, dtype=bool)
        last[:-1] = req[1:] != req[:-1]
        self.last_np[req[last]] = np.maximum(self.last_np[req[last]], t[last])
        return ((t[keep] - prev[keep]) * 1e-6).astype(np.float32)


class StreamingHistogram:
    """Persistent histogram for live inter-token-latency statistics.

    ``update()`` bins a new slice of samples (ms) into 2048 log-spaced bins per
    metric and keeps count/sum/min/max; nothing is ever re-read.  ``snapshot()``
    returns approximate quantiles (log-interpolated inside the bin, ~1 % bin
    width) — the live view; the exact offline numbers come from
    :func:`batched_quantiles` over the stored events.

    On a GPU the bins live on the device and ``update`` is one launch of
    ``stream_hist_kernel``; with ``device='cpu'`` (no GPU in the load generator's
    host, CI) the same bins are kept in numpy with the same binning formula.
    Histograms of several ranks merge by adding bins: :meth:`all_reduce`."""

    def __init__(self, n_metrics: int = 1, device: Any = "cuda"):
        torch = _torch()
        self.m = n_metrics
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.is_double = False
        if self.on_gpu:
            self.state, self.hist = _new_stream_state(n_metrics, self.device)
        else:
            self.hist = torch.zeros((n_metrics, BINS), dtype=torch.int32)
            self._count = np.zeros(n_metrics, dtype=np.int64)
            self._sum = np.zeros(n_metrics, dtype=np.float64)
            self._min = np.full(n_metrics, np.inf)
            self._max = np.full(n_metrics, -np.inf)

    def update(self, samples: Any) -> None:
        torch = _torch()
        if not self.on_gpu:
            x = np.asarray(samples.cpu().numpy() if hasattr(samples, "cpu") else samples)
            if x.ndim == 1:
                x = x.reshape(1, -1)
            assert x.shape[0] == self.m
            h = self.hist.numpy()
            for i in range(self.m):
                row = x[i][~np.isnan(x[i])]
                if not row.size:
                    continue
                h[i] += np.bincount(log_bins_numpy(row), minlength=BINS).astype(np.int32)
                self._count[i] += row.size
                self._sum[i] += float(row.astype(np.float64).sum())
                self._min[i] = min(self._min[i], float(row.min()))
                self._max[i] = max(self._max[i], float(row.max()))
            return
        x = torch.as_tensor(samples)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.float()
        self.is_double = x.dtype == torch.float64
        x = x.to(self.device).contiguous()
        if x.dim() == 1:
            x = x.unsqueeze(0)
        assert x.shape[0] == self.m
        if x.shape[1]:
            native().stream_hist(x, x.shape[1], self.hist, self.state, LO_LOG2, HI_LOG2)

    def stats(self) -> List[Dict[str, Any]]:
        if self.on_gpu:
            return _decode_stream_state(self.state.cpu().numpy(), self.m, self.is_double)
        out = []
        for i in range(self.m):
            d: Dict[str, Any] = {"count": int(self._count[i]), "sum": float(self._sum[i])}
            if d["count"]:
                d.update(min=float(self._min[i]), max=float(self._max[i]), mean=d["sum"] / d["count"])
            out.append(d)
        return out

    def merge_counts(self, other_hist: Any) -> None:
        """Add another rank's bin counts (after an all-reduce/all-gather)."""
        self.hist += other_hist.to(self.hist.device)

    def all_reduce(self, dist: Any, group: Any = None) -> None:
        """Merge the histograms of every rank of a ``torch.distributed`` group in place: one
        sum all-reduce of the ``[n_metrics, 2048]`` int32 bins (8 KiB per metric; RCCL's
        ``ncclAllReduce`` on GPUs) plus a 4-scalar exchange for count / sum / min / max."""
        torch = _torch()
        st = self.stats()
        sc = torch.tensor([[s["count"], s["sum"]] for s in st], dtype=torch.float64, device=self.device)
        mn = torch.tensor([s.get("min", float("inf")) for s in st], dtype=torch.float64, device=self.device)
        mx = torch.tensor([s.get("max", float("-inf")) for s in st], dtype=torch.float64, device=self.device)
        dist.all_reduce(self.hist, group=group)
        dist.all_reduce(sc, group=group)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        self._merged = [{"count": int(sc[i, 0].item()), "sum": float(sc[i, 1].item()), "min": float(mn[i].item()), "max": float(mx[i].item())} for i in range(self.m)]
        for d in self._merged:
            if d["count"]:
                d["mean"] = d["sum"] / d["count"]
            else:
                d.pop("min"), d.pop("max")

    def snapshot(self, ps: Sequence[float] = (0.5, 0.95, 0.99)) -> List[Dict[str, Any]]:
        hist = self.hist.cpu().numpy().astype(np.int64)
        stats = getattr(self, "_merged", None) or self.stats()
        stats = [dict(s) for s in stats]
        step = (HI_LOG2 - LO_LOG2) / BINS
        for i, st in enumerate(stats):
            n = int(hist[i].sum())
            st["quantiles"] = {}
            if n == 0:
                continue
            cum = np.cumsum(hist[i])
            for p in ps:
                k = p * (n - 1)
                b = int(np.searchsorted(cum, k, side="right"))
                b = min(b, BINS - 1)
                below = cum[b - 1] if b else 0
                frac = (k - below + 0.5) / max(1, hist[i][b])
                v = 2.0 ** (LO_LOG2 + (b + min(max(frac, 0.0), 1.0)) * step)
                if "min" in st:
                    v = min(max(v, st["min"]), st["max"])
                st["quantiles"][p] = v
        return stats
