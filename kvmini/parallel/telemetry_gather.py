"""Lock-step multi-GPU telemetry over RCCL (``torch.distributed`` backend ``nccl``).

The reference gets one node-summed power number per tick from Prometheus
(``energy/collector.py:226-242``).  On an 8 x MI355X node every rank (one process
per GPU) samples its own GPU with amd-smi and the per-rank snapshots are
all-gathered over xGMI so that every rank — and the report — sees the same
``[world, ticks, fields]`` table with rows aligned on a common tick grid.

A snapshot is 16 float64 = 128 bytes: ``rank, seq, t_s, power_w, gfx_util,
vram_used, vram_total, temp_c, sclk_mhz, energy_uj`` + 6 reserved.  The payload
is latency-bound (128 B per rank per tick), so two collective shapes are offered:

* :meth:`TelemetryGather.gather_tick` — one ``all_gather_into_tensor`` of
  ``world x 128 B`` per tick on a dedicated communicator and side stream (the
  live daemon: ``kvmini telemetry``); a tick is *emitted* only when all slots
  carry the same ``seq`` (lock-step check).
* :meth:`TelemetryGather.gather_table` — resample the locally buffered samples
  onto the shared tick grid ``t0 + k*dt`` and exchange the whole table in ONE
  collective at a step boundary (what ``bench.py`` does: nothing competes with
  the serving stream during the timed region, and fewer, larger collectives are
  what the point-to-point xGMI links like).

On CPU (tests) the same code runs over ``gloo``.
"""

from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

FIELDS = ("rank", "seq", "t_s", "power_w", "gfx_util", "vram_used", "vram_total", "temp_c", "sclk_mhz", "energy_uj")
SNAPSHOT_F64 = 16  # 128 bytes


def snapshot_vector(rank: int, seq: int, sample: Optional[Dict[str, Any]]) -> np.ndarray:
    v = np.full(SNAPSHOT_F64, np.nan, dtype=np.float64)
    v[0], v[1] = rank, seq
    if sample:
        g = (sample.get("per_gpu") or [{}])[0] if sample.get("per_gpu") else {}
        v[2] = sample.get("ts_s", np.nan)
        for i, key in ((3, "watts"), (4, "gfx_util"), (5, "vram_used"), (7, "temp_c"), (8, "sclk_mhz")):
            val = sample.get(key)
            v[i] = np.nan if val is None else float(val)
        if g.get("vram_total") is not None:
            v[6] = float(g["vram_total"])
        if g.get("energy_uj") is not None:
            v[9] = float(g["energy_uj"])
    return v


def samples_to_matrix(samples: Sequence[Dict[str, Any]]) -> np.ndarray:
    """``[n, 16]`` snapshot rows (rank / seq columns left at 0) of a list of sampler readings,
    built column by column (no per-sample Python objects beyond the one pass that reads the dicts)."""
    n = len(samples)
    out = np.full((n, SNAPSHOT_F64), np.nan, dtype=np.float64)
    if not n:
        return out
    out[:, 0] = out[:, 1] = 0.0

    def col(key: str) -> np.ndarray:
        return np.array([np.nan if s.get(key) is None else s[key] for s in samples], dtype=np.float64)

    out[:, 2] = col("ts_s")
    for i, key in ((3, "watts"), (4, "gfx_util"), (5, "vram_used"), (7, "temp_c"), (8, "sclk_mhz")):
        out[:, i] = col(key)
    firsts = [(s.get("per_gpu") or [{}])[0] for s in samples]
    out[:, 6] = np.array([np.nan if g.get("vram_total") is None else g["vram_total"] for g in firsts], dtype=np.float64)
    out[:, 9] = np.array([np.nan if g.get("energy_uj") is None else g["energy_uj"] for g in firsts], dtype=np.float64)
    return out


class TelemetryGather:
    def __init__(self, group: Optional[Any] = None, device: Optional[torch.device] = None, native: bool = False, device_index: int = 0, impl: str = "rccl"):
        """``native=True`` routes the exchange through ``libkvmini_gather.so`` (C++: its own
        RCCL communicator and HIP stream; ``impl`` = ``rccl`` for ``ncclAllGather`` or ``p2p`` for
        the one-shot peer-store kernel) instead of ``torch.distributed``."""
        self.group = group
        self.native = None
        self.impl = impl
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None:
            backend = dist.get_backend(group) if dist.is_initialized() else "gloo"
            device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.device = device
        self.stream = torch.cuda.Stream(device, priority=0) if device.type == "cuda" else None
        self.seq = 0
        self.dropped_ticks = 0
        self.native_error: Optional[str] = None
        if native and self.world > 1:
            from kvmini.telemetry.native_rt import NativeGather

            # The native path is taken by every rank or by none: a rank that cannot set it up
            # (library missing, communicator not created) must not leave the others waiting in
            # a collective it never joins, so the ranks agree over the torch group first.
            try:
                self.native = NativeGather(device_index, impl=impl)
            except Exception as e:  # noqa: BLE001 - any failure means "use the torch path"
                self.native_error = f"{type(e).__name__}: {e}"
            if not self._all_ranks_ok(self.native is not None):
                self._drop_native()

    def _all_ranks_ok(self, ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _drop_native(self) -> None:
        if self.native is not None:
            try:
                self.native.close()
            except Exception:  # noqa: BLE001
                pass
            self.native = None
        if self.native_error is None:
            self.native_error = "another rank could not set up the native gather"

    @property
    def source(self) -> str:
        """What actually moved the snapshots (goes into power.json)."""
        if self.world == 1:
            return "amd-smi"
        if self.native is not None:
            return "amd-smi+" + ("xgmi-p2p-allgather" if self.impl == "p2p" else "rccl-allgather(ncclAllGather)")
        backend = dist.get_backend(self.group) if dist.is_initialized() else "none"
        return "amd-smi+" + ("rccl-allgather" if backend == "nccl" else f"{backend}-allgather")

    @classmethod
    def with_own_group(cls, device: Optional[torch.device] = None) -> "TelemetryGather":
        """Dedicated communicator so telemetry collectives never interleave with
        the model's (TP) collectives on the default group."""
        grp = dist.new_group() if dist.is_initialized() else None
        return cls(grp, device)

    # ------------------------------------------------------------------
    def _all_gather(self, local: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return local.unsqueeze(0)
        if self.native is not None:
            got = None
            try:
                got = self.native.gather(local.cpu().numpy().reshape(-1, SNAPSHOT_F64))
            except Exception as e:  # noqa: BLE001
                self.native_error = f"{type(e).__name__}: {e}"
            if self._all_ranks_ok(got is not None):
                return torch.from_numpy(got).reshape((self.world,) + tuple(local.shape))
            self._drop_native()  # some rank's native exchange failed: all ranks repeat it over torch.distributed
            local = local.to(self.device)
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=self.group)
            self.stream.synchronize()
        else:
            dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=self.group)
        return out

    def gather_tick(self, sample: Optional[Dict[str, Any]]) -> Optional[np.ndarray]:
        """All-gather this tick's snapshot.  Returns the ``[world, 16]`` table, or
        None when the ranks are not in lock-step (differing ``seq``)."""
        v = torch.from_numpy(snapshot_vector(self.rank, self.seq, sample)).to(self.device)
        self.seq += 1
        table = self._all_gather(v).cpu().numpy()
        if not np.all(table[:, 1] == table[0, 1]):
            self.dropped_ticks += 1
            return None
        return table

    def gather_table(self, samples: Sequence[Dict[str, Any]], t0: float, t1: float, dt: float = 0.1) -> np.ndarray:
        """Resample ``samples`` (this rank's time-ordered readings) onto the tick
        grid and exchange all ticks at once: returns ``[world, n_ticks, 16]``.
        ``t0``/``t1`` must be identical on every rank (broadcast them first)."""
        n_ticks = max(1, int(math.floor((t1 - t0) / dt)) + 1)
        local = np.full((n_ticks, SNAPSHOT_F64), np.nan, dtype=np.float64)
        local[:, 0] = self.rank
        local[:, 1] = np.arange(n_ticks)
        local[:, 2] = t0 + dt * np.arange(n_ticks)
        if samples:
            ts = np.array([s["ts_s"] for s in samples], dtype=np.float64)
            # nearest earlier-or-equal sample for each tick (sample-and-hold)
            idx = np.clip(np.searchsorted(ts, local[:, 2], side="right") - 1, 0, len(samples) - 1)
            local[:, 3:] = samples_to_matrix(samples)[idx, 3:]
        src = torch.from_numpy(local)
        table = self._all_gather(src if self.native is not None else src.to(self.device)).cpu().numpy()
        assert np.all(table[:, :, 1] == np.arange(n_ticks)[None, :]), "telemetry table out of lock-step"
        return table


def table_to_power_json(table: np.ndarray, dt: float, source: Optional[str] = None) -> Dict[str, Any]:
    """Node-level power.json from a gathered table: per tick, watts summed over
    ranks (NaN-aware) and per-GPU detail."""
    world, n_ticks, _ = table.shape
    samples: List[Dict[str, Any]] = []
    for k in range(n_ticks):
        per = []
        for r in range(world):
            row = table[r, k]
            per.append({"index": int(row[0]), "watts": None if np.isnan(row[3]) else float(row[3]), "gfx_util": None if np.isnan(row[4]) else float(row[4]), "vram_used": None if np.isnan(row[5]) else float(row[5])})
        w = [p["watts"] for p in per if p["watts"] is not None]
        u = [p["gfx_util"] for p in per if p["gfx_util"] is not None]
        m = [p["vram_used"] for p in per if p["vram_used"] is not None]
        samples.append({"ts_s": float(table[0, k, 2]), "watts": sum(w) if w else None, "gfx_util": sum(u) / len(u) if u else None, "vram_used": sum(m) if m else None, "per_gpu": per})
    if source is None:
        source = "amd-smi" if world == 1 else "amd-smi+rccl-allgather"
    return {"samples": samples, "interval_s": dt, "source": source, "world_size": int(world)}
