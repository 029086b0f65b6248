"""GPU telemetry sampling through amd-smi (no DCGM, no Prometheus needed).

The reference polls ``sum(DCGM_FI_DEV_POWER_USAGE{...})`` from Prometheus every
>= 0.5 s (``energy/collector.py:203-251``).  On an MI355X node the counters are
read directly: socket power, gfx activity, VRAM use, temperature and clocks via
the ``amdsmi`` library, at 10-100 Hz from a background thread.  Fallback chain
when amd-smi is not usable (permissions, container without /dev/kfd):
``rocm-smi`` Python bindings are not assumed; we read
``/sys/class/drm/card*/device/hwmon/*/power1_average`` (microwatts), and a
``FakeSampler`` produces deterministic values for CPU-only tests.

A sample is the dict that also defines one row of ``power.json``::

    {"ts_s": wall seconds, "watts": W (sum over sampled GPUs), "gfx_util": %,
     "vram_used": bytes, "temp_c": C, "sclk_mhz": MHz, "per_gpu": [{...}, ...]}
"""

from __future__ import annotations

import glob
import math
import os
import threading
import time
from typing import Any, Callable, Dict, List, Optional, Sequence


class _AmdSmi:
    """Lazy singleton wrapper over the amdsmi python module."""

    _lock = threading.Lock()
    _mod: Any = None
    _handles: List[Any] = []
    _err: Optional[str] = None

    @classmethod
    def init(cls) -> bool:
        with cls._lock:
            if cls._mod is not None:
                return True
            if cls._err is not None:
                return False
            try:
                import amdsmi  # type: ignore

                amdsmi.amdsmi_init()
                cls._handles = list(amdsmi.amdsmi_get_processor_handles())
                cls._mod = amdsmi
                return True
            except BaseException as e:  # library missing, no devices, no permission
                cls._err = f"{type(e).__name__}: {e}"
                return False


def _num(v: Any) -> Optional[float]:
    try:
        f = float(v)
        return None if math.isnan(f) else f
    except (TypeError, ValueError):
        return None


def gpu_inventory() -> List[Dict[str, Any]]:
    """Static facts per GPU: product name, VRAM bytes, compute/memory partition."""
    out: List[Dict[str, Any]] = []
    if _AmdSmi.init():
        smi = _AmdSmi._mod
        for i, h in enumerate(_AmdSmi._handles):
            d: Dict[str, Any] = {"index": i}
            try:
                asic = smi.amdsmi_get_gpu_asic_info(h)
                d["product"] = asic.get("market_name") or asic.get("asic_name")
                d["target"] = asic.get("target_graphics_version")
                d["num_cu"] = asic.get("num_compute_units")
            except Exception:
                pass
            try:
                vr = smi.amdsmi_get_gpu_vram_usage(h)
                d["vram_total_bytes"] = int(vr.get("vram_total", 0)) * (1 << 20)
            except Exception:
                pass
            for key, fn in (("compute_partition", "amdsmi_get_gpu_compute_partition"), ("memory_partition", "amdsmi_get_gpu_memory_partition")):
                try:
                    d[key] = str(getattr(smi, fn)(h))
                except Exception:
                    pass
            out.append(d)
        return out
    try:
        import torch

        if torch.cuda.is_available():
            for i in range(torch.cuda.device_count()):
                p = torch.cuda.get_device_properties(i)
                out.append({"index": i, "product": p.name, "vram_total_bytes": int(p.total_memory), "num_cu": getattr(p, "multi_processor_count", None)})
    except Exception:
        pass
    return out


def min_gpu_memory_gb() -> Optional[float]:
    """Smallest VRAM across visible GPUs in GB (config validator's autodetect;
    the reference shells out to ``nvidia-smi`` for this, ``validate_config.py:215-238``)."""
    sizes = [g["vram_total_bytes"] / 1e9 for g in gpu_inventory() if g.get("vram_total_bytes")]
    return min(sizes) if sizes else None


def read_once(gpu_indices: Optional[Sequence[int]] = None) -> Optional[Dict[str, Any]]:
    """One telemetry reading (None when no source works)."""
    ts = time.time()
    if _AmdSmi.init():
        smi = _AmdSmi._mod
        per: List[Dict[str, Any]] = []
        for i, h in enumerate(_AmdSmi._handles):
            if gpu_indices is not None and i not in gpu_indices:
                continue
            g: Dict[str, Any] = {"index": i}
            try:
                pw = smi.amdsmi_get_power_info(h)
                w = _num(pw.get("current_socket_power"))
                if w is None or w <= 0:
                    w = _num(pw.get("average_socket_power"))
                g["watts"] = w
            except Exception:
                g["watts"] = None
            try:
                act = smi.amdsmi_get_gpu_activity(h)
                g["gfx_util"] = _num(act.get("gfx_activity"))
                g["umc_util"] = _num(act.get("umc_activity"))
            except Exception:
                pass
            try:
                vr = smi.amdsmi_get_gpu_vram_usage(h)
                g["vram_used"] = _num(vr.get("vram_used")) and float(vr["vram_used"]) * (1 << 20)
                g["vram_total"] = _num(vr.get("vram_total")) and float(vr["vram_total"]) * (1 << 20)
            except Exception:
                pass
            try:
                g["temp_c"] = _num(smi.amdsmi_get_temp_metric(h, smi.AmdSmiTemperatureType.HOTSPOT, smi.AmdSmiTemperatureMetric.CURRENT))
            except Exception:
                pass
            try:
                ck = smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)
                g["sclk_mhz"] = _num(ck.get("clk") if isinstance(ck, dict) else None)
            except Exception:
                pass
            try:
                en = smi.amdsmi_get_energy_count(h)
                # hardware accumulator: counter * resolution (uJ)
                cnt = _num(en.get("energy_accumulator") or en.get("power"))
                res = _num(en.get("counter_resolution"))
                if cnt is not None and res is not None:
                    g["energy_uj"] = cnt * res
            except Exception:
                pass
            per.append(g)
        if per:
            return _aggregate(ts, per)
    # hwmon fallback
    per = []
    for i, path in enumerate(sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))):
        if gpu_indices is not None and i not in gpu_indices:
            continue
        try:
            with open(path) as f:
                per.append({"index": i, "watts": float(f.read().strip()) / 1e6})
        except (OSError, ValueError):
            continue
    if per:
        return _aggregate(ts, per)
    return None


def _aggregate(ts: float, per: List[Dict[str, Any]]) -> Dict[str, Any]:
    def s(key: str, how: str) -> Optional[float]:
        vals = [g[key] for g in per if g.get(key) is not None]
        if not vals:
            return None
        return sum(vals) if how == "sum" else sum(vals) / len(vals)

    return {
        "ts_s": ts,
        "watts": s("watts", "sum"),
        "gfx_util": s("gfx_util", "avg"),
        "vram_used": s("vram_used", "sum"),
        "temp_c": s("temp_c", "avg"),
        "sclk_mhz": s("sclk_mhz", "avg"),
        "per_gpu": per,
    }


class FakeSampler:
    """Deterministic telemetry source for tests: ``watts = base + amp*sin``."""

    def __init__(self, base_w: float = 500.0, amp_w: float = 50.0, n_gpus: int = 1):
        self.base_w, self.amp_w, self.n_gpus = base_w, amp_w, n_gpus

    def __call__(self, gpu_indices: Optional[Sequence[int]] = None) -> Dict[str, Any]:
        ts = time.time()
        per = [{"index": i, "watts": self.base_w + self.amp_w * math.sin(ts + i), "gfx_util": 80.0, "vram_used": 1e9, "temp_c": 60.0, "sclk_mhz": 2000.0} for i in range(self.n_gpus)]
        return _aggregate(ts, per)


class TelemetrySampler:
    """Background sampling thread: ``start()`` ... ``stop()`` -> list of samples."""

    def __init__(self, interval_s: float = 0.1, gpu_indices: Optional[Sequence[int]] = None, source: Optional[Callable[..., Optional[Dict[str, Any]]]] = None, max_samples: int = 1_000_000):
        self.interval_s = max(0.005, float(interval_s))
        self.gpu_indices = gpu_indices
        self.source = source or read_once
        self.samples: List[Dict[str, Any]] = []
        self.max_samples = max_samples
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None

    def _run(self) -> None:
        nxt = time.perf_counter()
        while not self._stop.is_set():
            s = self.source(self.gpu_indices)
            if s is not None and len(self.samples) < self.max_samples:
                self.samples.append(s)
            nxt += self.interval_s
            delay = nxt - time.perf_counter()
            if delay > 0:
                self._stop.wait(delay)
            else:
                nxt = time.perf_counter()

    def start(self) -> "TelemetrySampler":
        self._stop.clear()
        self._th = threading.Thread(target=self._run, daemon=True, name="kvmini-telemetry")
        self._th.start()
        return self

    def stop(self) -> List[Dict[str, Any]]:
        self._stop.set()
        if self._th is not None:
            self._th.join(5)
            self._th = None
        return self.samples

    def working(self) -> bool:
        return self.source(self.gpu_indices) is not None

    def to_power_json(self) -> Dict[str, Any]:
        return {"samples": self.samples, "interval_s": self.interval_s, "source": "amd-smi" if self.source is read_once else getattr(self.source, "__class__", type(self.source)).__name__}
