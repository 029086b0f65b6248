"""ctypes bindings for ``libkvmini_rt.so`` — the native runtime pieces:

* :class:`NativeSampler` — a C++ thread polling the AMD SMI library into a ring
  buffer (``csrc/amdsmi_sampler.cpp``); same interface as
  :class:`kvmini.telemetry.amdsmi_sampler.TelemetrySampler`, but the cadence does
  not depend on the GIL.
* :class:`NativeGather` — one ``ncclAllGather`` (RCCL) of 128-byte snapshots on a
  private communicator and stream (``csrc/telemetry_gather.cpp``).

The reference samples power by shelling out to ``nvidia-smi`` once per second
(``energy/collector.py:103-145``) and aggregates multi-GPU telemetry through
Prometheus scrapes (``analyze.py:259-299``); both are replaced by these.
"""

from __future__ import annotations

import ctypes
import math
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

_LIB: Optional[ctypes.CDLL] = None
_TRIED = False


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libkvmini_rt.so")


def load() -> Optional[ctypes.CDLL]:
    """The sampler library, or None when it is not built.  It links no GPU runtime (amd-smi is
    opened with dlopen inside), so loading it never brings a second HIP runtime into a
    process that also imports torch."""
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    try:
        lib = ctypes.CDLL(lib_path())
    except OSError:
        return None
    dp = ctypes.POINTER(ctypes.c_double)
    lib.kvs_gpu_count.restype = ctypes.c_int
    lib.kvs_start.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_longlong]
    lib.kvs_start.restype = ctypes.c_int
    lib.kvs_start_synthetic.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_longlong]
    lib.kvs_start_synthetic.restype = ctypes.c_int
    lib.kvs_count.restype = ctypes.c_longlong
    lib.kvs_read.argtypes = [dp, ctypes.c_longlong]
    lib.kvs_read.restype = ctypes.c_longlong
    lib.kvs_stop.restype = ctypes.c_int
    lib.kvs_trapz_wh.argtypes = [dp, dp, ctypes.c_longlong]
    lib.kvs_trapz_wh.restype = ctypes.c_double
    _LIB = lib
    return lib


_GATHER: Optional[ctypes.CDLL] = None


def gather_lib_path() -> str:
    return os.path.join(os.path.dirname(lib_path()), "libkvmini_gather.so")


def _gpu_runtime_global() -> None:
    """``libkvmini_gather.so`` is linked without HIP / RCCL; make the process's copies of
    them global first.  With torch imported those are the ones bundled with torch (a second,
    system HIP runtime in the same process breaks kernel launches); otherwise the system's."""
    import sys

    cands: List[List[str]] = []
    torch = sys.modules.get("torch")
    if torch is not None:
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        cands = [[os.path.join(d, "libamdhip64.so")], [os.path.join(d, "librccl.so")]]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    sysl = [[os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"], [os.path.join(rocm, "lib", "librccl.so"), "librccl.so"]]
    for i in range(2):
        for name in (cands[i] if cands else []) + sysl[i]:
            try:
                ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                break
            except OSError:
                continue


def load_gather() -> Optional[ctypes.CDLL]:
    """The RCCL telemetry-gather library bound to this process's HIP runtime and RCCL."""
    global _GATHER
    if _GATHER is not None:
        return _GATHER
    _gpu_runtime_global()
    try:
        lib = ctypes.CDLL(gather_lib_path())
    except OSError:
        return None
    dp = ctypes.POINTER(ctypes.c_double)
    lib.kvt_unique_id_bytes.restype = ctypes.c_int
    lib.kvt_unique_id.argtypes = [ctypes.c_void_p]
    lib.kvt_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.kvt_gather_many.argtypes = [dp, dp, ctypes.c_longlong]
    lib.kvt_gather.argtypes = [dp, dp]
    _GATHER = lib
    return lib


def _dp(a: np.ndarray) -> Any:
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def trapz_wh(t_s: Sequence[float], watts: Sequence[float]) -> float:
    """Trapezoidal energy through the native runtime (NaN samples skipped)."""
    lib = load()
    t = np.ascontiguousarray(t_s, dtype=np.float64)
    w = np.ascontiguousarray(watts, dtype=np.float64)
    if lib is None:
        ok = ~(np.isnan(t) | np.isnan(w))
        return float(np.trapezoid(w[ok], t[ok]) / 3600.0) if ok.sum() > 1 else 0.0
    return float(lib.kvs_trapz_wh(_dp(t), _dp(w), len(t)))


def native_gpu_count() -> int:
    lib = load()
    if lib is None:
        return 0
    return max(0, int(lib.kvs_gpu_count()))


class NativeSampler:
    """Drop-in for ``TelemetrySampler`` backed by the C++ sampling thread."""

    RECORD = 8  # ts_s, gpu, watts, gfx_util, vram_used_bytes, temp_c, sclk_mhz, energy_uj

    def __init__(self, interval_s: float = 0.05, gpu_indices: Optional[Sequence[int]] = None, capacity: int = 1 << 20, synthetic: int = 0):
        """``synthetic`` > 0: that many fake GPUs fed from a counter instead of the SMI library
        (the same thread and ring; for dry runs and tests on machines without an AMD GPU)."""
        self.synthetic = int(synthetic)
        self.interval_s = max(0.001, float(interval_s))
        self.gpu_indices = list(gpu_indices) if gpu_indices is not None else None
        self.capacity = capacity
        self.samples: List[Dict[str, Any]] = []
        self._lib = load()
        self._started = False

    def working(self) -> bool:
        return self._lib is not None and (self.synthetic > 0 or native_gpu_count() > 0)

    def start(self) -> "NativeSampler":
        if not self.working():
            raise RuntimeError("native telemetry runtime unavailable (library missing or no AMD GPU visible)")
        idx = self.gpu_indices or []
        arr = (ctypes.c_int * max(1, len(idx)))(*idx)
        if self.synthetic > 0:
            rc = self._lib.kvs_start_synthetic(self.interval_s, self.synthetic, self.capacity)
        else:
            rc = self._lib.kvs_start(self.interval_s, arr, len(idx), self.capacity)
        if rc != 0:
            raise RuntimeError(f"kvs_start failed: {rc}")
        self._started = True
        return self

    def _drain(self) -> List[Dict[str, Any]]:
        n = int(self._lib.kvs_count())
        n = min(n, self.capacity)
        if n <= 0:
            return []
        buf = np.empty((n, self.RECORD), dtype=np.float64)
        got = int(self._lib.kvs_read(_dp(buf), n))
        return records_to_samples(buf[:got])

    def latest(self) -> Optional[Dict[str, Any]]:
        s = self._drain()
        return s[-1] if s else None

    def stop(self) -> List[Dict[str, Any]]:
        if self._started:
            self.samples = self._drain()
            self._lib.kvs_stop()
            self._started = False
        return self.samples

    def to_power_json(self) -> Dict[str, Any]:
        return {"samples": self.samples, "interval_s": self.interval_s, "source": "synthetic (native sampler thread)" if self.synthetic else "amd-smi (native sampler thread)"}


def records_to_samples(rec: np.ndarray) -> List[Dict[str, Any]]:
    """Group per-(tick, gpu) records into per-tick samples in the collector's schema
    (watts summed over GPUs, utilisation averaged).  A tick = a run of records with
    distinct GPU ids."""
    out: List[Dict[str, Any]] = []
    cur: List[np.ndarray] = []
    seen: set = set()

    def flush() -> None:
        if not cur:
            return

        def agg(col: int, how: str) -> Optional[float]:
            vals = [float(r[col]) for r in cur if not math.isnan(r[col])]
            if not vals:
                return None
            return sum(vals) if how == "sum" else sum(vals) / len(vals)

        per = [{"index": int(r[1]), "watts": None if math.isnan(r[2]) else float(r[2]), "gfx_util": None if math.isnan(r[3]) else float(r[3]), "vram_used": None if math.isnan(r[4]) else float(r[4]),
                "temp_c": None if math.isnan(r[5]) else float(r[5]), "sclk_mhz": None if math.isnan(r[6]) else float(r[6]), "energy_uj": None if math.isnan(r[7]) else float(r[7])} for r in cur]
        out.append({"ts_s": float(cur[0][0]), "watts": agg(2, "sum"), "gfx_util": agg(3, "avg"), "vram_used": agg(4, "sum"), "temp_c": agg(5, "avg"), "sclk_mhz": agg(6, "avg"), "per_gpu": per})

    for r in rec:
        gid = int(r[1])
        if gid in seen:
            flush()
            cur, seen = [], set()
        cur.append(r)
        seen.add(gid)
    flush()
    return out


class NativeGather:
    """All-gather of telemetry snapshots through native code, two implementations:

    * ``impl="rccl"`` — one ``ncclAllGather`` on a private communicator and stream
      (``csrc/telemetry_gather.cpp``); the unique id travels over the already-initialised
      ``torch.distributed`` group.  A ring: W - 1 latency-bound hops for W x 128 bytes.
    * ``impl="p2p"`` — the one-shot peer-memory kernel (``csrc/xgmi_collectives.hip``): every
      rank stores its snapshots straight into every peer's buffer over its own xGMI link, one
      flag per workgroup, one hop.
    """

    def __init__(self, device_index: int, impl: str = "rccl"):
        import torch
        import torch.distributed as dist

        lib = load_gather()
        if lib is None:
            raise RuntimeError("libkvmini_gather.so not available")
        if impl not in ("rccl", "p2p"):
            raise ValueError(f"gather impl {impl!r}: rccl or p2p")
        self._lib = lib
        self.impl = impl
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._xgmi = None
        if impl == "p2p":
            from kvmini.parallel.xgmi import XgmiComm

            self._dev = torch.device("cuda", device_index)
            self._xgmi = XgmiComm(group=dist.group.WORLD if dist.is_initialized() else None, device=self._dev)
            return
        nbytes = int(lib.kvt_unique_id_bytes())
        buf = ctypes.create_string_buffer(nbytes)
        id_ok = True
        if self.rank == 0 and lib.kvt_unique_id(buf) != 0:
            id_ok = False  # still broadcast (zeros): the other ranks are waiting in the collective
            buf = ctypes.create_string_buffer(nbytes)
        if self.world > 1:
            t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if dist.get_backend() == "nccl":
                t = t.cuda(device_index)
            dist.broadcast(t, 0)
            raw = bytes(t.cpu().numpy().tobytes())
            id_ok = any(raw)  # an all-zero id = rank 0 could not create one: every rank sees that
            buf = ctypes.create_string_buffer(raw, nbytes)
        if not id_ok:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        rc = lib.kvt_init(self.rank, self.world, buf, device_index)
        if rc != 0:
            raise RuntimeError(f"kvt_init failed: {rc}")

    def gather(self, local: np.ndarray) -> np.ndarray:
        """``local``: ``[n_ticks, 16]`` float64 -> ``[world, n_ticks, 16]``."""
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, 16)
        n = local.shape[0]
        if self._xgmi is not None:
            import torch

            per = max(1, self._xgmi.slot_bytes // 128)  # ticks per call: a snapshot is 128 bytes
            parts = [self._xgmi.all_gather(torch.from_numpy(local[i : i + per]).to(self._dev)).cpu().numpy() for i in range(0, n, per)]
            return np.concatenate(parts, axis=1)
        table = np.empty((self.world, n, 16), dtype=np.float64)
        rc = self._lib.kvt_gather_many(_dp(local), _dp(table), n)
        if rc != 0:
            raise RuntimeError(f"kvt_gather_many failed: {rc}")
        return table

    def close(self) -> None:
        if self._xgmi is not None:
            self._xgmi.close()
            self._xgmi = None
            return
        self._lib.kvt_destroy()


def make_sampler(interval_s: float = 0.05, gpu_indices: Optional[Sequence[int]] = None) -> Any:
    """Native sampler when the runtime and a GPU are available, else the Python one."""
    ns = NativeSampler(interval_s, gpu_indices)
    if ns.working():
        return ns
    from .amdsmi_sampler import TelemetrySampler

    return TelemetrySampler(interval_s=interval_s, gpu_indices=gpu_indices)
