"""Shared helpers used across the harness.

Behavioural references (reference repo, read for semantics only):
``cost_estimator.py:48-83`` (Kubernetes quantity grammar),
``scripts/trtllm_build_vs_perf.py:34-54`` (secret redaction, DNS-1123 labels),
``analyze.py:606-617`` (results.json merge-update contract).
"""

from __future__ import annotations

import hashlib
import json
import math
import os
import re
import time
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional


def now_ms() -> float:
    """Wall-clock milliseconds (float). Request timestamps are wall clock so
    that they can be intersected with pod start times and power samples."""
    return time.time() * 1000.0


def now_s() -> float:
    return time.time()


def repo_root() -> Path:
    """Directory that holds bench.sh / cost.yaml / runners/ (the checkout root)."""
    return Path(__file__).resolve().parents[2]


def safe_float(v: Any, default: Optional[float] = None) -> Optional[float]:
    """float(v) that maps '', None, 'NaN' and garbage to ``default``."""
    if v is None:
        return default
    if isinstance(v, (int, float)):
        f = float(v)
        return default if math.isnan(f) else f
    s = str(v).strip()
    if s == "" or s.lower() in ("nan", "none", "null"):
        return default
    try:
        f = float(s)
    except ValueError:
        return default
    return default if math.isnan(f) else f


def load_json(path: os.PathLike | str, default: Any = None) -> Any:
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        if default is not None:
            return default
        raise


def _json_default(o: Any) -> Any:
    # numpy scalars / arrays and Paths sneak into results dicts
    if hasattr(o, "tolist"):
        return o.tolist()
    if hasattr(o, "item"):
        return o.item()
    if isinstance(o, Path):
        return str(o)
    raise TypeError(f"not JSON serialisable: {type(o).__name__}")


def _nan_to_none(obj: Any) -> Any:
    """results.json must be strict JSON: NaN/inf become null."""
    if isinstance(obj, float):
        return None if (math.isnan(obj) or math.isinf(obj)) else obj
    if isinstance(obj, dict):
        return {k: _nan_to_none(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_nan_to_none(v) for v in obj]
    return obj


def write_json(path: os.PathLike | str, obj: Any, indent: int = 2) -> None:
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    tmp = p.with_suffix(p.suffix + ".tmp")
    with open(tmp, "w") as f:
        json.dump(_nan_to_none(json.loads(json.dumps(obj, default=_json_default))), f, indent=indent)
        f.write("\n")
    os.replace(tmp, p)


def merge_json_file(path: os.PathLike | str, updates: Dict[str, Any]) -> Dict[str, Any]:
    """Merge ``updates`` into the JSON object stored at ``path`` (created if
    missing) and return the merged dict.  This is the blackboard pattern every
    pipeline stage uses on results.json."""
    merged: Dict[str, Any] = {}
    if os.path.exists(path):
        try:
            prev = load_json(path)
            if isinstance(prev, dict):
                merged.update(prev)
        except (OSError, ValueError):
            pass
    merged.update(updates)
    write_json(path, merged)
    return merged


def load_yaml(path: os.PathLike | str) -> Any:
    import yaml

    with open(path) as f:
        return yaml.safe_load(f)


_BIN_SUFFIX = {"Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "Pi": 2**50, "Ei": 2**60}
_DEC_SUFFIX = {"n": 1e-9, "u": 1e-6, "m": 1e-3, "k": 1e3, "K": 1e3, "M": 1e6, "G": 1e9, "T": 1e12, "P": 1e15, "E": 1e18}


def parse_k8s_quantity(q: Any) -> float:
    """Parse a Kubernetes resource quantity ("500m", "12Gi", "2", "1.5G", "1e3")
    into a plain float (cores, bytes, or count).  Unparseable input -> 0.0."""
    if q is None:
        return 0.0
    if isinstance(q, (int, float)):
        return float(q)
    s = str(q).strip()
    if not s:
        return 0.0
    for suf, mul in _BIN_SUFFIX.items():
        if s.endswith(suf):
            try:
                return float(s[: -len(suf)]) * mul
            except ValueError:
                return 0.0
    if s[-1] in _DEC_SUFFIX and not s[-1].isdigit():
        # avoid eating the exponent marker of "1e3" (handled by float() below)
        try:
            return float(s[:-1]) * _DEC_SUFFIX[s[-1]]
        except ValueError:
            return 0.0
    try:
        return float(s)
    except ValueError:
        return 0.0


_SECRET_FLAG = re.compile(r"(?i)(token|secret|password|passwd|api[-_]?key|auth|credential)")


def redact_cmd(argv: Iterable[str]) -> List[str]:
    """Mask values of secret-looking flags/assignments in a command line that is
    about to be logged (``--api-key X`` -> ``--api-key ***``, ``FOO_TOKEN=abc`` ->
    ``FOO_TOKEN=***``)."""
    out: List[str] = []
    mask_next = False
    for a in argv:
        if mask_next:
            out.append("***")
            mask_next = False
            continue
        if "=" in a:
            k, _, _v = a.partition("=")
            if _SECRET_FLAG.search(k):
                out.append(f"{k}=***")
                continue
        elif a.startswith("-") and _SECRET_FLAG.search(a):
            mask_next = True
        out.append(a)
    return out


def dns1123_label(name: str, max_len: int = 63) -> str:
    """Lower-case RFC-1123 label usable as a Kubernetes object name."""
    s = re.sub(r"[^a-z0-9-]+", "-", name.lower()).strip("-")
    s = re.sub(r"-{2,}", "-", s)
    if not s:
        s = "x"
    return s[:max_len].rstrip("-") or "x"


def stable_hash(obj: Any, n: int = 8) -> str:
    """Deterministic short id of a JSON-able object (process-independent,
    unlike the salted builtin ``hash``)."""
    blob = json.dumps(obj, sort_keys=True, default=str).encode()
    return hashlib.sha1(blob).hexdigest()[:n]
