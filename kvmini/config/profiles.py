"""Typed benchmark profiles with one precedence rule: CLI > profile > default.

The reference loads a profile with an inline ``python3 -c`` inside ``bench.sh``
*after* parsing flags, so profile values silently override explicit flags, and
forwards every ``vllm_features`` entry to the load generator as ``--kebab-flags``,
which its strict argparse rejects (``bench.sh:143-184``, ``scripts/loadtest.py:678``).
Here a profile is split by destination: load-shape keys go to the load generator,
``vllm_features`` go to the *deployment* (vLLM args / env), request-side features
(``test_schema``, ``tools``, ``prompt_tokens``) go to the payload builder.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

from kvmini.utils import load_yaml, repo_root

LOAD_DEFAULTS: Dict[str, Any] = {"requests": 200, "concurrency": 10, "max_tokens": 64, "pattern": "steady"}


@dataclass
class Profile:
    name: str = "default"
    path: Optional[str] = None
    pattern: Optional[str] = None
    requests: Optional[int] = None
    concurrency: Optional[int] = None
    max_tokens: Optional[int] = None
    target_rps: Optional[float] = None
    stream: Optional[bool] = None
    prompt_tokens: Optional[int] = None
    description: str = ""
    vllm_features: Dict[str, Any] = field(default_factory=dict)
    native_engine: Dict[str, Any] = field(default_factory=dict)
    model_requirements: Dict[str, Any] = field(default_factory=dict)
    validation_hints: Dict[str, Any] = field(default_factory=dict)
    test_schema: Optional[Dict[str, Any]] = None
    tools: Optional[List[Dict[str, Any]]] = None
    raw: Dict[str, Any] = field(default_factory=dict)

    def openai_tools(self) -> Optional[List[Dict[str, Any]]]:
        """Profile tool definitions in OpenAI ``tools`` form."""
        if not self.tools:
            return None
        out = []
        for t in self.tools:
            if "function" in t:
                out.append(t)
            else:
                out.append({"type": "function", "function": {"name": t.get("name"), "description": t.get("description", ""), "parameters": t.get("parameters", {"type": "object", "properties": {}})}})
        return out

    def vllm_args(self) -> List[str]:
        """``vllm_features`` as vLLM server CLI args (booleans become bare flags,
        nulls are dropped).  ``device: cpu`` and friends pass through unchanged."""
        args: List[str] = []
        for k, v in self.vllm_features.items():
            flag = "--" + str(k).replace("_", "-")
            if v is None or v is False:
                continue
            if v is True:
                args.append(flag)
            else:
                args += [flag, str(v)]
        return args


def resolve_profile_path(name_or_path: str) -> Path:
    """Accept a path, or a bare profile name looked up under ``runners/profiles``
    (``fp8`` -> ``runners/profiles/quantization/fp8.yaml``)."""
    p = Path(name_or_path)
    if p.exists():
        return p
    root = repo_root() / "runners" / "profiles"
    for cand in (root / name_or_path, root / f"{name_or_path}.yaml", root / "quantization" / f"{name_or_path}.yaml", root / "partition" / f"{name_or_path}.yaml"):
        if cand.exists():
            return cand
    raise FileNotFoundError(f"profile not found: {name_or_path}")


def load_profile(name_or_path: Optional[str]) -> Profile:
    if not name_or_path:
        return Profile()
    path = resolve_profile_path(name_or_path)
    data = load_yaml(path) or {}
    if not isinstance(data, dict):
        raise ValueError(f"profile {path} is not a mapping")
    schema = data.get("test_schema")
    if isinstance(schema, str):
        try:
            schema = json.loads(schema)
        except ValueError:
            schema = None
    return Profile(
        name=path.stem,
        path=str(path),
        pattern=data.get("pattern"),
        requests=data.get("requests"),
        concurrency=data.get("concurrency"),
        max_tokens=data.get("max_tokens"),
        target_rps=data.get("target_rps"),
        stream=data.get("stream"),
        prompt_tokens=data.get("prompt_tokens"),
        description=data.get("description", ""),
        vllm_features=dict(data.get("vllm_features") or {}),
        native_engine=dict(data.get("native_engine") or {}),
        model_requirements=dict(data.get("model_requirements") or {}),
        validation_hints=dict(data.get("validation_hints") or {}),
        test_schema=schema if isinstance(schema, dict) else None,
        tools=data.get("tools"),
        raw=data,
    )


def resolve_load_settings(cli: Dict[str, Any], profile: Profile) -> Dict[str, Any]:
    """CLI value (when given, i.e. not None) > profile value > default."""
    out: Dict[str, Any] = {}
    for key, default in LOAD_DEFAULTS.items():
        cv = cli.get(key)
        pv = getattr(profile, key, None)
        out[key] = cv if cv is not None else pv if pv is not None else default
    for key in ("target_rps", "stream", "prompt_tokens"):
        cv = cli.get(key)
        out[key] = cv if cv is not None else getattr(profile, key, None)
    return out


def list_profiles() -> Dict[str, List[str]]:
    root = repo_root()
    groups = {
        "runners/profiles": sorted(str(p.relative_to(root)) for p in (root / "runners" / "profiles").rglob("*.yaml")),
        "profiles/partition": sorted(str(p.relative_to(root)) for p in (root / "profiles" / "partition").glob("*.yaml")),
        "profiles/engine": sorted(str(p.relative_to(root)) for p in (root / "profiles" / "engine").glob("*.yaml")),
    }
    return groups


def load_slo(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return json.load(f)
