"""Repository-wide config validation (``kvmini validate --all`` / ``--yaml-tree``).

What the reference spreads over ``make validate-configs``, ``yaml-lint`` and the
``helm lint`` schema check (reference ``Makefile:136-139,402-420``), as one
in-process pass that needs no extra tools: every YAML parses, every benchmark
profile validates, the pricing / SLO / reference-matrix files have the keys
their consumers read, chart values (defaults and examples) satisfy
``values.schema.json``, chart templates have balanced ``{{ }}`` actions, and the
Grafana dashboards are valid JSON with a uid and at least one target per panel.
"""

from __future__ import annotations

import fnmatch
import json
import os
import re
from typing import Any, Dict, Iterable, List, Optional, Tuple

import yaml

TYPE_MAP = {"object": dict, "array": list, "string": str, "boolean": bool, "integer": int, "number": (int, float), "null": type(None)}


def validate_schema(inst: Any, schema: Dict[str, Any], root: Optional[Dict[str, Any]] = None, path: str = "$") -> List[str]:
    """Subset of JSON-Schema draft-07: type, required, properties, items, enum,
    pattern, minLength/maxLength, minimum/maximum, local ``$ref``."""
    root = root or schema
    errs: List[str] = []
    if "$ref" in schema:
        ref = schema["$ref"]
        if not ref.startswith("#/"):
            return [f"{path}: unsupported $ref {ref}"]
        node: Any = root
        for part in ref[2:].split("/"):
            node = node[part]
        return validate_schema(inst, node, root, path)
    t = schema.get("type")
    if t:
        types = t if isinstance(t, list) else [t]
        ok = False
        for tt in types:
            py = TYPE_MAP.get(tt, object)
            if isinstance(inst, py) and not (tt in ("integer", "number") and isinstance(inst, bool)):
                ok = True
        if not ok:
            return [f"{path}: expected {t}, got {type(inst).__name__}"]
    if "enum" in schema and inst not in schema["enum"]:
        errs.append(f"{path}: {inst!r} not in {schema['enum']}")
    if isinstance(inst, str):
        if "pattern" in schema and not re.search(schema["pattern"], inst):
            errs.append(f"{path}: {inst!r} does not match {schema['pattern']}")
        if len(inst) < schema.get("minLength", 0):
            errs.append(f"{path}: shorter than {schema['minLength']}")
        if "maxLength" in schema and len(inst) > schema["maxLength"]:
            errs.append(f"{path}: longer than {schema['maxLength']}")
    if isinstance(inst, (int, float)) and not isinstance(inst, bool):
        if "minimum" in schema and inst < schema["minimum"]:
            errs.append(f"{path}: below minimum {schema['minimum']}")
        if "maximum" in schema and inst > schema["maximum"]:
            errs.append(f"{path}: above maximum {schema['maximum']}")
    if isinstance(inst, dict):
        for k in schema.get("required", []):
            if k not in inst:
                errs.append(f"{path}: missing required key '{k}'")
        for k, sub in (schema.get("properties") or {}).items():
            if k in inst:
                errs += validate_schema(inst[k], sub, root, f"{path}.{k}")
    if isinstance(inst, list) and isinstance(schema.get("items"), dict):
        for i, v in enumerate(inst):
            errs += validate_schema(v, schema["items"], root, f"{path}[{i}]")
    return errs


def deep_merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in over.items():
        out[k] = deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def iter_files(root: str, patterns: Iterable[str], exclude: Iterable[str] = ()) -> Iterable[str]:
    skip_dirs = {".git", "build", "runs", "gpurun_out", "__pycache__", "node_modules", "artifacts", ".pytest_cache"}
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in skip_dirs]
        for f in files:
            p = os.path.relpath(os.path.join(d, f), root)
            if any(fnmatch.fnmatch(f, pat) for pat in patterns) and not any(fnmatch.fnmatch(p, ex) or fnmatch.fnmatch(os.path.dirname(p), ex) for ex in exclude):
                yield p


def check_yaml_tree(root: str, exclude: Iterable[str] = ("charts/*/templates", "charts/*/templates/*")) -> List[str]:
    errs = []
    for p in iter_files(root, ("*.yaml", "*.yml"), exclude):
        try:
            with open(os.path.join(root, p)) as f:
                list(yaml.safe_load_all(f))
        except yaml.YAMLError as e:
            errs.append(f"{p}: {str(e).splitlines()[0]}")
    return errs


def check_templates(root: str) -> List[str]:
    """Helm templates cannot be rendered without helm; at least the action
    delimiters and if/with/range/define ... end blocks must balance."""
    errs = []
    for p in iter_files(root, ("*.yaml", "*.tpl", "*.txt"), ()):
        if "/templates/" not in "/" + p.replace(os.sep, "/"):
            continue
        text = open(os.path.join(root, p)).read()
        if text.count("{{") != text.count("}}"):
            errs.append(f"{p}: unbalanced {{{{ }}}}")
        opens = len(re.findall(r"{{-?\s*(?:if|with|range|define|block)\b", text))
        ends = len(re.findall(r"{{-?\s*end\b", text))
        if opens != ends:
            errs.append(f"{p}: {opens} block openers vs {ends} end")
    return errs


def check_charts(root: str) -> List[str]:
    errs = []
    charts = os.path.join(root, "charts")
    if not os.path.isdir(charts):
        return errs
    for name in sorted(os.listdir(charts)):
        cdir = os.path.join(charts, name)
        if not os.path.isfile(os.path.join(cdir, "Chart.yaml")):
            continue
        chart = yaml.safe_load(open(os.path.join(cdir, "Chart.yaml")))
        for k in ("apiVersion", "name", "version"):
            if k not in chart:
                errs.append(f"charts/{name}/Chart.yaml: missing {k}")
        values = yaml.safe_load(open(os.path.join(cdir, "values.yaml"))) or {}
        schema_path = os.path.join(cdir, "values.schema.json")
        if os.path.exists(schema_path):
            schema = json.load(open(schema_path))
            errs += [f"charts/{name}/values.yaml {e}" for e in validate_schema(values, schema)]
            ex_dir = os.path.join(cdir, "examples")
            for ex in sorted(os.listdir(ex_dir)) if os.path.isdir(ex_dir) else []:
                over = yaml.safe_load(open(os.path.join(ex_dir, ex))) or {}
                errs += [f"charts/{name}/examples/{ex} {e}" for e in validate_schema(deep_merge(values, over), schema)]
    return errs


def check_dashboards(root: str) -> List[str]:
    errs = []
    ddir = os.path.join(root, "dashboards")
    for f in sorted(os.listdir(ddir)) if os.path.isdir(ddir) else []:
        if not f.endswith(".json"):
            continue
        try:
            d = json.load(open(os.path.join(ddir, f)))
        except ValueError as e:
            errs.append(f"dashboards/{f}: {e}")
            continue
        if not d.get("uid") or not d.get("title"):
            errs.append(f"dashboards/{f}: missing uid/title")
        for p in d.get("panels", []):
            if not p.get("targets"):
                errs.append(f"dashboards/{f}: panel '{p.get('title')}' has no targets")
            for t in p.get("targets", []):
                if "DCGM" in t.get("expr", "") or "nvidia" in t.get("expr", ""):
                    errs.append(f"dashboards/{f}: panel '{p.get('title')}' queries an NVIDIA-only metric")
    return errs


def check_configs(root: str) -> List[str]:
    from kvmini.config.profiles import list_profiles, load_profile
    from kvmini.cost.pricing import load_pricing
    from kvmini.gates.validate_config import ConfigValidator

    errs: List[str] = []
    cwd = os.getcwd()
    os.chdir(root)
    try:
        for group, names in list_profiles().items():
            for n in names:
                if group != "runners/profiles":
                    # partition / engine profiles are plain documents: they must parse and name themselves
                    doc = yaml.safe_load(open(n)) if os.path.exists(n) else None
                    if not isinstance(doc, dict) or not (doc.get("profile") or doc.get("name") or doc.get("metadata")):
                        errs.append(f"{n}: not a mapping with a profile/name key")
                    continue
                try:
                    prof = load_profile(n)
                except Exception as e:
                    errs.append(f"profile {n}: {e}")
                    continue
                v = ConfigValidator()
                if not v.validate_profile(dict(prof.raw)):
                    errs += [f"profile {n}: {e}" for e in v.errors]
        if os.path.exists("cost.yaml"):
            try:
                pr = load_pricing("cost.yaml")
                if pr.gpu_hourly("mi355x") <= 0:
                    errs.append("cost.yaml: no positive MI355X price")
            except Exception as e:
                errs.append(f"cost.yaml: {e}")
        if os.path.exists("slo.json"):
            slo = json.load(open("slo.json"))
            if not any(k in slo for k in ("p95_ms", "error_rate", "cost_per_1k_tokens")):
                errs.append("slo.json: none of p95_ms / error_rate / cost_per_1k_tokens present")
        if os.path.exists("reference-matrix.yaml"):
            m = yaml.safe_load(open("reference-matrix.yaml"))
            for k in ("gpus", "models", "traffic_patterns"):
                if not (m.get("matrix") or {}).get(k):
                    errs.append(f"reference-matrix.yaml: matrix.{k} is empty")
            for k in ("error_rate_max", "min_throughput_rps", "cold_multiplier_max", "p95_variance_pct"):
                if k not in (m.get("thresholds") or {}):
                    errs.append(f"reference-matrix.yaml: thresholds.{k} missing")
    finally:
        os.chdir(cwd)
    return errs


def run_all(root: str = ".") -> Tuple[int, Dict[str, List[str]]]:
    sections = {
        "yaml": check_yaml_tree(root),
        "helm templates": check_templates(root),
        "chart values vs schema": check_charts(root),
        "dashboards": check_dashboards(root),
        "configs": check_configs(root),
    }
    n = sum(len(v) for v in sections.values())
    return n, sections


def report(sections: Dict[str, List[str]]) -> None:
    for name, errs in sections.items():
        print(f"[{'ok' if not errs else 'FAIL'}] {name}" + (f" ({len(errs)} problems)" if errs else ""))
        for e in errs:
            print(f"    {e}")
