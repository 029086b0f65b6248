"""The part of JSON Schema's composition keywords that schemas written by tools use: ``$ref`` into
``$defs`` / ``definitions`` (pydantic's nested models), ``allOf`` (merged), ``anyOf`` / ``oneOf``
(pydantic's ``Optional[X]`` is ``anyOf: [X, {"type": "null"}]``).  Shared by the guided-decoding
script builder, which has to pick ONE shape to write, and the answer validator, which accepts any
alternative."""

from __future__ import annotations

from typing import Any, Dict, List

_MAX_HOPS = 32  # references followed for one node (a cycle of references ends here)


def deref(spec: Any, root: Any) -> Dict[str, Any]:
    """``spec`` with its ``$ref`` chain followed (local references only: ``#/...``) and its
    ``allOf`` members merged in; keywords next to a ``$ref`` win over the referenced schema's.
    Always a dict (anything else constrains nothing)."""
    hops = 0
    while isinstance(spec, dict) and isinstance(spec.get("$ref"), str) and hops < _MAX_HOPS:
        hops += 1
        ref = spec["$ref"]
        target: Any = root if ref.startswith("#") else None
        for part in [p for p in ref[1:].split("/") if p] if target is not None else []:
            part = part.replace("~1", "/").replace("~0", "~")
            target = target.get(part) if isinstance(target, dict) else None
        rest = {k: v for k, v in spec.items() if k != "$ref"}
        spec = {**target, **rest} if isinstance(target, dict) else rest
    if not isinstance(spec, dict):
        return {}
    if isinstance(spec.get("allOf"), list):
        merged: Dict[str, Any] = {k: v for k, v in spec.items() if k != "allOf"}
        for member in spec["allOf"]:
            m = deref(member, root)
            for k, v in m.items():
                if k == "properties" and isinstance(v, dict):
                    merged["properties"] = {**v, **(merged.get("properties") or {})}
                elif k == "required" and isinstance(v, list):
                    merged["required"] = list(dict.fromkeys(list(merged.get("required") or []) + v))
                else:
                    merged.setdefault(k, v)
        spec = merged
    return spec


def alternatives(spec: Dict[str, Any], root: Any) -> List[Dict[str, Any]]:
    """The dereferenced members of ``anyOf`` / ``oneOf`` ([] when the node has neither)."""
    alts = spec.get("anyOf") if isinstance(spec.get("anyOf"), list) else spec.get("oneOf")
    return [deref(a, root) for a in alts] if isinstance(alts, list) else []


def preferred(spec: Any, root: Any) -> Dict[str, Any]:
    """The one shape a writer produces for this node: the node itself, or — for ``anyOf`` /
    ``oneOf`` — its first alternative that is not just ``null`` (``Optional[X]`` is written as an
    X), with the node's own keywords kept beside it."""
    node = deref(spec, root)
    for _ in range(_MAX_HOPS):
        alts = alternatives(node, root)
        if not alts:
            return node
        pick = next((a for a in alts if a.get("type") != "null"), alts[0])
        node = {**{k: v for k, v in node.items() if k not in ("anyOf", "oneOf")}, **pick}
    return node
