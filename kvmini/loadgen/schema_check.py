"""Does a structured-output answer honour the JSON schema that was sent with the request?

The reference's ``structured-output`` profile promises "100 % format compliance"
(``runners/profiles/structured-output.yaml:26``) but nothing in it looks at the answers.  The load
generator checks every answer of a request that carried a schema: the text parses as JSON, and the
document satisfies the parts of JSON Schema that serving runtimes' guided decoding enforces —
``type`` (also a list of types), ``required``, ``properties`` (recursively), ``items``,
``minItems`` / ``maxItems``, ``enum``, ``const``, with ``$ref`` / ``allOf`` resolved and any
alternative of ``anyOf`` / ``oneOf`` accepted, and numeric bounds (``minimum`` / ``maximum`` and
the exclusive forms; the native engine writes unsigned integers, so a lower bound of 0 holds by
construction and an upper bound is up to the model).  String formats and patterns are not checked.
"""

from __future__ import annotations

import json
from typing import Any, Optional, Tuple

from ..utils.jsonschema_lite import alternatives, deref

_TYPES = {"object": dict, "array": list, "string": str, "boolean": bool, "null": type(None)}


def _is_type(v: Any, t: str) -> bool:
    if t == "integer":
        return isinstance(v, int) and not isinstance(v, bool) or (isinstance(v, float) and v.is_integer())
    if t == "number":
        return isinstance(v, (int, float)) and not isinstance(v, bool)
    py = _TYPES.get(t)
    return py is not None and isinstance(v, py)


def violation(doc: Any, schema: Any, path: str = "$", root: Any = None) -> Optional[str]:
    """First way ``doc`` breaks ``schema`` (None: it does not).  ``root`` is the document that
    ``$ref`` pointers refer to (the schema itself at the top)."""
    root = schema if root is None else root
    if not isinstance(schema, dict):
        return None
    schema = deref(schema, root)
    alts = alternatives(schema, root)
    if alts:
        whys = [violation(doc, a, path, root) for a in alts]
        if all(whys):
            return f"{path}: no alternative of anyOf / oneOf fits ({whys[0]})"
    if "const" in schema and doc != schema["const"]:
        return f"{path}: not the constant {schema['const']!r}"
    if isinstance(schema.get("enum"), list) and doc not in schema["enum"]:
        return f"{path}: {doc!r} is not one of {schema['enum']!r}"
    t = schema.get("type")
    if t is not None:
        types = [t] if isinstance(t, str) else list(t)
        if not any(_is_type(doc, x) for x in types):
            return f"{path}: {type(doc).__name__} where {' / '.join(types)} is required"
    if isinstance(doc, (int, float)) and not isinstance(doc, bool):
        for key, bad in (("minimum", lambda b: doc < b), ("maximum", lambda b: doc > b), ("exclusiveMinimum", lambda b: doc <= b), ("exclusiveMaximum", lambda b: doc >= b)):
            bound = schema.get(key)
            if isinstance(bound, (int, float)) and not isinstance(bound, bool) and bad(bound):
                return f"{path}: {doc!r} violates {key} {bound!r}"
    if isinstance(doc, dict):
        for k in schema.get("required") or []:
            if k not in doc:
                return f"{path}: required property {k!r} is missing"
        for k, sub in (schema.get("properties") or {}).items():
            if k in doc:
                why = violation(doc[k], sub, f"{path}.{k}", root)
                if why:
                    return why
    if isinstance(doc, list):
        lo, hi = schema.get("minItems"), schema.get("maxItems")
        if isinstance(lo, int) and len(doc) < lo:
            return f"{path}: {len(doc)} items where at least {lo} are required"
        if isinstance(hi, int) and len(doc) > hi:
            return f"{path}: {len(doc)} items where at most {hi} are allowed"
    if isinstance(doc, list) and isinstance(schema.get("items"), dict):
        for i, v in enumerate(doc):
            why = violation(v, schema["items"], f"{path}[{i}]", root)
            if why:
                return why
    return None


def check_document(text: str, schema: Any) -> Tuple[bool, bool, Optional[str]]:
    """(parses as JSON, satisfies the schema, what is wrong) for one answer."""
    try:
        doc = json.loads(text)
    except (TypeError, ValueError) as e:
        return False, False, f"not JSON: {e}"
    why = violation(doc, schema)
    return True, why is None, why
