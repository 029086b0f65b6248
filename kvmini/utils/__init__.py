"""Small shared helpers: clocks, JSON/CSV IO, Kubernetes quantity parsing, redaction."""

from .common import (  # noqa: F401
    dns1123_label,
    load_json,
    load_yaml,
    merge_json_file,
    now_ms,
    now_s,
    parse_k8s_quantity,
    redact_cmd,
    repo_root,
    safe_float,
    stable_hash,
    write_json,
)
