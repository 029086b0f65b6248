"""Wire-protocol plug-ins for the load engine: OpenAI chat, TGI, KServe-v2 generate.

The reference has one real load generator (OpenAI chat,
``scripts/loadtest.py:260-342``) and two closed-loop heredoc scripts for TGI and
Triton that emit a different 4-column CSV (``runners/backends/tgi/invoke.sh:81-217``,
``runners/backends/triton/invoke.sh:86-245``).  Here a protocol only decides the
path, the payload, the SSE dialect and how to read a non-streaming body; arrival
scheduling, timing, tracing and the canonical ``requests.csv`` are shared.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple


@dataclass
class RequestSpec:
    """Everything request-shaped that the CLI / a profile can set."""

    model: str = "placeholder"
    prompt: str = "Hello, world!"
    max_tokens: int = 64
    stream: bool = True
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    n: Optional[int] = None
    presence_penalty: Optional[float] = None
    frequency_penalty: Optional[float] = None
    json_mode: bool = False
    include_usage: bool = True
    extra_payload: Dict[str, Any] = field(default_factory=dict)
    tools: Optional[List[Dict[str, Any]]] = None
    response_schema: Optional[Dict[str, Any]] = None


class Protocol:
    name = "abstract"
    sse_dialect = "openai"

    def path(self, spec: RequestSpec) -> str:
        raise NotImplementedError

    def payload(self, spec: RequestSpec, prompt: Optional[str] = None) -> Dict[str, Any]:
        raise NotImplementedError

    def parse_body(self, body: Any) -> Tuple[Optional[Dict[str, int]], str]:
        """(usage, text) from a non-streaming JSON body."""
        raise NotImplementedError

    def health_paths(self) -> List[str]:
        return ["/v1/models", "/health", "/"]


class OpenAIChat(Protocol):
    name = "openai"
    sse_dialect = "openai"

    def path(self, spec: RequestSpec) -> str:
        return "/v1/chat/completions"

    def payload(self, spec: RequestSpec, prompt: Optional[str] = None) -> Dict[str, Any]:
        p: Dict[str, Any] = {
            "model": spec.model,
            "messages": [{"role": "user", "content": spec.prompt if prompt is None else prompt}],
            "max_tokens": spec.max_tokens,
            "temperature": 0 if spec.temperature is None else spec.temperature,
            "stream": spec.stream,
        }
        if spec.top_p is not None:
            p["top_p"] = spec.top_p
        if spec.top_k is not None:
            p["top_k"] = spec.top_k
        if spec.n is not None and spec.n > 1:
            p["n"] = spec.n
        if spec.presence_penalty is not None:
            p["presence_penalty"] = spec.presence_penalty
        if spec.frequency_penalty is not None:
            p["frequency_penalty"] = spec.frequency_penalty
        if spec.response_schema is not None:
            p["response_format"] = {"type": "json_schema", "json_schema": {"name": "response", "schema": spec.response_schema}}
        elif spec.json_mode:
            p["response_format"] = {"type": "json_object"}
        if spec.tools:
            p["tools"] = spec.tools
            p.setdefault("tool_choice", "auto")
        if spec.stream and spec.include_usage:
            p["stream_options"] = {"include_usage": True}
        if spec.extra_payload:
            p.update(spec.extra_payload)
        return p

    def parse_body(self, body: Any) -> Tuple[Optional[Dict[str, int]], str]:
        if not isinstance(body, dict):
            return None, ""
        text = ""
        for ch in body.get("choices") or []:
            msg = ch.get("message") or {}
            text += msg.get("content") or ch.get("text") or ""
        usage = body.get("usage") if isinstance(body.get("usage"), dict) else None
        return usage, text


class OpenAICompletions(OpenAIChat):
    name = "openai-completions"

    def path(self, spec: RequestSpec) -> str:
        return "/v1/completions"

    def payload(self, spec: RequestSpec, prompt: Optional[str] = None) -> Dict[str, Any]:
        p = super().payload(spec, prompt)
        p.pop("messages", None)
        p["prompt"] = spec.prompt if prompt is None else prompt
        return p


class TGI(Protocol):
    """Hugging Face text-generation-inference: ``/generate`` and ``/generate_stream``."""

    name = "tgi"
    sse_dialect = "tgi"

    def path(self, spec: RequestSpec) -> str:
        return "/generate_stream" if spec.stream else "/generate"

    def payload(self, spec: RequestSpec, prompt: Optional[str] = None) -> Dict[str, Any]:
        params: Dict[str, Any] = {"max_new_tokens": spec.max_tokens, "return_full_text": False, "details": True}
        if spec.temperature is not None and spec.temperature > 0:
            params["do_sample"] = True
            params["temperature"] = spec.temperature
        else:
            params["do_sample"] = False
        if spec.top_p is not None:
            params["top_p"] = spec.top_p
        if spec.top_k is not None:
            params["top_k"] = spec.top_k
        if spec.frequency_penalty is not None:
            params["frequency_penalty"] = spec.frequency_penalty
        params.update(spec.extra_payload.get("parameters", {}) if spec.extra_payload else {})
        return {"inputs": spec.prompt if prompt is None else prompt, "parameters": params, "stream": spec.stream}

    def parse_body(self, body: Any) -> Tuple[Optional[Dict[str, int]], str]:
        if isinstance(body, list) and body:
            body = body[0]
        if not isinstance(body, dict):
            return None, ""
        text = body.get("generated_text") or ""
        det = body.get("details") or {}
        usage = None
        if det.get("generated_tokens") is not None:
            usage = {"completion_tokens": int(det["generated_tokens"])}
            if det.get("prefill") is not None:
                usage["prompt_tokens"] = len(det["prefill"])
                usage["total_tokens"] = usage["prompt_tokens"] + usage["completion_tokens"]
        return usage, text

    def health_paths(self) -> List[str]:
        return ["/health", "/info", "/"]


def count_tokens_from_outputs(outputs: List[Dict[str, Any]], text: str = "") -> int:
    """Token count from a KServe-v2 ``outputs`` list: an explicit length tensor
    (``sequence_length`` / ``output_len`` / ``output_lengths``) wins, else ~4 chars
    per token of the text (reference ``scripts/triton_token_utils.py:4-22``)."""
    for o in outputs or []:
        if str(o.get("name", "")).lower() in ("sequence_length", "output_len", "output_lengths") and o.get("data"):
            try:
                return int(o["data"][0])
            except (TypeError, ValueError, IndexError):
                continue
    if isinstance(text, str) and text:
        return max(1, len(text) // 4)
    return 0


def update_tokens_from_stream_event(tokens_generated: int, prev_text_len: int, data: Any) -> Tuple[int, int]:
    """Advance streaming token counters by one KServe-v2 SSE event
    (reference ``scripts/triton_token_utils.py:25-52``)."""
    if not isinstance(data, dict):
        return tokens_generated, prev_text_len
    if "token_id" in data or "output_token" in data:
        return tokens_generated + 1, prev_text_len
    if "tokens_generated" in data:
        try:
            return tokens_generated + int(data["tokens_generated"]), prev_text_len
        except (TypeError, ValueError):
            pass
    txt = data.get("text")
    if isinstance(txt, str) and len(txt) > prev_text_len:
        tokens_generated += max(0, (len(txt) - prev_text_len) // 4)
        prev_text_len = len(txt)
    return tokens_generated, prev_text_len


class KServeV2Generate(Protocol):
    """KServe v2 "generate" extension (``/v2/models/<m>/generate[_stream]``) with
    ``text_input`` / ``max_tokens`` tensors."""

    name = "kserve-v2"
    sse_dialect = "kserve-v2"

    def __init__(self, model_path_name: str = "ensemble"):
        self.model_path_name = model_path_name

    def path(self, spec: RequestSpec) -> str:
        return f"/v2/models/{self.model_path_name}/" + ("generate_stream" if spec.stream else "generate")

    def payload(self, spec: RequestSpec, prompt: Optional[str] = None) -> Dict[str, Any]:
        def t(name: str, dtype: str, val: Any) -> Dict[str, Any]:
            return {"name": name, "shape": [1, 1], "datatype": dtype, "data": [val]}

        return {
            "inputs": [
                t("text_input", "BYTES", spec.prompt if prompt is None else prompt),
                t("max_tokens", "UINT32", spec.max_tokens),
                t("bad_words", "BYTES", ""),
                t("stop_words", "BYTES", ""),
            ],
            "outputs": [{"name": "text_output"}],
        }

    def parse_body(self, body: Any) -> Tuple[Optional[Dict[str, int]], str]:
        if not isinstance(body, dict):
            return None, ""
        outputs = body.get("outputs") or []
        text = body.get("text_output") or ""
        for o in outputs:
            if o.get("name") == "text_output" and o.get("data"):
                text = o["data"][0]
        n = count_tokens_from_outputs(outputs, text)
        return ({"completion_tokens": n} if n else None), text

    def health_paths(self) -> List[str]:
        return ["/v2/health/ready", "/v2", "/"]


PROTOCOLS = {
    "openai": OpenAIChat,
    "openai-completions": OpenAICompletions,
    "tgi": TGI,
    "kserve-v2": KServeV2Generate,
}


def get_protocol(name: str) -> Protocol:
    try:
        return PROTOCOLS[name]()
    except KeyError:
        raise ValueError(f"unknown protocol {name!r}; expected one of {sorted(PROTOCOLS)}")
