"""OpenAI-compatible HTTP front-end of the native MI355X engine (``kvmini serve``).

Endpoints: ``POST /v1/chat/completions``, ``POST /v1/completions`` (stream and
non-stream, ``stream_options.include_usage``, ``n`` choices, ``stop`` strings, ``logprobs`` /
``top_logprobs``), ``GET /v1/models``,
``GET /health``, ``GET /metrics`` (Prometheus text with the ``vllm:*`` names the
dashboards and the analyzer's cache-hit query use), TGI ``/generate`` and
``/generate_stream`` for the A/B harness.

This is the stand-in for "KServe vLLM-ROCm runtime" on a box with no vLLM and no
Kubernetes: the benchmark harness talks to it exactly as it would talk to a
KServe predictor URL.  Tokens flow engine thread -> per-request ``asyncio.Queue``
with one ``call_soon_threadsafe`` per engine step; the first content chunk is
written as soon as the first sampled token arrives (TCP_NODELAY).
"""

from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys
import time
import uuid
from typing import Any, AsyncIterator, Dict, List, Optional, Tuple

from kvmini.engine.engine import EngineConfig, LLMEngine, TokenEvent
from kvmini.engine.scheduler import SamplingParams
from kvmini.engine.tokenizer import get_tokenizer

from .httpd import HttpServer, Request, Response, StreamingResponse, sse_event


class NativeServer:
    def __init__(self, engine: LLMEngine, model_name: Optional[str] = None, tokenizer_path: Optional[str] = None, ignore_eos: bool = True, api_key: Optional[str] = None):
        self.engine = engine
        self.model_name = model_name or engine.mcfg.name
        cfg = engine.mcfg
        self.tok = get_tokenizer(cfg.vocab_size, cfg.bos_token_id, cfg.eos_token_id, tokenizer_path)
        # random-init weights have no meaningful EOS: generate exactly max_tokens
        self.ignore_eos = ignore_eos
        self.api_key = api_key
        self.started_at = time.time()
        self.ready_at: Optional[float] = None
        self.loop: Optional[asyncio.AbstractEventLoop] = None
        self._queues: Dict[int, Tuple[asyncio.Queue, int]] = {}  # request id -> (queue, choice index)
        self.requests_total = 0
        self.errors_total = 0
        engine.on_tokens = self._on_tokens

    # engine thread -> event loop ---------------------------------------
    def _on_tokens(self, events: List[TokenEvent]) -> None:
        loop = self.loop
        if loop is not None:
            loop.call_soon_threadsafe(self._deliver, events)

    def _deliver(self, events: List[TokenEvent]) -> None:
        for req_id, toks, done, reason, seq in events:
            ent = self._queues.get(req_id)
            if ent is not None:
                q, idx = ent
                q.put_nowait((idx, toks, done, reason, seq))

    # helpers --------------------------------------------------------------
    def _params(self, body: Dict[str, Any], chat: bool = True) -> SamplingParams:
        max_tokens = body.get("max_tokens") or body.get("max_completion_tokens") or 16
        ignore = body.get("ignore_eos")
        if chat:  # chat: logprobs is a flag, top_logprobs the count; completions: logprobs is the count
            n_lp = int(body.get("top_logprobs") or 0) or (1 if body.get("logprobs") else 0)
        else:
            n_lp = int(body.get("logprobs") or 0)
        ignore_eos = self.ignore_eos if ignore is None else bool(ignore)
        stops = [int(t) for t in (body.get("stop_token_ids") or [])]
        if not ignore_eos:  # every end-of-sequence id the checkpoint declares ends a reply, not only the first
            stops += [t for t in getattr(self.engine.mcfg, "extra_eos_token_ids", ()) if t not in stops]
        return SamplingParams(
            max_tokens=int(max_tokens),
            temperature=float(body.get("temperature") or 0.0),
            top_p=float(body.get("top_p") or 1.0),
            top_k=int(body.get("top_k") or 0),
            seed=self._seed(body.get("seed")),
            ignore_eos=ignore_eos,
            logprobs=max(0, min(n_lp, 20)),
            stop_token_ids=stops,
            presence_penalty=float(body.get("presence_penalty") or 0.0),
            frequency_penalty=float(body.get("frequency_penalty") or 0.0),
            repetition_penalty=float(body.get("repetition_penalty") or 1.0),
            logit_bias=self._logit_bias(body.get("logit_bias")),
            min_tokens=int(body.get("min_tokens") or 0),
        )

    @staticmethod
    def _logit_bias(v: Any) -> Optional[Dict[int, float]]:
        """OpenAI's ``logit_bias``: a JSON object whose keys are token ids written as strings."""
        if not v:
            return None
        if not isinstance(v, dict):
            raise ValueError("logit_bias must be an object of token id -> bias")
        try:
            return {int(k): float(b) for k, b in v.items()}
        except (TypeError, ValueError):
            raise ValueError("logit_bias must map token ids to numbers") from None

    @staticmethod
    def _seed(v: Any) -> Optional[int]:
        """A request's seed as an int (or None); anything else is the client's error."""
        if v is None:
            return None
        if isinstance(v, bool) or not isinstance(v, (int, float, str)):
            raise ValueError(f"seed must be an integer, got {v!r}")
        if isinstance(v, float) and v != int(v):
            raise ValueError(f"seed must be an integer, got {v!r}")
        return int(v)

    def _reject(self, prompt_ids: List[int], params: SamplingParams) -> Optional[Response]:
        """400 for a request the engine could never serve (too long for the model or the KV
        cache, non-finite sampling parameters): checked here, before anything is queued."""
        from kvmini.engine.scheduler import Sequence

        why = self.engine.scheduler.validate(Sequence(0, prompt_ids, params))
        if why is None:
            return None
        self.errors_total += 1
        return Response.json({"error": {"message": why, "type": "invalid_request_error"}}, 400)

    # guided decoding ------------------------------------------------------
    def _schema_start(self, schema: Any) -> Tuple[str, Tuple[int, int, int], Optional[List[int]], Optional[List[str]]]:
        """(text the server writes itself, automaton start, skeleton script, text of its entries)
        for a JSON schema: the opening brace and the first required key go in front of the
        generation, every further required key (with its punctuation, and the closing brace) is a
        run of forced tokens, and each value is restricted to the property's type
        (``kvmini.engine.guided.schema_script``)."""
        from kvmini.engine import guided as G

        return G.schema_script(schema if isinstance(schema, dict) else {}, self.tok.encode_pieces)

    @staticmethod
    def _tool_choice_schema(tools: List[Dict[str, Any]], allow_text: bool, parallel: bool) -> Dict[str, Any]:
        """The document a tool-choosing sequence writes: ``{"tool_calls": [{"name": <one of the
        tools>, "arguments": <that tool's parameters>}, ...]}`` with at least one call, or — when
        the model may also answer in text — ``{"content": "..."}``.  Every decision in it (text or
        calls, which tool, one more call or none, optional arguments, enum values) is a choice of
        the guided-decoding script, so the model takes it with its own logits."""
        calls = []
        for t in tools:
            params = t["function"].get("parameters")
            calls.append({"type": "object", "required": ["name", "arguments"],
                          "properties": {"name": {"const": str(t["function"]["name"])}, "arguments": params if isinstance(params, dict) and params else {"type": "object"}}})
        arr: Dict[str, Any] = {"type": "array", "minItems": 1, "items": calls[0] if len(calls) == 1 else {"oneOf": calls}}
        if not parallel:
            arr["maxItems"] = 1
        called = {"type": "object", "required": ["tool_calls"], "properties": {"tool_calls": arr}}
        if not allow_text:
            return called
        return {"oneOf": [called, {"type": "object", "required": ["content"], "properties": {"content": {"type": "string"}}}]}

    @staticmethod
    def _chosen_reply(text: str) -> Tuple[Optional[str], List[Dict[str, Any]]]:
        """(content, tool calls) of the document a tool-choosing sequence wrote (a document cut
        short by ``max_tokens`` that does not parse is returned as content, as it is)."""
        try:
            doc = json.loads(text)
        except ValueError:
            return text, []
        if not isinstance(doc, dict):
            return text, []
        calls = [{"id": "call_" + uuid.uuid4().hex[:24], "type": "function",
                  "function": {"name": str(c.get("name")), "arguments": json.dumps(c.get("arguments", {}), separators=(",", ":"))}}
                 for c in doc.get("tool_calls") or [] if isinstance(c, dict)]
        content = doc.get("content")
        return (content if isinstance(content, str) else None), calls

    def _guided_plan(self, body: Dict[str, Any], chat: bool) -> Tuple[Optional[List[Dict[str, Any]]], bool]:
        """What a request asks the automaton for.  Returns (sub-requests, they are tool calls):
        ``response_format`` json_object / json_schema (or vLLM's ``guided_json``) = one guided
        sequence per choice.  ``tools`` with a ``tool_choice`` other than "none": a NAMED function
        is one guided sequence that writes the arguments object of that call; "auto" and
        "required" are one guided sequence in which the MODEL chooses — which tools to call, how
        many calls (one when ``parallel_tool_calls`` is false), and under "auto" whether to call
        at all or to answer in text — by walking the choices of :meth:`_tool_choice_schema`."""
        rf = body.get("response_format") or {}
        tools = body.get("tools") if chat else None
        choice = body.get("tool_choice", "auto" if tools else "none")
        if tools and choice != "none":
            if not isinstance(tools, list) or not all(isinstance(t, dict) and isinstance(t.get("function"), dict) and t["function"].get("name") for t in tools):
                raise ValueError("tools must be a list of {type: function, function: {name, parameters}}")
            if isinstance(choice, dict):
                want = (choice.get("function") or {}).get("name")
                called = [t for t in tools if t["function"]["name"] == want]
                if not called:
                    raise ValueError(f"tool_choice names an unknown function {want!r}")
            elif choice in ("auto", "required"):
                schema = self._tool_choice_schema(tools[:16], allow_text=choice == "auto", parallel=bool(body.get("parallel_tool_calls", True)))
                prefix, start, script, texts = self._schema_start(schema)
                return [{"name": None, "chosen": True, "prefix": prefix, "start": start, "lead": prefix, "script": script, "texts": texts}], True
            else:
                raise ValueError(f"tool_choice must be none, auto, required or a function, got {choice!r}")
            subs = []
            for t in called[:16]:
                prefix, start, script, texts = self._schema_start(t["function"].get("parameters") or {})
                subs.append({"name": str(t["function"]["name"]), "prefix": prefix, "start": start, "lead": f"call {t['function']['name']} {prefix}", "script": script, "texts": texts})
            return subs, True
        schema = None
        if isinstance(rf, dict) and rf.get("type") == "json_schema":
            schema = (rf.get("json_schema") or {}).get("schema") or {}
        elif body.get("guided_json") is not None:
            schema = body["guided_json"] if isinstance(body["guided_json"], dict) else json.loads(str(body["guided_json"]))
        elif not (isinstance(rf, dict) and rf.get("type") == "json_object"):
            return None, False
        prefix, start, script, texts = self._schema_start(schema or {})
        return [{"name": None, "prefix": prefix, "start": start, "lead": prefix, "script": script, "texts": texts}], False

    def _ensure_guided(self) -> None:
        if not self.engine.guided_ready:
            from kvmini.engine.guided import token_classes

            cfg = self.engine.mcfg
            self.engine.set_token_classes(token_classes(self.tok, cfg.vocab_size, cfg.eos_token_id))

    def _auth(self, req: Request) -> Optional[Response]:
        if self.api_key and req.headers.get("authorization", "") != f"Bearer {self.api_key}":
            return Response.json({"error": {"message": "invalid api key", "type": "authentication_error"}}, 401)
        return None

    async def _run(self, prompt_ids: List[int], params: SamplingParams, q: Optional[asyncio.Queue] = None, idx: int = 0) -> Tuple[int, asyncio.Queue]:
        """Submit one sequence; its token events arrive on ``q`` tagged with ``idx``."""
        q = q if q is not None else asyncio.Queue()
        rid = self.engine.submit(prompt_ids, params)
        self._queues[rid] = (q, idx)
        return rid, q

    def _logprob_entries(self, seq: Any, first: int, toks: List[int]) -> List[Dict[str, Any]]:
        """OpenAI chat ``logprobs.content`` entries for output tokens ``first ..`` of a sequence."""
        out = []
        for k, t in enumerate(toks):
            if first + k >= len(seq.logprob_log):
                break
            lp, top = seq.logprob_log[first + k]
            piece = self.tok.decode_token(t) or " "
            out.append({"token": piece, "logprob": lp, "bytes": list(piece.encode()),
                        "top_logprobs": [{"token": self.tok.decode_token(i) or " ", "logprob": v, "bytes": list((self.tok.decode_token(i) or " ").encode())} for i, v in top]})
        return out

    # OpenAI ---------------------------------------------------------------
    async def chat(self, req: Request) -> Any:
        return await self._openai(req, chat=True)

    async def completions(self, req: Request) -> Any:
        return await self._openai(req, chat=False)

    async def _openai(self, req: Request, chat: bool) -> Any:
        bad = self._auth(req)
        if bad:
            return bad
        body = req.json() or {}
        self.requests_total += 1
        if chat:
            msgs = body.get("messages")
            if not isinstance(msgs, list) or not msgs:
                self.errors_total += 1
                return Response.json({"error": {"message": "messages must be a non-empty list", "type": "invalid_request_error"}}, 400)
            prompt_ids = self.tok.apply_chat_template(msgs)
        else:
            prompt = body.get("prompt", "")
            if isinstance(prompt, list):
                prompt = prompt[0] if prompt else ""
            prompt_ids = self.tok.encode(str(prompt), add_bos=True)
        try:
            n = max(1, min(int(body.get("n") or 1), 16))
            params = self._params(body, chat)
        except (TypeError, ValueError) as e:
            self.errors_total += 1
            return Response.json({"error": {"message": f"bad sampling parameter: {e}", "type": "invalid_request_error"}}, 400)
        try:
            subs, as_tools = self._guided_plan(body, chat)
        except (TypeError, ValueError) as e:
            self.errors_total += 1
            return Response.json({"error": {"message": f"bad structured-output request: {e}", "type": "invalid_request_error"}}, 400)
        if subs is not None and self.engine.spec is not None:
            self.errors_total += 1
            return Response.json({"error": {"message": "guided decoding is not available on an engine that runs speculative decoding", "type": "invalid_request_error"}}, 400)
        bad = self._reject(prompt_ids, params)
        if bad:
            return bad
        stops = body.get("stop") or []
        stops = [stops] if isinstance(stops, str) else [str(x) for x in stops if x]
        if subs is not None:
            self._ensure_guided()
            stops = []  # the automaton ends the document; stop strings could only cut it short
        stream = bool(body.get("stream"))
        include_usage = bool((body.get("stream_options") or {}).get("include_usage"))
        model = body.get("model") or self.model_name
        oid = ("chatcmpl-" if chat else "cmpl-") + uuid.uuid4().hex[:24]
        created = int(time.time())
        want_lp = params.logprobs > 0
        # n choices = n sequences that share the prompt (its KV blocks through the prefix cache)
        q: asyncio.Queue = asyncio.Queue()
        rids: List[int] = []
        lead_of = [""] * n  # text the server wrote in front of a choice's generated tokens
        chosen = False  # one guided sequence in which the model chooses its tool calls (tool_choice auto / required)
        detok: List[Any] = [None] * n  # schema-guided choices: forced skeleton tokens stand for the schema's own text
        if subs is None:
            for i in range(n):
                p_i = params if i == 0 else SamplingParams(**{**vars(params), "stop_token_ids": list(params.stop_token_ids), "seed": None if params.seed is None else params.seed + i})
                rid, _ = await self._run(prompt_ids, p_i, q, i)
                rids.append(rid)
        else:
            # guided sequences: one per choice (JSON mode) or one per tool call; the text the server
            # writes itself (schema key, tool name) is appended to the prompt so the model sees it
            if as_tools:
                n = len(subs)
                chosen = bool(subs[0].get("chosen"))
            plan = subs if as_tools else subs * n
            budget = max(8, params.max_tokens // len(plan)) if as_tools else params.max_tokens
            lead_of = [sb["prefix"] for sb in plan]
            detok = [None] * len(plan)
            for i, sb in enumerate(plan):
                p_i = SamplingParams(**{**vars(params), "stop_token_ids": list(params.stop_token_ids), "seed": None if params.seed is None else params.seed + i,
                                        "max_tokens": budget, "ignore_eos": False, "guided": "json", "guided_start": tuple(sb["start"]),
                                        "guided_script": tuple(sb["script"]) if sb.get("script") else None})
                if sb.get("script"):
                    from kvmini.engine.guided import ScriptedText

                    detok[i] = ScriptedText(self.engine._guided_cls_np, tuple(sb["start"]), budget, sb["script"], sb["texts"], self.tok.decode_token)
                ids_i = prompt_ids + (self.tok.encode(sb["lead"]) if sb["lead"] else [])
                bad = self._reject(ids_i, p_i)
                if bad:
                    for rid in rids:
                        self._queues.pop(rid, None)
                        self.engine.abort(rid)
                    return bad
                rid, _ = await self._run(ids_i, p_i, q, i)
                rids.append(rid)
        n_prompt = len(prompt_ids)
        obj = "chat.completion" if chat else "text_completion"
        toks_of: List[List[int]] = [[] for _ in range(n)]
        text_of = list(lead_of)
        reason_of: List[Optional[str]] = [None] * n
        lp_of: List[List[Dict[str, Any]]] = [[] for _ in range(n)]
        open_choices = set(range(n))

        def piece_of(idx: int, t: int, blank: str = " ") -> str:
            d = detok[idx]
            return d.piece(t) if d is not None else (self.tok.decode_token(t) or blank)

        def take(idx: int, toks: List[int], done: bool, reason: Optional[str], seq: Any) -> Tuple[str, List[Dict[str, Any]], bool]:
            """Fold one token event into choice ``idx``.  Returns (new text, logprob entries,
            finished now) after applying the stop strings (the text is cut in front of the match
            and the sequence aborted: the engine only knows stop TOKENS)."""
            if idx not in open_choices:
                return "", [], False
            first = len(toks_of[idx])
            toks_of[idx] += toks
            scripted = detok[idx] is not None
            piece = "".join(piece_of(idx, t) for t in toks) if (stream or scripted) else ""
            lps = self._logprob_entries(seq, first, toks) if want_lp else []
            full = text_of[idx] + piece if (stream or scripted) else lead_of[idx] + self.tok.decode(toks_of[idx])
            hit = min((full.find(sx) for sx in stops if sx in full), default=-1)
            if hit >= 0:
                piece = full[len(text_of[idx]) : hit] if stream else ""
                full = full[:hit]
                if not done:
                    self.engine.abort(rids[idx])
                done, reason = True, "stop"
            text_of[idx] = full
            lp_of[idx] += lps
            if done:
                reason_of[idx] = reason
                open_choices.discard(idx)
            return piece, lps, done

        def cleanup(abort_open: bool) -> None:
            for i, rid in enumerate(rids):
                self._queues.pop(rid, None)
                if abort_open and i in open_choices:
                    self.engine.abort(rid)

        if not stream:
            try:
                while open_choices:
                    take(*(await q.get()))
            finally:
                cleanup(abort_open=True)
            errs = [r for r in reason_of if r and r.startswith("error")]
            if errs:
                self.errors_total += 1
                return Response.json({"error": {"message": errs[0], "type": "internal_error"}}, 500)
            n_out = sum(len(t) for t in toks_of)
            usage = {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}
            choices = []
            if as_tools:
                cut = any(r == "length" for r in reason_of)
                if chosen:
                    content, calls = self._chosen_reply(text_of[0])
                    msg: Dict[str, Any] = {"role": "assistant", "content": content}
                    if calls:
                        msg["tool_calls"] = calls
                    choices.append({"index": 0, "message": msg, "finish_reason": "length" if cut else ("tool_calls" if calls else "stop")})
                else:
                    calls = [{"id": "call_" + uuid.uuid4().hex[:24], "type": "function", "function": {"name": sb["name"], "arguments": text_of[i]}} for i, sb in enumerate(subs or [])]
                    choices.append({"index": 0, "message": {"role": "assistant", "content": None, "tool_calls": calls}, "finish_reason": "length" if cut else "tool_calls"})
            for i in range(0 if as_tools else n):
                if chat:
                    ch: Dict[str, Any] = {"index": i, "message": {"role": "assistant", "content": text_of[i]}, "finish_reason": reason_of[i]}
                    if want_lp:
                        ch["logprobs"] = {"content": lp_of[i]}
                else:
                    ch = {"index": i, "text": text_of[i], "finish_reason": reason_of[i]}
                    if want_lp:
                        ch["logprobs"] = {"tokens": [e["token"] for e in lp_of[i]], "token_logprobs": [e["logprob"] for e in lp_of[i]],
                                          "top_logprobs": [{t["token"]: t["logprob"] for t in e["top_logprobs"]} for e in lp_of[i]]}
                choices.append(ch)
            return Response.json({"id": oid, "object": obj, "created": created, "model": model, "choices": choices, "usage": usage})

        async def gen() -> AsyncIterator[bytes]:
            base = {"id": oid, "object": (obj + ".chunk") if chat else obj, "created": created, "model": model}
            finished = False
            try:
                if chosen:
                    # the model decides between text and calls inside the document: it is folded into
                    # the reply when it is complete (a call's name is not known before its text is)
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {"role": "assistant", "content": None}, "finish_reason": None}]})
                    doc_text = lead_of[0]
                    while open_choices:
                        idx, toks, done, reason, seq = await q.get()
                        toks_of[idx] += toks
                        doc_text += "".join(piece_of(idx, t, "") for t in toks)
                        if done:
                            reason_of[idx] = reason
                            open_choices.discard(idx)
                    content, calls = self._chosen_reply(doc_text)
                    for i, call in enumerate(calls):
                        yield sse_event({**base, "choices": [{"index": 0, "delta": {"tool_calls": [{"index": i, **call}]}, "finish_reason": None}]})
                    if content is not None:
                        yield sse_event({**base, "choices": [{"index": 0, "delta": {"content": content}, "finish_reason": None}]})
                    cut = reason_of[0] == "length"
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {}, "finish_reason": "length" if cut else ("tool_calls" if calls else "stop")}]})
                elif as_tools:
                    # one choice; every tool call is an entry of delta.tool_calls that the argument pieces are appended to
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {"role": "assistant", "content": None}, "finish_reason": None}]})
                    for i, sb in enumerate(subs or []):
                        call = {"index": i, "id": "call_" + uuid.uuid4().hex[:24], "type": "function", "function": {"name": sb["name"], "arguments": lead_of[i]}}
                        yield sse_event({**base, "choices": [{"index": 0, "delta": {"tool_calls": [call]}, "finish_reason": None}]})
                    while open_choices:
                        idx, toks, done, reason, seq = await q.get()
                        toks_of[idx] += toks
                        piece = "".join(piece_of(idx, t, "") for t in toks)
                        if piece:
                            yield sse_event({**base, "choices": [{"index": 0, "delta": {"tool_calls": [{"index": idx, "function": {"arguments": piece}}]}, "finish_reason": None}]})
                        if done:
                            reason_of[idx] = reason
                            open_choices.discard(idx)
                    cut = any(r == "length" for r in reason_of)
                    yield sse_event({**base, "choices": [{"index": 0, "delta": {}, "finish_reason": "length" if cut else "tool_calls"}]})
                elif chat:
                    for i in range(n):
                        yield sse_event({**base, "choices": [{"index": i, "delta": {"role": "assistant", "content": ""}, "finish_reason": None}]})
                for i in range(n):
                    if lead_of[i] and not as_tools:
                        ch = {"index": i, "delta": {"content": lead_of[i]}, "finish_reason": None} if chat else {"index": i, "text": lead_of[i], "finish_reason": None}
                        yield sse_event({**base, "choices": [ch]})
                while open_choices:
                    idx, toks, done, reason, seq = await q.get()
                    if not stops and not want_lp:
                        # the common case, one chunk per token
                        toks_of[idx] += toks
                        for t in toks:
                            piece = piece_of(idx, t)
                            ch = {"index": idx, "delta": {"content": piece}, "finish_reason": None} if chat else {"index": idx, "text": piece, "finish_reason": None}
                            yield sse_event({**base, "choices": [ch]})
                        if done:
                            reason_of[idx] = reason
                            open_choices.discard(idx)
                    else:
                        piece, lps, done = take(idx, toks, done, reason, seq)
                        if piece or lps:
                            ch = {"index": idx, "delta": {"content": piece}, "finish_reason": None} if chat else {"index": idx, "text": piece, "finish_reason": None}
                            if want_lp:
                                ch["logprobs"] = {"content": lps}
                            yield sse_event({**base, "choices": [ch]})
                    if done:
                        fin = {"index": idx, "delta": {}, "finish_reason": reason_of[idx]} if chat else {"index": idx, "text": "", "finish_reason": reason_of[idx]}
                        yield sse_event({**base, "choices": [fin]})
                finished = True
                if include_usage:
                    n_out = sum(len(t) for t in toks_of)
                    yield sse_event({**base, "choices": [], "usage": {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}})
                yield sse_event("[DONE]")
            finally:
                cleanup(abort_open=not finished)  # client went away: stop generating for it

        return StreamingResponse(gen())

    # TGI ------------------------------------------------------------------
    async def tgi(self, req: Request) -> Any:
        bad = self._auth(req)
        if bad:
            return bad
        body = req.json() or {}
        self.requests_total += 1
        prm = body.get("parameters") or {}
        try:
            params = SamplingParams(
                max_tokens=int(prm.get("max_new_tokens") or 20),
                temperature=float(prm.get("temperature") or 0.0) if prm.get("do_sample") else 0.0,
                top_p=float(prm.get("top_p") or 1.0),
                top_k=int(prm.get("top_k") or 0),
                seed=self._seed(prm.get("seed")),
                ignore_eos=self.ignore_eos,
            )
        except (TypeError, ValueError) as e:
            self.errors_total += 1
            return Response.json({"error": f"bad parameter: {e}", "error_type": "validation"}, 422)
        prompt_ids = self.tok.encode(str(body.get("inputs", "")), add_bos=True)
        bad = self._reject(prompt_ids, params)
        if bad:
            return bad
        rid, q = await self._run(prompt_ids, params)
        if not req.path.endswith("_stream"):
            out: List[int] = []
            try:
                while True:
                    _i, toks, done, reason, _ = await q.get()
                    out += toks
                    if done:
                        break
            finally:
                self._queues.pop(rid, None)
            return Response.json({"generated_text": self.tok.decode(out), "details": {"finish_reason": reason, "generated_tokens": len(out), "prefill": [{"id": t} for t in prompt_ids]}})

        async def gen() -> AsyncIterator[bytes]:
            out: List[int] = []
            finished = False
            try:
                while True:
                    _i, toks, done, reason, _ = await q.get()
                    for i, t in enumerate(toks):
                        out.append(t)
                        ev: Dict[str, Any] = {"index": len(out) - 1, "token": {"id": t, "text": self.tok.decode_token(t) or " ", "logprob": 0.0, "special": False}, "generated_text": None, "details": None}
                        if done and i == len(toks) - 1:
                            ev["generated_text"] = self.tok.decode(out)
                            ev["details"] = {"finish_reason": reason, "generated_tokens": len(out)}
                        yield sse_event(ev)
                    if done:
                        finished = True
                        break
            finally:
                self._queues.pop(rid, None)
                if not finished:
                    self.engine.abort(rid)

        return StreamingResponse(gen())

    # misc -----------------------------------------------------------------
    async def tokenize(self, req: Request) -> Response:
        """vLLM's ``POST /tokenize``: ``{"prompt": ...}`` (``add_special_tokens``, default true, puts BOS in front) or
        ``{"messages": [...]}`` (the chat template) -> the ids the server would feed the model, and their count."""
        bad = self._auth(req)
        if bad:
            return bad
        body = req.json() or {}
        if isinstance(body.get("messages"), list) and body["messages"]:
            try:
                ids = self.tok.apply_chat_template(body["messages"])
            except Exception as e:  # a message the template cannot render is the client's error
                return Response.json({"error": {"message": f"messages cannot be rendered: {e}", "type": "invalid_request_error"}}, 400)
        elif isinstance(body.get("prompt"), str):
            ids = self.tok.encode(body["prompt"], add_bos=bool(body.get("add_special_tokens", True)))
        else:
            return Response.json({"error": {"message": "give 'prompt' (a string) or 'messages' (a non-empty list)", "type": "invalid_request_error"}}, 400)
        return Response.json({"count": len(ids), "max_model_len": self.engine.ecfg.max_model_len, "tokens": ids})

    async def detokenize(self, req: Request) -> Response:
        """vLLM's ``POST /detokenize``: ``{"tokens": [ids]}`` -> ``{"prompt": text}``."""
        bad = self._auth(req)
        if bad:
            return bad
        toks = (req.json() or {}).get("tokens")
        v = self.engine.mcfg.vocab_size
        if not isinstance(toks, list) or any(isinstance(t, bool) or not isinstance(t, int) or not 0 <= t < v for t in toks):
            return Response.json({"error": {"message": f"'tokens' must be a list of token ids in [0, {v})", "type": "invalid_request_error"}}, 400)
        return Response.json({"prompt": self.tok.decode(toks)})

    async def models(self, req: Request) -> Response:
        return Response.json({"object": "list", "data": [{"id": self.model_name, "object": "model", "created": int(self.started_at), "owned_by": "kvmini", "max_model_len": self.engine.ecfg.max_model_len}]})

    async def health(self, req: Request) -> Response:
        return Response.json({"status": "ok", "model": self.model_name, "uptime_s": time.time() - self.started_at, "ready_after_s": (self.ready_at or time.time()) - self.started_at})

    async def info(self, req: Request) -> Response:
        e = self.engine
        return Response.json({"model_id": self.model_name, "max_total_tokens": e.ecfg.max_model_len, "version": "kvmini-native", "model_device_type": e.device.type, "engine": e.metrics()})

    async def metrics(self, req: Request) -> Response:
        m = self.engine.metrics()
        name = self.model_name
        lb = f'{{model_name="{name}"}}'
        lines = [
            "# TYPE vllm:num_requests_running gauge",
            f"vllm:num_requests_running{lb} {m['running']}",
            "# TYPE vllm:num_requests_waiting gauge",
            f"vllm:num_requests_waiting{lb} {m['waiting']}",
            "# TYPE vllm:prompt_tokens_total counter",
            f"vllm:prompt_tokens_total{lb} {m['prompt_tokens']}",
            "# TYPE vllm:generation_tokens_total counter",
            f"vllm:generation_tokens_total{lb} {m['generation_tokens']}",
            "# TYPE vllm:gpu_prefix_cache_hits_total counter",
            f"vllm:gpu_prefix_cache_hits_total{lb} {m['prefix_cache_hits']}",
            "# TYPE vllm:gpu_prefix_cache_queries_total counter",
            f"vllm:gpu_prefix_cache_queries_total{lb} {m['prefix_cache_queries']}",
            "# TYPE vllm:gpu_cache_usage_perc gauge",
            f"vllm:gpu_cache_usage_perc{lb} {1.0 - m['kv_blocks_free'] / max(1, m['kv_blocks_total']):.6f}",
            "# TYPE vllm:num_preemptions_total counter",
            f"vllm:num_preemptions_total{lb} {m['preemptions']}",
            "# TYPE kvmini_engine_steps_total counter",
            f"kvmini_engine_steps_total {m['steps']}",
            "# TYPE kvmini_engine_step_ms gauge",
            f"kvmini_engine_step_ms {m['step_ms_ewma']:.4f}",
            "# TYPE kvmini_spec_tokens_proposed_total counter",
            f"kvmini_spec_tokens_proposed_total {m['spec_proposed']}",
            "# TYPE kvmini_spec_tokens_accepted_total counter",
            f"kvmini_spec_tokens_accepted_total {m['spec_accepted']}",
            "# TYPE kvmini_requests_total counter",
            f"kvmini_requests_total {self.requests_total}",
            f"kvmini_request_errors_total {self.errors_total}",
        ]
        return Response.text("\n".join(lines) + "\n", content_type="text/plain; version=0.0.4")

    def build_server(self) -> HttpServer:
        s = HttpServer()
        s.route("POST", "/v1/chat/completions", self.chat)
        s.route("POST", "/v1/completions", self.completions)
        s.route("GET", "/v1/models", self.models)
        s.route("GET", "/health", self.health)
        s.route("GET", "/", self.health)
        s.route("GET", "/info", self.info)
        s.route("GET", "/metrics", self.metrics)
        s.route("POST", "/generate", self.tgi)
        s.route("POST", "/generate_stream", self.tgi)
        s.route("POST", "/tokenize", self.tokenize)
        s.route("POST", "/detokenize", self.detokenize)
        return s

    async def serve(self, host: str, port: int, announce: bool = True) -> None:
        self.loop = asyncio.get_running_loop()
        srv = self.build_server()
        p = await srv.start(host, port)
        self.engine.start()
        self.ready_at = time.time()
        if announce:
            print(f"kvmini serve: {self.model_name} on http://{host}:{p} (load {self.engine.load_s:.1f}s, warmup {self.engine.warm_s:.1f}s)", flush=True)
        await srv.serve_forever()


def start_native_in_thread(engine: LLMEngine, port: int = 0, **kw: Any):
    """(server, base_url, stop) — engine + HTTP front-end on a background thread."""
    import threading

    ns = NativeServer(engine, **kw)
    ready = threading.Event()
    box: Dict[str, Any] = {}

    def _run() -> None:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
        ns.loop = loop
        srv = ns.build_server()
        box["port"] = loop.run_until_complete(srv.start("127.0.0.1", port))
        box["loop"], box["srv"] = loop, srv
        engine.start()
        ns.ready_at = time.time()
        ready.set()
        try:
            loop.run_forever()
        finally:
            loop.run_until_complete(srv.stop())
            loop.close()

    th = threading.Thread(target=_run, daemon=True, name="kvmini-native-http")
    th.start()
    if not ready.wait(60):
        raise RuntimeError("native server did not start")

    def stop() -> None:
        engine.stop()
        box["loop"].call_soon_threadsafe(box["loop"].stop)
        th.join(5)

    return ns, f"http://127.0.0.1:{box['port']}", stop


def add_engine_args(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--model", default="llama-3-8b", help="architecture preset (see kvmini.models.llama.PRESETS) or a Hugging Face checkpoint directory (config.json + *.safetensors [+ tokenizer.json]; Llama / Mistral form)")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--random-init", action="store_true", help="random-init weights (default when no --weights)")
    ap.add_argument("--weights", default=None, help="directory with HF *.safetensors (default: --model itself when that is a checkpoint directory)")
    ap.add_argument("--tokenizer", default=None, help="tokenizer.json (default: synthetic tokenizer)")
    ap.add_argument("--max-num-seqs", type=int, default=64)
    ap.add_argument("--max-batched-tokens", type=int, default=1024, help="token budget of one engine step (decode tokens + prompt chunk); larger = faster long-prompt prefill, longer worst-case step")
    ap.add_argument("--max-model-len", type=int, default=8192)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--num-blocks", type=int, default=0)
    ap.add_argument("--kv-memory-fraction", type=float, default=0.5)
    ap.add_argument("--no-graphs", action="store_true", help="disable HIP-graph capture (eager launches)")
    ap.add_argument("--kernels", default="auto", choices=["auto", "hip", "torch"])
    ap.add_argument("--no-prefix-caching", action="store_true")
    ap.add_argument("--quantization", default=None, choices=[None, "fp8", "mxfp4", "int4", "awq", "gptq"],
                    help="weight-only quantisation at load: fp8 (e4m3, per-row scale) or mxfp4 (OCP microscaling FP4; int4 / awq / gptq select the same 4-bit format)")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8", "fp8_e4m3"], help="fp8: e4m3 KV cache (half the bytes per token, twice the blocks), converted in registers by the attention kernels")
    ap.add_argument("--speculative-model", default=None, help="draft model preset for speculative decoding, or self:<K> (early exit after K layers of the served model)")
    ap.add_argument("--tail-damping", type=float, default=0.0, help="random-init + self:<K> only: damp layers >= K so the early-exit draft agrees with the full model (synthetic)")
    ap.add_argument("--num-speculative-tokens", type=int, default=0)
    ap.add_argument("--tensor-parallel-size", type=int, default=1, help="shard the model over this many GPUs: the command starts one rank process per GPU itself (RCCL over xGMI); under torchrun it joins the ranks torchrun started")
    ap.add_argument("--no-async-scheduling", action="store_true", help="read every step's tokens back before launching the next one (default: launch ahead; greedy requests only)")
    ap.add_argument("--launch-margin-ms", type=float, default=0.7, help="launch-ahead stepping: enqueue the next step this long before the one in flight should end, so late arrivals still make it (0 = at once)")
    ap.add_argument("--ride-along-limit", type=int, default=0, help="prompt backlogs up to this many tokens are fed to the running decodes in steps of <= 64 tokens instead of one wide step (better first-token time under bursts, slightly lower throughput; 0 = off)")
    ap.add_argument("--tune-gemm", action="store_true", help=argparse.SUPPRESS)  # accepted for old command lines: no library GEMM is left to tune
    ap.add_argument("--tp-graphs", action="store_true", help=argparse.SUPPRESS)  # the default now; kept for old command lines
    ap.add_argument("--no-tp-graphs", action="store_true", help="tensor parallelism without HIP graphs (eager launches)")
    ap.add_argument("--tp-rccl-graphs", action="store_true", help="also capture the wide-step buckets whose all-reduce is RCCL's (every rank must capture in step)")
    ap.add_argument("--tp-allreduce", default="xgmi", choices=["xgmi", "rccl"], help="tensor-parallel all-reduce: hand-written xGMI peer-memory kernels (default) or RCCL")
    ap.add_argument("--honor-eos", action="store_true", help="stop at EOS (default: ignore EOS with random-init weights)")


def engine_config_from_args(a: argparse.Namespace) -> EngineConfig:
    return EngineConfig(
        model=a.model,
        device=a.device,
        dtype=a.dtype,
        seed=a.seed,
        max_num_seqs=a.max_num_seqs,
        max_batched_tokens=a.max_batched_tokens,
        max_model_len=a.max_model_len,
        block_size=a.block_size,
        num_blocks=a.num_blocks,
        kv_memory_fraction=a.kv_memory_fraction,
        use_graphs=not a.no_graphs,
        kernels=a.kernels,
        enable_prefix_caching=not a.no_prefix_caching,
        weights_path=a.weights,
        quantization=a.quantization,
        kv_cache_dtype=a.kv_cache_dtype,
        draft_model=a.speculative_model,
        tail_damping=a.tail_damping,
        num_speculative_tokens=a.num_speculative_tokens,
        tp_size=getattr(a, "tensor_parallel_size", 1),
        tp_rank=getattr(a, "tp_rank", 0),
        tp_graphs=not getattr(a, "no_tp_graphs", False),
        tp_rccl_graphs=getattr(a, "tp_rccl_graphs", False),
        tp_custom_allreduce=getattr(a, "tp_allreduce", "xgmi") == "xgmi",
        async_scheduling=not getattr(a, "no_async_scheduling", False),
        launch_margin_ms=getattr(a, "launch_margin_ms", 0.7),
        ride_along_limit=getattr(a, "ride_along_limit", 0),
    )


def init_tensor_parallel(a: argparse.Namespace) -> Any:
    """One process per GPU (torchrun): join the RCCL world, pin this rank to its GPU and
    return the TP process group.  ``--device cpu`` uses gloo (tests)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL's peer-memory exchange needs on hosts without the legacy mode
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.tensor_parallel_size:
        raise SystemExit(f"--tensor-parallel-size {a.tensor_parallel_size} does not match the launcher's WORLD_SIZE {world} (start the command without a launcher, or with torchrun --nproc-per-node {a.tensor_parallel_size})")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.device == "cpu":
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        a.device = f"cuda:{local_rank}"
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    a.tp_rank = dist.get_rank()
    return dist.group.WORLD


def launch_tensor_parallel(argv: List[str], n: int) -> int:
    """``kvmini serve --tensor-parallel-size N`` typed without a launcher: start the N ranks as fresh
    child processes of this one (which never touches the GPU and replaces no process image), the
    way ``bench.py --gpus N`` does — rank r gets RANK / LOCAL_RANK = r and sees every GPU (RCCL's
    and the peer-memory kernels' xGMI paths need the peers' devices visible).  SIGTERM / SIGINT are
    handed on to the ranks; a rank that fails takes the others down and the exit code is non-zero.
    The reference only ever sets a number (``runners/backends/vllm/deploy.sh:78-79``)."""
    import signal
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = int(sk.getsockname()[1])
    pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs: List[Any] = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env["PYTHONPATH"] = pkg_root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        procs.append(subprocess.Popen([sys.executable, "-m", "kvmini", "serve"] + list(argv), env=env))

    def hand_on(signum: int, _frame: Any) -> None:
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(signum)
                except OSError:
                    pass

    signal.signal(signal.SIGTERM, hand_on)
    signal.signal(signal.SIGINT, hand_on)
    rc = 0
    deadline = None
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    print(f"kvmini serve: tensor-parallel rank {r} exited with {code}", file=sys.stderr, flush=True)
                    rc = rc or (code if code > 0 else 1)
                # once any rank has left (the server was told to stop, or a rank failed) the others cannot
                # finish a collective without it: give them a moment to leave by themselves
                deadline = deadline or time.time() + (15.0 if code == 0 else 5.0)
            if deadline is not None and time.time() > deadline:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                rc = rc or 1
    return rc


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini serve", description="Native MI355X OpenAI-compatible LLM server")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--service", default="demo-llm", help="name in the local cluster registry")
    ap.add_argument("--namespace", default="local")
    ap.add_argument("--no-register", action="store_true")
    add_engine_args(ap)
    a = ap.parse_args(argv)
    if a.tensor_parallel_size > 1 and "RANK" not in os.environ:
        return launch_tensor_parallel(list(sys.argv[1:] if argv is None else argv), a.tensor_parallel_size)
    proc_start = time.time()
    import signal

    def _term(*_: Any) -> None:  # SIGTERM = orderly shutdown (the finally block releases TP followers)
        raise KeyboardInterrupt

    signal.signal(signal.SIGTERM, _term)
    tp_group = init_tensor_parallel(a) if a.tensor_parallel_size > 1 else None
    engine = LLMEngine(engine_config_from_args(a), tp_group=tp_group)
    engine.warmup()
    if tp_group is not None and a.tp_rank != 0:
        # follower rank: no HTTP server, no scheduler — run the steps rank 0 broadcasts
        import torch.distributed as dist

        served = engine.follow()
        print(f"[tp rank {a.tp_rank}] served {served} steps", flush=True)
        dist.destroy_process_group()
        return 0
    tokenizer = a.tokenizer
    if tokenizer is None and os.path.isfile(os.path.join(a.model, "tokenizer.json")):
        tokenizer = os.path.join(a.model, "tokenizer.json")  # --model names a checkpoint directory: its tokenizer goes with it
    ns = NativeServer(engine, a.served_model_name or a.model, tokenizer, ignore_eos=not a.honor_eos, api_key=a.api_key)
    ns.started_at = proc_start
    if not a.no_register:
        from kvmini.cluster import LocalBackend

        LocalBackend().register_server(a.service, f"http://{a.host}:{a.port}", os.getpid(), proc_start, namespace=a.namespace, gpus=a.tensor_parallel_size if engine.device.type == "cuda" else 0)
    try:
        asyncio.run(ns.serve(a.host, a.port))
    except KeyboardInterrupt:
        pass
    finally:
        engine.stop()  # also releases tensor-parallel followers
        if not a.no_register:
            from kvmini.cluster import LocalBackend

            LocalBackend().mark_finished(os.getpid())
        if tp_group is not None:
            import torch.distributed as dist

            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
