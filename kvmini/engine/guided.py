"""Guided (structured) decoding: a JSON pushdown automaton over token CLASSES.

The reference's ``structured-output`` and ``tool-calling`` profiles ask the serving runtime for
guided JSON (``runners/profiles/structured-output.yaml:17``, ``tool-calling.yaml:18-19``) and its
parity probe checks ``response_format={"type":"json_object"}`` and ``tool_calls``
(``scripts/openai_parity_probe.py:71-186``).  Here the constraint is grammar-free: every token of
the vocabulary is put in one of 14 classes by its text (``{`` ``}`` ``[`` ``]`` ``:`` ``,`` ``"``,
string-safe text, digit runs, ``true``/``false``, ``null``, end of sequence, everything else), and a
small automaton over (state, nesting depth, container stack bits, tokens left) says which classes
may come next.  A class is allowed only if, after taking it, the document can still be closed
within the tokens that are left, so the output is complete JSON whenever ``max_tokens >= 3``.

A JSON *schema* adds a SCRIPT on top (:func:`schema_script`): the document's skeleton — every
required key, in order, with its punctuation — is a run of literal tokens that are FORCED, and
between them sit typed value slots in which the class automaton runs as above, restricted to the
property's type; when a value is complete at the top level the next literal is forced.  So the
output has every required key with a value of the declared type, in at most ``max_tokens`` tokens
whenever the skeleton with the shortest values fits.

The automaton runs ON THE DEVICE (``csrc/guided.hip``: one kernel masks the logits row with the
allowed classes in front of argmax / sampling, one advances the state with the token that was
picked), so a guided request is stepped launch-ahead like any other: the host never needs the
previous token to build the next mask.  This module is the same automaton in Python — the
reference the kernels are tested against, and the implementation of the CPU path.
"""

from __future__ import annotations

import re
from typing import Dict, Any, List, Optional, Sequence, Tuple

import numpy as np

# token classes
C_OTHER, C_LBRACE, C_RBRACE, C_LBRACK, C_RBRACK, C_COLON, C_COMMA, C_QUOTE, C_STR, C_DIGIT_NZ, C_DIGIT, C_LIT, C_EOS, C_NULL = range(14)
N_CLASSES = 14  # C_LIT = true / false, C_NULL = null (a class of its own so that a boolean slot can exclude it)
# automaton states
S_OFF, S_START, S_OBJ_FIRST, S_KEY, S_AFTER_KEY, S_VALUE, S_STR, S_NUM, S_AFTER_VALUE, S_OBJ_NEXT, S_ARR_FIRST, S_DONE, S_END = range(13)
MAX_DEPTH = 16

_STRUCT = {"{": C_LBRACE, "}": C_RBRACE, "[": C_LBRACK, "]": C_RBRACK, ":": C_COLON, ",": C_COMMA, '"': C_QUOTE}
_UNSAFE = re.compile(r'["\\\x00-\x1f\x7f�]')


def classify_text(text: str) -> int:
    """Class of a token by the text it decodes to."""
    if text in _STRUCT:
        return _STRUCT[text]
    if text in ("true", "false"):
        return C_LIT
    if text == "null":
        return C_NULL
    if not text or _UNSAFE.search(text):
        return C_OTHER  # empty (special / partial UTF-8 byte) or would need escaping inside a string
    if text.isascii() and text.isdigit():
        return C_DIGIT_NZ if text[0] != "0" else C_DIGIT
    return C_STR


def token_classes(tokenizer: Any, vocab_size: int, eos_id: int) -> np.ndarray:
    """uint8 class per token id (ids past the tokenizer's vocabulary: never allowed)."""
    cls = np.zeros(vocab_size, dtype=np.uint8)
    n = min(vocab_size, int(getattr(tokenizer, "vocab_size", vocab_size)))
    for t in range(n):
        cls[t] = classify_text(tokenizer.decode_token(t))
    if 0 <= eos_id < vocab_size:
        cls[eos_id] = C_EOS
    return cls


def value_cost(cls: np.ndarray) -> int:
    """Fewest tokens of a JSON value: 1 with a one-token literal or number, else 2 (``""``)."""
    return 1 if bool(np.isin(cls, (C_LIT, C_NULL, C_DIGIT_NZ)).any()) else 2


State = Tuple[int, int, int, int]  # (state, depth, stack bits, tokens left)
# Bits 16.. of the stack word: classes the NEXT token is restricted to (0 = no restriction); the
# word is cleared of them by the first advance.  A server that has written ``{"age":`` into the
# prompt uses it to make the value that follows a number.
FIRST_SHIFT = 16
TYPE_CLASSES = {"string": 1 << C_QUOTE, "integer": 1 << C_DIGIT_NZ, "number": 1 << C_DIGIT_NZ, "boolean": 1 << C_LIT, "null": 1 << C_NULL,
                "array": 1 << C_LBRACK, "object": 1 << C_LBRACE}


def _pop(depth: int, stack: int) -> Tuple[int, int, int]:
    depth -= 1
    stack &= ~(1 << depth)
    return (S_DONE if depth == 0 else S_AFTER_VALUE), depth, stack


def transition(s: int, depth: int, stack: int, c: int) -> Optional[Tuple[int, int, int]]:
    """(state, depth, stack) after a token of class ``c``; None if the grammar forbids it."""
    top_arr = depth > 0 and (stack >> (depth - 1)) & 1
    in_string = c in (C_LBRACE, C_RBRACE, C_LBRACK, C_RBRACK, C_COLON, C_COMMA, C_STR, C_DIGIT_NZ, C_DIGIT, C_LIT, C_NULL)
    if s == S_START:
        return (S_OBJ_FIRST, 1, 0) if c == C_LBRACE else None
    if s == S_OBJ_FIRST:
        if c == C_QUOTE:
            return S_KEY, depth, stack
        return _pop(depth, stack) if c == C_RBRACE else None
    if s == S_KEY:
        if c == C_QUOTE:
            return S_AFTER_KEY, depth, stack
        return (S_KEY, depth, stack) if in_string else None
    if s == S_AFTER_KEY:
        return (S_VALUE, depth, stack) if c == C_COLON else None
    if s in (S_VALUE, S_ARR_FIRST):
        if c == C_QUOTE:
            return S_STR, depth, stack
        if c == C_LBRACE and depth < MAX_DEPTH:
            return S_OBJ_FIRST, depth + 1, stack
        if c == C_LBRACK and depth < MAX_DEPTH:
            return S_ARR_FIRST, depth + 1, stack | (1 << depth)
        if c == C_DIGIT_NZ:
            return S_NUM, depth, stack
        if c in (C_LIT, C_NULL):
            return S_AFTER_VALUE, depth, stack
        if s == S_ARR_FIRST and c == C_RBRACK:
            return _pop(depth, stack)
        return None
    if s == S_STR:
        if c == C_QUOTE:
            return S_AFTER_VALUE, depth, stack
        return (S_STR, depth, stack) if in_string else None
    if s in (S_NUM, S_AFTER_VALUE):
        if s == S_NUM and c in (C_DIGIT, C_DIGIT_NZ):
            return S_NUM, depth, stack
        if c == C_COMMA:
            return (S_VALUE if top_arr else S_OBJ_NEXT), depth, stack
        if (c == C_RBRACK and top_arr) or (c == C_RBRACE and not top_arr):
            return _pop(depth, stack)
        return None
    if s == S_OBJ_NEXT:
        return (S_KEY, depth, stack) if c == C_QUOTE else None
    if s == S_DONE:
        return (S_END, 0, 0) if c == C_EOS else None
    return None


def close_cost(s: int, depth: int, vc: int) -> int:
    """Fewest tokens from here to the end of the sequence (closers + the EOS token)."""
    tail = depth + 1
    return {S_START: 3, S_OBJ_FIRST: tail, S_KEY: 2 + vc + tail, S_AFTER_KEY: 1 + vc + tail, S_VALUE: vc + tail, S_ARR_FIRST: tail, S_STR: 1 + tail,
            S_NUM: tail, S_AFTER_VALUE: tail, S_OBJ_NEXT: 3 + vc + tail, S_DONE: 1}.get(s, 0)


def allowed_classes(state: State, vc: int) -> int:
    """Bit mask over the classes that may come next (all ones when the automaton is off / done)."""
    s, depth, stack, left = state
    if s in (S_OFF, S_END):
        return (1 << N_CLASSES) - 1
    first, stack = stack >> FIRST_SHIFT, stack & ((1 << FIRST_SHIFT) - 1)
    fits = grammar = 0
    for c in range(1, N_CLASSES):
        nxt = transition(s, depth, stack, c)
        if nxt is None:
            continue
        grammar |= 1 << c
        if close_cost(nxt[0], nxt[1], vc) <= left - 1:
            fits |= 1 << c
    if first & grammar:  # (a restriction the grammar cannot honour is dropped)
        grammar &= first
        fits &= first
    return fits or grammar  # budget too small from the start: stay in the grammar, the output is cut short


def advance(state: State, c: int) -> State:
    s, depth, stack, left = state
    if s in (S_OFF, S_END):
        return state
    stack &= (1 << FIRST_SHIFT) - 1
    nxt = transition(s, depth, stack, c)
    if nxt is None:
        return s, depth, stack, left - 1
    return nxt[0], nxt[1], nxt[2], left - 1


def state_after_text(prefix: str) -> Tuple[int, int, int]:
    """(state, depth, stack) of the automaton after a prefix that the SERVER wrote (canonical
    JSON without whitespace, e.g. ``{"location":``): the generation continues from there."""
    s, depth, stack = S_START, 0, 0
    i = 0
    while i < len(prefix):
        ch = prefix[i]
        if s in (S_KEY, S_STR) and ch != '"':
            if ch == "\\":
                i += 1
            i += 1
            continue
        if ch.isspace():
            i += 1
            continue
        m = re.match(r"true|false|null", prefix[i:]) if s in (S_VALUE, S_ARR_FIRST) else None
        if m:
            c, i = (C_NULL if m.group(0) == "null" else C_LIT), i + len(m.group(0)) - 1
        elif ch.isdigit() and s in (S_VALUE, S_ARR_FIRST, S_NUM):
            c = C_DIGIT_NZ if (ch != "0" or s == S_NUM) else C_DIGIT
            if s != S_NUM and ch == "0":
                raise ValueError("numbers in a guided prefix may not start with 0")
        else:
            c = _STRUCT.get(ch, C_OTHER)
        nxt = transition(s, depth, stack, c)
        if nxt is None:
            raise ValueError(f"not a JSON prefix at offset {i}: {prefix!r}")
        s, depth, stack = nxt
        i += 1
    return s, depth, stack


def replay(cls: np.ndarray, start: Tuple[int, int, int], max_tokens: int, output_ids: Sequence[int], script: Optional[Sequence[int]] = None) -> Tuple[State, int]:
    """The automaton's state and script cursor after ``output_ids`` (a sequence that is taken up
    again after a preemption gets its device state rebuilt from this)."""
    st: State = (start[0], start[1], start[2], max_tokens)
    cur = 0 if script else -1
    for t in output_ids:
        st, cur = advance_scripted(st, cur, int(t), cls, script)
    return st, cur


# ---- scripts: the skeleton of a schema's document ----------------------------------------------
SCRIPT_END = -1
SCRIPT_MAX = 256  # entries per sequence the device buffer holds


def script_value(type_mask: int) -> int:
    """Script entry of a value slot whose first token is restricted to ``type_mask`` (0 = any)."""
    return -2 - int(type_mask)


def _forcing(s: int, depth: int) -> bool:
    """At the top level with a value just completed (or inside a literal): the script's next
    literal token comes next."""
    return depth == 1 and s in (S_AFTER_VALUE, S_NUM, S_OBJ_NEXT)


def _value_min_cost(type_mask: int, vc: int) -> int:
    if type_mask == 0:
        return vc
    return 1 if type_mask & ((1 << C_DIGIT_NZ) | (1 << C_LIT) | (1 << C_NULL)) else 2  # a one-token number / literal, else "" / {} / []


def script_rest(script: Sequence[int], vc: int) -> List[int]:
    """``rest[i]`` = fewest tokens from script entry ``i`` to the end of the sequence (literal
    tokens + the shortest value of every slot + the EOS token)."""
    rest = [0] * (len(script) + 1)
    rest[len(script)] = 1
    for i in range(len(script) - 1, -1, -1):
        e = script[i]
        rest[i] = rest[i + 1] + (0 if e == SCRIPT_END else 1 if e >= 0 else _value_min_cost(-2 - e, vc))
    return rest[: len(script)]


def allowed_scripted(state: State, cursor: int, script: Optional[Sequence[int]], rest: Optional[Sequence[int]], vc: int) -> Tuple[int, int]:
    """(class mask, forced token id or -1) for the next token of a scripted sequence."""
    s, depth, stack, left = state
    if not script or cursor < 0 or cursor >= len(script) or script[cursor] == SCRIPT_END or s in (S_OFF, S_END):
        return allowed_classes(state, vc), -1
    e = script[cursor]
    stack &= (1 << FIRST_SHIFT) - 1
    if e >= 0 and _forcing(s, depth):
        more_digits = s == S_NUM and left - 1 >= rest[cursor]  # room for one more digit and all that has to follow
        return (((1 << C_DIGIT) | (1 << C_DIGIT_NZ)) if more_digits else 0), e
    # inside (or at the start of) a value: the class automaton, with the budget of everything the
    # script still has to write after this value (close_cost counts "}" + EOS from the top level: 2)
    after = rest[cursor + 1] if e < 0 else rest[cursor]
    first = (-2 - e) if (e < 0 and s == S_VALUE and depth == 1) else 0
    fits = grammar = 0
    for c in range(1, N_CLASSES):
        nxt = transition(s, depth, stack, c)
        if nxt is None or nxt[1] < 1:
            continue  # (closing the top-level object is the script's business)
        if depth == 1 and s in (S_AFTER_VALUE, S_NUM) and c in (C_COMMA, C_RBRACE):
            continue
        grammar |= 1 << c
        if close_cost(nxt[0], nxt[1], vc) - 2 + after <= left - 1:
            fits |= 1 << c
    if first & grammar:
        grammar &= first
        fits &= first
    return (fits or grammar), -1


def advance_scripted(state: State, cursor: int, tok: int, cls: np.ndarray, script: Optional[Sequence[int]]) -> Tuple[State, int]:
    """State and cursor after token ``tok``."""
    s, depth, stack, left = state
    c = int(cls[tok]) if 0 <= tok < len(cls) else C_OTHER
    if not script or cursor < 0 or cursor >= len(script) or script[cursor] == SCRIPT_END or s in (S_OFF, S_END):
        return advance(state, c), cursor
    e = script[cursor]
    if e >= 0 and _forcing(s, depth) and tok == e:
        cursor += 1
        nxt = script[cursor] if cursor < len(script) else SCRIPT_END
        # the automaton is SET at the end of a literal (merged tokens such as '":' have no class
        # the grammar could follow): a value slot next, or the document is closed
        if nxt == SCRIPT_END:
            return (S_DONE, 0, 0, left - 1), cursor
        if nxt < 0:
            return (S_VALUE, 1, 0, left - 1), cursor
        return (S_OBJ_NEXT, 1, 0, left - 1), cursor
    if e < 0 and s == S_VALUE and depth == 1:
        cursor += 1  # the slot's value has begun
    return advance(state, c), cursor


def schema_script(schema: Any, encode_pieces: Any) -> Tuple[str, Tuple[int, int, int], Optional[List[int]], Optional[List[str]]]:
    """(text the server writes in front, automaton start, script, text of every script entry) for
    a JSON schema of an object.  ``encode_pieces(text) -> [(token id, text piece)]`` (no BOS; the
    pieces concatenate to ``text``).  The schema is flattened into ONE line of literals and value
    slots, at any depth:

    * an object with ``properties`` writes every ``required`` property (all of them when nothing
      is required) in order, recursively;
    * an array whose ``items`` is a schema writes ``max(minItems, 1)`` items (none when
      ``maxItems`` is 0), each flattened like any other value;
    * ``enum`` / ``const`` values are written literally (the first enum entry);
    * ``$ref`` (into ``$defs`` / ``definitions``) and ``allOf`` are resolved; of ``anyOf`` /
      ``oneOf`` the first alternative that is not ``null`` is written (``Optional[X]``: an X);
    * every other value is a slot restricted to the declared ``type`` — scalars, and objects /
      arrays that declare nothing about their content (free JSON of that container type).

    The automaton is SET at the end of every literal and a slot always runs at its own depth 1,
    so the brackets inside the literals need no tracking.  Without properties at the top: no
    script, the document is free JSON.  The texts are what a forced token stands for in the
    response (a tokenizer need not decode a token back to the piece it encoded)."""
    import json as _json

    from ..utils.jsonschema_lite import preferred

    top = preferred(schema, schema)
    props = top.get("properties")
    if not isinstance(props, dict) or not props:
        return "", (S_START, 0, 0), None, None
    script: List[int] = []
    texts: List[str] = []
    out = {"prefix": None, "pending": ""}  # pending: literal text waiting in front of the next slot / the end

    def literal(text: str) -> None:
        for tid, piece in encode_pieces(text):
            script.append(int(tid))
            texts.append(piece)

    def slot(mask: int) -> None:
        if out["prefix"] is None:
            out["prefix"] = out["pending"]  # the first literal goes into the prompt: the model sees it, nothing to force
        else:
            literal(out["pending"])
        out["pending"] = ""
        script.append(script_value(mask))
        texts.append("")

    on_path: List[str] = []  # references being expanded: a recursive schema stops at its first repeat

    def value(spec: Any, depth: int) -> None:
        ref = spec.get("$ref") if isinstance(spec, dict) else None
        spec = preferred(spec, schema)  # $ref / allOf resolved; of anyOf / oneOf the first alternative that is not null
        if isinstance(ref, str):
            if ref in on_path:  # a node of its own kind below itself: free JSON of that type from here on
                spec = {"type": spec.get("type")} if "type" in spec else {}
            on_path.append(ref)
            try:
                emit(spec, depth)
            finally:
                on_path.pop()
            return
        emit(spec, depth)

    def emit(spec: Dict[str, Any], depth: int) -> None:
        if "const" in spec or (isinstance(spec.get("enum"), list) and spec["enum"]):
            out["pending"] += _json.dumps(spec["const"] if "const" in spec else spec["enum"][0], separators=(",", ":"))
            return
        types = spec.get("type")
        tl = [types] if isinstance(types, str) else (list(types) if isinstance(types, list) else [])
        sub = spec.get("properties")
        if isinstance(sub, dict) and sub and (not tl or "object" in tl) and depth < 16:
            keys = [k for k in (spec.get("required") or []) if k in sub] or list(sub)
            out["pending"] += "{"
            for i, key in enumerate(keys):
                out["pending"] += ("," if i else "") + _json.dumps(str(key)) + ":"
                value(sub[key], depth + 1)
            out["pending"] += "}"
            return
        items = spec.get("items")
        if isinstance(items, dict) and items and (not tl or "array" in tl) and depth < 16:
            lo = spec.get("minItems") if isinstance(spec.get("minItems"), int) else 0
            hi = spec.get("maxItems") if isinstance(spec.get("maxItems"), int) else None
            n = max(lo, 1) if hi is None else min(max(lo, 1), max(hi, 0))
            out["pending"] += "["
            for i in range(n):
                out["pending"] += "," if i else ""
                value(items, depth + 1)
            out["pending"] += "]"
            return
        mask = 0
        for t in tl:
            mask |= TYPE_CLASSES.get(t, 0)
        slot(mask)

    value({"type": "object", "properties": props, "required": top.get("required")}, 0)
    literal(out["pending"])
    script.append(SCRIPT_END)
    texts.append("")
    if len(script) > SCRIPT_MAX:
        raise ValueError(f"schema skeleton of {len(script)} tokens exceeds the {SCRIPT_MAX} the guided-decoding script holds")
    prefix = out["prefix"]
    if prefix is None:  # every value was a constant: the whole document is one literal
        return "", (S_OBJ_NEXT, 1, 0), script, texts
    state_after_text(prefix)  # raises on a prefix that is not JSON
    return prefix, (S_VALUE, 1, 0), script, texts  # the first slot: the state every literal ends in


class ScriptedText:
    """Text of a scripted sequence, token by token: a forced skeleton token stands for the piece
    of the schema's literal it was encoded from, every other token for what it decodes to (the
    host-side twin of the device automaton; also what a response is rebuilt from)."""

    def __init__(self, cls: np.ndarray, start: Tuple[int, int, int], max_tokens: int, script: Sequence[int], texts: Sequence[str], decode_token: Any):
        self.cls, self.script, self.texts, self.decode_token = cls, list(script), list(texts), decode_token
        self.state: State = (start[0], start[1], start[2], max_tokens)
        self.cursor = 0

    def piece(self, tok: int) -> str:
        cur = self.cursor
        self.state, self.cursor = advance_scripted(self.state, cur, int(tok), self.cls, self.script)
        if self.cursor == cur + 1 and 0 <= cur < len(self.script) and self.script[cur] >= 0:
            return self.texts[cur]
        return self.decode_token(int(tok))


# ---- reference implementation of the two device ops (CPU path) -------------------------------
def _slot_script(script: Any, slot: int) -> Tuple[Optional[List[int]], Optional[List[int]]]:
    """(entries, rest) of a slot from the ``[slots, 2, SCRIPT_MAX]`` script tensor (None: no scripts)."""
    if script is None:
        return None, None
    rows = script[slot].tolist()
    return rows[0], rows[1]


def mask_rows_reference(logits: Any, rows: List[int], slots: List[int], init: List[Sequence[int]], state: Any, cls: np.ndarray, vc: int,
                        script: Any = None, cursor: Any = None, cur_init: Optional[List[int]] = None) -> None:
    """``logits[rows[i]]``: every token that is not allowed gets -inf.  ``init[i][0] >= 0``
    first (re)sets the automaton of ``slots[i]`` (and ``cur_init[i]`` its script cursor).
    ``state``: int32 ``[slots, 4]``; ``script``: int32 ``[slots, 2, SCRIPT_MAX]`` (entries, rest);
    ``cursor``: int32 ``[slots]`` (-1: the slot has no script)."""
    import torch

    cls_t = torch.from_numpy(cls.astype(np.int64)).to(logits.device)
    for i, (r, slot, ini) in enumerate(zip(rows, slots, init)):
        if ini[0] >= 0:
            state[slot] = torch.tensor(list(ini), dtype=state.dtype, device=state.device)
            if cursor is not None:
                cursor[slot] = cur_init[i] if cur_init is not None else -1
        st = tuple(int(v) for v in state[slot].tolist())
        cur = int(cursor[slot]) if cursor is not None else -1
        ent, rest = _slot_script(script, slot) if cur >= 0 else (None, None)
        allow, forced = allowed_scripted(st, cur, ent, rest, vc)  # type: ignore[arg-type]
        if allow == (1 << N_CLASSES) - 1:
            continue
        ok = torch.tensor([(allow >> c) & 1 for c in range(N_CLASSES)], dtype=torch.bool, device=logits.device)[cls_t]
        if 0 <= forced < ok.numel():
            ok[forced] = True
        logits[r, : ok.numel()] = torch.where(ok, logits[r, : ok.numel()], torch.full_like(logits[r, : ok.numel()], float("-inf")))
        logits[r, ok.numel() :] = float("-inf")


def advance_rows_reference(out_ids: Any, rows: List[int], slots: List[int], state: Any, cls: np.ndarray, script: Any = None, cursor: Any = None) -> None:
    import torch

    for r, slot in zip(rows, slots):
        tok = int(out_ids[r])
        st = tuple(int(v) for v in state[slot].tolist())
        cur = int(cursor[slot]) if cursor is not None else -1
        ent, _ = _slot_script(script, slot) if cur >= 0 else (None, None)
        new, cur2 = advance_scripted(st, cur, tok, cls, ent)  # type: ignore[arg-type]
        state[slot] = torch.tensor(list(new), dtype=state.dtype, device=state.device)
        if cursor is not None and cur2 != cur:
            cursor[slot] = cur2
