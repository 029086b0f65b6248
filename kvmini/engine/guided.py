"""Guided (structured) decoding: a JSON pushdown automaton over token CLASSES.

The reference's ``structured-output`` and ``tool-calling`` profiles ask the serving runtime for
guided JSON (``runners/profiles/structured-output.yaml:17``, ``tool-calling.yaml:18-19``) and its
parity probe checks ``response_format={"type":"json_object"}`` and ``tool_calls``
(``scripts/openai_parity_probe.py:71-186``).  Here the constraint is grammar-free: every token of
the vocabulary is put in one of 13 classes by its text (``{`` ``}`` ``[`` ``]`` ``:`` ``,`` ``"``,
string-safe text, digit runs, ``true``/``false``/``null``, end of sequence, everything else), and a
small automaton over (state, nesting depth, container stack bits, tokens left) says which classes
may come next.  A class is allowed only if, after taking it, the document can still be closed
within the tokens that are left, so the output is complete JSON whenever ``max_tokens >= 3``.

The automaton runs ON THE DEVICE (``csrc/guided.hip``: one kernel masks the logits row with the
allowed classes in front of argmax / sampling, one advances the state with the token that was
picked), so a guided request is stepped launch-ahead like any other: the host never needs the
previous token to build the next mask.  This module is the same automaton in Python — the
reference the kernels are tested against, and the implementation of the CPU path.
"""

from __future__ import annotations

import re
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

# token classes
C_OTHER, C_LBRACE, C_RBRACE, C_LBRACK, C_RBRACK, C_COLON, C_COMMA, C_QUOTE, C_STR, C_DIGIT_NZ, C_DIGIT, C_LIT, C_EOS = range(13)
N_CLASSES = 13
# automaton states
S_OFF, S_START, S_OBJ_FIRST, S_KEY, S_AFTER_KEY, S_VALUE, S_STR, S_NUM, S_AFTER_VALUE, S_OBJ_NEXT, S_ARR_FIRST, S_DONE, S_END = range(13)
MAX_DEPTH = 16

_STRUCT = {"{": C_LBRACE, "}": C_RBRACE, "[": C_LBRACK, "]": C_RBRACK, ":": C_COLON, ",": C_COMMA, '"': C_QUOTE}
_UNSAFE = re.compile(r'["\\\x00-\x1f\x7f�]')


def classify_text(text: str) -> int:
    """Class of a token by the text it decodes to."""
    if text in _STRUCT:
        return _STRUCT[text]
    if text in ("true", "false", "null"):
        return C_LIT
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
    return 1 if bool(np.isin(cls, (C_LIT, C_DIGIT_NZ)).any()) else 2


State = Tuple[int, int, int, int]  # (state, depth, stack bits, tokens left)
# Bits 16.. of the stack word: classes the NEXT token is restricted to (0 = no restriction); the
# word is cleared of them by the first advance.  A server that has written ``{"age":`` into the
# prompt uses it to make the value that follows a number.
FIRST_SHIFT = 16
TYPE_CLASSES = {"string": 1 << C_QUOTE, "integer": 1 << C_DIGIT_NZ, "number": 1 << C_DIGIT_NZ, "boolean": 1 << C_LIT, "null": 1 << C_LIT,
                "array": 1 << C_LBRACK, "object": 1 << C_LBRACE}


def _pop(depth: int, stack: int) -> Tuple[int, int, int]:
    depth -= 1
    stack &= ~(1 << depth)
    return (S_DONE if depth == 0 else S_AFTER_VALUE), depth, stack


def transition(s: int, depth: int, stack: int, c: int) -> Optional[Tuple[int, int, int]]:
    """(state, depth, stack) after a token of class ``c``; None if the grammar forbids it."""
    top_arr = depth > 0 and (stack >> (depth - 1)) & 1
    in_string = c in (C_LBRACE, C_RBRACE, C_LBRACK, C_RBRACK, C_COLON, C_COMMA, C_STR, C_DIGIT_NZ, C_DIGIT, C_LIT)
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
        if c == C_LIT:
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
            c, i = C_LIT, i + len(m.group(0)) - 1
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


def replay(cls: np.ndarray, start: Tuple[int, int, int], max_tokens: int, output_ids: Sequence[int]) -> State:
    """The automaton's state after ``output_ids`` (a sequence that is taken up again after a
    preemption gets its device state rebuilt from this)."""
    st: State = (start[0], start[1], start[2], max_tokens)
    for t in output_ids:
        st = advance(st, int(cls[t]) if 0 <= t < len(cls) else C_OTHER)
    return st


# ---- reference implementation of the two device ops (CPU path) -------------------------------
def mask_rows_reference(logits: Any, rows: List[int], slots: List[int], init: List[Sequence[int]], state: Any, cls: np.ndarray, vc: int) -> None:
    """``logits[rows[i]]``: every token whose class is not allowed gets -inf.  ``init[i][0] >= 0``
    first (re)sets the automaton of ``slots[i]``.  ``state``: int32 ``[slots, 4]`` tensor."""
    import torch

    cls_t = torch.from_numpy(cls.astype(np.int64)).to(logits.device)
    for r, slot, ini in zip(rows, slots, init):
        if ini[0] >= 0:
            state[slot] = torch.tensor(list(ini), dtype=state.dtype, device=state.device)
        st = tuple(int(v) for v in state[slot].tolist())
        allow = allowed_classes(st, vc)  # type: ignore[arg-type]
        if allow == (1 << N_CLASSES) - 1:
            continue
        ok = torch.tensor([(allow >> c) & 1 for c in range(N_CLASSES)], dtype=torch.bool, device=logits.device)[cls_t]
        logits[r, : ok.numel()] = torch.where(ok, logits[r, : ok.numel()], torch.full_like(logits[r, : ok.numel()], float("-inf")))
        logits[r, ok.numel() :] = float("-inf")


def advance_rows_reference(out_ids: Any, rows: List[int], slots: List[int], state: Any, cls: np.ndarray) -> None:
    import torch

    for r, slot in zip(rows, slots):
        tok = int(out_ids[r])
        st = tuple(int(v) for v in state[slot].tolist())
        new = advance(st, int(cls[tok]) if 0 <= tok < len(cls) else C_OTHER)  # type: ignore[arg-type]
        state[slot] = torch.tensor(list(new), dtype=state.dtype, device=state.device)
