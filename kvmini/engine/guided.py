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
# A script is a small program over int32 entries, walked by a cursor:
#   e >= 0            literal token of the skeleton: FORCED once the value in front of it is complete
#   SCRIPT_END (-1)   the document is closed
#   -2 - mask         value slot whose first token is restricted to the classes in ``mask`` (0 = any)
#   JMP_BASE - t      continue at entry t (the joins of alternatives, the back edge of an array's item loop)
#   CHOICE_BASE - n   the MODEL chooses among n alternatives; 2 n entries follow: (key, target) pairs.
#                     key >= 0: the alternative starts with that token; key = a slot code: it starts with
#                     a value of those classes.  After the key the script continues at ``target``.
# Choices are what ``enum`` (a trie over the values' tokens), ``anyOf`` / ``oneOf`` (alternatives told
# apart by their first token) and arrays between ``minItems`` and ``maxItems`` ("," + another item, or
# "]") compile to.  The cursor never rests on a jump.
SCRIPT_END = -1
SCRIPT_MAX = 1024  # entries per sequence the device buffer holds
JMP_BASE = -20000
CHOICE_BASE = -30000
MAX_ALTS = 16
MAX_OPTIONAL = 3  # optional properties of an object the model may add (each is a choice; the compiled size grows with their square)
_INF = 1 << 20
# While a schema is being compiled, a jump / choice target at or above _LABEL_USE is symbolic:
# _LABEL_USE + k means "where label k is", and a jump to _LABEL_DEF + k IS label k (it stands for
# the entry behind it).  _link() resolves both and drops the definitions.
_LABEL_USE = 4000
_LABEL_DEF = 7000


def script_value(type_mask: int) -> int:
    """Script entry of a value slot whose first token is restricted to ``type_mask`` (0 = any)."""
    return -2 - int(type_mask)


def script_jump(target: int) -> int:
    return JMP_BASE - int(target)


def script_choice(n: int) -> int:
    return CHOICE_BASE - int(n)


def entry_kind(e: int) -> str:
    if e >= 0:
        return "lit"
    if e == SCRIPT_END:
        return "end"
    if e > JMP_BASE:
        return "slot"
    if e > CHOICE_BASE:
        return "jmp"
    return "choice"


def resolve(script: Sequence[int], cur: int) -> int:
    """``cur`` with jumps followed (at most 8 hops: compiled scripts chain two at most)."""
    for _ in range(8):
        if not (0 <= cur < len(script)) or entry_kind(script[cur]) != "jmp":
            break
        cur = JMP_BASE - script[cur]
    return cur


def _forcing(s: int, depth: int) -> bool:
    """At the top level with a value just completed (or inside a literal): the script's next
    literal token (or choice) comes next."""
    return depth == 1 and s in (S_AFTER_VALUE, S_NUM, S_OBJ_NEXT)


def _value_min_cost(type_mask: int, vc: int) -> int:
    if type_mask == 0:
        return vc
    return 1 if type_mask & ((1 << C_DIGIT_NZ) | (1 << C_LIT) | (1 << C_NULL)) else 2  # a one-token number / literal, else "" / {} / []


def _instruction_starts(script: Sequence[int]) -> List[int]:
    """Indices of the entries that are instructions (the pairs behind a choice header are data)."""
    out, i = [], 0
    while i < len(script):
        out.append(i)
        e = script[i]
        i += 1 + (2 * (CHOICE_BASE - e) if entry_kind(e) == "choice" else 0)
    return out


def choice_alternatives(script: Sequence[int], cur: int) -> List[Tuple[int, int, int]]:
    """(key, target, index of the key entry) of every alternative of the choice at ``cur``."""
    n = CHOICE_BASE - script[cur]
    return [(script[cur + 1 + 2 * k], script[cur + 2 + 2 * k], cur + 1 + 2 * k) for k in range(n)]


def script_rest(script: Sequence[int], vc: int) -> List[int]:
    """``rest[i]`` = fewest tokens from script entry ``i`` to the end of the sequence (literal
    tokens + the shortest value of every slot + the cheapest alternative of every choice + the
    EOS token).  Scripts with loops (arrays without ``maxItems``) are relaxed to a fixed point."""
    n = len(script)
    rest = [_INF] * (n + 1)
    rest[n] = 1
    starts = _instruction_starts(script)
    for _ in range(len(starts) + 2):
        changed = False
        for i in reversed(starts):
            e, kind = script[i], entry_kind(script[i])
            if kind == "end":
                v = 1
            elif kind == "lit":
                v = 1 + rest[i + 1]
            elif kind == "slot":
                v = _value_min_cost(-2 - e, vc) + rest[i + 1]
            elif kind == "jmp":
                v = rest[JMP_BASE - e]
            else:
                v = min((1 if key >= 0 else _value_min_cost(-2 - key, vc)) + rest[tgt] for key, tgt, _ in choice_alternatives(script, i))
                for k in range(i + 1, i + 1 + 2 * (CHOICE_BASE - e)):
                    rest[k] = min(v, _INF)
            if v < rest[i]:
                rest[i] = v
                changed = True
        if not changed:
            break
    return [min(v, _INF) for v in rest[:n]]


def _set_after(script: Sequence[int], cur: int, left: int) -> State:
    """The automaton right after a literal (or a choice's token) was written, by what the script
    holds next: merged tokens such as '":' have no class the grammar could follow, so the state is
    SET — a value slot next, a literal / choice next (forcing), or the document is closed."""
    kind = entry_kind(script[cur]) if 0 <= cur < len(script) else "end"
    if kind == "end":
        return (S_DONE, 0, 0, left)
    if kind == "slot":
        return (S_VALUE, 1, 0, left)
    return (S_OBJ_NEXT, 1, 0, left)


def script_start(script: Sequence[int]) -> Tuple[int, int, int]:
    """(state, depth, stack) a scripted sequence starts in (cursor 0)."""
    return _set_after(script, resolve(script, 0), 0)[:3]


def _slot_classes(mask: int, after: int, left: int, vc: int) -> Tuple[int, int]:
    """(fits, grammar) class masks of the FIRST token of a value slot at the top level."""
    fits = grammar = 0
    for c in range(1, N_CLASSES):
        nxt = transition(S_VALUE, 1, 0, c)
        if nxt is None or nxt[1] < 1:
            continue
        grammar |= 1 << c
        if close_cost(nxt[0], nxt[1], vc) - 2 + after <= left - 1:
            fits |= 1 << c
    if mask & grammar:
        grammar &= mask
        fits &= mask
    return fits, grammar


def allowed_scripted(state: State, cursor: int, script: Optional[Sequence[int]], rest: Optional[Sequence[int]], vc: int) -> Tuple[int, List[int]]:
    """(class mask, token ids allowed whatever their class) for the next token of a scripted sequence."""
    s, depth, stack, left = state
    if not script or cursor < 0 or cursor >= len(script) or script[cursor] == SCRIPT_END or s in (S_OFF, S_END):
        return allowed_classes(state, vc), []
    e = script[cursor]
    kind = entry_kind(e)
    stack &= (1 << FIRST_SHIFT) - 1
    if kind != "slot" and _forcing(s, depth):
        more_digits = s == S_NUM and left - 1 >= rest[cursor]  # room for one more digit and all that has to follow
        digits = ((1 << C_DIGIT) | (1 << C_DIGIT_NZ)) if more_digits else 0
        if kind == "lit":
            return digits, [e]
        # a choice: every alternative after which the document can still be closed in the tokens left
        toks_fit, toks_all, cls_fit, cls_all = [], [], 0, 0
        for key, tgt, _ in choice_alternatives(script, cursor):
            if key >= 0:
                toks_all.append(key)
                if rest[tgt] <= left - 1:
                    toks_fit.append(key)
            else:
                f, g = _slot_classes(-2 - key, rest[tgt], left, vc)
                cls_fit |= f
                cls_all |= g
        if toks_fit or cls_fit:
            return cls_fit | digits, toks_fit
        return cls_all | digits, toks_all  # budget too small from the start: stay in the script, the output is cut short
    # inside (or at the start of) a value: the class automaton, with the budget of everything the
    # script still has to write after this value (close_cost counts "}" + EOS from the top level: 2)
    after = rest[cursor + 1] if kind == "slot" else rest[cursor]
    first = (-2 - e) if (kind == "slot" and s == S_VALUE and depth == 1) else 0
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
    return (fits or grammar), []


def step_scripted(state: State, cursor: int, tok: int, cls: np.ndarray, script: Optional[Sequence[int]]) -> Tuple[State, int, int]:
    """State, cursor and the index of the script entry the token stands for (-1: none — the token
    is part of a value) after token ``tok``."""
    s, depth, stack, left = state
    c = int(cls[tok]) if 0 <= tok < len(cls) else C_OTHER
    if not script or cursor < 0 or cursor >= len(script) or script[cursor] == SCRIPT_END or s in (S_OFF, S_END):
        return advance(state, c), cursor, -1
    e = script[cursor]
    kind = entry_kind(e)
    if kind == "lit" and _forcing(s, depth) and tok == e:
        nxt = resolve(script, cursor + 1)
        return _set_after(script, nxt, left - 1), nxt, cursor
    if kind == "choice" and _forcing(s, depth):
        alts = choice_alternatives(script, cursor)
        for key, tgt, at in alts:
            if key >= 0 and key == tok:
                nxt = resolve(script, tgt)
                return _set_after(script, nxt, left - 1), nxt, at
        if not (s == S_NUM and c in (C_DIGIT, C_DIGIT_NZ)):  # (a digit goes on with the number in front of the choice)
            for key, tgt, _at in alts:
                if key < 0:
                    mask = -2 - key
                    nx = transition(S_VALUE, 1, 0, c)
                    if nx is not None and nx[1] >= 1 and (mask == 0 or (mask >> c) & 1):
                        return (nx[0], nx[1], nx[2], left - 1), resolve(script, tgt), -1
    if kind == "slot" and s == S_VALUE and depth == 1:
        cursor = resolve(script, cursor + 1)  # the slot's value has begun
    return advance(state, c), cursor, -1


def advance_scripted(state: State, cursor: int, tok: int, cls: np.ndarray, script: Optional[Sequence[int]]) -> Tuple[State, int]:
    """State and cursor after token ``tok``."""
    st, cur, _ = step_scripted(state, cursor, tok, cls, script)
    return st, cur


class _Fragment:
    """Script entries being compiled, with the literal text that is still waiting to be encoded in
    front of the next slot / choice / the end (merging it lets a tokenizer use its longest pieces)."""

    def __init__(self, encode_pieces: Any):
        self.encode_pieces = encode_pieces
        self.script: List[int] = []
        self.texts: List[str] = []
        self.pending = ""

    def literal(self, text: str) -> None:
        for tid, piece in self.encode_pieces(text):
            self.script.append(int(tid))
            self.texts.append(piece)

    def flush(self) -> None:
        if self.pending:
            self.literal(self.pending)
            self.pending = ""

    def slot(self, mask: int) -> None:
        self.flush()
        self.script.append(script_value(mask))
        self.texts.append("")

    def raw(self, e: int, text: str = "") -> int:
        self.script.append(int(e))
        self.texts.append(text)
        return len(self.script) - 1

    def append_relocated(self, other: "_Fragment", skip: int = 0) -> None:
        """Entries ``skip``.. of another fragment, with its jump / choice targets moved to where they land here."""
        off = len(self.script) - skip
        starts = set(_instruction_starts(other.script))
        i = 0
        pair_targets: set = set()
        for st in sorted(starts):
            if entry_kind(other.script[st]) == "choice":
                n = CHOICE_BASE - other.script[st]
                pair_targets |= {st + 2 + 2 * k for k in range(n)}
        for i in range(skip, len(other.script)):
            e = other.script[i]
            if i in pair_targets:
                e = e + off if e < _LABEL_USE else e
            elif i in starts and entry_kind(e) == "jmp" and JMP_BASE - e < _LABEL_USE:
                e = script_jump(JMP_BASE - e + off)
            self.script.append(e)
            self.texts.append(other.texts[i])


def _emit_choice(frag: _Fragment, alts: List[_Fragment]) -> None:
    """A choice among compiled alternatives (each a fragment whose pending text has been flushed),
    told apart by their first entry: a literal token or a value slot.  Alternatives that share a
    first token are merged below it (a trie); alternatives that start with a slot are merged into
    one slot of all their classes when nothing follows them, else the first one is kept."""
    alts = [a for a in alts if a.script]
    groups: Dict[int, List[_Fragment]] = {}
    order: List[int] = []
    slot_alts: List[_Fragment] = []
    nested: List[_Fragment] = []  # alternatives that begin with a choice: their own alternatives join this one
    for a in alts:
        e = a.script[0]
        k = entry_kind(e)
        if k == "lit":
            if e not in groups:
                groups[e] = []
                order.append(e)
            groups[e].append(a)
        elif k == "slot":
            slot_alts.append(a)
        elif k == "choice":
            nested.append(a)
        # (an alternative that starts with a jump, or is empty, cannot be told apart here: dropped)
    keys: List[Tuple[int, List[_Fragment], str]] = [(tokid, groups[tokid], groups[tokid][0].texts[0]) for tokid in order]
    if slot_alts:
        if all(len(a.script) == 1 for a in slot_alts):
            mask = 0
            any_free = False
            for a in slot_alts:
                m = -2 - a.script[0]
                any_free |= m == 0
                mask |= m
            merged = _Fragment(frag.encode_pieces)
            merged.raw(script_value(0 if any_free else mask))
            slot_alts = [merged]
        keys.append((slot_alts[0].script[0], [slot_alts[0]], ""))
    keys = keys[:MAX_ALTS]
    if not keys and not nested:
        return
    if len(keys) == 1 and not nested:
        key, members, text = keys[0]
        if len(members) == 1:
            frag.append_relocated(members[0])  # no choice left: the one alternative, as it is
            return
        # every alternative starts with the same token (the opening quote of string values): write
        # it, then choose among what follows
        frag.raw(key, text)
        tails = []
        for m in members:
            t = _Fragment(frag.encode_pieces)
            t.append_relocated(m, skip=1)
            tails.append(t)
        if all(t.script for t in tails):
            _emit_choice(frag, tails)
        return
    taken = {key for key, _m, _t in keys}
    inner: List[List[Tuple[int, int, str]]] = []  # per nested alternative: (key, target inside it, text) of the alternatives it brings
    room = MAX_ALTS - len(keys)
    for a in nested:
        mine = []
        for key, tgt, at in choice_alternatives(a.script, 0):
            if key not in taken and room > 0:  # (a key that is already there keeps its first meaning)
                taken.add(key)
                mine.append((key, tgt, a.texts[at]))
                room -= 1
        inner.append(mine)
    header = frag.raw(script_choice(len(keys) + sum(len(m) for m in inner)))
    pair_at = []
    for key, _members, text in keys:
        pair_at.append(frag.raw(key, text))
        frag.raw(0)  # target, patched below
    inner_at = [[(frag.raw(key, text), frag.raw(0))[0] for key, _tgt, text in mine] for mine in inner]
    assert header >= 0
    joins: List[int] = []
    for (key, members, _text), at in zip(keys, pair_at):
        frag.script[at + 1] = len(frag.script)
        if len(members) == 1:
            frag.append_relocated(members[0], skip=1)
        else:  # several alternatives behind the same first token: choose among what follows it
            tails = []
            for m in members:
                t = _Fragment(frag.encode_pieces)
                t.append_relocated(m, skip=1)
                tails.append(t)
            if any(not t.script for t in tails):  # one value is a token-wise prefix of another: the shorter one is written
                pass
            else:
                _emit_choice(frag, tails)
        joins.append(frag.raw(0))  # jump to the join, patched below
    for a, mine, ats in zip(nested, inner, inner_at):
        if not mine:
            continue
        off = len(frag.script)
        frag.append_relocated(a)  # with its own header (not reached from here: its alternatives are entered directly)
        for (key, tgt, _text), at in zip(mine, ats):
            frag.script[at + 1] = tgt + off if tgt < _LABEL_USE else tgt
        joins.append(frag.raw(0))
    end = len(frag.script)
    for j in joins:
        frag.script[j] = script_jump(end)


def _link(script: List[int], texts: List[str]) -> Tuple[List[int], List[str]]:
    """The compiled script with its labels resolved: every symbolic target becomes the index of
    the entry behind the label's definition, and the definitions are dropped."""
    n = len(script)
    starts = _instruction_starts(script)
    is_def = [False] * n
    where: Dict[int, int] = {}
    pair_targets = set()
    for i in starts:
        e = script[i]
        kind = entry_kind(e)
        if kind == "jmp" and JMP_BASE - e >= _LABEL_DEF:
            is_def[i] = True
            where[JMP_BASE - e - _LABEL_DEF] = i
        elif kind == "choice":
            pair_targets |= {i + 2 + 2 * k for k in range(CHOICE_BASE - e)}
    new_index, k = [], 0
    for i in range(n):
        new_index.append(k)
        k += 0 if is_def[i] else 1
    new_index.append(k)

    def moved(t: int) -> int:
        if t >= _LABEL_USE:
            t = where[t - _LABEL_USE]
        return new_index[t]

    out_s, out_t = [], []
    for i in range(n):
        if is_def[i]:
            continue
        e = script[i]
        if i in pair_targets:
            e = moved(e)
        elif i in starts and entry_kind(e) == "jmp":
            e = script_jump(moved(JMP_BASE - e))
        out_s.append(e)
        out_t.append(texts[i])
    return out_s, out_t


def schema_script(schema: Any, encode_pieces: Any) -> Tuple[str, Tuple[int, int, int], Optional[List[int]], Optional[List[str]]]:
    """(text the server writes in front, automaton start, script, text of every script entry) for
    a JSON schema of an object.  ``encode_pieces(text) -> [(token id, text piece)]`` (no BOS; the
    pieces concatenate to ``text``).  The schema is compiled into ONE script of literals, value
    slots and choices, at any depth:

    * an object with ``properties`` writes every ``required`` property (all of them when nothing
      is required) in order, recursively;
    * ``const`` is written literally; ``enum`` is a CHOICE among its values (a trie over their
      tokens: the model picks, sampled or greedy, at the first token that tells them apart);
    * ``anyOf`` / ``oneOf``: a choice among the alternatives, told apart by their first token
      (scalars of different types merge into one slot of all their classes: ``Optional[str]`` is a
      string or ``null``); alternatives that cannot be told apart there fall back to the first;
    * an array whose ``items`` is a schema has between ``minItems`` and ``maxItems`` items: after
      the required ones the model chooses between "," + another item and "]" (a loop when
      ``maxItems`` is absent, unrolled otherwise), always within the token budget;
    * ``$ref`` (into ``$defs`` / ``definitions``) and ``allOf`` are resolved; a recursive schema
      stops at its first repeat: an array of such nodes without ``minItems`` is written empty, an
      alternative that repeats is dropped when another one remains, and what is left is free JSON
      of the node's type;
    * every other value is a slot restricted to the declared ``type`` — scalars, and objects /
      arrays that declare nothing about their content (free JSON of that container type).

    The automaton is SET at the end of every literal and a slot always runs at its own depth 1,
    so the brackets inside the literals need no tracking.  Without properties at the top: no
    script, the document is free JSON.  The texts are what a forced token stands for in the
    response (a tokenizer need not decode a token back to the piece it encoded)."""
    import json as _json

    from ..utils.jsonschema_lite import alternatives, deref

    top = deref(schema, schema)
    root: Optional[Dict[str, Any]] = None
    top_alts = alternatives(top, schema)
    if top_alts:
        # alternatives at the top: a choice when every one of them is an object with properties
        if all(isinstance(deref(a, schema).get("properties"), dict) and deref(a, schema)["properties"] for a in top_alts):
            root = top
        else:
            from ..utils.jsonschema_lite import preferred

            top = preferred(schema, schema)
    if root is None:
        props = top.get("properties")
        if not isinstance(props, dict) or not props:
            return "", (S_START, 0, 0), None, None
        root = {"type": "object", "properties": props, "required": top.get("required")}
    on_path: List[str] = []  # references being expanded: a recursive schema stops at its first repeat
    n_labels = [0]

    def new_label() -> int:
        n_labels[0] += 1
        if n_labels[0] >= _LABEL_DEF - _LABEL_USE:
            raise ValueError("schema too large for the guided-decoding script")
        return n_labels[0] - 1

    def compile_value(spec: Any, depth: int, frag: _Fragment) -> None:
        ref = spec.get("$ref") if isinstance(spec, dict) else None
        node = deref(spec, schema)
        if isinstance(ref, str):
            if ref in on_path:  # a node of its own kind below itself: free JSON of that type from here on
                node = {"type": node.get("type")} if "type" in node else {}
            on_path.append(ref)
            try:
                emit(node, depth, frag)
            finally:
                on_path.pop()
            return
        emit(node, depth, frag)

    def repeats(spec: Any) -> bool:
        return isinstance(spec, dict) and isinstance(spec.get("$ref"), str) and spec["$ref"] in on_path

    def emit(spec: Dict[str, Any], depth: int, frag: _Fragment) -> None:
        alts = alternatives(spec, schema)
        if alts and not all(repeats(a) for a in alts):
            alts = [a for a in alts if not repeats(a)]  # Optional[Node] below a Node: null, which is valid, not free JSON, which may not be
        if alts and depth < 16:
            own = {k: v for k, v in spec.items() if k not in ("anyOf", "oneOf")}
            frag.flush()
            compiled = []
            for a in alts:
                f = _Fragment(encode_pieces)
                compile_value({**own, **a}, depth + 1, f)
                f.flush()
                compiled.append(f)
            _emit_choice(frag, compiled)
            return
        if "const" in spec:
            frag.pending += _json.dumps(spec["const"], separators=(",", ":"))
            return
        if isinstance(spec.get("enum"), list) and spec["enum"]:
            values = list(dict.fromkeys(_json.dumps(v, separators=(",", ":")) for v in spec["enum"]))
            if len(values) == 1:
                frag.pending += values[0]
                return
            frag.flush()
            compiled = []
            for v in values[: 4 * MAX_ALTS]:
                f = _Fragment(encode_pieces)
                f.literal(v)
                compiled.append(f)
            _emit_choice(frag, compiled)
            return
        types = spec.get("type")
        tl = [types] if isinstance(types, str) else (list(types) if isinstance(types, list) else [])
        sub = spec.get("properties")
        if isinstance(sub, dict) and sub and (not tl or "object" in tl) and depth < 16:
            required = [k for k in (spec.get("required") or []) if k in sub]
            keys = required or list(sub)
            optional = [k for k in sub if k not in keys][:MAX_OPTIONAL]
            frag.pending += "{"
            for i, key in enumerate(keys):
                frag.pending += ("," if i else "") + _json.dumps(str(key)) + ":"
                compile_value(sub[key], depth + 1, frag)
            if not optional:
                frag.pending += "}"
                return
            # optional properties, in the schema's order: after the required ones the model chooses
            # between "}" and any optional key that has not been passed yet
            frag.flush()
            at = [new_label() for _ in range(len(optional) + 1)]  # at[i]: optional keys i.. may still come
            out = new_label()
            for i in range(len(optional)):
                frag.raw(script_jump(_LABEL_DEF + at[i]))
                alts = []
                for j in range(i, len(optional)):
                    f = _Fragment(encode_pieces)
                    f.pending = "," + _json.dumps(str(optional[j])) + ":"
                    compile_value(sub[optional[j]], depth + 1, f)
                    f.flush()
                    f.raw(script_jump(_LABEL_USE + at[j + 1]))
                    alts.append(f)
                close = _Fragment(encode_pieces)
                close.literal("}")
                close.raw(script_jump(_LABEL_USE + out))
                alts.append(close)
                _emit_choice(frag, alts)
            frag.raw(script_jump(_LABEL_DEF + at[-1]))
            frag.literal("}")
            frag.raw(script_jump(_LABEL_DEF + out))
            return
        items = spec.get("items")
        if isinstance(items, dict) and items and (not tl or "array" in tl) and depth < 16:
            lo = spec.get("minItems") if isinstance(spec.get("minItems"), int) else 0
            hi = spec.get("maxItems") if isinstance(spec.get("maxItems"), int) else None
            lo = max(lo, 0)
            if hi is not None:
                hi = max(hi, 0)
                lo = min(lo, hi)
            if repeats(items) and lo == 0:  # List[Node] below a Node: the empty list is valid, a free object may not be
                frag.pending += "[]"
                return
            frag.pending += "["
            if hi is not None and hi - lo <= 6:
                for i in range(lo):
                    frag.pending += "," if i else ""
                    compile_value(items, depth + 1, frag)
                if hi == lo:
                    frag.pending += "]"
                    return
                # optional items, unrolled from the inside out: after the last one only "]" is left
                frag.flush()

                def close_alt() -> _Fragment:
                    f = _Fragment(encode_pieces)
                    f.literal("]")
                    return f

                tail = close_alt()
                for k in range(hi - lo, 0, -1):
                    item = _Fragment(encode_pieces)
                    item.pending = "" if (lo == 0 and k == 1) else ","
                    compile_value(items, depth + 1, item)
                    item.flush()
                    item.append_relocated(tail)
                    nxt = _Fragment(encode_pieces)
                    _emit_choice(nxt, [close_alt(), item])
                    tail = nxt
                frag.append_relocated(tail)
                return
            # no upper bound (or a far one: the token budget is the bound): ONE copy of the item as a loop,
            #     [ (item ,){lo-1}  B: item  H: choice { "]" -> out ; "," -> B }
            # and without required items the array opens with  H0: choice { "]" -> out ; <what an item starts with> -> into B }
            for i in range(lo - 1):
                compile_value(items, depth + 1, frag)
                frag.pending += ","
            frag.flush()
            body, out = new_label(), new_label()
            item = _Fragment(encode_pieces)
            compile_value(items, depth + 1, item)
            item.flush()
            enter_at: List[Tuple[int, int]] = []  # (index of a target entry of the opening choice, target inside the item)
            if lo == 0:
                first = item.script[0]
                kind = entry_kind(first)
                if kind in ("lit", "slot"):
                    enter = [(first, 1, item.texts[0])]
                elif kind == "choice":
                    enter = [(key, tgt, item.texts[k_at]) for key, tgt, k_at in choice_alternatives(item.script, 0)][: MAX_ALTS - 1]
                else:
                    raise ValueError("array items that begin with nothing the model could choose by")
                h0 = len(frag.script)
                _emit_keyed_choice(frag, [("]", _LABEL_USE + out)], [(key, 0, text) for key, _tgt, text in enter])
                enter_at = [(h0 + 4 + 2 * k, tgt) for k, (_key, tgt, _text) in enumerate(enter)]
            frag.raw(script_jump(_LABEL_DEF + body))
            b0 = len(frag.script)
            frag.append_relocated(item)
            for at, tgt in enter_at:
                frag.script[at] = tgt + b0 if tgt < _LABEL_USE else tgt
            _emit_keyed_choice(frag, [("]", _LABEL_USE + out), (",", _LABEL_USE + body)])
            frag.raw(script_jump(_LABEL_DEF + out))
            return
        mask = 0
        for t in tl:
            mask |= TYPE_CLASSES.get(t, 0)
        frag.slot(mask)

    def _emit_keyed_choice(frag: _Fragment, texts_to: List[Tuple[str, int]], more: Sequence[Tuple[int, int, str]] = ()) -> None:
        """A choice whose alternatives are short literals (``text -> target``; a text of several
        tokens continues through a stub) plus ready (key, target, text) pairs."""
        stubs = []
        frag.raw(script_choice(len(texts_to) + len(more)))
        for text, target in texts_to:
            pieces = encode_pieces(text)
            frag.raw(int(pieces[0][0]), pieces[0][1])
            t_at = frag.raw(target)
            if len(pieces) > 1:
                stubs.append((t_at, pieces[1:], target))
        for key, target, text in more:
            frag.raw(key, text)
            frag.raw(target)
        for t_at, pieces, target in stubs:
            frag.script[t_at] = len(frag.script)
            for tid, piece in pieces:
                frag.raw(int(tid), piece)
            frag.raw(script_jump(target))

    main = _Fragment(encode_pieces)
    compile_value(root, 0, main)
    # the first literal goes into the prompt: the model sees it, nothing to force
    prefix = None
    main.flush()
    main.raw(SCRIPT_END)
    script, texts = main.script, main.texts
    # leading literal tokens become the prompt prefix (only when something other than a literal follows them)
    lead = 0
    while lead < len(script) and entry_kind(script[lead]) == "lit":
        lead += 1
    if lead and entry_kind(script[lead]) != "end":
        prefix = "".join(texts[:lead])
        rest_frag = _Fragment(encode_pieces)
        rest_frag.append_relocated(main, skip=lead)
        script, texts = rest_frag.script, rest_frag.texts
    script, texts = _link(script, texts)
    if len(script) > SCRIPT_MAX:
        raise ValueError(f"schema skeleton of {len(script)} script entries exceeds the {SCRIPT_MAX} the guided-decoding script holds")
    if prefix is None:  # every value was a constant: the whole document is one literal
        return "", script_start(script), script, texts
    state_after_text(prefix)  # raises on a prefix that is not JSON
    return prefix, script_start(script), script, texts


class ScriptedText:
    """Text of a scripted sequence, token by token: a forced skeleton token stands for the piece
    of the schema's literal it was encoded from, every other token for what it decodes to (the
    host-side twin of the device automaton; also what a response is rebuilt from)."""

    def __init__(self, cls: np.ndarray, start: Tuple[int, int, int], max_tokens: int, script: Sequence[int], texts: Sequence[str], decode_token: Any):
        self.cls, self.script, self.texts, self.decode_token = cls, list(script), list(texts), decode_token
        self.state: State = (start[0], start[1], start[2], max_tokens)
        self.cursor = 0

    def piece(self, tok: int) -> str:
        self.state, self.cursor, at = step_scripted(self.state, self.cursor, int(tok), self.cls, self.script)
        if at >= 0:  # a skeleton token (a literal, or the token that picked an alternative): the schema's own text
            return self.texts[at]
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
        allow, toks = allowed_scripted(st, cur, ent, rest, vc)  # type: ignore[arg-type]
        if allow == (1 << N_CLASSES) - 1:
            continue
        ok = torch.tensor([(allow >> c) & 1 for c in range(N_CLASSES)], dtype=torch.bool, device=logits.device)[cls_t]
        for forced in toks:
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
