"""Guided (JSON / tool-call) decoding: the automaton, the engine path, the server surface.

Reference behaviour: ``scripts/openai_parity_probe.py:71-186`` (tool_calls list, >= 2 parallel
calls, ``json.loads`` of a json_object response), ``runners/profiles/structured-output.yaml:17``,
``runners/profiles/tool-calling.yaml:18-19``.
"""

import json
import random

import numpy as np
import pytest

from kvmini.engine import guided as G


def _text_of(cls_seq):
    piece = {G.C_LBRACE: "{", G.C_RBRACE: "}", G.C_LBRACK: "[", G.C_RBRACK: "]", G.C_COLON: ":", G.C_COMMA: ",", G.C_QUOTE: '"', G.C_STR: " w", G.C_DIGIT_NZ: "7",
             G.C_DIGIT: "0", G.C_LIT: "true", G.C_NULL: "null", G.C_EOS: ""}
    return "".join(piece[c] for c in cls_seq)


@pytest.mark.parametrize("vc", [1, 2])
def test_random_walks_inside_the_allowed_classes_are_complete_json_within_the_budget(vc):
    rng = random.Random(7)
    for trial in range(400):
        budget = rng.randint(3, 60)
        st = (G.S_START, 0, 0, budget)
        seq = []
        while st[0] != G.S_END:
            allow = G.allowed_classes(st, vc)
            options = [c for c in range(1, G.N_CLASSES) if (allow >> c) & 1 and not (vc == 2 and c in (G.C_LIT, G.C_NULL, G.C_DIGIT_NZ) and st[0] in (G.S_VALUE, G.S_ARR_FIRST))]
            assert options, (st, seq)
            c = rng.choice(options)
            seq.append(c)
            st = G.advance(st, c)
            assert len(seq) <= budget, (budget, seq)
        assert seq[-1] == G.C_EOS
        doc = json.loads(_text_of(seq))
        assert isinstance(doc, dict)


def test_prefix_state_type_restriction_and_replay():
    st = G.state_after_text('{"age":')
    assert st == (G.S_VALUE, 1, 0)
    first = G.TYPE_CLASSES["integer"]
    allow = G.allowed_classes((st[0], st[1], st[2] | (first << G.FIRST_SHIFT), 20), 1)
    assert allow == 1 << G.C_DIGIT_NZ
    nxt = G.advance((st[0], st[1], st[2] | (first << G.FIRST_SHIFT), 20), G.C_DIGIT_NZ)
    assert nxt == (G.S_NUM, 1, 0, 19)  # the restriction is gone after the first token
    assert G.state_after_text('{"a":[1,{"b":"x') == (G.S_STR, 3, 0b010)
    with pytest.raises(ValueError):
        G.state_after_text('{"a"}')
    cls = np.array([G.C_OTHER, G.C_LBRACE, G.C_QUOTE, G.C_STR, G.C_COLON, G.C_LIT, G.C_RBRACE, G.C_EOS], dtype=np.uint8)
    assert G.replay(cls, (G.S_START, 0, 0), 30, [1, 2, 3, 2, 4, 5]) == ((G.S_AFTER_VALUE, 1, 0, 24), -1)


SCHEMA = {"type": "object", "required": ["name", "age", "tags", "ok", "kind", "address"],
          "properties": {"name": {"type": "string"}, "age": {"type": "integer"}, "tags": {"type": "array"}, "ok": {"type": "boolean"},
                         "kind": {"enum": ["a b", "c"]}, "address": {"type": "object"}, "nickname": {"type": "string"}}}
PY_TYPES = {"name": str, "age": int, "tags": list, "ok": bool, "kind": str, "address": dict}


def _check_schema_doc(doc):
    assert list(doc)[: len(PY_TYPES)] == list(PY_TYPES), doc  # every required key, in order
    for k, t in PY_TYPES.items():
        assert type(doc[k]) is t, (k, doc[k])
    assert doc["kind"] in ("a b", "c") and isinstance(doc.get("nickname", ""), str) and list(doc)[:6] == SCHEMA["required"]


def _options(allow, toks, by_class, rng):
    return [rng.choice(by_class[c]) for c in range(1, G.N_CLASSES) if (allow >> c) & 1 and by_class[c]] + list(toks)


def test_scripted_random_walks_give_every_required_key_with_its_type_within_the_budget():
    """Token-level random walks through the scripted automaton (forced skeleton tokens + class
    automaton inside the value slots): whatever the model picks, the document parses, has every
    required key in order with a value of the declared type, and ends within max_tokens once the
    shortest document fits."""
    from kvmini.engine.tokenizer import get_tokenizer

    tok = get_tokenizer(2048, 1, 2)
    cls = G.token_classes(tok, 2048, 2)
    vc = G.value_cost(cls)
    prefix, start, script, texts = G.schema_script(SCHEMA, tok.encode_pieces)
    assert prefix == '{"name":' and start == (G.S_VALUE, 1, 0) and script[-1] == G.SCRIPT_END and len(texts) == len(script)
    assert script[0] == G.script_value(G.TYPE_CLASSES["string"])
    assert sum(1 for i in G._instruction_starts(script) if G.entry_kind(script[i]) == "choice") == 2  # the enum ('"', then 'a b"' or 'c"') and the optional nickname (',"nickname":' or '}')
    rest = G.script_rest(script, vc)
    shortest = '{"name":"","age":7,"tags":[],"ok":true,"kind":"c","address":{}}'
    assert rest[0] == len(tok.encode_pieces(shortest[len(prefix):])) + 1  # the cheapest document behind the prefix + EOS
    kinds, nick = set(), set()
    by_class = {c: [t for t in range(16, 2048) if cls[t] == c] for c in range(G.N_CLASSES)}
    by_class[G.C_EOS] = [2]
    rng = random.Random(11)
    for trial in range(300):
        budget = rng.randint(rest[0], rest[0] + 60)
        st, cur, out = (start[0], start[1], start[2], budget), 0, []
        while st[0] != G.S_END:
            allow, toks = G.allowed_scripted(st, cur, script, rest, vc)
            options = _options(allow, toks, by_class, rng)
            assert options, (st, cur, out)
            t = rng.choice(options)
            out.append(t)
            st, cur = G.advance_scripted(st, cur, t, cls, script)
            assert len(out) <= budget, (budget, tok.decode(out))
        assert out[-1] == 2
        text = G.ScriptedText(cls, start, budget, script, texts, tok.decode_token)
        doc = json.loads(prefix + "".join(text.piece(t) for t in out))
        _check_schema_doc(doc)
        kinds.add(doc["kind"])
        nick.add("nickname" in doc)
        # replay (a preempted sequence) reproduces state and cursor at every cut
        cut = rng.randrange(len(out))
        st2, cur2 = (start[0], start[1], start[2], budget), 0
        for t in out[:cut]:
            st2, cur2 = G.advance_scripted(st2, cur2, t, cls, script)
        assert G.replay(cls, start, budget, out[:cut], script) == (st2, cur2)
    assert kinds == {"a b", "c"} and nick == {False, True}  # the model's pick, not the schema's first entry; optional keys come and go
    # a schema of constants only is one forced literal; a schema without properties has no script
    p2, s2, sc2, tx2 = G.schema_script({"properties": {"v": {"const": 3}, "w": {"enum": ["x"]}}}, tok.encode_pieces)
    assert p2 == "" and s2 == (G.S_OBJ_NEXT, 1, 0) and "".join(tx2) == '{"v":3,"w":"x"}' and sc2[-1] == G.SCRIPT_END
    assert G.schema_script({"type": "object"}, tok.encode_pieces) == ("", (G.S_START, 0, 0), None, None)


NESTED = {"type": "object", "required": ["user", "items", "total"],
          "properties": {
              "user": {"type": "object", "required": ["id", "name", "geo"],
                       "properties": {"id": {"type": "integer"}, "name": {"type": "string"}, "email": {"type": "string"},
                                      "geo": {"type": "object", "properties": {"lat": {"type": "number"}, "lon": {"type": "number"}}}}},
              "items": {"type": "array", "minItems": 2,
                        "items": {"type": "object", "required": ["sku", "qty", "flags"],
                                  "properties": {"sku": {"type": "string"}, "qty": {"type": "integer"},
                                                 "flags": {"type": "array", "items": {"type": "boolean"}}, "note": {"type": "string"}}}},
              "total": {"type": ["number", "null"]},
              "comment": {"type": "string"}}}


def test_scripted_random_walks_through_a_nested_schema():
    """Nested objects and arrays of objects are flattened into the same line of literals and value
    slots: whatever the model picks inside the slots, the document satisfies the schema at every
    depth (checked by the load generator's schema checker), arrays have at least ``minItems``
    items, and optional properties are left out."""
    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.loadgen.schema_check import violation

    tok = get_tokenizer(2048, 1, 2)
    cls = G.token_classes(tok, 2048, 2)
    vc = G.value_cost(cls)
    prefix, start, script, texts = G.schema_script(NESTED, tok.encode_pieces)
    assert prefix == '{"user":{"id":' and start == (G.S_VALUE, 1, 0)
    T = G.TYPE_CLASSES
    rest = G.script_rest(script, vc)
    n_items, n_flags = set(), set()
    by_class = {c: [t for t in range(16, 2048) if cls[t] == c] for c in range(G.N_CLASSES)}
    by_class[G.C_EOS] = [2]
    rng = random.Random(5)
    for trial in range(200):
        budget = rng.randint(rest[0], rest[0] + 80)
        st, cur, out = (start[0], start[1], start[2], budget), 0, []
        while st[0] != G.S_END:
            allow, toks = G.allowed_scripted(st, cur, script, rest, vc)
            options = _options(allow, toks, by_class, rng)
            assert options, (st, cur, out)
            t = rng.choice(options)
            out.append(t)
            st, cur = G.advance_scripted(st, cur, t, cls, script)
            assert len(out) <= budget
        text = G.ScriptedText(cls, start, budget, script, texts, tok.decode_token)
        doc = json.loads(prefix + "".join(text.piece(t) for t in out))
        assert violation(doc, NESTED) is None, (violation(doc, NESTED), doc)
        assert list(doc)[:3] == ["user", "items", "total"] and list(doc["user"])[:3] == ["id", "name", "geo"] and list(doc["user"]["geo"]) == ["lat", "lon"]
        assert len(doc["items"]) >= 2 and all(list(it)[:3] == ["sku", "qty", "flags"] for it in doc["items"])
        n_items.add(len(doc["items"]))
        n_flags |= {len(it["flags"]) for it in doc["items"]}
    assert 2 in n_items and max(n_items) > 2 and 0 in n_flags and max(n_flags) >= 2  # minItems is the floor, not the count; arrays without bounds may be empty
    # what pydantic emits: nested models behind $ref, Optional[X] as anyOf [X, null]; a recursive model stops at its first repeat
    pyd = {"$defs": {"Geo": {"type": "object", "properties": {"lat": {"type": "number"}, "lon": {"type": "number"}}, "required": ["lat", "lon"]},
                     "Node": {"type": "object", "properties": {"v": {"type": "integer"}, "kids": {"type": "array", "items": {"$ref": "#/$defs/Node"}}}, "required": ["v", "kids"]}},
           "type": "object", "required": ["id", "geo", "note", "tree"],
           "properties": {"id": {"type": "integer"}, "geo": {"allOf": [{"$ref": "#/$defs/Geo"}]}, "note": {"anyOf": [{"type": "string"}, {"type": "null"}], "default": None},
                          "tree": {"$ref": "#/$defs/Node"}}}
    p4, st4, sc4, tx4 = G.schema_script(pyd, tok.encode_pieces)
    assert p4 == '{"id":'
    slots4 = [-2 - sc4[i] for i in G._instruction_starts(sc4) if G.entry_kind(sc4[i]) == "slot"]
    assert slots4[:4] == [T["integer"], T["number"], T["number"], T["string"] | T["null"]]  # Optional[str]: a string or null, the model's pick
    rest4 = G.script_rest(sc4, vc)
    notes = set()
    for trial in range(120):
        budget = rng.randint(rest4[0], rest4[0] + 60)
        st, cur, out = (st4[0], st4[1], st4[2], budget), 0, []
        while st[0] != G.S_END:
            allow, toks = G.allowed_scripted(st, cur, sc4, rest4, vc)
            t = rng.choice(_options(allow, toks, by_class, rng))
            out.append(t)
            st, cur = G.advance_scripted(st, cur, t, cls, sc4)
            assert len(out) <= budget
        text = G.ScriptedText(cls, st4, budget, sc4, tx4, tok.decode_token)
        doc = json.loads(p4 + "".join(text.piece(t) for t in out))
        assert violation(doc, pyd) is None, (violation(doc, pyd), doc)
        notes.add(type(doc["note"]).__name__)
    assert notes == {"str", "NoneType"}
    assert violation({"id": 1, "geo": {"lat": 1.5, "lon": 2}, "note": None, "tree": {"v": 1, "kids": [{"v": 2, "kids": []}]}}, pyd) is None
    assert "geo.lat" in violation({"id": 1, "geo": {"lat": "x", "lon": 2}, "note": "n", "tree": {"v": 1, "kids": []}}, pyd)
    assert "kids[0].v" in violation({"id": 1, "geo": {"lat": 1, "lon": 2}, "note": "n", "tree": {"v": 1, "kids": [{"v": "x", "kids": []}]}}, pyd)
    assert "anyOf" in violation({"id": 1, "geo": {"lat": 1, "lon": 2}, "note": 3, "tree": {"v": 1, "kids": []}}, pyd)
    # maxItems 0 writes an empty array; an array or object that says nothing about its content stays a free slot
    p3, _, sc3, tx3 = G.schema_script({"properties": {"a": {"type": "array", "maxItems": 0, "items": {"type": "integer"}}, "b": {"type": "array"},
                                                      "c": {"type": "object"}}}, tok.encode_pieces)
    assert p3 == '{"a":[],"b":' and "".join(tx3) == ',"c":}' and [-2 - e for e in sc3 if e <= -2] == [T["array"], T["object"]]


def _walks(schema, trials, seed, vocab=2048):
    """Random walks through the script of ``schema``: the documents (parsed), each within its budget."""
    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.loadgen.schema_check import violation

    tok = get_tokenizer(vocab, 1, 2)
    cls = G.token_classes(tok, vocab, 2)
    vc = G.value_cost(cls)
    prefix, start, script, texts = G.schema_script(schema, tok.encode_pieces)
    assert script is not None and len(script) <= G.SCRIPT_MAX and not any(G.entry_kind(script[i]) == "jmp" and G.JMP_BASE - script[i] >= 4000 for i in G._instruction_starts(script))
    assert G.entry_kind(script[0]) != "jmp"  # the cursor starts on entry 0
    rest = G.script_rest(script, vc)
    by_class = {c: [t for t in range(16, vocab) if cls[t] == c] for c in range(G.N_CLASSES)}
    by_class[G.C_EOS] = [2]
    rng = random.Random(seed)
    docs = []
    for trial in range(trials):
        budget = rest[0] if trial == 0 else rng.randint(rest[0], rest[0] + 60)
        st, cur, out = (start[0], start[1], start[2], budget), 0, []
        while st[0] != G.S_END:
            allow, toks = G.allowed_scripted(st, cur, script, rest, vc)
            options = _options(allow, toks, by_class, rng)
            assert options, (st, cur, out)
            t = rng.choice(options)
            out.append(t)
            st, cur = G.advance_scripted(st, cur, t, cls, script)
            assert len(out) <= budget
        text = G.ScriptedText(cls, start, budget, script, texts, tok.decode_token)
        doc = json.loads(prefix + "".join(text.piece(t) for t in out))
        assert violation(doc, schema) is None, (violation(doc, schema), doc)
        docs.append(doc)
    return docs


def test_scripted_optional_keys_choice_first_items_and_top_level_alternatives():
    """Optional properties come and go in the schema's order; an array of enum values (items that
    begin with a choice of their own) loops through ONE copy of the item; alternatives at the top
    of a schema (what a tool-choosing request compiles to) are told apart below their shared text."""
    opt = {"type": "object", "required": ["a"], "properties": {"a": {"type": "integer"}, "b": {"type": "boolean"}, "c": {"enum": ["x", "y"]}, "d": {"type": "string"}, "e": {"type": "null"}}}
    seen = set()
    for doc in _walks(opt, 300, 5):
        keys = list(doc)
        assert keys[0] == "a" and keys == [k for k in "abcd" if k in doc]  # schema order; the fifth property is beyond MAX_OPTIONAL
        seen.add(tuple(keys))
    assert len(seen) == 8, seen
    arr = {"type": "object", "required": ["v", "w"], "properties": {"v": {"type": "array", "items": {"enum": [1, 2, "z"]}},
                                                                    "w": {"type": "array", "minItems": 2, "items": {"type": "integer"}}}}
    lens, vals = set(), set()
    for doc in _walks(arr, 300, 6):
        lens.add(min(len(doc["v"]), 3))
        vals |= set(doc["v"])
        assert len(doc["w"]) >= 2
    assert lens == {0, 1, 2, 3} and vals == {1, 2, "z"}
    from kvmini.engine.tokenizer import get_tokenizer

    tok = get_tokenizer(2048, 1, 2)
    _p, _s, script, _t = G.schema_script(arr, tok.encode_pieces)
    assert sum(1 for e in script if e == G.script_value(G.TYPE_CLASSES["integer"])) == 2  # minItems 2 = one inline item + the loop's one
    tools = {"oneOf": [{"type": "object", "required": ["tool_calls"], "properties": {"tool_calls": {"type": "array", "minItems": 1, "items": {"oneOf": [
        {"type": "object", "required": ["name", "arguments"], "properties": {"name": {"const": "get_weather"}, "arguments": {"type": "object", "required": ["location"], "properties": {
            "location": {"type": "string"}, "unit": {"enum": ["celsius", "fahrenheit"]}}}}},
        {"type": "object", "required": ["name", "arguments"], "properties": {"name": {"const": "get_time"}, "arguments": {"type": "object", "properties": {"city": {"type": "string"}}}}}]}}}},
        {"type": "object", "required": ["content"], "properties": {"content": {"type": "string"}}}]}
    kinds, names, units, n_calls = set(), set(), set(), set()
    for doc in _walks(tools, 400, 7):
        kinds.add(list(doc)[0])
        for c in doc.get("tool_calls", []):
            names.add(c["name"])
            units.add(c["arguments"].get("unit"))
        n_calls.add(min(len(doc.get("tool_calls", [])), 3))
    assert kinds == {"tool_calls", "content"} and names == {"get_weather", "get_time"} and units == {None, "celsius", "fahrenheit"} and n_calls >= {0, 1, 2}


CHOICES = {"type": "object", "required": ["location", "unit", "days", "alerts", "extra"],
           "properties": {"location": {"type": "string"}, "unit": {"type": "string", "enum": ["celsius", "fahrenheit"]},
                          "days": {"type": "array", "items": {"type": "integer"}, "minItems": 1, "maxItems": 3},
                          "alerts": {"type": "array", "items": {"type": "object", "required": ["level"], "properties": {"level": {"enum": ["low", "high", 3]}}}},
                          "extra": {"oneOf": [{"type": "null"}, {"type": "object", "required": ["k"], "properties": {"k": {"type": "boolean"}}}, {"type": "integer"}]}}}


def test_scripted_choices_reach_every_enum_value_every_alternative_and_both_array_bounds():
    """enum values (the reference's tool profile: ``runners/profiles/tool-calling.yaml:33``), anyOf /
    oneOf alternatives and array lengths (``structured-output.yaml:29``) are the MODEL's choice:
    random walks reach every enum value, every alternative, both bounds of a bounded array and
    0 / 1 / many items of an unbounded one; every document validates; the budget is never exceeded."""
    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.loadgen.schema_check import violation

    tok = get_tokenizer(2048, 1, 2)
    cls = G.token_classes(tok, 2048, 2)
    vc = G.value_cost(cls)
    prefix, start, script, texts = G.schema_script(CHOICES, tok.encode_pieces)
    assert prefix == '{"location":' and len(script) <= G.SCRIPT_MAX
    rest = G.script_rest(script, vc)
    by_class = {c: [t for t in range(16, 2048) if cls[t] == c] for c in range(G.N_CLASSES)}
    by_class[G.C_EOS] = [2]
    rng = random.Random(17)
    units, days, alerts, levels, extras = set(), set(), set(), set(), set()
    for trial in range(500):
        budget = rng.randint(rest[0], rest[0] + 70)
        st, cur, out = (start[0], start[1], start[2], budget), 0, []
        while st[0] != G.S_END:
            allow, toks = G.allowed_scripted(st, cur, script, rest, vc)
            options = _options(allow, toks, by_class, rng)
            assert options, (st, cur, out)
            t = rng.choice(options)
            out.append(t)
            st, cur = G.advance_scripted(st, cur, t, cls, script)
            assert len(out) <= budget
        text = G.ScriptedText(cls, start, budget, script, texts, tok.decode_token)
        doc = json.loads(prefix + "".join(text.piece(t) for t in out))
        assert violation(doc, CHOICES) is None, (violation(doc, CHOICES), doc)
        units.add(doc["unit"]); days.add(len(doc["days"])); alerts.add(min(len(doc["alerts"]), 3)); extras.add(type(doc["extra"]).__name__)
        levels |= {json.dumps(a["level"]) for a in doc["alerts"]}
        cut = rng.randrange(len(out))  # replay (a preempted sequence) reproduces state and cursor at every cut
        st2, cur2 = (start[0], start[1], start[2], budget), 0
        for t in out[:cut]:
            st2, cur2 = G.advance_scripted(st2, cur2, t, cls, script)
        assert G.replay(cls, start, budget, out[:cut], script) == (st2, cur2)
    assert units == {"celsius", "fahrenheit"} and days == {1, 2, 3} and alerts == {0, 1, 2, 3}
    assert levels == {'"low"', '"high"', "3"} and extras == {"NoneType", "dict", "int"}
    # the smallest budget still closes the document
    st, cur, n = (start[0], start[1], start[2], rest[0]), 0, 0
    while st[0] != G.S_END:
        allow, toks = G.allowed_scripted(st, cur, script, rest, vc)
        st, cur = G.advance_scripted(st, cur, rng.choice(_options(allow, toks, by_class, rng)), cls, script)
        n += 1
    assert n == rest[0]


def test_token_classes_of_the_synthetic_tokenizer():
    from kvmini.engine.tokenizer import get_tokenizer

    tok = get_tokenizer(2048, 1, 2)
    cls = G.token_classes(tok, 2048, 2)
    assert cls[2] == G.C_EOS and cls[0] == G.C_OTHER
    for piece, c in (("{", G.C_LBRACE), ('"', G.C_QUOTE), ("true", G.C_LIT), ("7", G.C_DIGIT_NZ), ("0", G.C_DIGIT)):
        (tid,) = tok.encode(piece)
        assert cls[tid] == c and tok.decode_token(tid) == piece
    assert G.value_cost(cls) == 1
    assert (cls == G.C_STR).sum() > 1900


def _engine(**kw):
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.tokenizer import get_tokenizer

    eng = LLMEngine(EngineConfig(model="tiny", device="cpu", max_num_seqs=8, max_batched_tokens=64, max_model_len=512, dtype="fp32", **kw))
    tok = get_tokenizer(eng.mcfg.vocab_size, eng.mcfg.bos_token_id, eng.mcfg.eos_token_id)
    eng.set_token_classes(G.token_classes(tok, eng.mcfg.vocab_size, eng.mcfg.eos_token_id))
    return eng, tok


@pytest.mark.parametrize("async_scheduling", [True, False])
def test_engine_guided_generation_is_valid_json_greedy_and_sampled(async_scheduling):
    from kvmini.engine.scheduler import SamplingParams

    eng, tok = _engine(async_scheduling=async_scheduling)
    prompts = [tok.encode(f"request number {i}", add_bos=True) for i in range(5)]
    for mt, temp in ((3, 0.0), (9, 0.0), (30, 0.0), (30, 1.0), (64, 0.7)):
        outs = eng.generate(prompts, SamplingParams(max_tokens=mt, temperature=temp, guided="json"))
        for o in outs:
            assert len(o) <= mt and o[-1] == eng.mcfg.eos_token_id
            assert isinstance(json.loads(tok.decode(o)), dict)
    if async_scheduling:
        assert eng.pipelined_steps > 0  # guided requests are stepped launch-ahead: the automaton is not on the host
    # a guided and a plain request side by side: the plain one is untouched by the mask
    plain = eng.generate([prompts[0]], SamplingParams(max_tokens=12))[0]
    seqs = [eng.make_sequence(prompts[0], SamplingParams(max_tokens=12)), eng.make_sequence(prompts[1], SamplingParams(max_tokens=12, guided="json"))]
    for s in seqs:
        assert eng.scheduler.add(s) is None
    while eng.scheduler.has_work():
        eng.step()
    assert seqs[0].output_ids == plain
    json.loads(tok.decode(seqs[1].output_ids))


def test_native_server_json_mode_schema_tools_and_parity_matrix(in_tmp):
    import httpx

    from kvmini.harness import parity_probe
    from kvmini.server.native import start_native_in_thread

    eng, _tok = _engine()
    _ns, url, stop = start_native_in_thread(eng)
    try:
        c = httpx.Client(timeout=60)
        msg = [{"role": "user", "content": "Return a JSON with keys city and temp."}]
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "response_format": {"type": "json_object"}})
        assert r.status_code == 200 and r.json()["choices"][0]["finish_reason"] == "stop"
        assert isinstance(json.loads(r.json()["choices"][0]["message"]["content"]), dict)
        schema = {"type": "object", "properties": {"name": {"type": "string"}, "age": {"type": "integer", "minimum": 0}}, "required": ["age", "name"]}
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "max_tokens": 32, "temperature": 0.9, "n": 2,
                                                       "response_format": {"type": "json_schema", "json_schema": {"name": "r", "schema": schema}}})
        assert r.status_code == 200 and len(r.json()["choices"]) == 2
        for ch in r.json()["choices"]:
            doc = json.loads(ch["message"]["content"])
            # every required key in the order of `required`, each value of the declared type
            assert list(doc) == ["age", "name"] and type(doc["age"]) is int and type(doc["name"]) is str and ch["finish_reason"] == "stop"
        # a richer schema (string, integer, array, boolean, enum, object; one optional property), streamed and not
        rf = {"type": "json_schema", "json_schema": {"name": "r", "schema": SCHEMA}}
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "max_tokens": 96, "temperature": 1.0, "response_format": rf})
        _check_schema_doc(json.loads(r.json()["choices"][0]["message"]["content"]))
        txt = ""
        with c.stream("POST", url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "max_tokens": 96, "response_format": rf, "stream": True}) as rs:
            for line in rs.iter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    d = json.loads(line[6:])["choices"]
                    txt += (d[0]["delta"].get("content") or "") if d else ""
        _check_schema_doc(json.loads(txt))
        # nested: objects in objects, an array of objects, streamed and not — the schema holds at every depth
        from kvmini.loadgen.schema_check import violation

        rf = {"type": "json_schema", "json_schema": {"name": "order", "schema": NESTED}}
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "max_tokens": 200, "temperature": 0.8, "response_format": rf})
        assert r.status_code == 200 and r.json()["choices"][0]["finish_reason"] == "stop"
        doc = json.loads(r.json()["choices"][0]["message"]["content"])
        assert violation(doc, NESTED) is None and len(doc["items"]) >= 2, doc
        txt = ""
        with c.stream("POST", url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "max_tokens": 200, "response_format": rf, "stream": True}) as rs:
            for line in rs.iter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    d = json.loads(line[6:])["choices"]
                    txt += (d[0]["delta"].get("content") or "") if d else ""
        assert violation(json.loads(txt), NESTED) is None
        # the load generator checks every answer of a schema-carrying run and the analyzer reports the rate
        from kvmini.analysis.reduce import analyze_run
        from kvmini.cluster.backends import FakeBackend
        from kvmini.loadgen.protocols import RequestSpec
        from kvmini.loadgen.runner import LoadConfig, run_load_sync

        for streaming in (True, False):
            res = run_load_sync(LoadConfig(url=url, run_dir="so_run", requests=6, concurrency=3, quiet=True, spec=RequestSpec(model="tiny", max_tokens=96, stream=streaming, response_schema=SCHEMA)))
            assert all(r.status == 200 and r.json_ok and r.schema_ok for r in res), [(r.status, r.schema_why) for r in res]
        so = analyze_run("so_run", backend=FakeBackend(), quiet=True)["structured_output"]
        assert so["checked"] == so["valid_json"] == so["schema_valid"] == 6 and so["schema_valid_rate"] == 1.0 and so["first_failures"] == []
        # vLLM's guided_json spelling on the completions endpoint
        r = c.post(url + "/v1/completions", json={"model": "tiny", "prompt": "person:", "max_tokens": 96, "guided_json": SCHEMA})
        _check_schema_doc(json.loads(r.json()["choices"][0]["text"]))
        # tool arguments carry every required parameter
        two = [{"type": "function", "function": {"name": "forecast", "parameters": {"type": "object", "required": ["city", "days"],
                                                                                  "properties": {"city": {"type": "string"}, "days": {"type": "integer"}, "units": {"enum": ["c", "f"]}}}}}]
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": two, "tool_choice": "required", "max_tokens": 48})
        a = json.loads(r.json()["choices"][0]["message"]["tool_calls"][0]["function"]["arguments"])
        assert list(a)[:2] == ["city", "days"] and type(a["city"]) is str and type(a["days"]) is int and a.get("units", "c") in ("c", "f")
        # tool_choice auto: the MODEL decides between text and calls, which tool, and how many calls
        tools = [{"type": "function", "function": {"name": f, "parameters": {"type": "object", "properties": {"city": {"type": "string"}}}}} for f in ("get_weather", "get_timezone")]
        seen, most = set(), 0
        for k in range(40):
            r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": "auto", "max_tokens": 120, "temperature": 1.5})
            ch = r.json()["choices"][0]
            calls = ch["message"].get("tool_calls")
            if calls:
                assert ch["finish_reason"] == "tool_calls" and ch["message"]["content"] is None
                for t in calls:
                    assert t["function"]["name"] in ("get_weather", "get_timezone") and t["id"].startswith("call_") and t["type"] == "function"
                    assert isinstance(json.loads(t["function"]["arguments"])["city"], str)
                    seen.add(t["function"]["name"])
                most = max(most, len(calls))
            else:
                assert ch["finish_reason"] == "stop" and isinstance(ch["message"]["content"], str)
                seen.add("text")
        # (how MANY calls is the model's pick too: the random walks above cover it — this model's strings eat most budgets)
        assert seen == {"text", "get_weather", "get_timezone"} and most >= 1, (seen, most)
        # required: at least one call, always; one call exactly without parallel calls
        for k in range(6):
            r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": "required", "max_tokens": 120, "temperature": 1.5, "parallel_tool_calls": k % 2 == 0})
            ch = r.json()["choices"][0]
            assert ch["finish_reason"] == "tool_calls" and len(ch["message"]["tool_calls"]) >= 1 and (k % 2 == 0 or len(ch["message"]["tool_calls"]) == 1)
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": {"type": "function", "function": {"name": "get_timezone"}}})
        assert [t["function"]["name"] for t in r.json()["choices"][0]["message"]["tool_calls"]] == ["get_timezone"]
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": "none", "max_tokens": 4})
        assert "tool_calls" not in r.json()["choices"][0]["message"]
        assert c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": {"type": "function", "function": {"name": "nope"}}}).status_code == 400
        # the reference's tool profile (runners/profiles/tool-calling.yaml:22-34): over 100 sampled calls the enum takes
        # both of its values (and, being optional, is sometimes left out); every arguments object is valid
        weather = [{"type": "function", "function": {"name": "get_weather", "description": "Get current weather for a location", "parameters": {
            "type": "object", "properties": {"location": {"type": "string", "description": "City name"}, "unit": {"type": "string", "enum": ["celsius", "fahrenheit"]}}, "required": ["location"]}}}]
        # (this random model never closes a string by itself — the token budget does — so with the profile as it is
        # the optional `unit` behind `location` rarely gets room; with `unit` required and first its two values are
        # the first thing the model decides, 100 times over different prompts)
        both = json.loads(json.dumps(weather))
        both[0]["function"]["parameters"]["required"] = ["unit", "location"]
        for tools_k, need_both in ((weather, False), (both, True)):
            units = []
            for k in range(100 if need_both else 30):
                ask = [{"role": "user", "content": f"What is the weather in city number {k}?"}]
                r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": ask, "tools": tools_k, "tool_choice": {"type": "function", "function": {"name": "get_weather"}},
                                                               "max_tokens": 24, "temperature": 0.8})
                a = json.loads(r.json()["choices"][0]["message"]["tool_calls"][0]["function"]["arguments"])
                assert violation(a, tools_k[0]["function"]["parameters"]) is None, a
                units.append(a.get("unit"))
            assert set(units) <= {None, "celsius", "fahrenheit"}
            if need_both:
                assert set(units) == {"celsius", "fahrenheit"} and min(units.count("celsius"), units.count("fahrenheit")) >= 5, (units.count("celsius"), units.count("fahrenheit"))
        # streamed: the calls arrive whole once the model's document is complete; a named function streams its argument pieces
        for choice in ("required", {"type": "function", "function": {"name": "get_weather"}}):
            args = {}
            finish = None
            with c.stream("POST", url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": choice, "stream": True, "max_tokens": 120}) as rs:
                for line in rs.iter_lines():
                    if line.startswith("data: ") and line != "data: [DONE]":
                        d = json.loads(line[6:])["choices"][0]
                        for tc in d["delta"].get("tool_calls", []):
                            args[tc["index"]] = args.get(tc["index"], "") + tc["function"]["arguments"]
                        finish = d["finish_reason"] or finish
            assert finish == "tool_calls" and sorted(args) == list(range(len(args))) and args
            for v in args.values():
                assert isinstance(json.loads(v)["city"], str)
        # streamed JSON mode
        txt = ""
        with c.stream("POST", url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "response_format": {"type": "json_object"}, "stream": True, "max_tokens": 20}) as rs:
            for line in rs.iter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    txt += json.loads(line[6:])["choices"][0]["delta"].get("content") or ""
        json.loads(txt)
        # the reference's capability matrix: all five on the native server
        assert parity_probe.main(["--url", url, "--model", "tiny", "--out", "cap.json"]) == 0
        res = json.load(open("cap.json"))
        # (function calling is certain: "required" always calls; >= 2 calls in one completion is the model's pick, re-asked a few times)
        caps = res["capabilities"]
        assert all(caps[k]["supported"] for k in ("tools_function_calling", "json_mode", "logprobs", "streaming")) and caps["parallel_tool_calls"]["details"].startswith("tool_calls="), res
    finally:
        stop()


# ---------------------------------------------------------------- GPU: the kernels against the Python automaton
@pytest.mark.gpu
def test_guided_kernels_match_the_python_automaton():
    import torch

    from kvmini.ops import hip_ops

    hip_ops.require()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    V, n_cls, S = 4099, 4090, 48  # logits wider than the class table, odd sizes: scalar tail + vector body
    cls = rng.integers(0, G.N_CLASSES, n_cls).astype(np.uint8)
    cls_d = torch.from_numpy(cls).to(dev)
    # states reached by random walks of random length (so that depth / stack are consistent)
    states = []
    py_rng = random.Random(5)
    for s in range(S):
        st = (G.S_START, 0, 0, py_rng.randint(3, 50))
        for _ in range(py_rng.randint(0, 25)):
            if st[0] == G.S_END:
                break
            allow = G.allowed_classes(st, 1)
            st = G.advance(st, py_rng.choice([c for c in range(1, G.N_CLASSES) if (allow >> c) & 1]))
        states.append(st)
    states[0] = (G.S_OFF, 0, 0, 0)
    states[1] = (G.S_VALUE, 1, G.TYPE_CLASSES["string"] << G.FIRST_SHIFT, 30)
    state = torch.tensor(states, dtype=torch.int32, device=dev)
    rows = torch.tensor(py_rng.sample(range(64), S), dtype=torch.int64, device=dev)
    slots = torch.arange(S, dtype=torch.int32, device=dev)
    # a third of the rows arrive with an init record that replaces a garbage device state
    init = torch.full((S, 4), -1, dtype=torch.int32)
    for s in range(0, S, 3):
        init[s] = torch.tensor(states[s], dtype=torch.int32)
        state[s] = torch.tensor([G.S_DONE, 9, 5, 1], dtype=torch.int32)
    logits = torch.randn(64, V, device=dev).to(torch.bfloat16)
    ref = logits.clone()
    for vc in (1, 2):
        got = ref.clone()
        st_d = state.clone()
        hip_ops.guided_mask(got, rows, slots, init.to(dev).contiguous(), st_d, cls_d, vc)
        torch.cuda.synchronize()
        assert st_d.cpu().tolist() == [list(s) for s in states]  # init records landed
        for s in range(S):
            allow = G.allowed_classes(states[s], vc)
            ok = np.array([(allow >> int(c)) & 1 for c in cls], dtype=bool)
            ok = np.concatenate([ok, np.full(V - n_cls, allow == (1 << G.N_CLASSES) - 1)])
            r = int(rows[s])
            want = torch.where(torch.from_numpy(ok).to(dev), ref[r].float(), torch.full((V,), float("-inf"), device=dev))
            assert torch.equal(got[r].float(), want), (s, states[s])
        untouched = sorted(set(range(64)) - set(rows.tolist()))
        assert torch.equal(got[untouched], ref[untouched])
    # advance: feed one token per row
    out_ids = torch.from_numpy(rng.integers(0, V, 64)).to(dev)
    st_d = torch.tensor(states, dtype=torch.int32, device=dev)
    hip_ops.guided_advance(out_ids, rows, slots, st_d, cls_d)
    torch.cuda.synchronize()
    for s in range(S):
        tok = int(out_ids[int(rows[s])])
        want = G.advance(states[s], int(cls[tok]) if tok < n_cls else G.C_OTHER)
        assert tuple(st_d[s].tolist()) == want, (s, states[s], tok)


@pytest.mark.gpu
@pytest.mark.parametrize("vocab,V", [(2048, 2051), (128256, 128256)])
def test_guided_script_kernels_walk_in_step_with_the_python_automaton(vocab, V):
    """Scripted (schema-guided) sequences through the two kernels, one token at a time: at every
    step the masked logits row, the automaton state and the script cursor equal the Python
    reference; the tokens are drawn at random from what the mask left, so forced skeleton tokens,
    typed value slots, numbers that end with the next literal, choices, loops and nested containers
    all occur.  Two vocabularies: a small one (with padding columns past the class table) and the
    Llama-3 sized one, whose tokenizer merges the skeleton into different tokens."""
    import torch

    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.ops import hip_ops

    hip_ops.require()
    dev = torch.device("cuda:0")
    tok = get_tokenizer(vocab, 1, 2)
    cls = G.token_classes(tok, vocab, 2)
    vc = G.value_cost(cls)
    cls_d = torch.from_numpy(cls).to(dev)
    schema = dict(SCHEMA, required=SCHEMA["required"] + ["days", "extra", "alerts"],
                  properties=dict(SCHEMA["properties"], days=CHOICES["properties"]["days"], extra=CHOICES["properties"]["extra"], alerts=CHOICES["properties"]["alerts"]))
    prefix, start, script, texts = G.schema_script(schema, tok.encode_pieces)
    rest = G.script_rest(script, vc)
    S, L = 6, G.SCRIPT_MAX
    budgets = [rest[0], rest[0] + 3, rest[0] + 40, rest[0] + 11, 200, rest[0] + 25]
    pad = L - len(script)
    script_d = torch.zeros((S, 2, L), dtype=torch.int32, device=dev)
    for s_ in range(S - 1):  # the last slot carries no script: plain JSON mode next to the scripted ones
        script_d[s_] = torch.tensor([script + [G.SCRIPT_END] * pad, rest + [1] * pad], dtype=torch.int32, device=dev)
    cursor_d = torch.full((S,), 7, dtype=torch.int32, device=dev)  # garbage until the init records land
    state_d = torch.full((S, 4), 3, dtype=torch.int32, device=dev)
    rows = torch.tensor([5, 0, 3, 1, 4, 2], dtype=torch.int64, device=dev)
    slots = torch.arange(S, dtype=torch.int32, device=dev)
    py = [((start[0], start[1], start[2], budgets[s_]), 0) for s_ in range(S - 1)] + [((G.S_START, 0, 0, budgets[S - 1]), -1)]
    out = [[] for _ in range(S)]
    gen = torch.Generator().manual_seed(3)
    rng = random.Random(9)
    step = 0
    while any(st[0] != G.S_END for st, _ in py):
        logits = torch.randn(S, V, generator=gen).to(torch.bfloat16).to(dev)
        ref = logits.clone()
        init = torch.tensor([list(st) for st, _ in py] if step == 0 else [[-1, 0, 0, 0]] * S, dtype=torch.int32, device=dev)
        cur_init = torch.tensor([c for _, c in py] if step == 0 else [-1] * S, dtype=torch.int32, device=dev)
        hip_ops.guided_mask(logits, rows, slots, init, state_d, cls_d, vc, script_d, cursor_d, cur_init)
        picks = torch.zeros(S, dtype=torch.int64)
        for s_ in range(S):
            st, cur = py[s_]
            allow, toks = G.allowed_scripted(st, cur, script if s_ < S - 1 else None, rest if s_ < S - 1 else None, vc)
            ok = np.array([(allow >> int(c)) & 1 for c in cls] + [allow == (1 << G.N_CLASSES) - 1] * (V - len(cls)), dtype=bool)
            for forced in toks:
                ok[forced] = True
            r = int(rows[s_])
            want = torch.where(torch.from_numpy(ok).to(dev), ref[r].float(), torch.full((V,), float("-inf"), device=dev))
            assert torch.equal(logits[r].float(), want), (step, s_, st, cur)
            options = np.flatnonzero(ok[: len(cls)])
            picks[r] = int(rng.choice(list(options))) if st[0] != G.S_END else 2
        out_ids = picks.to(dev)
        hip_ops.guided_advance(out_ids, rows, slots, state_d, cls_d, script_d, cursor_d)
        torch.cuda.synchronize()
        for s_ in range(S):
            st, cur = py[s_]
            t = int(picks[int(rows[s_])])
            if st[0] != G.S_END:
                out[s_].append(t)
            py[s_] = G.advance_scripted(st, cur, t, cls, script if s_ < S - 1 else None)
            assert tuple(state_d[s_].tolist()) == py[s_][0] and int(cursor_d[s_]) == py[s_][1], (step, s_, st, cur, t)
        step += 1
        assert step < 600
    for s_ in range(S - 1):
        assert len(out[s_]) <= budgets[s_] and out[s_][-1] == 2
        text = G.ScriptedText(cls, start, budgets[s_], script, texts, tok.decode_token)
        doc = json.loads(prefix + "".join(text.piece(t) for t in out[s_]))
        _check_schema_doc({k: v for k, v in doc.items() if k not in ("days", "extra", "alerts")})
        assert all(a["level"] in ("low", "high", 3) for a in doc["alerts"])
        assert 1 <= len(doc["days"]) <= 3 and (doc["extra"] is None or isinstance(doc["extra"], (int, dict)))
    assert isinstance(json.loads(tok.decode(out[S - 1])), dict)


@pytest.mark.gpu
def test_guided_mask_at_a_choice_leaves_exactly_the_alternatives():
    """A sequence that stands at a choice between two tokens (the enum of the reference's tool
    profile): the mask leaves those two logits and nothing else, for one row and for many, on a
    contiguous logits tensor and on a view of a wider one."""
    import torch

    from kvmini.engine.tokenizer import get_tokenizer
    from kvmini.ops import hip_ops

    hip_ops.require()
    dev = torch.device("cuda:0")
    V = 128256
    tok = get_tokenizer(V, 128000, 128001)
    cls = G.token_classes(tok, V, 128001)
    vc = G.value_cost(cls)
    cls_d = torch.from_numpy(cls).to(dev)
    _prefix, _start, script, _texts = G.schema_script(CHOICES, tok.encode_pieces)
    rest = G.script_rest(script, vc)
    at = next(i for i in G._instruction_starts(script) if G.entry_kind(script[i]) == "choice")
    state = (G.S_OBJ_NEXT, 1, 0, 25)
    allow, toks = G.allowed_scripted(state, at, script, rest, vc)
    assert allow == 0 and len(toks) == 2
    L, pad = G.SCRIPT_MAX, G.SCRIPT_MAX - len(script)
    for S in (1, 16):
        script_d = torch.tensor([[script + [G.SCRIPT_END] * pad, rest + [1] * pad]] * S, dtype=torch.int32, device=dev)
        cursor_d = torch.full((S,), at, dtype=torch.int32, device=dev)
        state_d = torch.tensor([list(state)] * S, dtype=torch.int32, device=dev)
        for width in (V, V + 64):
            full = torch.randn(S, width, device=dev).to(torch.bfloat16)
            logits, before = full[:, :V], full.clone()
            hip_ops.guided_mask(logits, torch.arange(S, dtype=torch.int64, device=dev), torch.arange(S, dtype=torch.int32, device=dev),
                                torch.tensor([[-1, 0, 0, 0]] * S, dtype=torch.int32, device=dev), state_d, cls_d, vc, script_d, cursor_d, torch.full((S,), -1, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            for r in range(S):
                kept = torch.nonzero(torch.isfinite(logits[r].float())).view(-1).tolist()
                assert kept == sorted(toks), (S, width, r, kept[:20])
                assert torch.equal(logits[r, kept], before[r, kept]) and torch.equal(full[r, V:], before[r, V:])


@pytest.mark.gpu
def test_engine_schema_guided_generation_on_the_hip_path():
    """A schema-guided batch on the HIP engine (graphs, launch-ahead): every output has all
    required keys with the declared types; greedy and sampled."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.engine.tokenizer import get_tokenizer

    eng = LLMEngine(EngineConfig(model="llama-3.2-1b", device="cuda:0", max_num_seqs=16, max_batched_tokens=128, max_model_len=512, kernels="hip", kv_memory_fraction=0.1))
    eng.warmup()
    tok = get_tokenizer(eng.mcfg.vocab_size, eng.mcfg.bos_token_id, eng.mcfg.eos_token_id)
    cls = G.token_classes(tok, eng.mcfg.vocab_size, eng.mcfg.eos_token_id)
    eng.set_token_classes(cls)
    prefix, start, script, texts = G.schema_script(SCHEMA, tok.encode_pieces)
    prompts = [tok.encode(f"describe person {i} as json " + prefix, add_bos=True) for i in range(10)]
    for temp in (0.0, 0.9):
        sp = SamplingParams(max_tokens=80, temperature=temp, guided="json", guided_start=start, guided_script=tuple(script))
        outs = eng.generate(prompts, sp)
        for o in outs:
            assert o[-1] == eng.mcfg.eos_token_id and len(o) <= 80
            text = G.ScriptedText(cls, start, 80, script, texts, tok.decode_token)
            _check_schema_doc(json.loads(prefix + "".join(text.piece(t) for t in o)))
    assert eng.pipelined_steps > 0
    # a nested schema (objects in objects, an array of objects): the same kernels, a longer script
    from kvmini.loadgen.schema_check import violation

    prefix, start, script, texts = G.schema_script(NESTED, tok.encode_pieces)
    prompts = [tok.encode(f"order {i} as json " + prefix, add_bos=True) for i in range(10)]
    for temp in (0.0, 0.9):
        sp = SamplingParams(max_tokens=160, temperature=temp, guided="json", guided_start=start, guided_script=tuple(script))
        for o in eng.generate(prompts, sp):
            assert o[-1] == eng.mcfg.eos_token_id and len(o) <= 160
            text = G.ScriptedText(cls, start, 160, script, texts, tok.decode_token)
            doc = json.loads(prefix + "".join(text.piece(t) for t in o))
            assert violation(doc, NESTED) is None and len(doc["items"]) >= 2, doc
    # choices (enum values, alternatives, array lengths) are the model's: sampled, every unit shows up
    prefix, start, script, texts = G.schema_script(CHOICES, tok.encode_pieces)
    prompts = [tok.encode(f"weather report {i} as json " + prefix, add_bos=True) for i in range(16)]
    units, days = set(), set()
    for rep in range(3):
        sp = SamplingParams(max_tokens=120, temperature=1.0, guided="json", guided_start=start, guided_script=tuple(script))
        for o in eng.generate(prompts, sp):
            assert o[-1] == eng.mcfg.eos_token_id and len(o) <= 120
            text = G.ScriptedText(cls, start, 120, script, texts, tok.decode_token)
            doc = json.loads(prefix + "".join(text.piece(t) for t in o))
            assert violation(doc, CHOICES) is None, doc
            units.add(doc["unit"])
            days.add(len(doc["days"]))
    assert units == {"celsius", "fahrenheit"} and days, (units, days)  # (this model never closes a string by itself: the budget does, so the arrays stay at their minimum)


@pytest.mark.gpu
def test_engine_guided_json_on_the_hip_path_is_pipelined_and_valid():
    import torch

    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.engine.tokenizer import get_tokenizer

    eng = LLMEngine(EngineConfig(model="llama-3.2-1b", device="cuda:0", max_num_seqs=16, max_batched_tokens=128, max_model_len=512, kernels="hip", kv_memory_fraction=0.1))
    eng.warmup()
    tok = get_tokenizer(eng.mcfg.vocab_size, eng.mcfg.bos_token_id, eng.mcfg.eos_token_id)
    eng.set_token_classes(G.token_classes(tok, eng.mcfg.vocab_size, eng.mcfg.eos_token_id))
    prompts = [tok.encode(f"tell me about item {i} as json", add_bos=True) for i in range(12)]
    for temp in (0.0, 0.8):
        outs = eng.generate(prompts, SamplingParams(max_tokens=48, temperature=temp, top_p=0.95, guided="json"))
        for o in outs:
            assert o[-1] == eng.mcfg.eos_token_id and len(o) <= 48
            assert isinstance(json.loads(tok.decode(o)), dict)
    assert eng.pipelined_steps > 0
    torch.cuda.synchronize()


def test_schema_check_of_structured_answers():
    """The load generator's compliance check: parse errors, missing required keys, wrong types
    (a boolean is not an integer), enum / const, nested properties and array items."""
    from kvmini.loadgen.schema_check import check_document

    good = '{"name":"x","age":3,"tags":["a"],"ok":false,"kind":"c","address":{}}'
    assert check_document(good, SCHEMA) == (True, True, None)
    assert check_document('{"name":"x"', SCHEMA)[:2] == (False, False)
    assert check_document('[1]', SCHEMA)[:2] == (True, False)
    for bad, needle in (('{"name":"x","age":3,"tags":[],"ok":true,"kind":"c"}', "'address' is missing"),
                        ('{"name":"x","age":true,"tags":[],"ok":true,"kind":"c","address":{}}', "$.age"),
                        ('{"name":"x","age":3,"tags":[],"ok":null,"kind":"c","address":{}}', "$.ok"),
                        ('{"name":"x","age":3,"tags":[],"ok":true,"kind":"zzz","address":{}}', "$.kind")):
        ok_json, ok_schema, why = check_document(bad, SCHEMA)
        assert ok_json and not ok_schema and needle in why, (bad, why)
    nested = {"type": "object", "properties": {"a": {"type": "array", "items": {"type": "object", "required": ["b"], "properties": {"b": {"const": 1}}}}, "n": {"type": ["number", "null"]}}}
    assert check_document('{"a":[{"b":1},{"b":1}],"n":null}', nested)[1] and check_document('{"n":2.5}', nested)[1]
    assert "$.a[1].b" in check_document('{"a":[{"b":1},{"b":2}]}', nested)[2]
    assert "$.a[0]" in check_document('{"a":[{}]}', nested)[2]
    # numeric bounds are checked (a lower bound of 0 holds by construction on the native engine; an upper one may not)
    bounded = {"type": "object", "properties": {"age": {"type": "integer", "minimum": 0, "maximum": 150}, "p": {"type": "number", "exclusiveMinimum": 0}}}
    assert check_document('{"age":5,"p":0.5}', bounded)[1]
    for bad, needle in (('{"age":-1}', "minimum"), ('{"age":200}', "maximum"), ('{"p":0}', "exclusiveMinimum")):
        assert needle in check_document(bad, bounded)[2]
