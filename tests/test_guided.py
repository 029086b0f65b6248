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
             G.C_DIGIT: "0", G.C_LIT: "true", G.C_EOS: ""}
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
            options = [c for c in range(1, G.N_CLASSES) if (allow >> c) & 1 and not (vc == 2 and c in (G.C_LIT, G.C_DIGIT_NZ) and st[0] in (G.S_VALUE, G.S_ARR_FIRST))]
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
    assert G.replay(cls, (G.S_START, 0, 0), 30, [1, 2, 3, 2, 4, 5]) == (G.S_AFTER_VALUE, 1, 0, 24)


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
            assert isinstance(doc["age"], int)  # the first required key is written by the server, its value typed by the automaton
        tools = [{"type": "function", "function": {"name": f, "parameters": {"type": "object", "properties": {"city": {"type": "string"}}}}} for f in ("get_weather", "get_timezone")]
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": "auto", "max_tokens": 40})
        ch = r.json()["choices"][0]
        assert ch["finish_reason"] == "tool_calls" and ch["message"]["content"] is None
        assert [t["function"]["name"] for t in ch["message"]["tool_calls"]] == ["get_weather", "get_timezone"]
        for t in ch["message"]["tool_calls"]:
            assert isinstance(json.loads(t["function"]["arguments"])["city"], str) and t["id"].startswith("call_") and t["type"] == "function"
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": {"type": "function", "function": {"name": "get_timezone"}}})
        assert [t["function"]["name"] for t in r.json()["choices"][0]["message"]["tool_calls"]] == ["get_timezone"]
        r = c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": "none", "max_tokens": 4})
        assert "tool_calls" not in r.json()["choices"][0]["message"]
        assert c.post(url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "tool_choice": {"type": "function", "function": {"name": "nope"}}}).status_code == 400
        # streamed tool calls: the argument pieces of each call concatenate to JSON
        args = {}
        finish = None
        with c.stream("POST", url + "/v1/chat/completions", json={"model": "tiny", "messages": msg, "tools": tools, "stream": True, "max_tokens": 30}) as rs:
            for line in rs.iter_lines():
                if line.startswith("data: ") and line != "data: [DONE]":
                    d = json.loads(line[6:])["choices"][0]
                    for tc in d["delta"].get("tool_calls", []):
                        args[tc["index"]] = args.get(tc["index"], "") + tc["function"]["arguments"]
                    finish = d["finish_reason"] or finish
        assert finish == "tool_calls" and sorted(args) == [0, 1]
        for v in args.values():
            json.loads(v)
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
        assert res["summary"] == {"supported_count": 5, "total": 5}, res
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
