"""The native engine against Hugging Face ``transformers`` on tiny random checkpoints: the architecture is read
from ``config.json``, the weights from safetensors, and greedy tokens / logits must match the library's
implementation of the same model (rotary convention, grouped-query attention, "llama3" rotary scaling, tied
embeddings, explicit head widths, Mistral's config, Qwen2's Q / K / V bias).  CPU, fp32, reference ops."""

import json
import os

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _save(tmp_path, kind, **kw):
    torch.manual_seed(7)
    base = dict(vocab_size=160, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512,
                rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    base.update(kw)
    if kind == "mistral":
        cfg = transformers.MistralConfig(**base)
        model = transformers.MistralForCausalLM(cfg)
    elif kind == "qwen2":
        cfg = transformers.Qwen2Config(**base)
        model = transformers.Qwen2ForCausalLM(cfg)
        with torch.no_grad():  # (the library initialises biases to zero: give them values that matter)
            for layer in model.model.layers:
                for proj in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj):
                    proj.bias.normal_(0.0, 0.5)
    else:
        cfg = transformers.LlamaConfig(**base)
        model = transformers.LlamaForCausalLM(cfg)
    model = model.eval().float()
    d = str(tmp_path / kind)
    model.save_pretrained(d, safe_serialization=True)
    return d, model


def _engine(path, **kw):
    from kvmini.engine.engine import EngineConfig, LLMEngine

    return LLMEngine(EngineConfig(model=path, device="cpu", dtype="fp32", kernels="torch", max_model_len=256, max_num_seqs=4, max_batched_tokens=64, use_graphs=False, **kw))


CASES = {
    "llama-gqa": ("llama", dict(rope_theta=10000.0)),
    "llama3-scaled-tied": ("llama", dict(rope_theta=500000.0, tie_word_embeddings=True,
                                         rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 64})),
    "llama-mha-wide-heads": ("llama", dict(num_key_value_heads=4, head_dim=32)),
    "mistral": ("mistral", dict(rope_theta=10000.0, sliding_window=None)),
    "qwen2-qkv-bias": ("qwen2", dict(rope_theta=1000000.0, rms_norm_eps=1e-6, tie_word_embeddings=True)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_greedy_tokens_and_logits_equal_transformers(tmp_path, case):
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import config_from_hf

    kind, kw = CASES[case]
    path, hf = _save(tmp_path, kind, **kw)
    cfg = config_from_hf(path)
    assert cfg.num_layers == 2 and cfg.num_heads == 4 and cfg.vocab_size == 160
    assert cfg.tie_word_embeddings == bool(kw.get("tie_word_embeddings", False))
    assert (cfg.rope_scaling is not None) == ("rope_scaling" in kw)
    assert cfg.head_dim == kw.get("head_dim", 16)
    eng = _engine(path)
    prompts = [[1, 5, 9, 33, 77, 100, 4], [1] + [(7 * i) % 150 + 3 for i in range(90)]]  # the long one passes the scaled rotary table's original range
    mine = eng.generate(prompts, SamplingParams(max_tokens=10, ignore_eos=True))
    for p, got in zip(prompts, mine):
        with torch.no_grad():
            want = hf.generate(torch.tensor([p]), max_new_tokens=10, do_sample=False, pad_token_id=0)[0, len(p):].tolist()
        assert got == want, (case, got, want)
    # logits of the prompt's last position, through the reference forward pass on the loaded weights
    from kvmini.models.reference import dense_logits

    ids = torch.tensor([prompts[0]])
    with torch.no_grad():
        want = hf(ids).logits[0, -1]
    got = dense_logits(eng.weights, prompts[0])[-1]
    assert torch.allclose(got.float(), want.float(), atol=2e-4, rtol=1e-3), (got - want).abs().max()


def test_unsupported_architectures_are_refused(tmp_path):
    from kvmini.models.llama import config_from_hf

    base = dict(model_type="llama", vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2)
    for bad, msg in ((dict(model_type="gemma2"), "model_type"), (dict(attention_bias=True), "bias"), (dict(rope_scaling={"rope_type": "yarn", "factor": 4.0}), "rope_scaling"),
                     (dict(num_local_experts=8), "experts"), (dict(hidden_act="gelu"), "hidden_act")):
        with pytest.raises(ValueError, match=msg):
            config_from_hf({**base, **bad})
    cfg = config_from_hf({**base, "model_type": "mistral", "sliding_window": 128, "max_position_embeddings": 4096, "eos_token_id": [7, 9]})
    assert cfg.max_position == 128 and cfg.eos_token_id == 7 and cfg.extra_eos_token_ids == (9,) and cfg.num_kv_heads == 2 and cfg.head_dim == 16
    # a chat checkpoint's generation_config.json adds its end-of-turn id
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({**base, "eos_token_id": 5}))
    (d / "generation_config.json").write_text(json.dumps({"eos_token_id": [5, 11, 12]}))
    assert config_from_hf(str(d)).extra_eos_token_ids == (11, 12)


def test_tensor_parallel_ranks_read_only_their_slices(tmp_path):
    """Two ranks' shards of a checkpoint, concatenated, are the one-rank weights."""
    from kvmini.models.llama import config_from_hf, load_hf_safetensors

    path, _ = _save(tmp_path, "llama")
    cfg = config_from_hf(path)
    full = load_hf_safetensors(path, cfg, "cpu", torch.float32)
    parts = [load_hf_safetensors(path, cfg, "cpu", torch.float32, tp_rank=r, tp_size=2) for r in range(2)]
    hd, nh, nkv = cfg.head_dim, cfg.num_heads, cfg.num_kv_heads
    for li in range(cfg.num_layers):
        f = full.layers[li]
        q = torch.cat([p.layers[li].wqkv[: nh // 2 * hd] for p in parts])
        k = torch.cat([p.layers[li].wqkv[nh // 2 * hd : nh // 2 * hd + nkv // 2 * hd] for p in parts])
        assert torch.equal(q, f.wqkv[: nh * hd]) and torch.equal(k, f.wqkv[nh * hd : nh * hd + nkv * hd])
        assert torch.equal(torch.cat([p.layers[li].wo for p in parts], 1), f.wo)
        assert torch.equal(torch.cat([p.layers[li].wdown for p in parts], 1), f.wdown)


def test_server_takes_a_checkpoint_directory_with_its_tokenizer(tmp_path):
    """`kvmini serve --model DIR`: architecture, weights and tokenizer.json all come from the directory."""
    from tokenizers import Tokenizer, models, pre_tokenizers

    path, _ = _save(tmp_path, "llama")
    vocab = {f"w{i}": i for i in range(160)}
    vocab["[UNK]"] = 0
    tok = Tokenizer(models.WordLevel({**{k: v for k, v in vocab.items() if v != 0}, "[UNK]": 0}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(os.path.join(path, "tokenizer.json"))
    from kvmini.server.native import NativeServer

    eng = _engine(path)
    ns = NativeServer(eng, "tiny-hf", os.path.join(path, "tokenizer.json"))
    ids = ns.tok.encode("w5 w9 w33")
    assert [i for i in ids if i in (5, 9, 33)] == [5, 9, 33]
    assert json.load(open(os.path.join(path, "config.json")))["vocab_size"] == eng.mcfg.vocab_size


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["llama", "qwen2"])
def test_hip_kernels_follow_transformers_on_a_checkpoint_directory(tmp_path, kind):
    """The same path on the GPU: bf16 weights through the HIP kernels (head width 128, grouped heads, HIP graphs)
    against the library's fp32 forward pass on the host, teacher-forced along the engine's tokens."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams

    extra = dict(head_dim=128) if kind == "llama" else {}  # (Qwen2's config has no head_dim: hidden / heads = 128)
    path, hf = _save(tmp_path, kind, vocab_size=1024, hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=2,
                     rope_theta=500000.0, initializer_range=0.08, **extra)
    eng = LLMEngine(EngineConfig(model=path, device="cuda:0", kernels="hip", max_model_len=512, max_num_seqs=4, max_batched_tokens=256, kv_memory_fraction=0.02))
    eng.warmup()
    assert (eng.weights.layers[0].bqkv is not None) == (kind == "qwen2")
    prompts = [[1, 5, 9, 33, 77, 100, 4], [1] + [(7 * i) % 900 + 3 for i in range(150)]]  # (a decode-sized step with slabs, a 151-token prompt on the tiled kernel)
    mine = eng.generate(prompts, SamplingParams(max_tokens=12, ignore_eos=True))
    for p, got in zip(prompts, mine):
        # teacher-forced: along the engine's own tokens, every token is the library's argmax or within a bf16 rounding
        # of it (a near-tie may flip; after a flip a free-running comparison would diverge for good)
        with torch.no_grad():
            lg = hf(torch.tensor([p + got])).logits[0]
        margins = [float(lg[len(p) - 1 + i].max() - lg[len(p) - 1 + i][t]) for i, t in enumerate(got)]
        assert max(margins) <= 0.08 and sum(m == 0.0 for m in margins) >= 10, (got, margins)
        assert margins[0] == 0.0


def _word_tokenizer(tmp_path, words, specials, metaspace=False):
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers

    if metaspace:
        vocab = {"[UNK]": 0, **{"▁" + w: i + 1 for i, w in enumerate(words)}}
        tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
        tok.pre_tokenizer = pre_tokenizers.Metaspace()
        tok.decoder = decoders.Metaspace()
    else:
        vocab = {"[UNK]": 0, **{w: i + 1 for i, w in enumerate(words)}}
        tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
        tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.add_special_tokens([AddedToken(t, special=True) for t in specials])
    path = str(tmp_path / "tokenizer.json")
    tok.save(path)
    return path


def test_chat_template_of_the_checkpoint_is_rendered_and_streamed_pieces_keep_their_spaces(tmp_path):
    from kvmini.engine.tokenizer import HFTokenizer

    words = ["hello", "there", "you", "are", "a", "bot", "system", "user", "assistant", "hi"]
    path = _word_tokenizer(tmp_path, words, ["<|im_start|>", "<|im_end|>"], metaspace=True)
    with open(tmp_path / "tokenizer_config.json", "w") as f:
        json.dump({"bos_token": "<|im_start|>", "eos_token": {"content": "<|im_end|>"},
                   "chat_template": "{% for m in messages %}<|im_start|>{{ m['role'] }} {{ m['content'] }}<|im_end|>{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant{% endif %}"}, f)
    tok = HFTokenizer(path, 11, 12)
    msgs = [{"role": "system", "content": "you are a bot"}, {"role": "user", "content": "hello there"}]
    assert tok.render_chat(msgs) == "<|im_start|>system you are a bot<|im_end|><|im_start|>user hello there<|im_end|><|im_start|>assistant"
    ids = tok.apply_chat_template(msgs)
    start, end = tok.tk.token_to_id("<|im_start|>"), tok.tk.token_to_id("<|im_end|>")
    assert ids[0] == start and ids.count(start) == 3 and ids.count(end) == 2 and 0 not in ids
    assert tok.eos_token == "<|im_end|>"
    # streamed one token at a time, the words keep the spaces between them
    reply = tok.encode("hello there you are")
    assert "".join(tok.decode_token(t) for t in reply).strip() == "hello there you are" == tok.decode(reply).strip()
    # a template that calls into Python internals is stopped by the sandbox
    with open(tmp_path / "tokenizer_config.json", "w") as f:
        json.dump({"chat_template": "{{ ''.__class__.__mro__[1].__subclasses__() }}"}, f)
    with pytest.raises(Exception):
        HFTokenizer(path, 11, 12).render_chat(msgs)


def test_chat_format_follows_the_vocabulary_without_a_template(tmp_path):
    from kvmini.engine.tokenizer import HFTokenizer

    words = ["hello", "user", "assistant", "system", "be", "brief"]
    msgs = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "hello"}]
    chatml = HFTokenizer(_word_tokenizer(tmp_path, words, ["<|im_start|>", "<|im_end|>"]), 1, 2)
    assert chatml.render_chat(msgs).endswith("<|im_start|>user\nhello<|im_end|>\n<|im_start|>assistant\n")
    inst = HFTokenizer(_word_tokenizer(tmp_path, words, ["[INST]", "[/INST]"]), 1, 2)
    assert "[INST] be brief\n\nhello [/INST]" in inst.render_chat(msgs)
    l3 = HFTokenizer(_word_tokenizer(tmp_path, words, ["<|begin_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"]), 1, 2)
    assert l3.render_chat(msgs).startswith("<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\nbe brief<|eot_id|>")
    plain = HFTokenizer(_word_tokenizer(tmp_path, words, []), 1, 2)
    assert plain.render_chat(msgs) == "system: be brief\nuser: hello\nassistant:"


def test_offline_generation_on_a_checkpoint_directory(tmp_path, capsys):
    """`kvmini generate --model DIR`: prompts from flags and a file (plain lines, JSON lines with a token budget of
    their own), replies as JSON lines whose tokens are the library's greedy continuation of the same ids."""
    from tokenizers import Tokenizer, models, pre_tokenizers

    from kvmini.server.offline import main

    path, hf = _save(tmp_path, "llama")
    tok = Tokenizer(models.WordLevel({**{f"w{i}": i for i in range(1, 160)}, "[UNK]": 0}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(os.path.join(path, "tokenizer.json"))
    pf = tmp_path / "prompts.txt"
    pf.write_text('w5 w9 w33\n{"prompt": "w7 w8", "max_tokens": 3}\n\n')
    out = tmp_path / "gen.jsonl"
    rc = main(["--model", path, "--device", "cpu", "--dtype", "fp32", "--kernels", "torch", "--no-graphs", "--max-model-len", "64", "--max-num-seqs", "4",
               "--max-batched-tokens", "32", "--prompt", "w11 w12 w13 w14", "--prompts-file", str(pf), "--max-tokens", "6", "--out", str(out)])
    assert rc == 0
    rows = [json.loads(ln) for ln in out.read_text().splitlines()]
    assert [r["completion_tokens"] for r in rows] == [6, 6, 3]
    summary = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert summary["requests"] == 3 and summary["completion_tokens"] == 15
    from kvmini.engine.tokenizer import get_tokenizer

    t = get_tokenizer(hf.config.vocab_size, 1, 2, os.path.join(path, "tokenizer.json"))
    for text, r in zip(["w11 w12 w13 w14", "w5 w9 w33", "w7 w8"], rows):
        ids = t.encode(text, add_bos=True)
        assert r["prompt_tokens"] == len(ids)
        with torch.no_grad():
            want = hf.generate(torch.tensor([ids]), max_new_tokens=r["completion_tokens"], do_sample=False, eos_token_id=None, pad_token_id=0)[0, len(ids):].tolist()
        assert r["token_ids"] == want


def test_offline_generation_reads_prompts_from_stdin_and_refuses_what_cannot_run(monkeypatch, capsys):
    """`kvmini generate --prompts-file -` on a preset: plain lines, JSON lines with messages, blank lines skipped; a
    prompt longer than the model length ends the command with the scheduler's reason before anything is generated."""
    import io

    from kvmini.server.offline import main

    common = ["--model", "tiny", "--device", "cpu", "--dtype", "fp32", "--kernels", "torch", "--no-graphs", "--max-model-len", "64", "--max-num-seqs", "4", "--max-batched-tokens", "32"]
    monkeypatch.setattr("sys.stdin", io.StringIO('hello there\n\n{"messages": [{"role": "user", "content": "hi"}]}\n{not json\n'))
    assert main(common + ["--prompts-file", "-", "--max-tokens", "4", "--temperature", "0.8", "--sampling-seed", "7"]) == 0
    out = capsys.readouterr()
    assert out.out.count("--- [") == 3 and "4 generated" in out.out
    summary = json.loads(out.err.strip().splitlines()[-1])
    assert summary["requests"] == 3 and summary["completion_tokens"] == 12
    with pytest.raises(SystemExit) as e:
        main(common + ["--prompt", " ".join(["word"] * 200)])
    assert "max_model_len" in str(e.value)
    with pytest.raises(SystemExit):
        main(common)  # no prompts at all
