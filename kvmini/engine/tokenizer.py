"""Tokenizers for the native server.

There is no network on a benchmark box, so real vocabularies are usually not
available; random-init weights make the text meaningless anyway.  The
:class:`SyntheticTokenizer` is a deterministic word/punctuation tokenizer that
maps pieces into the model's id space by hashing and wraps chat messages in a
Llama-3-shaped template (BOS, role header, content, end-of-turn, assistant
header), so prompt *lengths* — which are what TTFT and throughput depend on — are
realistic: the reference's default prompt ``"Hello, world!"``
(``scripts/loadtest.py:635``) becomes 14 tokens.  When a ``tokenizer.json`` is
available (``--tokenizer PATH``) the Hugging Face ``tokenizers`` runtime is used
instead.
"""

from __future__ import annotations

import re
import zlib
from typing import Any, Dict, List, Optional

_PIECE = re.compile(r"\w+|[^\w\s]")
_WORDS = (
    "the of and to in is that for it as was with be by on not he this are or his from at which but have an had they you were "
    "their one all we can her has there been if more when will would who so no out up said what its about than into them only "
    "time could new then some these may other do first any now such like our over man even most made after also did many must "
    "before back through years where much your way well down should because each just those people how too little state good "
    "very make world still own see men work long get here between both life being under never day same another know while last"
).split()


# ids [16, 48): the JSON alphabet, one token per piece, so that guided decoding has real structural
# tokens to allow and the text of a guided completion parses
_JSON_PIECES = ("{", "}", "[", "]", ":", ",", '"', "true", "false", "null", "0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "42", "100", "2024", "-", ".", "_", "/", "(", ")", "'", "!")


class SyntheticTokenizer:
    n_special = 16  # ids [0, 16) are reserved: 0 pad, 1 bos, 2 eos, 3 eot, 4.. headers
    n_reserved = n_special + len(_JSON_PIECES)

    def __init__(self, vocab_size: int, bos_id: int = 1, eos_id: int = 2):
        self.vocab_size = int(vocab_size)
        self.bos_id = min(bos_id, self.vocab_size - 1)
        self.eos_id = min(eos_id, self.vocab_size - 1)
        # tiny test vocabularies keep the plain hashed layout
        self._json = {p: self.n_special + i for i, p in enumerate(_JSON_PIECES)} if self.vocab_size >= 4 * self.n_reserved else {}
        self._json_text = {i: p for p, i in self._json.items()}
        self._base = self.n_reserved if self._json else self.n_special
        self._span = max(1, self.vocab_size - self._base)

    def _piece_id(self, piece: str) -> int:
        fixed = self._json.get(piece)
        if fixed is not None:
            return fixed
        return self._base + (zlib.crc32(piece.encode("utf-8")) % self._span)

    def encode(self, text: str, add_bos: bool = False) -> List[int]:
        ids = [self._piece_id(p) for p in _PIECE.findall(text)]
        return ([self.bos_id] + ids) if add_bos else ids

    def encode_pieces(self, text: str) -> List[Any]:
        """``[(token id, the text it stands for)]`` — word ids are hashes, so a token does not
        decode back to its word; callers that must reproduce ``text`` exactly (forced skeleton
        tokens of guided decoding) keep the pieces."""
        return [(self._piece_id(m.group(1)), m.group(0)) for m in re.finditer(r"\s*(\w+|[^\w\s])", text)]

    def apply_chat_template(self, messages: List[Dict[str, Any]]) -> List[int]:
        """BOS, then per message <|start_header|> role <|end_header|> \\n\\n content <|eot|>,
        then the assistant header — the token budget of the Llama-3 template."""
        ids = [self.bos_id]
        for m in messages:
            ids += [4, self._piece_id(str(m.get("role", "user"))), 5, 6]
            content = m.get("content")
            if isinstance(content, list):  # OpenAI content parts
                content = " ".join(str(p.get("text", "")) for p in content if isinstance(p, dict))
            ids += self.encode(str(content or ""))
            ids.append(3)
        ids += [4, self._piece_id("assistant"), 5, 6]
        return [i % self.vocab_size for i in ids]

    def decode_token(self, tok: int) -> str:
        if tok < self.n_special or tok in (self.eos_id, self.bos_id):  # (model configs put BOS / EOS anywhere in the id space)
            return ""
        fixed = self._json_text.get(tok)
        if fixed is not None:
            return fixed
        return " " + _WORDS[tok % len(_WORDS)]

    def decode(self, ids: List[int]) -> str:
        return "".join(self.decode_token(t) for t in ids).lstrip()


class HFTokenizer:
    """Adapter over a ``tokenizer.json`` (Hugging Face ``tokenizers``).  A ``tokenizer_config.json`` beside it
    supplies the chat template (Jinja, rendered in a sandbox: the file comes with the checkpoint) and the BOS / EOS
    strings; without one the format is chosen by the control tokens the vocabulary has (ChatML, Llama-3 headers,
    ``[INST]``), else plain ``role: content`` lines."""

    def __init__(self, path: str, bos_id: int, eos_id: int):
        import json
        import os

        from tokenizers import Tokenizer

        self.tk = Tokenizer.from_file(path)
        self.vocab_size = self.tk.get_vocab_size()
        self.bos_id, self.eos_id = bos_id, eos_id
        self._pieces: Dict[int, str] = {}
        self._anchor: Optional[int] = None
        self._anchor_text = ""
        self.chat_template: Optional[str] = None
        self.bos_token = self.tk.id_to_token(bos_id) or ""
        self.eos_token = self.tk.id_to_token(eos_id) or ""
        cfg_path = os.path.join(os.path.dirname(os.path.abspath(path)), "tokenizer_config.json")
        if os.path.exists(cfg_path):
            with open(cfg_path) as f:
                tc = json.load(f)
            tpl = tc.get("chat_template")
            if isinstance(tpl, list):  # several named templates: the default one
                tpl = next((t.get("template") for t in tpl if t.get("name") == "default"), tpl[0].get("template") if tpl else None)
            self.chat_template = tpl if isinstance(tpl, str) else None

            def tok_str(v: Any, default: str) -> str:
                return v.get("content", default) if isinstance(v, dict) else (v if isinstance(v, str) else default)

            self.bos_token, self.eos_token = tok_str(tc.get("bos_token"), self.bos_token), tok_str(tc.get("eos_token"), self.eos_token)
        vocab = self.tk.get_vocab()
        self._style = "chatml" if "<|im_start|>" in vocab else "llama3" if "<|start_header_id|>" in vocab else "inst" if "[INST]" in vocab else "plain"

    def encode(self, text: str, add_bos: bool = False) -> List[int]:
        ids = self.tk.encode(text, add_special_tokens=False).ids
        return ([self.bos_id] + ids) if add_bos else ids

    def encode_pieces(self, text: str) -> List[Any]:
        """``[(token id, text)]`` whose texts concatenate to ``text`` (by the encoding's offsets)."""
        enc = self.tk.encode(text, add_special_tokens=False)
        return [(i, text[a:b]) for i, (a, b) in zip(enc.ids, enc.offsets)]

    def render_chat(self, messages: List[Dict[str, Any]]) -> str:
        """The prompt text of a conversation, ending where the assistant's reply begins."""
        if self.chat_template:
            try:
                from jinja2.sandbox import ImmutableSandboxedEnvironment

                def raise_exception(msg: str) -> None:
                    raise ValueError(msg)

                env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
                env.globals["raise_exception"] = raise_exception
                return env.from_string(self.chat_template).render(messages=messages, add_generation_prompt=True, bos_token=self.bos_token, eos_token=self.eos_token)
            except ImportError:
                pass
        def text(m: Dict[str, Any]) -> str:
            c = m.get("content", "")
            return c if isinstance(c, str) else "".join(p.get("text", "") for p in c if isinstance(p, dict))

        if self._style == "chatml":
            return "".join(f"<|im_start|>{m.get('role', 'user')}\n{text(m)}<|im_end|>\n" for m in messages) + "<|im_start|>assistant\n"
        if self._style == "llama3":
            return ("<|begin_of_text|>" + "".join(f"<|start_header_id|>{m.get('role', 'user')}<|end_header_id|>\n\n{text(m)}<|eot_id|>" for m in messages)
                    + "<|start_header_id|>assistant<|end_header_id|>\n\n")
        if self._style == "inst":
            out, system = self.bos_token, ""
            for m in messages:
                role = m.get("role", "user")
                if role == "system":
                    system = text(m) + "\n\n"
                elif role == "user":
                    out += f"[INST] {system}{text(m)} [/INST]"
                    system = ""
                else:
                    out += f" {text(m)}{self.eos_token}"
            return out
        return "".join(f"{m.get('role', 'user')}: {text(m)}\n" for m in messages) + "assistant:"

    def apply_chat_template(self, messages: List[Dict[str, Any]]) -> List[int]:
        return self.tk.encode(self.render_chat(messages), add_special_tokens=False).ids

    def decode_token(self, tok: int) -> str:
        """The text one token adds to a reply.  Decoded behind an anchor token: decoders that strip the space in
        front of a text's first word (SentencePiece-style vocabularies) would otherwise glue streamed words together."""
        s = self._pieces.get(tok)
        if s is None:
            if self._anchor is None:
                ids = self.tk.encode("a", add_special_tokens=False).ids
                self._anchor = ids[-1] if ids else -1
                self._anchor_text = self.tk.decode([self._anchor], skip_special_tokens=True) if ids else ""
            s = self.tk.decode([tok], skip_special_tokens=True)
            if self._anchor >= 0 and self._anchor_text:
                both = self.tk.decode([self._anchor, tok], skip_special_tokens=True)
                if both.startswith(self._anchor_text):
                    s = both[len(self._anchor_text):]
            self._pieces[tok] = s
        return s

    def decode(self, ids: List[int]) -> str:
        return self.tk.decode(ids, skip_special_tokens=True)


def get_tokenizer(vocab_size: int, bos_id: int = 1, eos_id: int = 2, path: Optional[str] = None) -> Any:
    if path:
        return HFTokenizer(path, bos_id, eos_id)
    return SyntheticTokenizer(vocab_size, bos_id, eos_id)
