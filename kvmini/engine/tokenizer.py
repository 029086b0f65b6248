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
    """Thin adapter over a ``tokenizer.json`` (Hugging Face ``tokenizers``)."""

    def __init__(self, path: str, bos_id: int, eos_id: int):
        from tokenizers import Tokenizer

        self.tk = Tokenizer.from_file(path)
        self.vocab_size = self.tk.get_vocab_size()
        self.bos_id, self.eos_id = bos_id, eos_id

    def encode(self, text: str, add_bos: bool = False) -> List[int]:
        ids = self.tk.encode(text, add_special_tokens=False).ids
        return ([self.bos_id] + ids) if add_bos else ids

    def encode_pieces(self, text: str) -> List[Any]:
        """``[(token id, text)]`` whose texts concatenate to ``text`` (by the encoding's offsets)."""
        enc = self.tk.encode(text, add_special_tokens=False)
        return [(i, text[a:b]) for i, (a, b) in zip(enc.ids, enc.offsets)]

    def apply_chat_template(self, messages: List[Dict[str, Any]]) -> List[int]:
        parts = ["<|begin_of_text|>"]
        for m in messages:
            parts.append(f"<|start_header_id|>{m.get('role', 'user')}<|end_header_id|>\n\n{m.get('content', '')}<|eot_id|>")
        parts.append("<|start_header_id|>assistant<|end_header_id|>\n\n")
        return self.tk.encode("".join(parts), add_special_tokens=False).ids

    def decode_token(self, tok: int) -> str:
        return self.tk.decode([tok], skip_special_tokens=True)

    def decode(self, ids: List[int]) -> str:
        return self.tk.decode(ids, skip_special_tokens=True)


def get_tokenizer(vocab_size: int, bos_id: int = 1, eos_id: int = 2, path: Optional[str] = None) -> Any:
    if path:
        return HFTokenizer(path, bos_id, eos_id)
    return SyntheticTokenizer(vocab_size, bos_id, eos_id)
