"""Paged KV-cache block manager with hash-chained prefix caching.

Blocks hold ``block_size`` tokens.  A *full* block whose content is the token
chain ``(parent_hash, tokens)`` is registered in a hash table when it is computed;
a later sequence whose prompt starts with the same chain re-uses the block
(ref-counted) and skips its prefill.  Freed cached blocks go to an LRU list and
are evicted only when a fresh block is needed.  The cache-probe harness measures
exactly this effect (TTFT of repeated vs unique prompts; reference
``cache-probe.sh:76-135,269-280``); ``hit``/``query`` counters feed the server's
``/metrics`` and the analyzer's cache-hit ratio.
"""

from __future__ import annotations

from array import array
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int, enable_prefix_caching: bool = True):
        if num_blocks < 2:
            raise ValueError("need at least 2 KV blocks")
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.enable_prefix_caching = enable_prefix_caching
        # block 0 is the scratch block padding tokens point at; never handed out
        self._free: List[int] = list(range(num_blocks - 1, 0, -1))
        self._ref: List[int] = [0] * num_blocks
        self._hash_of: Dict[int, int] = {}  # block -> chain hash
        self._cached: Dict[int, int] = {}  # chain hash -> block
        self._content: Dict[int, Tuple[int, bytes]] = {}  # block -> (parent key, its tokens): what a lookup is verified against
        self._lru: "OrderedDict[int, None]" = OrderedDict()  # unreferenced cached blocks
        self.prefix_queries = 0
        self.prefix_hits = 0

    # ------------------------------------------------------------------
    @property
    def num_free(self) -> int:
        return len(self._free) + len(self._lru)

    @staticmethod
    def chain_hash(parent: int, tokens: Sequence[int]) -> int:
        """Key of a block in the hash table: (parent key, the block's tokens as bytes).  ``hash`` of a
        bytes object is SipHash under the interpreter's per-process random key, so keys cannot be
        predicted from outside; a key alone never serves a block — see ``_same``."""
        return hash((parent, array("q", tokens).tobytes()))

    def _chain(self, token_ids: Sequence[int], first: int, last: int, parent: int):
        """(key, tokens as bytes) of blocks ``first .. last - 1`` of ``token_ids`` given the key of
        block ``first - 1`` (0 before the first block): the ids become bytes once, every block is
        one slice of them."""
        bs8 = self.block_size * 8
        buf = array("q", token_ids[first * self.block_size : last * self.block_size]).tobytes()
        for i in range(last - first):
            raw = buf[i * bs8 : (i + 1) * bs8]
            key = hash((parent, raw))
            yield key, parent, raw
            parent = key

    def _same(self, block: int, parent: int, raw: bytes) -> bool:
        """A block found under a key is served to another request as that request's own KV, so the
        key is only a lookup: the block must hold exactly these tokens behind exactly this parent
        chain (compared, not hashed — by induction over the chain the whole prefix is equal)."""
        return self._content.get(block) == (parent, raw)

    def _pop_block(self) -> Optional[int]:
        if self._free:
            return self._free.pop()
        if self._lru:
            blk, _ = self._lru.popitem(last=False)  # evict least-recently-used cached block
            h = self._hash_of.pop(blk, None)
            self._content.pop(blk, None)
            if h is not None and self._cached.get(h) == blk:
                del self._cached[h]
            return blk
        return None

    def can_allocate(self, n: int) -> bool:
        return self.num_free >= n

    def allocate(self, n: int) -> Optional[List[int]]:
        if not self.can_allocate(n):
            return None
        out = []
        for _ in range(n):
            b = self._pop_block()
            assert b is not None
            self._ref[b] = 1
            out.append(b)
        return out

    def free(self, blocks: Sequence[int]) -> None:
        for b in reversed(list(blocks)):
            self._ref[b] -= 1
            assert self._ref[b] >= 0
            if self._ref[b] == 0:
                if b in self._hash_of and self.enable_prefix_caching:
                    self._lru[b] = None
                else:
                    self._hash_of.pop(b, None)
                    self._content.pop(b, None)
                    self._free.append(b)

    # prefix cache -------------------------------------------------------
    def match_prefix(self, token_ids: Sequence[int]) -> Tuple[List[int], int]:
        """Longest cached chain of full blocks that prefixes ``token_ids``; takes
        a reference on each returned block.  At least one token is always left to
        compute (the last prompt token produces the first output token)."""
        if not self.enable_prefix_caching:
            return [], 0
        self.prefix_queries += 1
        bs = self.block_size
        blocks: List[int] = []
        parent = 0
        max_full = (len(token_ids) - 1) // bs
        # (the loop of a 100k-token prompt runs 6000+ times per request: everything in locals, no calls; the ids become
        # bytes 512 blocks at a time, so a prompt that misses early does not pay for the conversion of all of it)
        bs8, step = bs * 8, 512
        cached, content, ref, lru = self._cached, self._content, self._ref, self._lru
        done = False
        for i0 in range(0, max_full, step):
            n = min(step, max_full - i0)
            buf = array("q", token_ids[i0 * bs : (i0 + n) * bs]).tobytes()
            for i in range(n):
                raw = buf[i * bs8 : (i + 1) * bs8]
                h = hash((parent, raw))
                b = cached.get(h)
                if b is None or content.get(b) != (parent, raw):  # the key finds, the content decides (see _same)
                    done = True
                    break
                if ref[b] == 0:
                    lru.pop(b, None)
                ref[b] += 1
                blocks.append(b)
                parent = h
            if done:
                break
        if blocks:
            self.prefix_hits += 1
        return blocks, len(blocks) * bs

    def register_full_blocks(self, token_ids: Sequence[int], blocks: Sequence[int], num_computed: int, already: int) -> int:
        """Publish blocks ``[already, num_computed // block_size)`` of a sequence in
        the hash table; returns the new count of published blocks."""
        if not self.enable_prefix_caching:
            return already
        bs = self.block_size
        n_full = num_computed // bs
        if n_full <= already:
            return already
        # chain hash of the last published block: remembered per canonical block; only a
        # sequence whose previous block duplicated an already cached one recomputes the chain
        parent = self._hash_of.get(blocks[already - 1]) if already > 0 else 0
        if parent is None:
            parent = 0
            for parent, _p, _r in self._chain(token_ids, 0, already, 0):
                pass
        for i, (h, par, raw) in enumerate(self._chain(token_ids, already, n_full, parent), already):
            b = blocks[i]
            if h not in self._cached:
                self._cached[h] = b
                self._hash_of[b] = h
                self._content[b] = (par, raw)
        return n_full

    def reset_prefix_cache(self) -> None:
        for b in list(self._lru):
            self._free.append(b)
        self._lru.clear()
        self._cached.clear()
        self._hash_of.clear()
        self._content.clear()
