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
        self._lru: "OrderedDict[int, None]" = OrderedDict()  # unreferenced cached blocks
        self.prefix_queries = 0
        self.prefix_hits = 0

    # ------------------------------------------------------------------
    @property
    def num_free(self) -> int:
        return len(self._free) + len(self._lru)

    @staticmethod
    def chain_hash(parent: int, tokens: Sequence[int]) -> int:
        return hash((parent, tuple(tokens)))

    def _pop_block(self) -> Optional[int]:
        if self._free:
            return self._free.pop()
        if self._lru:
            blk, _ = self._lru.popitem(last=False)  # evict least-recently-used cached block
            h = self._hash_of.pop(blk, None)
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
        for i in range(max_full):
            h = self.chain_hash(parent, token_ids[i * bs : (i + 1) * bs])
            b = self._cached.get(h)
            if b is None:
                break
            if self._ref[b] == 0:
                self._lru.pop(b, None)
            self._ref[b] += 1
            blocks.append(b)
            parent = h
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
            for i in range(already):
                parent = self.chain_hash(parent, token_ids[i * bs : (i + 1) * bs])
        for i in range(already, n_full):
            h = self.chain_hash(parent, token_ids[i * bs : (i + 1) * bs])
            b = blocks[i]
            if h not in self._cached:
                self._cached[h] = b
                self._hash_of[b] = h
            parent = h
        return n_full

    def reset_prefix_cache(self) -> None:
        for b in list(self._lru):
            self._free.append(b)
        self._lru.clear()
        self._cached.clear()
        self._hash_of.clear()
