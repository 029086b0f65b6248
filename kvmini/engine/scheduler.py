"""Continuous-batching scheduler with chunked prefill over a ragged token batch.

Every engine step runs ONE forward pass over a flat list of tokens: one token
for each decoding sequence plus chunks of prompt tokens for sequences that are
still prefilling, up to ``max_batched_tokens``.  Decodes are scheduled first (no
running request is ever stalled by an arriving prompt), then prefill chunks of
running sequences, then newly admitted sequences — so a new request normally
joins the very next step, which is what keeps time-to-first-token at "one step +
one chunk" under concurrency.  When KV blocks run out the most recently admitted
sequence is preempted (its blocks are freed and it is recomputed later).
"""

from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Deque, Dict, List, Optional, Tuple

from .kv_cache import BlockManager


# stands in ``Sequence.output_ids`` for a token that a launched step has sampled but the host has
# not read back yet (pipelined stepping: the next step is launched first)
PLACEHOLDER = -1
LOGIT_BIAS_MAX = 300  # entries of one request's logit_bias map (OpenAI's limit)


class SeqStatus(Enum):
    WAITING = 0
    RUNNING = 1
    FINISHED = 2


@dataclass
class SamplingParams:
    max_tokens: int = 64
    temperature: float = 0.0
    top_p: float = 1.0
    top_k: int = 0
    seed: Optional[int] = None
    ignore_eos: bool = False
    logprobs: int = 0
    stop_token_ids: List[int] = field(default_factory=list)
    # OpenAI's penalties over the tokens generated so far (logit -= presence * [count > 0] + frequency * count) and the
    # multiplicative repetition penalty of vLLM / HF over prompt + generated tokens (positive logits / r, negative * r)
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    repetition_penalty: float = 1.0
    # OpenAI's logit_bias (token id -> value added to its logit, -100 .. 100) and vLLM's min_tokens (EOS and the stop
    # tokens cannot be picked before this many tokens were generated); applied after the penalties, before masks and picks
    logit_bias: Optional[Dict[int, float]] = None
    min_tokens: int = 0
    # guided decoding: "json" constrains the output to one JSON object (kvmini/engine/guided.py);
    # guided_start = (state, depth, stack) of the automaton at the first generated token (a server
    # that pre-writes part of the document into the prompt starts it further in)
    guided: Optional[str] = None
    guided_start: Tuple[int, int, int] = (1, 0, 0)
    # skeleton script of a schema-guided sequence (kvmini.engine.guided.schema_script): forced
    # literal tokens (>= 0), typed value slots (<= -2), -1 at the end
    guided_script: Optional[Tuple[int, ...]] = None

    @property
    def penalised(self) -> bool:
        return self.presence_penalty != 0.0 or self.frequency_penalty != 0.0 or self.repetition_penalty != 1.0 or bool(self.logit_bias) or self.min_tokens > 0


@dataclass(eq=False)  # identity semantics: sequences live in lists / sets and are compared per step
class Sequence:
    req_id: int
    prompt_ids: List[int]
    params: SamplingParams
    arrival: float = field(default_factory=time.perf_counter)
    output_ids: List[int] = field(default_factory=list)
    status: SeqStatus = SeqStatus.WAITING
    num_computed: int = 0  # tokens whose K/V are in the cache
    blocks: List[int] = field(default_factory=list)
    blocks_epoch: int = 0  # bumped whenever ``blocks`` is replaced (not merely extended)
    published_blocks: int = 0
    slot: int = -1  # row in the block-table tensor
    cached_prefix_tokens: int = 0
    first_token_time: Optional[float] = None
    finish_reason: Optional[str] = None
    user: Any = None  # opaque handle for the front-end (stream queue, ...)
    # speculative decoding state: draft tokens proposed for the next verify step
    draft_ids: List[int] = field(default_factory=list)
    draft_feed: bool = False  # a draft runner is proposing launch-ahead: the last draft id may be a placeholder (fed on the device)
    preemptions: int = 0
    # pipelined stepping: row of this sequence's sample in the step in flight; set when the
    # in-flight sample is known (by length) to be the sequence's last
    pending_row: int = -1
    pending_finish: bool = False
    # per output token (requests with logprobs > 0): (log-probability of the token, [(id, logprob)] of the top ones)
    logprob_log: List[Any] = field(default_factory=list)

    @property
    def all_ids(self) -> List[int]:
        return self.prompt_ids + self.output_ids

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_ids) + len(self.output_ids)

    @property
    def is_prefilling(self) -> bool:
        return self.num_computed < len(self.prompt_ids)


@dataclass
class ScheduledChunk:
    seq: Sequence
    start: int  # first position computed in this step
    n_tokens: int  # number of positions computed in this step
    sample: bool  # the chunk ends at the sequence's last known token -> sample next
    # trailing positions to sample; -1 = the last one, or with draft tokens to verify
    # every position from the last accepted token on
    n_sample: int = -1
    out_index: int = -1  # pipelined stepping: index in ``output_ids`` of the placeholder this chunk fills


@dataclass
class Batch:
    chunks: List[ScheduledChunk]
    num_tokens: int
    # set by the engine for a step launched ahead that carries rows with logprobs: where their values land on the host
    logprob_rows: Any = None

    @property
    def num_samples(self) -> int:
        return sum(1 for c in self.chunks if c.sample)


class Scheduler:
    def __init__(
        self,
        block_manager: BlockManager,
        max_num_seqs: int = 64,
        max_batched_tokens: int = 256,
        max_model_len: int = 8192,
        eos_token_id: int = 2,
        ride_along_limit: int = 0,
    ):
        self.bm = block_manager
        self.max_num_seqs = max_num_seqs
        self.max_batched_tokens = max_batched_tokens
        self.max_model_len = max_model_len
        self.eos_token_id = eos_token_id
        # While sequences are decoding, a SMALL backlog of prompt tokens (<= this many; 0 = off)
        # is fed to them in steps that stay within the weight-streaming GEMM's row limits (64
        # tokens, or 128 next to more than 48 decodes) instead of one wide step: the prompts ride
        # along almost for free (the weights are streamed for the decodes anyway), the running
        # requests see no long step, and a burst of arrivals is spread over consecutive steps
        # rather than kept in lock-step.  Larger backlogs (long prompts) take the full budget.
        self.ride_along_limit = ride_along_limit
        self.waiting: Deque[Sequence] = deque()
        self.running: List[Sequence] = []
        self._free_slots: List[int] = list(range(max_num_seqs - 1, -1, -1))
        self.num_preemptions = 0
        # sequences the scheduler gave up on by itself (finish reason "error: ..."); the engine
        # drains this list into token events
        self.failed: List[Sequence] = []
        # False on an engine whose steps do not run the guided sampler (speculative decoding verifies
        # drafts against unmasked logits): such a request is refused at intake instead of running unconstrained
        self.guided_supported = True
        # token ids a request may name (logit_bias, stop tokens); the engine sets it to the model's vocabulary size
        self.vocab_size: Optional[int] = None

    # ------------------------------------------------------------------
    def validate(self, seq: Sequence) -> Optional[str]:
        """Why this request can never be served (None = it can).  Checked at intake so that a
        malformed request fails alone instead of raising inside a step that carries others."""
        import math

        p = seq.params
        if not seq.prompt_ids:
            return "empty prompt"
        if len(seq.prompt_ids) >= self.max_model_len:
            return f"prompt has {len(seq.prompt_ids)} tokens; max_model_len is {self.max_model_len}"
        if p.seed is not None and (isinstance(p.seed, bool) or not isinstance(p.seed, int)):
            return f"seed must be an integer, got {p.seed!r}"
        for name in ("temperature", "top_p"):
            v = getattr(p, name)
            if not isinstance(v, (int, float)) or isinstance(v, bool) or not math.isfinite(v) or v < 0:
                return f"{name} must be a finite non-negative number, got {v!r}"
        for name, lo, hi in (("presence_penalty", -2.0, 2.0), ("frequency_penalty", -2.0, 2.0), ("repetition_penalty", 0.01, 100.0)):
            v = getattr(p, name)
            if not isinstance(v, (int, float)) or isinstance(v, bool) or not math.isfinite(v) or not lo <= v <= hi:
                return f"{name} must be a number in [{lo}, {hi}], got {v!r}"
        if p.logit_bias is not None:
            lb = p.logit_bias
            if not isinstance(lb, dict) or len(lb) > LOGIT_BIAS_MAX:
                return f"logit_bias must be a map of at most {LOGIT_BIAS_MAX} token ids to numbers"
            for t, v in lb.items():
                if isinstance(t, bool) or not isinstance(t, int) or t < 0 or (self.vocab_size is not None and t >= self.vocab_size):
                    return f"logit_bias names token {t!r}; the vocabulary has ids 0 .. {(self.vocab_size or 0) - 1}"
                if not isinstance(v, (int, float)) or isinstance(v, bool) or not math.isfinite(v) or not -100.0 <= v <= 100.0:
                    return f"logit_bias values must be numbers in [-100, 100], got {v!r}"
        if isinstance(p.min_tokens, bool) or not isinstance(p.min_tokens, int) or p.min_tokens < 0:
            return f"min_tokens must be a non-negative integer, got {p.min_tokens!r}"
        if isinstance(p.max_tokens, int) and p.min_tokens > p.max_tokens:
            return f"min_tokens ({p.min_tokens}) is larger than max_tokens ({p.max_tokens})"
        if p.min_tokens > 0 and self.vocab_size is not None and any(isinstance(t, bool) or not isinstance(t, int) or not 0 <= t < self.vocab_size for t in p.stop_token_ids):
            return "min_tokens needs every stop token id inside the vocabulary"
        if p.penalised and not self.guided_supported:
            return "sampling penalties are not available on an engine that runs speculative decoding"
        if p.guided not in (None, "json"):
            return f"unknown guided decoding mode {p.guided!r}"
        if p.guided and not self.guided_supported:
            return "guided decoding is not available on an engine that runs speculative decoding"
        if p.guided_script is not None:
            from .guided import SCRIPT_END, SCRIPT_MAX

            sc = p.guided_script
            if p.guided != "json" or not 0 < len(sc) <= SCRIPT_MAX or sc[-1] != SCRIPT_END or any(isinstance(e, bool) or not isinstance(e, int) for e in sc):
                return f"bad guided script (integers, at most {SCRIPT_MAX} entries, the last one {SCRIPT_END})"
        if not isinstance(p.max_tokens, int) or p.max_tokens < 1:
            return f"max_tokens must be a positive integer, got {p.max_tokens!r}"
        bs = self.bm.block_size
        n_max = min(len(seq.prompt_ids) + p.max_tokens, self.max_model_len)
        if (n_max + bs - 1) // bs > self.bm.num_blocks - 1:
            return f"request needs {(n_max + bs - 1) // bs} KV blocks of {bs} tokens; the cache has {self.bm.num_blocks - 1}"
        return None

    def add(self, seq: Sequence) -> Optional[str]:
        """Queue a request.  Returns the rejection reason (and does not queue) when the
        request can never run."""
        why = self.validate(seq)
        if why is not None:
            seq.status = SeqStatus.FINISHED
            seq.finish_reason = f"error: {why}"
            return why
        if len(seq.prompt_ids) + seq.params.max_tokens > self.max_model_len:
            seq.params.max_tokens = max(1, self.max_model_len - len(seq.prompt_ids))
        self.waiting.append(seq)
        return None

    def has_work(self) -> bool:
        return bool(self.waiting or self.running)

    def abort(self, req_id: int) -> Optional[Sequence]:
        for s in list(self.running):
            if s.req_id == req_id:
                self._finish(s, "abort")
                return s
        for s in list(self.waiting):
            if s.req_id == req_id:
                self.waiting.remove(s)
                s.status = SeqStatus.FINISHED
                s.finish_reason = "abort"
                return s
        return None

    def _blocks_needed(self, seq: Sequence, upto_tokens: int) -> int:
        bs = self.bm.block_size
        return max(0, (upto_tokens + bs - 1) // bs - len(seq.blocks))

    def _preempt_one(self, keep: Sequence, protected: "set[int]") -> bool:
        """Evict the most recently admitted sequence that is neither ``keep`` nor
        already part of the batch being built."""
        for victim in reversed(self.running):
            if victim is keep or victim.req_id in protected:
                continue
            self.bm.free(victim.blocks)
            victim.blocks = []
            victim.blocks_epoch += 1
            victim.published_blocks = 0
            victim.num_computed = 0
            victim.draft_ids = []
            victim.status = SeqStatus.WAITING
            victim.preemptions += 1
            self._free_slots.append(victim.slot)
            victim.slot = -1
            self.running.remove(victim)
            self.waiting.appendleft(victim)
            self.num_preemptions += 1
            return True
        return False

    def _step_budget(self) -> int:
        """Token budget of the next step (see ``ride_along_limit``)."""
        budget = self.max_batched_tokens
        if self.ride_along_limit <= 0:
            return budget
        n_decode = sum(1 for s in self.running if not s.is_prefilling and not s.pending_finish)
        if n_decode == 0 or n_decode > 112:
            return budget
        backlog = 0
        for s in self.running:
            if s.is_prefilling:
                backlog += len(s.prompt_ids) - s.num_computed
        for s in self.waiting:
            backlog += s.num_tokens - s.num_computed
            if backlog > self.ride_along_limit:
                return budget
        if backlog == 0 or backlog > self.ride_along_limit:
            return budget
        return min(budget, 64 if n_decode <= 48 else 128)

    def schedule(self, spec_tokens: int = 0) -> Batch:
        """Pick the chunks for the next step.  ``spec_tokens`` > 0 lets a decoding
        sequence carry that many draft tokens to verify in the same pass."""
        budget = self._step_budget()
        chunks: List[ScheduledChunk] = []
        scheduled = set()

        # 1) running sequences: decodes first, then in-flight prefills
        order = sorted(self.running, key=lambda s: s.is_prefilling)
        for seq in order:
            if budget <= 0:
                break
            if seq not in self.running:  # preempted while serving an earlier sequence
                continue
            if seq.pending_finish:  # its last token is in flight
                continue
            known = seq.num_tokens + (len(seq.draft_ids) if not seq.is_prefilling else 0)
            n_new = known - seq.num_computed
            if n_new <= 0:
                continue
            n = min(n_new, budget)
            need = self._blocks_needed(seq, seq.num_computed + n)
            while need and not self.bm.can_allocate(need):
                if not self._preempt_one(seq, scheduled):
                    break
            if need:
                got = self.bm.allocate(need)
                if got is None:
                    if len(self.running) == 1 and not chunks:
                        # alone, nothing left to preempt and still no room: it never will be
                        self._finish(seq, f"error: KV cache exhausted ({self.bm.num_blocks} blocks) at {seq.num_computed} tokens")
                        self.failed.append(seq)
                    continue  # cannot run this step
                seq.blocks += got
            chunks.append(ScheduledChunk(seq, seq.num_computed, n, seq.num_computed + n == known))
            scheduled.add(seq.req_id)
            budget -= n

        # 2) admit waiting sequences
        while self.waiting and budget > 0 and len(self.running) < self.max_num_seqs and self._free_slots:
            seq = self.waiting[0]
            if not seq.blocks:
                cached_blocks, cached_tokens = self.bm.match_prefix(seq.all_ids)
                seq.blocks = list(cached_blocks)
                seq.blocks_epoch += 1
                seq.num_computed = cached_tokens
                seq.cached_prefix_tokens = cached_tokens
                seq.published_blocks = len(cached_blocks)
            n_new = seq.num_tokens - seq.num_computed
            n = min(n_new, budget)
            need = self._blocks_needed(seq, seq.num_computed + n)
            if need and not self.bm.can_allocate(need):
                # give the cached prefix back; retry when blocks free up
                self.bm.free(seq.blocks)
                seq.blocks, seq.num_computed, seq.published_blocks = [], 0, 0
                seq.blocks_epoch += 1
                break
            if need:
                seq.blocks += self.bm.allocate(need) or []
            self.waiting.popleft()
            seq.slot = self._free_slots.pop()
            seq.status = SeqStatus.RUNNING
            self.running.append(seq)
            chunks.append(ScheduledChunk(seq, seq.num_computed, n, seq.num_computed + n == seq.num_tokens))
            budget -= n

        return Batch(chunks, sum(c.n_tokens for c in chunks))

    # ------------------------------------------------------------------
    def _finish(self, seq: Sequence, reason: str) -> None:
        seq.status = SeqStatus.FINISHED
        seq.finish_reason = reason
        self.bm.free(seq.blocks)
        seq.blocks = []
        if seq.slot >= 0:
            self._free_slots.append(seq.slot)
            seq.slot = -1
        if seq in self.running:
            self.running.remove(seq)

    # --- pipelined stepping: account for a step at LAUNCH, fill in its tokens later ----------
    def launch_commit(self, batch: Batch) -> None:
        """Advance every sequence of a step that has just been enqueued as if it had finished:
        its positions count as computed and a sampled token is a ``PLACEHOLDER`` until
        :meth:`finalize`.  A sequence that reaches its length limit with this token is not
        scheduled again (``pending_finish``); it keeps its blocks and slot until finalized."""
        row = 0
        bs = self.bm.block_size
        for ch in batch.chunks:
            seq = ch.seq
            seq.num_computed = ch.start + ch.n_tokens
            # blocks whose tokens are all known (prompt chunks above all) are published now: a
            # request admitted by the very next launch may reuse them, its step runs behind this
            # one on the stream
            known = seq.num_tokens - (1 if seq.output_ids and seq.output_ids[-1] == PLACEHOLDER else 0)
            n_reg = min(seq.num_computed, known)
            if n_reg // bs > seq.published_blocks:
                seq.published_blocks = self.bm.register_full_blocks(seq.all_ids, seq.blocks, n_reg, seq.published_blocks)
            if not ch.sample:
                continue
            seq.pending_row = row
            row += 1
            ch.out_index = len(seq.output_ids)
            seq.output_ids.append(PLACEHOLDER)
            if len(seq.output_ids) >= seq.params.max_tokens or seq.num_tokens >= self.max_model_len:
                seq.pending_finish = True

    def finalize(self, chunk: ScheduledChunk, token: Optional[int]) -> List[int]:
        """The launched step of ``chunk`` has finished and ``token`` is its sample (None for a
        chunk that did not sample).  Returns the tokens appended.  A stop token ends the
        sequence here, one step late: the token a later launch computed for it meanwhile is
        dropped (its KV write went to blocks the sequence still owned)."""
        seq = chunk.seq
        if seq.status != SeqStatus.RUNNING:  # aborted while the step was in flight
            return []
        n_done = chunk.start + chunk.n_tokens  # every token below this position is known by now
        if chunk.sample:
            assert token is not None
            seq.output_ids[chunk.out_index] = token
        if n_done // self.bm.block_size > seq.published_blocks:
            seq.published_blocks = self.bm.register_full_blocks(seq.all_ids, seq.blocks, n_done, seq.published_blocks)
        if not chunk.sample:
            return []
        if seq.first_token_time is None:
            seq.first_token_time = time.perf_counter()
        n_out = chunk.out_index + 1
        if (not seq.params.ignore_eos and token == self.eos_token_id) or token in seq.params.stop_token_ids:
            del seq.output_ids[n_out:]
            seq.pending_finish = False
            self._finish(seq, "stop")
        elif n_out >= seq.params.max_tokens or len(seq.prompt_ids) + n_out >= self.max_model_len:
            seq.pending_finish = False
            self._finish(seq, "length")
        return [token]

    def commit(self, chunk: ScheduledChunk, new_tokens: List[int], num_accepted_positions: Optional[int] = None) -> List[int]:
        """Account for a finished step.  ``new_tokens`` are the tokens to append
        (1 for plain decoding, up to k+1 with speculative verification); returns
        the tokens actually appended (after max_tokens / EOS truncation)."""
        seq = chunk.seq
        seq.num_computed = chunk.start + (chunk.n_tokens if num_accepted_positions is None else num_accepted_positions)
        n_done = min(seq.num_computed, seq.num_tokens)
        if n_done // self.bm.block_size > seq.published_blocks:  # a block filled up in this step
            seq.published_blocks = self.bm.register_full_blocks(seq.all_ids, seq.blocks, n_done, seq.published_blocks)
        appended: List[int] = []
        if not chunk.sample:
            return appended
        for tok in new_tokens:
            seq.output_ids.append(tok)
            appended.append(tok)
            if seq.first_token_time is None:
                seq.first_token_time = time.perf_counter()
            stop = (not seq.params.ignore_eos and tok == self.eos_token_id) or tok in seq.params.stop_token_ids
            if stop:
                self._finish(seq, "stop")
                break
            if len(seq.output_ids) >= seq.params.max_tokens or seq.num_tokens >= self.max_model_len:
                self._finish(seq, "length")
                break
        return appended
