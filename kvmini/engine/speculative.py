"""Draft-model speculative decoding with greedy verification.

Each engine iteration for a decoding sequence is *propose* (a small draft model
runs ``k`` cheap autoregressive steps) then *verify* (the target model scores
the last accepted token plus the ``k`` drafts in ONE ragged pass and keeps the
longest prefix it agrees with, plus its own next token).  Output is token-for-
token identical to plain greedy decoding of the target; the win is 1 + (accepted
drafts) tokens per target weight-streaming pass, which on MI355X is what a
decode step costs (≈16 GB of bf16 weights per step for an 8B model, and up to
64 tokens ride one pass of the streaming GEMM for free).

The draft runner has its own KV-cache tensor but shares block ids and slots with
the target (one BlockManager), so no second allocator is needed: a block id is
valid in both caches.  Sequences that sample with temperature > 0 fall back to
one-token steps (rejection sampling is not implemented).

The reference exposes speculation only as vLLM flags in a profile
(``runners/profiles/speculative-decoding.yaml``); this is the native-engine
counterpart, reported through ``spec_proposed`` / ``spec_accepted``.
"""

from __future__ import annotations

import time
from dataclasses import replace
from typing import TYPE_CHECKING, Dict, List

import torch

from kvmini.models.llama import LlamaWeights, get_config, random_weights

from .runner import ModelRunner, RunnerConfig
from .scheduler import Batch, ScheduledChunk, SeqStatus, Sequence

if TYPE_CHECKING:
    from .engine import LLMEngine, TokenEvent


class SpeculativeDecoder:
    def __init__(self, engine: "LLMEngine", draft_model: str, k: int):
        self.engine = engine
        self.k = int(k)
        tgt = engine.runner
        ecfg = engine.ecfg
        if ecfg.tp_size > 1:
            raise ValueError("speculative decoding is single-GPU in this engine (the draft is not sharded)")
        if draft_model.startswith("self:"):
            # early-exit self-speculation: the draft is the target's first K layers followed by
            # the target's own final norm and LM head.  It shares the (already packed) weight
            # tensors, so it costs no extra weight memory — only a K-layer KV cache.
            depth = int(draft_model.split(":", 1)[1])
            if not 0 < depth < engine.mcfg.num_layers:
                raise ValueError(f"self-draft depth must be in 1..{engine.mcfg.num_layers - 1}")
            tw = engine.weights
            dw = LlamaWeights(replace(engine.mcfg, name=f"{engine.mcfg.name}-exit{depth}", num_layers=depth), tw.embed, tw.layers[:depth], tw.final_norm, tw.lm_head, tw.tp_rank, tw.tp_size)
            for attr in ("_hip_prepared", "quantization"):
                if hasattr(tw, attr):
                    setattr(dw, attr, getattr(tw, attr))
        else:
            dcfg = get_config(draft_model)
            if dcfg.vocab_size != engine.mcfg.vocab_size:
                raise ValueError(f"draft vocab {dcfg.vocab_size} != target vocab {engine.mcfg.vocab_size}")
            dw = random_weights(dcfg, engine.device, engine.weights.embed.dtype, ecfg.seed + 1)
        rcfg = RunnerConfig(max_num_seqs=ecfg.max_num_seqs, max_batched_tokens=ecfg.max_batched_tokens, max_model_len=ecfg.max_model_len, block_size=ecfg.block_size,
                            num_blocks=tgt.num_blocks, use_graphs=ecfg.use_graphs, kernels=ecfg.kernels)
        self.draft = ModelRunner(dw, rcfg)
        tgt.set_spec_width(self.k + 1)
        # positions whose K/V are valid in the draft cache, per request id
        self.draft_computed: Dict[int, int] = {}

    def load_draft_weights(self, weights) -> None:
        """Swap in real draft weights (same architecture as the random-init ones)."""
        rcfg = self.draft.rcfg
        self.draft = ModelRunner(weights, rcfg)

    def warmup(self) -> None:
        try:
            self.draft.capture_graphs()
        except Exception as e:  # draft shapes the HIP attention kernel does not cover run eagerly
            print(f"[speculative] draft graphs disabled: {e}")
            self.draft.graphs.clear()
            self.draft.rcfg.use_graphs = False

    # ------------------------------------------------------------------
    def _draft_prefill(self, batch: Batch) -> None:
        """Mirror the target's prefill chunks into the draft cache (no sampling)."""
        chunks = []
        for ch in batch.chunks:
            seq = ch.seq
            if ch.start < len(seq.prompt_ids):
                start = min(ch.start, self.draft_computed.get(seq.req_id, seq.cached_prefix_tokens))
                end = min(ch.start + ch.n_tokens, len(seq.prompt_ids))
                if end > start:
                    chunks.append(ScheduledChunk(seq, start, end - start, False))
                    self.draft_computed[seq.req_id] = end
        budget = self.draft.tmax
        group: List[ScheduledChunk] = []
        used = 0
        for ch in chunks:
            # split to the draft runner's token budget
            off = 0
            while off < ch.n_tokens:
                n = min(ch.n_tokens - off, budget - used)
                if n == 0:
                    self.draft.execute(Batch(group, used))
                    group, used = [], 0
                    continue
                group.append(ScheduledChunk(ch.seq, ch.start + off, n, False))
                used += n
                off += n
        if group:
            self.draft.execute(Batch(group, used))

    def _propose(self, seqs: List[Sequence]) -> None:
        """k draft steps for every sequence in ``seqs`` (batched across sequences)."""
        for s in seqs:
            s.draft_ids = []
        for _ in range(self.k):
            chunks = []
            total = 0
            for s in seqs:
                start = self.draft_computed[s.req_id]
                n = s.num_tokens + len(s.draft_ids) - start
                chunks.append(ScheduledChunk(s, start, n, True, 1))
                total += n
            ids, _ = self.draft.execute(Batch(chunks, total))
            for s, tok in zip(seqs, ids):
                self.draft_computed[s.req_id] = s.num_tokens + len(s.draft_ids)
                s.draft_ids.append(int(tok))

    # ------------------------------------------------------------------
    def step(self) -> "List[TokenEvent]":
        eng = self.engine
        sch = eng.scheduler
        t0 = time.perf_counter()
        k = self.k
        # sequences that can take k drafts: decoding, greedy, room for k+1 more tokens, draft cache in step
        cands = [
            s for s in sch.running
            if not s.is_prefilling and s.params.temperature <= 0 and s.req_id in self.draft_computed
            and s.params.max_tokens - len(s.output_ids) > 1 and s.num_tokens + k + 1 <= sch.max_model_len
        ]
        # keep the verify pass within the token budget
        room = max(0, sch.max_batched_tokens - len(sch.running))
        cands = cands[: room // max(1, k)]
        # the draft writes K/V for positions up to num_tokens + k - 1: those blocks must exist
        # before proposing (after a fully accepted step the previous allocation is used up)
        ready = []
        for s in cands:
            need = sch._blocks_needed(s, s.num_tokens + k)
            if need:
                got = sch.bm.allocate(need) if sch.bm.can_allocate(need) else None
                if got is None:
                    continue  # decode this one without drafts in this step
                s.blocks += got
            ready.append(s)
        cands = ready
        with torch.no_grad():
            if cands:
                self._propose(cands)
            batch = sch.schedule(spec_tokens=k)
            if not batch.chunks:
                for s in cands:
                    s.draft_ids = []
                return []
            self._draft_prefill(batch)
            ids, logits = eng.runner.execute(batch)
        events = self._commit(batch, ids, logits)
        dt = time.perf_counter() - t0
        st = eng.stats
        st.steps += 1
        st.busy_s += dt
        st.step_ms_ewma = dt * 1e3 if st.step_ms_ewma == 0 else 0.9 * st.step_ms_ewma + 0.1 * dt * 1e3
        return events

    def _commit(self, batch: Batch, ids: List[int], logits) -> "List[TokenEvent]":
        from .engine import _sample_row

        eng = self.engine
        sch = eng.scheduler
        events = []
        row = 0
        for ch in batch.chunks:
            seq = ch.seq
            if ch.start < len(seq.prompt_ids):
                eng.stats.prompt_tokens += min(ch.n_tokens, len(seq.prompt_ids) - ch.start)
            if not ch.sample:
                if seq.draft_ids and not seq.is_prefilling:
                    # a verify chunk cut short by the token budget: keep only the known tokens
                    seq.draft_ids = []
                    sch.commit(ch, [], num_accepted_positions=max(0, min(ch.n_tokens, seq.num_tokens - 1 - ch.start)))
                else:
                    sch.commit(ch, [])
                continue
            drafts = seq.draft_ids if not seq.is_prefilling else []
            if drafts and ch.start + ch.n_tokens == seq.num_tokens + len(drafts):
                n_rows = seq.num_tokens + len(drafts) - max(ch.start, seq.num_tokens - 1)
                rows = ids[row : row + n_rows]
                row += n_rows
                # rows[i] = target's greedy token after (last accepted, d0..d_{i-1})
                a = 0
                while a < len(drafts) and a < len(rows) - 1 and drafts[a] == rows[a]:
                    a += 1
                new_tokens = list(drafts[:a]) + [int(rows[a])]
                eng.stats.spec_proposed += len(drafts)
                eng.stats.spec_accepted += a
                old_n = seq.num_tokens
                seq.draft_ids = []
                accepted_positions = (old_n - ch.start) + a  # positions start.. through the last accepted draft
                new = sch.commit(ch, new_tokens, num_accepted_positions=accepted_positions)
                # draft cache: inputs were last token + d0..d_{k-2}; valid where input == final token
                self.draft_computed[seq.req_id] = old_n + min(a, max(0, len(drafts) - 1))
            else:
                seq.draft_ids = []
                tok = ids[row]
                p = seq.params
                if p.temperature > 0 and logits is not None:
                    tok = _sample_row(logits[row], p, None)
                row += 1
                new = sch.commit(ch, [int(tok)])
                self.draft_computed.setdefault(seq.req_id, min(len(seq.prompt_ids), seq.num_computed))
            eng.stats.generation_tokens += len(new)
            done = seq.status == SeqStatus.FINISHED
            if done:
                eng.stats.requests_finished += 1
                self.draft_computed.pop(seq.req_id, None)
            events.append((seq.req_id, new, done, seq.finish_reason, seq))
        # forget draft state of sequences that left the running set (abort / preemption)
        live = {s.req_id for s in sch.running}
        for rid in [r for r in self.draft_computed if r not in live]:
            del self.draft_computed[rid]
        return events
