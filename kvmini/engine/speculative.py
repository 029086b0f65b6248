"""Draft-model speculative decoding: greedy verification, and rejection sampling for
requests that sample.

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
valid in both caches.

Requests with temperature > 0 are verified by speculative *rejection sampling* (Leviathan et
al., "Fast Inference from Transformers via Speculative Decoding", 2023; Chen et al., 2023): the
draft token x_i is drawn from the draft's distribution q_i (after the request's temperature /
top-k / top-p), the target's distribution p_i at the same position comes out of the one verify
pass, x_i is kept with probability min(1, p_i(x_i) / q_i(x_i)); the first rejected position is
re-drawn from normalise(max(0, p_i - q_i)), and if nothing is rejected a bonus token is drawn
from p_k.  The tokens that come out are distributed exactly as if the target had sampled them
one by one.

The reference exposes speculation only as vLLM flags in a profile
(``runners/profiles/speculative-decoding.yaml``); this is the native-engine
counterpart, reported through ``spec_proposed`` / ``spec_accepted``.
"""

from __future__ import annotations

import time
from dataclasses import replace
from typing import TYPE_CHECKING, Dict, List, Optional, Tuple

import torch

from kvmini.models.llama import LlamaWeights, get_config, random_weights

from .runner import ModelRunner, RunnerConfig
from .scheduler import Batch, ScheduledChunk, SeqStatus, Sequence

if TYPE_CHECKING:
    from .engine import LLMEngine, TokenEvent


def rejection_sample(p: torch.Tensor, q: torch.Tensor, drafts: torch.Tensor, gen: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One sequence's verification.  ``p``: target distributions ``[k + 1, V]`` at the k draft
    positions and the one after them, ``q``: draft distributions ``[k, V]``, ``drafts``: the k
    draft tokens (int64), each drawn from its ``q`` row.  Returns (number of accepted drafts a,
    the token that follows them) as 0-dim device tensors — nothing is read back here."""
    k = drafts.numel()
    idx = torch.arange(k, device=p.device)
    px, qx = p[idx, drafts], q[idx, drafts]
    u = torch.rand(k, device=p.device, generator=gen)
    ok = u * qx < px  # u < p/q without the division (q(x) > 0: x was drawn from q)
    a = torch.cumprod(ok.to(torch.int64), 0).sum()  # leading accepts
    # distribution of the token after the accepted prefix: residual at the rejected position, p_k if none was
    resid = torch.clamp(p[:k] - q, min=0.0)
    resid = torch.cat((resid, p[k:]), 0)  # row k: the bonus position
    row = resid[a]
    tot = row.sum()
    row = torch.where(tot > 0, row / tot.clamp_min(1e-30), p[a])  # p == q there (nothing left over): any draw from p is right
    return a, torch.multinomial(row, 1, generator=gen).view(())


class SpeculativeDecoder:
    def __init__(self, engine: "LLMEngine", draft_model: str, k: int):
        self.engine = engine
        self.k = int(k)
        # greedy proposals are enqueued back to back, each fed by its predecessor on the device
        # (KVMINI_SPEC_LAUNCH_AHEAD=0: one read-back per proposal, the A/B)
        import os

        self.launch_ahead = os.environ.get("KVMINI_SPEC_LAUNCH_AHEAD", "1") == "1"
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
            from kvmini.models.llama import load_hf_safetensors, resolve_model

            dcfg, dpath = resolve_model(draft_model)  # a preset name (random-init draft) or a checkpoint directory
            if dcfg.vocab_size != engine.mcfg.vocab_size:
                raise ValueError(f"draft vocab {dcfg.vocab_size} != target vocab {engine.mcfg.vocab_size}")
            if dpath:
                dw = load_hf_safetensors(dpath, dcfg, engine.device, engine.weights.embed.dtype)
            else:
                dw = random_weights(dcfg, engine.device, engine.weights.embed.dtype, ecfg.seed + 1)
        rcfg = RunnerConfig(max_num_seqs=ecfg.max_num_seqs, max_batched_tokens=ecfg.max_batched_tokens, max_model_len=ecfg.max_model_len, block_size=ecfg.block_size,
                            num_blocks=tgt.num_blocks, use_graphs=ecfg.use_graphs, kernels=ecfg.kernels)
        self.draft = ModelRunner(dw, rcfg)
        tgt.set_spec_width(self.k + 1)
        # positions whose K/V are valid in the draft cache, per request id
        self.draft_computed: Dict[int, int] = {}
        # sampling requests: the draft distributions their draft tokens were drawn from, [k, V] per request id
        self.draft_q: Dict[int, torch.Tensor] = {}

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
        """k draft steps for every sequence in ``seqs`` (batched across sequences).  Greedy
        requests take the draft's argmax; sampling requests draw from the draft's distribution
        (all of a step's draws in one batched call) and keep that distribution for the verifier."""
        from .engine import _probs_rows

        for s in seqs:
            s.draft_ids = []
        sampled = [i for i, s in enumerate(seqs) if s.params.temperature > 0]
        if not sampled and self.launch_ahead:
            return self._propose_launch_ahead(seqs)
        qs: List[torch.Tensor] = []
        for _ in range(self.k):
            chunks = []
            total = 0
            for s in seqs:
                start = self.draft_computed[s.req_id]
                n = s.num_tokens + len(s.draft_ids) - start
                chunks.append(ScheduledChunk(s, start, n, True, 1))
                total += n
            ids, logits = self.draft.execute(Batch(chunks, total))
            if sampled:
                q = _probs_rows(logits[sampled], [seqs[i].params for i in sampled])
                gens = [self.engine.generator_for(seqs[i]) for i in sampled]
                if any(g is not None for g in gens):  # seeded requests own their generator
                    drawn = torch.stack([torch.multinomial(q[j], 1, generator=g).view(()) for j, g in enumerate(gens)]).tolist()
                else:
                    drawn = torch.multinomial(q, 1).view(-1).tolist()
                qs.append(q)
                for i, tok in zip(sampled, drawn):
                    ids[i] = tok
            for s, tok in zip(seqs, ids):
                self.draft_computed[s.req_id] = s.num_tokens + len(s.draft_ids)
                s.draft_ids.append(int(tok))
        if sampled:
            stack = torch.stack(qs, 1)  # [sampled, k, V]
            for j, i in enumerate(sampled):
                self.draft_q[seqs[i].req_id] = stack[j]

    def _propose_launch_ahead(self, seqs: List[Sequence]) -> None:
        """The k greedy draft steps enqueued back to back: proposal j + 1 is launched on top of
        proposal j, whose token it takes on the device (a placeholder in ``draft_ids`` until the
        host reads it back one launch later), so the device never waits for the host between
        proposals — one exposed read-back per propose phase (the last one) instead of k."""
        from .scheduler import PLACEHOLDER

        prev = None
        try:
            for j in range(self.k):
                chunks = []
                total = 0
                for s in seqs:
                    s.draft_feed = j > 0
                    start = self.draft_computed[s.req_id]
                    n = s.num_tokens + len(s.draft_ids) - start
                    chunks.append(ScheduledChunk(s, start, n, True, 1))
                    total += n
                handle = self.draft.launch(Batch(chunks, total))
                for row, s in enumerate(seqs):
                    self.draft_computed[s.req_id] = s.num_tokens + len(s.draft_ids)
                    s.pending_row = row
                    s.draft_ids.append(PLACEHOLDER)
                if prev is not None:
                    ids, _ = self.draft.collect(prev)
                    for s, tok in zip(seqs, ids):
                        s.draft_ids[j - 1] = int(tok)
                prev = handle
            ids, _ = self.draft.collect(prev)
            for s, tok in zip(seqs, ids):
                s.draft_ids[self.k - 1] = int(tok)
        finally:
            for s in seqs:
                s.draft_feed = False
                s.pending_row = -1

    # ------------------------------------------------------------------
    def step(self) -> "List[TokenEvent]":
        eng = self.engine
        sch = eng.scheduler
        t0 = time.perf_counter()
        k = self.k
        # sequences that can take k drafts: decoding, greedy, room for k+1 more tokens, draft cache in step
        cands = [
            s for s in sch.running
            if not s.is_prefilling and not s.params.logprobs and not s.params.guided and s.req_id in self.draft_computed
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
        from .engine import _probs_rows, _sample_row

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
                q = self.draft_q.pop(seq.req_id, None)
                if seq.params.temperature > 0 and q is not None and logits is not None and n_rows == len(drafts) + 1:
                    # rejection sampling: rows of `logits` are the target's outputs after (last accepted, d0..d_{i-1})
                    p = _probs_rows(logits[row : row + n_rows], [seq.params] * n_rows)
                    a_t, tok_t = rejection_sample(p, q, torch.tensor(drafts, dtype=torch.int64, device=p.device), eng.generator_for(seq))
                    a, nxt = (int(v) for v in torch.stack((a_t, tok_t)).tolist())
                    new_tokens = list(drafts[:a]) + [nxt]
                else:
                    # greedy: rows[i] = target's argmax after (last accepted, d0..d_{i-1})
                    a = 0
                    while a < len(drafts) and a < len(rows) - 1 and drafts[a] == rows[a]:
                        a += 1
                    nxt = int(rows[a])
                    if seq.params.temperature > 0 and logits is not None:  # (a verify chunk the budget cut short)
                        a = 0
                        nxt = _sample_row(logits[row], seq.params, eng.generator_for(seq))
                    new_tokens = list(drafts[:a]) + [nxt]
                row += n_rows
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
                    tok = _sample_row(logits[row], p, eng.generator_for(seq))
                row += 1
                new = sch.commit(ch, [int(tok)])
                self.draft_computed.setdefault(seq.req_id, min(len(seq.prompt_ids), seq.num_computed))
            eng.stats.generation_tokens += len(new)
            done = seq.status == SeqStatus.FINISHED
            if done:
                eng.stats.requests_finished += 1
                self.draft_computed.pop(seq.req_id, None)
                self.draft_q.pop(seq.req_id, None)
                eng._gens.pop(seq.req_id, None)
            events.append((seq.req_id, new, done, seq.finish_reason, seq))
        # forget draft state of sequences that left the running set (abort / preemption)
        live = {s.req_id for s in sch.running}
        for rid in [r for r in self.draft_computed if r not in live]:
            del self.draft_computed[rid]
            self.draft_q.pop(rid, None)
        return events
