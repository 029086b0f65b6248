"""LLM engine: request intake, scheduling loop, token delivery.

The engine owns one model replica (one GPU, or one TP group) and runs its step
loop on a dedicated thread: drain newly submitted requests, ask the scheduler
for a ragged batch, execute it (one HIP-graph replay), commit the sampled
tokens, and hand every request its new tokens through a callback.  The HTTP
front-end (``kvmini.server.native``) bridges that callback into per-request
asyncio queues with ONE ``call_soon_threadsafe`` per step.
"""

from __future__ import annotations

import itertools
import os
import queue
import threading
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from kvmini.models.llama import LlamaWeights, get_config, random_weights

from .kv_cache import BlockManager
from .runner import ModelRunner, RunnerConfig
from .scheduler import Batch, SamplingParams, Scheduler, Sequence, SeqStatus


@dataclass
class EngineConfig:
    model: str = "tiny"
    device: str = "auto"
    dtype: str = "bf16"
    seed: int = 0
    max_num_seqs: int = 64
    max_batched_tokens: int = 256
    max_model_len: int = 8192
    block_size: int = 16
    num_blocks: int = 0
    kv_memory_fraction: float = 0.5
    use_graphs: bool = True
    kernels: str = "auto"
    enable_prefix_caching: bool = True
    weights_path: Optional[str] = None
    quantization: Optional[str] = None  # None | fp8
    kv_cache_dtype: str = "auto"  # auto | fp8
    # speculative decoding
    draft_model: Optional[str] = None  # preset name, or "self:<K>" = early exit after K layers of the target
    num_speculative_tokens: int = 0
    # random-init only: damp the residual branches of layers >= K by this factor (see
    # kvmini.models.llama.random_weights); K is the early-exit depth of a "self:<K>" draft
    tail_damping: float = 0.0
    # launch step n+1 before reading step n's tokens back (greedy requests, no KV pressure)
    async_scheduling: bool = True
    # pipelined stepping launches the next step this long before the step in flight is expected
    # to end (not right away), so that requests arriving meanwhile still make that launch; 0 =
    # launch immediately
    launch_margin_ms: float = 0.7
    # prompt backlogs up to this many tokens ride along with the running decodes in steps of at
    # most 64 (128) tokens instead of one wide step (see Scheduler.ride_along_limit); 0 = off.
    # Measured on the headline mix (18 prompts arriving together): first token 11.0 -> 9.4 ms, but
    # inter-token p95 3.6 -> 4.2 ms and 0.8 % fewer requests/s, hence off by default
    ride_along_limit: int = 0
    tp_size: int = 1
    tp_rank: int = 0
    tp_graphs: bool = True  # HIP graphs under tensor parallelism (peer-memory all-reduce inside)
    tp_rccl_graphs: bool = False  # also capture the buckets whose all-reduce is RCCL's
    tp_custom_allreduce: bool = True
    tp_force_collectives: bool = False


# (request id, new token ids, finished?, finish reason, sequence)
TokenEvent = Tuple[int, List[int], bool, Optional[str], Sequence]


@dataclass
class EngineStats:
    steps: int = 0
    prompt_tokens: int = 0
    generation_tokens: int = 0
    requests_finished: int = 0
    busy_s: float = 0.0
    spec_proposed: int = 0
    spec_accepted: int = 0
    step_ms_ewma: float = 0.0
    # submit -> first sampled token, inside the engine (sum of seconds, count)
    first_token_s: float = 0.0
    first_token_n: int = 0
    first_token_steps: int = 0  # steps launched between submit and the first token (sum)


def _resolve_device(name: str) -> torch.device:
    if name == "auto":
        return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    return torch.device(name)


def _sample_row(logits: torch.Tensor, p: SamplingParams, gen: Optional[torch.Generator]) -> int:
    """Temperature / top-k / top-p sampling of one logits row (fp32)."""
    x = logits.float() / max(p.temperature, 1e-5)
    if p.top_k and p.top_k > 0:
        kth = torch.topk(x, min(p.top_k, x.numel())).values[-1]
        x = torch.where(x < kth, torch.full_like(x, float("-inf")), x)
    probs = torch.softmax(x, dim=-1)
    if p.top_p < 1.0:
        sp, si = torch.sort(probs, descending=True)
        keep = (torch.cumsum(sp, 0) - sp) < p.top_p
        sp = sp * keep
        probs = torch.zeros_like(probs).scatter_(0, si, sp)
        probs = probs / probs.sum()
    return int(torch.multinomial(probs, 1, generator=gen).item())


def _host_to_device(values: List[Any], dtype: torch.dtype, dev: torch.device) -> torch.Tensor:
    """Small per-step parameter vector to the device without stalling the host: pinned staging
    and an asynchronous copy (a pageable copy would wait for everything queued on the stream)."""
    t = torch.tensor(values, dtype=dtype)
    if dev.type != "cuda":
        return t
    return t.pin_memory().to(dev, non_blocking=True)


def _probs_rows(logits: torch.Tensor, params: List[SamplingParams]) -> torch.Tensor:
    """The distributions tokens are drawn from: softmax of ``logits`` (``[R, V]``, one
    SamplingParams per row) after temperature, top-k and top-p — fp32 ``[R, V]``, rows sum to
    one; every op on the device, nothing read back."""
    dev = logits.device
    temps = _host_to_device([max(p.temperature, 1e-5) for p in params], torch.float32, dev)
    x = logits.float() / temps[:, None]
    ks = [min(p.top_k, x.shape[1]) if p.top_k and p.top_k > 0 else 0 for p in params]
    if any(ks):
        vals = torch.topk(x, max(ks), dim=-1).values
        kidx = _host_to_device([max(k, 1) - 1 for k in ks], torch.int64, dev)
        kth = vals.gather(1, kidx[:, None])
        use = _host_to_device([k > 0 for k in ks], torch.bool, dev)[:, None]
        x = torch.where(use & (x < kth), torch.full_like(x, float("-inf")), x)
    probs = torch.softmax(x, dim=-1)
    if any(p.top_p < 1.0 for p in params):
        tp = _host_to_device([p.top_p for p in params], torch.float32, dev)
        sp, si = torch.sort(probs, dim=-1, descending=True)
        keep = (torch.cumsum(sp, -1) - sp) < tp[:, None]
        sp = sp * keep
        probs = torch.zeros_like(probs).scatter_(1, si, sp)
        probs = probs / probs.sum(-1, keepdim=True)
    return probs


def seeded_uniform(seed: int, k: int) -> float:
    """The uniform variate in [0, 1) of draw ``k`` of a request with seed ``seed``: a counter-based
    hash (splitmix64 finaliser), so the draw for a request's k-th token is the same whichever
    step, batch, rank or replay computes it, and the host can hand it to a device sampler without
    owning a generator whose state would have to follow the tokens."""
    m = (1 << 64) - 1
    x = ((int(seed) & m) * 0x9E3779B97F4A7C15 + (int(k) + 1) * 0xD1B54A32D192ED03) & m
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return (x >> 40) / float(1 << 24)


def _uniforms(draws: Optional[List[Optional[float]]], n: int, dev: torch.device) -> torch.Tensor:
    """One uniform variate per sampled row: the request's own (seeded rows) or the device
    generator's (``None`` entries)."""
    u = torch.rand(n, dtype=torch.float32, device=dev)
    if draws is not None and any(d is not None for d in draws):
        own = _host_to_device([float("nan") if d is None else d for d in draws], torch.float32, dev)
        u = torch.where(torch.isnan(own), u, own)
    return u


def _sample_rows_device(logits: torch.Tensor, params: List[SamplingParams], draws: Optional[List[Optional[float]]] = None) -> torch.Tensor:
    """Temperature / top-k / top-p sampling of many rows at once: one softmax / sort / inverse
    CDF over the batch, one uniform variate per row (``draws``: those of seeded requests); the
    ids stay on the device (int64 ``[R]``)."""
    probs = _probs_rows(logits, params)
    cdf = torch.cumsum(probs, dim=-1)
    u = _uniforms(draws, probs.shape[0], probs.device) * cdf[:, -1]
    idx = torch.searchsorted(cdf, u[:, None].contiguous(), right=True).view(-1)
    # (a variate that rounds up to the total lands behind the last token: the last one with mass instead)
    last = (probs > 0).int().cumsum(-1).argmax(-1)
    return torch.minimum(idx, last)


def _sample_rows(logits: torch.Tensor, params: List[SamplingParams]) -> List[int]:
    """:func:`_sample_rows_device` with ONE device synchronisation at the end, instead of one
    per row — at 20 sampled rows per step the per-row form costs more than the whole forward pass."""
    return _sample_rows_device(logits, params).tolist()


class LLMEngine:
    def __init__(self, ecfg: EngineConfig, weights: Optional[LlamaWeights] = None, tp_group: Any = None):
        self.ecfg = ecfg
        self.device = _resolve_device(ecfg.device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        # `model`: a preset name, or a Hugging Face checkpoint directory (architecture from its config.json, weights from its *.safetensors)
        weights_path = ecfg.weights_path
        if weights is None:
            from kvmini.models.llama import resolve_model

            self.mcfg, weights_path = resolve_model(ecfg.model, ecfg.weights_path)
        else:
            self.mcfg = weights.cfg
        dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[ecfg.dtype]
        t0 = time.perf_counter()
        if weights is None:
            if weights_path:
                from kvmini.models.llama import load_hf_safetensors

                weights = load_hf_safetensors(weights_path, self.mcfg, self.device, dtype, ecfg.tp_rank, ecfg.tp_size)
            else:
                damp = None
                if ecfg.tail_damping > 0 and ecfg.draft_model and ecfg.draft_model.startswith("self:"):
                    damp = (int(ecfg.draft_model.split(":", 1)[1]), float(ecfg.tail_damping))
                weights = random_weights(self.mcfg, self.device, dtype, ecfg.seed, ecfg.tp_rank, ecfg.tp_size, tail_damping=damp)
        self.weights = weights
        rcfg = RunnerConfig(
            max_num_seqs=ecfg.max_num_seqs,
            max_batched_tokens=ecfg.max_batched_tokens,
            max_model_len=ecfg.max_model_len,
            block_size=ecfg.block_size,
            num_blocks=ecfg.num_blocks,
            kv_memory_fraction=ecfg.kv_memory_fraction,
            use_graphs=ecfg.use_graphs,
            kernels=ecfg.kernels,
            quantization=ecfg.quantization,
            kv_cache_dtype=ecfg.kv_cache_dtype,
            tp_graphs=ecfg.tp_graphs,
            tp_rccl_graphs=ecfg.tp_rccl_graphs,
            tp_custom_allreduce=ecfg.tp_custom_allreduce,
            tp_force_collectives=ecfg.tp_force_collectives,
        )
        self.runner = ModelRunner(weights, rcfg, tp_group)
        self.weights, self.mcfg = self.runner.w, self.runner.cfg  # the runner may have re-shaped the model (padded heads)
        self.bm = BlockManager(self.runner.num_blocks, ecfg.block_size, ecfg.enable_prefix_caching)
        self.scheduler = Scheduler(self.bm, ecfg.max_num_seqs, ecfg.max_batched_tokens, ecfg.max_model_len, self.mcfg.eos_token_id, ecfg.ride_along_limit)
        self.scheduler.vocab_size = self.mcfg.vocab_size
        self.spec = None
        if ecfg.draft_model and ecfg.num_speculative_tokens > 0:
            from .speculative import SpeculativeDecoder

            self.spec = SpeculativeDecoder(self, ecfg.draft_model, ecfg.num_speculative_tokens)
            self.scheduler.guided_supported = False
        self.load_s = time.perf_counter() - t0
        self.stats = EngineStats()
        self._ids = itertools.count(1)
        self._inbox: "queue.SimpleQueue[Any]" = queue.SimpleQueue()
        self._wake = threading.Event()
        self._stop = False
        self._thread: Optional[threading.Thread] = None
        self.on_tokens: Optional[Callable[[List[TokenEvent]], None]] = None
        self._gens: Dict[int, torch.Generator] = {}
        self._rejected: List[Sequence] = []
        self.warm_s = 0.0
        self._pending: Optional[Tuple[Batch, Any]] = None  # (batch, runner handle) of the step in flight
        # logprobs of a step that was launched ahead leave the device right behind its ids: two host
        # buffers in turn (the step in flight and the one being launched), pinned on a GPU
        self._lp_host: List[Optional[Tuple[torch.Tensor, torch.Tensor]]] = [None, None]
        self._lp_turn = 0
        self.pipelined_steps = 0
        self._step_dur: Dict[Tuple[int, int], float] = {}  # measured device time of a step, by runner key
        self._probe_ctr: Dict[Tuple[int, int], int] = {}
        self._launch_cost = 0.0  # host seconds from the end of the hold to the enqueued step (moving average)
        self._late_penalty = 0.0
        self._t_collect = 0.0  # when the previous step's ids arrived = when the step in flight started
        # guided decoding: token classes (set by the front-end, which owns the tokenizer), one
        # automaton state per sequence slot on the device, and who initialised each slot
        self._guided_cls_np: Any = None
        self._guided_cls: Optional[torch.Tensor] = None
        self._guided_vc = 1
        self._guided_state = torch.zeros((ecfg.max_num_seqs, 4), dtype=torch.int32, device=self.device)
        self._guided_owner: List[Any] = [None] * ecfg.max_num_seqs
        # sampling penalties: per slot, how often every token was generated (int32 [slots, V]) and which tokens the
        # prompt holds (bool [slots, V]); allocated when the first penalised request arrives
        self._pen_counts: Optional[torch.Tensor] = None
        self._pen_prompt: Optional[torch.Tensor] = None
        self._pen_owner: List[Any] = [None] * ecfg.max_num_seqs
        # schema-guided sequences: the skeleton script of each slot (entries, rest) and its cursor
        from .guided import SCRIPT_MAX

        self._guided_script = torch.zeros((ecfg.max_num_seqs, 2, SCRIPT_MAX), dtype=torch.int32, device=self.device)
        self._guided_cursor = torch.full((ecfg.max_num_seqs,), -1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------
    def warmup(self) -> float:
        """Capture the HIP graphs (and JIT/allocator warm-up).  Returns seconds."""
        t0 = time.perf_counter()
        with torch.no_grad():
            self.runner.capture_graphs()
            if self.spec is not None:
                self.spec.warmup()
        self.warm_s = time.perf_counter() - t0
        self.check_collectives()
        return self.warm_s

    def check_collectives(self) -> None:
        """Raise if a peer-memory collective timed out waiting for a peer (its results are garbage)."""
        for x in (self.runner.xgmi, getattr(self.runner, "xgmi_ids", None)):
            if x is not None:
                n = x.errors()
                if n:
                    # the communicator is poisoned from here on (a late peer leaves stale flags behind): the
                    # group has to be restarted, so this is raised to whoever drives the engine
                    raise RuntimeError(f"{n} xGMI collective wait(s) timed out: a tensor-parallel rank did not take part in a step")

    def set_token_classes(self, cls: Any) -> None:
        """Token classes for guided decoding (``kvmini.engine.guided.token_classes`` of the
        serving tokenizer): uint8 per token id."""
        import numpy as np

        from . import guided

        cls = np.ascontiguousarray(cls, dtype=np.uint8)[: self.mcfg.vocab_size]
        self._guided_cls_np = cls
        self._guided_cls = torch.from_numpy(cls.copy()).to(self.device)
        self._guided_vc = guided.value_cost(cls)

    @property
    def guided_ready(self) -> bool:
        return self._guided_cls is not None

    # --- synchronous API (tests, offline generation) ---------------------
    def make_sequence(self, prompt_ids: List[int], params: Optional[SamplingParams] = None, user: Any = None) -> Sequence:
        return Sequence(next(self._ids), list(prompt_ids), params or SamplingParams(), user=user)

    def step(self) -> List[TokenEvent]:
        """One engine iteration.  Empty list when nothing is schedulable (or, pipelined, when
        the first step of a burst has been launched and none was in flight before it)."""
        if self.spec is not None:
            return self.spec.step()
        t0 = time.perf_counter()
        launched = False
        pipelined = self.ecfg.async_scheduling and self._can_pipeline()
        unheld = True
        if pipelined:
            unheld = self._hold_launch()
            # the hold takes in new requests: one of them may need the synchronous path
            pipelined = unheld or self._can_pipeline()
        if pipelined:
            # launch the next step on top of the one in flight, THEN read that one's tokens:
            # the device never waits for the host between steps
            t_launch = time.perf_counter()
            batch = self.scheduler.schedule()
            handle = None
            if batch.chunks:
                with torch.no_grad():
                    handle = self.runner.launch(batch, self._device_sampler(batch))
                self.scheduler.launch_commit(batch)
                launched = True
                self.pipelined_steps += 1
            had_pending = self._pending is not None
            if handle is not None:
                handle.timed = unheld and had_pending
                self._launch_cost = 0.8 * self._launch_cost + 0.2 * (time.perf_counter() - t_launch)
            events = self._finalize_pending(held=not unheld)
            self._pending = (batch, handle) if handle is not None else None
            if not had_pending:  # nothing was in flight: the new step starts now
                self._t_collect = time.perf_counter()
        else:
            events = self._finalize_pending()
            batch = self.scheduler.schedule()
            if batch.chunks:
                with torch.no_grad():
                    ids, logits = self.runner.execute(batch, self._device_sampler(batch, guided_only=True))
                events += self._commit(batch, ids, logits)
                launched = True
        if launched:
            # a peer-memory collective that timed out leaves garbage behind and its flags unreliable:
            # look at the error word now and then (reading it waits for the device, hence not per step)
            self._collective_check = getattr(self, "_collective_check", 0) + 1
            if self.runner.xgmi is not None and self.runner.tp_size > 1 and self._collective_check % 512 == 0:
                self.check_collectives()
            dt = time.perf_counter() - t0
            st = self.stats
            st.steps += 1
            st.busy_s += dt
            st.step_ms_ewma = dt * 1e3 if st.step_ms_ewma == 0 else 0.9 * st.step_ms_ewma + 0.1 * dt * 1e3
        self._progress = launched or bool(events)
        return events

    def _hold_launch(self) -> bool:
        """Just-in-time launch: sleep until shortly before the step in flight should end (its
        duration is known from earlier steps of the same shape), then take in what arrived.  A
        request that arrives during a step thus rides the very next one instead of the one
        after.  Unknown shapes launch immediately, and so does every 16th step of a shape: a
        step enqueued at once behind one in flight runs back to back with it, so the time
        between the two read-backs IS its device time — the only clock this needs, and one that a
        late launch cannot inflate.  Returns True when the launch was not held."""
        margin = self.ecfg.launch_margin_ms * 1e-3
        if self._pending is None or margin <= 0 or self.device.type != "cuda":
            return True
        # the margin has to cover the host's own launch work (scheduling, staging, the graph
        # launch call: ~0.2 ms, far more under a profiler or on a crowded host) with room to
        # spare, plus a penalty that grows while read-backs find the step already finished
        margin = max(margin, 2.5 * self._launch_cost + 0.2e-3) + self._late_penalty
        key = self._pending[1].key
        dur = self._step_dur.get(key)
        n = self._probe_ctr.get(key, 0)
        self._probe_ctr[key] = n + 1
        if dur is None or n % 16 == 0:
            return True
        wait = self._t_collect + dur - margin - time.perf_counter()
        if wait > 50e-6:
            time.sleep(wait)
            self._drain_inbox()
            return False
        return True

    def _device_sampler(self, batch: Batch, guided_only: bool = False) -> Optional[Callable[[torch.Tensor, torch.Tensor], None]]:
        """Device ops queued right behind the forward pass that replace argmax ids in place, so
        that the next step can be launched on them before the host has seen them:

        * rows with temperature > 0 are sampled (``guided_only``: left to the host, which owns
          seeded generators and runs after the read-back in the synchronous path);
        * rows of guided requests first get the automaton's mask over their logits, are then
          picked (argmax or sampling) from what is left, and the pick is fed to the automaton.

        None when no row needs any of this."""
        rows: List[int] = []
        plist: List[SamplingParams] = []
        g_rows: List[int] = []
        g_slots: List[int] = []
        g_init: List[int] = []
        g_cur: List[int] = []
        g_scripts: List[Tuple[int, List[List[int]]]] = []  # (slot, [entries, rest]) to upload before the mask runs
        g_greedy: List[int] = []
        draws: List[Optional[float]] = []
        lp_rows: List[int] = []
        lp_k: List[int] = []
        pen_rows: List[int] = []
        pen_slots: List[int] = []
        pen_vals: List[Tuple[float, float, float]] = []
        pen_greedy: List[int] = []  # positions in pen_rows of greedy, unguided rows: their pick is the argmax of the penalised logits
        pen_reset: List[Tuple[int, List[int], List[int]]] = []  # (slot, generated so far, prompt) of sequences new to their slot
        pen_bias: List[Tuple[int, int, float]] = []  # (position in pen_rows, token, value added): logit_bias, and -inf on the stop tokens before min_tokens
        r = 0
        for ch in batch.chunks:
            if ch.sample:
                p = ch.seq.params
                if p.penalised:
                    self._stage_penalties(ch.seq, r, pen_rows, pen_slots, pen_vals, pen_greedy, pen_reset, pen_bias)
                if p.logprobs > 0 and not guided_only:  # (the synchronous path reads them off the logits it gets back)
                    lp_rows.append(r)
                    lp_k.append(int(p.logprobs))
                if p.guided:
                    self._stage_guided(ch.seq, r, g_rows, g_slots, g_init, g_cur, g_scripts)
                    if p.temperature <= 0:
                        g_greedy.append(r)
                if p.temperature > 0 and (p.guided or p.penalised or not guided_only):
                    rows.append(r)
                    plist.append(p)
                    draws.append(self._draw_for(ch.seq))
                r += 1
        if not rows and not g_rows and not lp_rows and not pen_rows:
            return None
        if lp_rows:
            batch.logprob_rows = self._logprob_holder(lp_rows, lp_k)

        def run(logits: torch.Tensor, out_ids: torch.Tensor) -> None:
            ops, dev = self.runner.ops, logits.device
            if pen_rows:
                # penalties first: masks and picks below work on the penalised logits
                cnts, prm = self._pen_counts, self._pen_prompt
                for slot, gen, prompt in pen_reset:
                    cnts[slot].zero_()
                    prm[slot].zero_()
                    if gen:
                        cnts[slot].index_add_(0, _host_to_device(gen, torch.int64, dev), torch.ones(len(gen), dtype=cnts.dtype, device=dev))
                    if prompt:
                        prm[slot].index_fill_(0, _host_to_device(prompt, torch.int64, dev), True)
                pr, ps = _host_to_device(pen_rows, torch.int64, dev), _host_to_device(pen_slots, torch.int64, dev)
                pv = _host_to_device([v for t in pen_vals for v in t], torch.float32, dev).view(-1, 3)
                cnt = cnts.index_select(0, ps)
                lg = logits.index_select(0, pr).float()
                if any(t[2] != 1.0 for t in pen_vals):
                    seen = (cnt > 0) | prm.index_select(0, ps)
                    rp = pv[:, 2:3]
                    lg = torch.where(seen, torch.where(lg > 0, lg / rp, lg * rp), lg)
                lg = lg - pv[:, 1:2] * cnt.float() - pv[:, 0:1] * (cnt > 0).float()
                if pen_bias:
                    lg.index_put_((_host_to_device([e[0] for e in pen_bias], torch.int64, dev), _host_to_device([e[1] for e in pen_bias], torch.int64, dev)),
                                  _host_to_device([e[2] for e in pen_bias], torch.float32, dev), accumulate=True)
                logits.index_copy_(0, pr, lg.to(logits.dtype))
                if pen_greedy:
                    pg = _host_to_device(pen_greedy, torch.int64, dev)
                    out_ids.index_copy_(0, pr.index_select(0, pg), lg.index_select(0, pg).argmax(-1).to(out_ids.dtype))
            if g_rows:
                gr, gs = _host_to_device(g_rows, torch.int64, dev), _host_to_device(g_slots, torch.int32, dev)
                for slot, rows2 in g_scripts:  # (stream-ordered: lands before the mask kernel reads it)
                    self._guided_script[slot].copy_(_host_to_device(rows2, torch.int32, dev), non_blocking=True)
                ops.guided_mask(logits, gr, gs, _host_to_device(g_init, torch.int32, dev), self._guided_state, self._guided_cls, self._guided_vc,
                                self._guided_script, self._guided_cursor, _host_to_device(g_cur, torch.int32, dev))
                if g_greedy:
                    idx = gr if len(g_greedy) == len(g_rows) else _host_to_device(g_greedy, torch.int64, dev)
                    out_ids.index_copy_(0, idx, ops.greedy_sample(logits.index_select(0, idx)))
            if rows:
                self._sample_into(logits, rows, plist, out_ids, draws)
            if g_rows:
                ops.guided_advance(out_ids, gr, gs, self._guided_state, self._guided_cls, self._guided_script, self._guided_cursor)
            if pen_rows:  # the picks count from the next step on
                self._pen_counts.index_put_((ps, out_ids.index_select(0, pr)), torch.ones(len(pen_rows), dtype=self._pen_counts.dtype, device=dev), accumulate=True)
            if lp_rows:
                # log-probability of the picked token and the top-k of its row, from the logits the pick was
                # made on (masked ones for guided rows): device ops, then a copy to the host buffer that the
                # step's completion event covers — nothing is read back here
                _rows, ks, hv, hi = batch.logprob_rows
                li = _host_to_device(lp_rows, torch.int64, dev)
                lp = torch.log_softmax(logits.index_select(0, li).float(), -1)
                kmax = min(max(ks), lp.shape[1])
                tv, ti = torch.topk(lp, kmax, dim=-1)
                hv[: len(lp_rows), 0:1].copy_(lp.gather(1, out_ids.index_select(0, li).view(-1, 1)), non_blocking=True)
                hv[: len(lp_rows), 1 : 1 + kmax].copy_(tv, non_blocking=True)
                hi[: len(lp_rows), :kmax].copy_(ti, non_blocking=True)

        return run

    LOGPROBS_MAX = 20  # top-logprobs per token the host buffers hold (the OpenAI API's limit)

    def _logprob_holder(self, rows: List[int], ks: List[int]) -> Tuple[Dict[int, int], List[int], torch.Tensor, torch.Tensor]:
        """(row -> line, k per line, values ``[lines, 1 + K]``, ids ``[lines, K]``) for the step being launched."""
        turn = self._lp_turn
        self._lp_turn ^= 1
        if self._lp_host[turn] is None:
            n, pin = self.ecfg.max_num_seqs * 4, self.device.type == "cuda"
            self._lp_host[turn] = (torch.zeros((n, 1 + self.LOGPROBS_MAX), dtype=torch.float32, pin_memory=pin), torch.zeros((n, self.LOGPROBS_MAX), dtype=torch.int64, pin_memory=pin))
        hv, hi = self._lp_host[turn]
        if len(rows) > hv.shape[0]:
            raise RuntimeError(f"{len(rows)} rows with logprobs in one step; the host buffers hold {hv.shape[0]}")
        return {r: j for j, r in enumerate(rows)}, [min(k, self.LOGPROBS_MAX) for k in ks], hv, hi

    def _stage_penalties(self, seq: Sequence, row: int, pen_rows: List[int], pen_slots: List[int], pen_vals: List[Tuple[float, float, float]], pen_greedy: List[int],
                         pen_reset: List[Tuple[int, List[int], List[int]]], pen_bias: List[Tuple[int, int, float]]) -> None:
        """One penalised row of the step being launched.  The slot's count / prompt rows are rebuilt by the device
        ops when this sequence has not used the slot before (first sample, or taken up again after a preemption:
        recounted over the tokens it already has — a token still in flight was counted by the step that picked it)."""
        from .scheduler import PLACEHOLDER

        if self._pen_counts is None:
            v, n = self.mcfg.vocab_size, self.ecfg.max_num_seqs
            self._pen_counts = torch.zeros((n, v), dtype=torch.int32, device=self.device)
            self._pen_prompt = torch.zeros((n, v), dtype=torch.bool, device=self.device)
        p = seq.params
        key = (seq.req_id, seq.preemptions)
        if self._pen_owner[seq.slot] != key:
            self._pen_owner[seq.slot] = key
            pen_reset.append((seq.slot, [t for t in seq.output_ids if t != PLACEHOLDER], list(seq.prompt_ids) if p.repetition_penalty != 1.0 else []))
        if p.temperature <= 0 and not p.guided:
            pen_greedy.append(len(pen_rows))
        if p.logit_bias:
            pen_bias.extend((len(pen_rows), int(t), float(v)) for t, v in p.logit_bias.items())
        if len(seq.output_ids) < p.min_tokens:  # (tokens in flight included: this pick is number len + 1)
            stops = set(p.stop_token_ids)
            if not p.ignore_eos:
                stops.add(self.mcfg.eos_token_id)
            pen_bias.extend((len(pen_rows), int(t), float("-inf")) for t in stops if 0 <= t < self.mcfg.vocab_size)
        pen_rows.append(row)
        pen_slots.append(seq.slot)
        pen_vals.append((float(p.presence_penalty), float(p.frequency_penalty), float(p.repetition_penalty)))

    def _stage_guided(self, seq: Sequence, row: int, g_rows: List[int], g_slots: List[int], g_init: List[int], g_cur: List[int],
                      g_scripts: List[Tuple[int, List[List[int]]]]) -> None:
        """One guided row of the step being launched.  The automaton of the sequence's slot is
        (re)initialised by the kernel when this sequence has not used the slot before (first
        sample, or taken up again after a preemption: replayed over the tokens it already has);
        a schema-guided sequence's skeleton script goes to the slot's device buffer then."""
        from . import guided
        from .scheduler import PLACEHOLDER

        if self._guided_cls is None:
            raise RuntimeError("guided decoding needs token classes: call set_token_classes() first")
        key = (seq.req_id, seq.preemptions)
        g_rows.append(row)
        g_slots.append(seq.slot)
        if self._guided_owner[seq.slot] != key:
            self._guided_owner[seq.slot] = key
            known = [t for t in seq.output_ids if t != PLACEHOLDER]
            script = list(seq.params.guided_script) if seq.params.guided_script else None
            st, cur = guided.replay(self._guided_cls_np, seq.params.guided_start, seq.params.max_tokens, known, script)
            g_init += list(st)
            g_cur.append(cur)
            if script:
                pad = guided.SCRIPT_MAX - len(script)
                g_scripts.append((seq.slot, [script + [guided.SCRIPT_END] * pad, guided.script_rest(script, self._guided_vc) + [1] * pad]))
        else:
            g_init += [-1, 0, 0, 0]
            g_cur.append(-1)

    def _sample_into(self, logits: torch.Tensor, rows: List[int], plist: List[SamplingParams], out_ids: torch.Tensor,
                     draws: Optional[List[Optional[float]]] = None) -> torch.Tensor:
        """Sample rows ``rows`` of ``logits`` (one SamplingParams each; ``draws``: the uniform
        variate of a seeded request's row, None for the others) into ``out_ids[rows]`` with device
        ops only; returns the device index tensor of the rows."""
        dev = logits.device
        idx = _host_to_device(rows, torch.int64, dev)
        kernel = getattr(self.runner.ops, "sample_rows", None)
        if kernel is not None and logits.dtype == torch.bfloat16 and logits.stride(1) == 1:
            # one launch, no sort: threshold search on the bf16 logits (csrc/sampling.hip)
            kernel(logits, idx, _host_to_device([max(p.temperature, 1e-5) for p in plist], torch.float32, dev),
                   _host_to_device([int(p.top_k or 0) for p in plist], torch.int32, dev), _host_to_device([float(p.top_p) for p in plist], torch.float32, dev), out_ids,
                   _uniforms(draws, len(rows), dev))
        else:
            out_ids.index_copy_(0, idx, _sample_rows_device(logits.index_select(0, idx), plist, draws))
        return idx

    @staticmethod
    def _draw_for(seq: Sequence) -> Optional[float]:
        """The uniform variate of the token a seeded sequence samples next (None: no seed).  The
        counter is the number of tokens it has produced or has in flight, so a step that is
        launched ahead draws what the synchronous path would have drawn."""
        if seq.params.seed is None:
            return None
        return seeded_uniform(int(seq.params.seed), len(seq.output_ids))

    @staticmethod
    def _needs_host_sampling(p: SamplingParams) -> bool:
        # Nothing a request can ask for needs the logits on the host any more: seeded requests are
        # sampled by the device ops like unseeded ones (their variates are a function of (seed, token
        # index) — seeded_uniform — that the host computes at launch time), guided rows are picked on
        # the device, and logprobs are extracted by device ops behind the pick (_logprob_holder).
        # Only a top-logprobs count beyond what the host buffers hold takes the synchronous path.
        return int(p.logprobs or 0) > LLMEngine.LOGPROBS_MAX

    def _can_pipeline(self) -> bool:
        """Pipelined stepping needs every token of the next step to be either known or already
        on the device when that step runs (argmax, or unseeded sampling done by device ops), and
        no preemption (a preempted sequence is rescheduled from its token list, which may not
        hold placeholders)."""
        sch = self.scheduler
        # (tensor parallelism: the followers run the forward pass and the argmax; when the scheduling
        # rank samples or masks rows on the device, its ids reach them on the device as well —
        # ModelRunner.sync_sampled_ids — so sampled and guided requests are launched ahead there too)
        for s in sch.running:
            if self._needs_host_sampling(s.params):
                return False
        room = sch.max_num_seqs - len(sch.running)
        if room > 0 and sch.waiting:
            for s in itertools.islice(sch.waiting, room):
                if self._needs_host_sampling(s.params):
                    return False
        return self.bm.num_free >= len(sch.running) + sch.max_batched_tokens // self.bm.block_size + 2

    def generator_for(self, seq: Sequence) -> Optional[torch.Generator]:
        """The request's own random generator (requests with a seed), created on first use."""
        if seq.params.seed is None:
            return None
        gen = self._gens.get(seq.req_id)
        if gen is None:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(seq.params.seed))
            self._gens[seq.req_id] = gen
        return gen

    def _finalize_pending(self, held: bool = False) -> List[TokenEvent]:
        """Read back the step in flight (if any) and commit its tokens.  ``held``: the launch
        that preceded this read-back was delayed on purpose (see :meth:`_hold_launch`)."""
        if self._pending is None:
            return []
        batch, handle = self._pending
        self._pending = None
        t_wait = time.perf_counter()
        ids, _ = self.runner.collect(handle)
        now = time.perf_counter()
        if held:
            if now - t_wait < 50e-6:  # nothing to wait for: the device had run dry, the launch came too late
                self._late_penalty = min(5e-3, 2.0 * self._late_penalty + 0.2e-3)
            else:
                self._late_penalty *= 0.95
        if handle.timed:  # ran back to back with its predecessor: started when that one was read back
            d = now - self._t_collect
            old = self._step_dur.get(handle.key)
            if old is None or d < 1.5 * old:
                self._step_dur[handle.key] = d if old is None else 0.5 * (old + d)
        self._t_collect = now
        events: List[TokenEvent] = []
        lp_map = lp_ks = lp_vals = lp_ids = None
        holder = getattr(batch, "logprob_rows", None)
        if holder is not None:  # (the copies into the host buffers completed with the step's event)
            lp_map, lp_ks, hv, hi = holder
            lp_vals, lp_ids = hv[: len(lp_map)].tolist(), hi[: len(lp_map)].tolist()
        row = 0
        for ch in batch.chunks:
            seq = ch.seq
            if ch.start < len(seq.prompt_ids):
                self.stats.prompt_tokens += min(ch.n_tokens, len(seq.prompt_ids) - ch.start)
            tok = None
            if ch.sample:
                tok = ids[row]
                row += 1
            if seq.status != SeqStatus.RUNNING:
                continue
            first = seq.first_token_time is None
            new = self.scheduler.finalize(ch, tok)
            if not ch.sample:
                continue
            if lp_map is not None and (row - 1) in lp_map and new:
                j = lp_map[row - 1]
                k = lp_ks[j]
                seq.logprob_log.append((float(lp_vals[j][0]), list(zip(lp_ids[j][:k], lp_vals[j][1 : 1 + k]))))
            if first and seq.first_token_time is not None:
                self.stats.first_token_s += seq.first_token_time - seq.arrival
                self.stats.first_token_n += 1
                self.stats.first_token_steps += self.stats.steps - getattr(seq, "submit_step", self.stats.steps)
            self.stats.generation_tokens += len(new)
            done = seq.status == SeqStatus.FINISHED
            if done:
                self.stats.requests_finished += 1
            events.append((seq.req_id, new, done, seq.finish_reason, seq))
        return events

    def _commit(self, batch: Batch, ids: List[int], logits: Optional[torch.Tensor]) -> List[TokenEvent]:
        events: List[TokenEvent] = []
        # temperature sampling: all such rows of the step in one batched call
        batched: Dict[int, int] = {}
        if logits is not None:
            sampled = [ch.seq for ch in batch.chunks if ch.sample]
            rows_b = [r for r, sq in enumerate(sampled) if sq.params.temperature > 0 and not sq.params.guided and not sq.params.penalised]  # (those were picked by the device ops)
            if rows_b:
                tmp = torch.zeros(logits.shape[0], dtype=torch.int64, device=logits.device)
                idx = self._sample_into(logits, rows_b, [sampled[r].params for r in rows_b], tmp, [self._draw_for(sampled[r]) for r in rows_b])
                batched = dict(zip(rows_b, tmp[idx].tolist()))
        row = 0
        for ch in batch.chunks:
            seq = ch.seq
            if ch.start < len(seq.prompt_ids):
                self.stats.prompt_tokens += min(ch.n_tokens, len(seq.prompt_ids) - ch.start)
            if not ch.sample:
                self.scheduler.commit(ch, [])
                continue
            tok = ids[row]
            p = seq.params
            if row in batched:
                tok = batched[row]
            lp_entry = None
            if p.logprobs > 0 and logits is not None:
                lp = torch.log_softmax(logits[row].float(), -1)
                tv, ti = torch.topk(lp, min(p.logprobs, lp.numel()))
                lp_entry = (float(lp[tok]), list(zip(ti.tolist(), tv.tolist())))
            row += 1
            first = seq.first_token_time is None
            new = self.scheduler.commit(ch, [tok])
            if first and seq.first_token_time is not None:
                self.stats.first_token_s += seq.first_token_time - seq.arrival
                self.stats.first_token_n += 1
            if lp_entry is not None and new:
                seq.logprob_log.append(lp_entry)
            self.stats.generation_tokens += len(new)
            done = seq.status == SeqStatus.FINISHED
            if done:
                self.stats.requests_finished += 1
                self._gens.pop(seq.req_id, None)
            events.append((seq.req_id, new, done, seq.finish_reason, seq))
        return events

    def generate(self, prompts: List[List[int]], params: "Optional[SamplingParams | List[SamplingParams]]" = None) -> List[List[int]]:
        """Offline batch generation (blocking).  ``params``: one set for all prompts, or one per prompt."""
        plist = params if isinstance(params, (list, tuple)) else [params] * len(prompts)
        if len(plist) != len(prompts):
            raise ValueError(f"{len(plist)} sampling parameter sets for {len(prompts)} prompts")
        seqs = [self.make_sequence(p, q and SamplingParams(**vars(q))) for p, q in zip(prompts, plist)]
        for s in seqs:
            why = self.scheduler.add(s)
            if why is not None:
                raise ValueError(why)
        while self.scheduler.has_work():
            if not self.step() and self._pending is None and not self.scheduler.running and self.scheduler.waiting:
                raise RuntimeError("scheduler cannot make progress (KV cache too small for the prompt?)")
            if self.scheduler.failed:
                raise RuntimeError(self.scheduler.failed.pop().finish_reason)
        return [s.output_ids for s in seqs]

    # --- asynchronous API (server) ----------------------------------------
    def start(self) -> None:
        if self._thread is not None:
            return
        self._stop = False
        self._thread = threading.Thread(target=self._loop, name="kvmini-engine", daemon=True)
        self._thread.start()

    def stop(self) -> None:
        self._stop = True
        self._wake.set()
        if self._thread is not None:
            self._thread.join(10)
            self._thread = None
        self.release_followers()

    # --- tensor parallelism: rank 0 of the TP group schedules, the others follow -------------
    def follow(self) -> int:
        """Serve as a tensor-parallel follower until the scheduling rank stops (blocking)."""
        with torch.no_grad():
            return self.runner.follow()

    def release_followers(self) -> None:
        if self.ecfg.tp_size > 1 and self.ecfg.tp_rank == 0 and not getattr(self, "_followers_released", False):
            self._followers_released = True
            self.runner.stop_followers()

    def submit(self, prompt_ids: List[int], params: SamplingParams, user: Any = None) -> int:
        seq = self.make_sequence(prompt_ids, params, user)
        seq.submit_step = self.stats.steps  # type: ignore[attr-defined]
        self._inbox.put(("add", seq))
        self._wake.set()
        return seq.req_id

    def abort(self, req_id: int) -> None:
        self._inbox.put(("abort", req_id))
        self._wake.set()

    def _drain_inbox(self) -> None:
        while True:
            try:
                kind, item = self._inbox.get_nowait()
            except queue.Empty:
                return
            if kind == "add":
                if self.scheduler.add(item) is not None:
                    self._rejected.append(item)
            elif kind == "abort":
                self.scheduler.abort(item)

    def _failure_events(self) -> List[TokenEvent]:
        """Requests that ended without a step of their own: rejected at intake, or given up by
        the scheduler.  Each fails alone; running requests are not touched."""
        out: List[TokenEvent] = [(s.req_id, [], True, s.finish_reason, s) for s in self._rejected]
        out += [(s.req_id, [], True, s.finish_reason, s) for s in self.scheduler.failed]
        self._rejected.clear()
        self.scheduler.failed.clear()
        return out

    def _loop(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        while not self._stop:
            self._drain_inbox()
            if not self.scheduler.has_work():
                fails = self._failure_events()
                if fails and self.on_tokens is not None:
                    self.on_tokens(fails)
                    continue
                self._wake.wait(0.05)
                self._wake.clear()
                continue
            try:
                events = self.step()
                events += self._failure_events()  # also what the hold inside step() took in
            except Exception as e:  # fail the in-flight requests, keep serving
                import traceback

                traceback.print_exc()
                events = []
                self._pending = None
                for s in list(self.scheduler.running):
                    self.scheduler._finish(s, "error")
                    events.append((s.req_id, [], True, f"error: {e}", s))
            if events and self.on_tokens is not None:
                self.on_tokens(events)
            elif not events and not getattr(self, "_progress", False):
                # nothing schedulable right now (e.g. waiting for KV blocks)
                time.sleep(0.0005)

    # ------------------------------------------------------------------
    def metrics(self) -> Dict[str, Any]:
        st = self.stats
        return {
            "steps": st.steps,
            "prompt_tokens": st.prompt_tokens,
            "generation_tokens": st.generation_tokens,
            "requests_finished": st.requests_finished,
            "running": len(self.scheduler.running),
            "waiting": len(self.scheduler.waiting),
            "busy_s": st.busy_s,
            "step_ms_ewma": st.step_ms_ewma,
            "kv_blocks_total": self.bm.num_blocks,
            "kv_blocks_free": self.bm.num_free,
            "prefix_cache_queries": self.bm.prefix_queries,
            "prefix_cache_hits": self.bm.prefix_hits,
            "preemptions": self.scheduler.num_preemptions,
            "spec_proposed": st.spec_proposed,
            "spec_accepted": st.spec_accepted,
            "graphs": sorted({k[0] for k in self.runner.graphs}),
            "variant_graphs": sum(1 for k in self.runner.graphs if len(k) > 2),
            "variant_graph_steps": getattr(self.runner, "variant_graph_steps", 0),
            "pipelined_steps": self.pipelined_steps,
            "engine_first_token_ms": st.first_token_s / st.first_token_n * 1e3 if st.first_token_n else None,
            "engine_first_token_steps": st.first_token_steps / st.first_token_n if st.first_token_n else None,
            "load_s": self.load_s,
            "warm_s": self.warm_s,
            "tp_allreduce": ("xgmi-peer-memory" if self.runner.xgmi is not None else ("rccl" if self.runner.tp_active else None)),
        }
