"""Model runner: static device buffers, the ragged forward pass, HIP-graph replay.

One forward pass serves a flat batch of ``T`` tokens (decode tokens and
prefill-chunk tokens mixed).  ``T`` is padded to a bucket; for every bucket the
whole pass — embedding gather, 32 x (norm, QKV GEMM, RoPE + KV write, paged
attention, O GEMM, norm, gate/up GEMM + SwiGLU, down GEMM), final norm, LM head
and greedy argmax — is captured once into a HIP graph (``torch.cuda.CUDAGraph``
records hipGraph nodes on ROCm) and replayed per step, so a decode step costs one
graph launch instead of ~300 kernel launches.  Per-step inputs are packed into
one pinned int32 staging buffer and moved with a single async H2D copy.

With tensor parallelism each rank holds a shard of the heads / MLP columns and the
partial O-proj / down-proj outputs are summed over the ranks.  On GPUs that sum runs
in the hand-written xGMI peer-memory kernels (``kvmini/parallel/xgmi.py``): for
messages up to the slot size ONE launch does split-K slab sum -> one-shot all-reduce
-> residual add -> sums of squares, captured inside the HIP graph like any other
kernel; larger messages (wide prefill steps) take the two-shot kernel or RCCL.  The
scheduling rank hands every step's staged inputs to the other ranks through a
shared-memory ring on the host (``kvmini/parallel/step_channel.py``), so followers
never wait for their GPU to learn what to run next and steps can be launched ahead.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from kvmini.models.llama import LlamaConfig, LlamaWeights, rope_tables

from .scheduler import PLACEHOLDER, Batch

# Token buckets a step is padded to (one captured graph each).  From 128 rows on they grow by a quarter to
# a half, which keeps the padding of a prompt step small: a 1100-token prompt runs as 1280 rows instead of
# 2048 (first token 27.9 -> 21.6 ms on Llama-3-8B), a 600-token one as 640 instead of 1024 (15.9 -> 13.2 ms).
DEFAULT_BUCKETS = (16, 32, 48, 64, 96, 128, 192, 256, 320, 384, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048)


@dataclass
class RunnerConfig:
    max_num_seqs: int = 64
    max_batched_tokens: int = 256
    max_model_len: int = 8192
    block_size: int = 16
    num_blocks: int = 0  # 0 = size from memory
    kv_memory_fraction: float = 0.5
    use_graphs: bool = True
    kernels: str = "auto"  # auto | hip | torch
    return_logits: bool = False
    quantization: Optional[str] = None  # None | fp8 | mxfp4 (weight-only, HIP kernels; int4 / awq / gptq are served as mxfp4)
    kv_cache_dtype: str = "auto"  # auto (the model dtype) | fp8 (e4m3 bytes + one scale per kv head, converted in registers by the attention kernels)
    tp_graphs: bool = True  # capture HIP graphs under tensor parallelism (buckets whose all-reduce runs in the peer-memory kernels)
    tp_rccl_graphs: bool = False  # also capture buckets whose all-reduce is RCCL's (every rank must capture in step)
    tp_custom_allreduce: bool = True  # hand-written xGMI peer-memory kernels for the TP all-reduce (HIP ops on GPUs)
    tp_force_collectives: bool = False  # run the TP collectives even for a group of one rank (tests)


def cfg_hidden(cfg: LlamaConfig) -> int:
    return cfg.hidden_size


def select_ops(kernels: str, device: torch.device) -> Any:
    """``hip`` = hand-written gfx950 kernels (required on a GPU unless ``torch``
    is requested explicitly: a missing extension is an error, not a silent
    fallback); ``torch`` = reference ops."""
    if kernels == "torch" or (kernels == "auto" and device.type != "cuda"):
        from kvmini.ops import torch_ops

        return torch_ops
    from kvmini.ops import hip_ops

    hip_ops.require()
    return hip_ops


class _Stage:
    """One pinned host image of the step inputs (+ the host copy of the sampled ids)."""

    def __init__(self, n_ints: int, o_bt: int, smax: int, max_blocks_per_seq: int, tmax: int, pin: bool):
        self.cpu = torch.zeros(n_ints, dtype=torch.int32, pin_memory=pin)
        self.np = self.cpu.numpy()
        self.bt_np = self.np[o_bt : o_bt + smax * max_blocks_per_seq].reshape(smax, max_blocks_per_seq)
        # per slot: ((request id, blocks epoch), block ids already staged in THIS image)
        self.bt_owner: List[Tuple[Any, int]] = [(None, 0)] * smax
        self.out_cpu = torch.zeros(tmax, dtype=torch.int64, pin_memory=pin)
        self.event: Any = None  # recorded behind the step (and the copy of its ids)


@dataclass
class StepHandle:
    """A launched step whose sampled ids have not been read back yet."""

    n_samples: int
    logits: Optional[torch.Tensor]
    stage: _Stage
    key: Tuple[int, int] = (0, 0)  # (token bucket, step kind): steps of one key take alike
    timed: bool = False  # launched at once on top of a step in flight: its wall time is its device time


class ModelRunner:
    def __init__(self, weights: LlamaWeights, rcfg: RunnerConfig, tp_group: Any = None):
        self.w = weights
        self.cfg: LlamaConfig = weights.cfg
        self.rcfg = rcfg
        self.device = weights.device
        self.tp_group = tp_group
        self.tp_size = weights.tp_size
        # the collective code path runs whenever a group is given — also for a group of one
        # (tests: the RCCL all-reduce and its capture into HIP graphs on a single GPU)
        self.tp_active = tp_group is not None and (weights.tp_size > 1 or rcfg.tp_force_collectives)
        self.ops = select_ops(rcfg.kernels, self.device)
        if hasattr(self.ops, "prepare_weights") and self.cfg.head_dim not in (64, 128) and self.cfg.head_dim < 128 and not hasattr(weights, "_hip_prepared"):
            # head widths the attention kernels are not specialised for (80, 96, ...): zero-pad every
            # head to 128 instead of dropping to the reference attention path
            from kvmini.models.llama import pad_head_dim

            weights = pad_head_dim(weights, 128)
            self.w, self.cfg = weights, weights.cfg
        if hasattr(self.ops, "prepare_weights") and not hasattr(weights, "_hip_prepared"):
            # query heads per kv head outside the kernels' set (5, 16, ...): store each K/V head
            # several times, so that the ratio becomes one of the set
            from kvmini.ops.hip_ops import GQA_RATIOS

            ratio = self.cfg.num_heads // self.cfg.num_kv_heads
            if ratio not in GQA_RATIOS:
                from kvmini.models.llama import replicate_kv_heads

                factor = min(f for f in range(2, ratio + 1) if ratio % f == 0 and ratio // f in GQA_RATIOS)
                weights = replicate_kv_heads(weights, factor)
                self.w, self.cfg = weights, weights.cfg
        if hasattr(self.ops, "prepare_weights"):
            # one-time re-order of every linear weight into MFMA fragment order
            self.ops.prepare_weights(weights, rcfg.quantization)
        elif rcfg.quantization == "fp8":
            # reference ops: emulate weight-only FP8 by a quantise/dequantise round trip
            from kvmini.models.llama import fake_quantize_fp8

            fake_quantize_fp8(weights)
        elif rcfg.quantization in ("mxfp4", "int4", "awq", "autoawq", "gptq", "fp4"):
            from kvmini.models.llama import fake_quantize_mxfp4

            fake_quantize_mxfp4(weights)
        cfg = self.cfg
        self.nh = cfg.num_heads // self.tp_size
        self.nkv = cfg.num_kv_heads // self.tp_size
        self.hd = cfg.head_dim
        self.scale = cfg.softmax_scale
        self.block_size = rcfg.block_size
        self.max_blocks_per_seq = (rcfg.max_model_len + rcfg.block_size - 1) // rcfg.block_size
        self.cos, self.sin = rope_tables(cfg, rcfg.max_model_len, self.device)

        if rcfg.kv_cache_dtype not in ("auto", "fp8", "fp8_e4m3"):
            raise ValueError(f"kv_cache_dtype must be auto or fp8, got {rcfg.kv_cache_dtype!r}")
        self.kv_fp8 = rcfg.kv_cache_dtype != "auto"
        self.xgmi = None
        if self.tp_active and self.device.type == "cuda" and rcfg.tp_custom_allreduce and hasattr(self.ops, "prepare_weights"):
            from kvmini.parallel.xgmi import XgmiComm

            # slots sized for the widest step the fused epilogue should serve: 256 rows of the residual stream
            slot = max(4 << 20, 256 * cfg_hidden(self.cfg) * 2)
            self.xgmi = XgmiComm(group=tp_group, device=self.device, slot_bytes=slot) if self.tp_size > 1 else XgmiComm(device=self.device, slot_bytes=slot, world=1, rank=0)
        # sampled / guided ids travel from the scheduling rank to the followers on the device, behind the
        # sampling ops (a communicator of its own: one call per step on every rank, so its slot parity
        # never depends on what the captured graphs hold)
        self.xgmi_ids = None
        if self.xgmi is not None and self.tp_size > 1:
            from kvmini.parallel.xgmi import XgmiComm

            self.xgmi_ids = XgmiComm(group=tp_group, device=self.device, slot_bytes=max(4096, rcfg.max_batched_tokens * 8))
        self.num_blocks = rcfg.num_blocks or self._auto_num_blocks()
        if tp_group is not None and self.tp_size > 1:
            # only the scheduling rank hands out block ids, so every rank's cache must hold the
            # same number of blocks: take the smallest (the rank with the least free memory)
            on_dev = self.device.type == "cuda" and torch.distributed.get_backend(tp_group) == "nccl"
            nb = torch.tensor([self.num_blocks], dtype=torch.int64, device=self.device if on_dev else "cpu")
            torch.distributed.all_reduce(nb, op=torch.distributed.ReduceOp.MIN, group=tp_group)
            self.num_blocks = int(nb.item())
        dt = torch.float8_e4m3fn if self.kv_fp8 else weights.embed.dtype
        # [layer, k/v, block, kv_head, slot_in_block, head_dim]
        self.kv_cache = torch.zeros((cfg.num_layers, 2, self.num_blocks, self.nkv, self.block_size, self.hd), dtype=dt, device=self.device)
        # FP8 cache: stored byte = e4m3(x / scale), one scale per (layer, K or V, kv head).  1.0
        # unless calibrated (set_kv_scales): K and V of a normalised residual stream stay far
        # inside e4m3's +-448.
        self.kv_scale = torch.ones((cfg.num_layers, 2, self.nkv), dtype=torch.float32, device=self.device) if self.kv_fp8 else None

        # the persistent decode-layers kernel (HIP ops, one rank, shapes it is built for)
        self.decode_layers = None
        dl = getattr(self.ops, "DecodeLayers", None)
        if dl is not None and self.device.type == "cuda" and not self.tp_active and dl.supported(weights, self.kv_cache[0, 0], self.nh, self.nkv):
            self.decode_layers = dl(weights, self.kv_cache, self.kv_scale, self.nh, self.nkv)

        self.buckets = tuple(b for b in DEFAULT_BUCKETS if b < rcfg.max_batched_tokens) + (rcfg.max_batched_tokens,)
        tmax = self.buckets[-1]
        self.tmax = tmax
        smax = rcfg.max_num_seqs
        # packed int32 staging layout: tokens | positions | slot_mapping | tok_seq | ctx_lens | sample_idx | block_tables
        self._off = {}
        off = 0
        for name, n in (("tokens", tmax), ("positions", tmax), ("slots", tmax), ("tok_seq", tmax), ("ctx", tmax), ("sample_idx", tmax)):
            self._off[name] = (off, n)
            off += n
        self._off["block_tables"] = (off, smax * self.max_blocks_per_seq)
        off += smax * self.max_blocks_per_seq
        # prompt chunks (more than one token of a sequence in the step) are served by the tiled
        # attention kernel: 16-query tiles (first row, valid queries, first position, block-table
        # row), and a context-length array that is zero for those tokens so that the per-token
        # kernel skips them
        # With a step budget that can fill the CUs with groups of four tiles (and head groups of at
        # most 4), every chunk's tiles start at a multiple of four (unused tiles in between), so that
        # a step deep inside long prompts can run the 8-wave attention kernel, which takes such a
        # group per workgroup (measured: +5 % at 14 k tokens of context, +10 % at 129 k; behind the
        # 4-wave kernels near position 0 and on grids that leave CUs idle).  The padded list serves
        # the other kernels as well: their tile pairs stay aligned.
        self.quads_min_groups = int(os.environ.get("KVMINI_QUADS_MIN_GROUPS", "256"))  # groups x kv heads a step needs
        self.quads_min_pos = int(os.environ.get("KVMINI_QUADS_MIN_POS", "8192"))       # first position of every multi-tile chunk
        self.tile_quads = bool(hasattr(self.ops, "pad_tiles_to_quads") and self.device.type == "cuda" and self.hd == 128 and self.nh // self.nkv <= 4
                               and (tmax // 64) * self.nkv >= self.quads_min_groups)
        self.max_tiles = tmax // 16 + smax
        if self.tile_quads:
            self.max_tiles = (tmax // 16 + 4 * smax + 3) // 4 * 4
        self._off["ctx_decode"] = (off, tmax)
        off += tmax
        self._off["tiles"] = (off, 4 * self.max_tiles)
        off += 4 * self.max_tiles
        # decode tokens with long contexts riding along in a step that also carries prompt chunks:
        # up to DEC_SPLIT_ROWS of them are listed here (-1 = unused) and served by the split-KV
        # attention kernel, which spreads each context over many workgroups — the per-token kernel
        # walks a context with one workgroup per kv head (a 128k-token context: 1.9 ms per layer)
        self.dec_split_ok = bool(hasattr(self.ops, "split_decode_attention") and self.device.type == "cuda" and self.hd == 128 and self.block_size == 16)
        # a few prompt tiles deep inside a long context: every tile's key range is shared out over several workgroups
        self.pf_split_ok = bool(hasattr(self.ops, "prefill_attention_split") and self.device.type == "cuda" and self.hd == 128 and self.block_size == 16
                                and os.environ.get("KVMINI_PREFILL_SPLIT", "1") != "0")
        # steps of the special attention forms get a graph of their own the second time their shape comes up
        self.variant_graphs_max = int(os.environ.get("KVMINI_VARIANT_GRAPHS", "24"))
        self._variant_seen: Dict[Tuple[int, ...], int] = {}
        self.variant_graph_steps = 0
        self._off["dec_rows"] = (off, self.DEC_SPLIT_ROWS)
        off += self.DEC_SPLIT_ROWS
        # step header (tensor parallelism: rank 0 broadcasts the whole staging buffer, followers
        # read the step shape from here): t_pad, n_samples, max_ctx, command (0 = run, 1 = stop)
        self._off["header"] = (off, 8)
        off += 8
        # token feed (pipelined stepping): feed[i] = r + 1 -> token i is row r of the previous
        # step's sampled ids, read on the device; 0 -> the staged token
        self._off["feed"] = (off, tmax)
        off += tmax
        pin = self.device.type == "cuda"
        # two host staging buffers, used alternately: a step can be staged while the copy of the
        # previous one may still be in flight
        self.stages = [_Stage(off, self._off["block_tables"][0], smax, self.max_blocks_per_seq, tmax, pin) for _ in range(2)]
        self._cur = 0
        self.stage_dev = torch.zeros(off, dtype=torch.int32, device=self.device)
        self.out_ids = torch.zeros(tmax, dtype=torch.int64, device=self.device)
        self.logits_buf: Optional[torch.Tensor] = None
        self.spec_width = 1  # positions sampled per sequence per step (k+1 with speculation)
        self.s_cap = tmax
        self.set_spec_width(1)
        # (token bucket, step kind) -> (graph, logits); kind 0 = mixed step, 1 = decode-only (fused
        # RoPE + attention kernel), 2 = decode-only with the context split over several workgroups
        self.graphs: Dict[Tuple[int, ...], Any] = {}
        self._graph_pool = None
        # Weight read-ahead on a second stream beside the decode attention kernel (a parallel branch of the
        # captured graph, joined at the end of the step): KVMINI_SIDE_PREFETCH = workgroups of the reading kernel.
        # OFF by default.  Measured on MI355X (Llama-3-8B, batch 20, graph replay): 3.33 ms per step without it,
        # 4.18 / 4.08 / 4.00 ms with 32 / 96 / 256 workgroups — a fork per layer inside a replayed graph costs
        # ~20 us of cross-queue signalling, four times what the output projection can gain from a warm cache.
        self._side_wgs = int(os.environ.get("KVMINI_SIDE_PREFETCH", "0")) if self.device.type == "cuda" else 0
        self._side_stream = torch.cuda.Stream(device=self.device) if self._side_wgs > 0 else None
        self.steps = 0
        # tensor parallelism: the scheduling rank publishes every step's staged image to the
        # followers through host shared memory (no device broadcast, no follower read-back)
        self.channel = None
        if tp_group is not None and self.tp_size > 1:
            from kvmini.parallel.step_channel import StepChannel

            me = torch.distributed.get_rank(tp_group)
            box = [None]
            if me == 0:
                import os as _os

                self.channel = StepChannel(f"kvmini_tp_{_os.getpid()}_{id(self) & 0xffff:x}", off, ring=4, create=True, n_readers=self.tp_size - 1)
                box[0] = self.channel.name
            torch.distributed.broadcast_object_list(box, src=self._tp_src(), group=tp_group)
            if me != 0:
                self.channel = StepChannel(box[0], off, ring=4, reader=me - 1)
            torch.distributed.barrier(group=tp_group)  # everybody is attached before the first image

    def set_spec_width(self, width: int) -> None:
        self.spec_width = width
        need = self.rcfg.max_num_seqs * width
        self.s_cap = next((b for b in self.buckets if b >= need), self.buckets[-1])

    # ------------------------------------------------------------------
    def _auto_num_blocks(self) -> int:
        cfg, r = self.cfg, self.rcfg
        per_block = cfg.num_layers * 2 * self.nkv * r.block_size * self.hd * (1 if self.kv_fp8 else self.w.embed.element_size())
        want = r.max_num_seqs * self.max_blocks_per_seq + 64
        if self.device.type == "cuda":
            free, _total = torch.cuda.mem_get_info(self.device)
            by_mem = int(free * r.kv_memory_fraction) // per_block
            return max(64, min(want, by_mem))
        return max(64, min(want, 4096))

    def set_kv_scales(self, scales: torch.Tensor) -> None:
        """Per-(layer, K / V, kv head) scales of the FP8 cache (``[layers, 2, nkv]``, e.g. the
        ``k_scale`` / ``v_scale`` of a calibrated checkpoint).  In place: captured graphs read the
        same buffer.  The cache must be empty (bytes written under other scales would be misread)."""
        if self.kv_scale is None:
            raise RuntimeError("the KV cache is not FP8")
        self.kv_scale.copy_(scales.to(self.kv_scale))

    def _dev(self, name: str, n: Optional[int] = None) -> torch.Tensor:
        off, full = self._off[name]
        return self.stage_dev[off : off + (full if n is None else n)]

    def sample_rows(self, t_pad: int) -> int:
        """Rows of the LM-head GEMM for a token bucket: at most one per sequence
        (times the speculative width), never more than the bucket."""
        return min(t_pad, self.s_cap)

    def bucket_for(self, t: int) -> int:
        for b in self.buckets:
            if t <= b:
                return b
        raise ValueError(f"{t} tokens exceed max_batched_tokens={self.buckets[-1]}")

    # ------------------------------------------------------------------
    # Widest token bucket that gets a decode-only graph (RoPE + KV write + attention in one launch instead of three).
    # Wider buckets gain nothing from it — measured on MI355X, Llama-3-8B, per step with the limit at 64 / 256:
    # batch 96: 5.39 / 5.36 ms, 128: 5.50 / 5.45, 192: 6.21 / 6.26, 256: 5.67 / 5.68 (one workgroup per token and
    # kv head: from ~500 workgroups on the one launch takes as long as the three) — and cost start-up time.
    DECODE_FUSED_MAX_ROWS = int(os.environ.get("KVMINI_DECODE_FUSED_MAX_ROWS", "64"))
    # With an e4m3 cache the limit is 32 rows: from 33 rows on the batch decode-attention kernel (three K/V stages in
    # flight per wave) reads the cache at least as fast as the one-launch kernel once contexts are long — one launch /
    # three launches, batch 48: 3.85 / 3.86 ms at 290 tokens of context, 4.38 / 4.29 at 1070, 6.56 / 6.39 at 4180; batch 64:
    # 4.11 / 4.10, 4.61 / 4.60, 7.12 / 7.17 (bf16 cache at batch 64: 4.20 / 4.29, 5.25 / 5.33, 9.72 / 9.81 — it stays on the
    # one launch).
    DECODE_FUSED_MAX_ROWS_KV8 = int(os.environ.get("KVMINI_DECODE_FUSED_MAX_ROWS_KV8", "32"))
    # contexts from this many tokens on take the split-KV variant of a decode-only step (when the
    # step's shape has one: few rows); below it the extra workgroups only cost launch time
    SPLIT_KV_MIN_CTX = 2048
    # steps that also carry prompt chunks: at most this many decode tokens, the longest context
    # from DEC_SPLIT_MIN_CTX tokens on (such steps capture their graph on demand, see _variant_graph)
    DEC_SPLIT_ROWS = 16  # from ~64 consecutive decode rows on the per-token kernel fills the chip as well (measured equal)
    DEC_SPLIT_MIN_CTX = int(os.environ.get("KVMINI_DEC_SPLIT_MIN_CTX", "4096"))
    # Prompt chunks of 17 to ~512 tokens whose queries see at least this many keys go to the split form of the tile kernel
    # (measured on MI355X behind a 128k-token prefix, per layer: 64 new tokens 2505 -> 257 us, 256: 2480 -> 632 us, 512: 2513 -> 1130 us).
    PF_SPLIT_MAX_TILES = 32
    PF_SPLIT_MIN_CTX = int(os.environ.get("KVMINI_PREFILL_SPLIT_MIN_CTX", "4096"))

    def step_kind(self, t_pad: int, decode_only: bool, max_ctx: int) -> int:
        if not decode_only:
            return 0
        splits = getattr(self.ops, "decode_attention_splits", None)
        return 2 if splits is not None and max_ctx >= self.SPLIT_KV_MIN_CTX and splits(t_pad, self.nkv) > 1 else 1

    def forward(self, t_pad: int, max_blocks: Optional[int] = None, decode_only: int = 0, quads: bool = False, dec_split: int = 0, pf_split: int = 0) -> torch.Tensor:
        """The ragged forward pass over the first ``t_pad`` staged tokens; writes
        sampled ids into ``out_ids[:t_pad]`` and returns the logits ``[t_pad, V]``.
        ``decode_only`` (a step kind, see ``self.graphs``): every staged token is the only new token of its sequence, which
        allows RoPE + KV write + attention to run as one kernel.  ``quads``: the step's prompt chunks go to
        the 8-wave attention kernel (see ``tile_quads``).  ``pf_split``: ``shares | tiles << 8`` — the step's first ``tiles`` prompt tiles
        go to the split form of the tile kernel with ``shares`` key ranges each (see ``PF_SPLIT_MAX_TILES``)."""
        ops, cfg, w = self.ops, self.cfg, self.w
        tokens = self._dev("tokens", t_pad)
        positions = self._dev("positions", t_pad)
        slots = self._dev("slots", t_pad)
        tok_seq = self._dev("tok_seq", t_pad)
        ctx = self._dev("ctx", t_pad)
        sample_idx = self._dev("sample_idx", t_pad)
        bt = self._dev("block_tables").view(self.rcfg.max_num_seqs, self.max_blocks_per_seq)

        # Steps the weight-streaming GEMM serves: the residual kernels and the fused decode attention
        # read the next projection's weights ahead while HBM would idle (hip_ops.prefetch_plan).
        plan = getattr(ops, "prefetch_plan", None)
        if self.tp_active or self.device.type != "cuda":
            plan = None

        def pf(wt: Any, site: str) -> Dict[str, Any]:
            p = plan(wt, t_pad, site) if plan is not None else None
            return {"prefetch": p} if p is not None else {}

        x = ops.embed_tokens(w.embed, tokens, self._dev("feed", t_pad), self.out_ids)
        if decode_only == 1 and self.decode_layers is not None and t_pad <= 64:
            # plain decoding step: every layer in one persistent launch (phases separated by device-wide
            # barriers instead of kernel boundaries, weight chunks in flight across them)
            h, residual = self._add_rmsnorm(x, None, w.layers[0].attn_norm, False)
            self.decode_layers.run(residual, h.ss, positions, self.cos, self.sin, slots, bt, tok_seq, ctx, cfg.rms_eps, self.scale)
            hs = ops.select_rows(h, sample_idx[: self.sample_rows(t_pad)])
            logits = ops.linear(hs, w.lm_head)
            self.out_ids[: logits.shape[0]].copy_(ops.greedy_sample(logits))
            return logits
        side = self._side_stream if (decode_only == 1 and plan is not None and hasattr(ops, "prefetch_weight")) else None
        residual = None  # the first residual add is a copy
        pending_reduce = False  # x is a row-parallel projection's partial output that still has to be summed over the TP ranks
        h = None  # set when a projection already folded its output into the residual stream (fused epilogue)
        for li, lw in enumerate(w.layers):
            if h is None:
                h, residual = self._add_rmsnorm(x, residual, lw.attn_norm, pending_reduce, **pf(lw.wqkv, "qkv"))
            qkv = ops.linear(h, lw.wqkv, allow_slabs=True)
            if lw.bqkv is not None:
                qkv = ops.add_bias(qkv, lw.bqkv)
            kc, vc = self.kv_cache[li, 0], self.kv_cache[li, 1]
            kw = {"kv_scale": self.kv_scale[li]} if self.kv_fp8 else {}
            if decode_only and t_pad <= (self.DECODE_FUSED_MAX_ROWS_KV8 if self.kv_fp8 else self.DECODE_FUSED_MAX_ROWS) and self._fused_ok(qkv, kc):
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        ops.prefetch_weight(lw.wo, self._side_wgs)
                a = ops.rope_attention_fused(qkv, positions, self.cos, self.sin, kc, vc, slots, bt, tok_seq, ctx, self.nh, self.scale, n_split=None if decode_only == 2 else 1, **kw,
                                             **(pf(lw.wo, "o") if decode_only == 1 else {}))
            elif not decode_only and self._tiled_ok(kc):
                # decode tokens: one workgroup per token; prompt chunks: 16-query MFMA tiles
                q = ops.rope_kv_write(qkv, positions, self.cos, self.sin, kc, vc, slots, self.nh, self.nkv, self.hd, **kw)
                if dec_split:  # few decode tokens, long contexts: split-KV over the listed rows (padding rows stay zero)
                    a = ops.split_decode_attention(q, kc, vc, bt, tok_seq, self._dev("ctx_decode", t_pad), self._dev("dec_rows", dec_split),
                                                   torch.zeros((t_pad, self.nh * self.hd), dtype=q.dtype, device=q.device), self.scale, **kw)
                else:
                    a = ops.paged_attention(q, kc, vc, bt, tok_seq, self._dev("ctx_decode", t_pad), self.scale, max_blocks, **kw)
                if pf_split:
                    ops.prefill_attention_split(q, kc, vc, bt, self._dev("tiles").view(4, self.max_tiles), pf_split >> 8, a, self.scale, pf_split & 255, **kw)
                else:
                    ops.prefill_attention(q, kc, vc, bt, self._dev("tiles").view(4, self.max_tiles), a, self.scale, **kw, **({"quads": True} if quads else {}))
            else:
                q = ops.rope_kv_write(qkv, positions, self.cos, self.sin, kc, vc, slots, self.nh, self.nkv, self.hd, **kw)
                a = ops.paged_attention(q, kc, vc, bt, tok_seq, ctx, self.scale, max_blocks, **kw)
            h, residual = self._proj_residual(a, lw.wo, residual, lw.mlp_norm, **pf(lw.wgu, "gu"))
            act = ops.linear_swiglu(h, lw.wgu)
            fused = self._fused_residual(act, lw.wdown, residual)
            if fused is not None:
                h, residual = fused
            else:
                h = None
                x = ops.linear(act, lw.wdown, allow_slabs=True)
                pending_reduce = True
        if h is None:
            h, residual = self._add_rmsnorm(x, residual, w.final_norm, pending_reduce, **pf(w.lm_head, "head"))
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        hs = ops.select_rows(h, sample_idx[: self.sample_rows(t_pad)])
        logits = ops.linear(hs, w.lm_head)
        self.out_ids[: logits.shape[0]].copy_(ops.greedy_sample(logits))
        return logits

    def _fused_residual(self, x: torch.Tensor, w: Any, residual: Optional[torch.Tensor]) -> Optional[Tuple[Any, torch.Tensor]]:
        """Row-parallel projection + residual update + RMSNorm statistics in one launch, where the
        ops have it (HIP: split-K GEMM whose last workgroup per feature group folds the slabs into
        the residual) and no cross-rank sum has to happen in between."""
        fn = getattr(self.ops, "linear_residual", None)
        if fn is None or self.tp_active or residual is None:
            return None
        return fn(x, w, residual, self.cfg.rms_eps)

    def _proj_residual(self, x: torch.Tensor, w: Any, residual: torch.Tensor, norm_w: torch.Tensor, **kw: Any) -> Tuple[Any, torch.Tensor]:
        fused = self._fused_residual(x, w, residual)
        if fused is not None:
            return fused
        return self._add_rmsnorm(self.ops.linear(x, w, allow_slabs=True), residual, norm_w, True, **kw)

    def _add_rmsnorm(self, x: Any, residual: Optional[torch.Tensor], norm_w: torch.Tensor, partial: bool, **kw: Any) -> Tuple[Any, torch.Tensor]:
        """Residual update + norm in front of a projection.  ``partial``: ``x`` is this rank's
        share of a row-parallel projection and is summed over the TP ranks on the way — in the
        fused peer-memory kernel (slab sum -> all-reduce -> residual add -> sums of squares, one
        launch) when the message fits its slots, else all-reduced first."""
        ops, eps = self.ops, self.cfg.rms_eps
        if not (partial and self.tp_active):
            return ops.add_rmsnorm(x, residual, norm_w, eps, **kw)
        if self.xgmi is not None:
            src = x.slabs if hasattr(x, "slabs") else (x if x.is_contiguous() else x.contiguous())
            t, hdim = (src.shape[1], src.shape[2]) if src.dim() == 3 else src.shape
            if self.xgmi.fits_residual(t, hdim):
                has_res = residual is not None
                if residual is None:
                    residual = torch.empty((t, hdim), dtype=torch.bfloat16, device=src.device)
                ss = self.xgmi.residual_allreduce(src, residual, has_res)
                return ops.Normed(residual, ss, eps), residual
        return ops.add_rmsnorm(self._tp_all_reduce(x), residual, norm_w, eps)

    def _tp_all_reduce(self, x: Any) -> torch.Tensor:
        """Sum a row-parallel projection's partial output over the TP ranks.  Split-K slabs are
        first summed locally into one bf16 ``[T, H]`` tensor, so that ONE message per
        projection crosses xGMI (T x H x 2 bytes: 0.5 MiB at 32 tokens of a 70B model) instead
        of one per slab; the all-reduce itself is RCCL's."""
        if hasattr(x, "reduce"):  # hip_ops.Partial
            x = x.reduce()
        x = x.contiguous()
        if self.xgmi is not None and x.dtype == torch.bfloat16 and self.xgmi.fits(x.numel() * 2) and x.numel() % (8 * self.xgmi.world) == 0:
            return self.xgmi.all_reduce(x, two_shot=True)  # wide steps: reduce-scatter + all-gather through peer memory
        if self.tp_size > 1:
            if x.is_cuda and torch.distributed.get_backend(self.tp_group) != "nccl":
                # a host-side group (tests: several ranks sharing one GPU): sum through the host
                xc = x.float().cpu()
                torch.distributed.all_reduce(xc, group=self.tp_group)
                x.copy_(xc.to(x.dtype))
            else:
                torch.distributed.all_reduce(x, group=self.tp_group)
        return x

    def _tiled_ok(self, k_cache: torch.Tensor) -> bool:
        fn = getattr(self.ops, "supports_prefill_attention", None)
        return fn is not None and fn(self.nh, k_cache)

    def _fused_ok(self, qkv: Any, k_cache: torch.Tensor) -> bool:
        fn = getattr(self.ops, "supports_fused_decode_attention", None)
        return fn is not None and fn(qkv, self.nh, k_cache)

    # ------------------------------------------------------------------
    def capture_graphs(self, buckets: Optional[Sequence[int]] = None) -> None:
        """Capture one HIP graph per token bucket (largest first so the smaller
        ones share its memory pool)."""
        if self.device.type != "cuda" or not self.rcfg.use_graphs:
            return
        if self.tp_active and not self.rcfg.tp_graphs:
            return
        self._stage_dummy()
        # Nothing may free device memory while a stream is capturing: a garbage-collection pass
        # that drops an older engine's graphs (and their private memory pool) in the middle of a
        # capture aborts the process.  Collect first, then keep the collector off until done.
        import gc

        gc.collect()
        torch.cuda.synchronize(self.device)
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_all(buckets)
        finally:
            if gc_was_enabled:
                gc.enable()
        torch.cuda.synchronize(self.device)

    def _capture_all(self, buckets: Optional[Sequence[int]]) -> None:
        variants = [(b, 0) for b in sorted(buckets or self.buckets, reverse=True)]
        # decode-sized buckets also get the decode-only variant (fused RoPE + attention kernel)
        # (up to the bucket that holds max_num_seqs one-token rows: with 2 sequences that is still the 16-row bucket)
        dec_max = self.bucket_for(min(self.rcfg.max_num_seqs, self.tmax))
        # (wider decode-only buckets, up to 256 rows, get the variant too: RoPE + KV write and the per-token attention
        # kernel without the launch of the prompt-tile kernel that the general graph carries — 5.3 us per layer)
        variants += [(b, 1) for b, _ in variants if b <= dec_max and b <= max(self.DECODE_FUSED_MAX_ROWS, 256) and hasattr(self.ops, "rope_attention_fused")]
        variants += [(b, 2) for b, k in variants if k == 1 and self.step_kind(b, True, self.rcfg.max_model_len) == 2]
        for b, dec in variants:
            key = (b, dec)
            if key in self.graphs:
                continue
            if self.tp_active and self.tp_size > 1 and not self.rcfg.tp_rccl_graphs and not (self.xgmi is not None and self.xgmi.fits(b * self.cfg.hidden_size * 2)):
                continue  # this bucket's all-reduce would be RCCL's: left eager unless asked for (every rank must capture in step)
            # warm-up on a side stream (allocator + lazy init), then capture
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.forward(b, decode_only=dec)
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (HTTP front-end, RCCL watchdog) may
            # touch the HIP runtime while this thread captures
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                logits = self.forward(b, decode_only=dec)
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            self.graphs[key] = (g, logits)

    def _variant_graph(self, t_pad: int, quads: bool, dec_split: int, pf_split: int) -> Optional[Tuple[Any, torch.Tensor]]:
        """The graph of a step shape that uses a special attention form (8-wave prompt kernel, listed-rows split-KV,
        split prompt tiles): their grids depend on the step, so they are not part of the start-up set.  The first step
        of a shape runs eagerly (which also brings the kernels' workspaces into being), the second one captures,
        later ones replay; at most ``variant_graphs_max`` shapes are kept."""
        key = (t_pad, 0, int(quads), dec_split, pf_split)
        entry = self.graphs.get(key)
        if entry is not None:
            return entry
        if self.device.type != "cuda" or not self.rcfg.use_graphs or self.tp_active or not self.graphs:
            return None
        seen = self._variant_seen.get(key, 0)
        self._variant_seen[key] = seen + 1
        if seen == 0 or seen < 0 or sum(1 for k in self.graphs if len(k) > 2) >= self.variant_graphs_max:
            return None
        import gc

        gc_was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()  # nothing may free device memory while the stream captures (see capture_graphs)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                logits = self.forward(t_pad, decode_only=0, quads=quads, dec_split=dec_split, pf_split=pf_split)
        except Exception as e:  # noqa: BLE001 - a shape that does not capture keeps running eagerly; the serving loop must not die over it
            self._variant_seen[key] = -(1 << 30)  # never tried again
            print(f"[runner] no graph for step shape {key}: {type(e).__name__}: {e}", flush=True)
            return None
        finally:
            if gc_was_enabled:
                gc.enable()
        self.graphs[key] = (g, logits)
        return g, logits

    def _run_staged(self, t_pad: int, max_ctx: int, decode_only: int = 0, quads: bool = False, dec_split: int = 0, pf_split: int = 0) -> torch.Tensor:
        if quads or dec_split or pf_split:
            entry = self._variant_graph(t_pad, quads, dec_split, pf_split)
            self.variant_graph_steps += entry is not None
        else:
            entry = self.graphs.get((t_pad, decode_only)) or self.graphs.get((t_pad, 0))
        if entry is not None:
            g, logits = entry
            g.replay()
        else:
            with torch.no_grad():
                logits = self.forward(t_pad, max_blocks=(max_ctx + self.block_size - 1) // self.block_size, decode_only=decode_only, quads=quads, dec_split=dec_split,
                                      pf_split=pf_split)
        self.steps += 1
        return logits

    def _tp_src(self) -> int:
        """Global rank of the tensor-parallel group's first member (the scheduling rank)."""
        g = self.tp_group
        return 0 if g is None else torch.distributed.get_global_rank(g, 0)

    def sync_sampled_ids(self, n: int) -> None:
        """Tensor parallelism: make the first ``n`` entries of ``out_ids`` on every rank those of the
        scheduling rank, with device ops only (queued behind the step, so steps stay launched
        ahead).  Every rank of the group calls this for the same steps (header field ``ids_sync``).
        Peer-memory all-gather when the hand-written collectives are up; else a broadcast of the
        group's backend (RCCL on GPUs; a host-side group of ranks sharing one GPU goes through the
        host, which waits for the step)."""
        if self.tp_size <= 1 or n <= 0:
            return
        ids = self.out_ids[: min((n + 1) // 2 * 2, self.out_ids.numel())]  # whole 16-byte units for the peer-memory kernel
        if self.xgmi_ids is not None:
            table = self.xgmi_ids.all_gather(ids.view(torch.uint8))
            ids.copy_(table[0].view(torch.int64))
            return
        import torch.distributed as dist

        if ids.is_cuda and dist.get_backend(self.tp_group) != "nccl":
            host = ids.cpu()
            dist.broadcast(host, src=self._tp_src(), group=self.tp_group)
            ids.copy_(host)
        else:
            dist.broadcast(ids, src=self._tp_src(), group=self.tp_group)

    def follow(self) -> int:
        """Tensor-parallel follower loop (ranks > 0 of the TP group): take every step's staged
        image from the host channel, copy it to the device and enqueue the same pass on the local
        weight shard (the all-reduces inside ``forward`` pair up with the other ranks').  The loop
        never waits for the GPU except to reuse a pinned staging buffer two steps later, so it
        runs ahead of the device exactly as far as the scheduling rank does.  Returns the number
        of steps served when the stop command arrives."""
        o_h = self._off["header"][0]
        n = 0
        while True:
            sb = self.stages[self._cur]
            self._cur ^= 1
            if sb.event is not None:
                sb.event.synchronize()  # the copy that last read this pinned buffer is done
            while not self.channel.receive(sb.np, timeout_s=1.0):
                if not self.channel.writer_alive():
                    raise RuntimeError("tensor parallelism: the scheduling rank is gone (no stop command was published); this follower exits")
            t_pad, n_samples, max_ctx, cmd, dec, tile_form, dec_split, ids_sync = (int(v) for v in sb.np[o_h : o_h + 8])
            if cmd == 1:
                return n
            self.stage_dev.copy_(sb.cpu, non_blocking=True)
            self._run_staged(t_pad, max_ctx, dec, bool(tile_form & 1), dec_split, tile_form >> 1)
            if ids_sync:
                self.sync_sampled_ids(n_samples)  # the scheduling rank sampled: take its ids instead of the local argmax
            if self.device.type == "cuda":
                if sb.event is None:
                    sb.event = torch.cuda.Event()
                sb.event.record(torch.cuda.current_stream(self.device))
            n += 1

    def stop_followers(self) -> None:
        """Scheduling rank: release the followers from :meth:`follow`."""
        if self.tp_size <= 1 or self.channel is None:
            return
        o_h = self._off["header"][0]
        img = np.zeros(self.stages[0].np.size, dtype=np.int32)
        img[o_h : o_h + 8] = (0, 0, 0, 1, 0, 0, 0, 0)
        try:
            self.channel.publish(img, timeout_s=10.0)
        except TimeoutError:
            pass

    def _stage_dummy(self) -> None:
        """Inputs that are safe for warm-up/capture: every token is padding."""
        o, n = self._off["slots"]
        for sb in self.stages:
            sb.np[:] = 0
            sb.bt_owner = [(None, 0)] * len(sb.bt_owner)
            sb.np[o : o + n] = -1
        self.stage_dev.copy_(self.stages[0].cpu)

    # ------------------------------------------------------------------
    def execute(self, batch: Batch, sampler: Any = None) -> Tuple[List[int], Optional[torch.Tensor]]:
        """Run one step.  Returns (sampled ids, logits) with one row per position
        flagged for sampling, in chunk order: a chunk flagged ``sample`` contributes
        its last position — or, when it carries draft tokens to verify, every
        position from the last accepted token on."""
        return self.collect(self.launch(batch, sampler))

    def collect(self, h: StepHandle) -> Tuple[List[int], Optional[torch.Tensor]]:
        """Wait for a launched step and read its sampled ids."""
        if h.stage.event is not None:
            h.stage.event.synchronize()
        n = h.n_samples
        return h.stage.out_cpu[:n].tolist(), (h.logits[:n] if h.logits is not None else None)

    def launch(self, batch: Batch, sampler: Any = None) -> StepHandle:
        """Stage and enqueue one step without waiting for it.  A one-token chunk whose token
        is still a placeholder (``PLACEHOLDER``: sampled by the previous step, which the host
        has not read back) is fed on the device from row ``seq.pending_row`` of that step's
        ids.  At most one launched step may be uncollected when the next one is launched.
        ``sampler(logits, out_ids)``, if given, queues device ops behind the forward pass that
        replace ids in ``out_ids`` (non-greedy rows) before they are copied out."""
        t = batch.num_tokens
        t_pad = self.bucket_for(t)
        sb = self.stages[self._cur]
        self._cur ^= 1
        st = sb.np
        o_fd = self._off["feed"][0]
        o_tok, o_pos, o_slot, o_seq, o_ctx, o_smp = (self._off[k][0] for k in ("tokens", "positions", "slots", "tok_seq", "ctx", "sample_idx"))
        o_cd, o_tl, mt = self._off["ctx_decode"][0], self._off["tiles"][0], self.max_tiles
        o_dr, n_dec, dec_max_ctx = self._off["dec_rows"][0], 0, 0
        n_tiles = 0
        quad_groups, quad_min_pos = 0, 1 << 30  # groups of four tiles in multi-tile chunks, their shallowest start
        tile_max_ctx = 0  # most keys any tiled prompt chunk of the step sees
        bs = self.block_size
        i = 0
        n_samples = 0
        max_ctx = 1
        for ch in batch.chunks:
            seq = ch.seq
            n = ch.n_tokens
            blocks = seq.blocks
            # block tables only grow while a sequence keeps its slot: write the new tail
            key = (seq.req_id, seq.blocks_epoch)
            owner = sb.bt_owner[seq.slot]
            nb_have = owner[1] if owner[0] == key and owner[1] <= len(blocks) else 0
            if nb_have < len(blocks):
                sb.bt_np[seq.slot, nb_have : len(blocks)] = blocks[nb_have:]
            sb.bt_owner[seq.slot] = (key, len(blocks))
            p0 = ch.start
            # (a draft runner launching proposal j + 1 on top of proposal j: the chunk's one token is the
            # previous proposal, still a placeholder in draft_ids and fed on the device like any other)
            drafting = bool(seq.draft_ids) and seq.draft_feed and p0 >= seq.num_tokens
            if n == 1 and (not seq.draft_ids or drafting):
                npmt = len(seq.prompt_ids)
                tok = seq.draft_ids[p0 - seq.num_tokens] if drafting else (seq.prompt_ids[p0] if p0 < npmt else seq.output_ids[p0 - npmt])
                if tok == PLACEHOLDER:
                    st[o_tok + i] = 0
                    st[o_fd + i] = seq.pending_row + 1
                else:
                    st[o_tok + i] = tok
                    st[o_fd + i] = 0
                st[o_pos + i] = p0
                st[o_slot + i] = blocks[p0 // bs] * bs + p0 % bs
                st[o_seq + i] = seq.slot
                st[o_ctx + i] = p0 + 1
                st[o_cd + i] = p0 + 1
                if n_dec < self.DEC_SPLIT_ROWS:
                    st[o_dr + n_dec] = i
                n_dec += 1
                dec_max_ctx = max(dec_max_ctx, p0 + 1)
                if ch.sample:
                    st[o_smp + n_samples] = i
                    n_samples += 1
                i += 1
            else:
                ids = seq.all_ids
                first_sampled = n - 1
                if seq.draft_ids and not seq.is_prefilling:
                    ids = ids + seq.draft_ids
                    first_sampled = max(0, seq.num_tokens - 1 - p0)
                if ch.n_sample > 0:
                    first_sampled = max(0, n - ch.n_sample)
                pos = np.arange(p0, p0 + n, dtype=np.int32)
                st[o_tok + i : o_tok + i + n] = ids[p0 : p0 + n]
                st[o_pos + i : o_pos + i + n] = pos
                blk = np.asarray(blocks, dtype=np.int32)[pos // bs]
                st[o_slot + i : o_slot + i + n] = blk * bs + pos % bs
                st[o_seq + i : o_seq + i + n] = seq.slot
                st[o_ctx + i : o_ctx + i + n] = pos + 1
                st[o_fd + i : o_fd + i + n] = 0
                # A short chunk deep inside a long context (drafts to verify, the tail of a prompt) would be ONE
                # tile whose workgroup walks the whole context per kv head; as listed rows (each with its own
                # causal length) the split-KV kernel spreads every row's walk over the chip instead.
                as_rows = self.dec_split_ok and n <= self.DEC_SPLIT_ROWS and p0 >= self.DEC_SPLIT_MIN_CTX and n_dec + n <= self.DEC_SPLIT_ROWS
                if as_rows:
                    st[o_cd + i : o_cd + i + n] = pos + 1
                    st[o_dr + n_dec : o_dr + n_dec + n] = np.arange(i, i + n, dtype=np.int32)
                    n_dec += n
                    dec_max_ctx = max(dec_max_ctx, p0 + n)
                else:
                    st[o_cd + i : o_cd + i + n] = -1  # rows of a prompt chunk: left to the tiled kernel (0 = padding row, zeroed)
                if self.tile_quads and not as_rows:
                    while n_tiles % 4:  # the chunk's tiles start a group of four
                        st[o_tl + mt + n_tiles] = 0
                        n_tiles += 1
                    if n > 16:
                        quad_groups += (n + 63) // 64
                        quad_min_pos = min(quad_min_pos, p0)
                if not as_rows:
                    tile_max_ctx = max(tile_max_ctx, p0 + n)
                for off in range(0, 0 if as_rows else n, 16):
                    st[o_tl + n_tiles], st[o_tl + mt + n_tiles], st[o_tl + 2 * mt + n_tiles], st[o_tl + 3 * mt + n_tiles] = i + off, min(16, n - off), p0 + off, seq.slot
                    n_tiles += 1
                if ch.sample:
                    k = n - first_sampled
                    st[o_smp + n_samples : o_smp + n_samples + k] = np.arange(i + first_sampled, i + n, dtype=np.int32)
                    n_samples += k
                i += n
            max_ctx = max(max_ctx, p0 + n)
        # padding tokens: token 0, position 0, no KV write, empty context
        if t_pad > t:
            st[o_tok + t : o_tok + t_pad] = 0
            st[o_pos + t : o_pos + t_pad] = 0
            st[o_slot + t : o_slot + t_pad] = -1
            st[o_seq + t : o_seq + t_pad] = 0
            st[o_ctx + t : o_ctx + t_pad] = 0
            st[o_cd + t : o_cd + t_pad] = 0
            st[o_fd + t : o_fd + t_pad] = 0
        st[o_smp + n_samples : o_smp + t_pad] = 0
        st[o_tl + mt + n_tiles : o_tl + 2 * mt] = 0  # unused tiles: zero valid queries

        # plain decoding step: every chunk is one token (no prompt chunk, no drafts to verify)
        decode_only = self.step_kind(t_pad, all(ch.n_tokens == 1 for ch in batch.chunks), max_ctx)
        o_h = self._off["header"][0]
        quads = int(self.tile_quads and not decode_only and quad_groups * self.nkv >= self.quads_min_groups and quad_min_pos >= self.quads_min_pos)
        # a few tiles, deep inside a long context: key ranges shared out (the tile count rounded up to a power of two: fewer graph shapes)
        pf_split = 0
        if self.pf_split_ok and not decode_only and not quads and 0 < n_tiles <= self.PF_SPLIT_MAX_TILES and tile_max_ctx >= self.PF_SPLIT_MIN_CTX:
            pf_tiles = min(self.max_tiles, max(2, 1 << (n_tiles - 1).bit_length()))
            shares = self.ops.prefill_split_plan(pf_tiles, self.nh, self.nkv, tile_max_ctx)
            if shares >= 2:
                pf_split = shares | (pf_tiles << 8)
        # (the header carries the listed rows rounded up to a power of two: fewer rows, more workgroups per context)
        dec_split = 0
        if self.dec_split_ok and not decode_only and 0 < n_dec <= self.DEC_SPLIT_ROWS and dec_max_ctx >= self.DEC_SPLIT_MIN_CTX:
            dec_split = 1 << (n_dec - 1).bit_length()
        st[o_dr + min(n_dec, self.DEC_SPLIT_ROWS) : o_dr + self.DEC_SPLIT_ROWS] = -1
        ids_sync = int(sampler is not None and self.tp_size > 1)  # the followers must take this rank's sampled ids
        st[o_h : o_h + 8] = (t_pad, n_samples, max_ctx, 0, decode_only, quads | (pf_split << 1), dec_split, ids_sync)
        if self.channel is not None:
            # every rank runs the same step: hand the staged image to the followers first (host
            # shared memory, see follow()), so that they enqueue it while this rank does
            self.channel.publish(st)
        self.stage_dev.copy_(sb.cpu, non_blocking=True)
        logits = self._run_staged(t_pad, max_ctx, decode_only, bool(quads), dec_split, pf_split)
        if sampler is not None:
            sampler(logits, self.out_ids)
            if ids_sync:
                self.sync_sampled_ids(n_samples)
        # the ids leave the device right behind the step, before a later launch overwrites them
        sb.out_cpu[:n_samples].copy_(self.out_ids[:n_samples], non_blocking=True)
        if self.device.type == "cuda":
            if sb.event is None:
                sb.event = torch.cuda.Event()
            sb.event.record(torch.cuda.current_stream(self.device))
        return StepHandle(n_samples, logits, sb, (t_pad, decode_only))
