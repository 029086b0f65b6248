"""Engine ops backed by the gfx950 HIP kernels (same call surface as ``torch_ops``).

RMSNorm is *folded*: at load time each norm's gain is multiplied into the
columns of the projection that consumes it (``W' = W * diag(gain)``), so at run
time ``add_rmsnorm`` only updates the residual stream and records per-row sums of
squares (a fully parallel kernel), the projection reads the un-normalised
residual, and the ``1/rms`` row scale is applied in the projection's epilogue
(or by the consumer of its split-K slabs).  :class:`Normed` and :class:`Partial`
carry that bookkeeping between ops.

Linear layers run on two hand-written MFMA kernels that read the SAME weights,
re-ordered once, at load time, into MFMA fragment order (:class:`PackedLinear`):
the weight-streaming kernel (``csrc/stream_gemm.hip``) up to 128 token rows, where
the op is bound by reading W once, and the tiled kernel (``csrc/wide_gemm.hip``:
both operands staged through LDS by LDS-DMA, software-pipelined MFMA loop) above,
where it is compute-bound.  No vendor GEMM is called and no second (row-major)
copy of the weights exists.  For narrow outputs both kernels can split K across
workgroups and return fp32 partial-sum *slabs* ``[splits, T, N]``; the next op
(:func:`add_rmsnorm` or :func:`rope_kv_write`) accepts slabs and sums them in its
prologue, so the reduction costs no extra launch.
"""

from __future__ import annotations

import functools
import os

from typing import Any, Dict, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

from . import gpu_available, native

MAX_M = 64
GQA_RATIOS = (1, 2, 3, 4, 6, 7, 8)  # query heads per kv head the attention kernels are instantiated for
WIDE_M = 256  # token rows the 8 / 16-tile variants of the streaming GEMM cover
STREAM_MAX_M = 128  # above this many token rows the tiled (compute-bound) GEMM takes over
# k-steps (of 32) per pipelined chunk.  Measured on MI355X (Llama-3-8B shapes): 8 (8 KiB of
# weights in flight per wave) wins by ~2 % for a single 16-row token tile, 4 wins by 3-7 % from
# 32 rows up (smaller LDS x-tiles, more workgroups resident per CU).
KS_PER_CHUNK = 8
MAX_SPLITS = 16


def require() -> None:
    """Fail loudly when the HIP path cannot run (missing build or no GPU)."""
    native()
    if not gpu_available():
        raise RuntimeError("kvmini HIP kernels need a ROCm GPU (torch.cuda.is_available() is False)")


class PackedLinear:
    """A bf16 ``[N, K]`` weight re-ordered into MFMA fragment order
    ``[N/16][K/32][64][8]`` (N zero-padded to a multiple of 64)."""

    __slots__ = ("wp", "n", "k", "n_pad", "ks", "wscale", "bits")

    def __init__(self, w: torch.Tensor, fp8: bool = False, mxfp4: bool = False):
        """``fp8=True`` stores the weight as OCP e4m3 bytes with one fp32 scale per
        output row (weight-only quantisation: activations and accumulation stay
        bf16 / fp32), halving the bytes the decode GEMM streams from HBM.
        ``mxfp4=True`` stores OCP MXFP4: two e2m1 values per byte and one E8M0 scale byte per
        32 values along K (0.53 bytes per weight), widened to bf16 in registers by the
        streaming GEMM (``in_features`` must be a multiple of 256)."""
        n, k = w.shape
        if w.dtype != torch.bfloat16 or not w.is_cuda:
            raise ValueError("PackedLinear needs a bf16 GPU weight")
        ks = KS_PER_CHUNK if k % (32 * KS_PER_CHUNK) == 0 else 4
        if k % (32 * ks):
            raise ValueError(f"in_features {k} must be a multiple of 128 for the streaming GEMM")
        n_pad = (n + 63) // 64 * 64
        if n_pad != n:
            w = torch.cat((w, torch.zeros(n_pad - n, k, dtype=w.dtype, device=w.device)), 0)
        self.wscale: Optional[torch.Tensor] = None
        self.bits = 4 if mxfp4 else 8 if fp8 else 16
        if mxfp4:
            from kvmini.models.llama import quantize_mxfp4

            if k % 256:
                raise ValueError(f"in_features {k} must be a multiple of 256 for MXFP4 weights")
            self.wp, self.wscale = pack_mxfp4(*quantize_mxfp4(w))
            _reserve_w4_scratch(w.device, n_pad * k)
        elif fp8:
            from kvmini.models.llama import quantize_fp8_rows

            q, scale = quantize_fp8_rows(w)
            self.wp = native().pack_weight_fp8(q.view(torch.uint8).contiguous())
            self.wscale = scale.contiguous()
        else:
            self.wp = native().pack_weight(w.contiguous())
        self.n, self.k, self.n_pad, self.ks = n, k, n_pad, ks

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.n, self.k)

    def numel(self) -> int:
        return self.wp.numel()

    def element_size(self) -> int:
        return self.wp.element_size()


def pack_mxfp4(codes: torch.Tensor, scales: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MXFP4 codes ``[N, K]`` (0..15) and scale bytes ``[N, K/32]`` in the streaming GEMM's
    fragment order.  Weights: uint8 ``[N/16, K/128, 64 lanes, 16]`` — lane ``l`` holds, for each
    of the four k-steps ``u`` of a 128-column group, ``W[row l & 15][32 u + 8 (l >> 4) .. + 8)`` as
    4 bytes (low nibble = the even element).  Scales: int32 ``[N/16, K/128, 16 rows]``, byte ``u``
    of a word = the E8M0 scale of k-step ``u``."""
    n, k = codes.shape
    c = codes.view(n // 16, 16, k // 128, 4, 4, 4, 2)  # tile, row, group, u, q, byte, half
    c = c.permute(0, 2, 4, 1, 3, 5, 6)                # tile, group, q, row, u, byte, half
    packed = (c[..., 0] | (c[..., 1] << 4)).contiguous().view(n // 16, k // 128, 64, 16)
    sc = scales.view(n // 16, 16, k // 128, 4).permute(0, 2, 1, 3).contiguous().to(torch.int32)  # tile, group, row, u
    words = sc[..., 0] | (sc[..., 1] << 8) | (sc[..., 2] << 16) | (sc[..., 3] << 24)
    return packed, words.contiguous()


# Wide steps (more than 128 token rows) with MXFP4 weights: the matrix is expanded, exactly, into a
# bf16 scratch buffer in the packed fragment order and the bf16 tiled GEMM runs on that — one
# streaming pass (0.5 bytes read, 2 written per weight) instead of one 128-row pass of the
# streaming kernel per 128 token rows (2048-token prompt, Llama-3-8B: 180 ms that way).  One buffer
# per device, sized for the largest matrix at packing time (nothing is allocated during a step);
# matrices above the cap (an LM head: wide steps need logits for a few rows only) keep the passes.
_W4_SCRATCH_MAX_ELEMS = 256 * 1024 * 1024  # 512 MiB of bf16
_w4_scratch: Dict[torch.device, torch.Tensor] = {}
_w4_scratch_outgrown: List[torch.Tensor] = []


def _reserve_w4_scratch(device: torch.device, elems: int) -> None:
    if elems > _W4_SCRATCH_MAX_ELEMS:
        return
    cur = _w4_scratch.get(device)
    if cur is None or cur.numel() < elems:
        if cur is not None:
            # a graph captured earlier (another engine of this process, before a draft model was
            # packed) expands into and reads from the old buffer on every replay: it stays allocated
            _w4_scratch_outgrown.append(cur)
        _w4_scratch[device] = torch.empty(elems, dtype=torch.bfloat16, device=device)


class _ExpandedView:
    """What the tiled GEMM needs to know of a weight: a PackedLinear's MXFP4 matrix expanded to bf16."""

    __slots__ = ("wp", "n", "k", "n_pad", "wscale", "bits")

    def __init__(self, w: "PackedLinear"):
        buf = _w4_scratch.get(w.wp.device)
        elems = w.n_pad * w.k
        if buf is None or buf.numel() < elems:
            raise RuntimeError("no MXFP4 expansion buffer was reserved for this matrix")
        native().unpack_mxfp4(w.wp, w.wscale, buf, w.n_pad, w.k)
        self.wp, self.n, self.k, self.n_pad, self.wscale, self.bits = buf[:elems], w.n, w.k, w.n_pad, None, 16


Weight = Union[torch.Tensor, PackedLinear]


class Normed:
    """Residual stream ``x`` (bf16 ``[T, H]``, NOT normalised) plus the per-row partial
    sums of squares ``ss`` (``[T, ceil(H/512)]``) from which consumers derive 1/rms."""

    __slots__ = ("x", "ss", "eps")

    def __init__(self, x: torch.Tensor, ss: torch.Tensor, eps: float):
        self.x, self.ss, self.eps = x, ss, eps


class Partial:
    """Split-K partial sums ``[S, T, N]`` (fp32) of a projection; ``ss``/``eps``/``h_in``
    are set when the projection's input was a :class:`Normed` whose row scale has
    not been applied yet."""

    __slots__ = ("slabs", "ss", "eps", "h_in")

    def __init__(self, slabs: torch.Tensor, ss: Optional[torch.Tensor] = None, eps: float = 0.0, h_in: int = 0):
        self.slabs, self.ss, self.eps, self.h_in = slabs, ss, eps, h_in

    def reduce(self) -> torch.Tensor:
        """Materialise the bf16 result (tests / TP all-reduce path)."""
        y = self.slabs.sum(0)
        if self.ss is not None:
            y = y * torch.rsqrt(self.ss.sum(-1, keepdim=True) / self.h_in + self.eps)
        return y.to(torch.bfloat16)


def add_bias(y: Union[torch.Tensor, Partial], b: torch.Tensor) -> Union[torch.Tensor, Partial]:
    """A projection's bias ``[N]`` on its output, in place.  Split-K partial sums carry it in their first slab —
    divided by the row scale their consumer still applies (the folded RMSNorm's 1 / rms), so that
    ``rs * (sum of slabs) = rs * (x W^T) + b`` and the kernels that sum the slabs need not know about it."""
    if isinstance(y, Partial):
        bf = b if b.dtype == torch.float32 else b.float()  # (prepare_weights keeps it in fp32)
        native().slab_add_bias(y.slabs, bf, y.ss, int(y.h_in) if y.ss is not None else 0, float(y.eps))
        return y
    y[:, : b.numel()].add_(b.to(y.dtype))
    return y


def prepare_weights(weights: Any, quantization: Optional[str] = None) -> None:
    """Fold every RMSNorm gain into the projection that consumes it and pack every
    linear weight of a ``LlamaWeights`` into MFMA fragment order, in place.
    ``quantization='fp8'`` stores the layer projections as e4m3 (the LM head stays
    bf16, as serving stacks do, to keep the logits exact)."""
    if getattr(weights, "_hip_prepared", False):
        return
    quantization = FOUR_BIT_ALIASES.get(quantization or "none", quantization)
    if quantization not in (None, "none", "fp8", "mxfp4"):
        raise ValueError(f"unsupported quantization {quantization!r} (native engine: fp8, mxfp4; int4 / awq / gptq checkpoints' 4-bit slot is served as mxfp4)")
    fp8, fp4 = quantization == "fp8", quantization == "mxfp4"

    def fold_pack(w: torch.Tensor, gain: Optional[torch.Tensor], q: bool = True) -> PackedLinear:
        if gain is not None:
            w = (w.float() * gain.float()[None, :]).to(torch.bfloat16)
        return PackedLinear(w, fp8=q and fp8 and w.shape[1] % 64 == 0, mxfp4=q and fp4 and w.shape[1] % 256 == 0)

    for lw in weights.layers:
        lw.wqkv = fold_pack(lw.wqkv, lw.attn_norm)
        if getattr(lw, "bqkv", None) is not None:
            lw.bqkv = lw.bqkv.float().contiguous()  # added to fp32 slabs (add_bias)
        lw.wgu = fold_pack(lw.wgu, lw.mlp_norm)
        lw.wo = fold_pack(lw.wo, None)
        lw.wdown = fold_pack(lw.wdown, None)
        lw.attn_norm = torch.ones_like(lw.attn_norm)
        lw.mlp_norm = torch.ones_like(lw.mlp_norm)
    weights.lm_head = fold_pack(weights.lm_head, weights.final_norm, False)  # a tied embedding stays untouched: fold_pack copies
    weights.final_norm = torch.ones_like(weights.final_norm)
    weights._hip_prepared = True
    weights.quantization = "fp8" if fp8 else "mxfp4" if fp4 else None
    torch.cuda.empty_cache()


# the reference's 4-bit profiles (runners/profiles/quantization/{int4,autoawq,gptq}.yaml:17) name checkpoint
# formats; with random-init or bf16 weights quantised at load there is one 4-bit weight format here
FOUR_BIT_ALIASES = {"int4": "mxfp4", "awq": "mxfp4", "autoawq": "mxfp4", "gptq": "mxfp4", "fp4": "mxfp4"}

@functools.lru_cache(maxsize=1)
def _num_cus() -> int:
    """Compute units of the device the process runs on: 256 on a whole MI355X, 128 / 64 / 32 inside a
    DPX / QPX / CPX partition (the GEMM planners count rounds of workgroups over them)."""
    try:
        if torch.cuda.is_available():
            return max(1, int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count))
    except Exception:  # noqa: BLE001 - a planner must not fail over a device query
        pass
    return 256




@functools.lru_cache(maxsize=None)
def _waves_per_workgroup(n_pad: int) -> int:
    """Waves (16-feature tiles) per workgroup of an unsplit streaming GEMM.  The kernel's time follows the CU that
    streams the most: rounds of workgroups over the CUs x (tiles per workgroup + the x chunk every workgroup fetches,
    worth about two tiles at 32 rows).  28672 features = 1792 tiles: seven waves are exactly one workgroup per CU
    (cost 9) where four are 448 workgroups, two rounds (12).  The last workgroup may be ragged, so every count serves
    every width: 22016 features (Llama-2-7B gate/up, 1376 tiles) take six waves — 230 workgroups, one round (8) —
    where only four or eight divide the tiles (two rounds of four, or 172 workgroups of eight: 12 / 10)."""
    tiles = -(-n_pad // 16)
    best, best_cost = 4, float("inf")
    for nw in (4, 6, 7, 8):
        wgs = -(-tiles // nw)
        cost = -(-wgs // _num_cus()) * (nw + 2.0)
        if cost < best_cost - 1e-9 or (abs(cost - best_cost) <= 1e-9 and nw > best):
            best, best_cost = nw, cost
    return best


@functools.lru_cache(maxsize=None)
def _plan_slabs(n_pad: int, k: int, m: int = 32, group_cols: int = 256) -> Tuple[int, int, int]:
    """(waves per workgroup, K splits, 2-chunk groups per split) of a split-K streaming GEMM,
    by a three-term cost model fitted to the measured shapes (32 token rows):
    the weight stream costs 1 / (share of the CU rounds the grid fills) — the kernel's time
    follows the busiest CU, so feature tiles x splits should be a whole number of rounds over
    the CUs, splits as equal as K allows; every slab is written and read back once (2 m splits / K of the
    weight bytes for m token rows, at half price: it stays on chip); x is fetched once per
    workgroup (m / (32 waves-per-workgroup), likewise)."""
    ngroups = k // group_cols  # a group = the columns of K one pass of the kernel's register ring covers (2 chunks of 128 by default)
    best: Optional[Tuple[float, int, int, int]] = None
    for nw in (4, 6, 7, 8):
        if n_pad % (16 * nw):
            continue
        tiles = n_pad // (16 * nw)
        seen = set()
        for want in range(1, min(MAX_SPLITS, ngroups) + 1):
            gps = -(-ngroups // want)
            splits = -(-ngroups // gps)  # the last split may be shorter (K = 11008: 43 groups)
            if (splits, gps) in seen:
                continue
            seen.add((splits, gps))
            wgs = tiles * splits
            fill = wgs / (-(-wgs // _num_cus()) * _num_cus())
            even = ngroups / (splits * gps)  # the longest split sets the time
            cost = 1.0 / (fill * even) + 2.0 * m * splits / k + m / (32.0 * nw)
            if best is None or cost < best[0] - 1e-9:
                best = (cost, nw, splits, gps)
    if best is None:
        return 4, 1, ngroups
    return best[1], best[2], best[3]


# Tiled (wide) GEMM: (token tile, feature tile, pipeline variant) -> measured microseconds per
# 64-wide K step of one workgroup with every CU busy (Llama-3-8B shapes, 1 x MI355X).  Per FLOP
# 256 x 256 and 256 x 128 are alike (~1.28 PFLOP/s with every CU busy); the smaller tiles trade
# efficiency for more workgroups; 256 x 224 and 256 x 192 exist for feature counts whose
# 256-wide tiles leave a ragged last round over the 256 CUs (28672 = 128 x 224: 1024 tiles at
# 2048 rows instead of 896; 6144 = 32 x 192: 256 tiles instead of 192).
_WIDE_TILES_BF16 = (((256, 256, 1), 1.68), ((256, 224, 1), 1.615), ((256, 192, 0), 1.36), ((256, 128, 0), 0.84), ((128, 256, 0), 0.87), ((128, 128, 0), 0.62), ((256, 64, 0), 0.65))
_WIDE_TILES_FP8 = (((256, 256, 0), 1.9), ((256, 224, 0), 1.85), ((256, 192, 0), 1.55), ((256, 128, 0), 0.84), ((128, 256, 0), 0.87), ((128, 128, 0), 0.62))
# The 128-wide tiles move 1.5 x the operand bytes per MAC of the 256-wide ones from L2 into LDS; on
# grids of many rounds that traffic shows: + this many microseconds per step and round
# (28672 x 4096 at 2048 rows: 1.01 us per step over 7 rounds, 0.86 over 1-2 rounds).
# FP8 weights x FP8 activations (scaled MFMA): byte for byte the bf16 kernel's data movement on K / 2
# columns, 64-wide stages only; the step times are per stage of 128 K-elements.
# Measured (Llama-3-8B shapes): 256 x 128 runs the gate/up projection at 2048 rows in 258 us (1.87 PFLOP/s; bf16: 413 us)
# and the down projection in 102 us (2.35 PFLOP/s; bf16: 200 us).  The 256 x 256 tile is not offered: its accumulators
# plus 32-byte operands do not fit the register file (630 us on the down projection).
# Third number: microseconds per round outside the K loop (ring prologue, epilogue), fitted on pairs of shapes with
# K = 4096 and K = 14336 (e.g. 128 x 128 at 256 rows: o projection 24.4 us over 32 stages, down 67.5 us over 112).
_WIDE_TILES_FP8A = (((256, 128, 0), 0.92, 7.4), ((128, 256, 0), 0.92, 8.8), ((128, 128, 0), 0.54, 7.2))
# Wide steps with FP8 weights quantise the activations per token to e4m3 as well and multiply on the
# block-scaled matrix instruction (twice the bf16 MFMA rate); "0" keeps bf16 activations (the weights
# are then widened to bf16 in registers).
FP8_ACTIVATIONS = os.environ.get("KVMINI_FP8_ACT", "1") == "1"
# Quantised weights (FP8, MXFP4) with FP8 activations: rows from which a step's projections run on the tiled
# scaled-MFMA kernel instead of the streaming kernel (which widens the weights to bf16 in registers and, from ~64
# rows on, is bound by the x operand rather than by the weight bytes: the quantised formats gain nothing there).
# Measured on MI355X, Llama-3-8B decode step at batch 64 / 96 / 128, streaming -> tiled: FP8 3.30 -> 3.90, 4.05 ->
# 4.28, 4.60 -> 4.57 ms (stays on the streaming kernel up to 128 rows); MXFP4 3.37 -> 3.62, 4.88 (128-row pass) ->
# 3.98, 4.88 -> 4.29 ms (tiled from 65 rows on).
QUANT_WIDE_FROM = int(os.environ.get("KVMINI_QUANT_WIDE_FROM", "129"))
QUANT_WIDE_FROM_FP4 = int(os.environ.get("KVMINI_QUANT_WIDE_FROM_FP4", "65"))
TILED_GATE_UP_128 = os.environ.get("KVMINI_TILED_GATE_UP_128", "0") == "1"
_WIDE_ROUND_SLOPE = {(256, 128): 0.025, (128, 256): 0.025}
_WIDE_STAGE_KSTEPS = {1: 1, 3: 1}  # pipeline variants with 32-wide stages (kv_wide_gemm_stage_ksteps)


@functools.lru_cache(maxsize=None)
def _plan_wide(m: int, n_pad: int, k: int, fp8: int, may_split: bool) -> Tuple[int, int, int, int, int]:
    """(token tile, feature tile, variant, K splits, pipeline stages per split) of the tiled GEMM
    for a shape: the candidate with the lowest modelled time = rounds over the 256 CUs x K steps
    x measured step time of the tile, plus, for split K, the slab bytes written and read back by
    the consumer (2 x splits x m x n x 4 bytes at ~2.5 TB/s: slabs stay in the Infinity Cache)."""
    best: Optional[Tuple[float, Tuple[int, int, int, int, int]]] = None
    # fp8: 0 = bf16 weights, 1 = FP8 weights with bf16 activations, 2 = FP8 weights and activations, 3 = MXFP4 weights and FP8
    # activations (2, 3: k counts PAIRS of elements, i.e. bytes of activations / 2)
    for entry in (_WIDE_TILES_BF16, _WIDE_TILES_FP8, _WIDE_TILES_FP8A, _WIDE_TILES_FP8A)[int(fp8)]:
        (bm, bn, variant), step_us, round_us = entry[0], entry[1], (entry[2] if len(entry) > 2 else 0.0)
        if n_pad % bn:
            continue
        tiles = -(-m // bm) * (n_pad // bn)
        per_stage = 32 * _WIDE_STAGE_KSTEPS.get(variant, 2)
        nst = k // per_stage
        for splits in ((1, 2, 4, 8) if may_split else (1,)):
            if splits > 1 and (tiles * splits > 512 or nst < 4 * splits):
                continue
            sps = -(-nst // splits)
            if (splits - 1) * sps >= nst:
                continue
            rounds = -(-tiles * splits // _num_cus())
            slope = _WIDE_ROUND_SLOPE.get((bm, bn), 0.0) if int(fp8) < 2 else (0.004 if (bm, bn) == (128, 128) else 0.0)
            t = rounds * ((sps * per_stage / 64.0) * (step_us + slope * rounds) + round_us) + (2.0 * splits * m * n_pad * 4 / 2.5e6 if splits > 1 else 0.0)
            # token rows of the last tile that are padding still cost MFMAs
            if best is None or t < best[0] - 1e-9:
                best = (t, (bm, bn, variant, splits, sps))
    if best is None:
        raise ValueError(f"no tiled-GEMM configuration for N={n_pad}, K={k}")
    return best[1]


def _wide(w: Union[PackedLinear, "_ExpandedView"], h: Union[torch.Tensor, "Normed"], epi: int, allow_slabs: bool) -> Union[torch.Tensor, "Partial"]:
    """More than 128 token rows: the tiled MFMA kernel on the packed weights."""
    C = native()
    ss, eps = (h.ss, h.eps) if isinstance(h, Normed) else (None, 0.0)
    x = h.x if isinstance(h, Normed) else h
    if x.stride(1) != 1 or x.stride(0) % 8:
        x = x.contiguous()
    m = x.shape[0]
    may_split = allow_slabs and epi == 0 and w.n == w.n_pad
    if FP8_ACTIVATIONS and isinstance(w, PackedLinear) and w.bits == 8 and w.k % 128 == 0:
        # FP8 x FP8: the activations get one dynamic scale per token (1/rms of a folded norm stays in the epilogue)
        x8, a_scale = C.quantize_rows_fp8(x)
        bm, bn, variant, splits, sps = _plan_wide(m, w.n_pad, w.k // 2, 2, may_split)
        if splits > 1:
            out = torch.empty((splits, m, w.n_pad), dtype=torch.float32, device=x.device)
            C.wide_gemm_fp8a(w.wp, w.n_pad, w.k, x8, a_scale, out, 1, bm, bn, variant, splits, sps, None, 0.0, w.wscale, True)
            return Partial(out, ss, eps, w.k)
        n_out = w.n_pad // 2 if epi == 2 else w.n_pad
        out = torch.empty((m, n_out), dtype=torch.bfloat16, device=x.device)
        C.wide_gemm_fp8a(w.wp, w.n_pad, w.k, x8, a_scale, out, epi, bm, bn, variant, 1, sps, ss, eps, w.wscale, True)
        n_true = w.n // 2 if epi == 2 else w.n
        return out if n_true == n_out else out[:, :n_true]
    bm, bn, variant, splits, sps = _plan_wide(m, w.n_pad, w.k, int(w.wscale is not None), may_split)
    if splits > 1:
        out = torch.empty((splits, m, w.n_pad), dtype=torch.float32, device=x.device)
        C.wide_gemm(w.wp, w.n_pad, w.k, x, out, 1, bm, bn, variant, splits, sps, None, 0.0, w.wscale, True)
        return Partial(out, ss, eps, w.k)
    n_out = w.n_pad // 2 if epi == 2 else w.n_pad
    out = torch.empty((m, n_out), dtype=torch.bfloat16, device=x.device)
    C.wide_gemm(w.wp, w.n_pad, w.k, x, out, epi, bm, bn, variant, 1, sps, ss, eps, w.wscale, True)
    n_true = w.n // 2 if epi == 2 else w.n
    return out if n_true == n_out else out[:, :n_true]


_warned: set = set()


def _warn_once(msg: str) -> None:
    if msg not in _warned:
        _warned.add(msg)
        import warnings

        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _pad_rows(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    m = x.shape[0]
    mp = (m + 15) // 16 * 16
    if mp == m:
        return x, m
    xp = torch.zeros((mp, x.shape[1]), dtype=x.dtype, device=x.device)
    xp[:m] = x
    return xp, m


def _stream(w: PackedLinear, h: Union[torch.Tensor, Normed], epi: int, allow_slabs: bool) -> Union[torch.Tensor, Partial]:
    rows = (h.x if isinstance(h, Normed) else h).shape[0]
    if w.bits == 4:
        return _stream_fp4(w, h, epi, allow_slabs)
    if rows > STREAM_MAX_M and w.k % 64 == 0:
        return _wide(w, h, epi, allow_slabs)
    if rows >= QUANT_WIDE_FROM and w.bits == 8 and FP8_ACTIVATIONS and w.k % 128 == 0:
        return _wide(w, h, epi, allow_slabs)  # FP8 weights: the scaled-MFMA tiled kernel from this many rows on (see QUANT_WIDE_FROM)
    if 96 < rows <= STREAM_MAX_M and epi == 2 and w.bits == 16 and w.k % 64 == 0 and w.n_pad % 128 == 0 and TILED_GATE_UP_128:
        # the gate/up projection at 97-128 rows on the tiled kernel's 128 x 128 tile (both operands by LDS-DMA).  OFF by
        # default: measured on MI355X it wins alone (28672 x 4096 with rotated weights: 53.9 against 61.6 us for the
        # streaming kernel's eight row tiles per wave) and loses inside the decode step at batch 128 (5.13 against 4.94 ms)
        return _wide(w, h, epi, False)
    C = native()
    ss, eps = (h.ss, h.eps) if isinstance(h, Normed) else (None, 0.0)
    x, m = _pad_rows(h.x if isinstance(h, Normed) else h)
    mp = x.shape[0]
    if ss is not None and mp != m:
        ss = torch.cat((ss, torch.ones((mp - m, ss.shape[1]), dtype=ss.dtype, device=ss.device)), 0)
    n_out = w.n_pad // 2 if epi == 2 else w.n_pad
    ks = w.ks if mp <= 16 else 4
    nchunks = (w.k // 32) // ks
    if mp <= MAX_M and w.k % 256 == 0:
        # deep-pipeline kernel (v3).  Measured on MI355X (scripts/gpu_microbench.py --what nw):
        # split-K shapes run best as 8-wave workgroups (x is fetched once per 128 features), 2-chunk
        # groups, ~256 workgroups; unsplit ones (gate/up, LM head) with the waves-per-workgroup
        # count that fills the 256 CUs most evenly, 4-chunk groups up to 32 rows.
        if allow_slabs and epi == 0 and m == mp and w.n == w.n_pad:
            nw, splits, gps = _plan_slabs(w.n_pad, w.k)
            if splits > 1:
                out = torch.empty((splits, mp, w.n_pad), dtype=torch.float32, device=x.device)
                C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, 1, 2, splits, gps, None, 0.0, w.wscale, nw)
                return Partial(out, ss, eps, w.k)
        nb = 4 if (mp <= 32 and w.k % 512 == 0) else 2
        out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
        C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, epi, nb, 1, w.k // (128 * nb), ss, eps, w.wscale, _waves_per_workgroup(w.n_pad))
    elif mp <= MAX_M:
        if allow_slabs and epi == 0:
            splits, cps = C.stream_gemm_plan(w.n_pad, w.k, ks, MAX_SPLITS)
            if splits > 1 and m == mp and w.n == w.n_pad:
                out = torch.empty((splits, mp, w.n_pad), dtype=torch.float32, device=x.device)
                C.stream_gemm(w.wp, w.n_pad, w.k, x, out, 1, ks, splits, cps, None, 0.0, w.wscale)
                return Partial(out, ss, eps, w.k)
        out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
        C.stream_gemm(w.wp, w.n_pad, w.k, x, out, epi, ks, 1, nchunks, ss, eps, w.wscale)
    elif m <= WIDE_M and w.k % 256 == 0:
        # 65..128 tokens (a handful of short prompts next to the running decodes): still one pass
        # over the weights, 8 row tiles per wave.  (The 16-tile / 256-row variant remains for
        # K that is not a multiple of 64, which the tiled kernel does not take.)
        mw = 96 if m <= 96 else 128 if m <= 128 else 256  # (96: six row tiles per wave, for the 96-token bucket)
        if x.shape[0] != mw:
            xw = torch.zeros((mw, x.shape[1]), dtype=x.dtype, device=x.device)
            xw[:m] = x[:m]
            if ss is not None:
                ss = torch.cat((ss[:m], torch.ones((mw - m, ss.shape[1]), dtype=ss.dtype, device=ss.device)), 0)
            x = xw
        if allow_slabs and epi == 0 and m == mw and w.n == w.n_pad:
            nw, splits, gps = _plan_slabs(w.n_pad, w.k, mw)
            if splits > 1:
                out = torch.empty((splits, mw, w.n_pad), dtype=torch.float32, device=x.device)
                C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, 1, 2, splits, gps, None, 0.0, w.wscale, nw)
                return Partial(out, ss, eps, w.k)
        out = torch.empty((mw, n_out), dtype=torch.bfloat16, device=x.device)
        C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, epi, 2, 1, w.k // 256, ss, eps, w.wscale, _waves_per_workgroup(w.n_pad))
    else:
        out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
        for r0 in range(0, mp, MAX_M):
            r1 = min(mp, r0 + MAX_M)
            C.stream_gemm(w.wp, w.n_pad, w.k, x[r0:r1], out[r0:r1], epi, ks, 1, nchunks, None if ss is None else ss[r0:r1].contiguous(), eps, w.wscale)
    n_true = w.n // 2 if epi == 2 else w.n
    if m != mp or n_true != n_out:
        out = out[:m, :n_true]
    return out


@functools.lru_cache(maxsize=None)
def _plan_slabs_w4(n_pad: int, k: int, rows: int) -> Optional[Tuple[int, int, int, int, int]]:
    """(waves per workgroup, ring depth, K splits, groups per split, feature tiles per wave) of a
    split-K streaming GEMM on MXFP4 weights, or None when the shape does not fit the rule.  A 4-bit
    projection of Llama-3-8B is 9-60 MB: the kernel is bound by launch and memory latency, not by
    the weight stream, so — unlike bf16 (:func:`_plan_slabs`) — it wants many small workgroups
    with equal K ranges.  Measured on MI355X (scripts/ft2_gemm_bench.py, microseconds, before -> after):
    32 rows, 8-wave workgroups, 4-chunk rings, the most splits that divide K evenly with at most
    384 workgroups: QKV 8.7 -> 8.0, output 7.9 -> 6.5, down 16.0 -> 12.2; 128 rows, two feature
    tiles per wave (the x fragment reads from LDS bound the one-tile form there), at most one
    workgroup per CU: QKV 18.4 -> 15.9, output 13.7 -> 12.2, down 27.7 -> 23.0."""
    wide = rows > MAX_M
    ft = 2 if wide else 1
    nb = 2 if (wide or rows > 32 or k % 512) else 4
    ngroups = k // (128 * nb)
    if k % (128 * nb) or ngroups < 1:
        return None
    best: Optional[Tuple[int, int, int, int, int]] = None
    for nw in (8, 4):
        if n_pad % (16 * nw * ft):
            continue
        tiles = n_pad // (16 * nw * ft)
        for splits in range(min(ngroups, MAX_SPLITS), 0, -1):
            if ngroups % splits or (wide and ngroups // splits < 2):
                continue
            if tiles * splits <= (_num_cus() if wide else 384):
                cand = (nw, nb, splits, ngroups // splits, ft)
                if best is None or (wide and best[0] == 8 and (n_pad // (16 * 8 * ft)) * best[2] < 192 and tiles * splits > (n_pad // (16 * 8 * ft)) * best[2]):
                    best = cand  # 128 rows: 8-wave workgroups unless they leave a quarter of the CUs idle
                break
        if best is not None and not wide:
            break
    return best if best is not None and best[2] > 1 else None


def _stream_fp4(w: PackedLinear, h: Union[torch.Tensor, Normed], epi: int, allow_slabs: bool) -> Union[torch.Tensor, Partial]:
    """MXFP4 weights: the deep-pipeline streaming kernel for every row count — up to 64 rows as
    the bf16 / FP8 path (split-K slabs where that pays), 65..128 rows in one 8-tile pass; more rows
    run the bf16 tiled GEMM on an exact bf16 expansion of the matrix (``_ExpandedView``), or — for
    a matrix above the expansion buffer's cap — 128-row passes."""
    C = native()
    ss, eps = (h.ss, h.eps) if isinstance(h, Normed) else (None, 0.0)
    x = h.x if isinstance(h, Normed) else h
    if x.stride(1) != 1 or x.stride(0) % 8:
        x = x.contiguous()
    m = x.shape[0]
    n_out = w.n_pad // 2 if epi == 2 else w.n_pad
    n_true = w.n // 2 if epi == 2 else w.n
    if m >= min(129, QUANT_WIDE_FROM_FP4) and FP8_ACTIVATIONS and w.k % 128 == 0 and w.n_pad % 128 == 0:
        # the tiled kernel reads the MXFP4 weights themselves: FP4 operand + hardware block scales on the scaled
        # MFMA against per-token FP8 activations (no expansion pass, no scratch buffer, any matrix size)
        x8, a_scale = C.quantize_rows_fp8(x, False)
        may_split = allow_slabs and epi == 0 and w.n == w.n_pad
        bm, bn, variant, splits, sps = _plan_wide(m, w.n_pad, w.k // 2, 3, may_split)
        if splits > 1:
            out = torch.empty((splits, m, w.n_pad), dtype=torch.float32, device=x.device)
            C.wide_gemm_w4a8(w.wp, w.wscale, w.n_pad, w.k, x8, a_scale, out, 1, bm, bn, variant, splits, sps, None, 0.0, True)
            return Partial(out, ss, eps, w.k)
        out = torch.empty((m, n_out), dtype=torch.bfloat16, device=x.device)
        C.wide_gemm_w4a8(w.wp, w.wscale, w.n_pad, w.k, x8, a_scale, out, epi, bm, bn, variant, 1, sps, ss, eps, True)
        return out if n_true == n_out else out[:, :n_true]
    if m > 128 and w.k % 64 == 0 and w.n_pad * w.k <= _W4_SCRATCH_MAX_ELEMS:
        return _wide(_ExpandedView(w), h, epi, allow_slabs)  # (KVMINI_FP8_ACT=0: bf16 activations on an exact bf16 expansion of the matrix)
    if m > 128:
        out = torch.empty((m, n_true), dtype=torch.bfloat16, device=x.device)
        for r0 in range(0, m, 128):
            r1 = min(m, r0 + 128)
            part = Normed(x[r0:r1], ss[r0:r1].contiguous(), eps) if ss is not None else x[r0:r1]
            out[r0:r1] = _stream_fp4(w, part, epi, False)
        return out
    mp = (m + 15) // 16 * 16 if m <= MAX_M else 128
    cols = 128  # columns of K per chunk of the kernel's register ring
    if mp != m:
        xp = torch.zeros((mp, x.shape[1]), dtype=x.dtype, device=x.device)
        xp[:m] = x
        x = xp
        if ss is not None:
            ss = torch.cat((ss, torch.ones((mp - m, ss.shape[1]), dtype=ss.dtype, device=ss.device)), 0)
    if allow_slabs and epi == 0 and m == mp and w.n == w.n_pad:
        plan = _plan_slabs_w4(w.n_pad, w.k, mp)
        if plan is not None:
            nw, nbs, splits, gps, ft = plan
        else:
            nbs, ft = 4 if (mp <= 32 and w.k % (4 * cols) == 0) else 2, 1
            nw, splits, gps = _plan_slabs(w.n_pad, w.k, mp, nbs * cols)
        if splits > 1:
            out = torch.empty((splits, mp, w.n_pad), dtype=torch.float32, device=x.device)
            C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, 1, nbs, splits, gps, None, 0.0, w.wscale, nw, 4, ft=ft)
            return Partial(out, ss, eps, w.k)
    # Unsplit (gate/up, LM head).  Up to 32 rows a ring of 4 chunks (measured 18.4 us on the gate/up
    # projection against 18.9 for a ring of 8 and 29.6 for 2); at 128 rows two feature tiles per
    # wave in 4-wave workgroups when that fills the CUs at least as well (gate/up 58.1 -> 49.6 us).
    nb = 4 if (mp <= 32 and w.k % (4 * cols) == 0) else 2
    nw, ft = _waves_per_workgroup(w.n_pad), 1
    if mp == 128 and w.n_pad % 128 == 0:
        wgs1, wgs2 = w.n_pad // (16 * nw), w.n_pad // 128
        fill = lambda g: g / (-(-g // _num_cus()) * _num_cus())  # noqa: E731
        if fill(wgs2) >= fill(wgs1) - 1e-9:
            nw, ft = 4, 2
    out = torch.empty((mp, n_out), dtype=torch.bfloat16, device=x.device)
    C.stream_gemm3(w.wp, w.n_pad, w.k, x, out, epi, nb, 1, w.k // (cols * nb), ss, eps, w.wscale, nw, 4, ft=ft)
    if m != mp or n_true != n_out:
        out = out[:m, :n_true]
    return out


# ---- weight prefetch beside the latency-bound kernels of a decode step -------------------------
# KVMINI_PREFETCH="site=KiB:workgroups,..." — KiB = how much of the start of every (16-feature
# tile, K split) range of the NEXT streaming GEMM is read ahead, workgroups = extra workgroups the
# carrying kernel gets for it.  qkv / gu / head ride on the residual kernel in front of the QKV,
# gate/up and LM-head projections (one-wave workgroups), o rides on the fused decode attention
# kernel (8-wave workgroups).  Example that keeps each XCD's share within its 4 MiB L2:
# "qkv=16:1536,o=16:64,gu=12:1792,head=4:1024".
# OFF by default.  Measured on MI355X (Llama-3-8B, batch 20): the GEMMs do get faster — output
# projection 9.2 -> 4.6 us from L2, QKV 10.6 -> 8.6, gate/up 40.6 -> 39.5 — but the carriers pay
# more than that: the attention kernel fills a CU per workgroup, so its prefetch rows only get
# the ~64-96 CUs the step leaves idle (~45 GB/s each: 32 MB take 12 us, 9.1 -> 21.5 us), and
# the residual kernels' idle window inside a replayed graph is shorter than the GEMM time
# saved (+1.7 us each).  Decode step 3.37 -> 3.41 (one residual site) ... 3.64 ms (all sites).
_PREFETCH_DEFAULT = "0"
_prefetch_sinks: Dict[Any, torch.Tensor] = {}


@functools.lru_cache(maxsize=None)
def _prefetch_config(spec: Optional[str] = None) -> Dict[str, Tuple[int, int]]:
    spec = os.environ.get("KVMINI_PREFETCH", _PREFETCH_DEFAULT) if spec is None else spec
    out: Dict[str, Tuple[int, int]] = {}
    if spec.strip() in ("", "0", "off", "none"):
        return out
    for item in spec.split(","):
        site, _, val = item.strip().partition("=")
        kib, _, wgs = val.partition(":")
        if int(kib) > 0 and int(wgs) >= 8:
            out[site] = (int(kib), int(wgs) // 8 * 8)
    return out


def prefetch_plan(w: Weight, rows: int, site: str) -> Optional[Tuple[torch.Tensor, list, torch.Tensor]]:
    """What a latency-bound kernel needs to read ahead for the streaming GEMM that follows it:
    ``(packed weights, [gx, gy, nw, kib, tile_bytes, split_bytes, workgroups], sink)`` — the grid
    :func:`_stream` will launch for ``w`` at ``rows`` token rows (feature groups x K splits, waves
    per workgroup, bytes of a 16-feature tile and of one K split inside it), or None when the
    site is switched off or the step is not one the streaming kernel's main path serves.
    ``site``: 'qkv' / 'o' (split-K slabs), 'gu' / 'head' (unsplit)."""
    cfg = _prefetch_config().get(site)
    if cfg is None or not isinstance(w, PackedLinear) or rows > MAX_M or rows % 16 or w.k % 256:
        return None
    kib, wgs = cfg
    col_bytes = 2 * w.bits  # bytes of one K column of a 16-feature tile
    tile_bytes = w.k * col_bytes
    nw, gy, split_bytes = 0, 1, tile_bytes
    if site in ("qkv", "o") and w.n == w.n_pad:
        p4 = _plan_slabs_w4(w.n_pad, w.k, rows) if w.bits == 4 else None
        if p4 is not None:
            nw, nb, gy, gps, ft = p4
            nw, group_cols = nw * ft, 128 * nb  # ranges (16-feature tiles) per workgroup
        elif w.bits == 4:
            group_cols = 128 * (4 if (rows <= 32 and w.k % 512 == 0) else 2)
            nw, gy, gps = _plan_slabs(w.n_pad, w.k, rows, group_cols)
        else:
            group_cols = 256
            nw, gy, gps = _plan_slabs(w.n_pad, w.k)
        split_bytes = gps * group_cols * col_bytes
    if gy <= 1:
        nw, gy, split_bytes = _waves_per_workgroup(w.n_pad), 1, tile_bytes
    if tile_bytes % 1024 or split_bytes % 1024 or w.n_pad % (16 * nw):  # (the plan mirrors whole workgroups: no ragged last one)
        return None
    dev = w.wp.device
    sink = _prefetch_sinks.get(dev)
    if sink is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        sink = _prefetch_sinks[dev] = torch.zeros(4, dtype=torch.int32, device=dev)
    return w.wp, [w.n_pad // (16 * nw), gy, nw, min(kib, split_bytes // 1024), tile_bytes, split_bytes, wgs], sink


def prefetch_weight(w: Weight, workgroups: int) -> None:
    """Reads the packed weights of ``w`` once on the current stream without using them (a kernel of
    its own, for a side stream that runs beside a latency-bound kernel): the bytes stay in the
    memory-side cache for the GEMM that streams them next."""
    dev = w.wp.device
    sink = _prefetch_sinks.get(dev)
    if sink is None:
        if torch.cuda.is_current_stream_capturing():
            return
        sink = _prefetch_sinks[dev] = torch.zeros(4, dtype=torch.int32, device=dev)
    native().prefetch(w.wp, int(workgroups), sink)


_resid_counters: Dict[Any, torch.Tensor] = {}
FUSE_RESIDUAL = os.environ.get("KVMINI_FUSE_RESIDUAL", "0") == "1"  # measured slower than the two-kernel path (see linear_residual): off unless asked for


# Row-parallel projections (output, down) of steps up to 64 rows: one workgroup per 16-feature tile
# with K split over its 8 waves, residual add and sums of squares in the epilogue — no slabs, no
# residual kernel (csrc/stream_gemm.hip: stream_gemm_rowsum_kernel).  OFF by default: measured on MI355X
# (Llama-3-8B, batch 20) the decode step is 3.39 ms with it on the output projection and 3.78 ms with it
# on both row-parallel projections, against 3.34 ms for the split-K GEMM + residual_reduce pair (the form
# makes every CU read the whole of x).  KVMINI_ROWSUM_GEMM=1 turns it on; KVMINI_ROWSUM_NB pins the ring depth.
ROWSUM_GEMM = os.environ.get("KVMINI_ROWSUM_GEMM", "0") == "1"
ROWSUM_NB = int(os.environ.get("KVMINI_ROWSUM_NB", "0"))
# Every CU reads the whole of x (rows x K) in this form, which is what bounds it: measured on MI355X at 32 rows,
# K = 4096 (Llama-3-8B output projection) 12.4 us against 13.2 for the pair of kernels, K = 14336 (down
# projection) 37 against 24.5; at 16 rows 9.2 / 12.7 and 25.0 / 24.2.  So: rows x K up to this bound.
ROWSUM_MAX_ROWS_K = int(os.environ.get("KVMINI_ROWSUM_MAX_ROWS_K", str(32 * 4096)))


def _linear_residual_rowsum(x: torch.Tensor, w: "PackedLinear", residual: torch.Tensor, eps: float) -> Optional[Tuple[Normed, torch.Tensor]]:
    m = x.shape[0]
    if m * w.k > ROWSUM_MAX_ROWS_K or w.bits == 4 or m > MAX_M or m % 16 or w.n != w.n_pad or w.k % 1024 or residual.shape != (m, w.n) or residual.dtype != torch.bfloat16 or not residual.is_contiguous():
        return None
    if x.stride(1) != 1 or x.stride(0) % 8:
        x = x.contiguous()
    nb = ROWSUM_NB or (3 if m <= 32 else 2)
    ss = native().stream_gemm_rowsum(w.wp, w.n_pad, w.k, x, residual, True, w.wscale, nb)
    return Normed(residual, ss, eps), residual


def linear_residual(x: torch.Tensor, w: Weight, residual: torch.Tensor, eps: float) -> Optional[Tuple[Normed, torch.Tensor]]:
    """``residual += x @ W^T`` plus the sums of squares of the new residual rows, in ONE launch.
    Default form: :func:`_linear_residual_rowsum` (K split inside the workgroup, no slabs).  The
    older form (``KVMINI_FUSE_RESIDUAL=1`` with ``KVMINI_ROWSUM_GEMM=0``): the split-K streaming GEMM
    writes its slabs and the last workgroup of each feature group to finish folds them into
    ``residual`` (bf16, in place).  Either way the separate residual / RMSNorm-statistics kernel
    between a row-parallel projection and the next one disappears.  Returns
    ``(Normed(residual, ss, eps), residual)``, or None when the step is outside what the fused
    epilogues serve (more than 64 rows, rows not a multiple of 16, padded features, 4-bit weights):
    the caller then runs the projection and the residual update separately."""
    if not isinstance(w, PackedLinear) or isinstance(x, Normed) or residual is None:
        return None
    if ROWSUM_GEMM:
        out = _linear_residual_rowsum(x, w, residual, eps)
        if out is not None:
            return out
    if not FUSE_RESIDUAL or w.bits == 4:
        return None
    m = x.shape[0]
    if m > MAX_M or m % 16 or w.n != w.n_pad or w.k % 256 or residual is None or residual.shape != (m, w.n):
        return None
    if x.stride(1) != 1 or x.stride(0) % 8:
        x = x.contiguous()
    nw, splits, gps = _plan_slabs(w.n_pad, w.k, m)
    dev = x.device
    cnt = _resid_counters.get(dev)
    if cnt is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        cnt = _resid_counters[dev] = torch.zeros(4096, dtype=torch.int32, device=dev)
    groups = w.n_pad // (16 * nw)
    if groups > cnt.numel():
        return None
    slabs = torch.empty((splits, m, w.n_pad), dtype=torch.float32, device=dev)
    ss = torch.empty((m, groups), dtype=torch.float32, device=dev)
    native().stream_gemm3(w.wp, w.n_pad, w.k, x, slabs, 3, 2, splits, gps, None, 0.0, w.wscale, nw, 0, residual, True, ss, cnt)
    return Normed(residual, ss, eps), residual


def add_rmsnorm(x: Union[torch.Tensor, Partial], residual: Optional[torch.Tensor], w: torch.Tensor, eps: float,
                prefetch: Optional[Tuple[torch.Tensor, list, torch.Tensor]] = None) -> Tuple[Normed, torch.Tensor]:
    """Residual update of the folded-norm pipeline.  ``x``: bf16 ``[T, H]`` or a
    :class:`Partial` (split-K slabs of the previous projection); ``residual``: bf16
    ``[T, H]`` updated in place, or None for the first layer.  ``w`` (the gain) must
    already be folded into the consuming projection (:func:`prepare_weights`).
    ``prefetch``: a :func:`prefetch_plan` for the projection that runs next."""
    C = native()
    if isinstance(x, Partial):
        assert x.ss is None, "slabs that still need a row scale cannot feed a residual add"
        src = x.slabs
        t, hdim = src.shape[1], src.shape[2]
    else:
        src = x if x.is_contiguous() else x.contiguous()
        t, hdim = src.shape
    has_res = residual is not None
    if residual is None:
        residual = torch.empty((t, hdim), dtype=torch.bfloat16, device=src.device)
    ss = C.residual_reduce(src, residual, has_res, *(prefetch or ()))
    return Normed(residual, ss, eps), residual


def embed_tokens(table: torch.Tensor, tokens: torch.Tensor, feed: torch.Tensor, prev_ids: torch.Tensor) -> torch.Tensor:
    """Embedding lookup with device-side token feed (see ``torch_ops.embed_tokens``): one
    kernel, so a step can be launched before the host has read the previous step's ids."""
    if table.dtype != torch.bfloat16 or table.shape[1] % 8:
        from . import torch_ops

        return torch_ops.embed_tokens(table, tokens, feed, prev_ids)
    out = torch.empty((tokens.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
    native().embed_fed(table, tokens, feed, prev_ids, out)
    return out


def select_rows(h: Normed, idx: torch.Tensor) -> Normed:
    return Normed(torch.index_select(h.x, 0, idx), torch.index_select(h.ss, 0, idx), h.eps)


def linear(x: Union[torch.Tensor, Normed], w: Weight, allow_slabs: bool = False) -> Union[torch.Tensor, Partial]:
    if isinstance(w, PackedLinear):
        return _stream(w, x, 0, allow_slabs)
    assert not isinstance(x, Normed)
    return F.linear(x, w)


def linear_swiglu(x: Union[torch.Tensor, Normed], wgu: Weight) -> torch.Tensor:
    if isinstance(wgu, PackedLinear):
        return _stream(wgu, x, 2, False)  # type: ignore[return-value]
    assert not isinstance(x, Normed)
    gu = F.linear(x, wgu)
    act = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=gu.dtype, device=gu.device)
    native().swiglu_interleaved(gu, act)
    return act


def swiglu_interleaved(gu: torch.Tensor) -> torch.Tensor:
    act = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=gu.dtype, device=gu.device)
    native().swiglu_interleaved(gu, act)
    return act


def rope_kv_write(
    qkv: Union[torch.Tensor, Partial],
    positions: torch.Tensor,
    cos: torch.Tensor,
    sin: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    n_heads: int,
    n_kv: int,
    head_dim: int,
    kv_scale: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """``qkv``: bf16 ``[T, W]`` (q is rotated in place and returned as a strided
    view) or a :class:`Partial` (slabs are summed, row-scaled, and q is written to a
    new ``[T, nh, hd]``).  ``kv_scale`` (fp32 ``[2, nkv]``): the cache holds e4m3 bytes of
    ``x / scale`` (FP8 KV cache); None: a bf16 cache."""
    if isinstance(qkv, Partial):
        t = qkv.slabs.shape[1]
        q = torch.empty((t, n_heads, head_dim), dtype=torch.bfloat16, device=qkv.slabs.device)
        native().rope_kv_write_slabs(qkv.slabs, q, positions, cos, sin, k_cache, v_cache, slot_mapping, n_heads, n_kv, head_dim, qkv.ss, qkv.h_in, qkv.eps, kv_scale)
        return q
    if not qkv.is_contiguous():
        qkv = qkv.contiguous()
    native().rope_kv_write(qkv, positions, cos, sin, k_cache, v_cache, slot_mapping, n_heads, n_kv, head_dim, kv_scale)
    return qkv[:, : n_heads * head_dim].view(qkv.shape[0], n_heads, head_dim)


def paged_attention(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    tok_seq: torch.Tensor,
    ctx_lens: torch.Tensor,
    scale: float,
    max_blocks: Optional[int] = None,
    kv_scale: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    t, nh, hd = q.shape
    if hd not in (64, 128) or k_cache.shape[2] != 16 or nh // k_cache.shape[1] not in GQA_RATIOS:
        # shapes outside the kernel's specialisation (test-sized models): reference path
        from . import torch_ops

        _warn_once(f"paged attention: head_dim {hd}, block {k_cache.shape[2]}, {nh // k_cache.shape[1]} query heads per kv head is outside the HIP kernels' "
                   f"specialisation (head_dim 64 / 128, block 16, ratios {GQA_RATIOS}): using the PyTorch reference path, which is far slower")

        return torch_ops.paged_attention(q, k_cache, v_cache, block_tables, tok_seq, ctx_lens, scale, max_blocks, kv_scale)
    out = torch.empty((t, nh * hd), dtype=q.dtype, device=q.device)
    native().paged_attention(q, k_cache, v_cache, block_tables, tok_seq, ctx_lens, out, scale, kv_scale)
    return out


def supports_fused_decode_attention(qkv: Any, n_heads: int, k_cache: torch.Tensor) -> bool:
    """Shapes the one-launch RoPE + KV write + attention kernel covers."""
    nkv, bs, hd = k_cache.shape[1], k_cache.shape[2], k_cache.shape[3]
    return isinstance(qkv, Partial) and hd == 128 and bs == 16 and n_heads % nkv == 0 and n_heads // nkv in GQA_RATIOS


# Waves per workgroup of the fused decode attention kernel: every wave keeps two 16-token K/V
# blocks in flight, so a long context streams at (workgroups x waves x 16 KB) / latency.  Eight
# waves when the (tokens x kv heads) grid does not fill the 256 CUs on its own (Llama-3-8B at
# batch 20: 160 workgroups); with more workgroups than that, two 4-wave workgroups per CU keep
# as much in flight without the wider merge (Llama-2-7B at batch 10, 320 workgroups: 3.00 vs
# 3.13 ms per step).  KVMINI_DECODE_WAVES=4|8 pins it.
DECODE_ATTENTION_WAVES = int(os.environ.get("KVMINI_DECODE_WAVES", "0"))


def _decode_attention_waves(tokens: int, n_kv_heads: int) -> int:
    if DECODE_ATTENTION_WAVES in (4, 8):
        return DECODE_ATTENTION_WAVES
    return 8 if tokens * n_kv_heads <= _num_cus() else 4


# Split-KV decode attention: with few (token, kv head) pairs in a step the grid cannot fill the
# chip and a long context is streamed by a handful of workgroups (a 128k-token sequence decoded
# alone: 8 workgroups, 0.28 TB/s).  Such steps run several workgroups per pair, each over a
# contiguous part of the context (the kernel uses fewer for short contexts), merged by the last
# one to finish.  The choice depends only on the step's shape, so a captured graph keeps it.
SPLIT_KV_MAX_PAIRS = int(os.environ.get("KVMINI_SPLIT_KV_MAX_PAIRS", "128"))  # 128: the smallest token bucket (16 rows, padding included) of an 8-kv-head model
SPLIT_KV_MAX = 32
_split_ws: Dict[Any, torch.Tensor] = {}


def decode_attention_splits(tokens: int, n_kv_heads: int) -> int:
    """Workgroups per (token, kv head) for a decode step of ``tokens`` rows (padding rows count:
    they are part of the captured grid and exit at once)."""
    forced = int(os.environ.get("KVMINI_DECODE_SPLITS", "0"))
    if forced > 0:
        return forced
    pairs = tokens * n_kv_heads
    if pairs > SPLIT_KV_MAX_PAIRS:
        return 1
    # (16 workgroups per CU's worth of pairs: the rows of a 16-row bucket that are padding, or short, leave at once,
    # and a lone 128k-token sequence in such a step is then streamed by 256 workgroups instead of 128:
    # 4.83 against 5.91 ms per step with the e4m3 cache, 6.09 against 6.50 with bf16)
    return max(1, min(SPLIT_KV_MAX, 16 * _num_cus() // pairs))


_SPLIT_WS_FLOATS = 4 << 20  # 16 MiB: 4096 counters + partials of up to ~4000 (pair, split) combinations of an 8-head group


def _split_workspace(device: torch.device, floats: int) -> torch.Tensor:
    """One workspace per device, allocated once at a fixed size: captured graphs keep its address."""
    if floats > _SPLIT_WS_FLOATS:
        raise ValueError(f"split-KV decode attention needs {floats} workspace floats, more than the fixed {_SPLIT_WS_FLOATS}: use fewer splits")
    ws = _split_ws.get(device)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the split-KV workspace must exist before graph capture (run the step eagerly once)")
        ws = torch.zeros(_SPLIT_WS_FLOATS, dtype=torch.float32, device=device)
        _split_ws[device] = ws
    return ws


def rope_attention_fused(
    qkv: Partial,
    positions: torch.Tensor,
    cos: torch.Tensor,
    sin: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    block_tables: torch.Tensor,
    tok_seq: torch.Tensor,
    ctx_lens: torch.Tensor,
    n_heads: int,
    scale: float,
    kv_scale: Optional[torch.Tensor] = None,
    n_split: Optional[int] = None,
    prefetch: Optional[Tuple[torch.Tensor, list, torch.Tensor]] = None,
) -> torch.Tensor:
    """RoPE + paged-KV write + attention for a step in which every token is the only new
    token of its sequence (plain decoding).  ``qkv`` must be the split-K slabs of the QKV
    projection; returns the attention output ``[T, nh*hd]``.  ``n_split``: workgroups per
    (token, kv head) (None: by the step's shape, see :func:`decode_attention_splits`).
    ``prefetch``: a :func:`prefetch_plan` for the output projection (unsplit launches only)."""
    t = qkv.slabs.shape[1]
    nkv, hd = k_cache.shape[1], k_cache.shape[3]
    out = torch.empty((t, n_heads * hd), dtype=torch.bfloat16, device=qkv.slabs.device)
    ns = decode_attention_splits(t, nkv) if n_split is None else n_split
    ws = _split_workspace(out.device, 4096 + t * n_heads * ns * (hd + 2)) if ns > 1 else None
    _apply_attention_mfma()
    native().decode_attention_fused(qkv.slabs, positions, cos, sin, k_cache, v_cache, slot_mapping, block_tables, tok_seq, ctx_lens, out, n_heads, qkv.ss, qkv.h_in, qkv.eps, scale,
                                    _decode_attention_waves(t, nkv), kv_scale, ns, ws, *((prefetch if ns == 1 else None) or ()))
    return out


# Long-context (split-KV) decode attention of head groups of 4 and 8: scores from the matrix cores
# (mfma_f32_16x16x32_bf16 on K blocks as they are loaded) instead of packed dot products; "0" keeps the
# dot-product form.  Applied once, when the extension is first used for attention.
ATTENTION_MFMA = os.environ.get("KVMINI_ATTN_MFMA", "1") == "1"
_attention_mfma_applied: Optional[bool] = None


def _apply_attention_mfma() -> None:
    global _attention_mfma_applied
    if _attention_mfma_applied != ATTENTION_MFMA:
        native().set_attention_mfma(ATTENTION_MFMA)
        _attention_mfma_applied = ATTENTION_MFMA


def split_decode_attention(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    tok_seq: torch.Tensor,
    ctx_lens: torch.Tensor,
    rows: torch.Tensor,
    out: torch.Tensor,
    scale: float,
    kv_scale: Optional[torch.Tensor] = None,
    n_split: Optional[int] = None,
) -> torch.Tensor:
    """Split-KV attention for the decode tokens of a step that also carries prompt chunks:
    ``rows`` (int32, -1 = unused entry) lists the token rows to serve, ``q`` ``[T, nh, hd]`` is
    rotated and the step's keys / values are in the cache (:func:`rope_kv_write` ran).  Writes
    the listed rows of ``out`` ``[T, nh*hd]``; the context of a listed row is spread over up to
    ``n_split`` workgroups per kv head (None: by the number of listed rows), where
    :func:`paged_attention` would walk it with one."""
    n_rows = rows.numel()
    nh, hd = q.shape[1], q.shape[2]
    nkv = k_cache.shape[1]
    ns = decode_attention_splits(n_rows, nkv) if n_split is None else n_split
    ws = _split_workspace(out.device, 4096 + n_rows * nh * ns * (hd + 2))
    _apply_attention_mfma()
    native().split_decode_attention(q, k_cache, v_cache, block_tables, tok_seq, ctx_lens, rows, out, scale, _decode_attention_waves(n_rows, nkv), kv_scale, ns, ws)
    return out


# ---- all decoder layers of a plain decoding step in one persistent launch ------------------------
# OFF by default (KVMINI_DECODE_LAYERS=1 turns it on; KVMINI_DECODE_LAYERS_GRID pins the workgroup count).
# Measured on MI355X, Llama-3-8B, batch 20 (profiles/bench/decode_layers_persistent_r3.json): bit-equal to
# the kernel-per-phase path, 4.48 ms per step against 3.32 ms.  Workgroup 0's timeline per layer: barriers
# 3.3-6.9 us each in situ (a kernel boundary costs ~1.8 us), gate/up 46.8 us against 38.2 (every workgroup
# reads x from the uncached exchange buffer), attention 13.4 against 9.1; the weight chunks issued before a
# barrier do not shorten the GEMM phases.  What it would take to win is a loader that keeps HBM busy
# independently of the phases (weights into an LDS ring by dedicated waves), not cheaper barriers.
DECODE_LAYERS = os.environ.get("KVMINI_DECODE_LAYERS", "0") == "1"


class DecodeLayers:
    """State of the persistent decode-layers kernel for one model on one device
    (``csrc/engine_kernels.hip``: ``decode_layers_kernel``): the per-layer pointer table, the
    GEMM plans (the streaming GEMM's own: same waves, same K splits, hence the same numbers as
    the kernel-per-phase path) and the uncached workspace the phases exchange their results
    through.  Built once, before any graph capture; :meth:`run` is capturable."""

    def __init__(self, weights: Any, kv_cache: torch.Tensor, kv_scale: Optional[torch.Tensor], n_heads: int, n_kv: int):
        C = native()
        cfg = weights.cfg
        layers = weights.layers
        self.nh, self.nkv, self.inter = n_heads, n_kv, layers[0].wdown.k
        self.hidden = cfg.hidden_size
        plans = []
        for w, split in ((layers[0].wqkv, True), (layers[0].wo, True), (layers[0].wgu, False), (layers[0].wdown, True)):
            if split:
                nw, splits, gps = _plan_slabs(w.n_pad, w.k)
                cps = 2 * gps
            else:
                nw, splits, cps = _waves_per_workgroup(w.n_pad), 1, w.k // 128
            plans += [w.n_pad, w.k, nw, splits, cps]
        self.plans = plans
        self.s_qkv, self.s_proj = plans[3], max(plans[8], plans[18])
        dev = kv_cache.device
        rows = []
        for li, lw in enumerate(layers):
            rows.append([lw.wqkv.wp.data_ptr(), lw.wo.wp.data_ptr(), lw.wgu.wp.data_ptr(), lw.wdown.wp.data_ptr(), kv_cache[li, 0].data_ptr(), kv_cache[li, 1].data_ptr()])
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.kv_scale = kv_scale
        self.ws = C.alloc_uncached(C.decode_layers_workspace_bytes(self.hidden, n_heads, n_kv, self.inter, self.s_qkv, self.s_proj))
        self.state_off = C.decode_layers_state_offset(self.hidden, n_heads, n_kv, self.inter, self.s_qkv, self.s_proj)
        self.grid = int(os.environ.get("KVMINI_DECODE_LAYERS_GRID", "0")) or min(_num_cus(), 512)
        self.keep = (weights, kv_cache)  # the table holds raw pointers into these
        self.timeline = os.environ.get("KVMINI_DECODE_LAYERS_TIMELINE", "0") == "1"

    @staticmethod
    def supported(weights: Any, kv_cache: torch.Tensor, n_heads: int, n_kv: int) -> bool:
        """bf16 packed weights, head_dim 128, KV blocks of 16, head groups of 4, shapes the streaming plans divide."""
        if not DECODE_LAYERS or not gpu_available():
            return False
        lw = weights.layers[0]
        ws = (lw.wqkv, lw.wo, lw.wgu, lw.wdown)
        if getattr(lw, "bqkv", None) is not None:  # (no bias in the persistent kernel's QKV phase)
            return False
        if not all(isinstance(w, PackedLinear) and w.bits == 16 and w.n == w.n_pad and w.k % 256 == 0 for w in ws):
            return False
        if lw.wgu.n_pad % (16 * _waves_per_workgroup(lw.wgu.n_pad)):  # (the persistent kernel's gate/up phase has no ragged last workgroup)
            return False
        if kv_cache.shape[-1] != 128 or kv_cache.shape[-2] != 16 or n_kv < 1 or n_heads % n_kv or n_heads // n_kv != 4:
            return False
        return weights.cfg.hidden_size % 128 == 0 and lw.wgu.n_pad == 2 * lw.wdown.k

    def run(self, residual: torch.Tensor, ss: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, slot_mapping: torch.Tensor, block_tables: torch.Tensor,
            tok_seq: torch.Tensor, ctx_lens: torch.Tensor, eps: float, scale: float) -> None:
        """``residual`` (bf16 ``[T, H]``, T in 16 / 32 / 48 / 64) and ``ss`` (its per-row sums of squares, ``[T, H/512]``)
        enter as the first layer's input and leave as the last layer's output."""
        native().decode_layers(self.table, self.plans, self.nh, self.nkv, self.inter, positions, cos, sin, slot_mapping, block_tables, tok_seq, ctx_lens, eps, scale,
                               self.kv_scale, residual, ss, self.ws, self.grid, self.timeline)

    def read_timeline(self, n_layers: int) -> "torch.Tensor":
        """Workgroup 0's phase stamps of the last launch (``timeline = True``): int64 ``[layers, 16]`` ticks of the 100 MHz real-time counter."""
        off = native().decode_layers_layout(self.hidden, self.nh, self.nkv, self.inter, self.s_qkv, self.s_proj)[9]
        return self.ws[off : off + n_layers * 16 * 8].view(torch.int64).view(n_layers, 16).cpu()

    def error_word(self) -> int:
        """Non-zero after a launch in which a barrier timed out (a workgroup never arrived): results of that launch are garbage."""
        return int(self.ws[self.state_off + 4 : self.state_off + 8].view(torch.int32).item())


def supports_prefill_attention(n_heads: int, k_cache: torch.Tensor) -> bool:
    nkv, bs, hd = k_cache.shape[1], k_cache.shape[2], k_cache.shape[3]
    return hd == 128 and bs == 16 and n_heads % nkv == 0 and n_heads // nkv in GQA_RATIOS


def prefill_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor, tiles: torch.Tensor, out: torch.Tensor, scale: float,
                      kv_scale: Optional[torch.Tensor] = None, quads: bool = False) -> torch.Tensor:
    """Tiled MFMA causal attention for prompt chunks, written into ``out`` ``[T, nh*hd]``
    in place (rows not covered by a tile keep their content).  ``tiles``: int32
    ``[4, n_tiles]`` = first token row, valid queries (0 = unused), first position,
    block-table row of each 16-query tile; K/V of the chunk must be in the cache already.
    ``quads``: the list is padded per chunk to a multiple of four tiles (``pad_tiles_to_quads``),
    so every aligned group of four is consecutive tiles of one chunk: the 8-wave kernel then
    serves a group (up to 64 queries) per workgroup (head groups of at most 4)."""
    native().prefill_attention(q, k_cache, v_cache, block_tables, tiles, out, scale, kv_scale, quads)
    return out


PREFILL_SPLIT_MAX = 32  # key-range shares per tile, at most


def prefill_split_plan(n_tiles: int, nh: int, nkv: int, max_ctx: int, min_keys: int = 1024) -> int:
    """Key-range shares for a step's ``n_tiles`` prompt tiles whose queries see up to ``max_ctx`` keys (0: the unsplit
    kernel serves them): a power of two that brings the launch to about two workgroups per CU (all resident at once),
    each share at least ``min_keys`` keys."""
    groups = ((n_tiles + 1) // 2 if nh // nkv <= 4 else n_tiles) * nkv
    if groups <= 0:
        return 0
    ns = min(PREFILL_SPLIT_MAX, 2 * _num_cus() // groups, max_ctx // max(32, min_keys))
    if ns < 2:
        return 0
    return 1 << (ns.bit_length() - 1)


def prefill_attention_split(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor, tiles: torch.Tensor, n_tiles: int, out: torch.Tensor,
                            scale: float, n_split: int, kv_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """:func:`prefill_attention` for the first ``n_tiles`` tiles of ``tiles`` (``[4, max_tiles]``) with every tile's
    key range walked by ``n_split`` workgroups (partials in an fp32 workspace, folded by a second launch): a few tiles
    deep inside a long context — a short new turn after a cached prefix of 100k tokens — otherwise walk the whole
    context with one workgroup per tile and kv head."""
    nh = q.shape[1]
    rows = n_split * 16 * n_tiles
    ws = torch.empty(rows * nh * 130, dtype=torch.float32, device=q.device)  # [rows][nh][128] outputs, then [rows][nh][2] (reference, sum)
    native().prefill_attention_split(q, k_cache, v_cache, block_tables, tiles, n_tiles, out, scale, kv_scale, n_split, ws[: rows * nh * 128], ws[rows * nh * 128 :])
    return out


def pad_tiles_to_quads(tiles: Sequence[Tuple[int, int, int, int]]) -> list:
    """``tiles`` (first row, valid queries, first position, block-table row; a chunk's tiles in
    order) with unused tiles inserted so that every chunk starts at a multiple of four."""
    out: list = []
    for k, t in enumerate(tiles):
        prev = tiles[k - 1] if k else None
        same_chunk = prev is not None and prev[1] == 16 and t[1] > 0 and t[3] == prev[3] and t[0] == prev[0] + 16 and t[2] == prev[2] + 16
        if not same_chunk:
            while len(out) % 4:
                out.append((0, 0, 0, 0))
        out.append(tuple(t))
    while len(out) % 4:
        out.append((0, 0, 0, 0))
    return out


def sample_rows(logits: torch.Tensor, rows: torch.Tensor, temperature: torch.Tensor, top_k: torch.Tensor, top_p: torch.Tensor, out: torch.Tensor,
                uniform: Optional[torch.Tensor] = None) -> None:
    """Temperature / top-k / top-p sampling of rows ``rows`` (int64, each < ``logits.shape[0]``)
    of bf16 ``logits`` ``[S, V]`` without a sort; ``out[rows[r]]`` receives the token id.  One
    parameter per sampled row (fp32 / int32 / fp32 device tensors); ``uniform``: one variate in
    [0, 1) per row (fp32; a seeded request's own), else drawn from the device generator."""
    u = uniform if uniform is not None else torch.rand(rows.numel(), dtype=torch.float32, device=logits.device)
    if u.dtype != torch.float32 or u.numel() != rows.numel() or u.device != logits.device:
        raise ValueError("sample_rows: one fp32 uniform variate per row, on the logits' device")
    native().sample_rows(logits, rows, temperature, top_k, top_p, u.contiguous(), out)


def guided_mask(logits: torch.Tensor, rows: torch.Tensor, slots: torch.Tensor, init: torch.Tensor, state: torch.Tensor, cls: torch.Tensor, vc: int,
                script: Optional[torch.Tensor] = None, cursor: Optional[torch.Tensor] = None, cur_init: Optional[torch.Tensor] = None) -> None:
    """Guided JSON decoding, step 1 (``csrc/guided.hip``): every token of ``logits[rows[r]]`` whose
    class the automaton of ``state[slots[r]]`` does not allow becomes -inf; ``init[r][0] >= 0``
    (re)sets that automaton first.  ``script`` int32 ``[S, 2, L]`` / ``cursor`` int32 ``[S]`` /
    ``cur_init`` int32 ``[R]``: skeleton scripts of schema-guided sequences (forced literal tokens)."""
    native().guided_mask(logits, rows, slots, init, state, cls, vc, script, cursor, cur_init)


def guided_advance(out_ids: torch.Tensor, rows: torch.Tensor, slots: torch.Tensor, state: torch.Tensor, cls: torch.Tensor, script: Optional[torch.Tensor] = None,
                   cursor: Optional[torch.Tensor] = None) -> None:
    """Guided JSON decoding, step 2: feed ``out_ids[rows[r]]`` to the automaton (and script cursor) of ``state[slots[r]]``."""
    native().guided_advance(out_ids, rows, slots, state, cls, script, cursor)


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    native().argmax(logits, out)
    return out
